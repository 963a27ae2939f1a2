"""The library's host-libm arbitration arithmetic (srla_amd/csrc/host_ties.cpp: order selection, LTP taps, the SVR refinement,
Levinson-Durbin) against the oracle, bit for bit, on the oracle's own inputs -- on the CPU, through the test hooks of
include/srla_mi355x.h.  On the GPU the same code only ever runs for flagged near-ties (tests/test_gpu_ties.py falsifies device
decisions to get there); here every call is compared."""
import ctypes as C

import numpy as np
import pytest

import helpers

DP = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def hooks():
    lib = C.CDLL(helpers.PRODUCT_SO)
    lib.SRLAMI355X_TestSelectOrder.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_uint32, C.c_uint32]
    lib.SRLAMI355X_TestSelectOrder.restype = C.c_uint32
    lib.SRLAMI355X_TestLtpTaps.argtypes = [C.c_void_p, C.c_uint32]
    lib.SRLAMI355X_TestLtpTaps.restype = C.c_uint32
    lib.SRLAMI355X_TestSvrRefine.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.SRLAMI355X_TestLevinson.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    return lib


def _signal(kind, seed, n, bps=16):
    pcm = helpers.synth(kind, seed, 48000, 1, n, bps)[0]
    return pcm.astype(np.float64) * 2.0 ** -(bps - 1)


def _lags(sig, num):
    o = helpers.Oracle(1, preset=4, max_block=max(4096, 1 << int(np.ceil(np.log2(sig.size)))), divisions=0)
    lags = np.zeros(num, dtype=np.float64)
    o.lib.oracle_autocorr(o.h, _p(sig), sig.size, _p(lags), num)
    return lags


CASES = [(helpers.MUSIC, 1, 4096), (helpers.VARIED, 2, 4096), (helpers.NOISE, 3, 2048), (helpers.SINE, 4, 1024), (helpers.MUSIC, 5, 1000),
         (helpers.VARIED, 6, 8192), (helpers.MUSIC, 7, 513)]


@pytest.mark.parametrize("kind,seed,n", CASES)
@pytest.mark.parametrize("order", [8, 16, 32, 64])
def test_levinson_and_order_selection_equal_the_oracle(hooks, kind, seed, n, order):
    sig = _signal(kind, seed, n)
    lags = _lags(sig, order + 1)
    lags[0] *= 1.0 + 1e-5                                                 # the ridge, lpc.c:466
    coefs = np.zeros((order, order), dtype=np.float64)
    errs = np.zeros(order + 1, dtype=np.float64)
    olib = helpers.oracle_lib()
    olib.oracle_levinson(_p(lags), order, n, _p(coefs), _p(errs))
    mine = np.zeros(order, dtype=np.float64)
    hooks.SRLAMI355X_TestLevinson(_p(lags), order, _p(mine))
    assert np.array_equal(mine.view(np.uint64), coefs[order - 1].view(np.uint64))
    for bps in (8, 16, 24):
        lens = np.zeros(order + 1, dtype=np.float64)
        want = olib.oracle_select_order(_p(errs), order, n, bps, _p(lens))
        # (the oracle's error variances are window compensated already: compensation 1.0 is an exact product)
        assert hooks.SRLAMI355X_TestSelectOrder(_p(errs), order, 1.0, n, bps) == want


@pytest.mark.parametrize("kind,seed,n", CASES[:5])
@pytest.mark.parametrize("order,iters", [(8, 1), (16, 3), (32, 2), (64, 1)])
def test_svr_refinement_equals_the_oracle(hooks, kind, seed, n, order, iters):
    sig = _signal(kind, seed, n)
    lags = _lags(sig, order + 1)
    lags[0] *= 1.0 + 1e-5
    start = np.zeros(order, dtype=np.float64)
    hooks.SRLAMI355X_TestLevinson(_p(lags), order, _p(start))
    want = start.copy()
    assert helpers.oracle_lib().oracle_svr_refine(_p(sig), n, _p(want), order, iters) == 0
    got = start.copy()
    hooks.SRLAMI355X_TestSvrRefine(_p(sig), n, _p(got), order, iters)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def _pack(coef):
    q = []
    for c in coef:
        d = c * 32.0
        v = int(np.floor(d + 0.5)) if d >= 0.0 else -int(np.floor(-d + 0.5))
        q.append(max(-32, min(31, v)))
    q = q[::-1] + [0] * (3 - len(q))
    return (q[0] & 63) | ((q[1] & 63) << 6) | ((q[2] & 63) << 12)


@pytest.mark.parametrize("ltp_order", [1, 3])
def test_ltp_taps_equal_the_oracle(hooks, ltp_order):
    found = 0
    for kind, seed, n in [(helpers.MUSIC, s, 4096) for s in range(10, 30)] + [(helpers.SINE, 31, 2048), (helpers.VARIED, 32, 8192)]:
        sig = _signal(kind, seed, n)
        o = helpers.Oracle(1, preset=4, max_block=8192, divisions=0, ltp_order=ltp_order)
        coef = np.zeros(3, dtype=np.float64)
        period = C.c_uint32(0)
        if o.lib.oracle_ltp_coefficients(o.h, _p(sig), n, ltp_order, _p(coef), C.byref(period)) != 0:
            continue
        found += 1
        r = _lags(sig, 263 + 2)
        p = period.value
        td = np.array([r[0], r[1], r[2], r[p - 1], r[p], r[p + 1]], dtype=np.float64)
        assert hooks.SRLAMI355X_TestLtpTaps(_p(td), ltp_order) == _pack(coef[:ltp_order]), (kind, seed, p)
    assert found >= 5
