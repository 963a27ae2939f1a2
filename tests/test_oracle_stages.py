"""Stage-level checks of the oracle: vectors produced by the compiled reference's exported functions
(tests/golden/stages.npz) and the reference tests' own known answers (tests/golden/kats.json)."""
import ctypes as C

import numpy as np
import pytest

import helpers
from tools_shared import KATS, stages

lib = helpers.oracle_lib()
ST = stages()


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64, 256, 1024, 2048, 4096, 8192])
def test_real_fft_bit_exact(n):
    x = ST["fft_in_%d" % n].copy()
    w = np.zeros(n)
    lib.oracle_fft_real(n, -1, _vp(x), _vp(w))
    assert x.tobytes() == ST["fft_fwd_%d" % n].tobytes()
    lib.oracle_fft_real(n, 1, _vp(x), _vp(w))
    assert x.tobytes() == ST["fft_inv_%d" % n].tobytes()


def test_fft_against_naive_dft():
    # test/fft/main.cpp:40 -- n = 32, tolerance 1e-8
    n = 32
    rng = np.random.RandomState(0)
    x = rng.uniform(-1, 1, n)
    y = x.copy(); w = np.zeros(n)
    lib.oracle_fft_real(n, -1, _vp(y), _vp(w))
    k = np.arange(n)
    dft = np.array([np.sum(x * np.exp(-2j * np.pi * k * f / n)) for f in range(n // 2 + 1)])
    assert abs(y[0] - dft[0].real) < 1e-8 and abs(y[1] - dft[n // 2].real) < 1e-8
    for f in range(1, n // 2):
        assert abs(y[2 * f] - dft[f].real) < 1e-8 and abs(y[2 * f + 1] - dft[f].imag) < 1e-8
    lib.oracle_fft_real(n, 1, _vp(y), _vp(w))
    assert np.max(np.abs(y * (2.0 / n) - x)) < 1e-8


@pytest.mark.parametrize("case", KATS["lpc_cases"], ids=lambda c: "n%d_p%d" % (c["n"], c["order"]))
def test_all_order_lpc_bit_exact(case):
    n, order, idx = case["n"], case["order"], case["index"]
    sig = helpers.synth(case["kind"], case["seed"], 48000, 1, n)[0].astype(np.float64) * 2.0 ** -15
    o = helpers.Oracle(1, max_block=8192, divisions=0, preset=6)
    lags = np.zeros(order + 1)
    lib.oracle_autocorr(o.h, _vp(sig), n, _vp(lags), order + 1)
    lags[0] *= (1.0 + 1e-5)
    rows = np.zeros((order, order)); ev = np.zeros(order + 1)
    lib.oracle_levinson(_vp(lags), order, n, _vp(rows), _vp(ev))
    assert ev.tobytes() == ST["lpc_errvars_%d" % idx].tobytes()
    want = ST["lpc_rows_%d" % idx]
    for k in range(order):
        assert rows[k, :k + 1].tobytes() == want[k, :k + 1].tobytes()
    for q in (1, 2, order // 2, order):
        g = ST["lpc_q_%d_%d" % (idx, q)]
        ic = np.zeros(q, dtype=np.int32); rs = C.c_uint32(0)
        row = np.ascontiguousarray(want[q - 1, :q])
        lib.oracle_quantize(_vp(row), q, _vp(ic), C.byref(rs))
        assert rs.value == g[0] and np.array_equal(ic, g[1:])


@pytest.mark.parametrize("case", KATS["ltp_cases"], ids=lambda c: "ltp%d" % c["index"])
def test_ltp_coefficients(case):
    n = case["n"]
    sig = helpers.synth(case["kind"], case["seed"], 48000, 1, n)[0].astype(np.float64) * 2.0 ** -15
    o = helpers.Oracle(1, max_block=8192, divisions=0, preset=4)
    coef = np.zeros(3); period = C.c_uint32(0)
    rc = lib.oracle_ltp_coefficients(o.h, _vp(sig), n, 3, _vp(coef), C.byref(period))
    if case["rc"] == 0:
        assert rc == 0 and period.value == case["period"]
        assert coef.tobytes() == ST["ltp_coef_%d" % case["index"]].tobytes()
    else:
        assert rc == 1


def test_ltp_finds_sine_periods():
    # test/lpc/main.cpp:232-262: periods 10, 20, ..., 190 on a 2048-sample sine
    o = helpers.Oracle(1, max_block=8192, divisions=0, preset=4)
    for k in KATS["ltp_sine_periods"]:
        sig = np.ascontiguousarray(ST["ltp_sine_in_%d" % k["period"]])
        coef = np.zeros(3); period = C.c_uint32(0)
        rc = lib.oracle_ltp_coefficients(o.h, _vp(sig), sig.size, 3, _vp(coef), C.byref(period))
        assert (rc == 0) == (k["rc"] == 0)
        if rc == 0:
            assert period.value == k["detected"] == k["period"]


@pytest.mark.parametrize("case", KATS["code_cases"], ids=lambda c: "code%d" % c["index"])
def test_residual_code_length(case):
    res = np.ascontiguousarray(ST["code_in_%d" % case["index"]])
    t = C.c_uint32(0); p = C.c_uint32(0); b = C.c_uint32(0)
    lib.oracle_residual_code_search(_vp(res), res.size, C.byref(t), C.byref(p), C.byref(b))
    assert b.value == case["bits"]


def test_fletcher16_known_answers():
    for k in KATS["fletcher16"]:
        data = k["text"].encode()
        assert lib.oracle_fletcher16(data, len(data)) == k["value"]


@pytest.mark.parametrize("g", KATS["dijkstra"], ids=lambda g: "nodes%d" % g["num_nodes"])
def test_dijkstra_known_answers(g):
    n = g["num_nodes"]
    adj = np.full((n, n), float(1 << 24))
    for i, j, w in g["edges"]:
        adj[i, j] = w
    path = np.zeros(n, dtype=np.uint32); cost = C.c_double(0)
    assert lib.oracle_dijkstra(_vp(adj), n, g["start"], g["goal"], C.byref(cost), _vp(path)) == 0
    assert cost.value == g["min_cost"]
    route = [g["goal"]]
    while route[-1] != g["start"]:
        route.append(int(path[route[-1]]))
    assert route[::-1] == g["route"]


def test_rice_parameter_is_monotone_step_function():
    # the device looks the Rice parameter up through thresholds; that needs monotonicity
    means = np.sort(np.concatenate([np.linspace(0, 4, 4001), np.logspace(0, 9.5, 4000)]))
    ks = [lib.oracle_rice_k(float(m)) for m in means]
    assert all(b >= a for a, b in zip(ks, ks[1:]))
    assert ks[0] == 0 and max(ks) >= 30
