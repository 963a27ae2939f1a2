"""History mode: parameters under which blocks ANYWHERE in the stream depend on the reference's earlier
calls -- the long-term predictor with blocks of at most 256 samples (263 lags copied out of a shorter FFT buffer,
lpc.c:371-373) and odd minimum blocks (the Welch window never writes the middle word, lpc.c:260-264).  Golden streams:
the compiled reference in a fresh process per stream (tools/gen_golden_history.py), e.g. `srla -e -B 1024 -V 2 -P 3`,
`-B 512 -V 1 -P 1`, `-B 2048 -V 3 -P 3`, `-B 4095 -V 0`, `-B 1000 -V 3`."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import helpers
from srla_amd import capi

HIST = json.load(open(os.path.join(helpers.GOLDEN, "history_streams.json")))["streams"]
IDS = [c["name"] for c in HIST]


def _input(case):
    sp = case["input"]
    pcm = helpers.synth(sp["kind"], sp["seed"], sp["rate"], sp["nch"], sp["n"], sp["bps"])
    assert helpers.sha256(pcm) == case["input_sha256"], "synthetic input differs from the one the golden was made from"
    return pcm


def _check(case, data):
    assert data.size == case["srl_size"]
    assert helpers.sha256(data) == case["srl_sha256"]
    if "file" in case:
        assert np.array_equal(data, np.fromfile(os.path.join(helpers.GOLDEN, case["file"]), dtype=np.uint8))


# ---------------------------------------------------------------------------------------------- CPU: the oracle is pinned
@pytest.mark.parametrize("case", [c for c in HIST if c["input"]["n"] <= 100000], ids=[c["name"] for c in HIST if c["input"]["n"] <= 100000])
def test_oracle_reproduces_the_reference_in_the_history_regimes(case):
    pcm = _input(case)
    sp = case["input"]
    data = helpers.Oracle(sp["nch"], bits_per_sample=sp["bps"], sampling_rate=sp["rate"], **case["cli"]).encode_whole(pcm)
    _check(case, data)
    assert np.array_equal(helpers.oracle_decode(data), pcm)


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("case", HIST, ids=IDS)
def test_history_golden_streams_from_the_reference(product, case):
    pcm = _input(case)
    sp = case["input"]
    got = product.encode(pcm, bits_per_sample=sp["bps"], sampling_rate=sp["rate"], **case["cli"])
    _check(case, got)


def _stats(product, enc):
    import bench
    st = bench.Stats()
    product.lib.SRLAMI355X_GetStats.argtypes = [C.c_void_p, C.POINTER(bench.Stats), C.c_int]
    product.lib.SRLAMI355X_GetStats(enc, C.byref(st), 0)
    return st


@pytest.mark.gpu
def test_history_mode_is_what_ran_and_it_is_declared_identical(product):
    """The statistics say that the windows went through history mode, and the library does not flag these parameters as
    non-identical (they were, until history mode existed)."""
    cli = dict(preset=4, max_block=1024, divisions=2, ltp_order=3)
    pcm = helpers.synth(helpers.MUSIC, 5, 48000, 2, 20000)
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = product.create(cfg)
    try:
        assert product.set_parameter(enc, par) == capi.OK
        product.lib.SRLAMI355X_NonIdenticalReasons.argtypes = [C.c_void_p, C.c_uint32]
        product.lib.SRLAMI355X_NonIdenticalReasons.restype = C.c_uint32
        assert product.lib.SRLAMI355X_NonIdenticalReasons(enc, 0) == 0
        assert product.lib.SRLAMI355X_NonIdenticalReasons(enc, 20001) == 0
        rc, data = product.encode_whole(enc, pcm)
        assert rc == capi.OK
        st = _stats(product, enc)
        assert st.num_history_windows == (20000 + 4095) // 4096
        assert st.num_nonidentical_calls == 0 and st.nonidentical_reasons == 0
        assert np.array_equal(data, helpers.Oracle(2, **cli).encode_whole(pcm))
    finally:
        product.destroy(enc)


HCLIS = {
    "B1024_V2_P3": dict(preset=4, max_block=1024, divisions=2, ltp_order=3),
    "B512_V1_P1": dict(preset=2, max_block=512, divisions=1, ltp_order=1),
    "B4095_V0": dict(preset=4, max_block=4095, divisions=0),
    "B1000_V3_P1": dict(preset=2, max_block=1000, divisions=3, ltp_order=1),
    "min375_P3": dict(preset=3, max_block=3000, min_block=375, lookahead=6000, ltp_order=3),
    "B2048_V3_L8_P3": dict(preset=1, max_block=2048, divisions=3, lookahead_factor=8, ltp_order=3),
}


@pytest.mark.gpu
@pytest.mark.parametrize("cli_name", sorted(HCLIS))
@pytest.mark.parametrize("kind,nch,bps", [(helpers.MUSIC, 2, 16), (helpers.VARIED, 1, 16), (helpers.NOISE, 3, 24), (helpers.VARIED, 2, 8)])
def test_history_regimes_equal_the_oracle(product, cli_name, kind, nch, bps):
    cli = HCLIS[cli_name]
    for n in (24576, 20001, 4100, 300, 131):
        pcm = helpers.synth(kind, 90 + nch, 48000, nch, n, bps)
        got = product.encode(pcm, bits_per_sample=bps, **cli)
        want = helpers.Oracle(nch, bits_per_sample=bps, **cli).encode_whole(pcm)
        assert np.array_equal(got, want), (cli_name, kind, nch, bps, n)


@pytest.mark.gpu
def test_history_mode_with_silence_offset_shift_and_callback(product):
    """Silent and RAW blocks make no call (the buffer keeps what it held), the offset shift is settled before the first window,
    callbacks arrive once per window, in order."""
    cli = HCLIS["B1024_V2_P3"]
    pcm = helpers.synth(helpers.MUSIC, 12, 48000, 2, 30000)
    pcm[:, 5000:14000] = 0                                     # whole windows and parts of windows of silence
    pcm = (pcm >> 2) << 2                                      # offset left shift 2
    pcm[:, 20000:20100] = np.random.RandomState(1).randint(-32768, 32767, size=(2, 100)) & ~3   # noise: RAW candidates
    want = helpers.Oracle(2, **cli).encode_whole(pcm)
    calls = []
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = product.create(cfg)
    try:
        assert product.set_parameter(enc, par) == capi.OK
        rc, got = product.encode_whole(enc, pcm, callback=lambda total, progress, ptr, size: calls.append((total, progress, size)))
        assert rc == capi.OK
    finally:
        product.destroy(enc)
    assert np.array_equal(got, want)
    assert [c[1] for c in calls] == [min(30000, 4096 * (k + 1)) for k in range(8)]
    assert sum(c[2] for c in calls) == want.size - capi.HEADER_SIZE
    assert np.array_equal(helpers.oracle_decode(got), pcm)


@pytest.mark.gpu
def test_history_mode_streams_of_a_batch_each_start_from_a_fresh_buffer(product):
    """`srla` creates its encoder per file: every stream of a batch is what EncodeWhole writes for it on a fresh handle."""
    cli = HCLIS["B1000_V3_P1"]
    pcms = [helpers.synth(helpers.MUSIC if i % 2 else helpers.VARIED, 200 + i, 48000, 2, n) for i, n in enumerate((9000, 12001, 500, 16000))]
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = product.create(cfg)
    try:
        assert product.set_parameter(enc, par) == capi.OK
        rc, outs, res = capi.encode_batch(product, enc, pcms)
        assert rc == capi.OK and all(r == capi.OK for r in res)
        for pcm, got in zip(pcms, outs):
            assert np.array_equal(got, helpers.Oracle(2, **cli).encode_whole(pcm))
        # PCM frames (a WAV data chunk) through the same regime
        frames = [np.ascontiguousarray(p.T.astype("<i2")).tobytes() for p in pcms]
        rc, outs2, res = capi.encode_batch_pcm(product, enc, frames, [p.shape[1] for p in pcms], 2)
        assert rc == capi.OK
        for a, b in zip(outs, outs2):
            assert np.array_equal(a, b)
        # a buffer that is too small fails that stream alone
        caps = [2 * p.size * 4 + 1024 for p in pcms]
        caps[1] = 2000
        rc, outs3, res = capi.encode_batch(product, enc, pcms, caps=caps)
        assert rc == capi.INSUFFICIENT_BUFFER and res[1] == capi.INSUFFICIENT_BUFFER and outs3[1] is None
        for k in (0, 2, 3):
            assert np.array_equal(outs3[k], outs[k])
    finally:
        product.destroy(enc)


@pytest.mark.gpu
def test_history_mode_from_device_memory(product):
    import torch
    cli = HCLIS["B4095_V0"]
    pcm = helpers.synth(helpers.MUSIC, 33, 48000, 2, 20000)
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = product.create(cfg)
    try:
        assert product.set_parameter(enc, par) == capi.OK
        d = torch.from_numpy(pcm).cuda()
        out = np.zeros(2 * pcm.size * 4 + 1024, dtype=np.uint8)
        size = C.c_uint32(0)
        fn = product.lib.SRLAMI355X_EncodeWholeDevice
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
        rc = fn(enc, C.c_void_p(d.data_ptr()), pcm.shape[1], pcm.shape[1], out.ctypes.data_as(C.c_void_p), out.size, C.byref(size), None)
        assert rc == capi.OK
        assert np.array_equal(out[:size.value], helpers.Oracle(2, **cli).encode_whole(pcm))
    finally:
        product.destroy(enc)


@pytest.mark.gpu
def test_window_ranges_are_refused_in_the_history_regimes(product, capfd):
    """One stream over several GPUs needs independent windows; in the history regimes they are a chain, so the range call says so."""
    cli = HCLIS["B1024_V2_P3"]
    pcm = helpers.synth(helpers.MUSIC, 3, 48000, 2, 8192)
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = product.create(cfg)
    try:
        assert product.set_parameter(enc, par) == capi.OK
        fn = product.lib.SRLAMI355X_EncodeWindows
        fn.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        out = np.zeros(1 << 20, dtype=np.uint8)
        size = C.c_uint32(0)
        rc = fn(enc, capi.planar_ptrs(pcm), 8192, 0, 1, out.ctypes.data_as(C.c_void_p), out.size, C.byref(size))
        assert rc == capi.INVALID_FORMAT
        assert "not independent" in capfd.readouterr().err
    finally:
        product.destroy(enc)


# ---------------------------------------------------------------------- what is NOT bit-identical is named, loudly, and counted
def _reasons(product, enc, n):
    product.lib.SRLAMI355X_NonIdenticalReasons.argtypes = [C.c_void_p, C.c_uint32]
    product.lib.SRLAMI355X_NonIdenticalReasons.restype = C.c_uint32
    return product.lib.SRLAMI355X_NonIdenticalReasons(enc, n)


LTP_TINY_BUFFER = 2


def test_parameters_that_are_not_bit_identical_are_named_at_set_parameter(product, capfd):
    """No GPU needed: SetEncodeParameter itself says what cannot be promised (and stays silent otherwise)."""
    def setup(**cli):
        cfg, par = capi.cli_setup(2, 16, 48000, **cli)
        enc = product.create(cfg)
        assert enc and product.set_parameter(enc, par) == capi.OK
        return enc
    # the baseline configurations and the history regimes: identical, silent
    for cli in (dict(preset=4, max_block=4096, divisions=1), dict(preset=4, max_block=4096, divisions=2, ltp_order=3),
                dict(preset=4, max_block=1024, divisions=2, ltp_order=3), dict(preset=4, max_block=4095, divisions=0)):
        enc = setup(**cli)
        assert _reasons(product, enc, 0) == 0 and _reasons(product, enc, 48001) == 0
        product.destroy(enc)
    assert "WARNING" not in capfd.readouterr().err
    # the long-term predictor on an encoder created for blocks of at most 256 samples (lpc.c:371-373 leaves the buffer)
    enc = setup(preset=4, max_block=256, divisions=0, ltp_order=3)
    assert _reasons(product, enc, 0) == LTP_TINY_BUFFER
    product.destroy(enc)
    err = capfd.readouterr().err
    assert "NOT guaranteed bit-identical" in err and "lpc.c:371-373" in err
    # SVR refinement, also together with history-dependent blocks: identical (the refinement's residual is one more writer of
    # the reference's buffer in chain / history mode), silent
    enc = setup(preset=4, max_block=4096, divisions=1, svr_iterations=2)
    assert _reasons(product, enc, 0) == 0 and _reasons(product, enc, 40000) == 0 and _reasons(product, enc, 40001) == 0
    product.destroy(enc)
    enc = setup(preset=2, max_block=1000, divisions=3, svr_iterations=1)   # odd minimum block: every window
    assert _reasons(product, enc, 0) == 0
    product.destroy(enc)
    assert "WARNING" not in capfd.readouterr().err


@pytest.mark.gpu
def test_calls_that_are_not_bit_identical_are_counted_and_still_lossless(product, capfd):
    pcm_odd = helpers.synth(helpers.MUSIC, 8, 48000, 2, 20001)
    pcm_even = pcm_odd[:, :20000].copy()
    # SVR with an odd-length last window: was counted as not identical until chain mode modelled the refinement's residual
    cli = dict(preset=4, max_block=4096, divisions=1, svr_iterations=1)
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = product.create(cfg)
    try:
        assert product.set_parameter(enc, par) == capi.OK
        for pcm in (pcm_even, pcm_odd):
            rc, data = product.encode_whole(enc, pcm)
            assert rc == capi.OK
            st = _stats(product, enc)
            assert st.num_nonidentical_calls == 0 and st.nonidentical_reasons == 0
            assert np.array_equal(data, helpers.Oracle(2, **cli).encode_whole(pcm))
        assert "WARNING" not in capfd.readouterr().err
    finally:
        product.destroy(enc)
    cfg, par = capi.cli_setup(2, 16, 48000, preset=4, max_block=256, divisions=0, ltp_order=3)
    enc = product.create(cfg)
    try:
        assert product.set_parameter(enc, par) == capi.OK
        rc, data = product.encode_whole(enc, pcm_even)
        assert rc == capi.OK
        st = _stats(product, enc)
        assert st.num_nonidentical_calls == 1 and st.nonidentical_reasons == LTP_TINY_BUFFER
        assert np.array_equal(helpers.oracle_decode(data), pcm_even)
    finally:
        product.destroy(enc)
