"""SRLAMI355X_EncodeBatch: many streams in one call, windows of different streams sharing the device jobs.  Every stream
must be, byte for byte, what SRLAEncoder_EncodeWhole writes for it (= the oracle's stream)."""
import ctypes as C

import numpy as np
import pytest

import helpers
from srla_amd import capi

pytestmark = pytest.mark.gpu

CLIS = {
    "m4_B4096": dict(preset=4, max_block=4096, divisions=1),
    "m4_B4096_V2_P3": dict(preset=4, max_block=4096, divisions=2, ltp_order=3),
    "m2_B2048_V0": dict(preset=2, max_block=2048, divisions=0),
    "m0_B2048": dict(preset=0, max_block=2048, divisions=1),
}


def _encoder(product, nch, bps=16, rate=48000, **cli):
    cfg, par = capi.cli_setup(nch, bps, rate, **cli)
    enc = product.create(cfg)
    assert product.set_parameter(enc, par) == capi.OK
    return enc


def _oracle(pcm, bps=16, **cli):
    return helpers.Oracle(pcm.shape[0], bits_per_sample=bps, **cli).encode_whole(pcm)


@pytest.mark.parametrize("cli_name", sorted(CLIS))
def test_batch_of_mixed_lengths_equals_the_oracle(product, cli_name):
    cli = CLIS[cli_name]
    # even and odd lengths (odd ones end in a chain-mode window), shorter than a window, exactly one window, several windows
    lengths = [48000, 5000, 16384, 4096, 300, 100001, 32768 + 11, 70000, 1, 2049, 65536]
    kinds = [helpers.MUSIC, helpers.VARIED, helpers.NOISE]
    pcms = [helpers.synth(kinds[i % 3], 500 + i, 48000, 2, n) for i, n in enumerate(lengths)]
    enc = _encoder(product, 2, **cli)
    try:
        rc, streams, res = capi.encode_batch(product, enc, pcms)
        assert rc == capi.OK and all(r == capi.OK for r in res)
        for i, pcm in enumerate(pcms):
            assert np.array_equal(streams[i], _oracle(pcm, **cli)), (cli_name, i, lengths[i])
        # the same handle again (cached tables, other slots) and a single stream through the ordinary entry point
        rc, streams2, _ = capi.encode_batch(product, enc, pcms[::-1])
        assert rc == capi.OK
        for a, b in zip(streams2, streams[::-1]):
            assert np.array_equal(a, b)
        rc, one = product.encode_whole(enc, pcms[0])
        assert rc == capi.OK and np.array_equal(one, streams[0])
    finally:
        product.destroy(enc)


def test_batch_spanning_several_jobs(product, monkeypatch):
    """small jobs (SRLA_MI355X_JOB_SAMPLES), so that streams are cut across jobs and many jobs are in flight"""
    monkeypatch.setenv("SRLA_MI355X_JOB_SAMPLES", "65536")
    cli = CLIS["m4_B4096"]
    lengths = [200000, 30000, 131072, 50001, 90000, 16384 * 3]
    pcms = [helpers.synth(helpers.MUSIC, 600 + i, 48000, 2, n) for i, n in enumerate(lengths)]
    enc = _encoder(product, 2, **cli)
    try:
        rc, streams, res = capi.encode_batch(product, enc, pcms)
        assert rc == capi.OK
        for i, pcm in enumerate(pcms):
            assert np.array_equal(streams[i], _oracle(pcm, **cli)), i
    finally:
        product.destroy(enc)


def test_batch_offset_shift_per_stream(product):
    """streams with different offset left shifts in one job (srla_utility.c:177: the shift is a per-stream quantity)"""
    cli = CLIS["m4_B4096"]
    base = [helpers.synth(helpers.MUSIC, 700 + i, 48000, 2, 40000 + 1000 * i, 24) for i in range(4)]
    shifts = [0, 8, 3, 0]
    pcms = [np.ascontiguousarray((p >> s) << s) for p, s in zip(base, shifts)]
    enc = _encoder(product, 2, bps=24, **cli)
    try:
        rc, streams, res = capi.encode_batch(product, enc, pcms)
        assert rc == capi.OK
        for i, pcm in enumerate(pcms):
            want = _oracle(pcm, bps=24, **cli)
            assert np.array_equal(streams[i], want), i
            assert streams[i][24] >= shifts[i]          # header.offset_lshift
    finally:
        product.destroy(enc)


def test_batch_small_buffer_fails_only_its_stream(product):
    cli = CLIS["m4_B4096"]
    pcms = [helpers.synth(helpers.NOISE, 800 + i, 48000, 2, 50000) for i in range(3)]
    enc = _encoder(product, 2, **cli)
    try:
        rc, streams, res = capi.encode_batch(product, enc, pcms, caps=[400000, 20000, 400000])
        assert rc == capi.INSUFFICIENT_BUFFER
        assert res == [capi.OK, capi.INSUFFICIENT_BUFFER, capi.OK]
        for i in (0, 2):
            assert np.array_equal(streams[i], _oracle(pcms[i], **cli))
    finally:
        product.destroy(enc)


def test_batch_pinned_outputs(product):
    """pinned output buffers: the device stores every stream's blocks into its buffer itself"""
    import torch
    cli = CLIS["m4_B4096"]
    lengths = [60000, 33000, 16385, 120000]
    pcms = [helpers.synth(helpers.MUSIC, 900 + i, 48000, 2, n) for i, n in enumerate(lengths)]
    outs_t = [torch.empty(4 * p.size + 4096, dtype=torch.uint8).pin_memory() for p in pcms]
    outs = [t.numpy() for t in outs_t]
    enc = _encoder(product, 2, **cli)
    try:
        call = capi.BatchCall(product, pcms, outs)
        sizes = (C.c_uint32 * len(pcms))()
        assert call.run(enc, sizes) == capi.OK
        for i, pcm in enumerate(pcms):
            assert np.array_equal(outs[i][:sizes[i]], _oracle(pcm, **cli)), i
    finally:
        product.destroy(enc)


def test_batch_argument_errors(product):
    enc = _encoder(product, 2, **CLIS["m4_B4096"])
    try:
        fn = product.lib.SRLAMI355X_EncodeBatch
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        assert fn(enc, 0, None, None, None, None, None, None) == capi.INVALID_ARGUMENT
        assert fn(None, 1, None, None, None, None, None, None) == capi.INVALID_ARGUMENT
    finally:
        product.destroy(enc)


def test_batch_ex_with_pinned_planes_and_given_or_masks(product):
    """SRLAMI355X_EncodeBatchEx: planes in SRLAMI355X_AllocHost memory (read by DMA where they lie) and the OR of every
    stream's samples supplied by the caller -- the library's host threads then touch no sample (what srla_corpus does)."""
    L = product.lib
    L.SRLAMI355X_AllocHost.restype = C.c_void_p
    L.SRLAMI355X_AllocHost.argtypes = [C.c_size_t]
    L.SRLAMI355X_FreeHost.argtypes = [C.c_void_p]
    fn = L.SRLAMI355X_EncodeBatchEx
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    cli = CLIS["m4_B4096_V2_P3"]
    base = [helpers.synth(helpers.MUSIC, 950 + i, 48000, 2, n) for i, n in enumerate((70000, 16384 * 3, 123457, 5000))]
    shifts = [0, 2, 0, 5]
    pcms = [np.ascontiguousarray((p >> s) << s) for p, s in zip(base, shifts)]
    enc = _encoder(product, 2, **cli)
    pinned = []
    try:
        views = []
        for p in pcms:
            ptr = L.SRLAMI355X_AllocHost(p.nbytes)
            assert ptr
            pinned.append(ptr)
            v = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=p.shape)
            v[:] = p
            views.append(v)
        n = len(pcms)
        keep = [capi.planar_ptrs(v) for v in views]
        inputs = (C.c_void_p * n)(*[C.cast(k, C.c_void_p) for k in keep])
        nsmp = (C.c_uint32 * n)(*[p.shape[1] for p in pcms])
        ors = (C.c_uint32 * n)(*[int(np.bitwise_or.reduce(p.astype(np.int64).ravel() & 0xFFFFFFFF)) for p in pcms])
        outs = [np.zeros(4 * p.size + 1024, np.uint8) for p in pcms]
        data = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_uint32 * n)(*[o.size for o in outs])
        sizes = (C.c_uint32 * n)()
        res = (C.c_int * n)()
        assert fn(enc, n, inputs, nsmp, ors, data, caps, sizes, res) == capi.OK
        for i, p in enumerate(pcms):
            assert np.array_equal(outs[i][:sizes[i]], _oracle(p, **cli)), i
    finally:
        product.destroy(enc)
        for ptr in pinned:
            L.SRLAMI355X_FreeHost(ptr)


def _interleave(pcm, bps):
    """planar int32 -> the bytes of a WAV data chunk (8-bit: unsigned + 128; else signed little endian)"""
    nch, n = pcm.shape
    t = np.ascontiguousarray(pcm.T)                     # frames
    if bps == 8:
        return (t + 128).astype(np.uint8).tobytes()
    if bps == 16:
        return t.astype("<i2").tobytes()
    b = t.astype("<i4").view(np.uint8).reshape(n, nch, 4)[:, :, :3]
    return np.ascontiguousarray(b).tobytes()


@pytest.mark.parametrize("how", ["pageable", "pinned", "staged"])
@pytest.mark.parametrize("bps,nch", [(16, 2), (24, 3), (8, 1)])
def test_batch_of_interleaved_pcm_frames(product, monkeypatch, how, bps, nch):
    """SRLAMI355X_EncodeBatchPcm: streams given as WAV data chunks.  The frames are uploaded as they are and de-interleaved,
    widened and OR-reduced on the device (pageable: locked in place for the call; pinned: SRLAMI355X_AllocHost memory; staged:
    SRLA_MI355X_STAGING, de-interleaved on the host).  Odd lengths (chain-mode tails), a stream shorter than a window, and one
    whose first 64 Ki frames suggest an offset shift the whole stream does not have."""
    if how == "staged":
        monkeypatch.setenv("SRLA_MI355X_STAGING", "1")
    L = product.lib
    L.SRLAMI355X_AllocHost.restype = C.c_void_p
    L.SRLAMI355X_AllocHost.argtypes = [C.c_size_t]
    L.SRLAMI355X_FreeHost.argtypes = [C.c_void_p]
    fn = L.SRLAMI355X_EncodeBatchPcm
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    cli = dict(preset=4, max_block=4096, divisions=1) if bps != 16 else CLIS["m4_B4096_V2_P3"]
    pcms = [helpers.synth(helpers.MUSIC if nch == 2 else helpers.VARIED, 970 + i, 48000, nch, n, bps) for i, n in enumerate((150001, 49152, 3000, 200000))]
    if bps > 8:
        pcms[3][:, :100000] = (pcms[3][:, :100000] >> 4) << 4           # the prefix looks shifted, the stream is not
        pcms[1] = np.ascontiguousarray((pcms[1] >> 2) << 2)               # this one really is
    raw = [_interleave(p, bps) for p in pcms]
    enc = _encoder(product, nch, bps=bps, **cli)
    pinned, keep = [], []
    try:
        ptrs = []
        for r in raw:
            if how == "pinned":
                ptr = L.SRLAMI355X_AllocHost(len(r))
                assert ptr
                pinned.append(ptr)
                C.memmove(ptr, r, len(r))
                ptrs.append(ptr)
            else:
                a = np.frombuffer(r, np.uint8).copy()
                keep.append(a)
                ptrs.append(a.ctypes.data)
        n = len(pcms)
        frames = (C.c_void_p * n)(*ptrs)
        nsmp = (C.c_uint32 * n)(*[p.shape[1] for p in pcms])
        outs = [np.zeros(4 * p.size + 1024, np.uint8) for p in pcms]
        data = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_uint32 * n)(*[o.size for o in outs])
        sizes = (C.c_uint32 * n)()
        res = (C.c_int * n)()
        assert fn(enc, n, frames, nsmp, bps // 8, data, caps, sizes, res) == capi.OK
        for i, p in enumerate(pcms):
            assert np.array_equal(outs[i][:sizes[i]], _oracle(p, bps=bps, **cli)), i
        # the container must be the sample format
        assert fn(enc, n, frames, nsmp, (bps // 8) % 3 + 1, data, caps, sizes, res) == capi.INVALID_FORMAT
    finally:
        product.destroy(enc)
        for ptr in pinned:
            L.SRLAMI355X_FreeHost(ptr)
