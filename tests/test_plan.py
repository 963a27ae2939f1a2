"""The job plan of a call (srla_amd/csrc/host_plan.cpp: plan_jobs) through the library's test hook -- no device needed.

What the pipeline relies on and a byte comparison cannot see (a violation is a race, not a wrong byte every time):
every stream's regular windows are covered exactly once and in order, segments start on window boundaries and on multiples
of 16 samples of the job's planes, and a buffer set is not taken again before its last job has been collected (the loop in
host_pipeline.cpp collects job k in iteration k + 4: five jobs apart)."""
import ctypes as C
import random

import numpy as np
import pytest

import helpers
from srla_amd import capi

REUSE_DISTANCE = 5          # host_pipeline.cpp: lag = depth + run-ahead = 4, so a set is free again five jobs later
CHAIN_SETS_FROM = 11        # host_impl.h: kMaxSlots - 3 (seed, search, encode of chain mode)


def _plan(lib, enc, lengths, device_input=False):
    fn = lib.lib.SRLAMI355X_TestPlanJobs
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32), C.c_uint32]
    arr = (C.c_uint32 * len(lengths))(*lengths)
    out = (C.c_uint32 * 65536)()
    n = fn(enc, len(lengths), arr, int(device_input), out, 65536)
    assert n >= 0
    jobs, w = [], 0
    while w < n:
        slot, nseg, total = out[w], out[w + 1], out[w + 2]
        w += 3
        segs = [tuple(out[w + 4 * k: w + 4 * k + 4]) for k in range(nseg)]
        w += 4 * nseg
        jobs.append((slot, total, segs))
    return jobs


def _check(jobs, lengths, window_len, tails, reuse_distance=REUSE_DISTANCE):
    pos = [0] * len(lengths)
    last_use = {}
    for j, (slot, total, segs) in enumerate(jobs):
        assert slot < CHAIN_SETS_FROM, (j, slot)
        if slot in last_use:
            assert j - last_use[slot] >= reuse_distance, "buffer set %d taken by jobs %d and %d" % (slot, last_use[slot], j)
        last_use[slot] = j
        assert segs and total % 16 == 0
        end = 0
        for stream, s0, ns, base in segs:
            assert ns > 0 and s0 == pos[stream], (j, stream, s0, pos[stream])          # in order, no gap, no overlap
            assert s0 % window_len == 0                                                   # whole windows up to the stream's end
            assert base % 16 == 0 and base >= end
            end = base + ns
            pos[stream] += ns
            body = lengths[stream] - tails[stream]
            assert pos[stream] == body or ns % window_len == 0
        assert end <= total
    for i, n in enumerate(lengths):
        assert pos[i] == n - tails[i], (i, pos[i], n, tails[i])


CASES = [dict(preset=4, max_block=4096, divisions=1), dict(preset=4, max_block=4096, divisions=2, ltp_order=3),
         dict(preset=2, max_block=8192, divisions=2), dict(preset=4, max_block=4096, divisions=0), dict(preset=4, max_block=1024, divisions=3)]


def _tail(n, window_len, grid, ltp):
    """host_pipeline.cpp: chain_tail -- the history-dependent last window stays out of the plan"""
    tn = n % window_len
    if (tn & 1) and (grid & 1) == 0:
        return tn
    if tn > 0 and ltp > 0 and grid > 256 and ((tn - 1) % grid) + 1 <= 256:
        return tn
    return 0


@pytest.mark.parametrize("cli", CASES, ids=lambda c: "B%d_V%d_P%d" % (c["max_block"], c["divisions"], c.get("ltp_order", 0)))
def test_every_plan_covers_its_streams_once_and_keeps_the_buffer_sets_apart(cli):
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    cfg, par = capi.cli_setup(2, 16, 48000, **cli)
    enc = lib.create(cfg)
    assert lib.set_parameter(enc, par) == capi.OK
    search = par.min_num_samples_per_block != par.max_num_samples_per_block
    window_len = par.num_lookahead_samples if search else par.max_num_samples_per_block
    grid = par.min_num_samples_per_block if search else par.max_num_samples_per_block
    rnd = random.Random(20261001)
    seconds = [0.01, 0.5, 3, 10, 30, 40, 60, 87.3, 90, 120, 150, 174.8, 200, 250, 300, 437.1, 600, 1800]
    singles = [int(s * 48000) for s in seconds] + [rnd.randrange(1, 40_000_000) for _ in range(150)]
    seen_shapes = set()
    for n in singles:
        for dev in (False, True):
            jobs = _plan(lib, enc, [n], dev)
            tails = [_tail(n, window_len, grid, par.ltp_order)]
            if n - tails[0] == 0:
                assert jobs == []
                continue
            _check(jobs, [n], window_len, tails)
            seen_shapes.add((len(jobs), dev))
    assert len({k for k, _ in seen_shapes}) >= 6                       # one job, pieces, whole jobs + tail jobs ...
    for _ in range(60):
        lengths = [rnd.choice([rnd.randrange(1, 400_000), rnd.randrange(1, 25_000_000)]) for _ in range(rnd.randrange(2, 40))]
        tails = [_tail(n, window_len, grid, par.ltp_order) for n in lengths]
        _check(_plan(lib, enc, lengths), lengths, window_len, tails)
    lib.destroy(enc)


@pytest.mark.parametrize("slots", [6, 7, 9])
def test_more_rotating_sets_keep_every_set_apart_by_the_deeper_pipeline(slots, monkeypatch):
    """SRLA_MI355X_SLOTS = 6 .. 9: the host runs further ahead (host_pipeline.cpp: lag = depth + min(8, kSlots - 1 - depth), a job
    is collected lag iterations after it was begun), so a buffer set must not come round again before kSlots jobs have passed --
    for the remainder jobs of a call of many streams too (round 5 gave them five sets of their own whatever kSlots was: with six
    rotating sets remainder job r + 5 restaged the set of job r while r was still in flight; found by the advisor)."""
    monkeypatch.setenv("SRLA_MI355X_SLOTS", str(slots))
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    rnd = random.Random(slots)
    for cli in (CASES[0], CASES[1]):
        cfg, par = capi.cli_setup(2, 16, 48000, **cli)
        enc = lib.create(cfg)
        assert lib.set_parameter(enc, par) == capi.OK
        window_len, grid = par.num_lookahead_samples, par.min_num_samples_per_block
        # many streams with remainders: whole jobs first, then at least a dozen remainder jobs
        for _ in range(12):
            lengths = [(4 << 20) * rnd.randrange(0, 3) + rnd.randrange(2_200_000, 4_000_000) for _ in range(rnd.randrange(14, 40))]
            tails = [_tail(n, window_len, grid, par.ltp_order) for n in lengths]
            jobs = _plan(lib, enc, lengths)
            assert sum(1 for _, _, segs in jobs if segs[0][2] != 4 << 20) >= 7      # remainder jobs: more than the five sets round 5 gave them
            _check(jobs, lengths, window_len, tails, reuse_distance=slots)
        for n in (28_800_000, 2_880_000, 9_600_000, 86_400_000):
            _check(_plan(lib, enc, [n]), [n], window_len, [_tail(n, window_len, grid, par.ltp_order)], reuse_distance=min(slots, 5))
        lib.destroy(enc)


def test_the_metric_stream_is_planned_as_documented():
    """600 s of stereo at the metric configuration: whole jobs of 4 Mi instants in the five rotating buffer sets, the rest cut so that
    the last job is small, both in sets of their own; 60 s: three pieces; 10 s: one job."""
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    cfg, par = capi.cli_setup(2, 16, 48000, preset=4, max_block=4096, divisions=1)
    enc = lib.create(cfg)
    assert lib.set_parameter(enc, par) == capi.OK
    jobs = _plan(lib, enc, [28_800_000])
    assert [s for s, _, _ in jobs] == [0, 1, 2, 3, 4, 0, 5, 6]
    assert [segs[0][2] for _, _, segs in jobs[:6]] == [4 << 20] * 6 and jobs[-1][2][0][2] <= 300_000
    assert len(_plan(lib, enc, [2_880_000])) == 3 and len(_plan(lib, enc, [480_000])) == 1
    # 200 s: two whole jobs' worth in seven pieces -- more than the five rotating sets, so the last piece (a length of its own) takes
    # the first tail job's set and the six equal pieces keep finding their shared table
    jobs = _plan(lib, enc, [9_600_000])
    assert [s for s, _, _ in jobs] == [0, 1, 2, 3, 4, 0, 5] and len({segs[0][2] for _, _, segs in jobs[:6]}) == 1
    lib.destroy(enc)
    # nine 300 s files at config 5's flags: 27 whole jobs in the rotating sets, the four jobs of remainders in sets of their own
    cfg, par = capi.cli_setup(2, 16, 48000, preset=4, max_block=4096, divisions=2, ltp_order=3)
    enc = lib.create(cfg)
    assert lib.set_parameter(enc, par) == capi.OK
    jobs = _plan(lib, enc, [14_400_000] * 9)
    assert [s for s, _, _ in jobs] == [k % 5 for k in range(27)] + [5, 6, 7, 8]
    assert all(len(segs) == 1 for _, _, segs in jobs[:27]) and all(len(segs) > 1 for _, _, segs in jobs[27:])
    lib.destroy(enc)


def test_the_staging_copy_packs_and_flags_what_does_not_fit():
    """host_support.cpp: pack16_or -- every length around the vector width, every destination alignment, the values either side of
    the int16 range (a stream declared 16 bits wide may hold wider samples: the job is then staged as int32)."""
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    fn = lib.lib.SRLAMI355X_TestPack16
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(7)
    for n in list(range(0, 70)) + [255, 256, 257, 4095, 65536 + 33]:
        for shift in (0, 1, 7, 16):
            src = rng.integers(-32768, 32768, size=n, dtype=np.int32)
            buf = np.zeros(n + 64, dtype=np.int16)
            dst = buf[shift:shift + n]
            wide = C.c_uint32(0)
            m = fn(dst.ctypes.data, src.ctypes.data, n, C.byref(wide))
            assert np.array_equal(dst, src.astype(np.int16)) and wide.value == 0
            assert m == (int(np.bitwise_or.reduce(src.view(np.uint32))) if n else 0)
            assert not buf[:shift].any() and not buf[shift + n:].any()
    for bad in (32768, -32769, 1 << 20, -(1 << 31)):
        for at in (0, 5, 31, 32, 100, 999):
            src = rng.integers(-32768, 32768, size=1000, dtype=np.int32)
            src[at] = bad
            dst = np.zeros(1000, dtype=np.int16)
            wide = C.c_uint32(0)
            fn(dst.ctypes.data, src.ctypes.data, 1000, C.byref(wide))
            assert wide.value != 0, (bad, at)
    for ok in (32767, -32768):
        src = np.full(1000, ok, dtype=np.int32)
        dst = np.zeros(1000, dtype=np.int16)
        wide = C.c_uint32(0)
        fn(dst.ctypes.data, src.ctypes.data, 1000, C.byref(wide))
        assert wide.value == 0 and (dst == ok).all()
