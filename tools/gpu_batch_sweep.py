#!/usr/bin/env python3
"""Randomised parity sweep of the many-streams entry points on an MI355X: SRLAMI355X_EncodeBatch and SRLAMI355X_EncodeBatchPcm
against the oracle, stream by stream, over random formats, parameters, stream counts and lengths (odd ones and streams shorter
than a window included), with small jobs so that streams share jobs and jobs are many.

    python tools/gpu_batch_sweep.py [batches] [seed]"""
import ctypes as C
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
from srla_amd import capi  # noqa: E402


def interleave(pcm, bps):
    nch, n = pcm.shape
    t = np.ascontiguousarray(pcm.T)
    if bps == 8:
        return (t + 128).astype(np.uint8).tobytes()
    if bps == 16:
        return t.astype("<i2").tobytes()
    return np.ascontiguousarray(t.astype("<i4").view(np.uint8).reshape(n, nch, 4)[:, :, :3]).tobytes()


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd = random.Random(seed)
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    L = lib.lib
    pcm_fn = L.SRLAMI355X_EncodeBatchPcm
    pcm_fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    pcm_fn.restype = C.c_int
    bad = done = 0
    for b in range(batches):
        nch = rnd.choice([1, 2, 2, 2, 3])
        bps = rnd.choice([16, 16, 16, 8, 24])
        preset = rnd.choice([1, 2, 4, 4, 4])
        log2b = rnd.choice([10, 11, 12, 12, 13])
        divisions = rnd.choice([0, 1, 1, 2])
        max_block = 1 << log2b
        min_block = max_block >> divisions
        ltp = rnd.choice([0, 0, 1, 3]) if min_block >= 512 else 0
        cli = dict(preset=preset, max_block=max_block, divisions=divisions, ltp_order=ltp)
        os.environ["SRLA_MI355X_JOB_SAMPLES"] = str(rnd.choice([65536, 131072, 1 << 20]))
        count = rnd.randint(1, 12)
        pcms = []
        for i in range(count):
            n = rnd.choice([rnd.randint(1, 3000), rnd.randint(3000, 60000), rnd.randint(60000, 400000)])
            p = helpers.synth(rnd.choice([helpers.MUSIC, helpers.VARIED, helpers.NOISE, helpers.SINE]), 9000 + 100 * b + i, 48000, nch, n, bps)
            if rnd.random() < 0.2 and bps > 8:
                p = np.ascontiguousarray((p >> 3) << 3)
            pcms.append(p)
        wants = [helpers.Oracle(nch, bits_per_sample=bps, **cli).encode_whole(p) for p in pcms]
        cfg, par = capi.cli_setup(nch, bps, 48000, **cli)
        enc = lib.create(cfg)
        assert lib.set_parameter(enc, par) == capi.OK
        try:
            rc, outs, res = capi.encode_batch(lib, enc, pcms)
            ok = rc == capi.OK and all(np.array_equal(o, w) for o, w in zip(outs, wants))
            raw = [np.frombuffer(interleave(p, bps), np.uint8).copy() for p in pcms]
            frames = (C.c_void_p * count)(*[r.ctypes.data for r in raw])
            nsmp = (C.c_uint32 * count)(*[p.shape[1] for p in pcms])
            bufs = [np.zeros(4 * p.size + 1024, np.uint8) for p in pcms]
            data = (C.c_void_p * count)(*[o.ctypes.data for o in bufs])
            caps = (C.c_uint32 * count)(*[o.size for o in bufs])
            sizes = (C.c_uint32 * count)()
            rc2 = pcm_fn(enc, count, frames, nsmp, bps // 8, data, caps, sizes, None)
            ok2 = rc2 == capi.OK and all(np.array_equal(bufs[i][:sizes[i]], wants[i]) for i in range(count))
        finally:
            lib.destroy(enc)
        done += 1
        if not (ok and ok2):
            bad += 1
            print("MISMATCH batch %d (seed %d): planar %s pcm %s  nch=%d bps=%d %s lengths %s" % (b, seed, ok, ok2, nch, bps, cli, [p.shape[1] for p in pcms]), flush=True)
    print("batch sweep: %d batches compared, %d mismatches" % (done, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
