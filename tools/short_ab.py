#!/usr/bin/env python3
"""Same-box A/B of short-stream call times: one 60 s / 10 s / 120 s stereo stream per SRLAEncoder_EncodeWhole call (pageable host memory
to pageable host memory), median of many calls, a fresh handle per environment setting, settings interleaved.
    python tools/short_ab.py ROUNDS "ENV=.." "ENV=.." ...          (a setting "X=0" is the default build)"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import numpy as np
    import helpers
    from srla_amd import capi
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    cfg, par = capi.cli_setup(2, 16, 48000, preset=4, max_block=4096, divisions=1)
    enc = lib.create(cfg); assert lib.set_parameter(enc, par) == 0
    full = helpers.synth(helpers.MUSIC, 1000, 48000, 2, 120 * 48000)
    out = np.zeros(4 * full.size + 4096, np.uint8)
    res = []
    for secs in (10, 60, 120):
        clip = np.ascontiguousarray(full[:, :secs * 48000]); ptr = capi.planar_ptrs(clip); sz = C.c_uint32(0)
        ts = []
        for k in range(43):
            t0 = time.perf_counter()
            rc = lib.lib.SRLAEncoder_EncodeWhole(enc, ptr, clip.shape[1], out.ctypes.data_as(C.c_void_p), out.size, C.byref(sz), None)
            assert rc == 0
            if k >= 3: ts.append(time.perf_counter() - t0)
        ts.sort()
        res.append("%3d s %7.3f ms %7.0f Msamples/s" % (secs, 1e3 * ts[len(ts) // 2], clip.shape[1] / ts[len(ts) // 2] / 1e6))
    print("   ".join(res), flush=True)
    lib.destroy(enc)
    sys.exit(0)
rounds = int(sys.argv[1])
for r in range(rounds):
    for env in sys.argv[2:]:
        e = dict(os.environ)
        for kv in env.split():
            k, _, v = kv.partition("="); e[k] = v
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=e, capture_output=True, text=True)
        print("%-44s %s" % (env, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "FAILED " + p.stderr[-300:]), flush=True)
