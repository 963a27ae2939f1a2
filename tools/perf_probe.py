#!/usr/bin/env python3
"""Timing probe: one stream, device-resident or host input, with the library's per-job timeline.

    SRLA_MI355X_TIMELINE=1 SRLA_MI355X_TIMING_STRIDE=1 python tools/perf_probe.py [seconds] [device|host|pinned] [reps] [V] [P] [B]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
from srla_amd import capi  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600
mode = sys.argv[2] if len(sys.argv) > 2 else "device"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
V = int(sys.argv[4]) if len(sys.argv) > 4 else 1
P = int(sys.argv[5]) if len(sys.argv) > 5 else 0
B = int(sys.argv[6]) if len(sys.argv) > 6 else 4096
n = int(seconds * 48000); n -= n % 2
lib = capi.EncoderLib(helpers.PRODUCT_SO)
L = lib.lib
L.SRLAMI355X_EncodeWholeDevice.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
pcm = helpers.synth(helpers.MUSIC, 1000, 48000, 2, n)
cfg, par = capi.cli_setup(2, 16, 48000, preset=4, max_block=B, divisions=V, ltp_order=P)
enc = lib.create(cfg); assert lib.set_parameter(enc, par) == 0
cap = 4 * pcm.size + 4096
sz = C.c_uint32(0)
if mode == "device":
    d = torch.from_numpy(pcm).cuda()
    out = torch.empty(cap, dtype=torch.uint8).pin_memory()
    run = lambda: L.SRLAMI355X_EncodeWholeDevice(enc, C.c_void_p(d.data_ptr()), n, n, C.c_void_p(out.data_ptr()), cap, C.byref(sz), None)
else:
    src = torch.from_numpy(pcm).pin_memory().numpy() if mode == "pinned" else pcm
    out = (torch.empty(cap, dtype=torch.uint8).pin_memory().numpy() if mode == "pinned" else np.zeros(cap, np.uint8))
    planes = capi.planar_ptrs(src)
    run = lambda: L.SRLAEncoder_EncodeWhole(enc, planes, n, out.ctypes.data_as(C.c_void_p), cap, C.byref(sz), None)
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = run()
    dt = time.perf_counter() - t0
    assert rc == 0, rc
    print("rep %d: %.3f ms  %.0f Msamples/s  (%d bytes)" % (r, 1e3 * dt, n / dt / 1e6, sz.value), flush=True)
lib.destroy(enc)
