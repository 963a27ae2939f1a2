#!/usr/bin/env python3
"""Where the wavefronts of srla_autocorr / srla_residual_cost spend their time in flight (a -DSRLA_DIAG_PHASES build):

    make -C srla_amd/csrc BUILD=build_phases OUT=$PWD/ab/libphases.so EXTRA=-DSRLA_DIAG_PHASES $PWD/ab/libphases.so
    SRLA_PRODUCT_SO=ab/libphases.so python tools/phase_probe.py [seconds] [V] [P] [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
from srla_amd import capi  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 174.8
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1
P = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
n = int(seconds * 48000); n -= n % 2
lib = capi.EncoderLib(helpers.PRODUCT_SO)
L = lib.lib
L.SRLAMI355X_EncodeWholeDevice.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
pcm = helpers.synth(helpers.MUSIC, 1000, 48000, 2, n)
cfg, par = capi.cli_setup(2, 16, 48000, preset=4, max_block=B, divisions=V, ltp_order=P)
enc = lib.create(cfg); assert lib.set_parameter(enc, par) == 0
cap = 4 * pcm.size + 4096
sz = C.c_uint32(0)
d = torch.from_numpy(pcm).cuda()
out = torch.empty(cap, dtype=torch.uint8).pin_memory()
run = lambda: L.SRLAMI355X_EncodeWholeDevice(enc, C.c_void_p(d.data_ptr()), n, n, C.c_void_p(out.data_ptr()), cap, C.byref(sz), None)
tab = (C.c_ulonglong * 48)()


def read_tables():
    for k, name in enumerate(("autocorr", "residual_cost", "pack")):
        part = (C.c_ulonglong * 16)()
        assert getattr(L, "SRLAMI355X_DiagPhases_" + name)(part) == 0
        for p in range(16):
            tab[16 * k + p] = part[p]


assert run() == 0
torch.cuda.synchronize()
read_tables()                                   # discard the warm-up
for _ in range(3):
    assert run() == 0
torch.cuda.synchronize()
read_tables()
names = [["loads landed", "tap sums + tap", "pre-emphasis + window", "first forward stage", "forward stages 2..", "spectrum pass", "first inverse stage",
          "inverse stages 2..", "lag stores", "item record fetched"],
         ["loads landed", "pre-emphasis, planes, taps", "LTP", "FIR + residual", "partition means", "residual store", "Rice parameters", "code bits + reductions", "arg-min + record"],
         ["record fetched, words zeroed", "payload header", "parameters + residuals fetched", "pass 1 + prefix sum", "pass 2: emit", "Fletcher-16", "store"]]
for k, kn in enumerate(("srla_autocorr (all classes)", "srla_residual_cost (fast path)", "srla_pack_blocks")):
    row = [tab[16 * k + p] for p in range(16)]
    tot = float(sum(row)) or 1.0
    print("%s: %.3g wave-ticks" % (kn, tot))
    for p, v in enumerate(row):
        if v:
            print("   %-28s %5.1f %%" % (names[k][p] if p < len(names[k]) else "phase %d" % p, 100.0 * v / tot))
lib.destroy(enc)
