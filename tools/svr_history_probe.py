#!/usr/bin/env python3
"""SVR refinement together with history-dependent blocks (odd lengths, short LTP blocks): product vs oracle, byte for byte."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from srla_amd import capi
product = capi.EncoderLib(helpers.PRODUCT_SO)
CASES = [
    ("odd tail, V1", dict(preset=4, max_block=4096, divisions=1, svr_iterations=2), 40001, 2),
    ("odd tail, V2 P3", dict(preset=2, max_block=4096, divisions=2, ltp_order=3, svr_iterations=1), 30001, 2),
    ("odd tail, V0", dict(preset=3, max_block=2048, divisions=0, svr_iterations=3), 10001, 1),
    ("odd min block", dict(preset=2, max_block=1000, divisions=3, svr_iterations=1), 20000, 2),
    ("short LTP blocks", dict(preset=4, max_block=1024, divisions=2, ltp_order=3, svr_iterations=1), 20000, 2),
    ("odd block V0", dict(preset=4, max_block=4095, divisions=0, svr_iterations=2), 20000, 2),
]
bad = 0
for name, cli, n, nch in CASES:
    for kind in (helpers.MUSIC, helpers.VARIED):
        pcm = helpers.synth(kind, 9, 48000, nch, n)
        cfg, par = capi.cli_setup(nch, 16, 48000, **cli)
        enc = product.create(cfg); assert product.set_parameter(enc, par) == 0
        rc, got = product.encode_whole(enc, pcm)
        product.destroy(enc)
        want = helpers.Oracle(nch, **cli).encode_whole(pcm)
        ok = rc == 0 and np.array_equal(got, want)
        first = -1
        if rc == 0 and not ok:
            m = min(got.size, want.size); d = np.nonzero(got[:m] != want[:m])[0]; first = int(d[0]) if d.size else m
        print("%-18s kind %d n %6d: rc %d %s (%d vs %d bytes, first difference at %d)" % (name, kind, n, rc, "identical" if ok else "DIFFERENT", got.size, want.size, first), flush=True)
        bad += not ok
sys.exit(1 if bad else 0)
