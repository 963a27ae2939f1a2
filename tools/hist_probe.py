#!/usr/bin/env python3
"""Bisect a history-mode mismatch on the GPU: variations of one sweep case, library vs oracle, first differing block."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import helpers
from srla_amd import capi
lib = capi.EncoderLib(helpers.PRODUCT_SO)

def run(tag, pcm, bps=16, dump=False, **cli):
    got = lib.encode(pcm, bits_per_sample=bps, **cli)
    want = helpers.Oracle(pcm.shape[0], bits_per_sample=bps, **cli).encode_whole(pcm)
    if np.array_equal(got, want):
        print("ok      ", tag, flush=True); return True
    bg, bw = helpers.list_blocks(got) if got.size == want.size else None, helpers.list_blocks(want)
    first = None
    if bg is not None:
        off = 30; pos = 0
        for i, (t, ns, nb) in enumerate(bw):
            if i >= len(bg) or bg[i] != bw[i] or not np.array_equal(got[off:off + nb], want[off:off + nb]):
                first = (i, bw[i], "at sample", pos)
                if dump:
                    print("  want", bytes(want[off:off + nb]).hex()); print("  got ", bytes(got[off:off + nb]).hex())
                break
            off += nb; pos += ns
    nd = int((got[:min(got.size, want.size)] != want[:min(got.size, want.size)]).sum())
    print("MISMATCH", tag, "sizes", got.size, want.size, "first differing block", first, "of", len(bw), "; differing bytes", nd, flush=True)
    return False

base = dict(preset=4, max_block=1000, min_block=125, lookahead=1375, ltp_order=3)
pcm0 = helpers.synth(helpers.VARIED, 5000 + 37, 48000, 2, 394375)
sh = np.ascontiguousarray((pcm0 >> 3) << 3)
a = 143000
run("section 3 only, from 143000, 11000 samples", np.ascontiguousarray(sh[:, a:a + 11000]), dump=True, **base)
run("from 144375 (window 105 first), 2750 samples", np.ascontiguousarray(sh[:, 144375:144375 + 2750]), dump=True, **base)
run("from 144000, 4125", np.ascontiguousarray(sh[:, 144000:144000 + 4125]), dump=True, **base)
run("from 137500, 8250", np.ascontiguousarray(sh[:, 137500:137500 + 8250]), **base)
run("n=150000", np.ascontiguousarray(sh[:, :150000]), dump=True, **base)
run("ltp 1", np.ascontiguousarray(sh[:, :150000]), **dict(base, ltp_order=1))
run("preset 0", np.ascontiguousarray(sh[:, :150000]), **dict(base, preset=0))
