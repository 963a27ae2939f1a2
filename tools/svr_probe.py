#!/usr/bin/env python3
"""Bisect the seed-48 case-60 mismatch of tools/gpu_sweep.py on the GPU: variations, library vs oracle."""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import helpers
from srla_amd import capi
lib = capi.EncoderLib(helpers.PRODUCT_SO)

def run(tag, pcm, bps=16, **cli):
    got = lib.encode(pcm, bits_per_sample=bps, **cli)
    want = helpers.Oracle(pcm.shape[0], bits_per_sample=bps, **cli).encode_whole(pcm)
    if np.array_equal(got, want):
        print("ok      ", tag, flush=True); return True
    bg, bw = helpers.list_blocks(got), helpers.list_blocks(want)
    first = next((i for i in range(min(len(bg), len(bw))) if bg[i] != bw[i]), None)
    print("MISMATCH", tag, "sizes", got.size, want.size, "blocks", len(bg), len(bw), "first differing block record", first,
          None if first is None else (bg[first], bw[first]), flush=True)
    return False

pcm = helpers.synth(helpers.SINE, 5000 + 60, 48000, 2, 100000)
base = dict(preset=6, max_block=4608, min_block=1536, lookahead=16896, ltp_order=1, svr_iterations=1)
W = 16896
run("full length 100000", pcm, **base)
run("five whole windows", np.ascontiguousarray(pcm[:, :5 * W]), **base)
run("one window + the tail (15520)", np.ascontiguousarray(pcm[:, :W + 15520]), **base)
tail = np.ascontiguousarray(pcm[:, 5 * W:])
run("the tail alone (15520 = 10 x 1536 + 160)", tail, **base)
run("tail alone, no svr", tail, **dict(base, svr_iterations=0))
run("tail alone, no ltp", tail, **dict(base, ltp_order=0))
run("tail alone, preset 4", tail, **dict(base, preset=4))
run("tail alone, preset 5", tail, **dict(base, preset=5))
run("tail alone, music", helpers.synth(helpers.MUSIC, 5060, 48000, 2, 15520), **base)
run("1536 + 160", np.ascontiguousarray(pcm[:, :1696]), **base)
run("3072 + 160", np.ascontiguousarray(pcm[:, :3232]), **base)
run("160 only", np.ascontiguousarray(pcm[:, :160]), **base)
