#!/usr/bin/env python3
"""Replays one case of tools/gpu_sweep.py and reports the first block that differs from the oracle's."""
import random
import sys
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers  # noqa: E402
from srla_amd import capi  # noqa: E402


def cases(seed, count):
    rnd = random.Random(seed)
    for case in range(count):
        nch = rnd.choice([1, 2, 2, 2, 3, 5, 8]); bps = rnd.choice([16, 16, 16, 8, 24]); preset = rnd.choice([0, 1, 2, 3, 4, 4, 4, 5, 6])
        log2b = rnd.choice([8, 9, 10, 11, 12, 12, 13]); divisions = rnd.choice([0, 1, 1, 2, 3]); ltp = rnd.choice([0, 0, 0, 1, 3])
        max_block = 1 << log2b; min_block = max_block >> divisions
        if min_block < 64: continue
        if ltp and min_block < 264: ltp = 0
        order = [0, 8, 16, 32, 64, 128, 255][preset]
        if order > min_block: continue
        lookahead_factor = rnd.choice([1, 2, 4]) if divisions else 4
        if (max_block * lookahead_factor) // min_block + 1 > 65: continue
        nblocks = rnd.randint(0, max(2, min(600000 // min_block, 3 * (2 << 20) // min_block // 4)))
        tail = rnd.choice([0, rnd.randint(1, min_block - 1), rnd.randint(1, min_block - 1)])
        n = nblocks * min_block + tail
        if n * nch > 1_500_000: n = (1_500_000 // nch // min_block) * min_block
        if n == 0: continue
        kind = rnd.choice([helpers.MUSIC, helpers.VARIED, helpers.VARIED, helpers.NOISE, helpers.SINE])
        cli = dict(preset=preset, max_block=max_block, divisions=divisions, ltp_order=ltp, lookahead_factor=lookahead_factor)
        shifted = rnd.random() < 0.15
        yield case, nch, bps, n, kind, cli, shifted


def main():
    seed, want_case = int(sys.argv[1]), int(sys.argv[2])
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    for case, nch, bps, n, kind, cli, shifted in cases(seed, want_case + 1):
        if case != want_case: continue
        pcm = helpers.synth(kind, 5000 + case, 48000, nch, n, bps)
        if shifted: pcm = (pcm >> 3) << 3
        print("case", case, nch, bps, n, kind, cli, "shifted", shifted)
        got = lib.encode(pcm, bits_per_sample=bps, **cli)
        want = helpers.Oracle(nch, bits_per_sample=bps, **cli).encode_whole(pcm)
        print("sizes", got.size, want.size, "header equal", np.array_equal(got[:30], want[:30]), list(got[:30]), list(want[:30]))
        bg, bw = helpers.list_blocks(got), helpers.list_blocks(want)
        print("blocks", len(bg), len(bw))
        og = ow = 30; pos = 0
        for i, (a, b) in enumerate(zip(bg, bw)):
            same = a == b and np.array_equal(got[og:og + a[2]], want[ow:ow + b[2]])
            if not same:
                print("first difference: block", i, "at sample", pos, "got", a, "want", b)
                break
            og += a[2]; ow += b[2]; pos += a[1]
        else:
            print("all common blocks equal")


if __name__ == "__main__":
    main()
