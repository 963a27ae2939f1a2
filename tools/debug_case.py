#!/usr/bin/env python3
"""Replays one case of tools/gpu_sweep.py and reports the first block that differs from the oracle's.

    python tools/debug_case.py <seed> <case> [max_samples]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers  # noqa: E402
import gpu_sweep  # noqa: E402
from srla_amd import capi  # noqa: E402


def main():
    seed, want_case = int(sys.argv[1]), int(sys.argv[2])
    max_samples = int(sys.argv[3]) if len(sys.argv) > 3 else 1_500_000
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    for case, nch, bps, n, kind, cli, shifted in gpu_sweep.cases(want_case + 1, seed, max_samples):
        if case != want_case:
            continue
        pcm = gpu_sweep.make_pcm(case, nch, bps, n, kind, shifted)
        print("case", case, nch, bps, n, kind, cli, "shifted", shifted)
        got = lib.encode(pcm, bits_per_sample=bps, **cli)
        want = helpers.Oracle(nch, bits_per_sample=bps, **cli).encode_whole(pcm)
        print("sizes", got.size, want.size, "header equal", np.array_equal(got[:30], want[:30]))
        bg, bw = helpers.list_blocks(got), helpers.list_blocks(want)
        print("blocks", len(bg), len(bw))
        og = ow = 30
        pos = 0
        for i, (a, b) in enumerate(zip(bg, bw)):
            if not (a == b and np.array_equal(got[og:og + a[2]], want[ow:ow + b[2]])):
                print("first difference: block", i, "at sample", pos, "got", a, "want", b)
                break
            og += a[2]; ow += b[2]; pos += a[1]
        else:
            print("all common blocks equal")


if __name__ == "__main__":
    main()
