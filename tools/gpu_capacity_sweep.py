#!/usr/bin/env python3
"""Randomised parity sweep of the capacity added in round 4: blocks above 32 768 samples (up to 65 535, the reference's limit) and
look-ahead windows of more than 128 minimum blocks (up to 1 024).  Library bytes vs oracle bytes.

    python tools/gpu_capacity_sweep.py [cases] [seed]"""
import random
import sys

import numpy as np

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import helpers  # noqa: E402
from srla_amd import capi  # noqa: E402


def cases(count, seed):
    rnd = random.Random(seed)
    for case in range(count):
        nch = rnd.choice([1, 2, 2, 3])
        bps = rnd.choice([16, 16, 8, 24])
        ltp = rnd.choice([0, 0, 1, 3])
        if rnd.random() < 0.5:
            # big blocks
            max_block = rnd.choice([32769, 40000, 49152, 50001, 65534, 65535])
            divisions = rnd.choice([0, 0, 1, 2])
            preset = rnd.choice([0, 2, 4, 5])
            cli = dict(preset=preset, max_block=max_block, divisions=divisions, ltp_order=ltp, lookahead_factor=rnd.choice([1, 2, 4]))
            n = rnd.randint(max_block // 2, 4 * max_block) | rnd.choice([0, 1])
            if rnd.random() < 0.15 and preset in (2, 4):
                cli["svr_iterations"] = 1
                n = min(n, 2 * max_block)
        else:
            # many search nodes: min block 16 .. 128, 129 .. 1025 nodes
            minb = rnd.choice([16, 24, 32, 48, 64, 100, 128])
            ratio = rnd.choice([2, 4, 8, 16, 32])
            nodes = rnd.choice([130, 160, 200, 257, 300, 400, 513, 700, 1025])
            preset = rnd.choice([0, 1, 1, 2])
            if [0, 8, 16][preset] > minb:
                preset = 0
            if ltp and minb * ratio <= 256:
                ltp = 0                                  # LTP on an encoder created for blocks <= 256: the reference writes out of bounds and usually crashes
            cli = dict(preset=preset, min_block=minb, max_block=minb * ratio, lookahead=minb * (nodes - 1), ltp_order=ltp)
            n = rnd.randint(minb * (nodes - 1) // 2, 3 * minb * (nodes - 1)) | rnd.choice([0, 1])
            n = min(n, 60000)
        kind = rnd.choice([helpers.MUSIC, helpers.VARIED, helpers.NOISE, helpers.SINE])
        yield case, nch, bps, n, kind, cli


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    count, seed = (int(args[0]) if args else 40), (int(args[1]) if len(args) > 1 else 1)
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    done = bad = 0
    for case, nch, bps, n, kind, cli in cases(count, seed):
        pcm = helpers.synth(kind, 7000 + case, 48000, nch, n, bps)
        try:
            got = lib.encode(pcm, bits_per_sample=bps, **cli)
        except RuntimeError as e:
            print("refused", cli, nch, bps, n, e, flush=True)
            continue
        try:
            want = helpers.Oracle(nch, bits_per_sample=bps, **cli).encode_whole(pcm)
        except (RuntimeError, ValueError) as e:
            print("oracle fails (%s): case %d %s; the library's stream %s" % (e, case, cli, "decodes" if np.array_equal(helpers.oracle_decode(got), pcm) else "DOES NOT DECODE"), flush=True)
            continue
        done += 1
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH case %d (seed %d): nch=%d bps=%d n=%d kind=%d %s sizes %d vs %d" % (case, seed, nch, bps, n, kind, cli, got.size, want.size), flush=True)
    print("capacity sweep: %d compared, %d mismatches" % (done, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
