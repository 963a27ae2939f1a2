#!/usr/bin/env python3
"""History-mode parity on an MI355X: parameters under which blocks anywhere in the stream depend on the calls before them
(an odd minimum block, lpc.c:260-264; the long-term predictor with blocks of at most 256 samples, lpc.c:371-373) --
library bytes vs oracle bytes.

    python tools/history_check.py [quick]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
from srla_amd import capi  # noqa: E402

CASES = [
    dict(preset=4, max_block=1024, divisions=2, ltp_order=3),
    dict(preset=2, max_block=512, divisions=1, ltp_order=1),
    dict(preset=4, max_block=2048, divisions=3, ltp_order=3),
    dict(preset=3, max_block=1024, divisions=3, ltp_order=3),
    dict(preset=4, max_block=256, divisions=0, ltp_order=3),
    dict(preset=4, max_block=4095, divisions=0),
    dict(preset=4, max_block=4095, divisions=0, ltp_order=3),
    dict(preset=2, max_block=1000, divisions=3),
    dict(preset=2, max_block=1000, divisions=3, ltp_order=1),
    dict(preset=0, max_block=1000, divisions=3),
    dict(preset=4, max_block=3000, divisions=2, ltp_order=3, lookahead_factor=2),
]


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    bad = 0
    for kind in (helpers.MUSIC, helpers.VARIED):
        for cli in (CASES[:3] if quick else CASES):
            for nch, n in ((2, 30000), (2, 30001), (1, 9000), (3, 12345)):
                pcm = helpers.synth(kind, 7, 48000, nch, n)
                t0 = time.perf_counter()
                got = lib.encode(pcm, **cli)
                dt = time.perf_counter() - t0
                want = helpers.Oracle(nch, **cli).encode_whole(pcm)
                ok = np.array_equal(got, want)
                bad += 0 if ok else 1
                first = -1 if ok else int(np.argmax(got[:min(got.size, want.size)] != want[:min(got.size, want.size)])) if got.size and want.size else 0
                print("%s kind=%d %s nch=%d n=%d: %d vs %d bytes, %.1f ms%s" % ("ok      " if ok else "MISMATCH", kind, cli, nch, n, got.size, want.size,
                                                                                  1e3 * dt, "" if ok else " first diff at %d" % first), flush=True)
    print("history check: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
