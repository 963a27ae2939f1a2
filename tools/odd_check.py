"""Odd-length streams: HIP path vs the oracle (chain mode).  Development aid; the pinned cases live in
tests/test_gpu_parity.py.  Usage: python tools/odd_check.py [quick]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from srla_amd import capi  # noqa: E402

CLIS = {
    "m4_B4096": dict(preset=4, max_block=4096, divisions=1),
    "m0_B2048": dict(preset=0, max_block=2048, divisions=1),
    "m2_B4096_V0": dict(preset=2, max_block=4096, divisions=0),
    "m4_B4096_V2": dict(preset=4, max_block=4096, divisions=2),
    "m4_B4096_V2_P3": dict(preset=4, max_block=4096, divisions=2, ltp_order=3),
    "m4_B8192_V2_P3": dict(preset=4, max_block=8192, divisions=2, ltp_order=3),
    "m6_B1024_V1_P1": dict(preset=6, max_block=1024, divisions=1, lookahead_factor=2, ltp_order=1),
    "m5_B2048_V3": dict(preset=5, max_block=2048, divisions=3, lookahead_factor=2),
    "m1_B512_V0": dict(preset=1, max_block=512, divisions=0),
}


def main():
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    bad = total = 0
    for name, cli in CLIS.items():
        for kind, nch in [(helpers.MUSIC, 2), (helpers.VARIED, 2), (helpers.VARIED, 1), (helpers.NOISE, 3)]:
            for n in (49152 + 1001, 49152 + 4097 + 512, 32768 + 77, 32768 + 2049, 3001, 301, 4095, 9001, 49152 + 82, 32768 + 4096 + 200, 131, 32768 + 8192 + 255):
                pcm = helpers.synth(kind, 70 + nch, 48000, nch, n)
                got = lib.encode(pcm, **cli)
                want = helpers.Oracle(nch, **cli).encode_whole(pcm)
                ok = got.size == want.size and bool((got == want).all())
                total += 1
                if not ok:
                    bad += 1
                    gb, wb = helpers.list_blocks(got), helpers.list_blocks(want)
                    first = next((i for i, (g, w) in enumerate(zip(gb, wb)) if g != w), min(len(gb), len(wb)))
                    print("DIFF", name, kind, nch, n, "blocks", len(gb), len(wb), "first differing block", first,
                          gb[first:first + 3], wb[first:first + 3], flush=True)
    print("odd_check: %d of %d differ" % (bad, total), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
