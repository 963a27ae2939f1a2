#!/usr/bin/env python3
"""Does the reference's output depend on what the SAME handle encoded before?  (deviation 8.)

The reference keeps one LPC calculator per encoder, so its FFT buffer (lpc.c:58,211) outlives a call: the middle word of an
odd-length block (lpc.c:260-264) and the lags beyond a short transform (lpc.c:371-373) of a LATER call's first window can hold
what an EARLIER call left.  The library starts every call from the buffer of a fresh handle (the `srla` tool's case).  This probe
runs the compiled reference (oracle/_ref, needs /root/reference to have been built here) over sequences of streams on one handle
and compares every stream after the first with the same stream encoded by a fresh process.  CPU only, test infrastructure."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers                   # noqa: E402
from srla_amd import capi        # noqa: E402


def run(ref, cli, seq, nch=2, bps=16):
    cfg, par = capi.cli_setup(nch, bps, 48000, **cli)
    enc = ref.create(cfg)
    assert enc and ref.set_parameter(enc, par) == capi.OK
    outs = []
    for pcm in seq:
        rc, d = ref.encode_whole(enc, pcm)
        assert rc == capi.OK
        outs.append(d.copy())
    ref.destroy(enc)
    return outs


def main():
    ref = helpers.reference_encoder()
    loud = helpers.synth(helpers.MUSIC, 3, 48000, 2, 20001)
    noise = helpers.synth(helpers.NOISE, 9, 48000, 2, 8000)
    rng = np.random.default_rng(1)
    total = differing = 0
    # (1) odd-length streams of every kind behind one another, regular and history regimes
    for cli in (dict(preset=4, max_block=4096, divisions=1), dict(preset=4, max_block=4096, divisions=2, ltp_order=3),
                dict(preset=4, max_block=1000, divisions=3), dict(preset=4, max_block=1024, divisions=2, ltp_order=3),
                dict(preset=4, max_block=4095, divisions=0), dict(preset=2, max_block=4096, divisions=1, svr_iterations=2)):
        seq = [helpers.synth((helpers.MUSIC, helpers.VARIED, helpers.NOISE)[i % 3], 100 + i, 48000, 2, int(rng.integers(300, 12000)) | 1)
               for i in range(8)]
        for pcm, got in list(zip(seq, run(ref, cli, seq)))[1:]:
            fresh = helpers.reference_encode_fresh(pcm, **cli)
            total += 1
            if not (got.size == fresh.size and np.array_equal(got, fresh)):
                differing += 1
                print("differs:", cli, "n =", pcm.shape[1], "sizes", got.size, fresh.size)
    # (2) a loud stream, then a short odd-length one with identical channels (S = 0: the analysis of S sees ONLY inherited words)
    for cli in (dict(preset=1, max_block=1000, divisions=3, ltp_order=3), dict(preset=4, max_block=1000, divisions=3),
                dict(preset=4, max_block=4096, divisions=2, ltp_order=3), dict(preset=3, max_block=500, divisions=1, ltp_order=1)):
        for k in range(6):
            m = helpers.synth(helpers.MUSIC if k % 2 else helpers.VARIED, 20 + k, 48000, 1, 251 + 250 * k)
            same = np.ascontiguousarray(np.vstack([m, m]))
            fresh = helpers.reference_encode_fresh(same, **cli)
            for first in (loud, noise):
                got = run(ref, cli, [first, same])[1]
                total += 1
                if not (got.size == fresh.size and np.array_equal(got, fresh)):
                    differing += 1
                    print("differs:", cli, "after", first.shape[1], "samples: identical channels, n =", same.shape[1], "sizes", got.size, fresh.size)
    print("streams encoded behind another one on the same handle: %d, differing from a fresh process: %d" % (total, differing))


if __name__ == "__main__":
    main()
