#!/usr/bin/env python3
"""Latency of the block-level entry points on an MI355X: a stream handed over block by block (ComputeBlockSize + EncodeBlock, or
EncodeOptimalPartitionedBlock per window), ms per call.  Since round 4 these calls run in history mode (they keep the handle's
buffer as the reference's calculator does); SRLA_MI355X_NO_CHAIN=1 gives the regular pipeline's time for comparison."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
from srla_amd import capi  # noqa: E402


def main():
    lib = capi.EncoderLib(helpers.PRODUCT_SO)
    sig = helpers.synth(helpers.MUSIC, 1, 48000, 2, 48000 * 20)
    for cli, api, n in ((dict(preset=4, max_block=4096, divisions=0), "block", 4096), (dict(preset=4, max_block=4096, divisions=1), "partitioned", 16384),
                        (dict(preset=4, max_block=4096, divisions=2, ltp_order=3), "partitioned", 16384), (dict(preset=4, max_block=4096, divisions=0), "size", 4096)):
        cfg, par = capi.cli_setup(2, 16, 48000, **cli)
        enc = lib.create(cfg)
        assert lib.set_parameter(enc, par) == capi.OK
        blocks = [np.ascontiguousarray(sig[:, o:o + n]) for o in range(0, sig.shape[1] - n, n)][:50]
        fn = {"block": lib.encode_block, "partitioned": lib.encode_partitioned, "size": lib.compute_block_size}[api]
        for b in blocks[:5]:
            fn(enc, b)
        t0 = time.perf_counter()
        for b in blocks:
            rc, _ = fn(enc, b)
            assert rc == capi.OK
        dt = time.perf_counter() - t0
        print("%-12s %6d samples  %s: %.3f ms per call, %.1f Msamples/s" % (api, n, cli, 1e3 * dt / len(blocks), 2 * n * len(blocks) / dt / 1e6), flush=True)
        lib.destroy(enc)


if __name__ == "__main__":
    main()
