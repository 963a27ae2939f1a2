#!/usr/bin/env python3
"""Golden streams for (1) the reference's own integration matrix (/root/reference/test/srla_encode_decode/main.cpp:393-767, 360
cases, each encoded twice on one handle as the test does) and (2) explicit (minimum block, maximum block, look-ahead) triples that
the `srla` tool's flags cannot express -- look-ahead = k x minimum with k not a multiple of maximum / minimum, maximum / minimum
not a power of two --, all made with the compiled reference (oracle/_ref/libsrla_ref.so) through its C API, one fresh process
per case.

    python tools/gen_golden_matrix.py          # writes tests/golden/matrix_streams.json

Runs only where /root/reference exists.  Inputs are re-creatable (tests/refmatrix.py; tools/synth) and pinned by SHA-256."""
import json
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
import refmatrix  # noqa: E402
from helpers import MUSIC, VARIED, NOISE  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

WORKER = r"""
import sys, pickle
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import numpy as np, helpers, refmatrix
from srla_amd import capi
job = pickle.load(open(sys.argv[1], 'rb'))
lib = helpers.reference_encoder()
if job['kind'] == 'matrix':
    outs = refmatrix.encode_twice(lib, job['case'], job['pcm'])
else:
    outs = [lib.encode(job['pcm'], bits_per_sample=job['bps'], sampling_rate=job['rate'], **job['cli'])]
pickle.dump(outs, open(sys.argv[2], 'wb'))
"""

# (2) explicit triples.  min / max / look-ahead; the comments say what each is there for.
TRIPLES = {
    "min512_max1024_L1536_m4": dict(preset=4, min_block=512, max_block=1024, lookahead=1536),                 # the matrix's shape, with a predictor
    "min512_max1024_L1536_m4_P3": dict(preset=4, min_block=512, max_block=1024, lookahead=1536, ltp_order=3),
    "min512_max1024_L2560_m2": dict(preset=2, min_block=512, max_block=1024, lookahead=2560),                 # 2.5 x max
    "min384_max1152_L2688_m4": dict(preset=4, min_block=384, max_block=1152, lookahead=2688),                 # max = 3 x min, look-ahead 7 x min
    "min1024_max4096_L5120_m4": dict(preset=4, min_block=1024, max_block=4096, lookahead=5120),               # 1.25 x max
    "min1024_max4096_L5120_m4_P3": dict(preset=4, min_block=1024, max_block=4096, lookahead=5120, ltp_order=3),
    "min1024_max3072_L7168_m3_P1": dict(preset=3, min_block=1024, max_block=3072, lookahead=7168, ltp_order=1),
    "min500_max1500_L3500_m2": dict(preset=2, min_block=500, max_block=1500, lookahead=3500),                 # nothing a power of two
    "min640_max3200_L4480_m4_svr2": dict(preset=4, min_block=640, max_block=3200, lookahead=4480, svr_iterations=2),
    "min256_max768_L1792_m2_P3": dict(preset=2, min_block=256, max_block=768, lookahead=1792, ltp_order=3),   # history regime (blocks <= 256 with the LTP)
    "min333_max999_L2331_m2": dict(preset=2, min_block=333, max_block=999, lookahead=2331),                   # history regime (odd blocks)
    "min2048_max8192_L10240_m4_P3": dict(preset=4, min_block=2048, max_block=8192, lookahead=10240, ltp_order=3),
    "min512_max1024_L1536_m0_P3_svr6": dict(preset=0, min_block=512, max_block=1024, lookahead=1536, ltp_order=3, svr_iterations=6),
    # preset 0 with SVR iterations in the history regimes (the refinement is skipped at srla_encoder.c:1084)
    "min125_max1000_L4000_m0_svr1": dict(preset=0, min_block=125, max_block=1000, lookahead=4000, svr_iterations=1),
    "min256_max1024_L4096_m0_P3_svr2": dict(preset=0, min_block=256, max_block=1024, lookahead=4096, ltp_order=3, svr_iterations=2),
    "min2048_max4096_L16384_m0_svr1": dict(preset=0, min_block=2048, max_block=4096, lookahead=16384, svr_iterations=1),   # chain mode on odd lengths
}
TRIPLE_INPUTS = [
    ("music", dict(kind=MUSIC, seed=91, rate=48000, nch=2, n=60000, bps=16)),
    ("varied_odd", dict(kind=VARIED, seed=92, rate=48000, nch=2, n=48003, bps=16)),
    ("noise_3ch24", dict(kind=NOISE, seed=93, rate=44100, nch=3, n=20010, bps=24)),
    ("music_mono8", dict(kind=MUSIC, seed=94, rate=44100, nch=1, n=30003, bps=8)),   # (a 1-sample tail block with preset 0 crashes the reference, srla_utility.c:228: lengths here avoid it)
]


# Inputs of their own.  "impulses": identical impulse trains in both channels between silent stretches, behind a busy block (a slice of
# the "varied" generator, low three bits cleared) -- S = R - L is ALL ZERO in blocks that are not silent, so with an odd block length the
# analysis of S sees nothing but the word the Welch window leaves untouched (lpc.c:260-264): rounding noise of the call before.  The
# long-term predictor then finds its "pitch" among the stale words beyond the FFT buffer and solves for taps of 1e30 and more, which
# the reference converts to int32 out of range (srla_encoder.c:1031-1037) -- found by tools/gpu_sweep.py, seed 42 case 37, round 4.
IMPULSES = dict(kind=VARIED, seed=5037, rate=48000, nch=2, n=160000, bps=16, first=143000, count=11000, lshift=3)
EXTRA = [
    ("impulses", IMPULSES, "min125_max1000_L1375_m4_P3", dict(preset=4, min_block=125, max_block=1000, lookahead=1375, ltp_order=3)),
    ("impulses", IMPULSES, "min125_max1000_L1000_m0_P1", dict(preset=0, min_block=125, max_block=1000, lookahead=1000, ltp_order=1)),
    ("impulses", IMPULSES, "B999_V0_m2_P3", dict(preset=2, min_block=999, max_block=999, lookahead=3996, ltp_order=3)),
    ("impulses", IMPULSES, "B4096_V2_m4_P3", dict(preset=4, max_block=4096, divisions=2, ltp_order=3)),
    # one look-ahead window and a tail whose last block (160 samples) is shorter than the LTP's 263 lags: ONE regular job beside the chain-mode
    # window, both with the SVR refinement of order 255 in global scratch (tools/gpu_sweep.py, seed 48 case 60, round 4: a shared scratch region)
    # more than 128 minimum blocks per look-ahead window (the library's limit until round 4): `srla -e -B 4096 -V 6` (257 search nodes,
    # 14 368 candidates per window), 513 nodes with the candidates beyond the pricing kernel's LDS, 513 nodes in history mode
    ("music_40000", dict(kind=MUSIC, seed=95, rate=48000, nch=2, n=40000, bps=16), "B4096_V6_m2", dict(preset=2, max_block=4096, divisions=6)),
    ("music_40000", dict(kind=MUSIC, seed=95, rate=48000, nch=2, n=40000, bps=16), "B2048_V6_L8_m1", dict(preset=1, max_block=2048, divisions=6, lookahead_factor=8)),
    ("varied_40001", dict(kind=VARIED, seed=96, rate=48000, nch=2, n=40001, bps=16), "B1024_V4_L32_m2_P3", dict(preset=2, max_block=1024, divisions=4, lookahead_factor=32, ltp_order=3)),
    # blocks above 32 768 samples (the library's limit until round 4; a block header holds 65 535 at most): the 65 536-point transform,
    # signal staging in global memory
    ("music_150001", dict(kind=MUSIC, seed=97, rate=48000, nch=2, n=150001, bps=16), "B65535_V0_m4", dict(preset=4, max_block=65535, divisions=0)),
    ("music_150001", dict(kind=MUSIC, seed=97, rate=48000, nch=2, n=150001, bps=16), "B50000_V1_m2_P3", dict(preset=2, max_block=50000, divisions=1, lookahead_factor=2, ltp_order=3)),
    ("varied_3ch24_140000", dict(kind=VARIED, seed=98, rate=96000, nch=3, n=140000, bps=24), "B40000_V2_m5_P1", dict(preset=5, max_block=40000, divisions=2, lookahead_factor=1, ltp_order=1)),
    ("music_150001", dict(kind=MUSIC, seed=97, rate=48000, nch=2, n=150001, bps=16), "B65535_V0_m3_svr1", dict(preset=3, max_block=65535, divisions=0, svr_iterations=1)),
    ("sine_window_and_tail", dict(kind=0, seed=5060, rate=48000, nch=2, n=32416, bps=16), "min1536_max4608_L16896_m6_P1_svr1",
     dict(preset=6, min_block=1536, max_block=4608, lookahead=16896, ltp_order=1, svr_iterations=1)),
    ("sine_window_and_tail", dict(kind=0, seed=5060, rate=48000, nch=2, n=32416, bps=16), "min1536_max4608_L16896_m5_P1_svr2",
     dict(preset=5, min_block=1536, max_block=4608, lookahead=16896, ltp_order=1, svr_iterations=2)),
]


def run_fresh(job):
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "job.pkl"), os.path.join(d, "out.pkl")
        pickle.dump(job, open(src, "wb"))
        rc = subprocess.call([sys.executable, "-c", WORKER % dict(tests=os.path.join(ROOT, "tests"), root=ROOT), src, dst])
        if rc != 0:
            raise RuntimeError("the reference failed (exit status %d) on %r" % (rc, job.get("cli") or job.get("case")))
        return pickle.load(open(dst, "rb"))


def main():
    from concurrent.futures import ThreadPoolExecutor
    assert helpers.have_reference(), "the compiled reference is required"
    dec = helpers.reference_decoder()
    pool = ThreadPoolExecutor(6)                       # six fresh reference processes at a time; results collected in order
    path = os.path.join(GOLD, "matrix_streams.json")
    head = dict(generator="tools/gen_golden_matrix.py",
                reference="aikiriao/SRLA codec 18 / format 10, gcc -std=c90 -O3 -mavx2 (oracle/Makefile ref), C API, one fresh process per case")
    if "--triples-only" in sys.argv:
        made, outs, matrix = [], [], json.load(open(path))["matrix"]
    else:
        made = refmatrix.generate_all()
        outs = list(pool.map(lambda cp: run_fresh(dict(kind="matrix", case=cp[0], pcm=cp[1])), made))
        matrix = []
    for (case, pcm), (first, second) in zip(made, outs):
        for d in (first, second):
            assert np.array_equal(dec.decode(d)[0], pcm), case["name"]
        e = dict(case, input_sha256=helpers.sha256(pcm), srl_sha256=helpers.sha256(first), srl_size=int(first.size),
                 second_srl_sha256=helpers.sha256(second), second_srl_size=int(second.size), num_blocks=len(helpers.list_blocks(first)))
        matrix.append(e)
        print("%-52s %7d -> %7d %s" % (case["name"], pcm.size, first.size, "" if np.array_equal(first, second) else "(second call differs)"), flush=True)
    json.dump(dict(head, matrix=matrix, triples=[]), open(path, "w"), indent=1)
    jobs = []
    for cname, cli in TRIPLES.items():
        for iname, sp in TRIPLE_INPUTS:
            if cli.get("svr_iterations") and cli["preset"] > 0:
                sp = dict(sp, n=min(sp["n"], 20001 if sp["n"] & 1 else 20000))     # the refinement is slow in the reference too
            pcm = helpers.synth_spec(sp)
            jobs.append(("triple_%s_%s" % (iname, cname), sp, cli, pcm))
    for iname, sp, cname, cli in EXTRA:
        jobs.append(("triple_%s_%s" % (iname, cname), sp, cli, helpers.synth_spec(sp)))
    outs = list(pool.map(lambda j: run_fresh(dict(kind="triple", pcm=j[3], bps=j[1]["bps"], rate=j[1]["rate"], cli=j[2])), jobs))
    triples = []
    for (name, sp, cli, pcm), (data,) in zip(jobs, outs):
        assert np.array_equal(dec.decode(data)[0], pcm)
        triples.append(dict(name=name, input=sp, cli=cli, input_sha256=helpers.sha256(pcm),
                            srl_sha256=helpers.sha256(data), srl_size=int(data.size), num_blocks=len(helpers.list_blocks(data))))
        print("%-52s %7d -> %7d" % (name, pcm.size, data.size), flush=True)
    json.dump(dict(head, matrix=matrix, triples=triples), open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
