#!/usr/bin/env python3
"""Golden streams for the history regimes, made with the reference's own `srla -e` tool (oracle/_ref/srla_ref: every object
compiled from /root/reference; the library in a fresh process for the one setting the tool's flags cannot express):
parameters under which blocks anywhere in the stream depend on the calls before them -- the long-term predictor with
blocks of at most 256 samples (lpc.c:371-373) and odd minimum blocks (lpc.c:260-264).

    python tools/gen_golden_history.py          # writes tests/golden/history_streams.json (+ a few small .srl files)

Runs only where /root/reference exists.  Inputs are re-creatable from (generator kind, seed, length) and pinned by SHA-256."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
from helpers import MUSIC, VARIED, NOISE  # noqa: E402

assert helpers.have_reference(), "the compiled reference is required"
GOLD = os.path.join(ROOT, "tests", "golden")

CLI = {
    # the review's three ("next round" 1b)
    "m4_B1024_V2_P3": dict(preset=4, max_block=1024, divisions=2, ltp_order=3),
    "m2_B512_V1_P1": dict(preset=2, max_block=512, divisions=1, ltp_order=1),
    "m4_B2048_V3_P3": dict(preset=4, max_block=2048, divisions=3, ltp_order=3),
    # minimum block 128: lags from two earlier calls
    "m3_B1024_V3_P3": dict(preset=3, max_block=1024, divisions=3, ltp_order=3),
    # (`-B 256 -V 0 -P 3` is NOT here: with a maximum block of at most 256 samples the reference's FFT buffer itself is shorter
    #  than the 263 lags and lpc.c:371-373 reads beyond it, into the transform's scratch buffer -- the library names that
    #  setting as not bit-identical, tests/test_history_mode.py::test_ltp_with_a_buffer_shorter_than_the_lags_is_named)
    # odd minimum blocks: `srla -e -B 4095 -V 0`, `-B 1000 -V 3` (125), `-B 3000 -V 2 -L 2` (750 even, for contrast: not a history regime without LTP)
    "m4_B4095_V0": dict(preset=4, max_block=4095, divisions=0),
    "m4_B4095_V0_P3": dict(preset=4, max_block=4095, divisions=0, ltp_order=3),
    "m2_B1000_V3": dict(preset=2, max_block=1000, divisions=3),
    "m2_B1000_V3_P1": dict(preset=2, max_block=1000, divisions=3, ltp_order=1),
    "m0_B1000_V3": dict(preset=0, max_block=1000, divisions=3),
    # through the API only (the tool derives min = max >> V): min 375, max 3000, look-ahead 6000
    "m4_min375_max3000_P3": dict(preset=4, max_block=3000, min_block=375, lookahead=6000, ltp_order=3),
    # --svr-filter-learning-iteration: the refinement leaves its residual in the buffer the next history-dependent call reads
    # (lpc.c:1047).  In the history regimes, and -- ordinary parameters, odd stream length -- in the last window alone.
    "m2_B1000_V3_svr1": dict(preset=2, max_block=1000, divisions=3, svr_iterations=1),
    "m4_B1024_V2_P3_svr1": dict(preset=4, max_block=1024, divisions=2, ltp_order=3, svr_iterations=1),
    "m4_B4095_V0_svr2": dict(preset=4, max_block=4095, divisions=0, svr_iterations=2),
    "m4_B4096_V1_svr2": dict(preset=4, max_block=4096, divisions=1, svr_iterations=2),
    "m2_B4096_V2_P3_svr1": dict(preset=2, max_block=4096, divisions=2, ltp_order=3, svr_iterations=1),
    "m3_B2048_V0_svr3": dict(preset=3, max_block=2048, divisions=0, svr_iterations=3),
}
SVR_ONLY_ODD = ("m4_B4096_V1_svr2", "m2_B4096_V2_P3_svr1", "m3_B2048_V0_svr3")   # not history regimes: only an odd-length stream's tail

cases = []


def add(name, spec, cli_name, store=False):
    cases.append(dict(name=name, input=spec, cli_name=cli_name, cli=CLI[cli_name], store_bytes=store))


for c in CLI:
    if "svr" in c:
        # (the refinement is slow in the reference too: shorter inputs)
        if c not in SVR_ONLY_ODD: add("hist_music_" + c, dict(kind=MUSIC, seed=81, rate=48000, nch=2, n=24000, bps=16), c)
        add("hist_varied_odd_" + c, dict(kind=VARIED, seed=82, rate=48000, nch=2, n=30003, bps=16), c)
        add("hist_music_odd_" + c, dict(kind=MUSIC, seed=87, rate=44100, nch=1, n=20001, bps=16), c)
        continue
    add("hist_music_" + c, dict(kind=MUSIC, seed=81, rate=48000, nch=2, n=60000, bps=16), c)
    add("hist_varied_odd_" + c, dict(kind=VARIED, seed=82, rate=48000, nch=2, n=48003, bps=16), c)
for c in ("m4_B1024_V2_P3", "m2_B512_V1_P1", "m4_B2048_V3_P3", "m4_B4095_V0", "m2_B1000_V3_P1"):
    add("hist_small_" + c, dict(kind=MUSIC, seed=83, rate=48000, nch=2, n=12000, bps=16), c, store=True)
for c, nch, bps in (("m4_B1024_V2_P3", 1, 16), ("m4_B1024_V2_P3", 3, 24), ("m2_B1000_V3_P1", 1, 8), ("m4_B4095_V0_P3", 3, 24), ("m3_B1024_V3_P3", 5, 16)):
    add("hist_%dch_%dbit_%s" % (nch, bps, c), dict(kind=VARIED, seed=84, rate=44100, nch=nch, n=30000 + 2 * nch, bps=bps), c)
add("hist_noise_m4_B1024_V2_P3", dict(kind=NOISE, seed=85, rate=48000, nch=2, n=40000, bps=16), "m4_B1024_V2_P3")
# 10 s of stereo at the review's first setting: 117 windows
add("hist_music10s_m4_B1024_V2_P3", dict(kind=MUSIC, seed=86, rate=48000, nch=2, n=480000, bps=16), "m4_B1024_V2_P3")

streams = []
for c in cases:
    sp = c["input"]
    pcm = helpers.synth(sp["kind"], sp["seed"], sp["rate"], sp["nch"], sp["n"], sp["bps"])
    if "min_block" in c["cli"] or sp["bps"] == 8:
        # not expressible with the tool's flags (or 8-bit input, which the tool refuses to open): the reference library in a fresh process
        data = helpers.reference_encode_fresh(pcm, bits_per_sample=sp["bps"], sampling_rate=sp["rate"], **c["cli"])
        how = "library, fresh process"
    else:
        data = helpers.reference_tool_encode(pcm, bits_per_sample=sp["bps"], sampling_rate=sp["rate"], **c["cli"])
        how = "srla -e (oracle/_ref/srla_ref)"
    assert np.array_equal(helpers.reference_decoder().decode(data)[0], pcm)
    blocks = helpers.list_blocks(data)
    entry = dict(name=c["name"], input=sp, cli=c["cli"], input_sha256=helpers.sha256(pcm), srl_sha256=helpers.sha256(data), srl_size=int(data.size),
                 num_blocks=len(blocks), made_with=how)
    if c["store_bytes"]:
        fn = c["name"] + ".srl"
        data.tofile(os.path.join(GOLD, fn))
        entry["file"] = fn
    streams.append(entry)
    print("%-46s %8d -> %8d  blocks %4d" % (c["name"], pcm.size, data.size, len(blocks)), flush=True)
json.dump(dict(generator="tools/gen_golden_history.py",
               reference="aikiriao/SRLA codec 18 / format 10, gcc -std=c90 -O3 -mavx2 (oracle/Makefile ref_tool: the `srla` tool itself, one run per stream)",
               streams=streams), open(os.path.join(GOLD, "history_streams.json"), "w"), indent=1)
