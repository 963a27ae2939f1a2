#!/usr/bin/env python3
"""Generate tests/golden/* from the COMPILED REFERENCE (oracle/_ref/libsrla_ref.so).

Runs only where /root/reference exists (this container).  What is committed is data: inputs are
re-creatable from (generator kind, seed, length) and are pinned by their SHA-256; outputs are the
reference's own bytes (small streams) or their SHA-256 + size (long streams), plus stage-level
vectors obtained by calling the reference's exported functions (FFT_RealFFT,
LPCCalculator_CalculateMultipleLPCCoefficients, LPC_QuantizeCoefficients, SRLACoder_ComputeCodeLength,
LPCCalculator_CalculateLTPCoefficients, SRLAUtility_CalculateFletcher16CheckSum ...).

    python tools/gen_golden.py            # rewrites tests/golden/
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers  # noqa: E402
from helpers import SINE, MUSIC, VARIED, NOISE  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
assert helpers.have_reference(), "the compiled reference is required"
ref = helpers.reference_encoder()
rl = ref.lib


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def edge_signal(name, nch, n, bps):
    """integer-only edge inputs in the spirit of test/srla_encode_decode/main.cpp:51-209"""
    full = (1 << (bps - 1)) - 1
    a = np.zeros((nch, n), dtype=np.int32)
    if name == "silence":
        pass
    elif name == "const_pos":
        a[:] = full
    elif name == "const_neg":
        a[:] = -full - 1
    elif name == "nyquist":
        a[:, 0::2] = full
        a[:, 1::2] = -full - 1
    elif name == "impulse":
        a[:, 0] = 1
        a[:, n // 2] = -1
    elif name == "one_silent":
        a[0] = helpers.synth(MUSIC, 77, 48000, 1, n, bps)[0]
    elif name == "sign_flip":
        s = helpers.synth(SINE, 1, 48000, 1, n, bps)[0]
        a[0] = s
        if nch > 1:
            a[1] = -s
    elif name == "white_full":
        a[:] = np.random.RandomState(99).randint(-full - 1, full + 1, size=(nch, n)).astype(np.int32)
    elif name == "lshift3":
        a[:] = helpers.synth(MUSIC, 78, 48000, nch, n, bps)
        a[:] = (a >> 3) << 3
    else:
        raise ValueError(name)
    return a


def make_input(spec):
    if "edge" in spec:
        return edge_signal(spec["edge"], spec["nch"], spec["n"], spec["bps"])
    return helpers.synth(spec["kind"], spec["seed"], spec["rate"], spec["nch"], spec["n"], spec["bps"])


# ---------------------------------------------------------------------------- whole streams ---
CLI = {
    "m0_B2048": dict(preset=0, max_block=2048, divisions=1),
    "m2_B4096": dict(preset=2, max_block=4096, divisions=1),
    "m4_B4096": dict(preset=4, max_block=4096, divisions=1),
    "m4_B4096_V0": dict(preset=4, max_block=4096, divisions=0),
    "m4_B4096_V2": dict(preset=4, max_block=4096, divisions=2),
    "m4_B8192_V2_P3": dict(preset=4, max_block=8192, divisions=2, ltp_order=3),
    "m4_B4096_V2_P3": dict(preset=4, max_block=4096, divisions=2, ltp_order=3),
    "m6_B1024_V1_P1": dict(preset=6, max_block=1024, divisions=1, lookahead_factor=2, ltp_order=1),
    "m1_B8192_V2_P3": dict(preset=1, max_block=8192, divisions=2, ltp_order=3),
    "m3_B4096_V2_P3": dict(preset=3, max_block=4096, divisions=2, ltp_order=3),
    # blocks above 8192 samples and more than 64 search nodes per window (the reference accepts any -B / -V / -L)
    "m4_B16384_V1": dict(preset=4, max_block=16384, divisions=1),
    "m4_B32768_V2_P3": dict(preset=4, max_block=32768, divisions=2, ltp_order=3),
    "m2_B32768_V0": dict(preset=2, max_block=32768, divisions=0),
    "m4_B2048_V3_L16": dict(preset=4, max_block=2048, divisions=3, lookahead_factor=16),
    # --svr-filter-learning-iteration (lpc.c:1036-1136)
    "m2_B4096_svr1": dict(preset=2, max_block=4096, divisions=1, svr_iterations=1),
    "m4_B4096_svr5": dict(preset=4, max_block=4096, divisions=1, svr_iterations=5),
    "m4_B4096_V2_P3_svr2": dict(preset=4, max_block=4096, divisions=2, ltp_order=3, svr_iterations=2),
    "m2_B4096_svr5": dict(preset=2, max_block=4096, divisions=1, svr_iterations=5),
    "m5_B4096_svr2": dict(preset=5, max_block=4096, divisions=1, svr_iterations=2),
    "m6_B2048_V0_svr3": dict(preset=6, max_block=2048, divisions=0, svr_iterations=3),
    "m4_B16384_svr2": dict(preset=4, max_block=16384, divisions=1, svr_iterations=2),
}

cases = []


def add(name, spec, cli_name, store_bytes=False):
    cases.append(dict(name=name, input=spec, cli_name=cli_name, cli=CLI[cli_name], store_bytes=store_bytes))


# BASELINE.json configs at full size (60 s stereo 48 kHz / 10 s mono 44.1 kHz sine)
add("C1_sine_mono", dict(kind=SINE, seed=1, rate=44100, nch=1, n=441000, bps=16), "m0_B2048")
music60 = dict(kind=MUSIC, seed=1, rate=48000, nch=2, n=2880000, bps=16)
for c in ("m2_B4096", "m4_B4096", "m4_B4096_V0", "m4_B4096_V2", "m4_B8192_V2_P3", "m4_B4096_V2_P3"):
    add("music60_" + c, music60, c)
varied30 = dict(kind=VARIED, seed=2, rate=48000, nch=2, n=1440000, bps=16)
for c in ("m4_B4096", "m4_B4096_V2_P3", "m4_B8192_V2_P3", "m0_B2048"):
    add("varied30_" + c, varied30, c)
# medium: channel counts, bit depths, even tails
for nch in (1, 2, 3):
    for c in ("m4_B4096", "m4_B4096_V2_P3", "m6_B1024_V1_P1"):
        add("varied5_%dch_%s" % (nch, c), dict(kind=VARIED, seed=10 + nch, rate=48000, nch=nch, n=240000 + 1234, bps=16), c)
for bps in (8, 24):
    for c in ("m4_B4096", "m3_B4096_V2_P3"):
        add("varied_%dbit_%s" % (bps, c), dict(kind=VARIED, seed=3, rate=44100, nch=2, n=60000, bps=bps), c)
        add("music_%dbit_%s" % (bps, c), dict(kind=MUSIC, seed=3, rate=44100, nch=2, n=60000, bps=bps), c)
add("noise5_m4", dict(kind=NOISE, seed=5, rate=48000, nch=2, n=240000, bps=16), "m4_B4096")
# small streams whose bytes are committed
for c in ("m0_B2048", "m2_B4096", "m4_B4096", "m4_B4096_V2", "m4_B4096_V2_P3", "m4_B8192_V2_P3"):
    add("music_small_" + c, dict(kind=MUSIC, seed=21, rate=48000, nch=2, n=20000, bps=16), c, store_bytes=True)
add("varied_small_m4", dict(kind=VARIED, seed=22, rate=8000, nch=2, n=40000, bps=16), "m4_B4096", store_bytes=True)
for e in ("silence", "const_pos", "const_neg", "nyquist", "impulse", "one_silent", "sign_flip", "lshift3", "white_full"):
    for nch in (1, 2):
        if e in ("one_silent", "sign_flip") and nch == 1:
            continue
        add("edge_%s_%dch" % (e, nch), dict(edge=e, nch=nch, n=8500 - 500 * nch, bps=16, rate=48000), "m4_B4096_V2_P3" if e != "lshift3" else "m4_B4096",
            store_bytes=(nch == 2 and e in ("silence", "nyquist", "lshift3")))
for c in ("m4_B16384_V1", "m4_B32768_V2_P3", "m2_B32768_V0", "m4_B2048_V3_L16"):
    add("big_music_" + c, dict(kind=MUSIC, seed=61, rate=48000, nch=2, n=300000, bps=16), c)
for c in ("m2_B4096_svr1", "m4_B4096_svr5", "m4_B4096_V2_P3_svr2", "m2_B4096_svr5"):
    add("svr_music_" + c, dict(kind=MUSIC, seed=71, rate=48000, nch=2, n=60000, bps=16), c)
    add("svr_varied_" + c, dict(kind=VARIED, seed=72, rate=48000, nch=2, n=49152, bps=16), c)
for c in ("m5_B4096_svr2", "m6_B2048_V0_svr3", "m4_B16384_svr2"):
    add("svrbig_music_" + c, dict(kind=MUSIC, seed=73, rate=48000, nch=2, n=49152, bps=16), c)
# odd lengths: the reference is history dependent here (LPC window skips the middle sample); the
# oracle reproduces it, the device path documents the deviation
add("odd_tail_music", dict(kind=MUSIC, seed=31, rate=48000, nch=2, n=20001, bps=16), "m4_B4096")
add("odd_tail_varied_P3", dict(kind=VARIED, seed=32, rate=48000, nch=2, n=30001, bps=16), "m4_B4096_V2_P3")

streams = []
for c in cases:
    pcm = make_input(c["input"])
    rate = c["input"].get("rate", 48000)
    data = ref.encode(pcm, bits_per_sample=c["input"]["bps"], sampling_rate=rate, **c["cli"])
    blocks = helpers.list_blocks(data)
    entry = dict(name=c["name"], input=c["input"], cli=c["cli"], input_sha256=sha(pcm), srl_sha256=sha(data), srl_size=int(data.size),
                 num_blocks=len(blocks), raw_blocks=sum(1 for b in blocks if b[0] == 2), silent_blocks=sum(1 for b in blocks if b[0] == 1),
                 odd_length=bool(c["input"]["n"] % 2))
    if c["store_bytes"]:
        fn = c["name"] + ".srl"
        with open(os.path.join(GOLD, fn), "wb") as f:
            f.write(data.tobytes())
        entry["file"] = fn
    streams.append(entry)
    print("%-34s %9d -> %9d  blocks %4d raw %3d silent %3d" % (c["name"], pcm.size, data.size, len(blocks), entry["raw_blocks"], entry["silent_blocks"]), flush=True)
json.dump(dict(generator="tools/gen_golden.py", reference="aikiriao/SRLA codec 18 / format 10, gcc -std=c90 -O3 -mavx2 (oracle/Makefile ref)",
               streams=streams), open(os.path.join(GOLD, "streams.json"), "w"), indent=1)

# ------------------------------------------------------------------- stage-level vectors ------
stage = {}
rng = np.random.RandomState(12345)

# FFT_RealFFT forward / inverse (libs/fft/include/fft.h:35)
rl.FFT_RealFFT.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
for n in (4, 8, 16, 32, 64, 256, 1024, 2048, 4096, 8192):
    x = rng.uniform(-1, 1, n)
    stage["fft_in_%d" % n] = x.copy()
    y = x.copy(); w = np.zeros(n)
    rl.FFT_RealFFT(n, -1, y.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
    stage["fft_fwd_%d" % n] = y.copy()
    rl.FFT_RealFFT(n, 1, y.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
    stage["fft_inv_%d" % n] = y.copy()


class LPCConfig(C.Structure):
    _fields_ = [("max_order", C.c_uint32), ("max_num_samples", C.c_uint32)]


rl.LPCCalculator_Create.restype = C.c_void_p
rl.LPCCalculator_Create.argtypes = [C.POINTER(LPCConfig), C.c_void_p, C.c_int32]
rl.LPCCalculator_Destroy.argtypes = [C.c_void_p]
rl.LPCCalculator_CalculateMultipleLPCCoefficients.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_double]
rl.LPCCalculator_CalculateLTPCoefficients.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), C.c_int, C.c_double]
rl.LPC_QuantizeCoefficients.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
rl.SRLACoder_Create.restype = C.c_void_p
rl.SRLACoder_Create.argtypes = [C.c_uint32, C.c_void_p, C.c_int32]
rl.SRLACoder_ComputeCodeLength.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
rl.SRLACoder_ComputeCodeLength.restype = C.c_uint32
rl.SRLAUtility_CalculateFletcher16CheckSum.argtypes = [C.c_void_p, C.c_size_t]
rl.SRLAUtility_CalculateFletcher16CheckSum.restype = C.c_uint16
LPC_WINDOWTYPE_WELCH = 2  # libs/lpc/include/lpc.h enum order: RECTANGULAR, SIN, WELCH
# all-order LPC (window + FFT autocorrelation + Levinson + compensation), lpc.c:535-570
lpc_cases = []
for idx, (n, order, kind, seed) in enumerate([(4096, 64, MUSIC, 41), (2048, 64, MUSIC, 42), (3072, 32, VARIED, 43), (4096, 16, NOISE, 44),
                                              (1000, 8, MUSIC, 45), (8192, 64, MUSIC, 46), (4096, 128, MUSIC, 47)]):
    sig = helpers.synth(kind, seed, 48000, 1, n)[0].astype(np.float64) * 2.0 ** -15
    cfg = LPCConfig(max(order, 3), n)
    h = rl.LPCCalculator_Create(C.byref(cfg), None, 0)
    rows = np.zeros((order, order)); ptrs = (C.c_void_p * order)(*[rows[k].ctypes.data for k in range(order)])
    ev = np.zeros(order + 1)
    rc = rl.LPCCalculator_CalculateMultipleLPCCoefficients(h, sig.ctypes.data_as(C.c_void_p), n, ptrs, ev.ctypes.data_as(C.c_void_p), order, LPC_WINDOWTYPE_WELCH, 1e-5)
    assert rc == 0
    rl.LPCCalculator_Destroy(h)
    stage["lpc_rows_%d" % idx] = rows.copy(); stage["lpc_errvars_%d" % idx] = ev.copy()
    # quantiser on a few rows
    q = {}
    for o in (1, 2, order // 2, order):
        ic = np.zeros(o, dtype=np.int32); rs = C.c_uint32(0)
        assert rl.LPC_QuantizeCoefficients(rows[o - 1].ctypes.data_as(C.c_void_p), o, 8, 16, ic.ctypes.data_as(C.c_void_p), C.byref(rs)) == 0
        stage["lpc_q_%d_%d" % (idx, o)] = np.concatenate([[rs.value], ic]).astype(np.int32)
    lpc_cases.append(dict(index=idx, n=n, order=order, kind=kind, seed=seed))

# LTP: pitch + 3 taps, lpc.c:1558-1649 ; plus the reference test's own KAT (test/lpc/main.cpp:232-262:
# a 2048-sample sine of period p is detected as p for p = 10, 20, ..., 190)
ltp_cases = []
for idx, (n, kind, seed) in enumerate([(4096, MUSIC, 51), (8192, MUSIC, 52), (4096, VARIED, 53), (2048, SINE, 1)]):
    sig = helpers.synth(kind, seed, 48000, 1, n)[0].astype(np.float64) * 2.0 ** -15
    cfg = LPCConfig(255, n)
    h = rl.LPCCalculator_Create(C.byref(cfg), None, 0)
    coef = np.zeros(3); period = C.c_int32(0)
    rc = rl.LPCCalculator_CalculateLTPCoefficients(h, sig.ctypes.data_as(C.c_void_p), n, 8, 262, coef.ctypes.data_as(C.c_void_p), 3, C.byref(period), LPC_WINDOWTYPE_WELCH, 1e-5)
    rl.LPCCalculator_Destroy(h)
    ltp_cases.append(dict(index=idx, n=n, kind=kind, seed=seed, rc=int(rc), period=int(period.value) if rc == 0 else 0))
    stage["ltp_coef_%d" % idx] = coef.copy()
sine_periods = []
for p in range(10, 200, 10):
    n = 2048
    sig = np.sin(2.0 * np.pi * np.arange(n) / p)
    cfg = LPCConfig(255, n)
    h = rl.LPCCalculator_Create(C.byref(cfg), None, 0)
    coef = np.zeros(3); period = C.c_int32(0)
    rc = rl.LPCCalculator_CalculateLTPCoefficients(h, sig.ctypes.data_as(C.c_void_p), n, 8, 262, coef.ctypes.data_as(C.c_void_p), 3, C.byref(period), LPC_WINDOWTYPE_WELCH, 1e-5)
    rl.LPCCalculator_Destroy(h)
    sine_periods.append(dict(period=p, rc=int(rc), detected=int(period.value)))
    stage["ltp_sine_in_%d" % p] = sig

# residual code length, srla_coder.c:701
code_cases = []
coder = rl.SRLACoder_Create(8192, None, 0)
for idx, (n, scale, seed) in enumerate([(4096, 300, 1), (4096, 1, 2), (2048, 20000, 3), (3072, 50, 4), (4095, 100, 5), (1024, 0, 6), (8192, 3, 7), (512, 1 << 20, 8)]):
    r = np.random.RandomState(seed)
    res = np.round(r.laplace(0, max(scale, 1e-9), n)).astype(np.int32) if scale else np.zeros(n, np.int32)
    if idx == 3:
        res[1000:2000] = np.round(r.laplace(0, 5000, 1000)).astype(np.int32)   # non-stationary: partitions matter
    bits = rl.SRLACoder_ComputeCodeLength(coder, res.ctypes.data_as(C.c_void_p), n)
    stage["code_in_%d" % idx] = res
    code_cases.append(dict(index=idx, n=n, bits=int(bits)))

np.savez_compressed(os.path.join(GOLD, "stages.npz"), **stage)

# ------------------------------------------------------------------------ the reference's KATs ---
kats = dict(
    # test/srla_internal/main.cpp:27-29
    fletcher16=[dict(text="abcde", value=0xC8F0), dict(text="abcdef", value=0x2057), dict(text="abcdefgh", value=0x0627)],
    # test/srla_encoder/srla_encoder_test.cpp:613-728 (graph data: node pairs and weights; answers)
    dijkstra=[
        dict(num_nodes=2, start=0, goal=1, min_cost=114514, edges=[[0, 1, 114514]], route=[0, 1]),
        dict(num_nodes=7, start=0, goal=6, min_cost=45,
             edges=[[0, 1, 30], [0, 3, 10], [0, 2, 15], [1, 3, 25], [1, 4, 60], [2, 3, 40], [2, 5, 20], [3, 6, 35], [4, 6, 20], [5, 6, 30]],
             route=[0, 3, 6]),
        dict(num_nodes=30, start=0, goal=29, min_cost=213, route=[0, 4, 5, 10, 15, 20, 24, 25, 29], edges=[
            [0, 1, 15], [0, 2, 58], [0, 3, 79], [0, 4, 1], [0, 5, 44], [0, 6, 78], [0, 7, 61], [0, 8, 90], [0, 9, 95],
            [1, 2, 53], [1, 3, 78], [1, 4, 49], [1, 5, 72], [1, 6, 50], [1, 7, 43], [1, 8, 25], [1, 9, 100],
            [2, 3, 51], [2, 4, 70], [2, 5, 59], [2, 6, 31], [2, 7, 71], [2, 8, 21], [2, 9, 55],
            [3, 4, 46], [3, 5, 7], [3, 6, 81], [3, 7, 92], [3, 8, 71], [3, 9, 48],
            [4, 5, 7], [4, 6, 18], [4, 7, 11], [4, 8, 36], [4, 9, 38],
            [5, 6, 54], [5, 7, 85], [5, 8, 84], [5, 9, 36], [5, 10, 1],
            [6, 7, 57], [6, 8, 85], [6, 9, 45], [7, 8, 28], [7, 9, 93], [8, 9, 11], [9, 10, 92],
            [10, 11, 29], [10, 12, 45], [10, 13, 53], [10, 14, 8], [10, 15, 16], [10, 16, 41], [10, 17, 51], [10, 18, 95], [10, 19, 94],
            [11, 12, 64], [11, 13, 31], [11, 14, 6], [11, 15, 91], [11, 16, 72], [11, 17, 90], [11, 18, 56], [11, 19, 41],
            [12, 13, 100], [12, 14, 68], [12, 15, 48], [12, 16, 73], [12, 17, 25], [12, 18, 31], [12, 19, 79],
            [13, 14, 1], [13, 15, 38], [13, 16, 17], [13, 17, 81], [13, 18, 21], [13, 19, 58],
            [14, 15, 47], [14, 16, 35], [14, 17, 36], [14, 18, 3], [14, 19, 64],
            [15, 16, 19], [15, 17, 22], [15, 18, 51], [15, 19, 58], [15, 20, 99],
            [16, 17, 11], [16, 18, 68], [16, 19, 86], [17, 18, 63], [17, 19, 97], [18, 19, 64], [19, 20, 86],
            [20, 21, 40], [20, 22, 28], [20, 23, 59], [20, 24, 14], [20, 25, 77], [20, 26, 90], [20, 27, 91], [20, 28, 74],
            [21, 22, 21], [21, 23, 78], [21, 24, 26], [21, 25, 76], [21, 26, 38], [21, 27, 32], [21, 28, 36],
            [22, 23, 12], [22, 24, 18], [22, 25, 68], [22, 26, 40], [22, 27, 86], [22, 28, 19],
            [23, 24, 32], [23, 25, 77], [23, 26, 63], [23, 27, 57], [23, 28, 78],
            [24, 25, 33], [24, 26, 81], [24, 27, 58], [24, 28, 3],
            [25, 26, 89], [25, 27, 28], [25, 28, 83], [25, 29, 42], [26, 27, 83], [26, 28, 87], [27, 28, 75], [28, 29, 85]]),
    ],
    ltp_sine_periods=sine_periods,   # test/lpc/main.cpp:232-262, values from running the compiled reference
    lpc_cases=lpc_cases, ltp_cases=ltp_cases, code_cases=code_cases,
)
# cross-check the Fletcher KATs against the compiled reference itself
for k in kats["fletcher16"]:
    b = k["text"].encode()
    assert rl.SRLAUtility_CalculateFletcher16CheckSum(b, len(b)) == k["value"]
json.dump(kats, open(os.path.join(GOLD, "kats.json"), "w"), indent=1)
print("wrote", len(streams), "stream cases,", len(stage), "stage arrays")
