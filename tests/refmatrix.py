"""The reference's own encode -> decode integration matrix as integer inputs and parameter sets.

/root/reference/test/srla_encode_decode/main.cpp:393-767 runs 360 cases: ten waveform generators (:51-209) x
{1, 2, 8 channels} x {8, 16, 24 bit} x minimum block 512 / maximum block 1024 x look-ahead {2048, 1536} x
{no LTP, no SVR, preset 0 ; LTP order 3 + 6 SVR iterations + preset 0}, 8 500 samples each, every case encoded TWICE
on the same handle.  (The test's initialisers `{ch, bits, 8000, 512, 1024, look, SRLA_MAX_LTP_ORDER,
SRLA_NUM_PARAMETER_PRESETS - 1}` fill `ltp_order` and `num_svr_filter_learning_iteration` of
include/srla_encoder.h:8-18 and leave `preset` zero.)

This module restates the INPUTS of that test (generators, double -> fixed conversion :211-236, the single `srand(0)`
of :772 with the C library's `rand()` running on through the cases) so that the goldens made from the compiled
reference (tools/gen_golden_matrix.py -> tests/golden/matrix_streams.json) can be replayed where the reference is
absent.  Every input is pinned by SHA-256 in the golden file: a C library whose rand() / libm differed would be
reported as such, not as an encoder mismatch.  Test infrastructure only."""
import ctypes as C
import ctypes.util
import math

import numpy as np

NUM_SAMPLES = 8500
RATE = 8000
MIN_BLOCK, MAX_BLOCK = 512, 1024
GENERATORS = ("silence", "sin", "sin_ch_sign_flipped", "white_noise", "chirp", "positive_constant", "negative_constant",
              "nyquist_osc", "gauss_noise", "mini_impulse")                    # main.cpp:394,431,468,505,542,579,616,653,690,727
RAND_MAX = 2147483647


def parameter_sets():
    """the 36 parameter sets every generator runs through, in the file's order: (nch, bps, lookahead, ltp_order, svr_iterations)"""
    out = []
    for ltp, svr in ((0, 0), (3, 6)):
        for look in (2048, 1536):
            for nch in (1, 2, 8):
                for bps in (8, 16, 24):
                    out.append((nch, bps, look, ltp, svr))
    return out


def cases():
    """-> list of dicts in the reference's order (the order matters: rand() is seeded once)"""
    out = []
    for g in GENERATORS:
        for nch, bps, look, ltp, svr in parameter_sets():
            out.append(dict(name="%s_%dch_%dbit_L%d_P%d_svr%d" % (g, nch, bps, look, ltp, svr), generator=g, nch=nch, bps=bps,
                            lookahead=look, ltp_order=ltp, svr_iterations=svr, preset=0, n=NUM_SAMPLES))
    return out


class _Rand:
    """the C library's srand / rand (main.cpp:772, :105, :185-186)"""

    def __init__(self):
        self.libc = C.CDLL(ctypes.util.find_library("c") or "libc.so.6")
        self.libc.rand.restype = C.c_int
        self.libc.srand.argtypes = [C.c_uint]

    def seed(self, s):
        self.libc.srand(s)

    def __call__(self):
        return self.libc.rand()


def _sin_table(n):
    return [math.sin(880.0 * math.pi * s / 44100.0) for s in range(n)]          # 440.0f * 2 * M_PI * smpl / 44100.0f, left to right


def _to_fixed(rows, bps):
    """main.cpp:211-236 with offset_lshift 0: Round(x * 2^(bps-1)) (srla_utility.c:22-25), clipped at the positive end"""
    scale = math.pow(2, bps - 1)
    top = 1 << (bps - 1)
    out = np.zeros((len(rows), len(rows[0])), dtype=np.int32)
    for c, row in enumerate(rows):
        for i, x in enumerate(row):
            d = x * scale
            v = int(math.floor(d + 0.5)) if d >= 0.0 else -int(math.floor(-d + 0.5))
            out[c, i] = top - 1 if v >= top else v
    return out


def generate_all():
    """every case's planar int32 input, in order -> list of (case, ndarray [nch][n])"""
    rnd = _Rand()
    rnd.seed(0)
    n = NUM_SAMPLES
    sin = _sin_table(n)
    chirp = [math.sin((2.0 * math.pi * s) / float(n - s)) for s in range(n)]
    fixed_cache = {}
    out = []
    for case in cases():
        g, nch, bps = case["generator"], case["nch"], case["bps"]
        key = (g, bps)
        if g == "white_noise":
            rows = [[2.0 * (float(rnd()) / RAND_MAX - 0.5) for _ in range(n)] for _ in range(nch)]
            pcm = _to_fixed(rows, bps)
        elif g == "gauss_noise":
            rows = []
            for _ in range(nch):
                row = []
                for _ in range(n):
                    x = float(rnd()) / RAND_MAX
                    y = float(rnd()) / RAND_MAX
                    c = math.cos(2.0 * math.pi * y)
                    if x > 0.0:
                        v = 0.25 * math.sqrt(-2.0 * math.log(x)) * c
                    else:                                                  # log(0) = -inf in C
                        v = math.copysign(math.inf, c)
                    v = 1.0 if v >= 1.0 else v
                    v = -1.0 if v <= -1.0 else v
                    row.append(v)
                rows.append(row)
            pcm = _to_fixed(rows, bps)
        else:
            if key not in fixed_cache:
                if g == "silence":
                    rows = [[0.0] * n, [0.0] * n]
                elif g == "sin":
                    rows = [sin, sin]
                elif g == "sin_ch_sign_flipped":
                    rows = [sin, [-1.0 * v for v in sin]]                  # pow(-1, ch) * sin(...)
                elif g == "chirp":
                    rows = [chirp, chirp]
                elif g == "positive_constant":
                    rows = [[1.0] * n] * 2
                elif g == "negative_constant":
                    rows = [[-1.0] * n] * 2
                elif g == "nyquist_osc":
                    r = [1.0 if (s % 2 == 0) else -1.0 for s in range(n)]
                    rows = [r, r]
                elif g == "mini_impulse":
                    r = [0.0] * n
                    r[1] = math.pow(2.0, -15.0)
                    rows = [r, r]
                else:
                    raise ValueError(g)
                fixed_cache[key] = _to_fixed(rows, bps)
            two = fixed_cache[key]
            pcm = np.ascontiguousarray(np.stack([two[c & 1] for c in range(nch)]))
        out.append((case, pcm))
    return out


def reference_config(case):
    """(SRLAEncoderConfig, SRLAEncodeParameter) exactly as main.cpp:262-271 / the test-case initialisers build them"""
    from srla_amd import capi
    cfg = capi.SRLAEncoderConfig(case["nch"], MIN_BLOCK, MAX_BLOCK, case["lookahead"], 0)    # preset 0: max_num_parameters 0
    par = capi.SRLAEncodeParameter(case["nch"], case["bps"], RATE, MIN_BLOCK, MAX_BLOCK, case["lookahead"], case["ltp_order"],
                                   case["svr_iterations"], case["preset"])
    return cfg, par


def encode_twice(lib, case, pcm):
    """main.cpp:286-347: Create, SetEncodeParameter, EncodeWhole, EncodeWhole again on the same handle (buffer 2x the PCM size)"""
    from srla_amd import capi
    cfg, par = reference_config(case)
    enc = lib.create(cfg)
    if not enc:
        raise RuntimeError("SRLAEncoder_Create failed")
    try:
        rc = lib.set_parameter(enc, par)
        if rc != capi.OK:
            raise RuntimeError("SRLAEncoder_SetEncodeParameter -> %d" % rc)
        cap = capi.HEADER_SIZE + (2 * case["nch"] * case["n"] * case["bps"]) // 8
        outs = []
        for _ in range(2):
            rc, data = lib.encode_whole(enc, pcm, cap=cap)
            if rc != capi.OK:
                raise RuntimeError("SRLAEncoder_EncodeWhole -> %d" % rc)
            outs.append(data)
        return outs
    finally:
        lib.destroy(enc)
