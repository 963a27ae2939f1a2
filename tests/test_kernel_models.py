"""CPU models of the integer / indexing tricks the round-3 kernels rest on (srla_amd/csrc/*.hip), each checked against the
plain formula it replaces.  No GPU: the GPU parity tests compare the kernels themselves with the oracle; these pin WHY they are
exact, at sizes and corner values the audio never reaches.

* FIR_DOT of srla_residual_cost: the signal as an int16 plane + an int8 plane, x = 2^16 h + l, the 8-bit taps packed in the four
  byte phases of the sample words (even / odd for the low plane), one closing word for the tap pair that straddles two groups:
  the wrap-around int32 sum of srla_lpc_predict.c:118-265 modulo 2^32.
* The code-bit pass on two samples per instruction: saturating 16-bit subtract, 16-bit shift, both halves summed -- against
  srla_coder.c:327-347 for every parameter 0..31 and every 16-bit value class.
* The first radix-4 stage's output permutation e ^ ((e >> 3) & 3): a bijection, conflict-free for the eight lanes a 16-byte LDS
  store is served in, and invisible to the 16-byte loads of the second stage (lane groups made of whole quads).
* The group offsets of the padded LDS layout (a pad group behind every FL groups for even FL) against the plain index formula.
"""
import numpy as np
import pytest

M32 = (1 << 32) - 1


def s8(v):
    v &= 0xFF
    return v - 256 if v >= 128 else v


def s16(v):
    v &= 0xFFFF
    return v - 65536 if v >= 32768 else v


def dot2(a, b, acc):
    """v_dot2_i32_i16: signed 16-bit halves, 32-bit wrap-around accumulate"""
    return (acc + s16(a) * s16(b) + s16(a >> 16) * s16(b >> 16)) & M32


def dot4(a, b, acc):
    """v_dot4_i32_i8"""
    return (acc + sum(s8(a >> (8 * i)) * s8(b >> (8 * i)) for i in range(4))) & M32


def fir_reference(x, taps, half):
    """acc[s] = half + sum_k cq[k] x[s - o4 + k] modulo 2^32 for s >= o4 (taps already reversed and front padded to o4)"""
    o4 = len(taps)
    return [(half + sum(int(taps[k]) * int(x[s - o4 + k]) for k in range(o4))) & M32 for s in range(o4, len(x))]


def fir_dot_model(x, taps, half):
    """The kernel's formulation for outputs s = o4 .. (chunks of four outputs, groups of four taps)."""
    o4 = len(taps)
    assert o4 % 4 == 0 and (len(x) - o4) % 4 == 0
    low = [s16(v) for v in x]
    high = [(v - s16(v)) >> 16 for v in x]
    assert all(-128 <= h <= 127 for h in high) and all(x[i] == 65536 * high[i] + low[i] for i in range(len(x)))
    lw = [(low[2 * i] & 0xFFFF) | ((low[2 * i + 1] & 0xFFFF) << 16) for i in range(len(x) // 2)]
    hw = [sum((high[4 * i + j] & 0xFF) << (8 * j) for j in range(4)) for i in range(len(x) // 4)]

    def cq(k):
        return int(taps[k]) if 0 <= k < o4 else 0
    ng = o4 // 4
    cpl, cph = [], []
    for g in range(ng + 1):
        b = 4 * g
        c = [cq(b + d) for d in range(-3, 4)]          # c[3 + d] = tap b + d
        cpl.append([(c[3] & 0xFFFF) | ((c[4] & 0xFFFF) << 16), (c[5] & 0xFFFF) | ((c[6] & 0xFFFF) << 16),
                    (c[2] & 0xFFFF) | ((c[3] & 0xFFFF) << 16), (c[4] & 0xFFFF) | ((c[5] & 0xFFFF) << 16)])
        cph.append([sum((c[3 - r + j] & 0xFF) << (8 * j) for j in range(4)) for r in range(4)])
    out = []
    for s0 in range(o4, len(x), 4):                   # one chunk of four outputs
        acc = [half] * 4
        base = (s0 - o4) // 4                          # the chunk's first window group (four samples = two low words)
        cur = (lw[2 * base], lw[2 * base + 1])
        for j in range(ng):
            nxt = (lw[2 * (base + j + 1)], lw[2 * (base + j + 1) + 1])
            cf = cpl[j]
            acc[0] = dot2(cf[1], cur[1], dot2(cf[0], cur[0], acc[0]))
            acc[1] = dot2(cf[3], cur[1], dot2(cf[2], cur[0], acc[1]))
            acc[2] = dot2(cf[1], nxt[0], dot2(cf[0], cur[1], acc[2]))
            acc[3] = dot2(cf[3], nxt[0], dot2(cf[2], cur[1], acc[3]))
            cur = nxt
        cl = cpl[ng][2]                                # the last tap of the odd outputs
        acc[1] = dot2(cl, cur[0], acc[1])
        acc[3] = dot2(cl, cur[1], acc[3])
        ah = [0] * 4
        for j in range(ng + 1):
            a = hw[base + j]
            for r in range(4):
                ah[r] = dot4(cph[j][r], a, ah[r])
        out += [(acc[r] + (ah[r] << 16)) & M32 for r in range(4)]
    return out


@pytest.mark.parametrize("order", [1, 2, 3, 4, 5, 31, 32, 33, 64])
def test_fir_on_packed_planes_equals_the_wraparound_sum(order):
    rng = np.random.default_rng(order)
    o4 = (order + 3) & ~3
    taps = np.zeros(o4, np.int64)
    taps[o4 - order:] = rng.integers(-128, 128, order)
    taps[o4 - 1] = -128                                 # extreme taps
    for span in (1 << 15, 1 << 17, 1 << 22):            # within 16 bits, needs the high plane, the 24-bit limit of FIR_DOT
        x = rng.integers(-span, span, o4 + 64).astype(np.int64)
        x[o4 + 3] = span - 1
        x[o4 + 4] = -span
        x[o4 + 9] = 32767
        x[o4 + 10] = -32768
        x[o4 + 11] = 32768                              # low half -32768, high half 1
        half = 1 << 9
        assert fir_dot_model(list(map(int, x)), taps, half) == fir_reference(x, taps, half), (order, span)


def test_split_into_planes_is_exact_and_the_high_plane_vanishes_within_16_bits():
    for v in list(range(-70000, 70000, 7)) + [-(1 << 23), (1 << 23) - 1, 32767, 32768, -32768, -32769]:
        low = s16(v)
        high = (v - low) >> 16
        assert v == 65536 * high + low and -128 <= high <= 128
        assert (high == 0) == (-32768 <= v <= 32767)
        # the kernel's form: h = (y + 0x8000) >> 16
        assert high == (v + 0x8000) >> 16


def code_cost_var(u, k, rice):
    """the sample-dependent part of srla_coder.c:327-347 (residual_cost.hip code_cost_var)"""
    thr = 0 if rice else (2 << k)
    return max(u - thr, 0) >> k


def pair_cost(w, thr2, sh2, acc):
    lo = max((w & 0xFFFF) - (thr2 & 0xFFFF), 0) >> (sh2 & 15)
    hi = max((w >> 16) - (thr2 >> 16), 0) >> ((sh2 >> 16) & 15)
    return acc + lo + hi


def thr16(k, rice):
    t2 = 0 if rice else (2 << (k & 15))
    return 0xFFFF if k >= 16 else min(t2, 0xFFFF)


def test_code_bits_of_two_samples_per_instruction():
    values = sorted(set([0, 1, 2, 3, 255, 256, 32767, 32768, 65534, 65535] + [(1 << b) - 1 for b in range(1, 17)]
                        + [1 << b for b in range(16)] + list(np.random.default_rng(5).integers(0, 65536, 200))))
    for rice in (False, True):
        for k0 in range(32):
            for k1 in (k0, (k0 + 5) % 32):
                thr2 = thr16(k0, rice) | (thr16(k1, rice) << 16)
                sh2 = (k0 & 15) | ((k1 & 15) << 16)
                for a in values:
                    for b in (values[(values.index(a) * 7 + 3) % len(values)], 65535, 0):
                        want = code_cost_var(int(a), k0, rice) + code_cost_var(int(b), k1, rice)
                        assert pair_cost(int(a) | (int(b) << 16), thr2, sh2, 0) == want, (rice, k0, k1, a, b)


def test_first_stage_permutation():
    perm = lambda e: e ^ ((e >> 3) & 3)
    for m in (512, 1024, 2048, 4096):
        assert sorted(perm(e) for e in range(m)) == list(range(m))
        # stores: lane i of a group of eight writes output k of butterfly bf0 + i: element 4 bf + k -> eight 16-byte columns
        for bf0 in range(0, m // 4, 8):
            for k in range(4):
                cols = {perm(4 * (bf0 + i) + k) % 8 for i in range(8)}
                assert len(cols) == 8, (m, bf0, k)
        # second-stage loads: lanes of a quad stay within their quad, so a group of whole quads still covers whole quads
        for bf in range(0, m // 4):
            assert perm(bf) >> 2 == bf >> 2
        # and the kernel's split form of the store address
        for bf in range(m // 4):
            kx = (bf >> 1) & 3
            for k in range(4):
                assert perm(4 * bf + k) == 4 * bf + (k ^ kx)


# lane groups in which a wave64 ds_read_b128 / ds_write_b128 is served (MI355X_MICROARCH.md, LDS): loads 4 x 16 lanes over 16
# columns of 16 bytes, stores 8 x 8 contiguous lanes over 8 columns
READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def test_register_fed_first_stage_permutation():
    """fft_first_stage_regs: thread t writes its eight outputs 8t .. 8t+7 at e ^ ((e >> 3) & 7); the second stage reads
    bf + k m/4 at the same permutation of bf."""
    perm = lambda e: e ^ ((e >> 3) & 7)
    for m in (2048, 4096):
        assert sorted(perm(e) for e in range(m)) == list(range(m))
        for t0 in range(0, m // 8, 8):                      # a store group: eight consecutive threads, output j of each
            for j in range(8):
                assert len({perm(8 * (t0 + i) + j) % 8 for i in range(8)}) == 8
        for wave0 in range(0, m // 4, 64):                  # a wavefront of the second stage: butterflies wave0 .. wave0 + 63
            for k in range(4):
                for grp in READ_GROUPS:
                    cols = {(perm(wave0 + lane) + k * (m // 4)) % 16 for lane in grp}
                    assert len(cols) == 16, (m, wave0, k)
        # the thread that computes butterflies 2t and 2t+1 holds their inputs: chunks of four samples 4t + c nfft/4 are the
        # complex elements 2t + c m/4 and 2t + 1 + c m/4
        nfft = 2 * m
        for t in (0, 1, 7, m // 8 - 1):
            for c in range(4):
                samples = [4 * t + c * (nfft // 4) + i for i in range(4)]
                assert [x // 2 for x in samples] == [2 * t + c * (m // 4)] * 2 + [2 * t + 1 + c * (m // 4)] * 2


def sig_index(fl, s_plus_pad):
    s = 4 * fl
    return s_plus_pad if fl & 1 else s_plus_pad + (s_plus_pad // s) * 4


def group_offset(fl, d):
    if fl & 1:
        return d
    return d - ((fl - 1 - d) // fl) if d < 0 else d


@pytest.mark.parametrize("fl", [1, 2, 3, 4, 5, 6, 7, 8])
def test_group_offsets_of_the_padded_layout(fl):
    """own + group_offset(d) is the padded group index of the sample group d groups from a thread's first own group, for every
    d the FIR reaches (back over the 256-sample padding, forward within the thread's own run)"""
    s = 4 * fl
    padf = ((256 + s - 1) // s) * s
    for tid in (0, 1, 2, 63, 64, 255):
        s_base = s * tid
        own = sig_index(fl, padf + s_base) >> 2
        for d in range(-64, fl):
            sample = padf + s_base + 4 * d
            if sample < 0:
                continue
            assert sig_index(fl, sample) % 4 == 0
            assert own + group_offset(fl, d) == sig_index(fl, sample) >> 2, (fl, tid, d)
