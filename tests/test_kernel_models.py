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
import os

import numpy as np
import pytest
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


@pytest.mark.parametrize("fl,order", [(1, 1), (2, 17), (4, 64), (5, 33), (6, 64), (7, 55), (8, 64), (8, 16), (8, 255), (5, 128)])
def test_fir_as_toeplitz_tiles_on_the_matrix_pipe(fl, order):
    """residual_cost.hip: FIR_MFMA -- tools/probes/mfma_fir_model.py replays the kernel's operand indices (byte planes, the four
    shifted copies of the zero-padded tap string, k-blocks, tile groups' rows and columns, the lane permutation that returns every
    chunk to its owner) for a block of 1024 fl samples and compares with the direct wrap-around sum; the model also asserts that
    no lane reaches outside the tap string (MF_OFFZ, MF_TZB) -- round 6 widened the form from fl <= 4 to fl <= 8."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mfma_fir_model", os.path.join(REPO, "tools", "probes", "mfma_fir_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    src = open(os.path.join(REPO, "srla_amd", "csrc", "residual_cost.hip")).read()
    assert "#define MF_OFFZ %d " % m.MF_OFFZ in src and "#define MF_TZB  %d " % m.MF_TZB in src
    assert m.emul(fl, order)


# ------------------------------------------------------------------------------------------------------------------------
# autocorr.hip, round 6: the 8192-point class (M = 4096 complex points on eight wavefronts) on SIXTEEN sub-regions
# (fft_second_stage_regions, fft_subregions, spectrum_power_pass_subregions).  The model replays every LDS address of the kernel
# functions -- which slot every butterfly of every stage reads and writes -- on an array of complex numbers, with the
# reference's butterfly (fft.c:71-128) in Python floats, and compares bit for bit with the plain Stockham transform the
# older kernel form (fft_complex_lds) is; it also checks every 16-byte access against the lane groups LDS serves it in.
WRITE_GROUPS = [list(range(g, g + 8)) for g in range(0, 64, 8)]


def _tables(m, flag):
    """per stage (sub-size n) three tables of n/4 entries: w^p, w^2p, w^3p with w = exp(flag 2 pi i / n) (any values would do for an
    address model; these make the result an FFT, which the test also checks against numpy)"""
    tabs, n = [], m
    while n > 2:
        w1 = [complex(np.cos(2 * np.pi * p / n), flag * np.sin(2 * np.pi * p / n)) for p in range(n // 4)]
        tabs.append((w1, [a * a for a in w1], [a * a * a for a in w1]))
        n //= 4
    return tabs


def _bfly(a, b, c, d, w1, w2, w3, flag):
    apc, amc, bpd, bmd = a + c, a - c, b + d, b - d
    jbmd = complex(-bmd.imag, bmd.real) if flag < 0 else complex(bmd.imag, -bmd.real)
    return apc + bpd, w1 * (amc - jbmd), w2 * (apc - bpd), w3 * (amc + jbmd)


def _stockham(x, flag, tabs):
    m, x, s, st = len(x), list(x), 1, 0
    n = m
    while n > 2:
        y = [0j] * m
        for bf in range(m // 4):
            q, p = bf & (s - 1), bf // s
            o = _bfly(x[bf], x[bf + m // 4], x[bf + 2 * m // 4], x[bf + 3 * m // 4], tabs[st][0][p], tabs[st][1][p], tabs[st][2][p], flag)
            for k in range(4):
                y[4 * bf - 3 * q + k * s] = o[k]
        x, n, s, st = y, n // 4, s * 4, st + 1
    assert n == 1
    return x


def _conflict_free(addrs_by_lane, groups, columns):
    for grp in groups:
        lanes = [l for l in grp if addrs_by_lane[l] is not None]
        cols = {}
        for l in lanes:
            cols.setdefault(addrs_by_lane[l] % columns, set()).add(addrs_by_lane[l])
        if any(len(v) > 1 for v in cols.values()):
            return False
    return True


def _subregion_transform(lds, flag, tabs, need, inswz, stats):
    """fft_second_stage_regions + fft_subregions on `lds` (the first stage done), M = len(lds), 512 threads, R = 2."""
    m = len(lds)
    qm, sq, ntk, r_count = m // 4, m // 16, 512, 2
    swz = lambda e: e ^ ((e >> 3) & 3)
    swz16 = lambda e: e ^ ((e >> 4) & 7)
    prune = need is not None
    need = m if need is None else need
    # ---- second stage, in place inside every region
    k_on = [True, (not prune) or 4 < need, (not prune) or 8 < need, (not prune) or 12 < need]
    for r in range(r_count):
        for wave0 in range(0, ntk, 64):
            addr = [None] * 64
            for lane in range(64):
                u = wave0 + lane + r * ntk
                rho, j = u // sq, u % sq
                if prune and not rho < need:
                    continue
                base = rho * qm + (swz16(j) if inswz else j)
                addr[lane] = base
                o = _bfly(lds[base], lds[base + sq], lds[base + 2 * sq], lds[base + 3 * sq], tabs[1][0][j], tabs[1][1][j], tabs[1][2][j], flag)
                for k in range(4):
                    if k_on[k]:
                        lds[base + k * sq] = o[k]
            stats["second_read"] &= _conflict_free(addr, READ_GROUPS, 16)
            stats["second_write"] &= _conflict_free(addr, WRITE_GROUPS, 8)
    # ---- stages 3 .. 6, every wavefront on its own two sub-regions (loads of a stage before its stores, per wavefront)
    bpg = sub4 = m // 64
    for wave in range(8):
        for st in range(2, 6):
            s = 1 << (2 * st)
            sl = s >> 4
            k_on = [True, (not prune) or s < need, (not prune) or 2 * s < need, (not prune) or 3 * s < need]
            work = []
            for r in range(r_count):
                rd = [[None] * 64 for _ in range(4)]
                for lane in range(64):
                    u = lane + 64 * r
                    g, jj = wave * 2 + u // bpg, u % bpg
                    ql, qres = jj & (sl - 1), 4 * (g & 3) + (g >> 2)
                    if prune and not 16 * ql + qres < need:
                        continue
                    p = jj // sl
                    if inswz and st == 2:
                        pe = jj ^ ((jj >> 4) & 3)
                        src = [g * sq + pe, g * sq + (pe ^ 4) + sub4, g * sq + pe + 2 * sub4, g * sq + (pe ^ 4) + 3 * sub4]
                        assert src == [g * sq + swz16(jj + k * sub4) for k in range(4)]
                    else:
                        b0 = g * sq + (swz(jj) if st == 3 else jj)
                        src = [b0 + k * sub4 for k in range(4)]
                    for k in range(4):
                        rd[k][lane] = src[k]
                    work.append((lane, r, g, jj, ql, _bfly(*(lds[a] for a in src), tabs[st][0][p], tabs[st][1][p], tabs[st][2][p], flag)))
                for k in range(4):
                    stats["sub_read_%d" % st] &= _conflict_free(rd[k], READ_GROUPS, 16)
            for r in range(r_count):
                wr = [[None] * 64 for _ in range(4)]
                for lane, rr, g, jj, ql, o in work:
                    if rr != r:
                        continue
                    for k in range(4):
                        if not k_on[k]:
                            continue
                        dst = g * sq + 4 * jj + (k ^ ((jj >> 1) & 3)) if st == 2 else g * sq + 4 * jj - 3 * ql + k * sl
                        wr[k][lane] = dst
                        lds[dst] = o[k]
                for k in range(4):
                    stats["sub_write_%d" % st] &= _conflict_free(wr[k], WRITE_GROUPS, 8)


def test_sixteen_sub_regions_of_the_eight_wavefront_class():
    m, rng = 4096, np.random.default_rng(6)
    qm, sq = m // 4, m // 16
    slot = lambda e: (e & 3) * qm + ((e >> 2) & 3) * sq + (e >> 4)
    swz16 = lambda e: e ^ ((e >> 4) & 7)
    assert sorted(slot(e) for e in range(m)) == list(range(m)) and sorted(swz16(e) for e in range(m)) == list(range(m))
    x = [complex(a, b) for a, b in rng.standard_normal((m, 2))]
    stats = {}
    for key in ["second_read", "second_write", "first_inv_read", "first_inv_write"] + ["sub_%s_%d" % (w, st) for w in ("read", "write") for st in range(2, 6)]:
        stats[key] = True
    # ---- forward: first stage from registers (fft_first_stage_regs_regions: outputs k of butterfly bf at k QM + bf), the rest on the layout
    tf = _tables(m, -1)
    lds = [0j] * m
    for bf in range(qm):
        o = _bfly(x[bf], x[bf + qm], x[bf + 2 * qm], x[bf + 3 * qm], tf[0][0][bf], tf[0][1][bf], tf[0][2][bf], -1)
        for k in range(4):
            lds[k * qm + bf] = o[k]
    _subregion_transform(lds, -1, tf, None, False, stats)
    want = _stockham(x, -1, tf)
    assert all(lds[slot(e)] == want[e] for e in range(m))                                  # the same bits, element by element
    assert np.allclose(want, np.fft.fft(np.array(x)), rtol=0, atol=1e-9)                   # (and it is the transform)
    # ---- the spectrum pass's pairs: every bin 1 .. M/2 once with its partner, both read where the forward transform left them,
    # contiguous over the lanes; the stores 16 bins apart on eight columns
    seen = []
    for it in range(4):
        for wave0 in range(0, 512, 64):
            ra, rb, wa, wb = [None] * 64, [None] * 64, [None] * 64, [None] * 64
            for lane in range(64):
                P = wave0 + lane + it * 512
                cls, tt = P // (m // 32), P % (m // 32)
                rho, sg, first = cls >> 2, cls & 3, 1 if cls == 0 else 0
                i = rho + 4 * sg + 16 * (tt + first)
                rho2, sg2 = (4 - rho) & 3, (3 - sg) if rho else ((4 - sg) & 3)
                ra[lane] = rho * qm + sg * sq + tt + first
                rb[lane] = rho2 * qm + sg2 * sq + (sq - 1 - tt)
                assert ra[lane] == slot(i) and rb[lane] == slot((m - i) % m if i != m - i else i), (P, i)
                wa[lane], wb[lane] = (None if i == m - i else swz16(i)), swz16(m - i)
                seen.append(i)
            assert _conflict_free(ra, READ_GROUPS, 16) and _conflict_free(rb, READ_GROUPS, 16)
            assert _conflict_free(wa, WRITE_GROUPS, 8) and _conflict_free(wb, WRITE_GROUPS, 8)
    assert sorted(seen) == list(range(1, m // 2 + 1))
    # ---- inverse: input in natural order at fft_swz16, first stage in place (fft_first_stage_regions<.., 2>), pruned to `need` outputs
    ti = _tables(m, 1)
    z = [complex(a, b) for a, b in rng.standard_normal((m, 2))]
    want = _stockham(z, 1, ti)
    for need in (None, 132, 33, 5, 1):
        lds = [0j] * m
        for e in range(m):
            lds[swz16(e)] = z[e]
        for r in range(2):
            for wave0 in range(0, 512, 64):
                addr = [None] * 64
                for lane in range(64):
                    bf = wave0 + lane + r * 512
                    b0 = swz16(bf)
                    addr[lane] = b0
                    o = _bfly(lds[b0], lds[b0 + qm], lds[b0 + 2 * qm], lds[b0 + 3 * qm], ti[0][0][bf], ti[0][1][bf], ti[0][2][bf], 1)
                    for k in range(4):
                        lds[b0 + k * qm] = o[k]
                    assert [b0 + k * qm for k in range(4)] == [swz16(bf + k * qm) for k in range(4)]
                stats["first_inv_read"] &= _conflict_free(addr, READ_GROUPS, 16)
                stats["first_inv_write"] &= _conflict_free(addr, WRITE_GROUPS, 8)
        _subregion_transform(lds, 1, ti, need, True, stats)
        assert all(lds[slot(e)] == want[e] for e in range(m if need is None else need)), need
    # ---- bank conflicts: which accesses are free of them (the rest are the known two-way cases of the strided stores)
    free = {k for k, v in stats.items() if v}
    assert {"second_read", "second_write", "first_inv_write", "sub_read_2", "sub_read_4", "sub_read_5", "sub_write_2", "sub_write_4", "sub_write_5"} <= free, sorted(stats.items())


def test_level_sums_through_lds():
    """residual_cost.hip (round 6): a wavefront lays its lanes' eleven level sums out as rows of RC_RED_ROW = 68 words; lane 4 l + g
    adds up words 16 g .. 16 g + 15 of row l with four 16-byte loads.  Every (level, lane) word is added exactly once, the loads are
    16-byte aligned, and load i of the lanes of every group a ds_read_b128 is served in lands on sixteen different columns."""
    row = 68
    src = open(os.path.join(REPO, "srla_amd", "csrc", "residual_cost.hip")).read()
    assert "#define RC_RED_ROW %du" % row in src
    seen = {}
    for lane in range(44):
        lv, seg = lane >> 2, lane & 3
        for i in range(4):
            base = lv * row + 16 * seg + 4 * i
            assert base % 4 == 0
            for w in range(4):
                seen[(lv, 16 * seg + 4 * i + w)] = seen.get((lv, 16 * seg + 4 * i + w), 0) + 1
                assert base + w == lv * row + (16 * seg + 4 * i + w)           # row lv, word = the lane that stored it
    assert seen == {(l, lane): 1 for l in range(11) for lane in range(64)}
    for i in range(4):
        for grp in READ_GROUPS:
            cols = [((lane >> 2) * row + 16 * (lane & 3) + 4 * i) // 4 % 16 for lane in grp if lane >> 2 <= 10]
            assert len(set(cols)) == len(cols), (i, grp)
