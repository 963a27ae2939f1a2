/*
 * solve.hip -- what stands between the lags and the taps: srla_pitch_solve (long-term predictor: pitch scan + 3 x 3 solve, one lane
 * per item), the Levinson-Durbin chain (srla_lpc_errvars(_lean) / srla_lpc_recursion: one lane per item; srla_order_select: one
 * wavefront per item; srla_lpc_taps / srla_lpc_quantize: one lane per item) and the SVR refinement (srla_svr_refine(_big),
 * srla_lpc_quantize_ws).  Serial fp64 recursions, so one LANE per item wherever the work is serial.  DESIGN.md 3.2, 3.7.
 */
#include "kernels_common.h"

/* ------------------------------------------------------------ order choice (H2: libm) ----- */
/* srla_encoder.c:873-885 */
/* logscale: 1.0 (exact) in production; the tie tests falsify the device's log with it (SrlaJobParams) */
__device__ __forceinline__ double geometric_entropy(double mean_abs, uint32_t bps, double logscale)
{
    const double intmean = mean_abs * (double)(1 << (bps - 1));
    const double rho = 1.0 / (1.0 + intmean);
    const double invrho = 1.0 - rho;
    if (mean_abs < 1e-16) return 0.0;
    return -(invrho * ((log(invrho) * logscale) * 1.4426950408889634) + rho * ((log(rho) * logscale) * 1.4426950408889634)) / rho;
}

/* correctly rounded x^-0.5 for the 3x3 LTP solve (lpc.c:591 uses pow(sum, -0.5)) */
__device__ __forceinline__ double inv_sqrt_cr(double x)
{
    const double s = sqrt(x);
    const double s_lo = __builtin_fma(-s, s, x) / (2.0 * s);          /* sqrt(x) = s + s_lo  */
    const double r = 1.0 / s;
    const double e = __builtin_fma(-s, r, 1.0);                        /* 1 - s*r             */
    return r + r * (e - s_lo * r);
}

/* ================================================================================================
 * K2p: srla_pitch_solve -- one lane per item (lpc.c:1473-1649, srla_encoder.c:1031-1047)
 * ============================================================================================== */
/* Near-ties (H2): an item whose decision hangs on a libm function the device cannot reproduce bit for bit is appended to
 * the job's tie list -- ties[0] = count, ties[1 + k] = item | kind << 30 (kind 0: LPC order, 1: LTP taps, 2: SVR refinement) -- and, for LTP items, the
 * numbers the host needs to redo the 3x3 solve with its own pow() go to tie_data[8 k ..]. */
__device__ __forceinline__ uint32_t tie_append(uint32_t *__restrict__ ties, uint32_t item, uint32_t kind)
{
    const uint32_t k = atomicAdd(&ties[0], 1u);
    ties[1u + k] = item | (kind << 30);
    return k;
}

/* One step of the pitch scan (lpc.c:1486-1527) as a state machine over the lag index j = 8 .. 263, so that every lane of a
 * wavefront walks the lags in the same order with loads that do not depend on the data.  The reference's loops: from i, the
 * first upward zero crossing `start` (262 when there is none); from start + 1 the first downward one `end` (at most 261, or
 * start + 1 when that is larger); the largest local maximum above zero of [start, end] is a candidate; on with i = end + 1
 * while i < 262 and fewer than 20 candidates.  rm, rc, rn = R(j - 1), R(j), R(j + 1); returns true when a candidate is
 * complete (value *cand_val at *cand_at). */
struct PitchScan {
    uint32_t start, peak_at, ncand;
    double peak;
    bool in_seg, done;
};
__device__ __forceinline__ bool pitch_scan_step(PitchScan &st, const uint32_t j, const double rm, const double rc, const double rn,
                                                uint32_t *cand_at, double *cand_val)
{
    bool cand = false;
    if (!st.done) {
        if (!st.in_seg) {
            /* (j == 262: no crossing in [i, 261], the reference goes on with start = 262, end = 263) */
            if (j >= SRLA_LTP_MAX_PERIOD || (rm < 0.0 && rc > 0.0)) { st.start = j; st.in_seg = true; st.peak = 0.0; st.peak_at = 0; }
        }
        if (st.in_seg) {
            if (rc > rm && rc > rn && rc > st.peak) { st.peak = rc; st.peak_at = j; }
            if (j > st.start && (j >= SRLA_LTP_MAX_PERIOD - 1u || (rc > 0.0 && rn < 0.0))) {
                if (st.peak_at != 0) { cand = true; *cand_at = st.peak_at; *cand_val = st.peak; st.ncand++; }
                st.in_seg = false;
                if (j + 1u >= SRLA_LTP_MAX_PERIOD || st.ncand >= 20u) st.done = true;
            }
        }
    }
    return cand;
}

__global__ __launch_bounds__(WAVE) void srla_pitch_solve(SrlaJobParams jp, const SrlaItemDesc *__restrict__ items,
                                                         const double *__restrict__ lags_ws,
                                                         SrlaItemResult *__restrict__ results,
                                                         const uint32_t *__restrict__ select, uint32_t round,
                                                         uint32_t *__restrict__ ties, double *__restrict__ tie_data)
{
    NARROW_KERNEL_PRIORITY();
    /* One LANE per item, 64 items per wavefront.  The lag table is [lag][item], so the lanes of a wavefront read one lag of
     * their 64 items with one coalesced load, and because the scan visits the lags in a fixed order (pitch_scan_step) the
     * loads run ahead of the arithmetic instead of forming a chain of data-dependent round trips (round 2: the lags of 8 items
     * staged in LDS and scanned by 8 lanes: 0.4 ms per job alone, 0.77 ms in flight at -V 2 -P 3).  Two passes over the lags:
     * the first finds the largest candidate peak, the second the first candidate within 0.9 of it (lpc.c:1540-1546) -- no
     * candidate list, whose dynamic indexing would live in scratch memory. */
    const size_t stride = jp.num_items;
    const uint32_t idx = blockIdx.x * WAVE + threadIdx.x;
    if (idx >= jp.num_items) return;
    if (select != nullptr && select[idx] != round) return;   /* chain mode: only the items whose LTP lags this round produced */
    const double *lg = lags_ws + idx;
    /* words 263 and 264 of the reference's lag buffer are never written: zero (fresh pages) */
    auto R = [&](uint32_t j) -> double { return (j < SRLA_LTP_LAGS) ? lg[(size_t)j * stride] : 0.0; };
    SrlaItemResult *out = &results[idx];
    const double r0 = R(0);
    uint32_t period = 0;
    if (!(fabs(r0) <= (double)FLT_MIN)) {
        double best = 0.0;
        uint32_t ncand = 0;
        for (int pass = 0; pass < 2; pass++) {
            PitchScan st;
            st.start = 0; st.peak_at = 0; st.ncand = 0; st.peak = 0.0; st.in_seg = false; st.done = false;
            double rm = R(SRLA_LTP_MIN_PERIOD - 1u), rc = R(SRLA_LTP_MIN_PERIOD);
            bool found = false;
            /* j = 8 .. 263 in blocks of 8: the eight loads of a block are issued together */
            for (uint32_t j0 = SRLA_LTP_MIN_PERIOD; j0 < SRLA_LTP_MAX_PERIOD + 2u; j0 += 8u) {
                double nx[8];
#pragma unroll
                for (int u = 0; u < 8; u++) nx[u] = R(j0 + 1u + (uint32_t)u);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    uint32_t at = 0; double val = 0.0;
                    if (pitch_scan_step(st, j0 + (uint32_t)u, rm, rc, nx[u], &at, &val)) {
                        if (pass == 0) { if (val > best) best = val; }
                        else if (!found && val >= 0.9 * best) { period = at; found = true; }
                    }
                    rm = rc; rc = nx[u];
                }
                /* (a wavefront leaves the loop when all of its lanes are done) */
                if (st.done || found) break;
            }
            if (pass == 0) {
                ncand = st.ncand;
                if (ncand == 0 || best < 0.1 * r0) break;          /* lpc.c:1530-1537 */
            }
        }
        if (period < (jp.ltp_order / 2) + 1) period = 0;
    }
    uint32_t flags = 0;
    int32_t q[3] = { 0, 0, 0 };
    if (period > 0) {
        const int dim = (int)jp.ltp_order;
        const double rl[3] = { r0 * (1.0 + 1e-5), R(1), R(2) };
        double am[3][3], inv_diag[3], xs[3];
        bool ok = true;
        for (int j = 0; j < dim; j++) for (int k = j; k < dim; k++) am[j][k] = am[k][j] = rl[k - j];
        for (int i2 = 0; i2 < dim && ok; i2++) {
            double sum = am[i2][i2];
            for (int k = i2 - 1; k >= 0; k--) sum -= am[i2][k] * am[i2][k];
            if (sum <= 0.0) { ok = false; break; }
            inv_diag[i2] = inv_sqrt_cr(sum);                     /* lpc.c:591: pow(sum, -0.5) of the platform libm */
            for (int j = i2 + 1; j < dim; j++) {
                sum = am[i2][j];
                for (int k = i2 - 1; k >= 0; k--) sum -= am[i2][k] * am[j][k];
                am[j][i2] = sum * inv_diag[i2];
            }
        }
        if (!ok) { flags |= SRLA_ITEM_LTP_FAIL; period = 0; }
        else {
            double b[3] = { 0.0, 0.0, 0.0 };
            for (int i2 = 0; i2 < dim; i2++) {
                const uint32_t j = period - jp.ltp_order / 2 + (uint32_t)i2;
                b[i2] = (j == 0) ? rl[0] : R(j);
            }
            for (int i2 = 0; i2 < dim; i2++) {
                double sum = b[i2];
                for (int j = i2 - 1; j >= 0; j--) sum -= am[i2][j] * xs[j];
                xs[i2] = sum * inv_diag[i2];
            }
            for (int i2 = dim - 1; i2 >= 0; i2--) {
                double sum = xs[i2];
                for (int j = i2 + 1; j < dim; j++) sum -= am[j][i2] * xs[j];
                xs[i2] = sum * inv_diag[i2];
            }
            for (int i2 = 0; i2 < dim; i2++) {
                const double scaled = xs[i2] * 32.0 + jp.tie_ltpbias;   /* bias: 0.0 in production (tie tests) */
                const double fr = fabs(scaled) + 0.5;
                if (fabs(fr - floor(fr + 0.5)) < jp.tie_ltp && fabs(scaled) < 40.0) flags |= SRLA_ITEM_LTP_TIE;
                int32_t c = cvt_i32_as_x86(round_half_away(scaled));
                c = (c < -32) ? -32 : ((c > 31) ? 31 : c);
                q[i2] = c;
            }
            for (int i2 = 0; i2 < dim / 2; i2++) { const int32_t t = q[i2]; q[i2] = q[dim - 1 - i2]; q[dim - 1 - i2] = t; }
            const uint32_t forced = items[idx].forced_ltp;
            if (forced >> 31) {
                /* the host has redone the solve with its libm (host_ties.cpp) */
                for (int i2 = 0; i2 < 3; i2++) q[i2] = ((int32_t)((forced >> (6 * i2)) << 26)) >> 26;
                flags &= ~SRLA_ITEM_LTP_TIE;
            } else if ((flags & SRLA_ITEM_LTP_TIE) && ties != nullptr) {
                const uint32_t k = tie_append(ties, idx, 1u);
                double *td = tie_data + 8u * (size_t)k;
                td[0] = r0; td[1] = R(1); td[2] = R(2);
                td[3] = R(period - 1); td[4] = R(period); td[5] = R(period + 1);
                td[6] = (double)period;
                td[7] = (double)(((uint32_t)q[0] & 63u) | (((uint32_t)q[1] & 63u) << 6) | (((uint32_t)q[2] & 63u) << 12));
            }
        }
    }
    out->ltp_period = period;
    out->ltp_coef[0] = (period > 0) ? q[0] : 0;
    out->ltp_coef[1] = (period > 0) ? q[1] : 0;
    out->ltp_coef[2] = (period > 0) ? q[2] : 0;
    if (flags) out->flags |= flags;
}

/* ================================================================================================
 * K2: Levinson-Durbin / order choice / quantiser.  Three kernels:
 *   srla_lpc_recursion<L>  one LANE per item: the full recursion (lpc.c:379-441) with the gamma dot product
 *                          summed in index order; a[] and r[] live in LDS column-major ([i][lane]) so the 64
 *                          recursions of a wavefront run without bank conflicts; loops are unrolled so that
 *                          several LDS loads are in flight per dependent add.  Writes the (uncompensated)
 *                          error variance of every order.
 *   srla_order_select      one WAVE per item, lane = order: window compensation (lpc.c:490-497), code-length
 *                          estimate (srla_encoder.c:934-957) and its first strict minimum -- fully parallel.
 *   srla_lpc_quantize<L>   one LANE per item: recursion up to the chosen order, 8-bit quantiser with error
 *                          feedback (lpc.c:1341-1405), tap order reversal (srla_encoder.c:1104-1108), Huffman
 *                          cost plain vs pair-summed (srla_encoder.c:1141-1174).
 * ============================================================================================== */
#define A_(i) a[(size_t)(i) * L + lane]
#define R_(i) r[(size_t)(i) * L + lane]

/* recursion up to `upto` (>= 1); err_out (may be null) receives the error variance of orders 1..upto at
 * err_out[order * stride]; on return A_(1..upto) is the predictor of order `upto` */
template <int L>
__device__ __forceinline__ void levinson_lane(double *a, double *r, uint32_t lane, double r0, uint32_t upto,
                                              double *err_out, size_t stride)
{
    const double a1 = -R_(1) / r0;
    A_(0) = 1.0; A_(1) = a1; A_(2) = 0.0;
    double e = r0 + R_(1) * a1;
    if (err_out) err_out[stride] = e;
    for (uint32_t k = 1; k < upto; k++) {
        /* gamma = sum_{i=0..k} a[i] * r[k+1-i], accumulated in index order (lpc.c:420-423) */
        double gamma = 0.0;
        uint32_t i = 0;
        for (; i + 4 <= k + 1; i += 4) {
            const double x0 = A_(i), x1 = A_(i + 1), x2 = A_(i + 2), x3 = A_(i + 3);
            const double y0 = R_(k + 1 - i), y1 = R_(k - i), y2 = R_(k - 1 - i), y3 = R_(k - 2 - i);
            const double p0 = x0 * y0, p1 = x1 * y1, p2 = x2 * y2, p3 = x3 * y3;
            gamma += p0; gamma += p1; gamma += p2; gamma += p3;
        }
        for (; i < k + 1; i++) gamma += A_(i) * R_(k + 1 - i);
        gamma /= -e;
        e = e * (1.0 - gamma * gamma);
        /* a'[i] = a[i] + gamma * a[k+1-i] for i = 0..k+1, pairwise in place (lpc.c:430-433) */
        uint32_t lo = 0, hi = k + 1;
        for (; lo + 1 < hi - 1; lo += 2, hi -= 2) {
            const double al0 = A_(lo), ah0 = A_(hi), al1 = A_(lo + 1), ah1 = A_(hi - 1);
            A_(lo) = al0 + gamma * ah0; A_(hi) = ah0 + gamma * al0;
            A_(lo + 1) = al1 + gamma * ah1; A_(hi - 1) = ah1 + gamma * al1;
        }
        for (; lo <= hi; lo++, hi--) {
            const double al = A_(lo), ah = A_(hi);
            A_(lo) = al + gamma * ah;
            if (lo != hi) A_(hi) = ah + gamma * al;
            if (hi == 0) break;
        }
        A_(k + 2) = 0.0;
        if (err_out) err_out[(size_t)(k + 1) * stride] = e;
    }
}

template <int L>
__global__ __launch_bounds__(WAVE) void srla_lpc_recursion(SrlaJobParams jp, const double *__restrict__ lags_ws,
                                                           double *__restrict__ err_ws, const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = blockIdx.x * L + lane;
    if (lane >= L || idx >= jp.num_items) return;   /* no barriers below: lanes are independent */
    if (sel != nullptr && sel[idx] != sel_round) return;   /* chain mode with SVR on: the solve chain round by round */
    const uint32_t p = jp.max_order;
    double *a = (double *)lds;                        /* a[i * L + lane], i < p + 2 */
    double *r = a + (size_t)(p + 2) * L;              /* r[i * L + lane], i < p + 1 */
    const size_t stride = jp.num_items;
    for (uint32_t i = 0; i <= p; i++) R_(i) = lags_ws[(size_t)i * stride + idx];
    const double r0 = R_(0) * (1.0 + 1e-5);          /* ridge, lpc.c:483 */
    double *err = err_ws + idx;
    err[0] = r0;
    if (fabs(r0) < (double)FLT_EPSILON) {
        for (uint32_t o = 1; o <= p; o++) err[(size_t)o * stride] = r0;   /* lpc.c:395-405 */
        return;
    }
    levinson_lane<L>(a, r, lane, r0, p, err, stride);
}

__global__ __launch_bounds__(WAVE) void srla_order_select(
    SrlaJobParams jp, const SrlaItemDesc *__restrict__ items, const SrlaGeom *__restrict__ geoms,
    const double *__restrict__ err_ws, SrlaItemResult *__restrict__ results, double *__restrict__ dbg,
    uint32_t *__restrict__ ties, const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    NARROW_KERNEL_PRIORITY();
    /* (neighbouring items share the 64-byte sectors of the [order][item] table: they go to the same XCD, hence the same L2 --
     * dealt round robin over the XCDs every sector was fetched from HBM eight times, 255 MB per launch at -V 2) */
    const uint32_t idx = xcd_position(blockIdx.x, jp.num_items), lane = threadIdx.x;
    if (idx >= jp.num_items) return;
    if (sel != nullptr && sel[idx] != sel_round) return;
    const SrlaItemDesc it = items[idx];
    const double comp = geoms[it.geom].welch_comp;
    const uint32_t p = jp.max_order, n = it.n, bps = jp.bits_per_sample;
    const size_t stride = jp.num_items;
    double *dbg_item = dbg ? dbg + (size_t)idx * SRLA_DBG_STRIDE : nullptr;
    if (dbg_item && lane == 0) dbg_item[SRLA_DBG_ERRVARS] = err_ws[idx] * comp;
    /* first strict minimum == the lowest order among the smallest lengths; lengths that are NaN or not
     * below FLT_MAX are never chosen (srla_encoder.c:938-950) */
    const double kInf = __builtin_inf();
    double best = kInf, second = kInf;
    uint32_t best_order = 0;
    for (uint32_t base = 1; base <= p; base += WAVE) {
        const uint32_t o = base + lane;
        double len = kInf;
        if (o <= p) {
            const double ev = err_ws[(size_t)o * stride + idx] * comp;          /* lpc.c:490-497 */
            const double mabse = 2.0 * sqrt(ev / 2.0);
            double l = geometric_entropy(mabse, bps, jp.tie_logscale) * (double)n;
            l += (double)(8u * o);
            if (dbg_item) { dbg_item[SRLA_DBG_ERRVARS + o] = ev; dbg_item[SRLA_DBG_LENS + o] = l; }
            if (l < (double)FLT_MAX) len = l;
        }
        /* wave reduction of (len, order) in lexicographic order, plus the runner-up length */
        double v = len, v2 = kInf; uint32_t vo = o;
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(v, off, WAVE), ov2 = __shfl_xor(v2, off, WAVE);
            const uint32_t oo = __shfl_xor(vo, off, WAVE);
            const bool other_wins = (ov < v) || (ov == v && oo < vo);
            const double loser = other_wins ? v : ov;
            double s2 = (ov2 < v2) ? ov2 : v2;
            s2 = (loser < s2) ? loser : s2;
            if (other_wins) { v = ov; vo = oo; }
            v2 = s2;
        }
        if (v < best) { second = (best < v2) ? best : v2; best = v; best_order = vo; }
        else { const double c = (v < v2) ? v : v2; second = (c < second) ? c : second; }
    }
    if (lane == 0) {
        uint32_t order = (best < kInf) ? best_order : 0u;
        uint32_t flags = 0;
        if (jp.order_fixed) order = p;
        else if (it.forced_order < 0 && order != 0 && (second - best) <= jp.tie_rel * fabs(best) + 1e-9) {
            flags |= SRLA_ITEM_ORDER_TIE;
            if (ties) (void)tie_append(ties, idx, 0u);
        }
        if (it.forced_order >= 0) order = (uint32_t)it.forced_order;
        results[idx].lpc_order = order;
        if (flags) results[idx].flags |= flags;
    }
}

/* shared tail of the quantiser kernels: cf(i) = tap i of the chosen predictor */
template <typename CF, typename QS, typename QL>
__device__ __forceinline__ void quantize_and_price(uint32_t order, bool silent, CF cf, QS qstore, QL qload,
                                                   const uint8_t *__restrict__ huff_len, SrlaItemResult *out,
                                                   const double band = 0.0, bool *near_boundary = nullptr)
{
    /* band > 0: *near_boundary is set when the outcome hangs on the last bits of a tap -- the largest tap within `band` (relative) of a
     * power of two (the shared shift), or a scaled tap plus the error fed back within `band` of a rounding boundary */
    uint32_t rshift = 0, use_sum = 0, coef_bits = 0;
    if (order > 0) {
        /* 8-bit quantisation with error feedback from the last tap (lpc.c:1341-1405) */
        double maxabs = 0.0;
        if (!silent) for (uint32_t i = 0; i < order; i++) { const double v = fabs(cf(i)); if (maxabs < v) maxabs = v; }
        if (maxabs <= 0.0078125) {
            rshift = 8;
            for (uint32_t i = 0; i < order; i++) qstore(i, 0);
        } else {
            int ndigit;
            const double mant = frexp(maxabs, &ndigit);
            rshift = (uint32_t)(7 - ndigit);
            if (rshift >= 16u) rshift = 15u;
            const double scale = __builtin_ldexp(1.0, (int)rshift);
            bool near = band > 0.0 && (mant - 0.5 < band || 1.0 - mant < band);
            double qerr = 0.0;
            for (int i = (int)order - 1; i >= 0; i--) {
                qerr += cf((uint32_t)i) * scale;
                if (band > 0.0) { const double a = fabs(qerr), fr = a - floor(a); if (fabs(fr - 0.5) < band) near = true; }
                int32_t qq = cvt_i32_as_x86(round_half_away(qerr));
                if (qq >= 128) qq = 127; else if (qq < -128) qq = -128;
                qerr -= (double)qq;
                qstore(order - 1 - (uint32_t)i, qq);        /* reversed: oldest sample first (srla_encoder.c:1104) */
            }
            if (near_boundary != nullptr && near) *near_boundary = true;
        }
        /* Huffman cost plain vs pair-summed (srla_encoder.c:1141-1174) */
        uint32_t plain = 0, summed = 0, overflow = 0;
        int32_t prevq = 0;
        for (uint32_t k = 0; k < order; k++) {
            const int32_t c = qload(k);
            out->lpc_coef[k] = (int8_t)c;
            plain += huff_len[zigzag32(c)];
            if (k == 0) summed += huff_len[zigzag32(c)];
            else {
                const uint32_t z = zigzag32(c + prevq);
                if (z >= 256u) overflow = 1; else summed += huff_len[256 + z];
            }
            prevq = c;
        }
        use_sum = (overflow == 0 && (order == 1 || summed < plain)) ? 1u : 0u;
        coef_bits = use_sum ? summed : plain;
    }
    out->lpc_rshift = rshift;
    out->use_sum = use_sum;
    out->pad[0] = coef_bits;
}

/* The solve chain of orders 8 .. 64 as three launches, each with the parallelism its part has:
 *   srla_lpc_errvars<P>   one LANE per item, registers: the recursion alone -- error variance of every order (err_ws) and the
 *                         reflection coefficient of every step (gamma_ws; row 0 holds a[1] of order 1).
 *   srla_order_select     one WAVE per item, lane = order: the 64 code-length estimates (a square root, two divisions and two
 *                         logarithms each -- more than half of the one-pass kernel's instructions, and there on the serial
 *                         chain of a single lane) side by side.
 *   srla_lpc_taps<P>      one LANE per item: the predictor of the chosen order rebuilt from the stored reflection coefficients
 *                         (lpc.c:430-433, the update alone: the same products and sums on the same operands, no dot
 *                         products), 8-bit quantiser, tap cost.
 * (Round 2's single launch for all of it held a whole SIMD's registers -- 262 per lane -- for 0.11 ms on its own and 0.17-0.24 ms
 * beside the wide kernels; the three together hold far less for far shorter.) */
template <int P>
__global__ __launch_bounds__(WAVE) void srla_lpc_errvars(SrlaJobParams jp, const double *__restrict__ lags_ws, double *__restrict__ err_ws,
                                                         double *__restrict__ gamma_ws, const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    NARROW_KERNEL_PRIORITY();
    const uint32_t idx = blockIdx.x * WAVE + threadIdx.x;
    if (idx >= jp.num_items) return;
    if (sel != nullptr && sel[idx] != sel_round) return;
    const size_t stride = jp.num_items;
    double r[P + 1];
#pragma unroll
    for (int i = 0; i <= P; i++) r[i] = lags_ws[(size_t)i * stride + idx];
    const double r0 = r[0] * (1.0 + 1e-5);                   /* ridge, lpc.c:483 */
    double *err = err_ws + idx, *gam = gamma_ws + idx;
    err[0] = r0;
    if (fabs(r0) < (double)FLT_EPSILON) {
        /* lpc.c:395-405: every error variance is r0, every predictor zero */
        for (uint32_t o = 1; o <= (uint32_t)P; o++) { err[(size_t)o * stride] = r0; gam[(size_t)(o - 1) * stride] = 0.0; }
        return;
    }
    double a[P + 2];
    const double a1 = -r[1] / r0;
    a[0] = 1.0; a[1] = a1; a[2] = 0.0;
    double e = r0 + r[1] * a1;
    err[stride] = e;
    gam[0] = a1;
#pragma unroll
    for (int k = 1; k < P; k++) {
        double gamma = 0.0;
#pragma unroll
        for (int i = 0; i <= k; i++) gamma += a[i] * r[k + 1 - i];          /* index order, lpc.c:420-423 */
        gamma /= -e;
        e = e * (1.0 - gamma * gamma);
#pragma unroll
        for (int i = 0; i <= (k + 1) / 2; i++) {
            const int j = k + 1 - i;
            const double ai = a[i], aj = a[j];
            a[i] = ai + gamma * aj;
            if (i != j) a[j] = aj + gamma * ai;
        }
        a[k + 2] = 0.0;
        err[(size_t)(k + 1) * stride] = e;
        gam[(size_t)k * stride] = gamma;
    }
}

/* srla_lpc_errvars with a footprint that fits beside the wide kernels.  The register form of order 64 holds 348 registers per
 * lane -- two thirds of a SIMD's file.  Beside srla_residual_cost (five wavefronts of 96 registers per SIMD, a fresh workgroup
 * taking every slot that frees up) such a wavefront finds no room until the wide launch drains: in a kernel trace of a 600 s
 * encode srla_lpc_errvars<64> took 31 us when it started just ahead of a wide kernel and 230-470 us otherwise, it ended exactly
 * where a wide kernel ended, stream N was busy back to back and srla_residual_cost of the job waited for it (0.5 ms of gaps on
 * stream W per call).  Here the lags r[] and the upper part of the predictor a[] stand in LDS ([index][lane]: conflict-free,
 * addresses are immediates because everything stays unrolled), AREG entries of a[] in registers and L = 32 items share a
 * wavefront: about 120 registers and 23 KB of LDS -- what ONE retiring workgroup of a wide kernel leaves behind.  The same
 * operations in the same order on the same operands (lpc.c:417-438): identical bits. */
template <int P, int L, int AREG>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4, 8))) /* at most 128 registers */ void srla_lpc_errvars_lean(SrlaJobParams jp, const double *__restrict__ lags_ws, double *__restrict__ err_ws,
                                                              double *__restrict__ gamma_ws, const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    NARROW_KERNEL_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = blockIdx.x * L + lane;
    if (lane >= (uint32_t)L || idx >= jp.num_items) return;   /* no barriers below: lanes are independent */
    if (sel != nullptr && sel[idx] != sel_round) return;
    const size_t stride = jp.num_items;
    double *rl = reinterpret_cast<double *>(lds) + lane;      /* r[k] at rl[k * L] */
    double *al = rl + (size_t)(P + 1) * L;                    /* a[i], i >= AREG, at al[(i - AREG) * L] */
#pragma unroll
    for (int i = 0; i <= P; i++) rl[i * L] = lags_ws[(size_t)i * stride + idx];
    const double r0 = rl[0] * (1.0 + 1e-5);                   /* ridge, lpc.c:483 */
    double *err = err_ws + idx, *gam = gamma_ws + idx;
    err[0] = r0;
    if (fabs(r0) < (double)FLT_EPSILON) {
        /* lpc.c:395-405: every error variance is r0, every predictor zero */
        for (uint32_t o = 1; o <= (uint32_t)P; o++) { err[(size_t)o * stride] = r0; gam[(size_t)(o - 1) * stride] = 0.0; }
        return;
    }
    double areg[AREG];
    /* i is a constant wherever these are called (the loops below are fully unrolled), so the choice folds away */
    auto A = [&](int i) -> double { return (i < AREG) ? areg[i < AREG ? i : 0] : al[(i - AREG) * L]; };
    auto setA = [&](int i, double v) { if (i < AREG) areg[i < AREG ? i : 0] = v; else al[(i - AREG) * L] = v; };
    const double r1 = rl[L];
    const double a1 = -r1 / r0;
    setA(0, 1.0); setA(1, a1); setA(2, 0.0);
    double e = r0 + r1 * a1;
    err[stride] = e;
    gam[0] = a1;
#pragma unroll
    for (int k = 1; k < P; k++) {
        /* (the fences keep the scheduler from hoisting a whole step's LDS loads to its top: chunks of eight terms, whose loads
         * are in flight together, stay within the register budget) */
        double gamma = 0.0;
#pragma unroll
        for (int i = 0; i <= k; i++) {
            gamma += A(i) * rl[(k + 1 - i) * L];                               /* index order, lpc.c:420-423 */
            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        gamma /= -e;
        e = e * (1.0 - gamma * gamma);
#pragma unroll
        for (int i = 0; i <= (k + 1) / 2; i++) {
            const int j = k + 1 - i;
            const double ai = A(i), aj = A(j);
            setA(i, ai + gamma * aj);
            if (i != j) setA(j, aj + gamma * ai);
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        setA(k + 2, 0.0);
        __builtin_amdgcn_sched_barrier(0);
        err[(size_t)(k + 1) * stride] = e;
        gam[(size_t)k * stride] = gamma;
    }
}

template <int P>
__global__ __launch_bounds__(WAVE) void srla_lpc_taps(SrlaJobParams jp, const double *__restrict__ err_ws, const double *__restrict__ gamma_ws,
                                                      const uint8_t *__restrict__ huff_len, SrlaItemResult *__restrict__ results,
                                                      double *__restrict__ coef_ws /* SVR refinement follows: the predictor of the chosen order goes here (row of 64 per item), unquantised */,
                                                      const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    NARROW_KERNEL_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int L = WAVE;
    const uint32_t lane = threadIdx.x;
    uint8_t *s_huff = lds;                                   /* 512 bytes: plain, pair-summed code lengths */
    double *snap = (double *)(lds + 512);                    /* snap[i * L + lane], i < P: taps of the chosen order */
    int32_t *q = (int32_t *)(snap + (size_t)P * L);          /* q[i * L + lane]: quantised taps */
    for (uint32_t i = lane; i < 128; i += WAVE) ((uint32_t *)s_huff)[i] = ((const uint32_t *)huff_len)[i];
    __syncthreads();
    const uint32_t idx = blockIdx.x * WAVE + lane;
    if (idx >= jp.num_items) return;                         /* no barriers below: lanes are independent */
    if (sel != nullptr && sel[idx] != sel_round) return;
    const size_t stride = jp.num_items;
    SrlaItemResult *out = &results[idx];
    const uint32_t order = out->lpc_order;
    const bool silent = fabs(err_ws[idx]) < (double)FLT_EPSILON;
    const double *gam = gamma_ws + idx;
    /* every reflection coefficient the wavefront can need, fetched at once (one coalesced load per step, all in flight together:
     * fetched step by step inside the branch below they were 63 dependent round trips, 50 us of the launch) */
    uint32_t top = order;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o2 = (uint32_t)__shfl_xor((int)top, off, WAVE); top = (o2 > top) ? o2 : top; }
    double g[P];
#pragma unroll
    for (int k = 0; k < P; k++) g[k] = ((uint32_t)k < top) ? gam[(size_t)k * stride] : 0.0;
    double a[P + 2];
    a[0] = 1.0; a[1] = g[0]; a[2] = 0.0;
#pragma unroll
    for (int k = 1; k < P; k++) {
        if ((uint32_t)k < order) {                           /* (a wavefront goes as far as the highest order among its items) */
            const double gamma = g[k];
#pragma unroll
            for (int i = 0; i <= (k + 1) / 2; i++) {
                const int j = k + 1 - i;
                const double ai = a[i], aj = a[j];
                a[i] = ai + gamma * aj;
                if (i != j) a[j] = aj + gamma * ai;
            }
        }
        a[k + 2] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < P; i++) snap[(size_t)i * L + lane] = a[1 + i];
    if (coef_ws != nullptr) {
        double *row = coef_ws + (size_t)idx * 64u;             /* SVR_P doubles per item whatever the preset */
        for (uint32_t i = 0; i < order; i++) row[i] = silent ? 0.0 : snap[(size_t)i * L + lane];
        return;
    }
    quantize_and_price(order, silent,
                       [&](uint32_t i) -> double { return snap[(size_t)i * L + lane]; },
                       [&](uint32_t i, int32_t v) { q[(size_t)i * L + lane] = v; },
                       [&](uint32_t i) -> int32_t { return q[(size_t)i * L + lane]; }, s_huff, out);
}

template <int L>
__global__ __launch_bounds__(WAVE) void srla_lpc_quantize(
    SrlaJobParams jp, const double *__restrict__ lags_ws, const uint8_t *__restrict__ huff_len,
    SrlaItemResult *__restrict__ results, double *__restrict__ coef_ws /* SVR refinement follows: the taps of the chosen order, unquantised, rows of 256 */,
    const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = blockIdx.x * L + lane;
    if (lane >= L || idx >= jp.num_items) return;
    if (sel != nullptr && sel[idx] != sel_round) return;
    const uint32_t p = jp.max_order;
    double *a = (double *)lds;
    double *r = a + (size_t)(p + 2) * L;
    const size_t stride = jp.num_items;
    SrlaItemResult *out = &results[idx];
    const uint32_t order = out->lpc_order;
    uint32_t rshift = 0, use_sum = 0, coef_bits = 0;
    if (order > 0) {
        for (uint32_t i = 0; i <= order; i++) R_(i) = lags_ws[(size_t)i * stride + idx];
        const double r0 = R_(0) * (1.0 + 1e-5);
        const bool silent = fabs(r0) < (double)FLT_EPSILON;
        if (!silent) levinson_lane<L>(a, r, lane, r0, order, nullptr, 0);
        if (coef_ws != nullptr) {
            double *row = coef_ws + (size_t)idx * 256u;
            for (uint32_t i = 0; i < order; i++) row[i] = silent ? 0.0 : A_(1 + i);
            return;
        }
        /* 8-bit quantisation with error feedback from the last tap (lpc.c:1341-1405); q[] reuses r[] */
        int32_t *q = (int32_t *)r;
#define Q_(i) q[(size_t)(i) * (2 * L) + lane]
        double maxabs = 0.0;
        if (!silent) for (uint32_t i = 0; i < order; i++) { const double v = fabs(A_(1 + i)); if (maxabs < v) maxabs = v; }
        if (maxabs <= 0.0078125) {
            rshift = 8;
            for (uint32_t i = 0; i < order; i++) Q_(i) = 0;
        } else {
            int ndigit;
            (void)frexp(maxabs, &ndigit);
            rshift = (uint32_t)(7 - ndigit);
            if (rshift >= 16u) rshift = 15u;
            const double scale = __builtin_ldexp(1.0, (int)rshift);
            double qerr = 0.0;
            for (int i = (int)order - 1; i >= 0; i--) {
                qerr += A_(1 + i) * scale;
                int32_t qq = cvt_i32_as_x86(round_half_away(qerr));
                if (qq >= 128) qq = 127; else if (qq < -128) qq = -128;
                qerr -= (double)qq;
                Q_(order - 1 - i) = qq;          /* reversed: oldest sample first (srla_encoder.c:1104) */
            }
        }
        /* Huffman cost plain vs pair-summed (srla_encoder.c:1141-1174) */
        uint32_t plain = 0, summed = 0, overflow = 0;
        int32_t prevq = 0;
        for (uint32_t k = 0; k < order; k++) {
            const int32_t c = Q_(k);
            out->lpc_coef[k] = (int8_t)c;
            plain += huff_len[zigzag32(c)];
            if (k == 0) summed += huff_len[zigzag32(c)];
            else {
                const uint32_t z = zigzag32(c + prevq);
                if (z >= 256u) overflow = 1; else summed += huff_len[256 + z];
            }
            prevq = c;
        }
        use_sum = (overflow == 0 && (order == 1 || summed < plain)) ? 1u : 0u;
        coef_bits = use_sum ? summed : plain;
#undef Q_
    }
    out->lpc_rshift = rshift;
    out->use_sum = use_sum;
    out->pad[0] = coef_bits;
}
#undef A_
#undef R_

/* ================================================================================================
 * SVR refinement of the predictor (--svr-filter-learning-iteration > 0; lpc.c:1036-1136, reached from
 * srla_encoder.c:1084-1097).  Off by default and an order of magnitude more work than the rest of the analysis -- for the
 * reference as well.  One workgroup per item; everything whose value depends on the ORDER of a floating-point summation is
 * summed in the reference's order: the residual of a sample tap by tap (samples in parallel), mabse and the p entries of
 * r_vec sample by sample (one lane per running sum), the Cholesky factor and the two triangular solves row by row.
 * The covariance matrix (lpc.c:987-1020) is exact integer arithmetic whenever the products of the block cannot leave 53
 * bits (16-bit audio): then every partial sum of the reference is exact and the order is free; otherwise one thread per
 * matrix entry replays the reference's sum.
 * libm (H2): pow(x, -0.5) of the factorisation is the correctly rounded x^-1/2; log / pow of the objective
 * (lpc.c:1023-1033) are the device's -- the objective only steers comparisons, which are flagged (SRLA_ITEM_SVR_TIE) when
 * they are close enough for a last bit to matter.
 * ============================================================================================== */
#define SVR_NT 256
#define SVR_P  64           /* orders up to 64 (presets 0..4) */
#define SVR_PS 65           /* row stride of the matrix in LDS */

/* logscale: 1.0 (exact) in production; the tie tests falsify the device's log with it (SrlaJobParams) */
__device__ __forceinline__ double svr_rgr_mean_code_length(double mean_abs_error, bool *near_tie, double logscale)
{
    /* lpc.c:1023-1033 with BITS_PER_SAMPLE = 16 (:1042) */
    const double intmean = mean_abs_error * 65536.0;
    const double rho = 1.0 / (1.0 + intmean);
    const double l2 = (log(log(0.5127629514) / log(1.0 - rho)) * logscale) * 1.4426950408889634;
    const double m = (0.0 > l2) ? 0.0 : l2;
    const uint32_t k2 = (uint32_t)m;
    if (m > 0.5 && fabs(m - floor(m + 0.5)) < 1e-9) *near_tie = true;       /* the integer part hangs on log()'s last bits */
    const uint32_t k1 = k2 + 1;
    const double k1factor = pow(1.0 - rho, (double)(1u << k1));
    const double k2factor = pow(1.0 - rho, (double)(1u << k2));
    return (1.0 + k1) * (1.0 - k1factor) + (1.0 + k2 + (1.0 / (1.0 - k2factor))) * k1factor;
}

/* BIG = false: orders up to 64 and blocks up to n_cap <= 8192 samples, everything in LDS, one workgroup per item.
 * BIG = true: the other items (orders 128 / 255, blocks up to 32768 samples): block, residuals and the matrix live in a
 * region of global scratch per (persistent) workgroup, the vectors in LDS; the same code.  ws_stride: doubles per row of coef_ws. */
#define SVR_PMAX 256
template <bool BIG>
__device__ void svr_refine_item(const SrlaJobParams &jp, const int32_t *__restrict__ input, const SrlaItemDesc &it, SrlaItemResult *out,
                                double *row, const uint32_t iterations, int32_t *xi, double *rr, double *cov, const uint32_t PS,
                                double *low, double *r_vec, double *delta, double *coef, double *init_coef, double *best_coef,
                                double *s_scalar, long long *s_lag, uint32_t *s_flag, const SrlaSvrExtra ex, const uint32_t item_idx)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p = out->lpc_order;
    double *dump = (it.svr_dump != 0 && ex.chain_pool != nullptr) ? ex.chain_pool + (it.svr_dump - 1u) : nullptr;
    const double *under = (it.svr_under != 0 && ex.chain_pool != nullptr) ? ex.chain_pool + (it.svr_under - 1u) : nullptr;
    /* where the refinement does not touch the reference's buffer, its first n words stay what the LPC-pass call left */
    auto leave_untouched = [&]() { if (dump != nullptr && under != nullptr) for (uint32_t i = tid; i < it.n; i += SVR_NT) dump[i] = under[i]; };
    if (it.forced_svr != 0 && ex.forced_rows != nullptr) {
        /* the host has redone the refinement with its libm (host_ties.cpp) */
        __syncthreads();
        for (uint32_t i = tid; i < p; i += SVR_NT) row[i] = ex.forced_rows[(size_t)(it.forced_svr - 1u) * 256u + i];
        leave_untouched();
        return;
    }
    const InputView iv = input_view(jp, it.lshift);
    const uint32_t n = it.n;
    const int32_t *in = input + it.sample_off;
    const double norm = __builtin_ldexp(1.0, -(int)(jp.bits_per_sample - 1));
    __syncthreads();                                                     /* a persistent workgroup: the previous item is done with the buffers */
    /* ---- the signal the LPC analysis saw: pre-emphasis (srla_utility.c:342), long-term predictor (srla_lpc_predict.c:267) ---- */
    {
        const int32_t pc = out->preemph_coef;
        uint32_t am = 0;
        for (uint32_t i = tid; i < n; i += SVR_NT) {
            const int32_t cur = load_variant(in, iv, it.variant, i);
            const int32_t prev = (i == 0) ? cur : load_variant(in, iv, it.variant, i - 1);
            xi[i] = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)pc) >> 4));
        }
        if (tid < 4) s_flag[tid] = 0;
        __syncthreads();
        const uint32_t period = out->ltp_period;
        if (period > 0) {
            const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
            const int32_t c0 = out->ltp_coef[0], c1 = out->ltp_coef[1], c2 = out->ltp_coef[2];
            const uint32_t nblk = (n + SVR_NT - 1) / SVR_NT;
            for (uint32_t b = nblk; b-- > 0;) {                          /* in place, from the top down: every output reads lower indices only */
                const uint32_t s = b * SVR_NT + tid;
                int32_t v = 0;
                const bool act = s < n && s >= period + half_order + 1;
                if (act) {
                    const uint32_t base = s - period - half_order;
                    uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)xi[base];
                    if (taps == 3) acc += (uint32_t)c1 * (uint32_t)xi[base + 1] + (uint32_t)c2 * (uint32_t)xi[base + 2];
                    v = (int32_t)((uint32_t)xi[s] - (uint32_t)((int32_t)acc >> 5));
                }
                __syncthreads();
                if (act) xi[s] = v;
                __syncthreads();
            }
        }
        for (uint32_t i = tid; i < n; i += SVR_NT) { const int32_t v = xi[i]; const uint32_t a = (v < 0) ? (uint32_t)(-(int64_t)v) : (uint32_t)v; am = (a > am) ? a : am; }
        am = wave_max_u32(am);
        if (lane == 0) atomicMax(&s_flag[3], am);
        for (uint32_t i = tid; i < p; i += SVR_NT) { coef[i] = row[i]; init_coef[i] = row[i]; best_coef[i] = row[i]; }
        __syncthreads();
    }
    const uint32_t absmax = s_flag[3];
    const uint32_t m_len = n - p;                                        /* terms of every covariance sum */
    /* ---- covariance matrix, lpc.c:987-1020 ---- */
    if ((double)absmax * (double)absmax * (double)m_len < 4503599627370496.0 /* 2^52 */) {
        /* exact: row 0 by parallel integer dot products, the other entries by the exact recurrence along the diagonals
         * cov[i+1][j+1] = cov[i][j] - x[i] x[j] + x[m+i] x[m+j] */
        for (uint32_t d = wave; d < p; d += SVR_NT / WAVE) {
            long long acc = 0;
            for (uint32_t s0 = lane; s0 < m_len; s0 += WAVE) acc += (long long)xi[s0] * (long long)xi[s0 + d];
            acc = wave_sum_i64(acc);
            if (lane == 0) s_lag[d] = acc;
        }
        __syncthreads();
        const double scale = norm * norm;
        if (tid < p) {
            const uint32_t d = tid;
            long long v = s_lag[d];
            for (uint32_t i = 0; i + d < p; i++) {
                cov[i * PS + i + d] = (double)v * scale;
                v += (long long)xi[m_len + i] * (long long)xi[m_len + i + d] - (long long)xi[i] * (long long)xi[i + d];
            }
        }
    } else {
        /* the reference's own running sums, one thread per entry */
        const uint32_t npairs = p * (p + 1) / 2;
        for (uint32_t t = tid; t < npairs; t += SVR_NT) {
            uint32_t i = 0, r = t;
            while (r >= p - i) { r -= p - i; i++; }
            const uint32_t j = i + r;
            double acc = 0.0;
            for (uint32_t s0 = 0; s0 < m_len; s0++) acc += ((double)xi[s0 + i] * norm) * ((double)xi[s0 + j] * norm);
            cov[i * PS + j] = acc;
        }
    }
    __syncthreads();
    for (uint32_t t = tid; t < p * p; t += SVR_NT) { const uint32_t i = t / p, j = t % p; if (j > i) cov[j * PS + i] = cov[i * PS + j]; }
    __syncthreads();
    if (tid < p) cov[tid * PS + tid] *= (1.0 + 1e-5);                /* ridge, lpc.c:1067-1069 */
    __syncthreads();
    /* ---- Cholesky factorisation, lpc.c:573-600 ---- */
    for (uint32_t i = 0; i < p; i++) {
        if (tid == 0) {
            double sum = cov[i * PS + i];
            for (int k = (int)i - 1; k >= 0; k--) sum -= cov[i * PS + k] * cov[i * PS + k];
            if (sum <= 0.0) s_flag[0] = 1;
            else low[i] = inv_sqrt_cr(sum);
        }
        __syncthreads();
        if (s_flag[0]) break;
        const uint32_t j = i + 1 + tid;
        if (j < p) {
            double sum = cov[i * PS + j];
            for (int k = (int)i - 1; k >= 0; k--) sum -= cov[i * PS + k] * cov[j * PS + k];
            cov[j * PS + i] = sum * low[i];
        }
        __syncthreads();
    }
    if (s_flag[0]) {                                                     /* singular: all-zero input (lpc.c:1071-1077) */
        for (uint32_t i = tid; i < p; i += SVR_NT) row[i] = 0.0;
        leave_untouched();
        return;
    }
    /* ---- the learning loop, lpc.c:1083-1127 ---- */
    const double margins[6] = { 0.0, 1.0 / 4096, 1.0 / 1024, 1.0 / 256, 1.0 / 64, 1.0 / 16 };   /* srla_internal.c:27 */
    double min_obj = (double)FLT_MAX;                                    /* uniform: every thread keeps its own copy */
    for (int mi = 0; mi < 6; mi++) {
        const double margin = margins[mi];
        double prev_obj = (double)FLT_MAX;
        __syncthreads();
        for (uint32_t i = tid; i < p; i += SVR_NT) coef[i] = init_coef[i];
        __syncthreads();
        for (uint32_t itr = 0; itr < iterations; itr++) {
            /* residual of every sample: taps in index order (lpc.c:1098-1100), samples in parallel */
            for (uint32_t s0 = p + tid; s0 < n; s0 += SVR_NT) {
                double res = (double)xi[s0] * norm;
                for (uint32_t i = 0; i < p; i++) res += coef[i] * ((double)xi[s0 - i - 1] * norm);
                rr[s0] = res;
            }
            __syncthreads();
            /* the running sums over the samples, in sample order: r_vec[i] on lane i of wave 0, mabse on wave 1 */
            if (tid < p) {
                double acc = 0.0;
                for (uint32_t s0 = p; s0 < n; s0++) {
                    const double r = rr[s0];
                    const double a = (r > 0) ? r : -r;
                    const double t = (double)((r > 0) - (r < 0)) * (((a - margin) > 0.0) ? (a - margin) : 0.0);   /* LPC_SOFT_THRESHOLD, lpc.c:34 */
                    acc += t * ((double)xi[s0 - tid - 1] * norm);
                }
                r_vec[tid] = acc;
            } else if (tid == SVR_NT - 1) {                               /* p <= 255: this thread has no r_vec entry */
                double acc = 0.0;
                for (uint32_t s0 = p; s0 < n; s0++) { const double r = rr[s0]; acc += (r > 0) ? r : -r; }
                s_scalar[0] = acc;
            }
            __syncthreads();
            if (tid == 0) {
                bool tie = false;
                const double obj = svr_rgr_mean_code_length(s_scalar[0] / (double)n, &tie, jp.tie_logscale);
                /* cov delta = r_vec by the factor, lpc.c:605-631 */
                for (uint32_t i = 0; i < p; i++) {
                    double sum = r_vec[i];
                    for (int k = (int)i - 1; k >= 0; k--) sum -= cov[i * PS + k] * delta[k];
                    delta[i] = sum * low[i];
                }
                for (int k = (int)p - 1; k >= 0; k--) {
                    double sum = delta[k];
                    for (uint32_t j = (uint32_t)k + 1; j < p; j++) sum -= cov[j * PS + k] * delta[j];
                    delta[k] = sum * low[k];
                }
                s_scalar[1] = obj;
                if (tie) s_flag[2] = 1;
            }
            __syncthreads();
            const double obj = s_scalar[1];
            /* comparisons of objective values that differ by less than the device's log / pow can be trusted for */
            if (tid == 0) {
                const double tol = jp.tie_rel;
                if ((obj != min_obj && fabs(obj - min_obj) <= tol * fabs(obj)) || (obj != prev_obj && fabs(obj - prev_obj) <= tol * fabs(obj))
                    || fabs(fabs(prev_obj - obj) - 1e-8) <= 1e-8 * tol) s_flag[2] = 1;
            }
            if (obj < min_obj) {
                for (uint32_t i = tid; i < p; i += SVR_NT) best_coef[i] = coef[i];
                min_obj = obj;
            }
            if ((prev_obj < obj) || (fabs(prev_obj - obj) < 1e-8)) break;
            for (uint32_t i = tid; i < p; i += SVR_NT) coef[i] += delta[i];
            prev_obj = obj;
            __syncthreads();
        }
    }
    __syncthreads();
    if (dump != nullptr) {
        /* What the reference's `residual` = the calculator's persistent buffer holds now (lpc.c:1047,1095-1106): the block itself
         * below the order, from there on the soft-thresholded residual of the LAST pass that ran (rr[] still holds that pass's
         * residuals; its margin is the last of the list) -- what an odd-length or a short LTP block analysed later inherits. */
        const double margin = margins[5];
        for (uint32_t s0 = tid; s0 < n; s0 += SVR_NT) {
            double v;
            if (s0 < p) v = (double)xi[s0] * norm;
            else {
                const double r = rr[s0];
                const double a = (r > 0) ? r : -r;
                v = (double)((r > 0) - (r < 0)) * (((a - margin) > 0.0) ? (a - margin) : 0.0);
            }
            dump[s0] = v;
        }
    }
    for (uint32_t i = tid; i < p; i += SVR_NT) row[i] = best_coef[i];
    if (tid == 0 && s_flag[2]) {
        out->flags |= SRLA_ITEM_SVR_TIE;
        if (ex.ties != nullptr) (void)tie_append(ex.ties, item_idx, 2u);
    }
}


__global__ __launch_bounds__(SVR_NT) void srla_svr_refine(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    SrlaItemResult *__restrict__ results, double *__restrict__ coef_ws, uint32_t ws_stride, uint32_t iterations, uint32_t n_cap, SrlaSvrExtra ex)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int32_t *xi = (int32_t *)lds;                                        /* the pre-emphasised (+ LTP) block */
    double *rr = (double *)(lds + (((size_t)n_cap * 4 + 15) & ~(size_t)15));   /* residual of every sample under the current taps */
    double *cov = rr + n_cap;                                            /* [SVR_P][SVR_PS]: upper = matrix, lower = Cholesky factor */
    double *low = cov + SVR_P * SVR_PS, *r_vec = low + SVR_P, *delta = r_vec + SVR_P, *coef = delta + SVR_P;
    double *init_coef = coef + SVR_P, *best_coef = init_coef + SVR_P;
    __shared__ double s_scalar[2];
    __shared__ long long s_lag[SVR_P];
    __shared__ uint32_t s_flag[4];                                       /* 0: singular, 1: break, 2: near-tie, 3: |x| max */
    const uint32_t item_idx = xcd_position(blockIdx.x, jp.num_items);
    if (item_idx >= jp.num_items) return;
    if (ex.select != nullptr && ex.select[item_idx] != ex.round) return;
    SrlaItemResult *out = &results[item_idx];
    const uint32_t p = out->lpc_order;
    const SrlaItemDesc it = items[item_idx];
    if (p == 0 && it.svr_dump != 0 && it.svr_under != 0 && ex.chain_pool != nullptr)   /* order 0: no refinement (srla_encoder.c:1084) */
        for (uint32_t i = threadIdx.x; i < it.n; i += SVR_NT) ex.chain_pool[(size_t)(it.svr_dump - 1u) + i] = ex.chain_pool[(size_t)(it.svr_under - 1u) + i];
    if (p == 0 || p > SVR_P || it.n > n_cap) return;                     /* the others: srla_svr_refine_big */
    svr_refine_item<false>(jp, input, it, out, coef_ws + (size_t)item_idx * ws_stride, iterations, xi, rr, cov, SVR_PS,
                           low, r_vec, delta, coef, init_coef, best_coef, s_scalar, s_lag, s_flag, ex, item_idx);
}

/* per workgroup in `scratch`: n_max int32, n_max doubles, SVR_PMAX x (SVR_PMAX + 1) doubles */
__device__ __forceinline__ size_t srla_svr_big_scratch_bytes_dev(uint32_t n_max)
{
    return (((size_t)n_max * 4 + 15) & ~(size_t)15) + (size_t)n_max * 8 + (size_t)SVR_PMAX * (SVR_PMAX + 1) * 8;
}
extern "C" size_t srla_svr_big_scratch_bytes(uint32_t n_max)
{
    return (((size_t)n_max * 4 + 15) & ~(size_t)15) + (size_t)n_max * 8 + (size_t)SVR_PMAX * (SVR_PMAX + 1) * 8;
}

__global__ __launch_bounds__(SVR_NT) void srla_svr_refine_big(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    SrlaItemResult *__restrict__ results, double *__restrict__ coef_ws, uint32_t ws_stride, uint32_t iterations, uint32_t n_cap,
    unsigned char *__restrict__ scratch, uint32_t n_max, SrlaSvrExtra ex)
{
    __shared__ double vec[6][SVR_PMAX];
    __shared__ double s_scalar[2];
    __shared__ long long s_lag[SVR_PMAX];
    __shared__ uint32_t s_flag[4];
    unsigned char *mine = scratch + (size_t)blockIdx.x * srla_svr_big_scratch_bytes_dev(n_max);
    int32_t *xi = (int32_t *)mine;
    double *rr = (double *)(mine + (((size_t)n_max * 4 + 15) & ~(size_t)15));
    double *cov = rr + n_max;
    for (uint32_t item_idx = blockIdx.x; item_idx < jp.num_items; item_idx += gridDim.x) {
        if (ex.select != nullptr && ex.select[item_idx] != ex.round) continue;
        SrlaItemResult *out = &results[item_idx];
        const uint32_t p = out->lpc_order;
        const SrlaItemDesc it = items[item_idx];
        if (p == 0 || (p <= SVR_P && it.n <= n_cap)) continue;           /* done by srla_svr_refine */
        svr_refine_item<true>(jp, input, it, out, coef_ws + (size_t)item_idx * ws_stride, iterations, xi, rr, cov, SVR_PMAX + 1,
                              vec[0], vec[1], vec[2], vec[3], vec[4], vec[5], s_scalar, s_lag, s_flag, ex, item_idx);
    }
}

/* the quantiser and tap cost (lpc.c:1341-1405, srla_encoder.c:1141-1174) from the refined taps: one lane per item */
__global__ __launch_bounds__(WAVE) void srla_lpc_quantize_ws(SrlaJobParams jp, const double *__restrict__ coef_ws, uint32_t ws_stride,
                                                             const uint8_t *__restrict__ huff_len, SrlaItemResult *__restrict__ results,
                                                             const uint32_t *__restrict__ sel, uint32_t sel_round, uint32_t *__restrict__ ties)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int L = WAVE;
    const uint32_t lane = threadIdx.x;
    uint8_t *s_huff = lds;
    int32_t *q = (int32_t *)(lds + 512);
    for (uint32_t i = lane; i < 128; i += WAVE) ((uint32_t *)s_huff)[i] = ((const uint32_t *)huff_len)[i];
    __syncthreads();
    const uint32_t idx = blockIdx.x * WAVE + lane;
    if (idx >= jp.num_items) return;
    if (sel != nullptr && sel[idx] != sel_round) return;
    SrlaItemResult *out = &results[idx];
    const uint32_t order = out->lpc_order;
    const double *row = coef_ws + (size_t)idx * ws_stride;
    /* The refined taps carry the last bits of the refinement's pow(x, -0.5) (lpc.c:591 via :1071; the device's is the correctly
     * rounded value, glibc's within an ulp of it): where the quantiser's outcome hangs on such bits the item is flagged like an
     * objective near-tie and the host redoes the refinement with its own libm (host_ties.cpp: arbitrate_svr).  The band is the
     * near-tie threshold of the comparisons (1e-9 in production): eight orders of magnitude above what an ulp of a tap moves. */
    bool near = false;
    quantize_and_price(order, false,
                       [&](uint32_t i) -> double { return row[i]; },
                       [&](uint32_t i, int32_t v) { q[(size_t)i * L + lane] = v; },
                       [&](uint32_t i) -> int32_t { return q[(size_t)i * L + lane]; }, s_huff, out, jp.tie_rel, &near);
    if (near && order > 0 && !(out->flags & SRLA_ITEM_SVR_TIE)) {
        out->flags |= SRLA_ITEM_SVR_TIE;
        if (ties != nullptr) (void)tie_append(ties, idx, 2u);
    }
}

/* --------------------------------------------------------------------------- launchers ---- */
extern "C" int srla_launch_pitch_solve(hipStream_t stream, const SrlaJobParams *jp, const SrlaItemDesc *items, const double *lags_ws,
                                       SrlaItemResult *results, hipEvent_t ev_start, hipEvent_t ev_stop,
                                       const uint32_t *select, uint32_t round, uint32_t *ties, double *tie_data)
{
    if (jp->num_items == 0) return 0;
    hipExtLaunchKernelGGL(srla_pitch_solve, dim3((jp->num_items + WAVE - 1) / WAVE), dim3(WAVE), 0, stream, ev_start, ev_stop, 0,
                          *jp, items, lags_ws, results, select, round, ties, tie_data);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

/* the recursion of the three-launch solve chain: order 64 in the lean form (srla_lpc_errvars_lean), smaller orders in registers */
#define ERRVARS_LEAN_L 32
#define ERRVARS_LEAN_AREG 36
template <int PP>
static void launch_errvars(hipStream_t stream, hipEvent_t ev_start, const SrlaJobParams *jp, const double *lags_ws, double *err_ws,
                           double *gamma_ws, const SrlaSvrExtra &ex)
{
    if constexpr (PP == 64) {
        /* a small job (a short stream, a piece of one) does not fill the chip: nothing to be starved by, and the register form is
         * twice as fast on its own (31 against 65 us) */
        if (jp->num_items >= 6144u || jp->crowded) {
            const uint32_t lds = ((PP + 1) + (PP + 2 - ERRVARS_LEAN_AREG)) * 8 * ERRVARS_LEAN_L;
            hipExtLaunchKernelGGL((srla_lpc_errvars_lean<PP, ERRVARS_LEAN_L, ERRVARS_LEAN_AREG>), dim3((jp->num_items + ERRVARS_LEAN_L - 1) / ERRVARS_LEAN_L),
                                  dim3(WAVE), lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, gamma_ws, ex.select, ex.round);
            return;
        }
    }
    hipExtLaunchKernelGGL(srla_lpc_errvars<PP>, dim3((jp->num_items + 63) / 64), dim3(WAVE), 0, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws,
                          gamma_ws, ex.select, ex.round);
}

extern "C" int srla_launch_lpc_solve(hipStream_t stream, const SrlaJobParams *jp, const SrlaItemDesc *items,
                                     const SrlaGeom *geoms, const double *lags_ws, double *err_ws, const uint8_t *huff_len,
                                     SrlaItemResult *results, double *dbg, uint32_t *ties, hipEvent_t ev_start, hipEvent_t ev_stop,
                                     const int32_t *input, double *coef_ws, uint32_t svr_iterations, uint32_t svr_n_cap,
                                     void *svr_scratch, uint32_t svr_groups, double *gamma_ws, const SrlaSvrExtra *svr_extra)
{
    if (jp->num_items == 0) return 0;
    const uint32_t p = jp->max_order;
    SrlaSvrExtra ex = { nullptr, nullptr, nullptr, nullptr, 0u };
    if (svr_extra) ex = *svr_extra;
    if (gamma_ws == nullptr) return -1;         /* orders 8 .. 64: errvars + order_select + taps */
    if (svr_iterations > 0) {
        /* solve (taps left unquantised) -> SVR refinement -> quantiser */
        const dim3 g64s((jp->num_items + 63) / 64), blks(WAVE);
        const uint32_t ws_stride = (p <= 64) ? 64u : 256u;
#define SVR_PATH(PP)                                                                                                     \
    do {                                                                                                                 \
        const uint32_t lds = 512 + PP * 8 * 64 + PP * 4 * 64;                                                            \
        SET_LDS_ATTR(srla_lpc_taps<PP>);                                                                                 \
        launch_errvars<PP>(stream, ev_start, jp, lags_ws, err_ws, gamma_ws, ex);                                         \
        hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blks, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round); \
        hipLaunchKernelGGL(srla_lpc_taps<PP>, g64s, blks, lds, stream, *jp, err_ws, gamma_ws, huff_len, results, coef_ws, ex.select, ex.round); \
    } while (0)
        if (p == 8) SVR_PATH(8); else if (p == 16) SVR_PATH(16); else if (p == 32) SVR_PATH(32); else if (p == 64) SVR_PATH(64);
        else if (p <= 128) {
            const uint32_t lds = (2 * p + 3) * 8 * 64;
            SET_LDS_ATTR(srla_lpc_recursion<64>);
            SET_LDS_ATTR(srla_lpc_quantize<64>);
            hipExtLaunchKernelGGL(srla_lpc_recursion<64>, g64s, blks, lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, ex.select, ex.round);
            hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blks, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round);
            hipLaunchKernelGGL(srla_lpc_quantize<64>, g64s, blks, lds, stream, *jp, lags_ws, huff_len, results, coef_ws, ex.select, ex.round);
        } else {
            const uint32_t lds = (2 * p + 3) * 8 * 32;
            SET_LDS_ATTR(srla_lpc_recursion<32>);
            SET_LDS_ATTR(srla_lpc_quantize<32>);
            hipExtLaunchKernelGGL(srla_lpc_recursion<32>, dim3((jp->num_items + 31) / 32), blks, lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, ex.select, ex.round);
            hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blks, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round);
            hipLaunchKernelGGL(srla_lpc_quantize<32>, dim3((jp->num_items + 31) / 32), blks, lds, stream, *jp, lags_ws, huff_len, results, coef_ws, ex.select, ex.round);
        }
#undef SVR_PATH
        {   /* items of order <= 64 in blocks that fit LDS, whatever the preset's maximum */
            const uint32_t lds_svr = ((svr_n_cap * 4u + 15u) & ~15u) + svr_n_cap * 8u + (SVR_P * SVR_PS + 6 * SVR_P) * 8u;
            {
                static bool done_ = false;
                if (!done_) { (void)hipFuncSetAttribute((const void *)srla_svr_refine, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048); done_ = true; }
                (void)hipGetLastError();
            }
            hipLaunchKernelGGL(srla_svr_refine, dim3(8u * ((jp->num_items + 7u) >> 3)), dim3(SVR_NT), lds_svr, stream, *jp, input, items, results, coef_ws,
                               ws_stride, svr_iterations, svr_n_cap, ex);
        }
        if (p > 64 || jp->max_block > svr_n_cap) {
            /* orders 128 / 255, blocks above 8192 samples: persistent workgroups on global scratch */
            if (svr_scratch == nullptr || svr_groups == 0) return -1;
            const uint32_t groups = jp->num_items < svr_groups ? jp->num_items : svr_groups;
            hipLaunchKernelGGL(srla_svr_refine_big, dim3(groups), dim3(SVR_NT), 0, stream, *jp, input, items, results, coef_ws,
                               ws_stride, svr_iterations, svr_n_cap, (unsigned char *)svr_scratch, jp->max_block, ex);
        }
        SET_LDS_ATTR(srla_lpc_quantize_ws);
        hipExtLaunchKernelGGL(srla_lpc_quantize_ws, g64s, blks, 512 + (p < 64 ? 64 : p) * 4 * 64, stream, nullptr, ev_stop, 0, *jp, coef_ws, ws_stride, huff_len, results, ex.select, ex.round, ex.ties);
        return (hipGetLastError() == hipSuccess) ? 0 : -2;
    }
    const dim3 g64((jp->num_items + 63) / 64), blk(WAVE);
    /* orders 8 .. 64: recursion (one lane per item), order choice (one wavefront per item), taps (one lane per item) */
#define REGS_PATH(PP)                                                                                                    \
    do {                                                                                                                 \
        const uint32_t lds = 512 + PP * 8 * 64 + PP * 4 * 64;                                                            \
        SET_LDS_ATTR(srla_lpc_taps<PP>);                                                                                 \
        launch_errvars<PP>(stream, ev_start, jp, lags_ws, err_ws, gamma_ws, ex);                                         \
        hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blk, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round); \
        hipExtLaunchKernelGGL(srla_lpc_taps<PP>, g64, blk, lds, stream, nullptr, ev_stop, 0, *jp, err_ws, gamma_ws, huff_len, results, (double *)nullptr, ex.select, ex.round); \
    } while (0)
    if (p == 8) REGS_PATH(8);
    else if (p == 16) REGS_PATH(16);
    else if (p == 32) REGS_PATH(32);
    else if (p == 64) REGS_PATH(64);
    else if (p <= 128) {
        const uint32_t lds = (2 * p + 3) * 8 * 64;
        SET_LDS_ATTR(srla_lpc_recursion<64>);
        SET_LDS_ATTR(srla_lpc_quantize<64>);
        hipExtLaunchKernelGGL(srla_lpc_recursion<64>, g64, blk, lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, ex.select, ex.round);
        hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blk, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round);
        hipExtLaunchKernelGGL(srla_lpc_quantize<64>, g64, blk, lds, stream, nullptr, ev_stop, 0, *jp, lags_ws, huff_len, results, (double *)nullptr, ex.select, ex.round);
    } else {
        const uint32_t lds = (2 * p + 3) * 8 * 32;
        SET_LDS_ATTR(srla_lpc_recursion<32>);
        SET_LDS_ATTR(srla_lpc_quantize<32>);
        hipExtLaunchKernelGGL(srla_lpc_recursion<32>, dim3((jp->num_items + 31) / 32), blk, lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, ex.select, ex.round);
        hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blk, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round);
        hipExtLaunchKernelGGL(srla_lpc_quantize<32>, dim3((jp->num_items + 31) / 32), blk, lds, stream, nullptr, ev_stop, 0, *jp, lags_ws, huff_len, results, (double *)nullptr, ex.select, ex.round);
    }
#undef REGS_PATH
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

