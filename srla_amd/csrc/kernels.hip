/*
 * kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the SRLA encode hot path.
 *
 *   srla_analyze_items   one workgroup per item (candidate block x channel variant): samples are
 *                        staged once in LDS and everything up to the item's code length stays
 *                        in LDS / registers: pre-emphasis, optional long-term predictor,
 *                        Welch window + real-FFT autocorrelation (fp64, bit-exact operation
 *                        order of libs/fft), Levinson-Durbin, order choice, 8-bit tap
 *                        quantisation, integer FIR residual, partitioned (recursive) Rice
 *                        code-length search.           (srla_encoder.c:966-1205 and callees)
 *   srla_price_windows   stereo decision + block sizes + the shortest path over block
 *                        divisions, one thread per window.   (srla_encoder.c:1208-1334,
 *                        :1477-1546, :249-424)
 *   srla_gather_blocks   copies the residuals / parameters of the chosen blocks into the
 *                        compact buffers the host bit-packer reads.
 *   srla_or_reduce       whole-stream OR for the offset left shift (srla_utility.c:177-203)
 *
 * No MFMA: the path is integer/fp64 reductions and butterflies, not a dense contraction.
 * All fp64 arithmetic must round exactly like the C90 reference: this file is compiled with
 * -ffp-contract=off and additionally pins it with the pragma below.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "device_layout.h"
#include "kernels.h"

#pragma clang fp contract(off)

#define NT 256
#define WAVE 64
#define NWAVES (NT / WAVE)

typedef double2 cplx;

/* fixed-size LDS scratch of kernel A */
struct Small {
    double   lags[272];        /* autocorrelation lags (LTP needs 263 + 2 stale reads) */
    double   dscratch[16];
    long long lscratch[16];
    uint32_t uscratch[32];
    int32_t  coef[256];        /* quantised taps, stream (reversed) order */
    uint32_t level_bits[16];
    uint8_t  ktab[2048];       /* heap layout: level p at [2^p - 1, 2^(p+1) - 1) */
    int32_t  preemph_coef;
    uint32_t order;
    uint32_t rshift;
    uint32_t period;
    int32_t  ltp_coef[4];
    uint32_t flags;
    uint32_t max_u;
    uint32_t code_type;
    uint32_t porder;
    uint32_t res_bits;
    uint32_t seq_path;
    uint32_t pad[2];
};

extern "C" uint32_t srla_kernel_small_bytes(void) { return (uint32_t)((sizeof(Small) + 15) & ~15u); }

/* ---------------------------------------------------------------- small device helpers --- */
__device__ __forceinline__ uint32_t zigzag32(int32_t s) { return ((uint32_t)s << 1) ^ (uint32_t)(-(int32_t)(s < 0)); }

__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx c_mul(cplx a, cplx b)
{
    /* fft.c:57-63: two roundings per product term, no fused multiply-add */
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ double round_half_away(double d)
{
    /* srla_utility.c:22-25 */
    return (d >= 0.0) ? floor(d + 0.5) : -floor(-d + 0.5);
}

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_down(v, off, WAVE); v = (o > v) ? o : v; }
    return v;
}

/* variant sample i of the job input (srla_encoder.c:1229-1253, srla_utility.c:91-103) */
__device__ __forceinline__ int32_t load_variant(const int32_t *__restrict__ in, const SrlaJobParams &jp,
                                                uint32_t variant, uint32_t idx)
{
    const uint32_t sh = jp.offset_lshift;
    if (variant < jp.num_channels) return in[(size_t)variant * jp.channel_stride + idx] >> sh;
    const int32_t l = in[idx] >> sh;
    const int32_t r = in[(size_t)jp.channel_stride + idx] >> sh;
    const int32_t s = (int32_t)((uint32_t)r - (uint32_t)l);
    if (variant == jp.num_channels + 1) return s;
    return (int32_t)((uint32_t)l + (uint32_t)(s >> 1));
}

/* ------------------------------------------------------------------------------ FFT ------ */
/* complex FFT of m points held interleaved in LDS, radix-4 decimation in frequency with the
 * reference's (Stockham) butterfly arithmetic; every butterfly output is staged in registers
 * so one LDS buffer suffices (two barriers per stage).  fft.c:71-136.
 * tw: per-stage w1 tables, stage with sub-size n holds n/4 entries. */
template <int R>
__device__ void fft_complex_lds(cplx *x, uint32_t m, int flag, const cplx *__restrict__ tw)
{
    const uint32_t tid = threadIdx.x;
    uint32_t n = m, s = 1, log2s = 0;
    const double jim = (double)(-flag);
    while (n > 2) {
        const uint32_t n1 = n >> 2, n2 = n >> 1, n3 = n1 + n2;
        const uint32_t nb = m >> 2;
        cplx a[R], b[R], c[R], d[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t bf = tid + (uint32_t)r * NT;
            if (bf < nb) {
                const uint32_t p = bf >> log2s, q = bf & (s - 1);
                a[r] = x[q + s * p];
                b[r] = x[q + s * (p + n1)];
                c[r] = x[q + s * (p + n2)];
                d[r] = x[q + s * (p + n3)];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t bf = tid + (uint32_t)r * NT;
            if (bf < nb) {
                const uint32_t p = bf >> log2s, q = bf & (s - 1);
                const cplx w1 = tw[p];
                const cplx w2 = c_mul(w1, w1);
                const cplx w3 = c_mul(w1, w2);
                const cplx apc = c_add(a[r], c[r]), amc = c_sub(a[r], c[r]), bpd = c_add(b[r], d[r]);
                const cplx bmd = c_sub(b[r], d[r]);
                /* (0, -flag) * (b - d), written out as the reference's complex product */
                const cplx jbmd = make_double2(0.0 * bmd.x - jim * bmd.y, 0.0 * bmd.y + jim * bmd.x);
                const uint32_t o = q + s * (p << 2);
                x[o] = c_add(apc, bpd);
                x[o + s] = c_mul(w1, c_sub(amc, jbmd));
                x[o + 2 * s] = c_mul(w2, c_sub(apc, bpd));
                x[o + 3 * s] = c_mul(w3, c_add(amc, jbmd));
            }
        }
        __syncthreads();
        tw += n1;
        n >>= 2;
        s <<= 2;
        log2s += 2;
    }
    if (n == 2) {
        cplx a[2 * R], b[2 * R];
#pragma unroll
        for (int r = 0; r < 2 * R; r++) {
            const uint32_t q = tid + (uint32_t)r * NT;
            if (q < s) { a[r] = x[q]; b[r] = x[q + s]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2 * R; r++) {
            const uint32_t q = tid + (uint32_t)r * NT;
            if (q < s) { x[q] = c_add(a[r], b[r]); x[q + s] = c_sub(a[r], b[r]); }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ uint32_t complex_table_len(uint32_t m)
{
    uint32_t t = 0;
    for (uint32_t n = m; n > 2; n >>= 2) t += n >> 2;
    return t;
}

/* spectral symmetry pass of the real FFT (fft.c:164-183); rtw[i-1] = (wr, wi) for pair i */
__device__ void real_fft_pairs(double *x, uint32_t nfft, int flag, const cplx *__restrict__ rtw)
{
    const double c2 = flag * 0.5;
    const uint32_t quarter = nfft >> 2;
    for (uint32_t i = 1 + threadIdx.x; i <= quarter; i += NT) {
        const uint32_t i1 = i << 1, i2 = i1 + 1, i3 = nfft - i1, i4 = i3 + 1;
        const double x1 = x[i1], x2 = x[i2], x3 = x[i3], x4 = x[i4];
        const double wr = rtw[i - 1].x, wi = rtw[i - 1].y;
        const double h1r = 0.5 * (x1 + x3);
        const double h1i = 0.5 * (x2 - x4);
        const double h2r = -c2 * (x2 + x4);
        const double h2i = c2 * (x1 - x3);
        if (i1 != i3) {
            x[i1] = h1r + (wr * h2r) - (wi * h2i);
            x[i2] = h1i + (wr * h2i) + (wi * h2r);
        }
        /* for the self-paired middle element the reference's second pair of stores wins */
        x[i3] = h1r - (wr * h2r) + (wi * h2i);
        x[i4] = -h1i + (wr * h2i) + (wi * h2r);
    }
    __syncthreads();
}

/* Welch window + circular autocorrelation through the FFT (lpc.c:236-266, 330-376).
 * y: pre-emphasised int32 signal in LDS; buf: nfft doubles in LDS; lags_out[0..num_lags). */
template <int R>
__device__ void windowed_autocorr(const int32_t *y, double *buf, const SrlaGeom &g, double norm_bps,
                                  const cplx *__restrict__ twbase, double *lags_out, uint32_t num_lags)
{
    const uint32_t n = g.n, nfft = g.nfft, m = nfft >> 1, half = n >> 1;
    const uint32_t ct = complex_table_len(m), quarter = nfft >> 2;
    const cplx *tw_fwd = twbase;
    const cplx *tw_inv = twbase + ct;
    const cplx *rtw_fwd = twbase + 2 * ct;
    const cplx *rtw_inv = rtw_fwd + quarter;

    for (uint32_t e = threadIdx.x; e < nfft; e += NT) {
        double v = 0.0;
        if (e < n) {
            const double in = (double)y[e] * norm_bps;
            uint32_t smpl;
            bool touched = true;
            if (e < half) smpl = e;
            else if (e >= n - half) smpl = n - 1 - e;
            else { smpl = 0; touched = false; }   /* middle sample of an odd block */
            if (touched) {
                const double w = g.welch_divisor * (double)smpl * (double)(n - 1 - smpl);
                v = in * w;
            }
        }
        buf[e] = v;
    }
    __syncthreads();

    /* forward: complex FFT of nfft/2 points, then the symmetry pass, then DC / Nyquist */
    fft_complex_lds<R>((cplx *)buf, m, -1, tw_fwd);
    real_fft_pairs(buf, nfft, -1, rtw_fwd);
    if (threadIdx.x == 0) {
        const double h1r = buf[0], im = buf[1];
        buf[0] = h1r + im;
        buf[1] = h1r - im;
    }
    __syncthreads();
    /* power spectrum, lpc.c:357-365 */
    for (uint32_t k = threadIdx.x; k < m; k += NT) {
        const double re = buf[2 * k], im = buf[2 * k + 1];
        if (k == 0) { buf[0] = re * re; buf[1] = im * im; }
        else { buf[2 * k] = re * re + im * im; buf[2 * k + 1] = 0.0; }
    }
    __syncthreads();
    /* inverse */
    real_fft_pairs(buf, nfft, 1, rtw_inv);
    if (threadIdx.x == 0) {
        const double h1r = buf[0], im = buf[1];
        buf[0] = 0.5 * (h1r + im);
        buf[1] = 0.5 * (h1r - im);
    }
    __syncthreads();
    fft_complex_lds<R>((cplx *)buf, m, 1, tw_inv);
    for (uint32_t i = threadIdx.x; i < num_lags; i += NT)
        lags_out[i] = (i < nfft) ? buf[i] * g.acorr_norm : 0.0;
    __syncthreads();
}

/* --------------------------------------------------------------------- Levinson-Durbin --- */
/* Runs on wave 0 only.  lev: a_prev[order+3], a_cur[order+3], prod[order+3], err[order+2].
 * Recursion of lpc.c:379-441 up to `upto` (<= order); the gamma dot product is accumulated in
 * index order.  On return the predictor a_{upto}[0..upto] is in *row_out (points into lev). */
__device__ void levinson_wave0(const double *r, uint32_t upto, double *lev, uint32_t stride,
                               double *err, double **row_out)
{
    const uint32_t lane = threadIdx.x;
    double *a_prev = lev, *a_cur = lev + stride, *prod = lev + 2 * stride;
    if (fabs(r[0]) < (double)FLT_EPSILON) {
        for (uint32_t i = lane; i < upto + 2; i += WAVE) { a_prev[i] = 0.0; }
        for (uint32_t i = lane; i < upto + 1; i += WAVE) err[i] = r[0];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        *row_out = a_prev;
        return;
    }
    if (lane == 0) {
        const double a1 = -r[1] / r[0];
        a_prev[0] = 1.0;
        a_prev[1] = a1;
        a_prev[2] = 0.0;
        err[0] = r[0];
        err[1] = r[0] + r[1] * a1;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (uint32_t k = 1; k < upto; k++) {
        for (uint32_t i = lane; i < k + 1; i += WAVE) prod[i] = a_prev[i] * r[k + 1 - i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        double gamma = 0.0;
        for (uint32_t i = 0; i < k + 1; i++) gamma += prod[i];   /* every lane: same order, same value */
        const double ek = err[k];
        gamma /= -ek;
        if (lane == 0) err[k + 1] = ek * (1.0 - gamma * gamma);
        for (uint32_t i = lane; i < k + 2; i += WAVE) a_cur[i] = a_prev[i] + gamma * a_prev[k + 1 - i];
        if (lane == 0) a_cur[k + 2] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        double *t = a_prev; a_prev = a_cur; a_cur = t;
    }
    *row_out = a_prev;
}

/* ------------------------------------------------------------ order choice (H2: libm) ----- */
/* srla_encoder.c:873-885 */
__device__ __forceinline__ double geometric_entropy(double mean_abs, uint32_t bps)
{
    const double intmean = mean_abs * (double)(1 << (bps - 1));
    const double rho = 1.0 / (1.0 + intmean);
    const double invrho = 1.0 - rho;
    if (mean_abs < 1e-16) return 0.0;
    return -(invrho * (log(invrho) * 1.4426950408889634) + rho * (log(rho) * 1.4426950408889634)) / rho;
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

/* ------------------------------------------------------------------------- kernel A ------- */
template <int R>
__global__ __launch_bounds__(NT) void srla_analyze_items(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    uint32_t item_first, const SrlaGeom *__restrict__ geoms, const cplx *__restrict__ twiddles,
    SrlaLdsPlan plan, const double *__restrict__ rice_thresholds, const uint8_t *__restrict__ huff_len,
    int32_t *__restrict__ res_ws, SrlaItemResult *__restrict__ results, double *__restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int32_t *y = (int32_t *)(lds + plan.y_off);
    double *fftbuf = (double *)(lds + plan.fft_off);
    double *lev = (double *)(lds + plan.lev_off);
    double *means = (double *)(lds + plan.means_off);
    Small *sm = (Small *)(lds + plan.small_off);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t item_idx = item_first + blockIdx.x;
    const SrlaItemDesc it = items[item_idx];
    const SrlaGeom g = geoms[it.geom];
    const uint32_t n = it.n, bps = jp.bits_per_sample;
    const int32_t *in = input + it.sample_off;
    const double norm_bps = __builtin_ldexp(1.0, -(int)(bps - 1));
    const cplx *twbase = twiddles + g.tw_off;
    double *dbg_item = dbg ? dbg + (size_t)item_idx * SRLA_DBG_STRIDE : nullptr;

#define STAMP(k) do { if (dbg_item && tid == 0) dbg_item[SRLA_DBG_TIMES + (k)] = (double)wall_clock64(); } while (0)
    STAMP(0);
    if (tid == 0) { sm->flags = (n & 1u) ? SRLA_ITEM_ODD_LENGTH : 0u; sm->max_u = 0; sm->seq_path = 0; sm->period = 0; }

    /* ---- stage the variant in LDS; exact integer correlations for the pre-emphasis tap ---- */
    uint32_t absmax = 0;
    for (uint32_t i = tid; i < n; i += NT) {
        const int32_t v = load_variant(in, jp, it.variant, i);
        y[i] = v;
        const uint32_t a = (v < 0) ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
        absmax = (a > absmax) ? a : absmax;
    }
    __syncthreads();
    long long r0 = 0, r1 = 0;
    for (uint32_t i = tid; i < n; i += NT) {
        const long long c = y[i];
        r0 += c * c;
        if (i + 1 < n) r1 += c * (long long)y[i + 1];
    }
    r0 = wave_sum_i64(r0); r1 = wave_sum_i64(r1); absmax = wave_max_u32(absmax);
    if (lane == 0) { sm->lscratch[wave] = r0; sm->lscratch[4 + wave] = r1; sm->uscratch[wave] = absmax; }
    __syncthreads();
    if (tid == 0) {
        long long s0 = 0, s1 = 0; uint32_t am = 0;
        for (int w = 0; w < NWAVES; w++) { s0 += sm->lscratch[w]; s1 += sm->lscratch[4 + w]; am = (sm->uscratch[w] > am) ? sm->uscratch[w] : am; }
        if (am == 0) sm->flags |= SRLA_ITEM_INPUT_ZERO;
        double d0, d1;
        if (am < (1u << 23) && s0 < (1LL << 53)) {
            /* every partial sum of the reference's double accumulation is an exactly
             * representable integer, so the summation order does not matter */
            d0 = (double)s0; d1 = (double)s1;
        } else {
            /* srla_utility.c:226-240 literally (rounding depends on the order) */
            double curr = y[0], succ = y[1];
            d0 = 0.0; d1 = 0.0;
            for (uint32_t i = 0; i + 2 < n; i++) {
                const double nn = y[i + 2];
                d0 += curr * curr; d1 += curr * succ; curr = succ; succ = nn;
            }
            d0 += curr * curr; d1 += curr * succ; curr = succ; d0 += curr * curr;
            sm->seq_path = 1;
        }
        int32_t c = 0;
        if (!(d0 < 1e-6)) {
            c = (int32_t)round_half_away((d1 / d0) * 16.0);
            c = (c < -16) ? -16 : ((c > 15) ? 15 : c);
        }
        sm->preemph_coef = c;
    }
    __syncthreads();
    {
        /* y[i] -= (x[i-1] * coef) >> 4, x[-1] = x[0]; the neighbour is re-derived from the
         * input so no thread reads an LDS word another thread rewrites (srla_utility.c:342) */
        const int32_t c = sm->preemph_coef;
        for (uint32_t i = tid; i < n; i += NT) {
            const int32_t cur = y[i];
            const int32_t prev = (i == 0) ? cur : load_variant(in, jp, it.variant, i - 1);
            y[i] = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)c) >> 4));
        }
    }
    const int32_t preemph_prev = load_variant(in, jp, it.variant, 0);
    __syncthreads();

    STAMP(1);
    /* ---- long-term (pitch) predictor, srla_encoder.c:1010-1057 + lpc.c:1558-1649 ---------- */
    if (jp.ltp_order > 0) {
        windowed_autocorr<R>(y, fftbuf, g, norm_bps, twbase, sm->lags, SRLA_LTP_LAGS + 2);
        if (tid == 0) {
            double *r = sm->lags;
            r[SRLA_LTP_LAGS] = 0.0; r[SRLA_LTP_LAGS + 1] = 0.0;   /* never-written words of the reference's buffer */
            uint32_t period = 0;
            if (!(fabs(r[0]) <= (double)FLT_MIN)) {
                /* lpc.c:1473-1555 */
                uint32_t cand[20]; uint32_t ncand = 0, i = SRLA_LTP_MIN_PERIOD; double best = 0.0;
                const uint32_t maxp = SRLA_LTP_MAX_PERIOD;
                while (i < maxp && ncand < 20) {
                    uint32_t start, end, peak_at = 0; double peak = 0.0;
                    for (start = i; start < maxp; start++) if (r[start - 1] < 0.0 && r[start] > 0.0) break;
                    for (end = start + 1; end < maxp - 1; end++) if (r[end] > 0.0 && r[end + 1] < 0.0) break;
                    for (uint32_t j = start; j <= end; j++)
                        if (r[j] > r[j - 1] && r[j] > r[j + 1] && r[j] > peak) { peak_at = j; peak = r[j]; }
                    if (peak_at != 0) { cand[ncand++] = peak_at; if (peak > best) best = peak; }
                    i = end + 1;
                }
                if (ncand > 0 && !(best < 0.1 * r[0])) {
                    for (uint32_t k = 0; k < ncand; k++)
                        if (r[cand[k]] >= 0.9 * best) { period = cand[k]; break; }
                }
                if (period < (jp.ltp_order / 2) + 1) period = 0;
            }
            if (period > 0) {
                const int dim = (int)jp.ltp_order;
                double am[3][3], inv_diag[3], xs[3];
                bool ok = true;
                r[0] *= (1.0 + 1e-5);
                for (int j = 0; j < dim; j++) for (int k = j; k < dim; k++) am[j][k] = am[k][j] = r[k - j];
                for (int i2 = 0; i2 < dim && ok; i2++) {
                    double sum = am[i2][i2];
                    for (int k = i2 - 1; k >= 0; k--) sum -= am[i2][k] * am[i2][k];
                    if (sum <= 0.0) { ok = false; break; }
                    inv_diag[i2] = inv_sqrt_cr(sum);
                    for (int j = i2 + 1; j < dim; j++) {
                        sum = am[i2][j];
                        for (int k = i2 - 1; k >= 0; k--) sum -= am[i2][k] * am[j][k];
                        am[j][i2] = sum * inv_diag[i2];
                    }
                }
                if (!ok) { sm->flags |= SRLA_ITEM_LTP_FAIL; period = 0; }
                else {
                    const double *b = &r[period - jp.ltp_order / 2];
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
                    int32_t q[3] = { 0, 0, 0 };
                    for (int i2 = 0; i2 < dim; i2++) {
                        const double scaled = xs[i2] * 32.0;
                        const double fr = fabs(scaled) + 0.5;
                        if (fabs(fr - floor(fr + 0.5)) < 1e-9 && fabs(scaled) < 40.0) sm->flags |= SRLA_ITEM_LTP_TIE;
                        int32_t c = (int32_t)round_half_away(scaled);
                        c = (c < -32) ? -32 : ((c > 31) ? 31 : c);
                        q[i2] = c;
                    }
                    for (int i2 = 0; i2 < dim / 2; i2++) { const int32_t t = q[i2]; q[i2] = q[dim - 1 - i2]; q[dim - 1 - i2] = t; }
                    sm->ltp_coef[0] = q[0]; sm->ltp_coef[1] = q[1]; sm->ltp_coef[2] = q[2];
                }
            }
            sm->period = period;
        }
        __syncthreads();
        if (dbg_item) for (uint32_t i = tid; i < SRLA_LTP_LAGS; i += NT) dbg_item[SRLA_DBG_LTPLAGS + i] = sm->lags[i];
        const uint32_t period = sm->period;
        if (period > 0) {
            /* srla_lpc_predict.c:267-294, out of place through the (idle) FFT buffer */
            int32_t *tmp = (int32_t *)fftbuf;
            const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
            const int32_t c0 = sm->ltp_coef[0], c1 = sm->ltp_coef[1], c2 = sm->ltp_coef[2];
            for (uint32_t i = tid; i < n; i += NT) {
                int32_t v = y[i];
                if (i >= period + half_order + 1) {
                    const uint32_t base = i - period - half_order;
                    uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)y[base];
                    if (taps == 3) acc += (uint32_t)c1 * (uint32_t)y[base + 1] + (uint32_t)c2 * (uint32_t)y[base + 2];
                    v = (int32_t)((uint32_t)v - (uint32_t)((int32_t)acc >> 5));
                }
                tmp[i] = v;
            }
            __syncthreads();
            for (uint32_t i = tid; i < n; i += NT) y[i] = tmp[i];
            __syncthreads();
        }
    }

    STAMP(2);
    /* ---- LPC analysis ---------------------------------------------------------------------- */
    const uint32_t pmax = jp.max_order;
    uint32_t order = 0;
    if (pmax > 0) {
        windowed_autocorr<R>(y, fftbuf, g, norm_bps, twbase, sm->lags, pmax + 1);
        if (dbg_item) for (uint32_t i = tid; i < pmax + 1; i += NT) dbg_item[SRLA_DBG_LAGS + i] = sm->lags[i];
        STAMP(3);
        const uint32_t stride = pmax + 3;
        double *err = lev + 3 * stride;
        if (wave == 0) {
            if (lane == 0) sm->lags[0] *= (1.0 + 1e-5);     /* ridge, lpc.c:483 */
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            double *row;
            levinson_wave0(sm->lags, pmax, lev, stride, err, &row);
            /* window power compensation (lpc.c:490-497) and the code-length estimate per order
             * (srla_encoder.c:934-957), one order per lane */
            double best_len = (double)FLT_MAX, second = (double)FLT_MAX; uint32_t best_order = 0;
            for (uint32_t base = 0; base <= pmax; base += WAVE) {
                const uint32_t o = base + lane;
                double len = __builtin_nan("");
                if (o <= pmax) {
                    const double ev = err[o] * g.welch_comp;
                    err[o] = ev;
                    if (o >= 1) {
                        const double mabse = 2.0 * sqrt(ev / 2.0);
                        len = geometric_entropy(mabse, bps) * (double)n;
                        len += (double)(8u * o);
                    }
                }
                if (dbg_item && o <= pmax) { dbg_item[SRLA_DBG_ERRVARS + o] = err[o]; dbg_item[SRLA_DBG_LENS + o] = len; }
                /* first strict minimum == lowest order among the smallest lengths */
                for (uint32_t l = 0; l < WAVE; l++) {
                    const double cl = __shfl(len, (int)l, WAVE);
                    if (base + l >= 1 && base + l <= pmax) {
                        if (best_len > cl) { second = best_len; best_len = cl; best_order = base + l; }
                        else if (second > cl) second = cl;
                    }
                }
            }
            if (jp.order_fixed) best_order = pmax;
            else if (best_order != 0 && (second - best_len) <= 1e-9 * fabs(best_len) + 1e-9) {
                if (lane == 0) sm->flags |= SRLA_ITEM_ORDER_TIE;
            }
            if (it.forced_order >= 0) best_order = (uint32_t)it.forced_order;
            order = best_order;
            /* recompute the recursion up to the chosen order to get that order's predictor */
            if (order > 0) {
                levinson_wave0(sm->lags, order, lev, stride, lev + 4 * stride, &row);
                if (lane == 0) {
                    /* 8-bit quantisation with error feedback from the last tap (lpc.c:1341-1405) */
                    const double *cf = row + 1;
                    double maxabs = 0.0;
                    for (uint32_t i = 0; i < order; i++) { const double a = fabs(cf[i]); if (maxabs < a) maxabs = a; }
                    uint32_t rshift;
                    if (maxabs <= 0.0078125) {
                        rshift = 8;
                        for (uint32_t i = 0; i < order; i++) sm->coef[i] = 0;
                    } else {
                        int ndigit;
                        (void)frexp(maxabs, &ndigit);
                        rshift = (uint32_t)(7 - ndigit);
                        if (rshift >= 16u) rshift = 15u;
                        const double scale = __builtin_ldexp(1.0, (int)rshift);
                        double qerr = 0.0;
                        for (int i = (int)order - 1; i >= 0; i--) {
                            qerr += cf[i] * scale;
                            int32_t q = (int32_t)round_half_away(qerr);
                            if (q >= 128) q = 127; else if (q < -128) q = -128;
                            qerr -= (double)q;
                            sm->coef[order - 1 - i] = q;       /* reversed: oldest sample first */
                        }
                    }
                    sm->rshift = rshift;
                }
            } else if (lane == 0) {
                sm->rshift = 0;
            }
            if (lane == 0) sm->order = order;
        }
        __syncthreads();
        order = sm->order;
    } else {
        if (tid == 0) { sm->order = 0; sm->rshift = 0; }
        __syncthreads();
    }
    const uint32_t rshift = sm->rshift;

    STAMP(4);
    /* ---- integer FIR residual (srla_lpc_predict.c:118-265), zig-zag copy kept in LDS -------- */
    uint32_t *u = (uint32_t *)fftbuf;
    int32_t *res_out = res_ws + it.res_off;
    {
        const int32_t half = (int32_t)(1u << ((rshift - 1u) & 31u));
        uint32_t max_u = 0;
        for (uint32_t s = tid; s < n; s += NT) {
            int32_t r;
            if (order == 0) r = y[s];
            else if (s == 0) r = y[0];
            else if (s < order) r = (int32_t)((uint32_t)y[s] - (uint32_t)y[s - 1]);
            else {
                uint32_t acc = (uint32_t)half;
                const int32_t *win = y + (s - order);
                for (uint32_t k = 0; k < order; k++) acc += (uint32_t)sm->coef[k] * (uint32_t)win[k];
                r = (int32_t)((uint32_t)y[s] + (uint32_t)((int32_t)acc >> rshift));
            }
            res_out[s] = r;
            const uint32_t z = zigzag32(r);
            u[s] = z;
            max_u = (z > max_u) ? z : max_u;
        }
        max_u = wave_max_u32(max_u);
        if (lane == 0) atomicMax(&sm->max_u, max_u);
    }
    STAMP(5);
    /* finest-level partition sums (exact integers), srla_coder.c:366-381 */
    const uint32_t mp = g.max_porder, nparts = 1u << mp, fl = g.fine_len;
    unsigned long long *sums = (unsigned long long *)(means + (nparts - 1));
    for (uint32_t p = tid; p < nparts; p += NT) sums[p] = 0ull;
    if (tid < 16) sm->level_bits[tid] = 0;
    __syncthreads();
    const uint32_t tpp = (nparts >= NT) ? 1u : (NT / nparts);     /* threads per finest partition */
    if (tpp == 1) {
        for (uint32_t p = tid; p < nparts; p += NT) {
            unsigned long long s = 0;
            const uint32_t *up = u + p * fl;
            for (uint32_t i = 0; i < fl; i++) s += up[i];
            sums[p] = s;
        }
    } else {
        const uint32_t p = tid / tpp, j = tid % tpp;
        unsigned long long s = 0;
        const uint32_t *up = u + p * fl;
        for (uint32_t i = j; i < fl; i += tpp) s += up[i];
        atomicAdd(&sums[p], s);
    }
    __syncthreads();
    for (uint32_t p = tid; p < nparts; p += NT) {
        const unsigned long long s = sums[p];
        means[(nparts - 1) + p] = (double)s / (double)fl;
    }
    __syncthreads();
    for (int lvl = (int)mp - 1; lvl >= 0; lvl--) {
        const uint32_t cnt = 1u << lvl;
        for (uint32_t p = tid; p < cnt; p += NT)
            means[(cnt - 1) + p] = (means[(2 * cnt - 1) + 2 * p] + means[(2 * cnt - 1) + 2 * p + 1]) / 2.0;
        __syncthreads();
    }
    uint32_t code_type;
    if (sm->max_u == 0) code_type = SRLA_CODE_ALLZERO;
    else if (means[0] < 2) code_type = SRLA_CODE_RICE;
    else code_type = SRLA_CODE_RECURSIVE_RICE;

    uint32_t best_porder = 0, best_bits = 0;
    if (code_type != SRLA_CODE_ALLZERO) {
        /* parameter per (level, partition) */
        for (uint32_t e = tid; e < 2 * nparts - 1; e += NT) {
            const double mean = means[e];
            uint32_t k;
            if (code_type == SRLA_CODE_RICE) {
                k = 0;   /* srla_coder.c:262-276 through the host-derived monotone thresholds */
                for (int t = 0; t < 32; t++) k += (mean >= rice_thresholds[t]) ? 1u : 0u;
            } else {
                const double gp = 0.66794162356 * (1.0 + mean);   /* srla_coder.c:298-311 */
                const uint32_t golomb = (uint32_t)((1.0 > gp) ? 1.0 : gp);
                k = 31u - (uint32_t)__clz((int)golomb);
            }
            sm->ktab[e] = (uint8_t)k;
        }
        __syncthreads();
        /* cost of every partition order in one pass over the residual */
        uint32_t acc[SRLA_MAX_PORDER + 1];
#pragma unroll
        for (int l = 0; l <= SRLA_MAX_PORDER; l++) acc[l] = 0;
        {
            uint32_t p_first, p_step, j_first, j_step;
            if (tpp == 1) { p_first = tid; p_step = NT; j_first = 0; j_step = 1; }
            else { p_first = tid / tpp; p_step = nparts; j_first = tid % tpp; j_step = tpp; }
            for (uint32_t p = p_first; p < nparts; p += p_step) {
                const uint32_t *up = u + p * fl;
                uint32_t kk[SRLA_MAX_PORDER + 1];
#pragma unroll
                for (int l = 0; l <= SRLA_MAX_PORDER; l++)
                    kk[l] = ((uint32_t)l <= mp) ? sm->ktab[((1u << l) - 1) + (p >> (mp - l))] : 0u;
                for (uint32_t i = j_first; i < fl; i += j_step) {
                    const uint32_t v = up[i];
#pragma unroll
                    for (int l = 0; l <= SRLA_MAX_PORDER; l++) {
                        if ((uint32_t)l <= mp) {
                            const uint32_t k = kk[l];
                            if (code_type == SRLA_CODE_RICE) {
                                acc[l] += 1u + k + (v >> k);                      /* srla_coder.c:327-330 */
                            } else {
                                int32_t over = (int32_t)v - (int32_t)(2u << k);  /* srla_coder.c:333-347 */
                                over = (over > 0) ? over : 0;
                                acc[l] += (k + 2u) + ((uint32_t)over >> k);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int l = 0; l <= SRLA_MAX_PORDER; l++) {
            if ((uint32_t)l <= mp) {
                const uint32_t s = wave_sum_u32(acc[l]);
                if (lane == 0) atomicAdd(&sm->level_bits[l], s);
            }
        }
        /* side information: 10 bits of partition order, 5 bits for the first parameter, then
         * zig-zag(delta) + 1 per further partition (srla_coder.c:415-427) */
        for (uint32_t e = tid; e < 2 * nparts - 1; e += NT) {
            const uint32_t lvl = 31u - (uint32_t)__clz((int)(e + 1));
            const uint32_t p = e + 1 - (1u << lvl);
            uint32_t side;
            if (p == 0) side = 10u + 5u;
            else side = zigzag32((int32_t)sm->ktab[e] - (int32_t)sm->ktab[e - 1]) + 1u;
            atomicAdd(&sm->level_bits[lvl], side);
        }
        __syncthreads();
        best_bits = 0xFFFFFFFFu;
        for (uint32_t l = 0; l <= mp; l++) {
            const uint32_t b = sm->level_bits[l];
            if (b < best_bits) { best_bits = b; best_porder = l; }
        }
    }
    const uint32_t res_bits = best_bits + 2u;
    STAMP(6);

    /* ---- coefficient cost (srla_encoder.c:1121-1187) and the item record -------------------- */
    SrlaItemResult *out = &results[item_idx];
    if (code_type != SRLA_CODE_ALLZERO)
        for (uint32_t p = tid; p < (1u << best_porder); p += NT) out->kparam[p] = sm->ktab[((1u << best_porder) - 1) + p];
    for (uint32_t k = tid; k < order; k += NT) out->lpc_coef[k] = (int8_t)sm->coef[k];
    if (wave == 0) {
        uint32_t plain = 0, summed = 0, overflow = 0;
        for (uint32_t k = lane; k < order; k += WAVE) {
            const int32_t c = sm->coef[k];
            plain += huff_len[zigzag32(c)];
            if (k == 0) summed += huff_len[zigzag32(c)];
            else {
                const uint32_t z = zigzag32(c + sm->coef[k - 1]);
                if (z >= 256u) overflow = 1; else summed += huff_len[256 + z];
            }
        }
        plain = wave_sum_u32(plain); summed = wave_sum_u32(summed); overflow = wave_sum_u32(overflow);
        if (lane == 0) {
            uint32_t use_sum = 0, coef_bits = 0;
            if (order > 0) {
                use_sum = (overflow == 0 && (order == 1 || summed < plain)) ? 1u : 0u;
                coef_bits = use_sum ? summed : plain;
            }
            uint32_t bits = res_bits;
            bits += bps + 1u;              /* pre-emphasis state */
            bits += 5u;                    /* pre-emphasis tap   */
            bits += 8u + 4u + 1u;          /* order, shift, sum flag */
            bits += coef_bits;
            bits += 1u;                    /* LTP flag */
            const uint32_t period = sm->period;
            if (period > 0) bits += 1u + 8u + jp.ltp_order * 6u;
            out->preemph_prev = preemph_prev;
            out->preemph_coef = sm->preemph_coef;
            out->lpc_order = order;
            out->lpc_rshift = rshift;
            out->use_sum = use_sum;
            out->ltp_period = period;
            out->ltp_coef[0] = (period > 0) ? sm->ltp_coef[0] : 0;
            out->ltp_coef[1] = (period > 0) ? sm->ltp_coef[1] : 0;
            out->ltp_coef[2] = (period > 0) ? sm->ltp_coef[2] : 0;
            out->code_length = bits;
            out->res_code_type = code_type;
            out->res_porder = best_porder;
            out->res_bits = res_bits;
            out->flags = sm->flags;
            out->pad[0] = 0; out->pad[1] = 0;
        }
    }
}

/* ------------------------------------------------------------------------- kernel B ------- */
/* One thread per window.  Block cost: ComputeBlockSize (srla_encoder.c:1477-1546) on top of the
 * stereo decision of ComputeCoefficients (:1275-1327); path: ApplyDijkstraMethod (:249-307)
 * with its exact tie behaviour; partition read-back (:397-421). */
__global__ void srla_price_windows(SrlaJobParams jp, const SrlaWindowDesc *__restrict__ windows,
                                   const SrlaCandDesc *__restrict__ cands,
                                   const SrlaItemResult *__restrict__ results,
                                   SrlaBlockRecord *__restrict__ blocks, uint32_t *__restrict__ cand_bytes)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= jp.num_windows) return;
    const SrlaWindowDesc wd = windows[w];
    const uint32_t nch = jp.num_channels, bps = jp.bits_per_sample, nodes = wd.num_nodes;

    /* candidate costs */
    for (uint32_t c = 0; c < wd.num_cands; c++) {
        const SrlaCandDesc cd = cands[wd.cand_base + c];
        const uint32_t raw_bytes = 11u + (bps * cd.n * nch) / 8u;
        uint32_t bytes = raw_bytes, type = SRLA_BLOCK_RAW, method = 0;
        if (cd.item_base != 0xFFFFFFFFu) {
            bool silent = true;
            for (uint32_t ch = 0; ch < nch; ch++)
                if (!(results[cd.item_base + ch].flags & SRLA_ITEM_INPUT_ZERO)) { silent = false; break; }
            if (silent) { type = SRLA_BLOCK_SILENT; bytes = 11u; }
            else {
                uint32_t bits;
                if (nch == 1) { bits = results[cd.item_base].code_length; method = 0; }
                else {
                    const uint32_t l = results[cd.item_base + 0].code_length, r = results[cd.item_base + 1].code_length;
                    const uint32_t m = results[cd.item_base + nch].code_length, s = results[cd.item_base + nch + 1].code_length;
                    uint32_t len[4] = { l + r, m + s, l + s, r + s };
                    bits = len[0]; method = 0;
                    for (uint32_t k = 1; k < 4; k++) if (bits > len[k]) { bits = len[k]; method = k; }
                }
                bits += 2u;
                bits = ((bits + 7u) / 8u) * 8u;
                if (bits >= bps * cd.n * nch) { type = SRLA_BLOCK_RAW; bytes = raw_bytes; }
                else { type = SRLA_BLOCK_COMPRESS; bytes = 11u + bits / 8u; }
            }
        }
        cand_bytes[wd.cand_base + c] = bytes | (type << 28) | (method << 30);
    }
    /* shortest path 0 -> nodes-1 over candidate edges */
    long long cost[SRLA_MAX_NODES];
    uint32_t path[SRLA_MAX_NODES];
    uint32_t via_cand[SRLA_MAX_NODES];
    uint8_t used[SRLA_MAX_NODES];
    for (uint32_t i = 0; i < nodes; i++) { cost[i] = SRLA_BIG_WEIGHT; path[i] = 0xFFFFFFFFu; used[i] = 0; via_cand[i] = 0xFFFFFFFFu; }
    cost[0] = 0;
    uint32_t target = 0;
    for (uint32_t guard = 0; guard <= nodes; guard++) {
        long long mn = SRLA_BIG_WEIGHT;
        for (uint32_t i = 0; i < nodes; i++) if (!used[i] && mn > cost[i]) { mn = cost[i]; target = i; }
        if (target == nodes - 1) break;
        /* relax every edge leaving `target` (edges are the candidates with node_i == target) */
        for (uint32_t c = 0; c < wd.num_cands; c++) {
            const SrlaCandDesc cd = cands[wd.cand_base + c];
            if (cd.node_i != target) continue;
            const long long via = (long long)(cand_bytes[wd.cand_base + c] & 0x0FFFFFFFu) + cost[target];
            if (cost[cd.node_j] > via) { cost[cd.node_j] = via; path[cd.node_j] = target; via_cand[cd.node_j] = c; }
        }
        used[target] = 1;
    }
    /* read the partition back and emit block records in stream order */
    uint32_t count = 0;
    for (uint32_t node = nodes - 1; node != 0 && path[node] != 0xFFFFFFFFu; node = path[node]) count++;
    uint32_t node = nodes - 1;
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t c = via_cand[node];
        const SrlaCandDesc cd = cands[wd.cand_base + c];
        const uint32_t packed = cand_bytes[wd.cand_base + c];
        SrlaBlockRecord rec;
        rec.valid = 1;
        rec.sample_off = cd.sample_off;
        rec.n = cd.n;
        rec.block_type = (packed >> 28) & 3u;
        rec.ch_method = (packed >> 30) & 3u;
        rec.bytes = packed & 0x0FFFFFFFu;
        if (rec.block_type == SRLA_BLOCK_COMPRESS && nch > 2) {
            /* ComputeBlockSize prices only the first two channels (srla_encoder.c:1287-1301) and that
             * price drives the search; EncodeBlock then writes every channel and applies its RAW
             * fall-back to the size actually written (srla_encoder.c:1605-1611) */
            uint32_t bits = 2u;
            for (uint32_t ch = 2; ch < nch; ch++) bits += results[cd.item_base + ch].code_length;
            const uint32_t l = results[cd.item_base + 0].code_length, r = results[cd.item_base + 1].code_length;
            const uint32_t m = results[cd.item_base + nch].code_length, s2 = results[cd.item_base + nch + 1].code_length;
            const uint32_t len[4] = { l + r, m + s2, l + s2, r + s2 };
            bits += len[rec.ch_method];
            const uint32_t payload = (bits + 7u) / 8u;
            if (8u * payload >= bps * cd.n * nch) { rec.block_type = SRLA_BLOCK_RAW; rec.bytes = 11u + (bps * cd.n * nch) / 8u; }
            else rec.bytes = 11u + payload;
        }
        for (uint32_t ch = 0; ch < SRLA_MAX_CH; ch++) rec.item[ch] = 0xFFFFFFFFu;
        if (rec.block_type == SRLA_BLOCK_COMPRESS) {
            for (uint32_t ch = 0; ch < nch; ch++) rec.item[ch] = cd.item_base + ch;
            if (nch >= 2) {
                const uint32_t mi = cd.item_base + nch, si = cd.item_base + nch + 1;
                if (rec.ch_method == 1) { rec.item[0] = mi; rec.item[1] = si; }
                else if (rec.ch_method == 2) { rec.item[1] = si; }
                else if (rec.ch_method == 3) { rec.item[0] = si; }
            }
        }
        rec.pad[0] = 0; rec.pad[1] = 0;
        blocks[wd.block_base + (count - 1 - k)] = rec;
        node = path[node];
    }
    for (uint32_t k = count; k < nodes - 1; k++) blocks[wd.block_base + k].valid = 0;
}

/* ------------------------------------------------------------------------- kernel C ------- */
/* grid = (block slots, channels).  Residuals of compress blocks (raw shifted-back samples for
 * RAW blocks) land at the block's own sample positions of out[ch][...]; the chosen item records
 * are copied next to them. */
__global__ __launch_bounds__(NT) void srla_gather_blocks(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    const SrlaBlockRecord *__restrict__ blocks, const SrlaItemResult *__restrict__ results,
    const int32_t *__restrict__ res_ws, int32_t *__restrict__ out, SrlaItemResult *__restrict__ chan_out)
{
    const uint32_t slot = blockIdx.x, ch = blockIdx.y;
    const SrlaBlockRecord rec = blocks[slot];
    if (!rec.valid) return;
    int32_t *dst = out + (size_t)ch * jp.out_stride + rec.sample_off;
    if (rec.block_type == SRLA_BLOCK_COMPRESS) {
        const uint32_t item = rec.item[ch];
        const int32_t *src = res_ws + items[item].res_off;
        for (uint32_t i = threadIdx.x; i < rec.n; i += NT) dst[i] = src[i];
        /* item record: 1344 bytes = 336 words */
        const uint32_t *rs = (const uint32_t *)&results[item];
        uint32_t *rd = (uint32_t *)&chan_out[(size_t)slot * jp.num_channels + ch];
        for (uint32_t i = threadIdx.x; i < sizeof(SrlaItemResult) / 4; i += NT) rd[i] = rs[i];
    } else if (rec.block_type == SRLA_BLOCK_RAW) {
        const int32_t *src = input + (size_t)ch * jp.channel_stride + rec.sample_off;
        for (uint32_t i = threadIdx.x; i < rec.n; i += NT) dst[i] = src[i];
    }
}

/* ------------------------------------------------------------------- offset left shift ---- */
__global__ __launch_bounds__(NT) void srla_or_reduce(const int32_t *__restrict__ in, size_t count, uint32_t *__restrict__ out)
{
    uint32_t m = 0;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < count; i += (size_t)gridDim.x * NT) m |= (uint32_t)in[i];
    for (int off = 32; off > 0; off >>= 1) m |= __shfl_down(m, off, WAVE);
    if ((threadIdx.x & 63) == 0 && m) atomicOr(out, m);
}

/* --------------------------------------------------------------------------- launchers ---- */
extern "C" int srla_launch_analyze(hipStream_t stream, int regs_per_thread_class, uint32_t num_items_in_group,
                                   const SrlaJobParams *jp, const int32_t *input, const SrlaItemDesc *items,
                                   uint32_t item_first, const SrlaGeom *geoms, const void *twiddles,
                                   const SrlaLdsPlan *plan, const double *rice_thresholds, const uint8_t *huff_len,
                                   int32_t *res_ws, SrlaItemResult *results, double *dbg)
{
    if (num_items_in_group == 0) return 0;
    dim3 grid(num_items_in_group), block(NT);
#define LAUNCH(RR)                                                                                          \
    do {                                                                                                    \
        static bool attr_set_##RR = false;                                                                  \
        if (!attr_set_##RR) {                                                                               \
            (void)hipFuncSetAttribute((const void *)srla_analyze_items<RR>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
            attr_set_##RR = true;                                                                           \
        }                                                                                                   \
        hipLaunchKernelGGL(srla_analyze_items<RR>, grid, block, plan->total, stream, *jp, input, items,     \
                           item_first, geoms, (const cplx *)twiddles, *plan, rice_thresholds, huff_len,     \
                           res_ws, results, dbg);                                                           \
    } while (0)
    switch (regs_per_thread_class) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    default: return -1;
    }
#undef LAUNCH
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" int srla_launch_price(hipStream_t stream, const SrlaJobParams *jp, const SrlaWindowDesc *windows,
                                 const SrlaCandDesc *cands, const SrlaItemResult *results,
                                 SrlaBlockRecord *blocks, uint32_t *cand_bytes)
{
    if (jp->num_windows == 0) return 0;
    const uint32_t bs = 64;
    hipLaunchKernelGGL(srla_price_windows, dim3((jp->num_windows + bs - 1) / bs), dim3(bs), 0, stream,
                       *jp, windows, cands, results, blocks, cand_bytes);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" int srla_launch_gather(hipStream_t stream, const SrlaJobParams *jp, uint32_t num_slots,
                                  const int32_t *input, const SrlaItemDesc *items, const SrlaBlockRecord *blocks,
                                  const SrlaItemResult *results, const int32_t *res_ws, int32_t *out,
                                  SrlaItemResult *chan_out)
{
    if (num_slots == 0) return 0;
    hipLaunchKernelGGL(srla_gather_blocks, dim3(num_slots, jp->num_channels), dim3(NT), 0, stream,
                       *jp, input, items, blocks, results, res_ws, out, chan_out);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" int srla_launch_or_reduce(hipStream_t stream, const int32_t *in, size_t count, uint32_t *out)
{
    size_t blocks = (count + NT * 16 - 1) / (NT * 16);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(srla_or_reduce, dim3((uint32_t)blocks), dim3(NT), 0, stream, in, count, out);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}
