/*
 * srla_oracle.c -- CPU oracle for the SRLA encode hot path (TEST INFRASTRUCTURE ONLY).
 *
 * A from-scratch restatement, in plain C99, of what the reference encoder computes between
 * SRLAEncoder_EncodeWhole's input samples and its output bytes, plus a decoder used to close
 * the loop.  Every function cites the reference file:line whose arithmetic it reproduces
 * (paths relative to /root/reference).  Compile with -ffp-contract=off: the reference is
 * C90 (no FMA contraction) and every integer in the stream is decided by double arithmetic.
 *
 * Parity pinned against the compiled reference (see srla_oracle.h).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#include "srla_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "huffman_codes.inc"

/* ---- format constants (libs/srla_internal/include/srla_internal.h:9-35, include/srla.h:7-22) */
#define FMT_VERSION        10u
#define CODEC_VERSION      18u
#define HEADER_BYTES       30u
#define BLOCK_HEADER_BYTES 11u
#define PREEMPH_SHIFT      4
#define LPC_COEF_BITS      8
#define LPC_RSHIFT_BITS    4
#define LPC_ORDER_BITS     8
#define LTP_ORDER_BITS     1
#define LTP_PERIOD_BITS    8
#define LTP_COEF_BITS      6
#define LTP_MIN_PERIOD     8u
#define LTP_MAX_PERIOD     (LTP_MIN_PERIOD + (1u << LTP_PERIOD_BITS) - 2u) /* 262 */
#define RIDGE              1e-5
#define PORDER_BITS        10   /* srla_coder.c:18 */
#define RICE_PARAM_BITS    5    /* srla_coder.c:22 */
#define BIG_WEIGHT         ((double)(1UL << 24)) /* srla_encoder.c:17 */

enum { BLOCK_COMPRESS = 0, BLOCK_SILENT = 1, BLOCK_RAW = 2 };
enum { CODE_RICE = 0, CODE_RECURSIVE_RICE = 1, CODE_ALLZERO = 2 };

/* max LPC order per preset (srla_internal.c:30-38); preset 0 uses the fixed maximum order
 * (= 0), all others pick the order by estimated code length. */
static const uint32_t k_preset_order[7] = { 0, 8, 16, 32, 64, 128, 255 };

struct Oracle {
    OracleConfig cfg;
    uint32_t offset_lshift;
    uint32_t svr_iterations; /* --svr-filter-learning-iteration (0: off, the default) */
    uint32_t fft_cap;      /* next pow2 >= max_block */
    double *fftbuf;        /* LPCCalculator::buffer: persistent across calls (lpc.c:58,211)   */
    double *fftwork;       /* LPCCalculator::work_buffer                                      */
    double *acorr;         /* LPCCalculator::auto_corr: persistent, never cleared (lpc.c:55)  */
    double *dsig;          /* encoder->buffer_double                                          */
    double *coefs;         /* [255][255] all-order predictor rows                             */
    double error_vars[ORACLE_MAX_ORDER + 1];
    int32_t *work_int[4];  /* L, R, M, S working copies      */
    int32_t *work_res[4];  /* L, R, M, S residuals           */
    int32_t *extra_int;    /* channels >= 2                  */
    int32_t *extra_res[ORACLE_MAX_CHANNELS];
};

static uint32_t zigzag(int32_t s) { return ((uint32_t)s << 1) ^ (uint32_t)(-(int32_t)(s < 0)); }
static int32_t unzigzag(uint32_t u) { return (int32_t)(u >> 1) ^ -(int32_t)(u & 1u); }
static uint32_t next_pow2(uint32_t v)
{
    /* lpc.c:79-89 */
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}
static double round_half_away(double d)
{
    /* srla_utility.c:22-25, lpc.c:65-68 */
    return (d >= 0.0) ? floor(d + 0.5) : -floor(-d + 0.5);
}
static double log2_via_ln(double d) { return log(d) * 1.4426950408889634; } /* srla_utility.c:28-33 */

/* =========================================================================================
 * Bit writer / reader: MSB-first, zero padded to a byte at flush
 * (libs/bit_stream/include/bit_stream.h:245-307, 400-437).
 * ========================================================================================= */
typedef struct { uint8_t *p; uint8_t *base; uint64_t acc; uint32_t nacc; } BitW;

static void bw_open(BitW *w, uint8_t *mem) { w->p = w->base = mem; w->acc = 0; w->nacc = 0; }
static void bw_put(BitW *w, uint32_t val, uint32_t nbits)
{
    if (nbits == 0) return;
    if (nbits < 32) val &= (1u << nbits) - 1u;
    w->acc = (w->acc << nbits) | val;
    w->nacc += nbits;
    while (w->nacc >= 8) {
        w->nacc -= 8;
        *w->p++ = (uint8_t)(w->acc >> w->nacc);
    }
}
static void bw_zero_run(BitW *w, uint32_t run)
{
    /* `run` zeros then a one; bit_stream.h:290-307 */
    while (run >= 24) { bw_put(w, 0, 24); run -= 24; }
    bw_put(w, 1, run + 1);
}
static uint32_t bw_flush(BitW *w)
{
    if (w->nacc > 0) { *w->p++ = (uint8_t)(w->acc << (8 - w->nacc)); w->nacc = 0; }
    return (uint32_t)(w->p - w->base);
}

typedef struct { const uint8_t *p; const uint8_t *end; uint64_t acc; uint32_t nacc; } BitR;
static void br_open(BitR *r, const uint8_t *mem, uint32_t size) { r->p = mem; r->end = mem + size; r->acc = 0; r->nacc = 0; }
static uint32_t br_get(BitR *r, uint32_t nbits)
{
    uint32_t v;
    if (nbits == 0) return 0;
    while (r->nacc < nbits) {
        r->acc = (r->acc << 8) | (uint64_t)((r->p < r->end) ? *r->p : 0);
        r->p++;
        r->nacc += 8;
    }
    r->nacc -= nbits;
    v = (uint32_t)((r->acc >> r->nacc) & ((nbits < 32) ? ((1ull << nbits) - 1ull) : 0xFFFFFFFFull));
    return v;
}
static uint32_t br_zero_run(BitR *r)
{
    uint32_t run = 0;
    while (br_get(r, 1) == 0) {
        run++;
        if (r->p > r->end + 8) break; /* corrupt stream guard */
    }
    return run;
}
/* bytes consumed once the reader is aligned back to a byte boundary (bit_stream.h:404-408) */
static uint32_t br_tell_aligned(const BitR *r, const uint8_t *base)
{
    return (uint32_t)((r->p - base) - (r->nacc >> 3));
}

static void put_u16be(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
static void put_u32be(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static uint32_t get_u16be(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }
static uint32_t get_u32be(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* =========================================================================================
 * Small utilities
 * ========================================================================================= */
uint16_t oracle_fletcher16(const uint8_t *data, uint32_t size)
{
    /* srla_utility.c:36-60: c0/c1 folded mod 255 every 5802 bytes with (x + x/255) & 0xFF */
    uint32_t c0 = 0, c1 = 0;
    while (size > 0) {
        uint32_t chunk = (size < 5802u) ? size : 5802u;
        size -= chunk;
        while (chunk--) { c0 += *data++; c1 += c0; }
        c0 = (c0 + c0 / 255u) & 0xFFu;
        c1 = (c1 + c1 / 255u) & 0xFFu;
    }
    return (uint16_t)((c1 << 8) | c0);
}

uint32_t oracle_offset_lshift(const int32_t *const *input, uint32_t num_channels, uint32_t num_samples)
{
    /* srla_utility.c:177-203: trailing zero count of the OR of all samples (0 if all zero) */
    uint32_t mask = 0, ch, i, tz = 0;
    for (ch = 0; ch < num_channels; ch++)
        for (i = 0; i < num_samples; i++) mask |= (uint32_t)input[ch][i];
    if (mask == 0) return 0;
    while (((mask >> tz) & 1u) == 0) tz++;
    return tz;
}

void oracle_lr_to_ms(int32_t *ch0, int32_t *ch1, uint32_t n)
{
    /* srla_utility.c:91-103: S = R - L, then M = L + (S >> 1) */
    uint32_t i;
    for (i = 0; i < n; i++) {
        ch1[i] = (int32_t)((uint32_t)ch1[i] - (uint32_t)ch0[i]);
        ch0[i] = (int32_t)((uint32_t)ch0[i] + (uint32_t)(ch1[i] >> 1));
    }
}

int32_t oracle_preemphasis_coef(const int32_t *x, uint32_t n)
{
    /* srla_utility.c:214-256: r0 = sum x[i]^2, r1 = sum x[i]x[i+1] accumulated in double in
     * index order; coef = clip(round(16 * r1 / r0), -16, 15); 0 when r0 < 1e-6. */
    double r0 = 0.0, r1 = 0.0, curr, succ;
    uint32_t i;
    int32_t c;
    curr = x[0];
    succ = x[1];
    for (i = 0; i + 2 < n; i++) {
        const double nextnext = x[i + 2];
        r0 += curr * curr;
        r1 += curr * succ;
        curr = succ;
        succ = nextnext;
    }
    r0 += curr * curr;
    r1 += curr * succ;
    curr = succ;
    r0 += curr * curr;
    if (r0 < 1e-6) return 0;
    c = (int32_t)round_half_away((r1 / r0) * pow(2.0f, PREEMPH_SHIFT));
    if (c < -(1 << PREEMPH_SHIFT)) c = -(1 << PREEMPH_SHIFT);
    if (c > (1 << PREEMPH_SHIFT) - 1) c = (1 << PREEMPH_SHIFT) - 1;
    return c;
}

void oracle_preemphasis(int32_t *x, uint32_t n, int32_t prev, int32_t coef)
{
    /* srla_utility.c:342-358: y[i] = x[i] - ((x[i-1] * coef) >> 4), x[-1] = prev */
    uint32_t i;
    for (i = 0; i < n; i++) {
        const int32_t cur = x[i];
        x[i] = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)coef) >> PREEMPH_SHIFT));
        prev = cur;
    }
}

/* =========================================================================================
 * FFT (libs/fft/src/fft.c).  Radix-4 Stockham on n complex points with a trailing radix-2
 * stage when log2(n) is odd; twiddles advance by a multiplicative recurrence, so their
 * values depend on the order of multiplication -- reproduced exactly here.
 * ========================================================================================= */
typedef struct { double re, im; } cplx;
static cplx c_add(cplx a, cplx b) { cplx r; r.re = a.re + b.re; r.im = a.im + b.im; return r; }
static cplx c_sub(cplx a, cplx b) { cplx r; r.re = a.re - b.re; r.im = a.im - b.im; return r; }
static cplx c_mul(cplx a, cplx b)
{
    /* fft.c:57-63 */
    cplx r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}

static void fft_complex(int n, int flag, cplx *x, cplx *y)
{
    /* fft.c:71-136 */
    const double pi = 3.14159265358979323846;
    cplx *origin = x;
    int stride = 1;
    while (n > 2) {
        const int quarter = n >> 2, half = n >> 1, three_q = quarter + half;
        const double theta = 2.0 * pi / n;
        cplx rot, step, w1;
        int p, q;
        rot.re = 0.0; rot.im = -flag;
        step.re = cos(theta); step.im = flag * sin(theta);
        w1.re = 1.0; w1.im = 0.0;
        for (p = 0; p < quarter; p++) {
            const cplx w2 = c_mul(w1, w1);
            const cplx w3 = c_mul(w1, w2);
            for (q = 0; q < stride; q++) {
                const cplx a = x[q + stride * p];
                const cplx b = x[q + stride * (p + quarter)];
                const cplx c = x[q + stride * (p + half)];
                const cplx d = x[q + stride * (p + three_q)];
                const cplx apc = c_add(a, c), amc = c_sub(a, c), bpd = c_add(b, d);
                const cplx jbmd = c_mul(rot, c_sub(b, d));
                y[q + stride * (4 * p + 0)] = c_add(apc, bpd);
                y[q + stride * (4 * p + 1)] = c_mul(w1, c_sub(amc, jbmd));
                y[q + stride * (4 * p + 2)] = c_mul(w2, c_sub(apc, bpd));
                y[q + stride * (4 * p + 3)] = c_mul(w3, c_add(amc, jbmd));
            }
            w1 = c_mul(w1, step);
        }
        n >>= 2;
        stride <<= 2;
        { cplx *t = x; x = y; y = t; }
    }
    if (n == 2) {
        int q;
        for (q = 0; q < stride; q++) {
            const cplx a = x[q], b = x[q + stride];
            y[q] = c_add(a, b);
            y[q + stride] = c_sub(a, b);
        }
        stride <<= 1;
        { cplx *t = x; x = y; y = t; }
    }
    if (origin != x) memcpy(y, x, sizeof(cplx) * (size_t)stride);
}

void oracle_fft_real(int n, int flag, double *x, double *work)
{
    /* fft.c:147-198: real FFT of n points through an n/2-point complex FFT */
    const double pi = 3.14159265358979323846;
    const double theta = flag * 2.0 * pi / n;
    const double wpi = sin(theta);
    const double wpr = cos(theta) - 1.0;
    const double c2 = flag * 0.5;
    double wr, wi;
    int i;
    if (flag == -1) fft_complex(n >> 1, -1, (cplx *)x, (cplx *)work);
    wr = 1.0 + wpr;
    wi = wpi;
    for (i = 1; i <= (n >> 2); i++) {
        const int i1 = 2 * i, i2 = i1 + 1, i3 = n - i1, i4 = i3 + 1;
        const double h1r = 0.5 * (x[i1] + x[i3]);
        const double h1i = 0.5 * (x[i2] - x[i4]);
        const double h2r = -c2 * (x[i2] + x[i4]);
        const double h2i = c2 * (x[i1] - x[i3]);
        double wtmp;
        x[i1] = h1r + (wr * h2r) - (wi * h2i);
        x[i2] = h1i + (wr * h2i) + (wi * h2r);
        x[i3] = h1r - (wr * h2r) + (wi * h2i);
        x[i4] = -h1i + (wr * h2i) + (wi * h2r);
        wtmp = wr;
        wr += wtmp * wpr - wi * wpi;
        wi += wi * wpr + wtmp * wpi;
    }
    {
        const double h1r = x[0];
        if (flag == -1) {
            x[0] = h1r + x[1];
            x[1] = h1r - x[1];
        } else {
            x[0] = 0.5 * (h1r + x[1]);
            x[1] = 0.5 * (h1r - x[1]);
            fft_complex(n >> 1, 1, (cplx *)x, (cplx *)work);
        }
    }
}

/* =========================================================================================
 * LPC analysis (libs/lpc/src/lpc.c)
 * ========================================================================================= */
void oracle_welch_window(const double *in, uint32_t n, double *out)
{
    /* lpc.c:256-266: symmetric pairs; the middle sample of an odd-length block is NOT
     * written (so `out` keeps whatever it held). */
    const double divisor = 4.0 * pow(n - 1, -2.0);
    uint32_t i;
    for (i = 0; i < (n >> 1); i++) {
        const double w = divisor * i * (n - 1 - i);
        out[i] = in[i] * w;
        out[n - i - 1] = in[n - i - 1] * w;
    }
}

static void autocorr_from_windowed(double *buf, double *work, uint32_t n, double *lags, uint32_t num_lags)
{
    /* lpc.c:330-376: zero pad to the next power of two >= n (NOT 2n: circular), |X|^2,
     * inverse, scale by 2/n. */
    const uint32_t size = next_pow2(n);
    const double norm = 2.0 / n;
    uint32_t i;
    for (i = n; i < size; i++) buf[i] = 0.0;
    oracle_fft_real((int)size, -1, buf, work);
    buf[0] *= buf[0];
    buf[1] *= buf[1];
    for (i = 2; i < size; i += 2) {
        const double re = buf[i], im = buf[i + 1];
        buf[i] = re * re + im * im;
        buf[i + 1] = 0.0;
    }
    oracle_fft_real((int)size, 1, buf, work);
    for (i = 0; i < num_lags; i++) lags[i] = buf[i] * norm;
}

void oracle_autocorr(struct Oracle *o, const double *signal, uint32_t n, double *lags, uint32_t num_lags)
{
    oracle_welch_window(signal, n, o->fftbuf);
    autocorr_from_windowed(o->fftbuf, o->fftwork, n, lags, num_lags);
}

static double welch_inverse_square_sum(uint32_t num_samples)
{
    /* lpc.c:275-290 */
    const double n = num_samples - 1;
    return (15 * (n - 1) * (n - 1) * (n - 1)) / (8 * n * (n - 2) * (n * n - 2 * n + 2));
}

void oracle_levinson(const double *r, uint32_t order, uint32_t num_samples, double *coefs, double *error_vars)
{
    /* lpc.c:379-441 (recursion) + lpc.c:490-497 (window compensation) +
     * lpc.c:563-567 (rows copied without the leading 1).  `r` already carries the ridge
     * factor on r[0] (lpc.c:483).  coefs[k*order + i] = a_{k+1}[i+1], i < order. */
    const uint32_t stride = ORACLE_MAX_ORDER + 2;
    double *a = (double *)calloc((size_t)(order + 1) * stride, sizeof(double));
    uint32_t k, i;
    if (fabs(r[0]) < FLT_EPSILON) {
        for (i = 0; i < order + 1; i++) error_vars[i] = r[0];
        /* predictor rows are all zero (already calloc'ed) */
    } else {
        a[0] = 1.0;
        error_vars[0] = r[0];
        a[1] = -r[1] / r[0];
        a[2] = 0.0;
        error_vars[1] = error_vars[0] + r[1] * a[1];
        for (k = 1; k < order; k++) {
            const double *prev = &a[(size_t)(k - 1) * stride];
            double *cur = &a[(size_t)k * stride];
            double gamma = 0.0;
            for (i = 0; i < k + 1; i++) gamma += prev[i] * r[k + 1 - i];
            gamma /= -error_vars[k];
            error_vars[k + 1] = error_vars[k] * (1.0 - gamma * gamma);
            for (i = 0; i < k + 2; i++) cur[i] = prev[i] + gamma * prev[k + 1 - i];
            cur[k + 2] = 0.0;
        }
    }
    {
        const double comp = welch_inverse_square_sum(num_samples);
        for (i = 0; i < order + 1; i++) error_vars[i] *= comp;
    }
    for (k = 0; k < order; k++)
        for (i = 0; i < order; i++) coefs[(size_t)k * order + i] = a[(size_t)k * stride + 1 + i];
    free(a);
}

static double geometric_entropy(double mean_abs, uint32_t bps)
{
    /* srla_encoder.c:873-885 */
    const double intmean = mean_abs * (1 << (bps - 1));
    const double rho = 1.0 / (1.0 + intmean);
    const double invrho = 1.0 - rho;
    if (mean_abs < 1e-16) return 0.0;
    return -(invrho * log2_via_ln(invrho) + rho * log2_via_ln(rho)) / rho;
}

uint32_t oracle_select_order(const double *error_vars, uint32_t max_order, uint32_t num_samples,
                             uint32_t bits_per_sample, double *lens_out)
{
    /* srla_encoder.c:934-957 (BRUTEFORCE_ESTIMATION): first strict minimum of
     * entropy(2*sqrt(var/2)) * N + 8 * order over order = 1..max. */
    double best = FLT_MAX;
    uint32_t order, best_order = 0;
    for (order = 1; order <= max_order; order++) {
        const double mabse = 2.0 * sqrt(error_vars[order] / 2.0);
        double len = geometric_entropy(mabse, bits_per_sample) * num_samples;
        len += LPC_COEF_BITS * order;
        if (lens_out) lens_out[order] = len;
        if (best > len) { best = len; best_order = order; }
    }
    return best_order;
}

void oracle_quantize(const double *coef, uint32_t order, int32_t *icoef, uint32_t *rshift_out)
{
    /* lpc.c:1341-1405 with nbits_precision = 8, max_bits = 16 */
    const int32_t qmax = 1 << (LPC_COEF_BITS - 1);
    double maxabs = 0.0, qerr = 0.0;
    int32_t i, ndigit;
    uint32_t rshift;
    for (i = 0; i < (int32_t)order; i++)
        if (maxabs < fabs(coef[i])) maxabs = fabs(coef[i]);
    if (maxabs <= pow(2.0, -(int32_t)(LPC_COEF_BITS - 1))) {
        *rshift_out = LPC_COEF_BITS;
        memset(icoef, 0, sizeof(int32_t) * order);
        return;
    }
    (void)frexp(maxabs, &ndigit);
    rshift = (uint32_t)((int32_t)(LPC_COEF_BITS - 1) - ndigit);
    if (rshift >= (1u << LPC_RSHIFT_BITS)) rshift = (1u << LPC_RSHIFT_BITS) - 1;
    for (i = (int32_t)order - 1; i >= 0; i--) {
        int32_t q;
        qerr += coef[i] * pow(2.0, rshift);
        q = (int32_t)round_half_away(qerr);
        if (q >= qmax) q = qmax - 1;
        else if (q < -qmax) q = -qmax;
        qerr -= q;
        icoef[i] = q;
    }
    *rshift_out = rshift;
}

static int32_t rounding_half(uint32_t rshift)
{
    /* `1 << (rshift - 1)`; rshift == 0 is undefined in C and evaluates to 1 << 31 on x86
     * (shift count masked to 5 bits), which is what the reference binary does. */
    return (int32_t)(1u << ((rshift - 1u) & 31u));
}

void oracle_lpc_predict(const int32_t *data, uint32_t n, const int32_t *coef, uint32_t order,
                        int32_t *residual, uint32_t rshift)
{
    /* srla_lpc_predict.c:236-265 (== the SSE4.1/AVX2 variants): int32 wrap-around */
    const int32_t half = rounding_half(rshift);
    uint32_t s, k;
    memcpy(residual, data, sizeof(int32_t) * n);
    for (s = 1; s < order; s++) residual[s] = (int32_t)((uint32_t)data[s] - (uint32_t)data[s - 1]);
    for (s = order; s < n; s++) {
        uint32_t acc = (uint32_t)half;
        for (k = 0; k < order; k++) acc += (uint32_t)coef[k] * (uint32_t)data[s - order + k];
        residual[s] = (int32_t)((uint32_t)residual[s] + (uint32_t)((int32_t)acc >> rshift));
    }
}

void oracle_ltp_predict(const int32_t *data, uint32_t n, const int32_t *coef, uint32_t order,
                        uint32_t period, int32_t *residual, uint32_t rshift)
{
    /* srla_lpc_predict.c:267-294 */
    const int32_t half = rounding_half(rshift);
    const uint32_t half_order = order >> 1;
    uint32_t s, k;
    memcpy(residual, data, sizeof(int32_t) * n);
    for (s = period + half_order + 1; s < n; s++) {
        uint32_t acc = (uint32_t)half;
        for (k = 0; k < order; k++) acc += (uint32_t)coef[k] * (uint32_t)data[s - period - half_order + k];
        residual[s] = (int32_t)((uint32_t)residual[s] - (uint32_t)((int32_t)acc >> rshift));
    }
}

int oracle_detect_pitch(const double *r, uint32_t min_period, uint32_t max_period, uint32_t *period)
{
    /* lpc.c:1473-1555: up to 20 zero-crossing-delimited lobes, each contributing its largest
     * local maximum; the pitch is the first candidate >= 0.9 * best, provided
     * best >= 0.1 * r[0].  Reads r[max_period + 2] at most. */
    uint32_t cand[20];
    uint32_t ncand = 0, i = min_period, k;
    double best = 0.0;
    while (i < max_period && ncand < 20) {
        uint32_t start, end, j, peak_at = 0;
        double peak = 0.0;
        for (start = i; start < max_period; start++)
            if (r[start - 1] < 0.0 && r[start] > 0.0) break;
        for (end = start + 1; end < max_period - 1; end++)
            if (r[end] > 0.0 && r[end + 1] < 0.0) break;
        for (j = start; j <= end; j++) {
            if (r[j] > r[j - 1] && r[j] > r[j + 1] && r[j] > peak) { peak_at = j; peak = r[j]; }
        }
        if (peak_at != 0) {
            cand[ncand++] = peak_at;
            if (peak > best) best = peak;
        }
        i = end + 1;
    }
    if (ncand == 0) return -1;
    if (best < 0.1 * r[0]) return -1;
    for (k = 0; k < ncand; k++) {
        if (r[cand[k]] >= 0.9 * best) { *period = cand[k]; return 0; }
    }
    return -1;
}

/* returns 0 ok, 1 no pitch, -1 numerical failure (encoder aborts with NG) */
int oracle_ltp_coefficients(struct Oracle *o, const double *signal, uint32_t n, uint32_t order,
                            double *coef, uint32_t *period_out)
{
    /* lpc.c:1558-1649 */
    double *r = o->acorr;
    double amat[ORACLE_LTP_TAPS][ORACLE_LTP_TAPS], inv_diag[ORACLE_LTP_TAPS], x[ORACLE_LTP_TAPS];
    const double *b;
    uint32_t period = 0;
    int32_t i, j, k, dim = (int32_t)order;
    oracle_welch_window(signal, n, o->fftbuf);
    autocorr_from_windowed(o->fftbuf, o->fftwork, n, r, LTP_MAX_PERIOD + 1);
    if (fabs(r[0]) <= FLT_MIN) return 1;
    if (oracle_detect_pitch(r, LTP_MIN_PERIOD, LTP_MAX_PERIOD, &period) != 0) return 1;
    if (period < (order / 2) + 1) return 1;
    r[0] *= (1.0 + RIDGE);
    for (j = 0; j < dim; j++)
        for (k = j; k < dim; k++) amat[j][k] = amat[k][j] = r[k - j];
    /* Cholesky, lpc.c:573-602 (inverse diagonal through pow(x, -0.5)) */
    for (i = 0; i < dim; i++) {
        double sum = amat[i][i];
        for (k = i - 1; k >= 0; k--) sum -= amat[i][k] * amat[i][k];
        if (sum <= 0.0) return -1;
        inv_diag[i] = pow(sum, -0.5);
        for (j = i + 1; j < dim; j++) {
            sum = amat[i][j];
            for (k = i - 1; k >= 0; k--) sum -= amat[i][k] * amat[j][k];
            amat[j][i] = sum * inv_diag[i];
        }
    }
    /* forward/backward substitution, lpc.c:605-631; right-hand side centred on the period */
    b = &r[period - order / 2];
    for (i = 0; i < dim; i++) {
        double sum = b[i];
        for (j = i - 1; j >= 0; j--) sum -= amat[i][j] * x[j];
        x[i] = sum * inv_diag[i];
    }
    for (i = dim - 1; i >= 0; i--) {
        double sum = x[i];
        for (j = i + 1; j < dim; j++) sum -= amat[j][i] * x[j];
        x[i] = sum * inv_diag[i];
    }
    for (i = 0; i < dim; i++) coef[i] = x[i];
    *period_out = period;
    return 0;
}

/* =========================================================================================
 * Residual coder: parameter choice and code-length search (libs/srla_coder/src/srla_coder.c)
 * ========================================================================================= */
uint32_t oracle_rice_k(double mean)
{
    /* srla_coder.c:262-276 */
    const double optx = 0.5127629514437670454896078808815218508243560791015625;
    const double rho = 1.0 / (1.0 + mean);
    const double v = round_half_away(log2_via_ln(log(optx) / log(1.0 - rho)));
    return (uint32_t)((0 > v) ? 0 : v);
}

uint32_t oracle_recursive_rice_k2(double mean)
{
    /* srla_coder.c:298-311: k2 = floor(log2((uint32)max(1, 0.66794162356 * (1 + mean)))) */
    const double g = 0.66794162356 * (1.0 + mean);
    const uint32_t golomb = (uint32_t)((1 > g) ? 1 : g);
    uint32_t k2 = 0;
    while ((golomb >> (k2 + 1)) != 0) k2++;
    return k2;
}

typedef struct {
    uint32_t code_type, porder, bits, max_porder;
    double *mean[PORDER_BITS + 1]; /* per-level partition means */
    uint32_t *uval;
} CodeSearch;

static void code_search(const int32_t *data, uint32_t n, CodeSearch *cs)
{
    /* srla_coder.c:349-484 */
    uint32_t max_porder = 1, nparts, part, porder, i, max_u = 0;
    uint32_t best_bits = UINT32_MAX, best_porder;
    int32_t lvl;
    while ((n % (1u << max_porder)) == 0) max_porder++;
    max_porder = (max_porder - 1 < PORDER_BITS) ? (max_porder - 1) : PORDER_BITS;
    nparts = 1u << max_porder;
    cs->max_porder = max_porder;
    for (part = 0; part < nparts; part++) {
        const uint32_t len = n / nparts;
        double sum = 0.0;
        for (i = 0; i < len; i++) {
            const uint32_t u = zigzag(data[part * len + i]);
            cs->uval[part * len + i] = u;
            sum += u;
            if (u > max_u) max_u = u;
        }
        cs->mean[max_porder][part] = sum / len;
    }
    for (lvl = (int32_t)max_porder - 1; lvl >= 0; lvl--)
        for (part = 0; part < (1u << lvl); part++)
            cs->mean[lvl][part] = (cs->mean[lvl + 1][2 * part] + cs->mean[lvl + 1][2 * part + 1]) / 2.0;

    if (max_u == 0) cs->code_type = CODE_ALLZERO;
    else if (cs->mean[0][0] < 2) cs->code_type = CODE_RICE;
    else cs->code_type = CODE_RECURSIVE_RICE;

    best_porder = max_porder + 1;
    if (cs->code_type == CODE_ALLZERO) {
        best_porder = 0;
        best_bits = 0;
    } else {
        for (porder = 0; porder <= max_porder; porder++) {
            const uint32_t len = n >> porder;
            uint32_t bits = PORDER_BITS, prev = 0;
            for (part = 0; part < (1u << porder); part++) {
                uint32_t param;
                const uint32_t *u = &cs->uval[part * len];
                if (cs->code_type == CODE_RICE) {
                    param = oracle_rice_k(cs->mean[porder][part]);
                    for (i = 0; i < len; i++) bits += 1 + param + (u[i] >> param); /* srla_coder.c:327-330 */
                } else {
                    const uint32_t k2 = oracle_recursive_rice_k2(cs->mean[porder][part]);
                    const uint32_t k1 = k2 + 1, k1pow = 1u << k1;
                    param = k2;
                    bits += (k1 + 1) * len; /* srla_coder.c:333-347 */
                    for (i = 0; i < len; i++) {
                        const int32_t over = (int32_t)u[i] - (int32_t)k1pow;
                        bits += (uint32_t)(((over > 0) ? over : 0) >> k2);
                    }
                }
                if (part == 0) bits += RICE_PARAM_BITS;
                else bits += zigzag((int32_t)param - (int32_t)prev) + 1;
                prev = param;
                if (bits >= best_bits) break;
            }
            if (bits < best_bits) { best_bits = bits; best_porder = porder; }
        }
    }
    cs->porder = best_porder;
    cs->bits = best_bits + 2;
}

static void code_search_alloc(CodeSearch *cs, uint32_t n)
{
    int i;
    for (i = 0; i <= PORDER_BITS; i++) cs->mean[i] = (double *)malloc(sizeof(double) << PORDER_BITS);
    cs->uval = (uint32_t *)malloc(sizeof(uint32_t) * n);
}
static void code_search_free(CodeSearch *cs)
{
    int i;
    for (i = 0; i <= PORDER_BITS; i++) free(cs->mean[i]);
    free(cs->uval);
}

void oracle_residual_code_search(const int32_t *data, uint32_t n, uint32_t *code_type,
                                 uint32_t *porder, uint32_t *bits)
{
    CodeSearch cs;
    code_search_alloc(&cs, n);
    code_search(data, n, &cs);
    *code_type = cs.code_type; *porder = cs.porder; *bits = cs.bits;
    code_search_free(&cs);
}

static void put_recursive_rice(BitW *w, uint32_t k2, uint32_t u)
{
    /* srla_coder.c:175-190 */
    const uint32_t k1 = k2 + 1, k1pow = 1u << k1;
    if (u < k1pow) {
        bw_put(w, k1pow | u, k1 + 1);
    } else {
        const uint32_t v = u - k1pow;
        bw_zero_run(w, 1 + (v >> k2));
        bw_put(w, v, k2);
    }
}

static void encode_residual(BitW *w, const int32_t *data, uint32_t n)
{
    /* srla_coder.c:532-595 */
    CodeSearch cs;
    uint32_t part, i, prev = 0;
    code_search_alloc(&cs, n);
    code_search(data, n, &cs);
    bw_put(w, cs.code_type, 2);
    if (cs.code_type != CODE_ALLZERO) {
        const uint32_t len = n >> cs.porder;
        bw_put(w, cs.porder, PORDER_BITS);
        for (part = 0; part < (1u << cs.porder); part++) {
            const double mean = cs.mean[cs.porder][part];
            const uint32_t param = (cs.code_type == CODE_RICE) ? oracle_rice_k(mean) : oracle_recursive_rice_k2(mean);
            if (part == 0) bw_put(w, param, RICE_PARAM_BITS);
            else bw_zero_run(w, zigzag((int32_t)param - (int32_t)prev));
            prev = param;
            for (i = 0; i < len; i++) {
                const uint32_t u = cs.uval[part * len + i];
                if (cs.code_type == CODE_RICE) {
                    bw_zero_run(w, u >> param); /* srla_coder.c:165-171 */
                    bw_put(w, u, param);
                } else {
                    put_recursive_rice(w, param, u);
                }
            }
        }
    }
    code_search_free(&cs);
}

static void decode_residual(BitR *r, int32_t *data, uint32_t n)
{
    /* srla_coder.c:650-698 */
    const uint32_t type = br_get(r, 2);
    uint32_t porder, len, part, i, param = 0;
    if (type == CODE_ALLZERO) { memset(data, 0, sizeof(int32_t) * n); return; }
    porder = br_get(r, PORDER_BITS);
    len = n >> porder;
    for (part = 0; part < (1u << porder); part++) {
        if (part == 0) param = br_get(r, RICE_PARAM_BITS);
        else param = (uint32_t)((int32_t)param + unzigzag(br_zero_run(r)));
        for (i = 0; i < len; i++) {
            uint32_t u;
            if (type == CODE_RICE) {
                const uint32_t quot = br_zero_run(r);
                u = (quot << param) + br_get(r, param);
            } else {
                /* srla_coder.c:223-241 */
                const uint32_t quot = br_zero_run(r);
                u = br_get(r, param + (quot == 0));
                u |= (quot + (quot != 0)) << param;
            }
            data[part * len + i] = unzigzag(u);
        }
    }
}

uint32_t oracle_coef_bits(const int32_t *coef, uint32_t order, uint32_t *use_sum)
{
    /* srla_encoder.c:1141-1174: plain Huffman cost vs pair-summed cost with early outs */
    uint32_t plain = 0, summed, p, use = 1;
    if (order == 0) { *use_sum = 0; return 0; }
    for (p = 0; p < order; p++) plain += srla_huff_plain_len[zigzag(coef[p])];
    summed = srla_huff_plain_len[zigzag(coef[0])];
    for (p = 1; p < order; p++) {
        const uint32_t u = zigzag(coef[p] + coef[p - 1]);
        if (u >= 256) { use = 0; break; }
        summed += srla_huff_summed_len[u];
        if (summed >= plain) { use = 0; break; }
    }
    *use_sum = use;
    return use ? summed : plain;
}

/* =========================================================================================
 * Encoder (libs/srla_encoder/src/srla_encoder.c)
 * ========================================================================================= */
struct Oracle *oracle_create(const OracleConfig *cfg)
{
    struct Oracle *o;
    uint32_t i;
    if (!cfg || cfg->num_channels == 0 || cfg->num_channels > ORACLE_MAX_CHANNELS || cfg->max_block == 0
        || cfg->min_block == 0 || cfg->min_block > cfg->max_block || cfg->lookahead < cfg->max_block
        || (cfg->lookahead % cfg->min_block) != 0 || cfg->preset > 6
        || (cfg->ltp_order > 0 && (cfg->ltp_order % 2) == 0) || cfg->ltp_order > ORACLE_LTP_TAPS)
        return NULL;
    o = (struct Oracle *)calloc(1, sizeof(*o));
    o->cfg = *cfg;
    o->fft_cap = next_pow2(cfg->max_block);
    o->fftbuf = (double *)calloc(o->fft_cap, sizeof(double));
    o->fftwork = (double *)calloc(o->fft_cap, sizeof(double));
    o->acorr = (double *)calloc((cfg->max_block > 512 ? cfg->max_block : 512), sizeof(double));
    o->dsig = (double *)calloc(cfg->max_block, sizeof(double));
    o->coefs = (double *)calloc((size_t)ORACLE_MAX_ORDER * ORACLE_MAX_ORDER, sizeof(double));
    for (i = 0; i < 4; i++) {
        o->work_int[i] = (int32_t *)calloc(cfg->max_block, sizeof(int32_t));
        o->work_res[i] = (int32_t *)calloc(cfg->max_block, sizeof(int32_t));
    }
    o->extra_int = (int32_t *)calloc(cfg->max_block, sizeof(int32_t));
    for (i = 0; i < ORACLE_MAX_CHANNELS; i++) o->extra_res[i] = (int32_t *)calloc(cfg->max_block, sizeof(int32_t));
    return o;
}

void oracle_destroy(struct Oracle *o)
{
    uint32_t i;
    if (!o) return;
    free(o->fftbuf); free(o->fftwork); free(o->acorr); free(o->dsig); free(o->coefs);
    for (i = 0; i < 4; i++) { free(o->work_int[i]); free(o->work_res[i]); }
    free(o->extra_int);
    for (i = 0; i < ORACLE_MAX_CHANNELS; i++) free(o->extra_res[i]);
    free(o);
}

int oracle_set_parameter(struct Oracle *o, const OracleConfig *cfg)
{
    if (!o || !cfg || cfg->num_channels != o->cfg.num_channels || cfg->max_block != o->cfg.max_block
        || cfg->min_block == 0 || cfg->min_block > cfg->max_block || cfg->lookahead < cfg->max_block
        || (cfg->lookahead % cfg->min_block) != 0 || cfg->preset > 6
        || (cfg->ltp_order > 0 && (cfg->ltp_order % 2) == 0) || cfg->ltp_order > ORACLE_LTP_TAPS)
        return -1;
    o->cfg = *cfg;
    o->offset_lshift = 0;          /* srla_encoder.c:743-750: a new header */
    return 0;
}

void oracle_set_offset_lshift(struct Oracle *o, uint32_t lshift) { o->offset_lshift = lshift; }
void oracle_set_svr_iterations(struct Oracle *o, uint32_t iterations) { o->svr_iterations = iterations; }

/* ---- SVR refinement of the predictor (--svr-filter-learning-iteration), libs/lpc/src/lpc.c:987-1136 -------------------- */
/* lpc.c:1023-1033 (BITS_PER_SAMPLE is the constant 16 there, :1042) */
static double svr_rgr_mean_code_length(double mean_abs_error, uint32_t bps)
{
    const double intmean = mean_abs_error * (1 << bps);
    const double rho = 1.0 / (1.0 + intmean);
    const double l2 = log(log(0.5127629514) / log(1.0 - rho)) * 1.4426950408889634;      /* LPC_Log2, lpc.c:71-76 */
    const uint32_t k2 = (uint32_t)((0 > l2) ? 0 : l2);
    const uint32_t k1 = k2 + 1;
    const double k1factor = pow(1.0 - rho, (double)(1 << k1));
    const double k2factor = pow(1.0 - rho, (double)(1 << k2));
    return (1.0 + k1) * (1.0 - k1factor) + (1.0 + k2 + (1.0 / (1.0 - k2factor))) * k1factor;
}

/* The reference keeps residual / best_coef / init_coef in the calculator's persistent buffer, work_buffer and auto_corr
 * (lpc.c:1044-1050): what they hold afterwards is part of the history an odd-length block inherits, so the same three
 * arrays are used here (NULL: private scratch, for the stage-level tests). */
int oracle_svr_refine_in(const double *data, uint32_t num_samples, double *coef, uint32_t order, uint32_t max_iter,
                         double *persistent_buffer, double *persistent_work, double *persistent_acorr);

int oracle_svr_refine(const double *data, uint32_t num_samples, double *coef, uint32_t order, uint32_t max_iter)
{
    return oracle_svr_refine_in(data, num_samples, coef, order, max_iter, NULL, NULL, NULL);
}

int oracle_svr_refine_in(const double *data, uint32_t num_samples, double *coef, uint32_t order, uint32_t max_iter,
                         double *persistent_buffer, double *persistent_work, double *persistent_acorr)
{
    static const double margin_list[] = { 0.0, 1.0 / 4096, 1.0 / 1024, 1.0 / 256, 1.0 / 64, 1.0 / 16 };   /* srla_internal.c:27 */
    const uint32_t p = order;
    double *cov, *low, *r_vec, *delta, *init_coef, *best_coef, *residual;
    double min_obj, prev_obj, obj;
    uint32_t i, j, smpl, itr, m;
    int k;
    if (max_iter == 0) return 0;                                                             /* lpc.c:1058-1061 */
    cov = (double *)calloc((size_t)p * p + 5 * (size_t)p + num_samples, sizeof(double));
    low = cov + (size_t)p * p; r_vec = low + p; delta = r_vec + p; init_coef = delta + p; best_coef = init_coef + p;
    residual = best_coef + p;
    if (persistent_buffer) residual = persistent_buffer;
    if (persistent_work) best_coef = persistent_work;
    if (persistent_acorr) init_coef = persistent_acorr;
#define COV(a, b) cov[(size_t)(a) * p + (b)]
    /* covariance, lpc.c:987-1020 */
    for (smpl = 0; smpl < num_samples - p; smpl++) {
        const double *pd = &data[smpl];
        for (i = 0; i < p; i++) {
            const double sv = pd[i];
            for (j = i; j < p; j++) COV(i, j) += sv * pd[j];
        }
    }
    for (i = 0; i < p; i++) for (j = i + 1; j < p; j++) COV(j, i) = COV(i, j);
    for (i = 0; i < p; i++) COV(i, i) *= (1.0 + RIDGE);                                      /* lpc.c:1067-1069 */
    /* Cholesky, lpc.c:573-600 */
    for (i = 0; i < p; i++) {
        double sum = COV(i, i);
        for (k = (int)i - 1; k >= 0; k--) sum -= COV(i, k) * COV(i, k);
        if (sum <= 0.0) { for (j = 0; j < p; j++) coef[j] = 0.0; free(cov); return 0; }       /* lpc.c:1071-1077 */
        low[i] = pow(sum, -0.5);
        for (j = i + 1; j < p; j++) {
            sum = COV(i, j);
            for (k = (int)i - 1; k >= 0; k--) sum -= COV(i, k) * COV(j, k);
            COV(j, i) = sum * low[i];
        }
    }
    memcpy(init_coef, coef, sizeof(double) * p);
    memcpy(best_coef, init_coef, sizeof(double) * p);
    min_obj = FLT_MAX;
    for (m = 0; m < sizeof(margin_list) / sizeof(margin_list[0]); m++) {
        const double margin = margin_list[m];
        prev_obj = FLT_MAX;
        memcpy(coef, init_coef, sizeof(double) * p);
        for (itr = 0; itr < max_iter; itr++) {
            double mabse = 0.0;
            memcpy(residual, data, sizeof(double) * num_samples);
            for (i = 0; i < p; i++) r_vec[i] = 0.0;
            for (smpl = p; smpl < num_samples; smpl++) {
                double r, a;
                for (i = 0; i < p; i++) residual[smpl] += coef[i] * data[smpl - i - 1];
                r = residual[smpl];
                a = (r > 0) ? r : -r;
                mabse += a;
                r = (double)((r > 0) - (r < 0)) * (((a - margin) > 0.0) ? (a - margin) : 0.0);  /* LPC_SOFT_THRESHOLD, lpc.c:34 */
                residual[smpl] = r;
                for (i = 0; i < p; i++) r_vec[i] += r * data[smpl - i - 1];
            }
            obj = svr_rgr_mean_code_length(mabse / num_samples, 16);
            /* cov delta = r_vec, lpc.c:605-631 */
            for (i = 0; i < p; i++) {
                double sum = r_vec[i];
                for (k = (int)i - 1; k >= 0; k--) sum -= COV(i, k) * delta[k];
                delta[i] = sum * low[i];
            }
            for (k = (int)p - 1; k >= 0; k--) {
                double sum = delta[k];
                for (j = (uint32_t)k + 1; j < p; j++) sum -= COV(j, k) * delta[j];
                delta[k] = sum * low[k];
            }
            if (obj < min_obj) { memcpy(best_coef, coef, sizeof(double) * p); min_obj = obj; }
            if ((prev_obj < obj) || (fabs(prev_obj - obj) < 1e-8)) break;
            for (i = 0; i < p; i++) coef[i] += delta[i];
            prev_obj = obj;
        }
    }
    memcpy(coef, best_coef, sizeof(double) * p);
#undef COV
    free(cov);
    return 0;
}

int oracle_analyze_channel(struct Oracle *o, int32_t *buf, uint32_t n, int32_t *residual, OracleChannelParams *out)
{
    /* srla_encoder.c:966-1205 */
    const uint32_t bps = o->cfg.bits_per_sample;
    const uint32_t max_order = k_preset_order[o->cfg.preset];
    const double norm = pow(2.0, -(int32_t)(bps - 1));
    uint32_t i, order, bits;
    CodeSearch cs;
    memset(out, 0, sizeof(*out));

    /* pre-emphasis: coefficient from the block, state seeded with the block's first sample */
    out->preemph_prev = buf[0];
    out->preemph_coef = oracle_preemphasis_coef(buf, n);
    oracle_preemphasis(buf, n, buf[0], out->preemph_coef);

    /* long-term (pitch) prediction, srla_encoder.c:1010-1057 */
    out->ltp_period = 0;
    if (o->cfg.ltp_order > 0) {
        double tap[ORACLE_LTP_TAPS] = { 0.0, 0.0, 0.0 };
        uint32_t period = 0;
        int rc;
        for (i = 0; i < n; i++) o->dsig[i] = buf[i] * norm;
        rc = oracle_ltp_coefficients(o, o->dsig, n, o->cfg.ltp_order, tap, &period);
        if (rc < 0) return -1;
        if (rc == 0 && period > 0) {
            int32_t q[ORACLE_LTP_TAPS];
            const uint32_t taps = o->cfg.ltp_order;
            for (i = 0; i < taps; i++) {
                int32_t c = (int32_t)round_half_away(tap[i] * pow(2.0, LTP_COEF_BITS - 1));
                if (c < -(1 << (LTP_COEF_BITS - 1))) c = -(1 << (LTP_COEF_BITS - 1));
                if (c > (1 << (LTP_COEF_BITS - 1)) - 1) c = (1 << (LTP_COEF_BITS - 1)) - 1;
                q[i] = c;
            }
            for (i = 0; i < taps / 2; i++) { const int32_t t = q[i]; q[i] = q[taps - 1 - i]; q[taps - 1 - i] = t; }
            oracle_ltp_predict(buf, n, q, taps, period, residual, LTP_COEF_BITS - 1);
            memcpy(buf, residual, sizeof(int32_t) * n);
            for (i = 0; i < taps; i++) out->ltp_coef[i] = q[i];
            out->ltp_period = period;
        }
    }

    /* LPC: windowed FFT autocorrelation -> ridge -> all-order Levinson -> order choice */
    for (i = 0; i < n; i++) o->dsig[i] = buf[i] * norm;
    oracle_welch_window(o->dsig, n, o->fftbuf);
    autocorr_from_windowed(o->fftbuf, o->fftwork, n, o->acorr, max_order + 1);
    /* lpc.c:474-480 (n < order) cannot happen: such blocks are RAW (srla_encoder.c:777) */
    o->acorr[0] *= (1.0 + RIDGE);
    oracle_levinson(o->acorr, max_order, n, o->coefs, o->error_vars);
    if (o->cfg.preset == 0) order = max_order; /* MAX_FIXED, srla_encoder.c:899-902 */
    else order = oracle_select_order(o->error_vars, max_order, n, bps, NULL);

    if (order > 0) {
        int32_t q[ORACLE_MAX_ORDER];
        uint32_t rshift;
        /* srla_encoder.c:1084-1097 (a no-op at the default of 0 iterations) */
        if (o->svr_iterations > 0)
            oracle_svr_refine_in(o->dsig, n, &o->coefs[(size_t)(order - 1) * max_order], order, o->svr_iterations, o->fftbuf, o->fftwork, o->acorr);
        oracle_quantize(&o->coefs[(size_t)(order - 1) * max_order], order, q, &rshift);
        for (i = 0; i < order / 2; i++) { const int32_t t = q[i]; q[i] = q[order - 1 - i]; q[order - 1 - i] = t; }
        oracle_lpc_predict(buf, n, q, order, residual, rshift);
        memcpy(out->lpc_coef, q, sizeof(int32_t) * order);
        out->lpc_rshift = rshift;
    } else {
        memcpy(residual, buf, sizeof(int32_t) * n);
        out->lpc_rshift = 0;
    }
    out->lpc_order = order;

    /* cost, srla_encoder.c:1121-1187 */
    code_search_alloc(&cs, n);
    code_search(residual, n, &cs);
    out->res_code_type = cs.code_type; out->res_porder = cs.porder; out->res_bits = cs.bits;
    code_search_free(&cs);
    bits = out->res_bits;
    bits += bps + 1;
    bits += PREEMPH_SHIFT + 1;
    bits += LPC_ORDER_BITS + LPC_RSHIFT_BITS + 1;
    bits += oracle_coef_bits(out->lpc_coef, order, &out->use_sum);
    bits += 1;
    if (out->ltp_period > 0) bits += LTP_ORDER_BITS + LTP_PERIOD_BITS + o->cfg.ltp_order * LTP_COEF_BITS;
    out->code_length = bits;
    return 0;
}

static uint32_t decide_block_type(const struct Oracle *o, const int32_t *const *input, uint32_t n)
{
    /* srla_encoder.c:766-796 (looks at the samples before the offset shift) */
    uint32_t ch, i;
    if (n <= k_preset_order[o->cfg.preset]) return BLOCK_RAW;
    for (ch = 0; ch < o->cfg.num_channels; ch++)
        for (i = 0; i < n; i++)
            if (input[ch][i] != 0) return BLOCK_COMPRESS;
    return BLOCK_SILENT;
}

/* srla_encoder.c:1208-1334.  chosen[ch] -> params, res[ch] -> residual pointers */
static int compute_coefficients(struct Oracle *o, const int32_t *const *input, uint32_t n,
                                uint32_t *method_out, uint32_t *bits_out,
                                OracleChannelParams *chosen, const int32_t **res,
                                OracleChannelParams *variants)
{
    const uint32_t nch = o->cfg.num_channels, sh = o->offset_lshift;
    OracleChannelParams v[4];
    OracleChannelParams tmp;
    uint32_t ch, i, method = 0, bits = 0;
    memset(v, 0, sizeof(v));
    for (ch = 0; ch < nch && ch < 2; ch++)
        for (i = 0; i < n; i++) o->work_int[ch][i] = input[ch][i] >> sh;
    if (nch >= 2) {
        /* reference order: M, S first, then every plain channel (matters only for the
         * persistent FFT buffer with odd n) */
        memcpy(o->work_int[2], o->work_int[0], sizeof(int32_t) * n);
        memcpy(o->work_int[3], o->work_int[1], sizeof(int32_t) * n);
        oracle_lr_to_ms(o->work_int[2], o->work_int[3], n);
        if (oracle_analyze_channel(o, o->work_int[2], n, o->work_res[2], &v[2]) != 0) return -1;
        if (oracle_analyze_channel(o, o->work_int[3], n, o->work_res[3], &v[3]) != 0) return -1;
    }
    for (ch = 0; ch < nch; ch++) {
        if (ch < 2) {
            if (oracle_analyze_channel(o, o->work_int[ch], n, o->work_res[ch], &v[ch]) != 0) return -1;
            chosen[ch] = v[ch];
            res[ch] = o->work_res[ch];
        } else {
            for (i = 0; i < n; i++) o->extra_int[i] = input[ch][i] >> sh;
            if (oracle_analyze_channel(o, o->extra_int, n, o->extra_res[ch], &tmp) != 0) return -1;
            chosen[ch] = tmp;
            res[ch] = o->extra_res[ch];
        }
    }
    if (nch == 1) {
        method = 0;
        bits = chosen[0].code_length;
    } else {
        uint32_t len[4], best;
        len[0] = v[0].code_length + v[1].code_length;
        len[1] = v[2].code_length + v[3].code_length;
        len[2] = v[0].code_length + v[3].code_length;
        len[3] = v[1].code_length + v[3].code_length;
        best = len[0]; method = 0;
        for (i = 1; i < 4; i++) if (best > len[i]) { best = len[i]; method = i; }
        bits = best;
        /* channels >= 2 are never added to the cost (srla_encoder.c:1287-1301) */
        if (method == 1) { chosen[0] = v[2]; chosen[1] = v[3]; res[0] = o->work_res[2]; res[1] = o->work_res[3]; }
        else if (method == 2) { chosen[1] = v[3]; res[1] = o->work_res[3]; }
        else if (method == 3) { chosen[0] = v[3]; res[0] = o->work_res[3]; }
    }
    bits += 2;
    bits = ((bits + 7) / 8) * 8;
    *method_out = method;
    *bits_out = bits;
    if (variants) memcpy(variants, v, sizeof(v));
    return 0;
}

int oracle_analyze_block(struct Oracle *o, const int32_t *const *input, uint32_t n,
                         OracleBlockInfo *info, OracleChannelParams *params,
                         OracleChannelParams *variants, int32_t *const *residual_out)
{
    const uint32_t nch = o->cfg.num_channels, bps = o->cfg.bits_per_sample;
    OracleChannelParams chosen[ORACLE_MAX_CHANNELS];
    const int32_t *res[ORACLE_MAX_CHANNELS];
    uint32_t type, ch;
    if (n == 0 || n > o->cfg.max_block) return -1;
    memset(info, 0, sizeof(*info));
    type = decide_block_type(o, input, n);
    info->block_bytes = BLOCK_HEADER_BYTES;
    if (type == BLOCK_COMPRESS) {
        if (compute_coefficients(o, input, n, &info->ch_method, &info->payload_bits, chosen, res, variants) != 0) return -1;
        if (params) memcpy(params, chosen, sizeof(OracleChannelParams) * nch);
        if (residual_out)
            for (ch = 0; ch < nch; ch++)
                if (residual_out[ch]) memcpy(residual_out[ch], res[ch], sizeof(int32_t) * n);
        if (info->payload_bits >= bps * n * nch) type = BLOCK_RAW; /* srla_encoder.c:1527-1530 */
        else info->block_bytes += info->payload_bits / 8;
    }
    if (type == BLOCK_RAW) info->block_bytes = BLOCK_HEADER_BYTES + (bps * n * nch) / 8;
    info->block_type = type;
    return 0;
}

int oracle_compute_block_size(struct Oracle *o, const int32_t *const *input, uint32_t n, uint32_t *output_size)
{
    /* srla_encoder.c:1477-1546 */
    OracleBlockInfo info;
    if (oracle_analyze_block(o, input, n, &info, NULL, NULL, NULL) != 0) return -1;
    *output_size = info.block_bytes;
    return 0;
}

static uint32_t write_raw_payload(const struct Oracle *o, const int32_t *const *input, uint32_t n, uint8_t *p)
{
    /* srla_encoder.c:823-852: interleaved, zig-zag mapped, big endian, bps/8 bytes each */
    const uint32_t nch = o->cfg.num_channels, bytes = o->cfg.bits_per_sample / 8;
    uint32_t i, ch, b;
    uint8_t *q = p;
    for (i = 0; i < n; i++)
        for (ch = 0; ch < nch; ch++) {
            const uint32_t u = zigzag(input[ch][i]);
            for (b = 0; b < bytes; b++) *q++ = (uint8_t)(u >> (8 * (bytes - 1 - b)));
        }
    return (uint32_t)(q - p);
}

static uint32_t write_compress_payload(const struct Oracle *o, uint32_t n, uint32_t method,
                                       const OracleChannelParams *pc, const int32_t *const *res, uint8_t *p)
{
    /* srla_encoder.c:1368-1452 */
    const uint32_t nch = o->cfg.num_channels, bps = o->cfg.bits_per_sample;
    BitW w;
    uint32_t ch, i;
    bw_open(&w, p);
    bw_put(&w, method, 2);
    for (ch = 0; ch < nch; ch++) {
        bw_put(&w, zigzag(pc[ch].preemph_prev), bps + 1);
        bw_put(&w, zigzag(pc[ch].preemph_coef), PREEMPH_SHIFT + 1);
    }
    for (ch = 0; ch < nch; ch++) {
        bw_put(&w, pc[ch].lpc_order, LPC_ORDER_BITS);
        bw_put(&w, pc[ch].lpc_rshift, LPC_RSHIFT_BITS);
        bw_put(&w, pc[ch].use_sum, 1);
        if (!pc[ch].use_sum) {
            for (i = 0; i < pc[ch].lpc_order; i++) {
                const uint32_t u = zigzag(pc[ch].lpc_coef[i]);
                bw_put(&w, srla_huff_plain_code[u], srla_huff_plain_len[u]);
            }
        } else {
            uint32_t u = zigzag(pc[ch].lpc_coef[0]);
            bw_put(&w, srla_huff_plain_code[u], srla_huff_plain_len[u]);
            for (i = 1; i < pc[ch].lpc_order; i++) {
                u = zigzag(pc[ch].lpc_coef[i] + pc[ch].lpc_coef[i - 1]);
                bw_put(&w, srla_huff_summed_code[u], srla_huff_summed_len[u]);
            }
        }
    }
    for (ch = 0; ch < nch; ch++) {
        bw_put(&w, pc[ch].ltp_period != 0, 1);
        if (pc[ch].ltp_period > 0) {
            bw_put(&w, (o->cfg.ltp_order - 1) / 2, LTP_ORDER_BITS);
            bw_put(&w, pc[ch].ltp_period - LTP_MIN_PERIOD, LTP_PERIOD_BITS);
            for (i = 0; i < o->cfg.ltp_order; i++) bw_put(&w, zigzag(pc[ch].ltp_coef[i]), LTP_COEF_BITS);
        }
    }
    for (ch = 0; ch < nch; ch++) encode_residual(&w, res[ch], n);
    return bw_flush(&w);
}

int oracle_encode_block(struct Oracle *o, const int32_t *const *input, uint32_t n,
                        uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
    /* srla_encoder.c:1549-1643 */
    const uint32_t nch = o->cfg.num_channels, bps = o->cfg.bits_per_sample;
    OracleChannelParams chosen[ORACLE_MAX_CHANNELS];
    const int32_t *res[ORACLE_MAX_CHANNELS];
    uint32_t type, payload = 0, method = 0, bits = 0;
    if (n == 0 || n > o->cfg.max_block) return -1;
    if (data_size < BLOCK_HEADER_BYTES + (bps * n * nch) / 8 + 64) return -2;
    type = decide_block_type(o, input, n);
    if (type == BLOCK_COMPRESS) {
        if (compute_coefficients(o, input, n, &method, &bits, chosen, res, NULL) != 0) return -1;
        payload = write_compress_payload(o, n, method, chosen, res, data + BLOCK_HEADER_BYTES);
        if (8 * payload >= bps * n * nch) type = BLOCK_RAW; /* srla_encoder.c:1608-1611 */
    }
    if (type == BLOCK_RAW) payload = write_raw_payload(o, input, n, data + BLOCK_HEADER_BYTES);
    if (type == BLOCK_SILENT) payload = 0;
    put_u16be(data, 0xFFFF);
    put_u32be(data + 2, payload + 5);
    data[8] = (uint8_t)type;
    put_u16be(data + 9, n);
    put_u16be(data + 6, oracle_fletcher16(data + 8, payload + 3));
    *output_size = BLOCK_HEADER_BYTES + payload;
    return 0;
}

int oracle_dijkstra(const double *adj, uint32_t n, uint32_t start, uint32_t goal,
                    double *min_cost, uint32_t *path)
{
    /* srla_encoder.c:249-307: dense O(n^2) Dijkstra, lowest index wins ties, the relaxation
     * also visits settled nodes, strict comparisons throughout. */
    double cost[ORACLE_MAX_NODES];
    uint8_t used[ORACLE_MAX_NODES];
    uint32_t i, target = start;
    if (n > ORACLE_MAX_NODES) return -1;
    for (i = 0; i < n; i++) { used[i] = 0; path[i] = ~0u; cost[i] = BIG_WEIGHT; }
    cost[start] = 0.0;
    for (;;) {
        double m = BIG_WEIGHT;
        for (i = 0; i < n; i++)
            if (!used[i] && m > cost[i]) { m = cost[i]; target = i; }
        if (target == goal) break;
        if (used[target]) return -2; /* unreachable goal: the reference would spin forever */
        for (i = 0; i < n; i++) {
            const double via = adj[(size_t)target * n + i] + cost[target];
            if (cost[i] > via) { cost[i] = via; path[i] = target; }
        }
        used[target] = 1;
    }
    if (min_cost) *min_cost = cost[goal];
    return 0;
}

int oracle_search_partitions(struct Oracle *o, const int32_t *const *input, uint32_t n,
                             uint32_t *num_partitions, uint32_t *partitions)
{
    /* srla_encoder.c:310-424 */
    const uint32_t minb = o->cfg.min_block, maxb = o->cfg.max_block, nch = o->cfg.num_channels;
    const uint32_t nodes = ((n + minb - 1) / minb) + 1;
    static double adj[ORACLE_MAX_NODES * ORACLE_MAX_NODES];   /* test infrastructure: one encoder at a time */
    uint32_t path[ORACLE_MAX_NODES];
    uint32_t i, j, ch, count, node;
    if (nodes > ORACLE_MAX_NODES) return -1;
    for (i = 0; i < nodes * nodes; i++) adj[i] = BIG_WEIGHT;
    for (i = 0; i < nodes; i++)
        for (j = i + 1; j < nodes; j++) {
            const int32_t *ptr[ORACLE_MAX_CHANNELS];
            const uint32_t off = i * minb;
            uint32_t len = (j - i) * minb, bytes;
            if (len > maxb) continue;
            if (len > n - off) len = n - off;
            for (ch = 0; ch < nch; ch++) ptr[ch] = &input[ch][off];
            if (oracle_compute_block_size(o, ptr, len, &bytes) != 0) return -1;
            adj[i * nodes + j] = bytes;
        }
    if (oracle_dijkstra(adj, nodes, 0, nodes - 1, NULL, path) != 0) return -1;
    count = 0;
    for (node = nodes - 1; node != 0; node = path[node]) count++;
    node = nodes - 1;
    for (i = 0; i < count; i++) {
        const uint32_t off = path[node] * minb;
        uint32_t len = (node - path[node]) * minb;
        if (len > n - off) len = n - off;
        partitions[count - 1 - i] = len;
        node = path[node];
    }
    *num_partitions = count;
    return 0;
}

static void write_stream_header(const struct Oracle *o, uint32_t num_samples, uint8_t *p)
{
    /* srla_encoder.c:134-161 */
    p[0] = '1'; p[1] = '2'; p[2] = '4'; p[3] = '9';
    put_u32be(p + 4, FMT_VERSION);
    put_u32be(p + 8, CODEC_VERSION);
    put_u16be(p + 12, o->cfg.num_channels);
    put_u32be(p + 14, num_samples);
    put_u32be(p + 18, o->cfg.sampling_rate);
    put_u16be(p + 22, o->cfg.bits_per_sample);
    p[24] = (uint8_t)o->offset_lshift;
    put_u32be(p + 25, o->cfg.max_block);
    p[29] = (uint8_t)o->cfg.preset;
}

int oracle_encode_whole(struct Oracle *o, const int32_t *const *input, uint32_t num_samples,
                        uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
    /* srla_encoder.c:1701-1788 */
    const uint32_t nch = o->cfg.num_channels;
    const int search = (o->cfg.min_block != o->cfg.max_block);
    const uint32_t window = search ? o->cfg.lookahead : o->cfg.max_block;
    uint32_t progress = 0, offset = HEADER_BYTES;
    if (num_samples == 0 || data_size < HEADER_BYTES) return -1;
    o->offset_lshift = oracle_offset_lshift(input, nch, num_samples);
    write_stream_header(o, num_samples, data);
    while (progress < num_samples) {
        const int32_t *ptr[ORACLE_MAX_CHANNELS];
        const uint32_t count = (window < num_samples - progress) ? window : (num_samples - progress);
        uint32_t ch, wrote = 0;
        for (ch = 0; ch < nch; ch++) ptr[ch] = &input[ch][progress];
        if (!search) {
            if (oracle_encode_block(o, ptr, count, data + offset, data_size - offset, &wrote) != 0) return -1;
        } else {
            /* srla_encoder.c:1646-1698: search, then encode every partition again */
            uint32_t parts[ORACLE_MAX_NODES], nparts = 0, k, done = 0;
            if (oracle_search_partitions(o, ptr, count, &nparts, parts) != 0) return -1;
            for (k = 0; k < nparts; k++) {
                const int32_t *bp[ORACLE_MAX_CHANNELS];
                uint32_t sz = 0;
                for (ch = 0; ch < nch; ch++) bp[ch] = &ptr[ch][done];
                if (oracle_encode_block(o, bp, parts[k], data + offset + wrote, data_size - offset - wrote, &sz) != 0) return -1;
                wrote += sz;
                done += parts[k];
            }
        }
        offset += wrote;
        progress += count;
    }
    *output_size = offset;
    return 0;
}

/* =========================================================================================
 * Decoder (libs/srla_decoder/src/srla_decoder.c, srla_lpc_synthesize.c) -- the verifier
 * ========================================================================================= */
int oracle_decode_header(const uint8_t *d, uint32_t size, OracleConfig *cfg, uint32_t *num_samples, uint32_t *lshift)
{
    if (size < HEADER_BYTES || d[0] != '1' || d[1] != '2' || d[2] != '4' || d[3] != '9') return -1;
    if (get_u32be(d + 4) != FMT_VERSION) return -2;
    memset(cfg, 0, sizeof(*cfg));
    cfg->num_channels = get_u16be(d + 12);
    *num_samples = get_u32be(d + 14);
    cfg->sampling_rate = get_u32be(d + 18);
    cfg->bits_per_sample = get_u16be(d + 22);
    *lshift = d[24];
    cfg->max_block = get_u32be(d + 25);
    cfg->preset = d[29];
    return 0;
}

static uint32_t huff_decode(BitR *r, const unsigned int *codes, const unsigned char *lens)
{
    /* prefix code lookup by linear search (tables are tiny; this is a verifier) */
    uint32_t code = 0, len = 0, s;
    for (;;) {
        code = (code << 1) | br_get(r, 1);
        len++;
        for (s = 0; s < 256; s++)
            if (lens[s] == len && codes[s] == code) return s;
        if (len > 32) return 0;
    }
}

static void lpc_synthesize(int32_t *d, uint32_t n, const int32_t *coef, uint32_t order, uint32_t rshift)
{
    /* srla_lpc_synthesize.c:238-266 */
    const int32_t half = rounding_half(rshift);
    uint32_t s, k;
    if (order == 0) return;
    for (s = 1; s < order; s++) d[s] = (int32_t)((uint32_t)d[s] + (uint32_t)d[s - 1]);
    for (s = order; s < n; s++) {
        uint32_t acc = (uint32_t)half;
        for (k = 0; k < order; k++) acc += (uint32_t)coef[k] * (uint32_t)d[s - order + k];
        d[s] = (int32_t)((uint32_t)d[s] - (uint32_t)((int32_t)acc >> rshift));
    }
}

static void ltp_synthesize(int32_t *d, uint32_t n, const int32_t *coef, uint32_t order, uint32_t period, uint32_t rshift)
{
    /* srla_lpc_synthesize.c:269-327 */
    const int32_t half = rounding_half(rshift);
    const uint32_t half_order = order >> 1;
    uint32_t s, k;
    if (order == 0 || period == 0) return;
    for (s = period + half_order + 1; s < n; s++) {
        uint32_t acc = (uint32_t)half;
        for (k = 0; k < order; k++) acc += (uint32_t)coef[k] * (uint32_t)d[s - period - half_order + k];
        d[s] = (int32_t)((uint32_t)d[s] + (uint32_t)((int32_t)acc >> rshift));
    }
}

static void deemphasis(int32_t *d, uint32_t n, int32_t prev, int32_t coef)
{
    /* srla_utility.c:361-380: inverse of the pre-emphasis, first output uses `prev` */
    uint32_t i;
    for (i = 0; i < n; i++) {
        d[i] = (int32_t)((uint32_t)d[i] + (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)coef) >> PREEMPH_SHIFT));
        prev = d[i];
    }
}

static int decode_compress_payload(const OracleConfig *cfg, uint32_t lshift, const uint8_t *p, uint32_t size,
                                   int32_t *const *buf, uint32_t n, uint32_t *consumed)
{
    /* srla_decoder.c:436-600 */
    const uint32_t nch = cfg->num_channels, bps = cfg->bits_per_sample;
    int32_t prev[ORACLE_MAX_CHANNELS], pcoef[ORACLE_MAX_CHANNELS];
    uint32_t order[ORACLE_MAX_CHANNELS], rshift[ORACLE_MAX_CHANNELS];
    uint32_t ltp_order[ORACLE_MAX_CHANNELS], ltp_period[ORACLE_MAX_CHANNELS];
    int32_t ltp_coef[ORACLE_MAX_CHANNELS][8];
    static int32_t coef[ORACLE_MAX_CHANNELS][256];
    BitR r;
    uint32_t method, ch, i;
    br_open(&r, p, size);
    method = br_get(&r, 2);
    for (ch = 0; ch < nch; ch++) {
        prev[ch] = unzigzag(br_get(&r, bps + 1));
        pcoef[ch] = unzigzag(br_get(&r, PREEMPH_SHIFT + 1));
    }
    for (ch = 0; ch < nch; ch++) {
        uint32_t use_sum;
        order[ch] = br_get(&r, LPC_ORDER_BITS);
        rshift[ch] = br_get(&r, LPC_RSHIFT_BITS);
        use_sum = br_get(&r, 1);
        if (!use_sum) {
            for (i = 0; i < order[ch]; i++) coef[ch][i] = unzigzag(huff_decode(&r, srla_huff_plain_code, srla_huff_plain_len));
        } else {
            coef[ch][0] = unzigzag(huff_decode(&r, srla_huff_plain_code, srla_huff_plain_len));
            for (i = 1; i < order[ch]; i++)
                coef[ch][i] = unzigzag(huff_decode(&r, srla_huff_summed_code, srla_huff_summed_len)) - coef[ch][i - 1];
        }
    }
    for (ch = 0; ch < nch; ch++) {
        ltp_order[ch] = 0;
        ltp_period[ch] = 0;
        if (br_get(&r, 1)) {
            ltp_order[ch] = 2 * br_get(&r, LTP_ORDER_BITS) + 1;
            ltp_period[ch] = br_get(&r, LTP_PERIOD_BITS) + LTP_MIN_PERIOD;
            for (i = 0; i < ltp_order[ch]; i++) ltp_coef[ch][i] = unzigzag(br_get(&r, LTP_COEF_BITS));
        }
    }
    for (ch = 0; ch < nch; ch++) decode_residual(&r, buf[ch], n);
    *consumed = br_tell_aligned(&r, p);
    for (ch = 0; ch < nch; ch++) {
        lpc_synthesize(buf[ch], n, coef[ch], order[ch], rshift[ch]);
        ltp_synthesize(buf[ch], n, ltp_coef[ch], ltp_order[ch], ltp_period[ch], LTP_COEF_BITS - 1);
        deemphasis(buf[ch], n, prev[ch], pcoef[ch]);
    }
    /* srla_utility.c:106-174 inverse channel transforms */
    if (method == 1) {
        for (i = 0; i < n; i++) {
            buf[0][i] = (int32_t)((uint32_t)buf[0][i] - (uint32_t)(buf[1][i] >> 1));
            buf[1][i] = (int32_t)((uint32_t)buf[1][i] + (uint32_t)buf[0][i]);
        }
    } else if (method == 2) {
        for (i = 0; i < n; i++) buf[1][i] = (int32_t)((uint32_t)buf[1][i] + (uint32_t)buf[0][i]);
    } else if (method == 3) {
        for (i = 0; i < n; i++) buf[0][i] = (int32_t)((uint32_t)buf[1][i] - (uint32_t)buf[0][i]);
    }
    if (lshift > 0)
        for (ch = 0; ch < nch; ch++)
            for (i = 0; i < n; i++) buf[ch][i] = (int32_t)((uint32_t)buf[ch][i] << lshift);
    return 0;
}

int oracle_decode_whole(const uint8_t *data, uint32_t data_size, int32_t *const *buffer,
                        uint32_t buffer_channels, uint32_t buffer_samples)
{
    /* srla_decoder.c:633-799 */
    OracleConfig cfg;
    uint32_t total, lshift, progress = 0, off = HEADER_BYTES;
    if (oracle_decode_header(data, data_size, &cfg, &total, &lshift) != 0) return -1;
    if (buffer_channels < cfg.num_channels || buffer_samples < total) return -2;
    while (progress < total && off < data_size) {
        const uint8_t *b = data + off;
        int32_t *ptr[ORACLE_MAX_CHANNELS];
        uint32_t size, type, n, ch, i, payload_used = 0;
        if (off + BLOCK_HEADER_BYTES > data_size) return -3;
        if (get_u16be(b) != 0xFFFF) return -4;
        size = get_u32be(b + 2);
        if (size + 6 > data_size - off) return -5;
        if (oracle_fletcher16(b + 8, size - 2) != get_u16be(b + 6)) return -6;
        type = b[8];
        n = get_u16be(b + 9);
        if (n > buffer_samples - progress) return -7;
        for (ch = 0; ch < cfg.num_channels; ch++) ptr[ch] = &buffer[ch][progress];
        if (type == BLOCK_RAW) {
            const uint32_t bytes = cfg.bits_per_sample / 8;
            const uint8_t *q = b + BLOCK_HEADER_BYTES;
            for (i = 0; i < n; i++)
                for (ch = 0; ch < cfg.num_channels; ch++) {
                    uint32_t u = 0, k;
                    for (k = 0; k < bytes; k++) u = (u << 8) | *q++;
                    ptr[ch][i] = unzigzag(u);
                }
            payload_used = (uint32_t)(q - (b + BLOCK_HEADER_BYTES));
        } else if (type == BLOCK_COMPRESS) {
            if (decode_compress_payload(&cfg, lshift, b + BLOCK_HEADER_BYTES, size - 5, ptr, n, &payload_used) != 0) return -8;
        } else if (type == BLOCK_SILENT) {
            for (ch = 0; ch < cfg.num_channels; ch++) memset(ptr[ch], 0, sizeof(int32_t) * n);
        } else {
            return -9;
        }
        if (payload_used != size - 5) return -10;
        off += BLOCK_HEADER_BYTES + payload_used;
        progress += n;
    }
    return (progress == total) ? 0 : -11;
}

int oracle_list_blocks(const uint8_t *data, uint32_t data_size, uint32_t *types,
                       uint32_t *nsamples, uint32_t *nbytes, uint32_t cap, uint32_t *count)
{
    uint32_t off = HEADER_BYTES, k = 0;
    while (off + BLOCK_HEADER_BYTES <= data_size) {
        const uint8_t *b = data + off;
        uint32_t size;
        if (get_u16be(b) != 0xFFFF) return -1;
        size = get_u32be(b + 2);
        if (k < cap) { types[k] = b[8]; nsamples[k] = get_u16be(b + 9); nbytes[k] = size + 6; }
        k++;
        off += size + 6;
    }
    *count = k;
    return (off == data_size) ? 0 : -2;
}
