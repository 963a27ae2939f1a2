/*
 * host_tables.cpp -- data-independent constants computed on the host with the host libm.
 *
 * The reference takes cos/sin/pow/log from the platform libm.  Every such use that
 * does not depend on the audio is evaluated here, once, with the same libm the reference would
 * use on this machine, and shipped to the device as tables:
 *   - FFT twiddles, advanced by the same multiplicative recurrences as libs/fft/src/fft.c:83-107
 *     and :149-183 (the values depend on the multiplication order, so they are reproduced step
 *     by step, not recomputed from angles);
 *   - per block length: Welch window divisor (lpc.c:259), window power compensation (lpc.c:283),
 *     autocorrelation scale (lpc.c:335);
 *   - Rice parameter thresholds: k(mean) of srla_coder.c:262-276 is a monotone step function of
 *     the partition mean; its 32 steps are located by bisection over the doubles.
 * Compile with -ffp-contract=off.
 */
#include "host_tables.h"

#include <math.h>
#include <string.h>

namespace srla {

namespace {
struct C2 { double re, im; };
inline C2 mul(C2 a, C2 b) { return C2{ a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; }
const double kPi = 3.14159265358979323846; /* fft.c:17 */
}  // namespace

uint32_t complex_table_len(uint32_t m)
{
    uint32_t t = 0;
    for (uint32_t n = m; n > 2; n >>= 2) t += 3 * (n >> 2);   /* w^p, w^2p, w^3p per stage */
    return t;
}

uint32_t twiddle_count(uint32_t nfft)
{
    return 2 * complex_table_len(nfft >> 1) + 2 * (nfft >> 2);
}

void build_twiddles(uint32_t nfft, double *out /* 2 doubles per entry */)
{
    const uint32_t m = nfft >> 1;
    C2 *o = reinterpret_cast<C2 *>(out);
    for (int pass = 0; pass < 2; pass++) {
        const int flag = (pass == 0) ? -1 : 1;
        for (uint32_t n = m; n > 2; n >>= 2) {
            const double theta0 = 2.0 * kPi / (int)n;
            C2 wdelta{ cos(theta0), flag * sin(theta0) };
            C2 w1{ 1.0, 0.0 };
            const uint32_t n1 = n >> 2;
            for (uint32_t p = 0; p < n1; p++) {
                const C2 w2 = mul(w1, w1);       /* fft.c:95 */
                const C2 w3 = mul(w1, w2);       /* fft.c:96 */
                o[p] = w1; o[n1 + p] = w2; o[2 * n1 + p] = w3;
                w1 = mul(w1, wdelta);            /* fft.c:107 */
            }
            o += 3 * n1;
        }
    }
    for (int pass = 0; pass < 2; pass++) {
        const int flag = (pass == 0) ? -1 : 1;
        const double theta = flag * 2.0 * kPi / (int)nfft;
        const double wpi = sin(theta);
        const double wpr = cos(theta) - 1.0;
        double wr = 1.0 + wpr, wi = wpi;
        for (uint32_t i = 1; i <= (nfft >> 2); i++) {
            *o++ = C2{ wr, wi };
            const double wtmp = wr;
            wr += wtmp * wpr - wi * wpi;
            wi += wi * wpr + wtmp * wpi;
        }
    }
}

static uint32_t next_pow2(uint32_t v)
{
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}

void fill_geom(uint32_t n, SrlaGeom *g)
{
    memset(g, 0, sizeof(*g));
    g->n = n;
    g->nfft = next_pow2(n);
    if (g->nfft < 2) g->nfft = 2;
    uint32_t l = 0;
    while ((1u << l) < g->nfft) l++;
    g->log2_nfft = l;
    uint32_t mp = 1;
    while ((n % (1u << mp)) == 0) mp++;          /* srla_coder.c:358-364 */
    mp = (mp - 1 < SRLA_MAX_PORDER) ? (mp - 1) : SRLA_MAX_PORDER;
    g->max_porder = mp;
    g->fine_len = n >> mp;
    g->welch_divisor = 4.0 * pow(n - 1, -2.0);   /* lpc.c:259 */
    {
        const double nn = n - 1;                 /* lpc.c:282-283 */
        g->welch_comp = (15 * (nn - 1) * (nn - 1) * (nn - 1)) / (8 * nn * (nn - 2) * (nn * nn - 2 * nn + 2));
    }
    g->acorr_norm = 2.0 / n;                      /* lpc.c:335 */
}

static uint32_t rice_k_host(double mean)
{
    /* srla_coder.c:262-276 with srla_utility.c:22-33 */
    const double optx = 0.5127629514437670454896078808815218508243560791015625;
    const double rho = 1.0 / (1.0 + mean);
    const double l2 = log(log(optx) / log(1.0 - rho)) * 1.4426950408889634;
    const double v = (l2 >= 0.0) ? floor(l2 + 0.5) : -floor(-l2 + 0.5);
    return (uint32_t)((0 > v) ? 0 : v);
}

void build_rice_thresholds(double *thr /* [32] */)
{
    for (int j = 1; j <= 32; j++) {
        /* smallest non-negative double `mean` with k(mean) >= j */
        uint64_t lo = 0;                      /* bits of +0.0: k = 0 < j */
        uint64_t hi;
        const double top = 17179869184.0;     /* 2^34, above any partition mean */
        memcpy(&hi, &top, 8);
        if (rice_k_host(top) < (uint32_t)j) { thr[j - 1] = INFINITY; continue; }
        while (hi - lo > 1) {
            const uint64_t mid = lo + (hi - lo) / 2;
            double d;
            memcpy(&d, &mid, 8);
            if (rice_k_host(d) >= (uint32_t)j) hi = mid; else lo = mid;
        }
        memcpy(&thr[j - 1], &hi, 8);
    }
}

}  // namespace srla
