/*
 * host_ties.cpp -- arbitration of near-ties with the host libm.
 *
 * Three decisions of the reference hang on libm functions whose last bit the device cannot promise to reproduce:
 *   the LPC order      srla_encoder.c:934-957 compares code-length estimates built on log() (:873-885);
 *   the LTP taps       lpc.c:591 inverts the Cholesky diagonal with pow(x, -0.5), and the taps are then rounded to 6 bits
 *                      (srla_encoder.c:1031-1047);
 *   the SVR refinement (--svr-filter-learning-iteration) lpc.c:1083-1127 keeps the best of its iterates and stops by comparing
 *                      objective values built on log() and pow() (:1023-1033).  A flagged item's whole refinement is redone here
 *                      (arbitrate_svr) from the device's bit-exact lags and samples; where the predictor differs in any bit the
 *                      host's replaces it (a row of the job's forced-predictor table).
 * The device decides both itself (its own log, a correctly rounded x^-1/2) and FLAGS an item whenever a different last bit
 * could change the outcome: the two best estimates closer than tie_rel, a tap within tie_ltp of a rounding boundary
 * (kernels.hip: srla_lpc_solve_regs / srla_order_select / srla_pitch_solve).  Flagged items are rare (none on ordinary
 * audio); for each the host redoes the decision here with the platform libm -- the very functions the reference would call
 * on this machine -- from the device's bit-exact inputs (error variances, lags).  Where the host decides otherwise the
 * item gets an override (forced order / forced taps in its descriptor) and its job is analysed again.
 */
#include "host_impl.h"

#include <algorithm>
#include <float.h>
#include <math.h>
#include <string.h>

namespace {

/* srla_utility.c:22-25 */
double round_half_away(double d) { return (d >= 0.0) ? floor(d + 0.5) : -floor(-d + 0.5); }

/* srla_encoder.c:873-885 with SRLAUtility_Log2 (srla_utility.c:28-33): log(x) * 1.4426950408889634 */
double geometric_entropy(double mean_abs_error, uint32_t bps)
{
    const double intmean = mean_abs_error * (double)(1 << (bps - 1));
    const double rho = 1.0 / (1.0 + intmean);
    const double invrho = 1.0 - rho;
    if (mean_abs_error < 1e-16) return 0.0;
    return -(invrho * (log(invrho) * 1.4426950408889634) + rho * (log(rho) * 1.4426950408889634)) / rho;
}

/* srla_encoder.c:934-957 on the compensated error variances err[o] * comp (lpc.c:490-497) */
uint32_t select_order(const double *err, uint32_t max_order, double comp, uint32_t num_samples, uint32_t bps)
{
    uint32_t best = 0;
    double minlen = FLT_MAX;
    for (uint32_t order = 1; order <= max_order; order++) {
        const double ev = err[order] * comp;
        const double mabse = 2.0 * sqrt(ev / 2.0);
        double len = geometric_entropy(mabse, bps) * num_samples;
        len += 8 * order;
        if (minlen > len) { minlen = len; best = order; }
    }
    return best;
}

/* lpc.c:1620-1645 (ridge, Toeplitz matrix, Cholesky :573-600 with pow(sum, -0.5), solve :605-631), then the 6-bit
 * quantiser and the tap reversal of srla_encoder.c:1031-1047.  td: R(0), R(1), R(2), R(p-1), R(p), R(p+1).
 * Returns the taps packed as the device packs them, or 0xFFFFFFFF if the matrix is singular. */
uint32_t ltp_taps(const double *td, uint32_t ltp_order)
{
    const int dim = (int)ltp_order;
    const double rl[3] = { td[0] * (1.0 + 1e-5), td[1], td[2] };
    double am[3][3], inv_diag[3], xs[3] = { 0, 0, 0 };
    for (int j = 0; j < dim; j++) for (int k = j; k < dim; k++) am[j][k] = am[k][j] = rl[k - j];
    for (int i = 0; i < dim; i++) {
        double sum = am[i][i];
        for (int k = i - 1; k >= 0; k--) sum -= am[i][k] * am[i][k];
        if (sum <= 0.0) return 0xFFFFFFFFu;
        inv_diag[i] = pow(sum, -0.5);
        for (int j = i + 1; j < dim; j++) {
            sum = am[i][j];
            for (int k = i - 1; k >= 0; k--) sum -= am[i][k] * am[j][k];
            am[j][i] = sum * inv_diag[i];
        }
    }
    double b[3] = { 0, 0, 0 };
    for (int i = 0; i < dim; i++) b[i] = td[3 + (dim == 3 ? i : 1)];      /* auto_corr[period - order / 2 + i] */
    for (int i = 0; i < dim; i++) {
        double sum = b[i];
        for (int j = i - 1; j >= 0; j--) sum -= am[i][j] * xs[j];
        xs[i] = sum * inv_diag[i];
    }
    for (int i = dim - 1; i >= 0; i--) {
        double sum = xs[i];
        for (int j = i + 1; j < dim; j++) sum -= am[j][i] * xs[j];
        xs[i] = sum * inv_diag[i];
    }
    int32_t q[3] = { 0, 0, 0 };
    for (int i = 0; i < dim; i++) {
        int32_t c = (int32_t)round_half_away(xs[i] * 32.0);
        c = (c < -32) ? -32 : ((c > 31) ? 31 : c);
        q[i] = c;
    }
    for (int i = 0; i < dim / 2; i++) { const int32_t t = q[i]; q[i] = q[dim - 1 - i]; q[dim - 1 - i] = t; }
    return ((uint32_t)q[0] & 63u) | (((uint32_t)q[1] & 63u) << 6) | (((uint32_t)q[2] & 63u) << 12);
}


/* lpc.c:1023-1033 with BITS_PER_SAMPLE = 16 (:1042) and LPC_Log2 = log(x) / log(2) ... as the reference spells it */
double svr_rgr_mean_code_length(double mean_abs_error)
{
    const double intmean = mean_abs_error * (double)(1 << 16);
    const double rho = 1.0 / (1.0 + intmean);
    const double l2 = log(log(0.5127629514) / log(1.0 - rho)) * 1.4426950408889634;   /* LPC_Log2, lpc.c:27 */
    const uint32_t k2 = (uint32_t)((0.0 > l2) ? 0.0 : l2);
    const uint32_t k1 = k2 + 1;
    const double k1factor = pow(1.0 - rho, (double)(1u << k1));
    const double k2factor = pow(1.0 - rho, (double)(1u << k2));
    return (1.0 + k1) * (1.0 - k1factor) + (1.0 + k2 + (1.0 / (1.0 - k2factor))) * k1factor;
}

/* LPC_CalculateCoefSVR (lpc.c:1036-1136) on the normalised block `data`, from the predictor in coef[0..p): the same sums in
 * the same order, with the platform libm behind the objective (log, pow) and the Cholesky diagonal (pow(x, -0.5)). */
void svr_refine(const std::vector<double> &data, std::vector<double> &coef, uint32_t max_iter)
{
    static const double margin_list[] = { 0.0, 1.0 / 4096, 1.0 / 1024, 1.0 / 256, 1.0 / 64, 1.0 / 16 };   /* srla_internal.c:27 */
    const uint32_t p = (uint32_t)coef.size(), n = (uint32_t)data.size();
    if (max_iter == 0 || p == 0 || n <= p) return;
    std::vector<double> cov((size_t)p * p, 0.0), low(p), r_vec(p), delta(p), init_coef(coef), best_coef(coef), residual(n);
#define COV(a, b) cov[(size_t)(a) * p + (b)]
    for (uint32_t smpl = 0; smpl < n - p; smpl++) {                                   /* lpc.c:987-1020 */
        const double *pd = &data[smpl];
        for (uint32_t i = 0; i < p; i++) {
            const double sv = pd[i];
            for (uint32_t j = i; j < p; j++) COV(i, j) += sv * pd[j];
        }
    }
    for (uint32_t i = 0; i < p; i++) for (uint32_t j = i + 1; j < p; j++) COV(j, i) = COV(i, j);
    for (uint32_t i = 0; i < p; i++) COV(i, i) *= (1.0 + 1e-5);                       /* lpc.c:1067-1069 */
    for (uint32_t i = 0; i < p; i++) {                                                /* lpc.c:573-600 */
        double sum = COV(i, i);
        for (int k = (int)i - 1; k >= 0; k--) sum -= COV(i, k) * COV(i, k);
        if (sum <= 0.0) { std::fill(coef.begin(), coef.end(), 0.0); return; }         /* lpc.c:1071-1077 */
        low[i] = pow(sum, -0.5);
        for (uint32_t j = i + 1; j < p; j++) {
            sum = COV(i, j);
            for (int k = (int)i - 1; k >= 0; k--) sum -= COV(i, k) * COV(j, k);
            COV(j, i) = sum * low[i];
        }
    }
    double min_obj = FLT_MAX;
    for (const double margin : margin_list) {
        double prev_obj = FLT_MAX;
        coef = init_coef;
        for (uint32_t itr = 0; itr < max_iter; itr++) {
            double mabse = 0.0;
            residual = data;
            std::fill(r_vec.begin(), r_vec.end(), 0.0);
            for (uint32_t smpl = p; smpl < n; smpl++) {
                for (uint32_t i = 0; i < p; i++) residual[smpl] += coef[i] * data[smpl - i - 1];
                double r = residual[smpl];
                const double a = (r > 0) ? r : -r;
                mabse += a;
                r = (double)((r > 0) - (r < 0)) * (((a - margin) > 0.0) ? (a - margin) : 0.0);   /* LPC_SOFT_THRESHOLD, lpc.c:34 */
                residual[smpl] = r;
                for (uint32_t i = 0; i < p; i++) r_vec[i] += r * data[smpl - i - 1];
            }
            const double obj = svr_rgr_mean_code_length(mabse / n);
            for (uint32_t i = 0; i < p; i++) {                                          /* lpc.c:605-631 */
                double sum = r_vec[i];
                for (int k = (int)i - 1; k >= 0; k--) sum -= COV(i, k) * delta[k];
                delta[i] = sum * low[i];
            }
            for (int k = (int)p - 1; k >= 0; k--) {
                double sum = delta[k];
                for (uint32_t j = (uint32_t)k + 1; j < p; j++) sum -= COV(j, k) * delta[j];
                delta[k] = sum * low[k];
            }
            if (obj < min_obj) { best_coef = coef; min_obj = obj; }
            if ((prev_obj < obj) || (fabs(prev_obj - obj) < 1e-8)) break;
            for (uint32_t i = 0; i < p; i++) coef[i] += delta[i];
            prev_obj = obj;
        }
    }
    coef = best_coef;
#undef COV
}

/* The predictor of `order` from the ridge-regularised lags r[0..order] (lpc.c:379-441: gamma summed in index order, the
 * update pairwise in place) -- only sums, products and one division per step, so the device's is the same to the bit. */
void levinson(const double *r, uint32_t order, std::vector<double> &coef)
{
    std::vector<double> a(order + 2, 0.0);
    const double r0 = r[0];
    a[0] = 1.0; a[1] = -r[1] / r0;
    double e = r0 + r[1] * a[1];
    for (uint32_t k = 1; k < order; k++) {
        double gamma = 0.0;
        for (uint32_t i = 0; i <= k; i++) gamma += a[i] * r[k + 1 - i];
        gamma /= -e;
        e = e * (1.0 - gamma * gamma);
        for (uint32_t i = 0; i <= (k + 1) / 2; i++) {
            const uint32_t j = k + 1 - i;
            const double ai = a[i], aj = a[j];
            a[i] = ai + gamma * aj;
            if (i != j) a[j] = aj + gamma * ai;
        }
    }
    coef.assign(a.begin() + 1, a.begin() + 1 + order);
}

}  // namespace

/* Test hooks (include/srla_mi355x.h, "test hooks"): the arbitration arithmetic above on its own, no device involved, so that a
 * CPU-only test can feed it the oracle's inputs and compare bit for bit (tests/test_host_ties.py). */
extern "C" {
uint32_t SRLAMI355X_TestSelectOrder(const double *error_vars, uint32_t max_order, double compensation, uint32_t num_samples, uint32_t bits_per_sample)
{
    return select_order(error_vars, max_order, compensation, num_samples, bits_per_sample);
}
uint32_t SRLAMI355X_TestLtpTaps(const double *lags6, uint32_t ltp_order) { return ltp_taps(lags6, ltp_order); }
void SRLAMI355X_TestSvrRefine(const double *data, uint32_t num_samples, double *coef, uint32_t order, uint32_t max_iter)
{
    std::vector<double> d(data, data + num_samples), c(coef, coef + order);
    svr_refine(d, c, max_iter);
    std::copy(c.begin(), c.end(), coef);
}
void SRLAMI355X_TestLevinson(const double *lags_ridged, uint32_t order, double *coef)
{
    std::vector<double> c;
    levinson(lags_ridged, order, c);
    std::copy(c.begin(), c.end(), coef);
}
}

bool Impl::apply_overrides(Job &job, uint32_t jobkey)
{
    if (overrides.empty()) return false;
    bool any = false;
    job.svr_rows.clear();
    for (SrlaItemDesc &it : job.items) it.forced_svr = 0;
    const auto lo = overrides.lower_bound(override_key(jobkey, 0)), hi = overrides.upper_bound(override_key(jobkey, 0xFFFFFFFFu));
    for (auto it = lo; it != hi; ++it) {
        const uint32_t item = (uint32_t)(it->first & 0xFFFFFFFFu);
        if (item >= job.items.size()) continue;
        if (it->second.forced_order >= 0) job.items[item].forced_order = it->second.forced_order;
        if (it->second.forced_ltp) job.items[item].forced_ltp = it->second.forced_ltp;
        if (!it->second.svr_row.empty()) {
            std::vector<double> row(256, 0.0);
            std::copy(it->second.svr_row.begin(), it->second.svr_row.begin() + std::min<size_t>(256, it->second.svr_row.size()), row.begin());
            job.svr_rows.insert(job.svr_rows.end(), row.begin(), row.end());
            job.items[item].forced_svr = (uint32_t)(job.svr_rows.size() / 256);
        }
        any = true;
    }
    return any;
}

int Impl::arbitrate(Slot &s, uint32_t jobkey)
{
    uint32_t count = 0;
    const size_t n_items = s.job.items.size();
    const uint32_t P = preset_order();
    /* a job that went through the block assembly left a few near-ties' numbers in pinned host memory (SrlaTieGather): no copies */
    const double *gathered = nullptr;
    const uint32_t gstride = std::max<uint32_t>(P + 2u, 8u);
    if (s.ties_gathered && s.h_info.p != nullptr) {
        const uint32_t c = s.h_info.as<SrlaJobInfo>()->num_tie_items;
        if (c != 0 && c <= SRLA_TIE_GATHER_CAP) { count = c; gathered = s.h_ties.as<double>(); }
    }
    if (gathered == nullptr && !d2h(&count, s.d_ties.p, 4)) return -1;
    if (count == 0) return 0;
    if (count > 3 * n_items) return -1;
    std::vector<uint32_t> list(count);
    if (gathered != nullptr) { for (uint32_t k = 0; k < count; k++) list[k] = (uint32_t)gathered[k]; }
    else if (!d2h(list.data(), s.d_ties.as<uint32_t>() + 1, (size_t)count * 4)) return -1;
    /* few flagged items: fetch their columns one by one; many (the tests' widened thresholds): everything at once */
    const bool bulk = count > 24;
    std::vector<double> err;
    std::vector<uint32_t> orders;
    if (bulk) {
        err.resize((size_t)(P + 1) * n_items);
        orders.resize(n_items);
        if (!d2h(err.data(), s.d_err.p, err.size() * sizeof(double))) return -1;
        if (!d2h_2d(orders.data(), 4, reinterpret_cast<const uint8_t *>(s.d_results.p) + offsetof(SrlaItemResult, lpc_order), sizeof(SrlaItemResult), 4, n_items)) return -1;
    }
    int mismatches = 0;
    std::vector<double> col(P + 1);
    /* LTP entries first: an item whose taps the host overrules is filtered differently next time, so what the device
     * decided about its LPC order in this run says nothing -- it is looked at again (and flagged again, if close) in the next */
    std::vector<uint8_t> retaken(n_items, 0);
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t item = list[k] & 0x3FFFFFFFu, kind = list[k] >> 30;
        if (item >= n_items) return -1;
        if (kind != 1) continue;
        double td[8];
        if (gathered != nullptr) memcpy(td, gathered + SRLA_TIE_GATHER_CAP + (size_t)k * gstride, sizeof(td));
        else if (!d2h(td, s.d_tie_data.as<double>() + 8 * (size_t)k, sizeof(td))) return -1;
        const uint32_t host_q = ltp_taps(td, par.ltp_order), dev_q = (uint32_t)td[7];
        if (host_q == 0xFFFFFFFFu || host_q == dev_q) stats.num_tie_resolved++;
        else {
            Override &ov = overrides[override_key(jobkey, item)];
            ov.forced_ltp = 0x80000000u | host_q;
            ov.forced_order = -1;
            retaken[item] = 1;
            stats.num_tie_overrides++; mismatches++;
        }
    }
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t item = list[k] & 0x3FFFFFFFu, kind = list[k] >> 30;
        if (kind != 0 || retaken[item]) continue;
        uint32_t dev_order = 0;
        if (gathered != nullptr) {
            const double *g = gathered + SRLA_TIE_GATHER_CAP + (size_t)k * gstride;
            for (uint32_t o = 0; o <= P; o++) col[o] = g[o];
            dev_order = (uint32_t)g[P + 1];
        } else if (bulk) {
            for (uint32_t o = 0; o <= P; o++) col[o] = err[(size_t)o * n_items + item];
            dev_order = orders[item];
        } else {
            if (!d2h_2d(col.data(), 8, s.d_err.as<double>() + item, n_items * 8, 8, P + 1)) return -1;
            if (!d2h(&dev_order, reinterpret_cast<const uint8_t *>(s.d_results.as<SrlaItemResult>() + item) + offsetof(SrlaItemResult, lpc_order), 4)) return -1;
        }
        const SrlaItemDesc &it = s.job.items[item];
        const uint32_t host_order = select_order(col.data(), P, geoms[it.geom].welch_comp, it.n, par.bits_per_sample);
        if (host_order == dev_order) stats.num_tie_resolved++;
        else { overrides[override_key(jobkey, item)].forced_order = (int32_t)host_order; stats.num_tie_overrides++; mismatches++; retaken[item] = 1; }
    }
    /* SVR entries last: an item whose taps or order were overruled above is refined from another start next time */
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t item = list[k] & 0x3FFFFFFFu, kind = list[k] >> 30;
        if (kind != 2 || retaken[item]) continue;
        stats.num_svr_tie_items++;
        const int r = arbitrate_svr(s, jobkey, item);
        if (r < 0) return -1;
        if (r == 0) stats.num_tie_resolved++;
        else { stats.num_tie_overrides++; mismatches++; }
    }
    return mismatches;
}

int Impl::arbitrate_svr(Slot &s, uint32_t jobkey, uint32_t item)
{
    const size_t n_items = s.job.items.size();
    const SrlaItemDesc &it = s.job.items[item];
    const uint32_t nch = par.num_channels, n = it.n, P = preset_order();
    SrlaItemResult head;
    if (!d2h(&head, s.d_results.as<SrlaItemResult>() + item, offsetof(SrlaItemResult, lpc_coef))) return -1;
    const uint32_t order = head.lpc_order;
    if (order == 0 || order > P) return 0;
    /* the block as the analysis saw it: variant (srla_utility.c:91-103), offset shift, pre-emphasis (srla_utility.c:342),
     * long-term predictor (srla_lpc_predict.c:267-294) */
    uint32_t lsh = it.lshift;
    if (s.jp.lshift_dev != nullptr) {
        if (hipEventSynchronize(ev_or) != hipSuccess) return -1;
        lsh = h_or.as<uint32_t>()[1];
    }
    std::vector<int32_t> a(n), b;
    auto plane = [&](uint32_t ch, std::vector<int32_t> &dst) {
        return d2h(dst.data(), s.in_cur + (size_t)ch * s.stride_cur + it.sample_off, (size_t)n * 4);
    };
    if (it.variant < nch) { if (!plane(it.variant, a)) return -1; for (auto &v : a) v >>= lsh; }
    else {
        b.resize(n);
        if (!plane(0, a) || !plane(1, b)) return -1;
        for (uint32_t i = 0; i < n; i++) {
            const int32_t l = a[i] >> lsh, r = b[i] >> lsh;
            const int32_t sd = (int32_t)((uint32_t)r - (uint32_t)l);
            a[i] = (it.variant == nch + 1) ? sd : (int32_t)((uint32_t)l + (uint32_t)(sd >> 1));
        }
    }
    {
        int32_t prev = a[0];
        for (uint32_t i = 0; i < n; i++) {
            const int32_t cur = a[i];
            a[i] = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)head.preemph_coef) >> 4));
            prev = cur;
        }
    }
    if (head.ltp_period > 0 && par.ltp_order > 0) {
        const uint32_t taps = par.ltp_order, half = taps >> 1, period = head.ltp_period;
        std::vector<int32_t> y(a);
        for (uint32_t sidx = period + half + 1; sidx < n; sidx++) {
            const uint32_t base = sidx - period - half;
            uint32_t acc = 16u + (uint32_t)head.ltp_coef[0] * (uint32_t)y[base];
            if (taps == 3) acc += (uint32_t)head.ltp_coef[1] * (uint32_t)y[base + 1] + (uint32_t)head.ltp_coef[2] * (uint32_t)y[base + 2];
            a[sidx] = (int32_t)((uint32_t)y[sidx] - (uint32_t)((int32_t)acc >> 5));
        }
    }
    const double norm = ldexp(1.0, -(int)(par.bits_per_sample - 1));
    std::vector<double> data(n);
    for (uint32_t i = 0; i < n; i++) data[i] = (double)a[i] * norm;
    /* the predictor the refinement started from: the recursion on the item's lags */
    std::vector<double> lags(order + 1);
    if (!d2h_2d(lags.data(), 8, s.d_lags.as<double>() + item, n_items * 8, 8, order + 1)) return -1;
    lags[0] *= (1.0 + 1e-5);                                              /* ridge, lpc.c:483 */
    std::vector<double> coef;
    if (fabs(lags[0]) < (double)FLT_EPSILON) coef.assign(order, 0.0);     /* lpc.c:395-405 */
    else levinson(lags.data(), order, coef);
    svr_refine(data, coef, par.num_svr_filter_learning_iteration);
    const uint32_t ws_stride = (P <= 64) ? 64u : 256u;
    std::vector<double> dev(order);
    if (!d2h(dev.data(), s.d_coef_ws.as<double>() + (size_t)item * ws_stride, (size_t)order * 8)) return -1;
    if (memcmp(dev.data(), coef.data(), (size_t)order * 8) == 0) return 0;
    overrides[override_key(jobkey, item)].svr_row = coef;
    if (jobkey >= kChainJobKey) {
        /* chain / history mode: later blocks of the window inherit the refinement's residual (lpc.c:1047), and an item whose
         * predictor is forced leaves the transform's words instead -- named and counted, not silent */
        stats.num_nonidentical_calls++;
        stats.nonidentical_reasons |= SRLAMI355X_NONIDENTICAL_SVR_HISTORY;
        if (!(warned_reasons & SRLAMI355X_NONIDENTICAL_SVR_HISTORY)) {
            warned_reasons |= SRLAMI355X_NONIDENTICAL_SVR_HISTORY;
            fprintf(stderr, "[srla-mi355x] WARNING: output valid and lossless but NOT guaranteed bit-identical to the reference: %s\n",
                    nonidentical_text(SRLAMI355X_NONIDENTICAL_SVR_HISTORY).c_str());
        }
    }
    return 1;
}
