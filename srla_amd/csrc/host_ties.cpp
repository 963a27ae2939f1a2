/*
 * host_ties.cpp -- arbitration of near-ties with the host libm.
 *
 * Two decisions of the reference hang on libm functions whose last bit the device cannot promise to reproduce:
 *   the LPC order      srla_encoder.c:934-957 compares code-length estimates built on log() (:873-885);
 *   the LTP taps       lpc.c:591 inverts the Cholesky diagonal with pow(x, -0.5), and the taps are then rounded to 6 bits
 *                      (srla_encoder.c:1031-1047).
 * The device decides both itself (its own log, a correctly rounded x^-1/2) and FLAGS an item whenever a different last bit
 * could change the outcome: the two best estimates closer than tie_rel, a tap within tie_ltp of a rounding boundary
 * (kernels.hip: srla_lpc_solve_regs / srla_order_select / srla_pitch_solve).  Flagged items are rare (none on ordinary
 * audio); for each the host redoes the decision here with the platform libm -- the very functions the reference would call
 * on this machine -- from the device's bit-exact inputs (error variances, lags).  Where the host decides otherwise the
 * item gets an override (forced order / forced taps in its descriptor) and its job is analysed again.
 */
#include "host_impl.h"

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

}  // namespace

bool Impl::apply_overrides(Job &job, uint32_t jobkey)
{
    if (overrides.empty()) return false;
    bool any = false;
    const auto lo = overrides.lower_bound(override_key(jobkey, 0)), hi = overrides.upper_bound(override_key(jobkey, 0xFFFFFFFFu));
    for (auto it = lo; it != hi; ++it) {
        const uint32_t item = (uint32_t)(it->first & 0xFFFFFFFFu);
        if (item >= job.items.size()) continue;
        if (it->second.forced_order >= 0) job.items[item].forced_order = it->second.forced_order;
        if (it->second.forced_ltp) job.items[item].forced_ltp = it->second.forced_ltp;
        any = true;
    }
    return any;
}

int Impl::arbitrate(Slot &s, uint32_t jobkey)
{
    uint32_t count = 0;
    if (hipMemcpy(&count, s.d_ties.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (count == 0) return 0;
    const size_t n_items = s.job.items.size();
    if (count > 2 * n_items) return -1;
    std::vector<uint32_t> list(count);
    if (hipMemcpy(list.data(), s.d_ties.as<uint32_t>() + 1, (size_t)count * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    const uint32_t P = preset_order();
    /* few flagged items: fetch their columns one by one; many (the tests' widened thresholds): everything at once */
    const bool bulk = count > 24;
    std::vector<double> err;
    std::vector<uint32_t> orders;
    if (bulk) {
        err.resize((size_t)(P + 1) * n_items);
        orders.resize(n_items);
        if (hipMemcpy(err.data(), s.d_err.p, err.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (hipMemcpy2D(orders.data(), 4, reinterpret_cast<const uint8_t *>(s.d_results.p) + offsetof(SrlaItemResult, lpc_order), sizeof(SrlaItemResult),
                        4, n_items, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    int mismatches = 0;
    std::vector<double> col(P + 1);
    /* LTP entries first: an item whose taps the host overrules is filtered differently next time, so what the device
     * decided about its LPC order in this run says nothing -- it is looked at again (and flagged again, if close) in the next */
    std::vector<uint8_t> retaken(n_items, 0);
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t item = list[k] & 0x7FFFFFFFu, kind = list[k] >> 31;
        if (item >= n_items) return -1;
        if (kind != 1) continue;
        double td[8];
        if (hipMemcpy(td, s.d_tie_data.as<double>() + 8 * (size_t)k, sizeof(td), hipMemcpyDeviceToHost) != hipSuccess) return -1;
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
        const uint32_t item = list[k] & 0x7FFFFFFFu, kind = list[k] >> 31;
        if (kind != 0 || retaken[item]) continue;
        uint32_t dev_order = 0;
        if (bulk) {
            for (uint32_t o = 0; o <= P; o++) col[o] = err[(size_t)o * n_items + item];
            dev_order = orders[item];
        } else {
            if (hipMemcpy2D(col.data(), 8, s.d_err.as<double>() + item, n_items * 8, 8, P + 1, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            if (hipMemcpy(&dev_order, reinterpret_cast<const uint8_t *>(s.d_results.as<SrlaItemResult>() + item) + offsetof(SrlaItemResult, lpc_order), 4,
                          hipMemcpyDeviceToHost) != hipSuccess) return -1;
        }
        const SrlaItemDesc &it = s.job.items[item];
        const uint32_t host_order = select_order(col.data(), P, geoms[it.geom].welch_comp, it.n, par.bits_per_sample);
        if (host_order == dev_order) stats.num_tie_resolved++;
        else { overrides[override_key(jobkey, item)].forced_order = (int32_t)host_order; stats.num_tie_overrides++; mismatches++; }
    }
    return mismatches;
}
