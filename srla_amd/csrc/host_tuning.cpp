/*
 * host_tuning.cpp -- every environment variable the library reads, in one place.
 *
 * Impl::read_environment() runs once per handle (from init_device) and fills the handle's fields; what the kernel
 * launchers need goes to kernels.hip through srla_set_launch_tuning.  INTEGRATION.md lists the same names for users.
 */
#include "host_impl.h"

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

namespace {
bool is_set(const char *name) { return getenv(name) != nullptr; }
long long number(const char *name, long long fallback)
{
    const char *e = getenv(name);
    return e ? atoll(e) : fallback;
}
}

void Impl::read_environment()
{
    /* ---- behaviour ------------------------------------------------------------------------------------------------- */
    no_chain = is_set("SRLA_MI355X_NO_CHAIN");                 /* history-dependent last windows analysed with zeros */
    no_speculation = is_set("SRLA_MI355X_NO_SPECULATION");     /* host input: the OR pass for the offset shift always runs first */
    no_pack16 = is_set("SRLA_MI355X_NO_PACK16");               /* host input always crosses PCIe as int32 */
    force_staging = is_set("SRLA_MI355X_STAGING");             /* never let the device read / write the caller's buffers */
    if (is_set("SRLA_MI355X_HYBRID")) hybrid_inplace = number("SRLA_MI355X_HYBRID", 1) != 0;
    if (is_set("SRLA_MI355X_PIN_INPLACE")) pin_inplace = number("SRLA_MI355X_PIN_INPLACE", 0) != 0 ? 1 : 0;   /* else by pool size */

    /* ---- sizing ---------------------------------------------------------------------------------------------------- */
    if (is_set("SRLA_MI355X_SLOTS")) {
        /* the software pipeline keeps up to depth + 1 = 5 jobs in flight: fewer buffer sets would be reused before their
         * job has been collected; + 2 slots for the tail jobs, + 3 for chain mode */
        const int v = (int)number("SRLA_MI355X_SLOTS", 0);
        if (v >= 5 && v <= (int)kMaxRotating) kSlots = (uint32_t)v;
        else fprintf(stderr, "[srla-mi355x] SRLA_MI355X_SLOTS=%d ignored: %d..%d job buffer sets are supported\n", v, 5, (int)kMaxRotating);
    }
    { const long long v = number("SRLA_MI355X_JOB_SAMPLES", 0); if (v >= 65536) job_samples = (uint64_t)v; }
    { const long long v = number("SRLA_MI355X_SHORT_MIN", 0); if (v >= 16384) short_min = (uint32_t)v; }
    { const long long v = number("SRLA_MI355X_MID_JOBS", -1); if (v >= 0 && v <= 16) mid_jobs = (uint32_t)v; }
    { const long long v = number("SRLA_MI355X_SHORT_DIV", 0); if (v >= 2 && v <= 64) short_div = (uint32_t)v; }
    { const long long v = number("SRLA_MI355X_PACK_THREADS", 0); if (v > 0) env_pack_threads = (uint32_t)v; }

    /* ---- measured alternatives kept as options -------------------------------------------------------- */
    split_ltp_stage = !is_set("SRLA_MI355X_NO_LTP_SKEW");                 /* set: the pitch solve back on stream W */
    if (is_set("SRLA_MI355X_DMA_OUT")) dma_out = number("SRLA_MI355X_DMA_OUT", 1) != 0;
    direct_tail = number("SRLA_MI355X_DIRECT_TAIL", 1) != 0;            /* 0: a call's last job leaves through srla_stream_out (round 5) */
    welch_table = number("SRLA_MI355X_WELCH_TABLE", 1) != 0;            /* 0: the Welch window's weights formed per sample in the kernel (round 5) */
    SrlaLaunchTuning lt = {};
    lt.pack_lds_cap_words = (uint32_t)std::max<long long>(0, number("SRLA_MI355X_PACK_LDS_WORDS", 0));   /* tests: reach the global-memory pack path */
    lt.fft_wp = (uint32_t)std::min<long long>(2, std::max<long long>(0, number("SRLA_MI355X_FFT_WP", 1)));   /* 0: round 4's transform (a workgroup barrier per stage); 2: the 8192-point class on sixteen sub-regions too (round 6: slower) */
    lt.fir_mfma = number("SRLA_MI355X_FIR_MFMA", 1) != 0 ? 1u : 0u;          /* 0: the FIR on v_dot2 / v_dot4 (round 4) */
    srla_set_launch_tuning(&lt);

    /* ---- diagnostics ------------------------------------------------------------------------------------------------ */
    timeline = is_set("SRLA_MI355X_TIMELINE");                 /* device-clock start / end of every stage of every job */
    timing = !is_set("SRLA_MI355X_NO_TIMING");
    { const long long v = number("SRLA_MI355X_TIMING_STRIDE", 0); if (v >= 1) timing_stride = (uint32_t)v; }
    chain_trace = is_set("SRLA_MI355X_CHAIN_TRACE");
    if (const char *e = getenv("SRLA_MI355X_TIE_TEST")) {
        /* "rel,ltp,logscale,ltpbias": widens the near-tie thresholds and falsifies the device's log / its scaled LTP taps, so
         * that the host arbitration has real work to do (tests/test_gpu_ties.py) */
        double a = 0, b = 0, c = 1, d = 0;
        if (sscanf(e, "%lf,%lf,%lf,%lf", &a, &b, &c, &d) == 4) { tie_rel = a; tie_ltp = b; tie_logscale = c; tie_ltpbias = d; }
    }
}
