/*
 * host_pipeline.cpp -- device set-up, staging of host input, the staged execution of jobs and the stream loop
 * (see host_impl.h for the overall picture).
 */
#include "host_impl.h"

#include <algorithm>
#include <stdlib.h>
#include <string.h>

#include <thread>

namespace srla {
int g_device_index = 0;
}

void Impl::tl_printf(const char *fmt, ...)
{
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    tl_log += buf;
}

Impl::~Impl()
{
    delete pool;
    if (dev_ready) {
        (void)hipSetDevice(device);
        for (auto &s : slot) {
            for (auto &st : streams) if (st) (void)hipStreamSynchronize(st);
            for (auto &e : s.t0) if (e) (void)hipEventDestroy(e);
            for (auto &e : s.t1) if (e) (void)hipEventDestroy(e);
            if (s.ev_in) (void)hipEventDestroy(s.ev_in);
            s.d_input16.release();
            DevBuf *db[] = { &s.d_input, &s.d_items, &s.d_cands, &s.d_windows, &s.d_results, &s.d_res_ws,
                             &s.d_blocks, &s.d_block_off, &s.d_ctl, &s.d_scratch, &s.d_dbg, &s.d_lags, &s.d_err, &s.d_class_index, &s.d_stream };
            for (auto *b : db) b->release();
            PinBuf *pb[] = { &s.h_in, &s.h_stream, &s.h_info };
            for (auto *b : pb) b->release();
        }
        for (auto &st : streams) if (st) (void)hipStreamDestroy(st);
        if (upload) (void)hipStreamDestroy(upload);
        if (chain_stream) { (void)hipStreamSynchronize(chain_stream); (void)hipStreamDestroy(chain_stream); }
        if (ev_or) (void)hipEventDestroy(ev_or);
        if (ev_ref) (void)hipEventDestroy(ev_ref);
        h_or.release();
        d_tw.release(); d_geoms.release(); d_thr.release(); d_huff.release(); d_huffcode.release(); d_pos.release(); d_or.release();
        d_chain_pool.release(); d_chain_tab.release();
        for (auto &b : d_chain_list) b.release();
        for (auto &b : d_chain_select) b.release();
    }
}

bool Impl::init_device()
{
    /* every entry point comes through here: the calling thread works on the handle's device from now on (handles of
     * several threads, or of several devices in one process, keep to their own) */
    if (dev_ready) return hipSetDevice(device) == hipSuccess;
    if (dev_failed) return false;
    dev_failed = true;
    if (const char *e = getenv("SRLA_MI355X_SLOTS")) {
        /* the software pipeline of encode_stream keeps depth + 1 = 4 jobs in flight: fewer buffer sets would be reused before
         * their job has been collected; + 2 slots for the tail jobs, + 3 for chain mode */
        const int v = atoi(e);
        if (v >= 4 && v + 5 <= (int)kMaxSlots) kSlots = (uint32_t)v;
        else fprintf(stderr, "[srla-mi355x] SRLA_MI355X_SLOTS=%d ignored: %d..%d job buffer sets are supported\n", v, 4, (int)kMaxSlots - 5);
    }
    if (const char *e = getenv("SRLA_MI355X_JOB_SAMPLES")) { const long long v = atoll(e); if (v >= 65536) job_samples = (uint64_t)v; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        fprintf(stderr, "[srla-mi355x] no HIP device available: the MI355X encode path cannot run "
                        "(there is no CPU fallback)\n");
        return false;
    }
    HIP_OK(hipSetDevice(device));
    {
        /* W (critical path) and N (its few wavefronts gate the next wide kernel) run at high priority, the block
         * assembly on C at low priority: measured +2 % over every other assignment (SRLA_MI355X_PRIO to experiment) */
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        int pr[3] = { hi, hi, lo };
        if (const char *e = getenv("SRLA_MI355X_PRIO")) {       /* experiment: e.g. "hlh": h = high, l = low, m = middle */
            for (int i = 0; i < 3 && e[i]; i++) pr[i] = (e[i] == 'h') ? hi : ((e[i] == 'm') ? (lo + hi) / 2 : lo);
        }
        HIP_OK(hipStreamCreateWithPriority(&streams[0], hipStreamNonBlocking, pr[0]));
        HIP_OK(hipStreamCreateWithPriority(&streams[1], hipStreamNonBlocking, pr[1]));
        HIP_OK(hipStreamCreateWithPriority(&streams[2], hipStreamNonBlocking, pr[2]));
    }
    HIP_OK(hipEventCreate(&ev_or));
    HIP_OK(hipEventCreate(&ev_ref));
    timeline = getenv("SRLA_MI355X_TIMELINE") != nullptr;
    HIP_OK(hipStreamCreateWithFlags(&upload, hipStreamNonBlocking));
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIP_OK(hipStreamCreateWithPriority(&chain_stream, hipStreamNonBlocking, hi));   /* a few workgroups per launch, latency bound */
    }
    if (!h_or.ensure(64)) return false;
    for (uint32_t si = 0; si < kMaxSlots; si++) {
        Slot &s = slot[si];
        s.stream = streams[0];
        for (auto &e : s.t0) HIP_OK(hipEventCreate(&e));
        for (auto &e : s.t1) HIP_OK(hipEventCreate(&e));
        HIP_OK(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
    }
    double thr[32];
    srla::build_rice_thresholds(thr);
    if (!d_thr.ensure(sizeof(thr))) return false;
    HIP_OK(hipMemcpy(d_thr.p, thr, sizeof(thr), hipMemcpyHostToDevice));
    uint8_t huff[512];
    memcpy(huff, srla::huffman_plain_lengths(), 256);
    memcpy(huff + 256, srla::huffman_summed_lengths(), 256);
    if (!d_huff.ensure(sizeof(huff))) return false;
    HIP_OK(hipMemcpy(d_huff.p, huff, sizeof(huff), hipMemcpyHostToDevice));
    {
        uint32_t codes[512];
        for (int i = 0; i < 256; i++) { codes[i] = srla::huffman_plain_codes()[i]; codes[256 + i] = srla::huffman_summed_codes()[i]; }
        if (!d_huffcode.ensure(sizeof(codes))) return false;
        HIP_OK(hipMemcpy(d_huffcode.p, codes, sizeof(codes), hipMemcpyHostToDevice));
    }
    if (!d_pos.ensure(64)) return false;
    HIP_OK(hipMemset(d_pos.p, 0, 64));
    force_staging = getenv("SRLA_MI355X_STAGING") != nullptr;
    no_pack16 = getenv("SRLA_MI355X_NO_PACK16") != nullptr;
    no_speculation = getenv("SRLA_MI355X_NO_SPECULATION") != nullptr;
    timing = getenv("SRLA_MI355X_NO_TIMING") == nullptr;
    if (const char *e = getenv("SRLA_MI355X_TAIL_BOOST")) { unsigned a = 0, b = 0; if (sscanf(e, "%u,%u", &a, &b) == 2 && a >= 1) { tail_boost = a; tail_boost_jobs = b; } }
    if (const char *e = getenv("SRLA_MI355X_TIMING_STRIDE")) { const int v = atoi(e); if (v >= 1) timing_stride = (uint32_t)v; }
    if (!d_or.ensure(64)) return false;
    unsigned hw = std::thread::hardware_concurrency();
    /* a container's CPU quota (cgroup v2 cpu.max = "<quota> <period>") bounds the useful thread count */
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
            const unsigned q = (unsigned)((quota + period - 1) / period);
            if (q > 0 && (hw == 0 || q < hw)) hw = q;
        }
        fclose(f);
    }
    /* half of the usable CPUs, at most 8: the enqueueing thread and the HIP runtime's own helper threads
     * need the rest (measured: more pack threads than that makes launches and D2H completion slower) */
    unsigned nthreads = pack_threads ? pack_threads : std::max(1u, std::min((hw ? hw : 2u) / 2u, 8u));
    if (const char *e = getenv("SRLA_MI355X_PACK_THREADS")) { const int v = atoi(e); if (v > 0) nthreads = (unsigned)v; }
    pool = new Pool(nthreads);
    dev_failed = false;
    dev_ready = true;
    return true;
}

bool Impl::prepare_job(Slot &s, const int32_t *d_in, uint32_t d_stride, const int32_t *const *host_in, bool want_dbg)
{
    Job &job = s.job;
    const uint32_t nch = par.num_channels;
    hipStream_t W = streams[0];
    if (!sync_tables()) return false;
    const size_t n_items = job.items.size(), n_cands = job.cands.size(), n_win = job.windows.size();
    {
        const void *pi = s.d_items.p, *pc = s.d_cands.p, *pw = s.d_windows.p;
        if (!s.d_items.ensure(std::max<size_t>(1, n_items) * sizeof(SrlaItemDesc))) return false;
        if (!s.d_cands.ensure(n_cands * sizeof(SrlaCandDesc))) return false;
        if (!s.d_windows.ensure(n_win * sizeof(SrlaWindowDesc))) return false;
        const void *px = s.d_class_index.p;
        if (!s.d_class_index.ensure(std::max<size_t>(1, n_items) * sizeof(SrlaAutocorrItem))) return false;
        if (pi != s.d_items.p || pc != s.d_cands.p || pw != s.d_windows.p || px != s.d_class_index.p) job.uploaded = false;
    }
    if (!s.d_results.ensure(std::max<size_t>(1, n_items) * sizeof(SrlaItemResult))) return false;
    if (!s.d_res_ws.ensure(std::max<uint64_t>(4, job.res_elems) * 4)) return false;
    if (!s.d_blocks.ensure((size_t)job.num_slots * sizeof(SrlaBlockRecord))) return false;
    if (!s.d_block_off.ensure((size_t)job.num_slots * 4 + 16)) return false;
    if (!s.d_ctl.ensure(64)) return false;
    if (!s.h_info.ensure(sizeof(SrlaJobInfo) + n_win * 4)) return false;
    {
        /* a block is never larger than its raw form (11 + n * nch * bytes): bound of the job's stream bytes */
        const size_t bound = (size_t)job.ns * nch * (par.bits_per_sample / 8) + (size_t)job.num_slots * SRLA_PACK_SLACK + 64;
        if (!s.out_direct && !s.h_stream.ensure(bound)) return false;
        if (!s.d_stream.ensure(bound + 32)) return false;
        const SrlaJobParams probe = job_params(job, d_stride);
        if (srla_pack_needs_scratch(&probe) && !s.d_scratch.ensure(bound)) return false;
    }
    if (want_dbg && !s.d_dbg.ensure(std::max<size_t>(1, n_items) * SRLA_DBG_STRIDE * sizeof(double))) return false;
    const uint32_t lag_rows = std::max<uint32_t>(par.ltp_order > 0 ? SRLA_LTP_LAGS : 0u, preset_order() + 1);
    if (!s.d_lags.ensure((size_t)lag_rows * std::max<size_t>(1, n_items) * sizeof(double))) return false;
    if (!s.d_err.ensure((size_t)(preset_order() + 1) * std::max<size_t>(1, n_items) * sizeof(double))) return false;
    s.want_dbg = want_dbg;
    s.in_cur = d_in;
    s.stride_cur = d_stride;
    s.used_h2d = false;
    if (!d_in) {
        if ((!in_pinned && !s.h_in.ensure((size_t)nch * job.ns * 4 + 64u * nch)) || !s.d_input.ensure((size_t)nch * job.ns * 4)) return false;
        if (in_pinned) {
            /* the caller's planes are pinned: DMA straight out of them (the OR of the job's samples, when it is
             * still being gathered, is computed by the pool threads meanwhile) */
            for (uint32_t ch = 0; ch < nch; ch++)
                HIP_OK(hipMemcpyAsync(s.d_input.as<int32_t>() + (size_t)ch * job.ns, host_in[ch] + job.s0, (size_t)job.ns * 4,
                                      hipMemcpyHostToDevice, upload));
            if (spec_or_active) {
                const uint32_t chunk = 256u << 10, per_ch = (job.ns + chunk - 1) / chunk;
                const Job *jb = &job;
                pool->parallel_for(per_ch * nch, [&](uint32_t i) {
                    const uint32_t ch = i / per_ch, o = (i % per_ch) * chunk, len = std::min(chunk, jb->ns - o);
                    const int32_t *src = host_in[ch] + jb->s0 + o;
                    uint32_t m = 0;
                    for (uint32_t k = 0; k < len; k++) m |= (uint32_t)src[k];
                    spec_or.fetch_or(m, std::memory_order_relaxed);
                });
            }
        } else {
            /* pageable -> pinned staging on the pool threads, then one DMA on the upload stream */
            const uint32_t chunk = 256u << 10, per_ch = (job.ns + chunk - 1) / chunk;
            const Job *jb = &job;
            const bool track = spec_or_active;
            bool packed = false;
            if (par.bits_per_sample <= 16 && !no_pack16) {
                /* as int16 (planes padded to 16 samples so that every chunk starts 32-byte aligned) */
                const size_t stride16 = ((size_t)job.ns + 15u) & ~(size_t)15u;
                if (!s.d_input16.ensure(nch * stride16 * 2)) return false;
                int16_t *dst = s.h_in.as<int16_t>();
                std::atomic<uint32_t> wide{ 0 };
                pool->parallel_for(per_ch * nch, [&](uint32_t i) {
                    const uint32_t ch = i / per_ch, o = (i % per_ch) * chunk, len = std::min(chunk, jb->ns - o);
                    uint32_t w = 0;
                    const uint32_t m = pack16_or(dst + (size_t)ch * stride16 + o, host_in[ch] + jb->s0 + o, len, &w);
                    if (track) spec_or.fetch_or(m, std::memory_order_relaxed);
                    if (w) wide.fetch_or(w, std::memory_order_relaxed);
                });
                if (wide.load() == 0) {
                    HIP_OK(hipMemcpyAsync(s.d_input16.p, s.h_in.p, nch * stride16 * 2, hipMemcpyHostToDevice, upload));
                    if (srla_launch_widen16(upload, s.d_input16.as<int16_t>(), stride16, s.d_input.as<int32_t>(), job.ns, nch) != 0) return false;
                    packed = true;
                }   /* else: samples beyond 16 bits in a stream declared narrower -- the reference does not mind, nor do we */
            }
            if (!packed) {
                int32_t *dst = s.h_in.as<int32_t>();
                pool->parallel_for(per_ch * nch, [&](uint32_t i) {
                    const uint32_t ch = i / per_ch, o = (i % per_ch) * chunk, len = std::min(chunk, jb->ns - o);
                    const int32_t *src = host_in[ch] + jb->s0 + o;
                    int32_t *d = dst + (size_t)ch * jb->ns + o;
                    const uint32_t m = copy_or(d, src, len);   /* the copy also gathers the OR of the samples it moves */
                    if (track) spec_or.fetch_or(m, std::memory_order_relaxed);
                });
                HIP_OK(hipMemcpyAsync(s.d_input.p, s.h_in.p, (size_t)nch * job.ns * 4, hipMemcpyHostToDevice, upload));
            }
        }
        HIP_OK(hipEventRecord(s.ev_in, upload));
        s.in_cur = s.d_input.as<int32_t>();
        s.stride_cur = job.ns;
        s.used_h2d = true;
    }
    if (!job.uploaded) {
        if (n_items) HIP_OK(hipMemcpyAsync(s.d_items.p, job.items.data(), n_items * sizeof(SrlaItemDesc), hipMemcpyHostToDevice, W));
        if (n_items) HIP_OK(hipMemcpyAsync(s.d_class_index.p, job.class_index.data(), n_items * sizeof(SrlaAutocorrItem), hipMemcpyHostToDevice, W));
        HIP_OK(hipMemcpyAsync(s.d_cands.p, job.cands.data(), n_cands * sizeof(SrlaCandDesc), hipMemcpyHostToDevice, W));
        HIP_OK(hipMemcpyAsync(s.d_windows.p, job.windows.data(), n_win * sizeof(SrlaWindowDesc), hipMemcpyHostToDevice, W));
        job.uploaded = true;
    }
    if (spec_or_active && !spec_guessed) {
        /* The first job's samples are staged: guess the stream's shift from them instead of assuming 0, so that a
         * stream whose samples all carry the same trailing zeros (16-bit audio in a 24-bit container) is not encoded
         * twice.  The whole stream's shift can only be smaller; if it is, the stream is encoded again (below). */
        const uint32_t m = spec_or.load();
        uint32_t sh = 0;
        if (m != 0) while (((m >> sh) & 1u) == 0) sh++;
        offset_lshift = sh;
        spec_guessed = true;
    }
    s.jp = job_params(job, s.stride_cur);
    s.busy = true;
    stats.num_windows += n_win; stats.num_candidates += n_cands; stats.num_items += n_items;
    stats.analyzed_samples += job.analyzed_samples;
    stats.analyze_launches++;
    return true;
}

bool Impl::run_stage(Slot &s, int st)
{
    Job &job = s.job;
    /* chain-mode jobs keep to their own stream up to the pricing, so that the regular jobs never queue behind their
     * many small dependent launches; the block assembly stays on C, where the order of the stream's blocks is made */
    hipStream_t W = s.own_stream ? s.own_stream : streams[0], N = s.own_stream ? s.own_stream : streams[1], C = streams[2];
    const SrlaJobParams &jp = s.jp;
    double *dbg = s.want_dbg ? s.d_dbg.as<double>() : nullptr;
    const bool have_items = !job.groups.empty();
    int rc = 0;
    /* Stage events ride on the kernel dispatches themselves (hipExtLaunchKernel): the end event on the stage's last
     * launch, the start event (timed jobs only) on its first -- no separate marker packets between the kernels
     * of the critical stream.  A stage without launches records its end event the ordinary way. */
    hipEvent_t ev0 = s.timed ? s.t0[st] : nullptr;
    switch (st) {
    case ST_A: {
        if (lshift_on_device) HIP_OK(hipStreamWaitEvent(W, ev_or, 0));
        if (s.used_h2d) HIP_OK(hipStreamWaitEvent(W, s.ev_in, 0));
        struct L { int kind, cls, pass; };                       /* kind 0: autocorr class launch, 1: pitch solve */
        L seq[12]; int nl = 0;
        if (have_items) {
            for (int pass = (par.ltp_order > 0) ? 1 : 0; pass >= 0; pass--) {
                for (int c = 0; c < 4; c++) if (job.class_count[c]) seq[nl++] = { 0, c, pass };
                if (pass == 1) seq[nl++] = { 1, 0, 1 };
            }
        }
        static const int kClass[4] = { 0, 1, 2, 4 };     /* FFT size / 2048 (0: at most 1024 points) */
        for (int i = 0; i < nl; i++) {
            hipEvent_t e0 = (i == 0) ? ev0 : nullptr, e1 = (i == nl - 1) ? s.t1[ST_A] : nullptr;
            if (seq[i].kind == 0) {
                const int c = seq[i].cls;
                rc |= srla_launch_autocorr(W, kClass[c], &jp, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), d_tw.p,
                                           (uint32_t)seq[i].pass, s.d_results.as<SrlaItemResult>(), s.d_lags.as<double>(), dbg,
                                           s.d_class_index.as<SrlaAutocorrItem>() + job.class_first[c], job.class_count[c], e0, e1, nullptr, nullptr);
            } else {
                rc |= srla_launch_pitch_solve(W, &jp, s.d_lags.as<double>(), s.d_results.as<SrlaItemResult>(), e0, e1, nullptr, 0);
            }
        }
        if (nl == 0) { if (ev0) HIP_OK(hipEventRecord(ev0, W)); HIP_OK(hipEventRecord(s.t1[ST_A], W)); }
        break; }
    case ST_B:
        HIP_OK(hipStreamWaitEvent(N, s.t1[ST_A], 0));
        if (have_items && jp.max_order > 0) {
            rc |= srla_launch_lpc_solve(N, &jp, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), s.d_lags.as<double>(),
                                        s.d_err.as<double>(), d_huff.as<uint8_t>(), s.d_results.as<SrlaItemResult>(), dbg, ev0, s.t1[ST_B]);
        } else { if (ev0) HIP_OK(hipEventRecord(ev0, N)); HIP_OK(hipEventRecord(s.t1[ST_B], N)); }
        break;
    case ST_C:
        HIP_OK(hipStreamWaitEvent(W, s.t1[ST_B], 0));
        if (have_items) {
            const Group &g = job.groups[0];
            /* the roofline kernel: start event on every job */
            rc |= srla_launch_residual_cost(W, g.rclass, &jp, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), &g.plan,
                                            d_thr.as<double>(), s.d_res_ws.as<int32_t>(), s.d_results.as<SrlaItemResult>(),
                                            timing ? s.t0[ST_C] : nullptr, s.t1[ST_C]);
        } else { if (timing) HIP_OK(hipEventRecord(s.t0[ST_C], W)); HIP_OK(hipEventRecord(s.t1[ST_C], W)); }
        break;
    case ST_D:
        HIP_OK(hipStreamWaitEvent(N, s.t1[ST_C], 0));
        if (jp.num_windows) {
            rc |= srla_launch_price(N, &jp, s.d_windows.as<SrlaWindowDesc>(), s.d_cands.as<SrlaCandDesc>(),
                                    s.d_results.as<SrlaItemResult>(), s.d_blocks.as<SrlaBlockRecord>(), ev0, s.t1[ST_D]);
        } else { if (ev0) HIP_OK(hipEventRecord(ev0, N)); HIP_OK(hipEventRecord(s.t1[ST_D], N)); }
        break;
    case ST_E:
        /* block offsets + complete blocks + stream-out to where the stream wants them (the caller's pinned buffer,
         * or this slot's pinned staging buffer); runs on its own stream and leaves W to autocorr / residual_cost */
        HIP_OK(hipStreamWaitEvent(C, s.t1[ST_D], 0));
        if (job.num_slots) {
            rc |= srla_launch_pack(C, &jp, job.num_slots, s.in_cur, s.d_items.as<SrlaItemDesc>(), s.d_windows.as<SrlaWindowDesc>(),
                                   s.d_blocks.as<SrlaBlockRecord>(), s.d_results.as<SrlaItemResult>(), s.d_res_ws.as<int32_t>(),
                                   d_huffcode.as<uint32_t>(), d_huff.as<uint8_t>(), s.d_block_off.as<uint32_t>(),
                                   d_pos.as<uint32_t>(), s.d_ctl.as<uint32_t>(), s.out_first, s.out_init_pos,
                                   s.out_direct ? 1u : 0u, s.out_limit, s.d_stream.as<uint8_t>(), s.out_direct ? s.out_direct : s.h_stream.as<uint8_t>(),
                                   s.d_scratch.as<uint8_t>(), s.h_info.as<SrlaJobInfo>(),
                                   reinterpret_cast<uint32_t *>(s.h_info.as<SrlaJobInfo>() + 1), ev0, s.t1[ST_E], s.out_boost);
        } else { if (ev0) HIP_OK(hipEventRecord(ev0, C)); HIP_OK(hipEventRecord(s.t1[ST_E], C)); }
        break;
    default: return false;
    }
    if (rc != 0) { fprintf(stderr, "[srla-mi355x] kernel launch failed in stage %d\n", st); return false; }
    return true;
}

bool Impl::launch_job(Slot &s, const int32_t *d_in, uint32_t d_stride, const int32_t *const *host_in, bool want_dbg)
{
    in_pinned = false;
    if (!prepare_job(s, d_in, d_stride, host_in, want_dbg)) return false;
    for (int st = 0; st < NUM_ST; st++) if (!run_stage(s, st)) return false;
    return true;
}

bool Impl::wait_job(Slot &s)
{
    HIP_OK(hipEventSynchronize(s.t1[ST_E]));
    float t = 0;
    double *acc[NUM_ST] = { &stats.autocorr_ms, &stats.solve_ms, &stats.residual_ms, &stats.price_ms, &stats.gather_ms };
    for (int st = 0; st < NUM_ST; st++)
        if ((s.timed || (timing && st == ST_C)) && hipEventElapsedTime(&t, s.t0[st], s.t1[st]) == hipSuccess) *acc[st] += t;
    if (s.timed) stats.timed_jobs++;
    if (timeline) {
        /* SRLA_MI355X_TIMELINE (with SRLA_MI355X_TIMING_STRIDE=1): where every stage of every job sat on the device's clock */
        char line[512]; int o = snprintf(line, sizeof(line), "[timeline] job %u+%u:", s.job.s0, s.job.ns);
        static const char *nm[NUM_ST] = { "A", "B", "C", "D", "E" };
        for (int st = 0; st < NUM_ST; st++) {
            float a = -1, b = 0;
            if ((s.timed || st == ST_C) && hipEventElapsedTime(&a, ev_ref, s.t0[st]) != hipSuccess) { a = -1; (void)hipGetLastError(); }
            if (hipEventElapsedTime(&b, ev_ref, s.t1[st]) == hipSuccess)
                o += snprintf(line + o, sizeof(line) - (size_t)o, "  %s %.3f-%.3f", nm[st], a, b);
            else (void)hipGetLastError();
        }
        tl_printf("%s\n", line);
    }
    stats.analyze_ms = stats.autocorr_ms + stats.solve_ms + stats.residual_ms;
    s.busy = false;
    return true;
}

SRLAApiResult Impl::finish_job(Slot &s, uint8_t *data, uint32_t write_off, uint32_t *written, const uint32_t **window_bytes)
{
    const auto t0 = Clock::now();
    const SrlaJobInfo info = *s.h_info.as<SrlaJobInfo>();
#ifdef SRLA_DIAG_STOP
    static const bool diag = getenv("SRLA_MI355X_K3_STOP") != nullptr;   /* timing experiments: the stream is garbage */
    if (diag) { *written = 0; *window_bytes = reinterpret_cast<const uint32_t *>(s.h_info.as<SrlaJobInfo>() + 1); return SRLA_APIRESULT_OK; }
#endif
    if (info.error & SRLA_JOBERR_OVERFLOW) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (info.error != 0 || info.base != write_off) {
        fprintf(stderr, "[srla-mi355x] internal error: device pack reported 0x%x (%s%s), stream offset %u vs %u\n", info.error,
                (info.error & SRLA_JOBERR_SIZE) ? "a packed block differs from its computed size " : "",
                (info.error & SRLA_JOBERR_COVER) ? "a window's blocks do not cover it" : "", info.base, write_off);
        return SRLA_APIRESULT_NG;
    }
    if (!s.out_direct && data != nullptr) {
        const uint8_t *src = s.h_stream.as<uint8_t>();
        const uint32_t chunk = 256u << 10, total = info.total_bytes;
        pool->parallel_for((total + chunk - 1) / chunk, [&](uint32_t i) {
            const uint32_t o = i * chunk;
            memcpy(data + write_off + o, src + o, std::min(chunk, total - o));
        });
    }
    stats.num_blocks += info.num_blocks; stats.num_raw_blocks += info.num_raw; stats.num_silent_blocks += info.num_silent;
    stats.num_tie_items += info.num_tie_items; stats.num_odd_items += info.num_odd_items;
    stats.pack_ms += ms_since(t0);
    *written = info.total_bytes;
    *window_bytes = reinterpret_cast<const uint32_t *>(s.h_info.as<SrlaJobInfo>() + 1);
    return SRLA_APIRESULT_OK;
}

srla::StreamInfo Impl::stream_info(uint32_t num_samples) const
{
    srla::StreamInfo si;
    si.num_channels = par.num_channels; si.bits_per_sample = par.bits_per_sample;
    si.sampling_rate = par.sampling_rate; si.num_samples = num_samples; si.offset_lshift = offset_lshift;
    si.max_block = par.max_num_samples_per_block; si.preset = par.preset; si.ltp_order = par.ltp_order;
    return si;
}

SRLAApiResult Impl::encode_stream(const int32_t *const *host_in, const int32_t *d_in, uint32_t d_stride,
                            uint32_t num_samples, uint8_t *data, uint32_t data_size, uint32_t *output_size,
                            SRLAEncoder_EncodeBlockCallback cb, bool with_header, bool search)
{
    const auto t0 = Clock::now();
    const uint32_t nch = par.num_channels;
    uint32_t write_off = 0;
    if (timeline) (void)hipEventRecord(ev_ref, streams[0]);
    spec_or_active = false;
    in_pinned = false;
    if (host_in && !force_staging) {
        in_pinned = true;
        for (uint32_t ch = 0; ch < nch && in_pinned; ch++) {
            hipPointerAttribute_t at;
            memset(&at, 0, sizeof(at));
            if (hipPointerGetAttributes(&at, host_in[ch]) != hipSuccess || at.type != hipMemoryTypeHost) { in_pinned = false; (void)hipGetLastError(); }
        }
    }
    if (with_header) {
        /* offset left shift: OR of every sample (srla_utility.c:177-203) */
        uint32_t mask = 0;
        spec_or_active = false;
        if (host_in && forced_lshift >= 0) {
            offset_lshift = (uint32_t)forced_lshift;
            mask = offset_lshift ? (1u << offset_lshift) : 1u;       /* reproduces the shift below */
        } else if (host_in && cb == nullptr && !no_speculation) {
            spec_or_active = true;
            spec_guessed = false;
            spec_or.store(0);
            mask = 1u;                                               /* shift 0 until the first job's samples have been seen (prepare_job) */
        } else if (host_in) {
            const uint32_t chunk = 1u << 20, per_ch = (num_samples + chunk - 1) / chunk;
            std::atomic<uint32_t> acc{ 0 };
            pool->parallel_for(per_ch * nch, [&](uint32_t i) {
                const uint32_t ch = i / per_ch, o = (i % per_ch) * chunk, len = std::min(chunk, num_samples - o);
                const int32_t *p = host_in[ch] + o;
                uint32_t m = 0;
                for (uint32_t k = 0; k < len; k++) m |= (uint32_t)p[k];
                acc.fetch_or(m, std::memory_order_relaxed);
            });
            mask = acc.load();
        } else {
            /* on the device, without a host round trip: the jobs read the shift from device memory */
            hipStream_t st = streams[0];
            if (hipMemsetAsync(d_or.p, 0, 8, st) != hipSuccess) return SRLA_APIRESULT_NG;
            if (srla_launch_or_reduce(st, d_in, d_stride, num_samples, nch, d_or.as<uint32_t>()) != 0) return SRLA_APIRESULT_NG;
            if (hipMemcpyAsync(h_or.p, d_or.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess) return SRLA_APIRESULT_NG;
            if (hipEventRecord(ev_or, st) != hipSuccess) return SRLA_APIRESULT_NG;
            lshift_on_device = true;
        }
        if (!lshift_on_device) {
            uint32_t sh = 0;
            if (mask != 0) while (((mask >> sh) & 1u) == 0) sh++;
            offset_lshift = sh;
        }
        write_off = SRLA_HEADER_SIZE;   /* the header itself is written once the shift is known (below) */
    }
    /* can the device store into the caller's buffer (pinned / registered host memory)? */
    uint8_t *out_direct = nullptr;
    bool out_in_hbm = false;
    if (!force_staging) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof(at));
        const hipError_t pe = hipPointerGetAttributes(&at, data);
        if (pe == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer != nullptr)
            out_direct = static_cast<uint8_t *>(at.devicePointer);
        else if (pe == hipSuccess && at.type == hipMemoryTypeDevice) {
            /* the caller wants the stream in device memory: same path, only the header needs a copy */
            out_direct = data;
            out_in_hbm = true;
        } else (void)hipGetLastError();
    }
    if (force_staging && data != nullptr) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof(at));
        if (hipPointerGetAttributes(&at, data) == hipSuccess && at.type == hipMemoryTypeDevice) { out_direct = data; out_in_hbm = true; }
        else (void)hipGetLastError();
    }
    const uint32_t init_pos = write_off;
    const uint32_t window_len = search ? par.num_lookahead_samples : par.max_num_samples_per_block;
    const uint32_t wpj = windows_per_job(search);
    const uint64_t job_len = (uint64_t)wpj * window_len;
    /* Job plan: full jobs rotate through the kSlots buffer sets.  What is left at the end of the stream is cut
     * once more so that the LAST job is small: after it nothing else runs on the wide stream, so its pricing,
     * block assembly and stream-out are pure latency (0.27 ms for a full job, 8 % of a 600 s stream's time).
     * The two tail jobs have buffer sets of their own, so that repeated calls of equal length keep finding
     * their descriptor tables cached. */
    /* an odd-length last window goes through chain mode (see chain_tail) once everything before it is out */
    uint32_t chain_n = 0;
    {
        static const bool no_chain = getenv("SRLA_MI355X_NO_CHAIN") != nullptr;
        const uint32_t tn = num_samples % window_len;
        const uint32_t grid = search ? par.min_num_samples_per_block : par.max_num_samples_per_block;
        if ((tn & 1u) && (grid & 1u) == 0 && (window_len % grid) == 0 && !no_chain) chain_n = tn;
        /* likewise an LTP analysis of a block shorter than the 263 lags reads what earlier calls left beyond its FFT
         * (lpc.c:371-373); with a minimum block above 256 samples only the window's last block can be that short */
        if (tn > 0 && par.ltp_order > 0 && grid > 256u && (window_len % grid) == 0 && ((tn - 1u) % grid) + 1u <= 256u && !no_chain) chain_n = tn;
    }
    const uint32_t body = num_samples - chain_n;
    chain.active = chain_n != 0; chain.begun = false; chain.early = false; chain.ad_done = false;
    chain.tail_start = body; chain.tail_n = chain_n; chain.search = search;
    chain.host_in = host_in; chain.d_in = d_in; chain.d_stride = d_stride;
    struct JobPlan { uint32_t s0, ns, slot; };
    std::vector<JobPlan> plan;
    {
        uint64_t nfull = body / job_len, rest = body - nfull * job_len;
        if (rest == 0 && nfull > 0) { nfull--; rest = job_len; }
        for (uint64_t k = 0; k < nfull; k++) plan.push_back({ (uint32_t)(k * job_len), (uint32_t)job_len, (uint32_t)(k % kSlots) });
        const uint64_t small = (uint64_t)std::max<uint32_t>(1u, 262144u / window_len) * window_len;
        const uint32_t tail0 = (uint32_t)(nfull * job_len);
        if (nfull > 0 && rest > 2 * small) {
            const uint32_t first = (uint32_t)(((rest - small) / window_len) * window_len);
            plan.push_back({ tail0, first, kSlots });
            plan.push_back({ tail0 + first, (uint32_t)(rest - first), kSlots + 1 });
        } else if (rest > 0) {
            plan.push_back({ tail0, (uint32_t)rest, nfull > 0 ? kSlots : 0u });
        }
    }
    const uint32_t njobs = (uint32_t)plan.size();
    uint32_t progress = 0;
    if (timeline) tl_printf("[timeline] stream of %u samples, %u jobs; host %.3f ms into the call\n", num_samples, njobs, ms_since(t0));

    auto fail = [&](SRLAApiResult rc) {
        for (auto &st : streams) if (st) (void)hipStreamSynchronize(st);
        if (upload) (void)hipStreamSynchronize(upload);
        if (chain_stream) (void)hipStreamSynchronize(chain_stream);
        for (auto &sl : slot) sl.busy = false;
        lshift_on_device = false;
        spec_or_active = false;
        return rc;
    };
    auto job_slot = [&](uint32_t k) -> Slot & { return slot[plan[k].slot]; };
    auto begin = [&](uint32_t k) -> bool {
        Slot &s = job_slot(k);
        const uint32_t s0 = plan[k].s0, ns = plan[k].ns;
        build_job(s.job, s0, ns, search);
        s.out_direct = out_direct; s.out_first = (k == 0); s.out_init_pos = init_pos; s.out_limit = data_size;
        s.timed = timing && (k % timing_stride == 0);
        s.out_boost = (k + tail_boost_jobs >= njobs) ? tail_boost : 1u;
        return prepare_job(s, d_in ? d_in + s0 : nullptr, d_stride, host_in, false);
    };
    /* Software pipeline over jobs: iteration t enqueues  autocorr + solve of job t,  residual_cost +
     * pricing of job t-1,  block assembly of job t-2,  then collects job t-3.  Needs 4 buffer sets. */
    const uint32_t depth = 3;
    uint32_t header_done = with_header ? 0 : 1;
    auto write_header = [&]() -> bool {
        if (header_done) return true;
        if (lshift_on_device) {
            if (hipEventSynchronize(ev_or) != hipSuccess) return false;
            offset_lshift = h_or.as<uint32_t>()[1];
        }
        if (out_in_hbm) {
            uint8_t hdr[SRLA_HEADER_SIZE];
            srla::write_stream_header(stream_info(num_samples), hdr);
            if (hipMemcpy(data, hdr, SRLA_HEADER_SIZE, hipMemcpyHostToDevice) != hipSuccess) return false;
        } else {
            srla::write_stream_header(stream_info(num_samples), data);
        }
        header_done = 1;
        return true;
    };
    uint32_t chain_seed_off = 0, chain_seed_n = 0;
    if (chain.active) {
        /* The window's search does not depend on the jobs before it, except through the last block encoded before
         * the window when the window's first history-dependent call can reach back that far: a window of a single
         * candidate (search), or any window when every block is a window of its own.  Without searching that
         * block is known now; otherwise it is read from the last regular job once that has been priced (below). */
        const uint32_t nodes = search ? (chain_n + par.min_num_samples_per_block - 1) / par.min_num_samples_per_block + 1 : 2u;
        if (body == 0 || (search && nodes >= 3)) chain.early = true;
        else if (!search) { chain.early = true; chain_seed_off = body - par.max_num_samples_per_block; chain_seed_n = par.max_num_samples_per_block; }
    }
    static const bool chain_trace = getenv("SRLA_MI355X_CHAIN_TRACE") != nullptr;
    for (uint32_t t = 0; t < njobs + depth; t++) {
        const auto t_enq = Clock::now();
        if (t < njobs) {
            if (!begin(t) || !run_stage(job_slot(t), ST_A) || !run_stage(job_slot(t), ST_B)) return fail(SRLA_APIRESULT_NG);
        }
        if (t >= 1 && t - 1 < njobs) {
            Slot &s = job_slot(t - 1);
            if (!run_stage(s, ST_C) || !run_stage(s, ST_D)) return fail(SRLA_APIRESULT_NG);
        }
        if (t >= 2 && t - 2 < njobs) {
            Slot &s = job_slot(t - 2);
            if (!run_stage(s, ST_E)) return fail(SRLA_APIRESULT_NG);
        }
        /* the host prepares the chain jobs while the device works on the first regular job */
        if (chain.early && !chain.begun) {
            const auto tc = Clock::now();
            if (!chain_begin(chain_seed_off, chain_seed_n)) return fail(SRLA_APIRESULT_NG);
            if (chain_trace) fprintf(stderr, "[chain] begin %.3f ms (%zu calls)\n", ms_since(tc), chain_calls.size());
        }
        if (chain.early && !chain.ad_done && (t == njobs || chain_search_done())) {
            const auto tc = Clock::now();
            if (!chain_encode_ad(out_direct, init_pos, data_size, njobs == 0)) return fail(SRLA_APIRESULT_NG);
            if (chain_trace) fprintf(stderr, "[chain] encode_ad %.3f ms (%zu calls)\n", ms_since(tc), chain_calls.size());
        }
        if (chain.early && t == njobs + 1 && !chain_encode_e()) return fail(SRLA_APIRESULT_NG);
        stats.h2d_ms += ms_since(t_enq);       /* host time spent enqueueing (no H2D of samples on this path) */
        if (timeline) tl_printf("[timeline] host: iteration %u enqueued at %.3f ms\n", t, ms_since(t0));
        if (t < depth) continue;
        const uint32_t k = t - depth;
        Slot &s = job_slot(k);
        if (!wait_job(s)) return fail(SRLA_APIRESULT_NG);
        if (!write_header()) return fail(SRLA_APIRESULT_NG);
        if (timeline) {
            float a = 0;
            if (lshift_on_device && k == 0 && hipEventElapsedTime(&a, ev_ref, ev_or) == hipSuccess) tl_printf("[timeline] offset-shift reduction done at %.3f\n", a);
            tl_printf("[timeline] host: job %u collected at %.3f ms\n", k, ms_since(t0));
        }
        uint32_t wrote = 0;
        const uint32_t *window_bytes = nullptr;
        const SRLAApiResult rc = finish_job(s, data, write_off, &wrote, &window_bytes);
        if (rc != SRLA_APIRESULT_OK) return fail(rc);
        /* callbacks: once per window, in order, pointing into the caller's buffer
         * (srla_encoder.c:1779-1782) */
        uint32_t off = write_off;
        for (size_t w = 0; w < s.job.windows.size(); w++) {
            progress += s.job.windows[w].n;
            if (cb) cb(num_samples, progress, data + off, window_bytes[w]);
            off += window_bytes[w];
        }
        write_off += wrote;
    }
    if (chain.active) {
        if (!write_header()) return fail(SRLA_APIRESULT_NG);
        if (!chain.early) {
            /* the last block encoded before the window: its final call is what the window's only candidate inherits from */
            uint32_t seed_off = 0, seed_n = 0;
            Slot &ls = job_slot(njobs - 1);
            const SrlaWindowDesc &wd = ls.job.windows.back();
            std::vector<SrlaBlockRecord> recs(wd.num_nodes - 1);
            if (hipMemcpy(recs.data(), ls.d_blocks.as<SrlaBlockRecord>() + wd.block_base, recs.size() * sizeof(SrlaBlockRecord), hipMemcpyDeviceToHost) != hipSuccess)
                return fail(SRLA_APIRESULT_NG);
            for (const SrlaBlockRecord &r : recs) if (r.valid) { seed_off = ls.job.s0 + r.sample_off; seed_n = r.n; }
            if (!chain_begin(seed_off, seed_n) || !chain_encode_ad(out_direct, init_pos, data_size, false) || !chain_encode_e())
                return fail(SRLA_APIRESULT_NG);
        }
        Slot &e = slot[kChainSlot + 2];
        const auto tc = Clock::now();
        if (!wait_job(e)) return fail(SRLA_APIRESULT_NG);
        if (chain_trace) fprintf(stderr, "[chain] waited %.3f ms for the encode job\n", ms_since(tc));
        uint32_t wrote = 0;
        const uint32_t *window_bytes = nullptr;
        const SRLAApiResult rc = finish_job(e, data, write_off, &wrote, &window_bytes);
        if (rc != SRLA_APIRESULT_OK) return fail(rc);
        progress += chain_n;
        if (cb) cb(num_samples, progress, data + write_off, wrote);
        write_off += wrote;
        chain.active = false;
    }
    lshift_on_device = false;
    if (spec_or_active) {
        spec_or_active = false;
        const uint32_t m = spec_or.load();
        uint32_t sh = 0;
        if (m != 0) while (((m >> sh) & 1u) == 0) sh++;
        if (sh != offset_lshift) {
            /* the guess was wrong: encode again with the shift that the whole stream has */
            forced_lshift = (int)sh;
            const SRLAApiResult rc = encode_stream(host_in, d_in, d_stride, num_samples, data, data_size, output_size, cb, with_header, search);
            forced_lshift = -1;
            return rc;
        }
    }
    *output_size = write_off;
    stats.total_ms += ms_since(t0);
    if (timeline) { tl_printf("[timeline] call returned at %.3f ms\n", ms_since(t0)); fputs(tl_log.c_str(), stderr); tl_log.clear(); }
    return SRLA_APIRESULT_OK;
}

