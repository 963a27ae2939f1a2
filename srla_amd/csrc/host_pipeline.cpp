/*
 * host_pipeline.cpp -- device set-up, staging of host input, the staged execution of jobs and the stream loop
 * (see host_impl.h for the overall picture).
 */
#include "host_impl.h"

#include <algorithm>
#include <stdlib.h>
#include <string.h>

#include <memory>
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
            s.d_price_ws.release(); s.d_big_sig.release();
            for (hipEvent_t e : { s.ev_a1, s.ev_p0, s.ev_p, s.ev_a0, s.ev_pk, s.ev_dma }) if (e) (void)hipEventDestroy(e);
            s.d_input16.release(); s.d_pcm.release();
            DevBuf *db[] = { &s.d_input, &s.d_items, &s.d_cands, &s.d_windows, &s.d_results, &s.d_res_ws,
                             &s.d_blocks, &s.d_block_off, &s.d_scratch, &s.d_dbg, &s.d_lags, &s.d_err, &s.d_gamma, &s.d_class_index, &s.d_stream,
                             &s.d_segs, &s.d_seg_ctl, &s.d_ties, &s.d_tie_data, &s.d_svr_rows, &s.d_big_scratch, &s.d_big_items, &s.d_coef_ws };
            for (auto *b : db) b->release();
            PinBuf *pb[] = { &s.h_in, &s.h_stream, &s.h_info, &s.h_segs, &s.h_ties };
            for (auto *b : pb) b->release();
        }
        for (auto &st : streams) if (st) (void)hipStreamDestroy(st);
        if (upload) (void)hipStreamDestroy(upload);
        if (dma_stream) { (void)hipStreamSynchronize(dma_stream); (void)hipStreamDestroy(dma_stream); }
        if (chain_stream) { (void)hipStreamSynchronize(chain_stream); (void)hipStreamDestroy(chain_stream); }
        if (ev_or) (void)hipEventDestroy(ev_or);
        if (ev_ref) (void)hipEventDestroy(ev_ref);
        h_or.release();
        d_tw.release(); d_welch.release(); welch_bps = 0; d_geoms.release(); d_thr.release(); d_huff.release(); d_huffcode.release(); d_pos.release(); d_or.release(); d_oracc.release(); d_svr_scratch.release(); d_svr_scratch_chain.release();
        d_chain_pool.release(); d_chain_tab.release(); d_hist.release(); tail.c.smp.release(); drop_pending(); for (Capture *c : spare) { c->smp.release(); delete c; } spare.clear(); for (auto &h : h_chain_up) h.release(); h_chain_recs.release(); h_bounce.release();
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
    read_environment();                                      /* host_tuning.cpp: every environment variable, in one place */
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        fprintf(stderr, "[srla-mi355x] no HIP device available: the MI355X encode path cannot run "
                        "(there is no CPU fallback)\n");
        return false;
    }
    HIP_OK(hipSetDevice(device));
    {
        /* W (critical path) and N (its few wavefronts gate the next wide kernel) run at high priority, the block
         * assembly on C at low priority: measured +2 % over every other assignment */
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const int pr[3] = { hi, hi, lo };
        HIP_OK(hipStreamCreateWithPriority(&streams[0], hipStreamNonBlocking, pr[0]));
        HIP_OK(hipStreamCreateWithPriority(&streams[1], hipStreamNonBlocking, pr[1]));
        HIP_OK(hipStreamCreateWithPriority(&streams[2], hipStreamNonBlocking, pr[2]));
    }
    HIP_OK(hipEventCreate(&ev_or));
    HIP_OK(hipEventCreate(&ev_ref));
    HIP_OK(hipStreamCreateWithFlags(&upload, hipStreamNonBlocking));
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIP_OK(hipStreamCreateWithPriority(&chain_stream, hipStreamNonBlocking, hi));   /* a few workgroups per launch, latency bound */
    }
    /* (created last: the runtime hands out its hardware queues in the order the streams are made, and a stream made before
     * `upload` moved that one onto a queue it shares with a compute stream -- host input -12 %, measured) */
    if (dma_out) HIP_OK(hipStreamCreateWithFlags(&dma_stream, hipStreamNonBlocking));
    if (!h_or.ensure(64)) return false;
    for (uint32_t si = 0; si < kMaxSlots; si++) {
        Slot &s = slot[si];
        for (auto &e : s.t0) HIP_OK(hipEventCreate(&e));
        for (auto &e : s.t1) HIP_OK(hipEventCreate(&e));
        HIP_OK(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        HIP_OK(hipEventCreate(&s.ev_a1)); HIP_OK(hipEventCreate(&s.ev_p0)); HIP_OK(hipEventCreate(&s.ev_p)); HIP_OK(hipEventCreate(&s.ev_a0));
        HIP_OK(hipEventCreateWithFlags(&s.ev_pk, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&s.ev_dma, hipEventDisableTiming));
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
    if (env_pack_threads) nthreads = env_pack_threads;
    pool = new Pool(nthreads);
    dev_failed = false;
    dev_ready = true;
    return true;
}

static uint32_t shift_of(uint32_t mask)
{
    uint32_t sh = 0;
    if (mask != 0) while (((mask >> sh) & 1u) == 0) sh++;
    return sh;
}

/* Brings the samples of the plan's segments to the device.  Device-resident input (one segment) is used where it lies.
 * Host input: pageable planes are staged into pinned memory by the pool threads (non-temporal stores; streams of at most
 * 16 bits are packed to int16 on the way and widened again by a small kernel behind the upload) and cross PCIe on the
 * upload stream; pinned planes are read by DMA directly.  The OR of the samples of streams whose offset shift is still
 * open is gathered on the way. */
bool Impl::stage_input(Slot &s, const JobPlan &plan)
{
    const uint32_t nch = par.num_channels;
    s.used_h2d = false;
    {
        const StreamCtx &first = sx[plan.segs[0].stream];
        if (first.d_in) {
            s.in_cur = first.d_in + plan.segs[0].s0;
            s.stride_cur = first.d_stride;
            return true;
        }
    }
    if (sx[plan.segs[0].stream].pcm) {
        /* interleaved PCM frames in page-locked host memory (every stream of such a call): the frames cross the link as
         * they are (2 bytes per 16-bit sample instead of 4), srla_deinterleave writes the planes, srla_or_reduce gathers
         * the OR for the offset shift -- no host thread touches a sample */
        const uint32_t total = plan.total, B = sx[plan.segs[0].stream].pcm_bytes;
        const size_t frame = (size_t)B * nch;
        if (!s.d_input.ensure((size_t)nch * total * 4) || !s.d_pcm.ensure((size_t)total * frame + 64)) return false;
        for (const SegPlan &sp : plan.segs) {
            StreamCtx &st = sx[sp.stream];
            uint8_t *raw = s.d_pcm.as<uint8_t>() + (size_t)sp.base * frame;
            HIP_OK(hipMemcpyAsync(raw, st.pcm + (size_t)sp.s0 * frame, (size_t)sp.ns * frame, hipMemcpyHostToDevice, upload));
            if (srla_launch_deinterleave(upload, raw, B, nch, sp.ns, s.d_input.as<int32_t>() + sp.base, total) != 0) return false;
            if (!st.lshift_final && st.or_on_device) {
                st.or_dev_end = std::max(st.or_dev_end, sp.s0 + sp.ns);
                if (srla_launch_or_accumulate(upload, s.d_input.as<int32_t>() + sp.base, total, sp.ns, nch, d_oracc.as<uint32_t>() + 2u * sp.stream) != 0) return false;
            }
        }
        HIP_OK(hipEventRecord(s.ev_in, upload));
        s.in_cur = s.d_input.as<int32_t>();
        s.stride_cur = total;
        s.used_h2d = true;
        return true;
    }
    const uint32_t total = plan.total, nseg = (uint32_t)plan.segs.size();
    if (!s.d_input.ensure((size_t)nch * total * 4)) return false;
    bool all_pinned = true;
    for (const SegPlan &sp : plan.segs) all_pinned = all_pinned && sx[sp.stream].in_pinned;
    std::unique_ptr<std::atomic<uint32_t>[]> seg_or(new std::atomic<uint32_t>[nseg]);
    for (uint32_t k = 0; k < nseg; k++) seg_or[k].store(0);
    struct Task { uint32_t seg, ch, off, len; };
    std::vector<Task> tasks;
    /* tasks of 256 Ki samples; a small job (a short stream, a piece of one) in smaller ones, so that every pool thread gets a few:
     * a 10 s stereo stream was four tasks for eight threads (staging 0.1 ms of a 0.44 ms call) */
    uint32_t chunk = 256u << 10;
    while (chunk > (16u << 10) && (uint64_t)total * nch < (uint64_t)chunk * 3u * pool->size()) chunk >>= 1;
    auto list_tasks = [&](bool only_open) {
        tasks.clear();
        for (uint32_t k = 0; k < nseg; k++) {
            const SegPlan &sp = plan.segs[k];
            const StreamCtx &st = sx[sp.stream];
            if (only_open && (st.lshift_final || sp.s0 != st.or_covered)) continue;
            for (uint32_t ch = 0; ch < nch; ch++)
                for (uint32_t o = 0; o < sp.ns; o += chunk) tasks.push_back({ k, ch, o, std::min(chunk, sp.ns - o) });
        }
    };
    if (all_pinned) {
        /* the caller's planes are pinned: DMA straight out of them (the OR of the samples, where it is still being
         * gathered, is computed by the pool threads meanwhile) */
        /* (one stream: the channels' copies spread over two streams / SDMA engines were 15 % slower) */
        /* Planes locked in place because the pool is too small to stage them (one host thread per rank on an 8-GPU node): as int32
         * they are 8 bytes per stereo instant on a link that carries 53 GB/s, and the link sets the pace (5 460 instead of 7 200
         * Msamples/s).  The one thread CAN pack half of them meanwhile (0.43 ms per channel of a 4 Mi-instant job at the
         * 40 GB/s one core reads, while the other channel's 16.7 MB cross the link in 0.31 ms): the first nch / 2 channels of a stream
         * of at most 16 bits go through the int16 staging buffer, the others are read where they lie -- 6 bytes per instant. */
        /* (a pool of four and more threads packs all of them in less time than the link needs for the packed planes: planes pinned by
         * the CALLER, 8 threads: 5 400 - 5 520 as they lie, 6 400 - 6 660 with half of them packed) */
        uint32_t staged_ch = (hybrid_inplace && par.bits_per_sample <= 16 && !no_pack16) ? ((pool->size() >= 4) ? nch : nch / 2u) : 0u;
        auto direct = [&](uint32_t ch0, uint32_t ch1) -> bool {
            for (const SegPlan &sp : plan.segs)
                for (uint32_t ch = ch0; ch < ch1; ch++)
                    HIP_OK(hipMemcpyAsync(s.d_input.as<int32_t>() + (size_t)ch * total + sp.base, sx[sp.stream].host_in[ch] + sp.s0,
                                          (size_t)sp.ns * 4, hipMemcpyHostToDevice, upload));
            return true;
        };
        if (!direct(staged_ch, nch)) return false;
        if (staged_ch != 0) {
            const size_t stride16 = total;
            if (!s.h_in.ensure((size_t)staged_ch * stride16 * 2 + 64u * nch) || !s.d_input16.ensure((size_t)staged_ch * stride16 * 2)) return false;
            tasks.clear();
            for (uint32_t k = 0; k < nseg; k++)
                for (uint32_t ch = 0; ch < staged_ch; ch++)
                    for (uint32_t o = 0; o < plan.segs[k].ns; o += chunk) tasks.push_back({ k, ch, o, std::min(chunk, plan.segs[k].ns - o) });
            int16_t *dst = s.h_in.as<int16_t>();
            std::atomic<uint32_t> wide{ 0 };
            const auto t_pack = Clock::now();
            pool->parallel_for((uint32_t)tasks.size(), [&](uint32_t i) {
                const Task &t = tasks[i];
                const SegPlan &sp = plan.segs[t.seg];
                uint32_t w = 0;
                (void)pack16_or(dst + (size_t)t.ch * stride16 + sp.base + t.off, sx[sp.stream].host_in[t.ch] + sp.s0 + t.off, t.len, &w);
                if (w) wide.fetch_or(w, std::memory_order_relaxed);
            });
            if (timeline) tl_printf("[timeline] host: %zu staging tasks (%u of %u channels) took %.3f ms\n", tasks.size(), staged_ch, nch, ms_since(t_pack));
            if (wide.load() == 0) {
                HIP_OK(hipMemcpyAsync(s.d_input16.p, s.h_in.p, (size_t)staged_ch * stride16 * 2, hipMemcpyHostToDevice, upload));
                if (srla_launch_widen16(upload, s.d_input16.as<int16_t>(), stride16, s.d_input.as<int32_t>(), total, staged_ch) != 0) return false;
                stats.num_hybrid_jobs++;
            } else if (!direct(0, staged_ch)) return false;   /* samples beyond 16 bits in a stream declared narrower: as they lie */
        }
        /* the OR of the samples, where it is still being gathered: on the device, from the uploaded copy */
        for (const SegPlan &sp : plan.segs) {
            const StreamCtx &st = sx[sp.stream];
            if (st.lshift_final || !st.or_on_device) continue;
            sx[sp.stream].or_dev_end = std::max(sx[sp.stream].or_dev_end, sp.s0 + sp.ns);
            if (srla_launch_or_accumulate(upload, s.d_input.as<int32_t>() + sp.base, total, sp.ns, nch, d_oracc.as<uint32_t>() + 2u * sp.stream) != 0) return false;
        }
    } else {
        if (!s.h_in.ensure((size_t)nch * total * 4 + 64u * nch)) return false;
        list_tasks(false);
        bool packed = false;
        if (par.bits_per_sample <= 16 && !no_pack16) {
            /* as int16 (segments start on multiples of 16 samples, so every chunk starts 32-byte aligned) */
            const size_t stride16 = total;
            if (!s.d_input16.ensure(nch * stride16 * 2)) return false;
            int16_t *dst = s.h_in.as<int16_t>();
            std::atomic<uint32_t> wide{ 0 };
            const auto t_pack = Clock::now();
            /* Channel by channel, each plane's upload enqueued as soon as it is packed: the link works on the first plane while the
             * pool packs the second, and the job's samples are on the device half an upload earlier (0.14 ms of the 0.47 ms between a
             * full job's first staged sample and its first kernel; the first three jobs of a call wait for their samples: host -> host
             * 7 290 -> 7 400 Msamples/s). */
            /* (calls of many jobs only: a short stream's one or three jobs lose more to the extra round than the link returns; halves and
             * quarters of a plane: less, profiles/r05/ab_host_path.txt) */
            const uint32_t rounds = (call_crowded && nch >= 2 && (size_t)total * 2 >= (512u << 10)) ? nch : 1u;
            /* the only job of a call: srla_widen16 reads the packed planes out of the page-locked staging buffer itself -- one launch
             * instead of the runtime's copy kernel and the widening behind it (a 10 s stream's 1.9 MB) */
            const bool direct_widen = call_solo;
            std::vector<uint32_t> mine;
            for (uint32_t r = 0; r < rounds && wide.load() == 0; r++) {
                mine.clear();
                for (uint32_t i = 0; i < tasks.size(); i++) if (rounds == 1 || tasks[i].ch == r) mine.push_back(i);
                pool->parallel_for((uint32_t)mine.size(), [&](uint32_t j) {
                    const Task &t = tasks[mine[j]];
                    const SegPlan &sp = plan.segs[t.seg];
                    uint32_t w = 0;
                    const uint32_t m = pack16_or(dst + (size_t)t.ch * stride16 + sp.base + t.off, sx[sp.stream].host_in[t.ch] + sp.s0 + t.off, t.len, &w);
                    seg_or[t.seg].fetch_or(m, std::memory_order_relaxed);
                    if (w) wide.fetch_or(w, std::memory_order_relaxed);
                });
                if (wide.load() != 0) break;
                if (direct_widen) continue;
                if (rounds == 1) HIP_OK(hipMemcpyAsync(s.d_input16.p, s.h_in.p, nch * stride16 * 2, hipMemcpyHostToDevice, upload));
                else HIP_OK(hipMemcpyAsync(s.d_input16.as<int16_t>() + (size_t)r * stride16, dst + (size_t)r * stride16, stride16 * 2, hipMemcpyHostToDevice, upload));
            }
            if (timeline) tl_printf("[timeline] host: %zu staging tasks took %.3f ms\n", tasks.size(), ms_since(t_pack));
            if (wide.load() == 0) {
                if (srla_launch_widen16(upload, direct_widen ? dst : s.d_input16.as<int16_t>(), stride16, s.d_input.as<int32_t>(), total, nch) != 0) return false;
                packed = true;
            }   /* else: samples beyond 16 bits in a stream declared narrower -- the reference does not mind, nor do we */
        }
        if (!packed) {
            int32_t *dst = s.h_in.as<int32_t>();
            pool->parallel_for((uint32_t)tasks.size(), [&](uint32_t i) {
                const Task &t = tasks[i];
                const SegPlan &sp = plan.segs[t.seg];
                /* the copy also gathers the OR of the samples it moves */
                const uint32_t m = copy_or(dst + (size_t)t.ch * total + sp.base + t.off, sx[sp.stream].host_in[t.ch] + sp.s0 + t.off, t.len);
                seg_or[t.seg].fetch_or(m, std::memory_order_relaxed);
            });
            HIP_OK(hipMemcpyAsync(s.d_input.p, s.h_in.p, (size_t)nch * total * 4, hipMemcpyHostToDevice, upload));
        }
    }
    HIP_OK(hipEventRecord(s.ev_in, upload));
    s.in_cur = s.d_input.as<int32_t>();
    s.stride_cur = total;
    s.used_h2d = true;
    for (uint32_t k = 0; k < nseg; k++) {
        const SegPlan &sp = plan.segs[k];
        StreamCtx &st = sx[sp.stream];
        if (!st.lshift_final && !st.or_on_device && sp.s0 == st.or_covered) { st.or_mask |= seg_or[k].load(); st.or_covered += sp.ns; }
        /* a pinned stream in a job that was staged after all (another stream of the job is not pinned) */
        if (!st.lshift_final && st.or_on_device && !all_pinned) { st.or_mask |= seg_or[k].load(); st.or_dev_end = std::max(st.or_dev_end, sp.s0 + sp.ns); }
    }
    return true;
}

void Impl::settle_lshift(const JobPlan &plan, std::vector<uint32_t> &lshift)
{
    lshift.assign(plan.segs.size(), 0u);
    for (size_t k = 0; k < plan.segs.size(); k++) {
        StreamCtx &st = sx[plan.segs[k].stream];
        if (st.lshift_on_device) continue;           /* read from device memory by the kernels (SrlaJobParams::lshift_dev) */
        if (!st.lshift_final && !st.lshift_spec) {
            /* the first job of the stream: final if it holds the whole stream, else the shift of what has been seen so far */
            st.lshift = shift_of(st.or_mask);
            if (st.or_covered >= st.num_samples) st.lshift_final = true;
            else st.lshift_spec = true;
        }
        lshift[k] = st.lshift;
    }
}

bool Impl::prepare_job(Slot &s, bool want_dbg)
{
    Job &job = s.job;
    const uint32_t nch = par.num_channels;
    /* (a piece of a short call runs on a stream of its own: its tables and its cleared tie list go there, ahead of its kernels) */
    hipStream_t W = (s.piece && s.own_stream) ? s.own_stream : streams[0], C = streams[2];
    if (!sync_tables()) return false;
    const size_t n_items = job.items.size(), n_cands = job.cands.size(), n_win = job.windows.size(), nseg = job.segs.size();
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
    if (!s.h_info.ensure(sizeof(SrlaJobInfo) + n_win * 4 + nseg * sizeof(SrlaSegInfo))) return false;
    if (!s.h_ties.ensure((size_t)SRLA_TIE_GATHER_CAP * (1u + std::max<uint32_t>(preset_order() + 2u, 8u)) * sizeof(double))) return false;
    if (!s.d_segs.ensure(nseg * sizeof(SrlaSegDesc)) || !s.h_segs.ensure(nseg * sizeof(SrlaSegDesc))) return false;
    if (!s.d_seg_ctl.ensure(nseg * SRLA_SEGCTL_WORDS_HOST * 4)) return false;
    if (!s.d_ties.ensure((3 * std::max<size_t>(1, n_items) + 2) * 4)) return false;   /* an order, an LTP and an SVR entry per item at most */
    if (par.ltp_order > 0 && !s.d_tie_data.ensure(std::max<size_t>(1, n_items) * 8 * sizeof(double))) return false;
    {
        /* a block is never larger than its raw form (11 + n * nch * bytes): bound of the job's bytes (+ the slack between segments) */
        uint64_t samples = 0;
        bool need_host_stage = false;
        for (const SegPlan &sp : job.segs) { samples += sp.ns; need_host_stage = need_host_stage || sx[sp.stream].out_direct == nullptr; }
        const size_t bound = (size_t)samples * nch * (par.bits_per_sample / 8) + (size_t)job.num_slots * SRLA_PACK_SLACK + 64 + 32 * nseg;
        if (need_host_stage && !s.h_stream.ensure(bound)) return false;
        if (!s.d_stream.ensure(bound + 32)) return false;
        const SrlaJobParams probe = job_params(job, s.stride_cur, false);
        if (srla_pack_needs_scratch(&probe) && !s.d_scratch.ensure(bound)) return false;
    }
    if (!job.big_items.empty()) {
        const uint32_t nfft_max = job.big_max_n > 32768u ? 65536u : (job.big_max_n > 16384u ? 32768u : 16384u);
        /* (two transform buffers per workgroup; for 65536 points also the signal, which no longer fits LDS: srla_launch_autocorr_big) */
        if (!s.d_big_scratch.ensure((size_t)SRLA_BIG_GROUPS * nfft_max * (16u + (nfft_max > 32768u ? 4u : 0u)))) return false;
        if (job.big_max_n > 32768u && !s.d_big_sig.ensure(job.big_items.size() * (size_t)srla_residual_big_sig_words(job.big_max_n) * 4u)) return false;
        const void *pb = s.d_big_items.p;
        if (!s.d_big_items.ensure(job.big_items.size() * 4)) return false;
        if (pb != s.d_big_items.p) job.uploaded = false;
    }
    if (par.num_svr_filter_learning_iteration > 0) {
        const uint32_t max_order = preset_order();
        if (!s.d_coef_ws.ensure(std::max<size_t>(1, n_items) * (max_order <= 64 ? 64 : 256) * sizeof(double))) return false;
        if ((max_order > 64 || par.max_num_samples_per_block > 8192u) &&
            (!d_svr_scratch.ensure((size_t)kSvrGroups * srla_svr_big_scratch_bytes(par.max_num_samples_per_block)) ||
             !d_svr_scratch_chain.ensure((size_t)kSvrGroups * srla_svr_big_scratch_bytes(par.max_num_samples_per_block)))) return false;
    }
    if (want_dbg && !s.d_dbg.ensure(std::max<size_t>(1, n_items) * SRLA_DBG_STRIDE * sizeof(double))) return false;
    const uint32_t lag_rows = std::max<uint32_t>(par.ltp_order > 0 ? SRLA_LTP_LAGS : 0u, preset_order() + 1);
    if (!s.d_lags.ensure((size_t)lag_rows * std::max<size_t>(1, n_items) * sizeof(double))) return false;
    if (!s.d_err.ensure((size_t)(preset_order() + 1) * std::max<size_t>(1, n_items) * sizeof(double))) return false;
    if (!s.d_gamma.ensure((size_t)(preset_order() + 1) * std::max<size_t>(1, n_items) * sizeof(double))) return false;
    s.want_dbg = want_dbg;
    if (!job.uploaded) {
        if (n_items) HIP_OK(hipMemcpyAsync(s.d_items.p, job.items.data(), n_items * sizeof(SrlaItemDesc), hipMemcpyHostToDevice, W));
        if (n_items) HIP_OK(hipMemcpyAsync(s.d_class_index.p, job.class_index.data(), n_items * sizeof(SrlaAutocorrItem), hipMemcpyHostToDevice, W));
        HIP_OK(hipMemcpyAsync(s.d_cands.p, job.cands.data(), n_cands * sizeof(SrlaCandDesc), hipMemcpyHostToDevice, W));
        HIP_OK(hipMemcpyAsync(s.d_windows.p, job.windows.data(), n_win * sizeof(SrlaWindowDesc), hipMemcpyHostToDevice, W));
        if (!job.big_items.empty()) HIP_OK(hipMemcpyAsync(s.d_big_items.p, job.big_items.data(), job.big_items.size() * 4, hipMemcpyHostToDevice, W));
        job.uploaded = true;
    }
    HIP_OK(hipMemsetAsync(s.d_ties.p, 0, 4, W));
    if (!job.svr_rows.empty()) {
        /* predictors the host refined with its own libm (rare: a plain copy) */
        if (!s.d_svr_rows.ensure(job.svr_rows.size() * sizeof(double))) return false;
        HIP_OK(hipMemcpy(s.d_svr_rows.p, job.svr_rows.data(), job.svr_rows.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    /* the segment table: where the blocks of each stream part go */
    {
        SrlaSegDesc *sd = s.h_segs.as<SrlaSegDesc>();
        for (size_t k = 0; k < nseg; k++) {
            StreamCtx &st = sx[job.segs[k].stream];
            sd[k].first_window = job.seg_first_window[k];
            sd[k].num_windows = job.seg_first_window[k + 1] - job.seg_first_window[k];
            sd[k].stream = job.segs[k].stream;
            sd[k].use_init = st.pass_started ? 0u : 1u;
            sd[k].init_pos = st.write_off;
            sd[k].limit = st.data ? st.data_size : 0xFFFFFFFFu;
            sd[k].dst = (uint64_t)reinterpret_cast<uintptr_t>(st.out_direct);
            if (s.emits) st.pass_started = true;
        }
        /* (on the stream the block assembly runs on: C, or the one stream of a call's only job) */
        HIP_OK(hipMemcpyAsync(s.d_segs.p, s.h_segs.p, nseg * sizeof(SrlaSegDesc), hipMemcpyHostToDevice, (s.solo && s.own_stream) ? s.own_stream : C));
    }
    s.jp = job_params(job, s.stride_cur, sx[job.segs[0].stream].lshift_on_device);
    s.busy = true; s.b_done = false;
    stats.num_windows += n_win; stats.num_candidates += n_cands; stats.num_items += n_items;
    stats.analyzed_samples += job.analyzed_samples;
    stats.analyze_launches++;
    return true;
}

bool Impl::run_stage(Slot &s, int st, int part)
{
    Job &job = s.job;
    /* chain-mode jobs keep to their own stream up to the pricing, so that the regular jobs never queue behind their
     * many small dependent launches; the block assembly stays on C, where the order of a stream's blocks is made */
    hipStream_t W = s.own_stream ? s.own_stream : streams[0], N = s.own_stream ? s.own_stream : streams[1], C = streams[2];
    const SrlaJobParams &jp = s.jp;
    double *dbg = s.want_dbg ? s.d_dbg.as<double>() : nullptr;
    const bool have_items = !job.groups.empty();
    const bool on_device = jp.lshift_dev != nullptr;
    int rc = 0;
    /* Stage events ride on the kernel dispatches themselves (hipExtLaunchKernel): the end event on the stage's last
     * launch, the start event (timed jobs only) on its first -- no separate marker packets between the kernels
     * of the critical stream.  A stage without launches records its end event the ordinary way. */
    hipEvent_t ev0 = s.timed ? s.t0[st] : nullptr;
    /* The only job of a call (Slot::solo) runs every stage on ONE stream: its stages need no end events to wait for one another --
     * an event on a launch costs about 3 us, a wait on it as much again, of a 10 s call's 0.34 ms -- except where the job is timed. */
    const bool lean = (s.solo || s.piece) && !s.timed;
    const bool lean_d = lean && s.solo;                          /* (a piece's block assembly runs on stream C: the pricing keeps its end event) */
    switch (st) {
    case ST_A: {
        s.ties_gathered = false;
        if (part != 2) {
            if (on_device) HIP_OK(hipStreamWaitEvent(W, ev_or, 0));
            if (s.used_h2d) HIP_OK(hipStreamWaitEvent(W, s.ev_in, 0));
        }
        const SrlaJobParams &jv = jp;
        struct L { int kind, cls, pass; };                       /* kind 0: autocorr class launch, 1: pitch solve */
        L seq[24]; int nl = 0;
        if (have_items) {
            for (int pass = (par.ltp_order > 0) ? 1 : 0; pass >= 0; pass--) {
                for (int c = 0; c < 8; c++) if (job.class_count[c]) seq[nl++] = { 0, c, pass };
                if (pass == 1) seq[nl++] = { 1, 0, 1 };
            }
        }
        /* the two-part form only for what it is made for: items, two passes */
        const bool split = part != 0 && have_items && par.ltp_order > 0;
        if (part == 2 && !split) break;                          /* part 1 did the whole stage */
        int first = 0, last = nl - 1, pitch_at = -1;
        for (int i = 0; i < nl; i++) if (seq[i].kind == 1) pitch_at = i;
        if (split && part == 1) last = pitch_at;
        if (split && part == 2) { first = pitch_at + 1; HIP_OK(hipStreamWaitEvent(W, s.ev_p, 0)); }
        static const int kClass[8] = { 0, 1, 2, 4, 0, 0, 0, 0 };     /* FFT size / 2048 (0: at most 1024 points) */
        static const uint32_t kBigFft[8] = { 0u, 0u, 0u, 0u, 16384u, 32768u, 0u, 65536u };   /* classes of srla_autocorr_big */
        auto start_event = [&](int i) -> hipEvent_t {
            hipEvent_t e = (i == 0) ? ev0 : nullptr;
            if (split) { if (i == pitch_at) e = s.ev_p0; if (i == pitch_at + 1 && s.timed) e = s.ev_a0; }
            return e;
        };
        auto stop_event = [&](int i) -> hipEvent_t {
            hipEvent_t e = (i == nl - 1 && !lean) ? s.t1[ST_A] : nullptr;
            if (split) { if (i == pitch_at - 1) e = s.ev_a1; /* the last launch of the LTP pass */ if (i == pitch_at) e = s.ev_p; }
            return e;
        };
        /* a small job's 2048- and 4096-point classes in ONE launch (srla_autocorr_pair): a short stream is a latency chain */
        const bool pair_ok = job.class_count[1] != 0 && job.class_count[2] != 0 && job.class_count[1] + job.class_count[2] <= kPairMaxItems;
        for (int i = first; i <= last; i++) {
            hipEvent_t e0 = start_event(i), e1 = stop_event(i);
            if (pair_ok && seq[i].kind == 0 && seq[i].cls == 1 && i + 1 <= last && seq[i + 1].kind == 0 && seq[i + 1].cls == 2 && seq[i + 1].pass == seq[i].pass) {
                rc |= srla_launch_autocorr_pair(W, &jv, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), d_tw.p, (uint32_t)seq[i].pass,
                                                s.d_results.as<SrlaItemResult>(), s.d_lags.as<double>(), dbg,
                                                s.d_class_index.as<SrlaAutocorrItem>() + job.class_first[2], job.class_count[2],
                                                s.d_class_index.as<SrlaAutocorrItem>() + job.class_first[1], job.class_count[1], e0, stop_event(i + 1));
                i++;
                continue;
            }
            if (seq[i].kind == 0) {
                const int c = seq[i].cls;
                if (kBigFft[c])
                    rc |= srla_launch_autocorr_big(W, &jp, s.in_cur, d_tw.p, (uint32_t)seq[i].pass, s.d_results.as<SrlaItemResult>(), s.d_lags.as<double>(), dbg,
                                                   s.d_class_index.as<SrlaAutocorrItem>() + job.class_first[c], job.class_count[c], kBigFft[c],
                                                   e0, e1, nullptr, nullptr, s.d_big_scratch.p, SRLA_BIG_GROUPS);
                else
                rc |= srla_launch_autocorr(W, kClass[c], &jv, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), d_tw.p,
                                           (uint32_t)seq[i].pass, s.d_results.as<SrlaItemResult>(), s.d_lags.as<double>(), dbg,
                                           s.d_class_index.as<SrlaAutocorrItem>() + job.class_first[c], job.class_count[c], e0, e1, nullptr, nullptr, c == 6 ? 1 : 0);
            } else {
                hipStream_t ps = W;
                if (split) { ps = N; HIP_OK(hipStreamWaitEvent(N, s.ev_a1, 0)); }
                rc |= srla_launch_pitch_solve(ps, &jp, s.d_items.as<SrlaItemDesc>(), s.d_lags.as<double>(), s.d_results.as<SrlaItemResult>(), e0, e1,
                                              nullptr, 0, s.d_ties.as<uint32_t>(), s.d_tie_data.as<double>());
            }
        }
        s.split_a = split;
        if (nl == 0 && !lean) { if (ev0) HIP_OK(hipEventRecord(ev0, W)); HIP_OK(hipEventRecord(s.t1[ST_A], W)); }
        break; }
    case ST_B:
        if (!lean) HIP_OK(hipStreamWaitEvent(N, s.t1[ST_A], 0));
        if (lean && !(have_items && jp.max_order > 0)) break;
        if (s.b_done) {
            /* chain mode with SVR on: the solve chain ran round by round inside stage A (host_chain.cpp) */
            if (ev0) HIP_OK(hipEventRecord(ev0, N));
            HIP_OK(hipEventRecord(s.t1[ST_B], N));
        } else if (have_items && jp.max_order > 0) {
            const SrlaSvrExtra ex = { s.d_ties.as<uint32_t>(), job.svr_rows.empty() ? nullptr : s.d_svr_rows.as<double>() };
            rc |= srla_launch_lpc_solve(N, &jp, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), s.d_lags.as<double>(),
                                        s.d_err.as<double>(), d_huff.as<uint8_t>(), s.d_results.as<SrlaItemResult>(), dbg,
                                        s.d_ties.as<uint32_t>(), ev0, lean ? nullptr : s.t1[ST_B], s.in_cur, s.d_coef_ws.as<double>(),
                                        par.num_svr_filter_learning_iteration, std::min<uint32_t>(par.max_num_samples_per_block, 8192u),
                                        d_svr_scratch.p, kSvrGroups, s.d_gamma.as<double>(), &ex);
        } else { if (ev0) HIP_OK(hipEventRecord(ev0, N)); HIP_OK(hipEventRecord(s.t1[ST_B], N)); }
        break;
    case ST_C:
        if (!lean) HIP_OK(hipStreamWaitEvent(W, s.t1[ST_B], 0));
        if (have_items) {
            hipEvent_t c1 = lean ? nullptr : s.t1[ST_C];
            const Group &g = job.groups[0];
            /* the roofline kernel: start event on every job */
            const bool big = !job.big_items.empty();
            const SrlaJobParams &jv = jp;
            if (g.split) {
                /* the large items first: they are what the launch waits for, the small ones fill in behind them */
                SrlaJobParams jl = jv, js = jv;
                jl.rc_lo = 4096u; jl.rc_hi = 8192u;
                js.rc_lo = 0u; js.rc_hi = 4096u;
                rc |= srla_launch_residual_cost(W, 4, &jl, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), &g.plan,
                                                d_thr.as<double>(), s.d_res_ws.as<int32_t>(), s.d_results.as<SrlaItemResult>(),
                                                s.c_start ? s.t0[ST_C] : nullptr, nullptr);
                rc |= srla_launch_residual_cost(W, 2, &js, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), &g.plan_small,
                                                d_thr.as<double>(), s.d_res_ws.as<int32_t>(), s.d_results.as<SrlaItemResult>(),
                                                nullptr, big ? nullptr : c1);
            } else
            rc |= srla_launch_residual_cost(W, g.rclass, &jv, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), &g.plan,
                                            d_thr.as<double>(), s.d_res_ws.as<int32_t>(), s.d_results.as<SrlaItemResult>(),
                                            s.c_start ? s.t0[ST_C] : nullptr, big ? nullptr : c1);
            if (big)
                rc |= srla_launch_residual_cost_big(W, &jp, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), d_thr.as<double>(),
                                                    s.d_res_ws.as<int32_t>(), s.d_results.as<SrlaItemResult>(), s.d_big_items.as<uint32_t>(),
                                                    (uint32_t)job.big_items.size(), job.big_max_n, nullptr, c1,
                                                    job.big_max_n > 32768u ? s.d_big_sig.as<int32_t>() : nullptr);
        } else if (!lean) { if (s.c_start) HIP_OK(hipEventRecord(s.t0[ST_C], W)); HIP_OK(hipEventRecord(s.t1[ST_C], W)); }
        break;
    case ST_D:
        if (!lean) HIP_OK(hipStreamWaitEvent(N, s.t1[ST_C], 0));
        if (jp.num_windows) {
            uint32_t *price_ws = nullptr;
            if (job.max_window_cands > srla_price_lds_cands()) {
                if (!s.d_price_ws.ensure(std::max<size_t>(16, job.cands.size() * 8))) return false;
                price_ws = s.d_price_ws.as<uint32_t>();
            }
            rc |= srla_launch_price(N, &jp, s.d_windows.as<SrlaWindowDesc>(), s.d_cands.as<SrlaCandDesc>(),
                                    s.d_results.as<SrlaItemResult>(), s.d_blocks.as<SrlaBlockRecord>(), ev0, lean_d ? nullptr : s.t1[ST_D],
                                    job.max_nodes, job.max_window_cands, price_ws);
        } else if (!lean_d) { if (ev0) HIP_OK(hipEventRecord(ev0, N)); HIP_OK(hipEventRecord(s.t1[ST_D], N)); }
        break;
    case ST_E:
        /* block offsets + complete blocks + stream-out to where the streams want them (their pinned buffers, or this
         * slot's pinned staging buffer); runs on its own stream and leaves W to autocorr / residual_cost */
        /* The offsets and the assembly right behind the pricing on N (high priority: on the low-priority stream they waited
         * for the tails of the wide kernels, 0.13-0.3 ms per job for 0.05 ms of work), the stream-out on C: the copy of job k
         * then also runs beside the assembly of job k + 1.  A job on a stream of its own keeps to it. */
        {
        hipStream_t P = s.solo ? s.own_stream : C;       /* (the only job of a call: no other job's blocks to keep in order with) */
        if (P == C) HIP_OK(hipStreamWaitEvent(C, s.t1[ST_D], 0));
        s.use_dma = call_dma && !s.own_stream && s.emits && !s.last_job;   /* (the call's last job: the copy-out kernel follows its assembly without a host round trip) */
        if (s.dma_pending) { HIP_OK(hipStreamWaitEvent(P, s.ev_dma, 0)); s.dma_pending = false; }   /* the staging buffer's last job has left it */
        if (job.num_slots) {
            SrlaJobInfo *info = s.h_info.as<SrlaJobInfo>();
            uint32_t *wbytes = reinterpret_cast<uint32_t *>(info + 1);
            const SrlaTieGather tg = { s.d_err.as<double>(), par.ltp_order > 0 ? s.d_tie_data.as<double>() : nullptr, s.h_ties.as<double>(),
                                       SRLA_TIE_GATHER_CAP, (uint32_t)job.items.size() };
            rc |= srla_launch_pack(P, &jp, job.num_slots, s.in_cur, s.d_items.as<SrlaItemDesc>(), s.d_windows.as<SrlaWindowDesc>(),
                                   s.d_blocks.as<SrlaBlockRecord>(), s.d_results.as<SrlaItemResult>(), s.d_res_ws.as<int32_t>(),
                                   d_huffcode.as<uint32_t>(), d_huff.as<uint8_t>(), s.d_block_off.as<uint32_t>(),
                                   d_pos.as<uint32_t>(), s.d_segs.as<SrlaSegDesc>(), s.d_seg_ctl.as<uint32_t>(),
                                   s.d_stream.as<uint8_t>(), s.h_stream.as<uint8_t>(), s.d_scratch.as<uint8_t>(), info, wbytes,
                                   reinterpret_cast<SrlaSegInfo *>(wbytes + job.windows.size()), s.d_ties.as<uint32_t>(),
                                   ev0, s.t1[ST_E], s.out_boost, nullptr, s.ev_pk,
                                   s.use_dma ? 1u : ((direct_tail && s.last_job && s.emits && !s.merge_cb) ? 2u : 0u), &tg);
            s.ties_gathered = true;
        } else { if (ev0) HIP_OK(hipEventRecord(ev0, P)); HIP_OK(hipEventRecord(s.t1[ST_E], P)); }
        }
        break;
    default: return false;
    }
    if (rc != 0) { fprintf(stderr, "[srla-mi355x] kernel launch failed in stage %d\n", st); return false; }
    return true;
}

bool Impl::wait_job(Slot &s)
{
    if (spin_collect) {
        /* a call of a few jobs is a latency chain: poll the event (a load of its signal) for a while instead of sleeping on it */
        const auto t0 = Clock::now();
        hipError_t q;
        while ((q = hipEventQuery(s.t1[ST_E])) == hipErrorNotReady && ms_since(t0) < 2.0) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        if (q != hipSuccess) (void)hipGetLastError();
    }
    HIP_OK(hipEventSynchronize(s.t1[ST_E]));
    float t = 0;
    double *acc[NUM_ST] = { &stats.autocorr_ms, &stats.solve_ms, &stats.residual_ms, &stats.price_ms, &stats.gather_ms };
    for (int st = 0; st < NUM_ST; st++) {
        if (st == ST_A && s.split_a) {
            /* stage A in two parts: the two autocorrelation passes (other jobs' kernels ran on W in between), the pitch solve on N */
            if (!s.timed) continue;
            if (hipEventElapsedTime(&t, s.t0[ST_A], s.ev_a1) == hipSuccess) stats.autocorr_ms += t;
            if (hipEventElapsedTime(&t, s.ev_a0, s.t1[ST_A]) == hipSuccess) stats.autocorr_ms += t;
            if (hipEventElapsedTime(&t, s.ev_p0, s.ev_p) == hipSuccess) stats.pitch_ms += t;
            continue;
        }
        if (s.timed && hipEventElapsedTime(&t, s.t0[st], s.t1[st]) == hipSuccess) *acc[st] += t;
    }
    if (s.timed) stats.timed_jobs++;
    if (timeline) {
        /* SRLA_MI355X_TIMELINE (with SRLA_MI355X_TIMING_STRIDE=1): where every stage of every job sat on the device's clock */
        char line[512]; int o = snprintf(line, sizeof(line), "[timeline] job of %u samples:", s.job.total);
        static const char *nm[NUM_ST] = { "A", "B", "C", "D", "E" };
        for (int st = 0; st < NUM_ST; st++) {
            float a = -1, b = 0;
            if (s.timed && hipEventElapsedTime(&a, ev_ref, s.t0[st]) != hipSuccess) { a = -1; (void)hipGetLastError(); }
            if (hipEventElapsedTime(&b, ev_ref, s.t1[st]) == hipSuccess)
                o += snprintf(line + o, sizeof(line) - (size_t)o, "  %s %.3f-%.3f", nm[st], a, b);
            else (void)hipGetLastError();
        }
        tl_printf("%s\n", line);
    }
    stats.analyze_ms = stats.autocorr_ms + stats.pitch_ms + stats.solve_ms + stats.residual_ms;
    s.busy = false;
    return true;
}

bool Impl::run_job_sync(Slot &s, const JobPlan &plan, bool search, bool want_dbg, uint32_t jobkey)
{
    for (int attempt = 0; attempt < 6; attempt++) {
        call_solo = false;
        if (!stage_input(s, plan)) return false;
        std::vector<uint32_t> lsh;
        settle_lshift(plan, lsh);
        build_job(s.job, plan, lsh, search);
        if (apply_overrides(s.job, jobkey)) { s.job.uploaded = false; s.job.key = 0; }
        s.own_stream = nullptr; s.solo = false; s.piece = false; s.timed = timing; s.c_start = timing; s.out_boost = 1; s.last_job = true; call_crowded = false;   /* (no DMA output: the copy-out kernel) */
        for (const SegPlan &sp : plan.segs) sx[sp.stream].pass_started = false;
        if (!prepare_job(s, want_dbg)) return false;
        for (int st = 0; st < NUM_ST; st++) if (!run_stage(s, st)) return false;
        if (!wait_job(s)) return false;
        const SrlaJobInfo *info = s.h_info.as<SrlaJobInfo>();
        if (info->num_tie_items == 0) return true;
        const int m = arbitrate(s, jobkey);
        if (m < 0) return false;
        if (m == 0) return true;
        stats.num_restarts++;
    }
    fprintf(stderr, "[srla-mi355x] internal error: near-tie arbitration did not settle\n");
    return false;
}

SRLAApiResult Impl::finish_job(Slot &s)
{
    const auto t0 = Clock::now();
    const SrlaJobInfo info = *s.h_info.as<SrlaJobInfo>();
    if (info.error & ~SRLA_JOBERR_OVERFLOW) {
        fprintf(stderr, "[srla-mi355x] internal error: device pack reported 0x%x (%s%s)\n", info.error,
                (info.error & SRLA_JOBERR_SIZE) ? "a packed block differs from its computed size " : "",
                (info.error & SRLA_JOBERR_COVER) ? "a window's blocks do not cover it" : "");
        return SRLA_APIRESULT_NG;
    }
    /* every way out once a copy has been enqueued: ev_dma marks the end of the copies, or the next job in this slot would wait
     * on a stale (or never recorded) event instead of on them */
    auto leave = [&](SRLAApiResult rc) {
        if (s.dma_pending && hipEventRecord(s.ev_dma, dma_stream) != hipSuccess) { (void)hipGetLastError(); s.dma_pending = false; (void)hipStreamSynchronize(dma_stream); }
        return rc;
    };
    const SrlaSegInfo *si = s.seg_info();
    const uint32_t *wb = s.window_bytes();
    SRLAApiResult worst = SRLA_APIRESULT_OK;
    for (size_t k = 0; k < s.job.segs.size(); k++) {
        const SegPlan &sp = s.job.segs[k];
        StreamCtx &st = sx[sp.stream];
        if (si[k].skip) { st.rc = SRLA_APIRESULT_INSUFFICIENT_BUFFER; worst = st.rc; continue; }
        if (si[k].pos != st.write_off) {
            fprintf(stderr, "[srla-mi355x] internal error: stream %u continues at %u, the host expected %u\n", sp.stream, si[k].pos, st.write_off);
            return leave(SRLA_APIRESULT_NG);
        }
        if (s.use_dma && si[k].bytes != 0) {
            /* (the segment stands in the job's staging buffer at stage_off, srla_block_offsets) */
            if (hipMemcpyAsync(st.data + si[k].pos, s.d_stream.as<uint8_t>() + si[k].stage_off, si[k].bytes, hipMemcpyDefault, dma_stream) != hipSuccess)
                return leave(SRLA_APIRESULT_NG);
            dma_used = true; s.dma_pending = true;
        }
        if (!st.out_direct && st.data != nullptr) {
            const uint8_t *src = s.h_stream.as<uint8_t>() + si[k].stage_off;
            uint8_t *dst = st.data + si[k].pos;
            const uint32_t total = si[k].bytes;
            uint32_t chunk = 256u << 10;
            while (chunk > (16u << 10) && total < chunk * 3u * pool->size()) chunk >>= 1;      /* (a short stream's bytes: a few tasks per thread) */
            pool->parallel_for((total + chunk - 1) / chunk, [&](uint32_t i) {
                const uint32_t o = i * chunk;
                memcpy(dst + o, src + o, std::min(chunk, total - o));
            });
        }
        /* callbacks: once per window, in order, pointing into the caller's buffer (srla_encoder.c:1779-1782) */
        uint32_t off = st.write_off, progress = st.progress;
        if (s.merge_cb) {
            progress += sp.ns;
            if (st.cb) st.cb(st.num_samples, progress, st.data + off, si[k].bytes);
        } else {
            for (uint32_t w = s.job.seg_first_window[k]; w < s.job.seg_first_window[k + 1]; w++) {
                progress += s.job.windows[w].n;
                if (st.cb) st.cb(st.num_samples, progress, st.data + off, wb[w]);
                off += wb[w];
            }
        }
        st.write_off += si[k].bytes;
        st.progress = progress;
    }
    if (s.dma_pending && hipEventRecord(s.ev_dma, dma_stream) != hipSuccess) { s.dma_pending = false; (void)hipStreamSynchronize(dma_stream); return SRLA_APIRESULT_NG; }
    stats.num_blocks += info.num_blocks; stats.num_raw_blocks += info.num_raw; stats.num_silent_blocks += info.num_silent;
    stats.num_tie_items += info.num_tie_items; stats.num_odd_items += info.num_odd_items;
    stats.pack_ms += ms_since(t0);
    return worst;
}

srla::StreamInfo Impl::stream_info(const StreamCtx &st) const
{
    srla::StreamInfo si;
    si.num_channels = par.num_channels; si.bits_per_sample = par.bits_per_sample;
    si.sampling_rate = par.sampling_rate; si.num_samples = st.num_samples; si.offset_lshift = st.lshift;
    si.max_block = par.max_num_samples_per_block; si.preset = par.preset; si.ltp_order = par.ltp_order;
    return si;
}

bool Impl::write_header(StreamCtx &st)
{
    if (!st.with_header || st.data == nullptr) return true;
    if (st.lshift_on_device) {
        if (hipEventSynchronize(ev_or) != hipSuccess) return false;
        st.lshift = h_or.as<uint32_t>()[1];
    }
    if (st.out_in_hbm) {
        uint8_t hdr[SRLA_HEADER_SIZE];
        srla::write_stream_header(stream_info(st), hdr);
        if (hipMemcpy(st.data, hdr, SRLA_HEADER_SIZE, hipMemcpyHostToDevice) != hipSuccess) return false;
    } else {
        srla::write_stream_header(stream_info(st), st.data);
    }
    return true;
}

void Impl::classify_buffers(StreamCtx &st, std::vector<const void *> &held)
{
    /* memory that another handle's call locked in place (the process-wide registry of host_support.cpp) stays locked only as
     * long as somebody holds a reference: take one for this call before enqueueing DMA on it */
    auto keep = [&](const void *p) { if (const void *key = host_pin_addref(p)) held.push_back(key); };
    st.in_pinned = false;
    if (st.host_in && !force_staging) {
        st.in_pinned = true;
        const size_t before = held.size();
        for (uint32_t ch = 0; ch < par.num_channels && st.in_pinned; ch++) {
            hipPointerAttribute_t at;
            memset(&at, 0, sizeof(at));
            if (hipPointerGetAttributes(&at, st.host_in[ch]) != hipSuccess || at.type != hipMemoryTypeHost) { st.in_pinned = false; (void)hipGetLastError(); }
            else keep(st.host_in[ch]);
        }
        if (!st.in_pinned) while (held.size() > before) { host_pin_release(held.back()); held.pop_back(); }
    }
    /* can the device store into the stream's buffer (pinned / registered host memory, or device memory)? */
    st.out_direct = nullptr;
    st.out_in_hbm = false;
    if (st.data != nullptr) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof(at));
        const hipError_t pe = hipPointerGetAttributes(&at, st.data);
        if (pe == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer != nullptr && !force_staging) {
            st.out_direct = static_cast<uint8_t *>(at.devicePointer);
            keep(st.data);
        } else if (pe == hipSuccess && at.type == hipMemoryTypeDevice) {
            /* the caller wants the stream in device memory: same path, only the header needs a copy */
            st.out_direct = st.data;
            st.out_in_hbm = true;
        } else if (pe != hipSuccess) (void)hipGetLastError();
    }
}

/* The history-dependent last window of a stream of num_samples samples (0: none) outside the history regimes: an odd-length
 * one (lpc.c:260-264), or -- LTP on -- one whose last block is shorter than the 263 lags (lpc.c:371-373; with a minimum block
 * above 256 samples only the window's last block can be that short). */
uint32_t Impl::chain_tail(uint32_t num_samples, bool search) const
{
    const uint32_t window_len = search ? par.num_lookahead_samples : par.max_num_samples_per_block;
    const uint32_t grid = search ? par.min_num_samples_per_block : par.max_num_samples_per_block;
    if (no_chain || (window_len % grid) != 0) return 0;
    const uint32_t tn = num_samples % window_len;
    if ((tn & 1u) && (grid & 1u) == 0) return tn;
    if (tn > 0 && par.ltp_order > 0 && grid > 256u && ((tn - 1u) % grid) + 1u <= 256u) return tn;
    return 0;
}

/* Why a stream of num_samples samples (0: the parameters alone) would not be guaranteed the reference's bytes (0: it is). */
uint32_t Impl::nonidentical_reasons(uint32_t num_samples) const
{
    uint32_t r = 0;
    const bool search = search_enabled();
    /* (SVR refinement together with history-dependent blocks: modelled -- the refinement's residual is one more writer of the
     * reference's buffer in chain / history mode, host_chain.cpp.  The one exception is counted where it happens: arbitrate_svr) */
    (void)num_samples;
    /* the reference's search takes its maximum from the configuration and then fails on the candidates beyond the parameters' */
    if (search && cfg.max_num_samples_per_block > par.max_num_samples_per_block) r |= SRLAMI355X_NONIDENTICAL_SEARCH_BEYOND_PARAMETERS;
    if (par.ltp_order > 0) {
        /* lpcc->buffer holds RoundUp2Powered(config max block) doubles (lpc.c:211): at most 256 of them => the 263 lags of
         * lpc.c:371-373 end in the transform's scratch buffer behind it */
        uint32_t p2 = 1;
        while (p2 < cfg.max_num_samples_per_block) p2 <<= 1;
        if (p2 < SRLA_LTP_LAGS) r |= SRLAMI355X_NONIDENTICAL_LTP_TINY_BUFFER;
    }
    return r;
}

/* counts a call made under such parameters and says so once per handle and reason */
void Impl::note_nonidentical(uint32_t num_samples) { note_reasons(nonidentical_reasons(num_samples)); }

void Impl::note_reasons(uint32_t r)
{
    if (r == 0) return;
    stats.num_nonidentical_calls++;
    stats.nonidentical_reasons |= r;
    if ((warned_reasons & r) != r) {
        warned_reasons |= r;
        fprintf(stderr, "[srla-mi355x] WARNING: output valid and lossless but NOT guaranteed bit-identical to the reference: %s\n",
                nonidentical_text(r).c_str());
    }
}

std::string Impl::nonidentical_text(uint32_t r)
{
    std::string t;
    if (r & SRLAMI355X_NONIDENTICAL_SVR_HISTORY)
        t += "an SVR refinement whose objective comparisons the host libm decided differently from the device, in a window whose later "
             "blocks inherit the refinement's residual (lpc.c:1047): the predictor is the host's, the residual left behind is not";
    if (r & SRLAMI355X_NONIDENTICAL_HANDLE_HISTORY) {
        if (!t.empty()) t += "; ";
        t += "a block of this call inherits a word of the reference's FFT buffer (lpc.c:260-264, 371-373) that an EARLIER call on this "
             "handle left there and that the library does not know: the last two windows of the stream before did not rewrite it (a "
             "stream that ends in digital silence, a call that failed) -- a fresh handle per stream (what the `srla` tool does) keeps the "
             "output bit-identical";
    }
    if (r & SRLAMI355X_NONIDENTICAL_SEARCH_BEYOND_PARAMETERS) {
        if (!t.empty()) t += "; ";
        t += "a block division search on an encoder created for a larger maximum block than its parameters name: the reference searches up "
             "to the CONFIGURATION's maximum (srla_encoder.c:598, :1669), refuses those candidates (:1499) and returns SRLA_APIRESULT_NG; "
             "this library searches within the parameters and succeeds -- create the encoder with the parameters' maximum block and the "
             "output is the reference's";
    }
    if (r & SRLAMI355X_NONIDENTICAL_LTP_TINY_BUFFER) {
        if (!t.empty()) t += "; ";
        t += "long-term predictor on an encoder created for blocks of at most 256 samples: the reference's FFT buffer is shorter than "
             "the 263 lags it copies out of it (lpc.c:371-373 reads its transform's scratch memory), which this library does not reproduce "
             "-- create the encoder with max_num_samples_per_block > 256 and the output is bit-identical";
    }
    return t;
}

/* The last valid block of segment k's last window of a priced job: (offset inside the stream, length). */
static bool last_block_of(Impl *im, Slot &ls, size_t k, uint32_t *off, uint32_t *n)
{
    const SegPlan &sp = ls.job.segs[k];
    const SrlaWindowDesc &wd = ls.job.windows[ls.job.seg_first_window[k + 1] - 1];
    std::vector<SrlaBlockRecord> recs(wd.num_nodes - 1);
    if (!im->d2h(recs.data(), ls.d_blocks.as<SrlaBlockRecord>() + wd.block_base, recs.size() * sizeof(SrlaBlockRecord)))
        return false;
    *off = 0; *n = 0;
    for (const SrlaBlockRecord &r : recs) if (r.valid) { *off = sp.s0 + (r.sample_off - sp.base); *n = r.n; }
    return true;
}

/* The body shared by every Encode* entry point: encodes the streams in `sx`. */
SRLAApiResult Impl::encode_streams(bool search)
{
    const auto t0 = Clock::now();
    const uint32_t nst = (uint32_t)sx.size(), nch = par.num_channels;
    const bool single = nst == 1;
    call_solo = false; planned_pieces = false;      /* per-call state: the history branch below returns before either is decided, and stage_input reads call_solo */
    auto drain = [&]() {
        for (auto &st : streams) if (st) (void)hipStreamSynchronize(st);
        if (upload) (void)hipStreamSynchronize(upload);
        if (chain_stream) (void)hipStreamSynchronize(chain_stream);
        if (dma_stream && dma_used) { (void)hipStreamSynchronize(dma_stream); dma_used = false; }
    };
    if (timeline) (void)hipEventRecord(ev_ref, streams[0]);
    if ((size_t)8 * nst > d_pos.cap) { drain(); if (!d_pos.ensure((size_t)8 * nst)) return SRLA_APIRESULT_NG; }
    /* page-lock pageable planes / buffers in place for this call (Impl::pin_inplace); dropped again when the call leaves */
    struct PinGuard {
        std::vector<const void *> held;
        ~PinGuard() { for (const void *p : held) host_pin_release(p); }
    } pins;
    /* the host-issued copies of finished jobs (Impl::dma_out) end before the call leaves -- and before `pins` lets go of the
     * buffers they write to (destroyed first: declared last) */
    struct DmaGuard {
        Impl *im;
        ~DmaGuard() { if (im->dma_stream && im->dma_used) { (void)hipStreamSynchronize(im->dma_stream); im->dma_used = false; } }
    } dma_guard{ this };
    /* pageable planes are locked in place when the pool is too small to stage them (all jobs by DMA then), and otherwise too
     */
    const bool few_threads = pool->size() < 4;     /* (round 5, with half of the channels packed on the way: 1 - 3 threads in place + 12 ... 30 %, 4 and more equal on long streams and staging ahead on short ones: profiles/r05/ab_hybrid_input.txt) */
    const bool want_pins = !force_staging && !pin_too_slow && (pin_inplace == 1 || (pin_inplace < 0 && few_threads));
    bool need_oracc = false;
    /* parameters under which blocks anywhere in the stream depend on the calls before them: window by window (host_chain.cpp) */
    /* ... and, for the reference's own entry points, calls of at most one window: they are where a handle's earlier calls can
     * reach into this one, and in history mode they leave the handle's buffer exactly as the reference's (host_impl.h, d_hist) */
    const bool tracked = single && sx[0].reference_call && !no_chain;
    /* (a call of at most one window reads the buffer only through an odd-length or short long-term-predictor block: chain_tail) */
    const bool one_window = tracked && sx[0].num_samples <= (search ? par.num_lookahead_samples : par.max_num_samples_per_block);
    const bool history = replaying || history_regime(search) || (one_window && chain_tail(sx[0].num_samples, search) != 0);
    if (history && tracked && !pending.empty() && !replaying) {
        /* this call may read the buffer: first what the regular calls on the handle left in it */
        std::vector<StreamCtx> mine;
        mine.swap(sx);
        const bool ok = replay_pending();
        sx.swap(mine);
        if (!ok) return SRLA_APIRESULT_NG;
    }
    call_tainted = false;
    tail.copied = false; tail.silent_stream = false;
    for (uint32_t si = 0; si < nst; si++) {
        StreamCtx &st = sx[si];
        classify_buffers(st, pins.held);
        /* (a stream below kPinMinMB MB of samples is not worth a registration, input or output: releasing it cost 0.07 ms of a 60 s
         * stream's 0.94 ms, and even ONE thread stages such a stream as fast as the link carries it in place -- 60 s of stereo with 1 / 2
         * / 3 pool threads: 3 170 / 3 400 - 3 530 / 3 650 - 3 700 Msamples/s staged, 2 940 - 3 170 / 3 090 / 3 100 in place; from 120 s on
         * the planes in place win (profiles/r05/ab_hybrid_input.txt).  SRLA_MI355X_PIN_INPLACE=1 asks for it from 4 MB on.) */
        const uint64_t sample_bytes = (uint64_t)st.num_samples * nch * 4u;
        const bool worth_pinning = sample_bytes >= ((pin_inplace == 1) ? (uint64_t)(4u << 20) : ((uint64_t)kPinMinMB << 20));
        if (want_pins && worth_pinning && st.host_in && !st.in_pinned) {
            const size_t before = pins.held.size();
            bool ok = true;
            double us_per_mb = 0.0;
            for (uint32_t ch = 0; ch < nch && ok; ch++) {
                ok = host_pin_acquire(st.host_in[ch], (size_t)st.num_samples * 4, &us_per_mb);
                if (ok) pins.held.push_back(st.host_in[ch]);
                /* without huge pages locking costs more than the staging copy it saves: remember, stage */
                if (ok && us_per_mb > 40.0 && pin_inplace != 1) { pin_too_slow = true; ok = false; }   /* (SRLA_MI355X_PIN_INPLACE=1 asked for it: "always" does not depend on what a box's registration costs -- a GPU test did, round 6) */
            }
            if (!ok) { while (pins.held.size() > before) { host_pin_release(pins.held.back()); pins.held.pop_back(); } }
            else { st.in_pinned = true; stats.num_inplace_pins += nch; }     /* (counted when the stream is read in place, not when a plane's registration turned out too slow and was dropped again) */
        }
        /* the output buffer always (unless switched off): the blocks then land in it straight from the device, which saves the
         * copy out of the staging buffers that the calling thread would otherwise make job by job (M: +5 %, and steadier) */
        const bool want_out_pin = want_pins || (!force_staging && !pin_too_slow && pin_inplace < 0);
        if (want_out_pin && worth_pinning && !pin_too_slow && st.data != nullptr && st.out_direct == nullptr && st.data_size > 0) {
            /* a stream never exceeds its raw size + block headers: no need to lock more of a generous buffer */
            const uint64_t blocks = (uint64_t)st.num_samples / std::max<uint32_t>(1u, par.min_num_samples_per_block) + 2u;
            const uint64_t bound = SRLA_HEADER_SIZE + ((uint64_t)st.num_samples * nch * par.bits_per_sample + 7u) / 8u + 16u * blocks + 4096u;
            const size_t bytes = (size_t)std::min<uint64_t>(st.data_size, bound);
            if (host_pin_acquire(st.data, bytes, nullptr)) {
                pins.held.push_back(st.data);
                stats.num_inplace_out_pins++;
                hipPointerAttribute_t at;
                memset(&at, 0, sizeof(at));
                if (hipPointerGetAttributes(&at, st.data) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer != nullptr)
                    st.out_direct = static_cast<uint8_t *>(at.devicePointer);
                else (void)hipGetLastError();
            }
        }
        st.write_off = st.with_header ? SRLA_HEADER_SIZE : 0u;
        st.progress = 0; st.pass_started = false; st.rc = SRLA_APIRESULT_OK;
        st.or_mask = 0; st.or_covered = 0; st.lshift_spec = false; st.lshift_on_device = false;
        st.or_on_device = false; st.or_dev_end = 0;
        if (st.lshift_final) { /* known: given by the caller (EncodeWindows), or the second attempt after a failed speculation */ }
        else if (!st.with_header) {
            /* block calls: encoder->header.offset_lshift, whatever the samples are */
            st.lshift = offset_lshift; st.lshift_final = true;
            st.raw_below_shift = false;
            if (st.lshift > 0 && st.host_in) {
                uint32_t m = 0;
                for (uint32_t ch = 0; ch < nch; ch++) m |= or_reduce(st.host_in[ch], st.num_samples);
                st.raw_below_shift = (m & ((1u << st.lshift) - 1u)) != 0;
            }
        }
        else if (st.d_in) {
            /* offset left shift: OR of every sample (srla_utility.c:177-203) on the device, without a host round trip: the
             * jobs read the shift from device memory */
            hipStream_t w = streams[0];
            if (!single || hipMemsetAsync(d_or.p, 0, 8, w) != hipSuccess) return SRLA_APIRESULT_NG;
            if (srla_launch_or_reduce(w, st.d_in, st.d_stride, st.num_samples, nch, d_or.as<uint32_t>()) != 0) return SRLA_APIRESULT_NG;
            if (hipMemcpyAsync(h_or.p, d_or.p, 8, hipMemcpyDeviceToHost, w) != hipSuccess) return SRLA_APIRESULT_NG;
            if (hipEventRecord(ev_or, w) != hipSuccess) return SRLA_APIRESULT_NG;
            st.lshift_on_device = true;
        } else if (st.pcm) {
            /* as for pinned planes below: a short look by the host (the whole stream with SRLA_MI355X_NO_SPECULATION), the
             * device gathers the rest */
            const uint32_t look = (no_speculation || history) ? st.num_samples : std::min<uint32_t>(st.num_samples, 65536u);
            uint32_t m = 0;
            for (uint32_t ch = 0; ch < nch; ch++) m |= pcm_channel(st.pcm, st.pcm_bytes, nch, ch, 0, look, nullptr);
            st.or_mask = m; st.or_covered = look;
            if (look == st.num_samples) { st.lshift = shift_of(m); st.lshift_final = true; }
            else { st.or_on_device = true; need_oracc = true; }
        } else if (st.in_pinned && st.cb == nullptr && !no_speculation && !history) {
            /* pinned planes: the host looks at the first 64 Ki samples per channel only; the device gathers the OR of
             * everything it uploads (stage_input) and the guess is checked against that at the end */
            const uint32_t look = std::min<uint32_t>(st.num_samples, 65536u);
            uint32_t m = 0;
            for (uint32_t ch = 0; ch < nch; ch++) m |= or_reduce(st.host_in[ch], look);
            st.or_mask = m; st.or_covered = look;
            if (look == st.num_samples) { st.lshift = shift_of(m); st.lshift_final = true; }
            else { st.or_on_device = true; need_oracc = true; }
        } else if (st.cb != nullptr || no_speculation || history) {
            /* delivered blocks cannot be taken back: the OR pass runs first */
            const uint32_t chunk = 1u << 20, per_ch = (st.num_samples + chunk - 1) / chunk;
            std::atomic<uint32_t> acc{ 0 };
            pool->parallel_for(per_ch * nch, [&](uint32_t i) {
                const uint32_t ch = i / per_ch, o = (i % per_ch) * chunk, len = std::min(chunk, st.num_samples - o);
                acc.fetch_or(or_reduce(st.host_in[ch] + o, len), std::memory_order_relaxed);
            });
            st.lshift = shift_of(acc.load());
            st.lshift_final = true;
        }
        /* the history-dependent last window goes through chain mode (host_chain.cpp) once everything before it is out: an
         * odd-length one (lpc.c:260-264), or -- LTP on -- one whose last block is shorter than the 263 lags (lpc.c:371-373;
         * with a minimum block above 256 samples only the window's last block can be that short) */
        uint32_t chain_n = history ? 0u : chain_tail(st.num_samples, search);
        st.chain_n = chain_n;
        st.body = st.num_samples - chain_n;
    }
    for (const StreamCtx &st : sx) if (st.with_header) note_nonidentical(st.num_samples);
    if (history) {
        overrides.clear();
        pool->set_linger_us(0u);                                  /* (window by window with host round trips: nothing for the pool to wait for) */
        const SRLAApiResult rc = history_encode(search);
        if (call_tainted && !replaying) note_reasons(SRLAMI355X_NONIDENTICAL_HANDLE_HISTORY);    /* (a replay's bytes are discarded; the taint reaches later real calls through hist_exact) */
        stats.total_ms += ms_since(t0);
        return rc;
    }
    /* a regular call of several windows: none of its blocks reaches back beyond its own stream, and what it leaves in the
     * reference's buffer is not tracked */
    /* (tracked: what the call leaves in the reference's buffer becomes a capture, keep_tail / push_capture) */
    if (need_oracc) {
        if ((size_t)8 * nst > d_oracc.cap) { drain(); if (!d_oracc.ensure((size_t)8 * nst)) return SRLA_APIRESULT_NG; }
        if (hipMemsetAsync(d_oracc.p, 0, (size_t)8 * nst, upload) != hipSuccess) return SRLA_APIRESULT_NG;
    }
    std::vector<JobPlan> plan;
    plan_jobs(plan, search);
    const uint32_t njobs = (uint32_t)plan.size();
    overrides.clear();
    call_crowded = njobs > 3;
    spin_collect = njobs <= 3;
    pool->set_linger_us(spin_collect ? kPoolLingerUs : 0u);      /* (a short call's two rounds -- staging, copy-out -- are 0.3 ms apart) */
    /* (a stream of a few pieces is a latency chain: its copies would start only when the host has collected each piece) */
    call_dma = dma_out && dma_stream != nullptr && njobs > 3;
    for (const StreamCtx &st : sx) call_dma = call_dma && st.out_direct != nullptr && st.data != nullptr && st.cb == nullptr;
    if (timeline) tl_printf("[timeline] %u stream(s), %u jobs; host %.3f ms into the call\n", nst, njobs, ms_since(t0));

    auto fail = [&](SRLAApiResult rc) {
        drain();
        for (auto &sl : slot) sl.busy = false;
        chain.active = false;
        if (tracked) { drop_pending(); hist_exact = 0; hist_fresh = false; }         /* (the reference's call stopped somewhere, too: its buffer is no longer the fresh handle's zeros) */
        return rc;
    };
    auto job_slot = [&](uint32_t k) -> Slot & { return slot[plan[k].slot]; };
    const bool chain_any = single && sx[0].chain_n != 0;          /* a history-dependent last window: its jobs' blocks follow the regular jobs' on stream C */
    call_solo = njobs == 1 && !chain_any && !timeline;
    auto begin = [&](uint32_t k) -> bool {
        Slot &s = job_slot(k);
        if (!stage_input(s, plan[k])) return false;
        std::vector<uint32_t> lsh;
        settle_lshift(plan[k], lsh);
        build_job(s.job, plan[k], lsh, search);
        if (apply_overrides(s.job, k)) { s.job.uploaded = false; s.job.key = 0; }
        if (single && sx[0].raw_below_shift) mark_raw_silence(s.job);
        /* a call of one job has nothing to overlap: its stages run on ONE stream, without the cross-stream hand-overs
         * (about 13 us each; a 10 s stream: 0.49 -> 0.465 ms) */
        s.own_stream = (njobs == 1) ? streams[0] : nullptr;
        /* The pieces of a short stream (plan_jobs): each runs its stages A - D on ONE stream (two streams, taken in turn) without end
         * events between them, only the block assembly stays on C, where the order of the stream's blocks is made -- one hand-over per
         * piece instead of four, and a piece's pricing no longer queues behind the next piece's solve chain (40 s: 3 300 -> 3 440, 60 s:
         * 3 880 -> 3 960, 90 s: 3 430 -> 3 980, 120 s: 3 890 -> 4 240 Msamples/s).  Full-size jobs keep the wide stream: two of them
         * side by side lose (300 s: - 2 %). */
        s.piece = planned_pieces && njobs >= 2 && !chain_any && !timeline;
        if (s.piece) s.own_stream = streams[k & 1u];
        s.solo = call_solo;
        s.emits = true; s.merge_cb = false;
        /* One job in `timing_stride` carries start events on its launches (a start event costs a launch about 3 us: all of them on
         * every job were 2 % of a long call and 10 % of a 10 s call); the jobs of short calls are counted across calls, so that a
         * call of ONE job is timed every fourth time instead of always. */
        s.timed = timing && ((njobs > 1 ? k : short_call_jobs++) % timing_stride == 0);
        /* (a call of two or three jobs keeps round 4's events -- its first job timed, srla_residual_cost's start event on every job:
         * measured FASTER than fewer or none, 60 s: 3 910 against 3 790 / 3 850 Msamples/s; profiles/r05/ab_host_path.txt) */
        s.c_start = s.timed || (timing && njobs >= 2 && njobs <= 3 && !s.piece);
        s.out_boost = (k + kTailBoostJobs >= njobs) ? kTailBoost : 1u;
        s.last_job = k + kDmaTailJobs >= njobs;              /* (the last jobs of the call: the copy-out kernel, no host round trip) */
        return prepare_job(s, false);
    };
    /* chain mode of the (single) stream, overlapped with the regular jobs */
    chain.active = single && sx[0].chain_n != 0;
    chain.begun = false; chain.early = false; chain.ad_done = false;
    uint32_t chain_seed_off = 0, chain_seed_n = 0;
    if (chain.active) {
        StreamCtx &st = sx[0];
        chain.stream = 0; chain.tail_start = st.body; chain.tail_n = st.chain_n; chain.search = search;
        /* The window's search does not depend on the jobs before it, except through the last block encoded before
         * the window when the window's first history-dependent call can reach back that far: a window of a single
         * candidate (search), or any window when every block is a window of its own.  Without searching that
         * block is known now; otherwise it is read from the last regular job once that has been priced (below). */
        const uint32_t nodes = search ? (st.chain_n + par.min_num_samples_per_block - 1) / par.min_num_samples_per_block + 1 : 2u;
        if (st.body == 0 || (search && nodes >= 3)) chain.early = true;
        else if (!search) { chain.early = true; chain_seed_off = st.body - par.max_num_samples_per_block; chain_seed_n = par.max_num_samples_per_block; }
    }
    /* the chain-mode window of stream `i`, synchronously: seed from the priced job in `ls` (segment k), or none */
    auto chain_sync = [&](uint32_t i, Slot *ls, size_t k) -> SRLAApiResult {
        StreamCtx &st = sx[i];
        chain.active = true; chain.begun = false; chain.early = false; chain.ad_done = false;
        chain.stream = i; chain.tail_start = st.body; chain.tail_n = st.chain_n; chain.search = search;
        uint32_t seed_off = 0, seed_n = 0;
        const uint32_t nodes = search ? (st.chain_n + par.min_num_samples_per_block - 1) / par.min_num_samples_per_block + 1 : 2u;
        if (st.body == 0 || (search && nodes >= 3)) { /* nothing before the window matters */ }
        else if (!search) { seed_off = st.body - par.max_num_samples_per_block; seed_n = par.max_num_samples_per_block; }
        else if (ls == nullptr || !last_block_of(this, *ls, k, &seed_off, &seed_n)) return SRLA_APIRESULT_NG;
        if (!chain_begin(seed_off, seed_n) || !chain_encode_ad() || !chain_encode_e()) return SRLA_APIRESULT_NG;
        const SRLAApiResult rc = chain_collect();
        chain.active = false;
        return rc;
    };

    /* Software pipeline over jobs: iteration t enqueues  autocorr + solve of job t,  residual_cost +
     * pricing of job t-1,  block assembly of job t-2,  then collects job t-3.  With the long-term predictor one step more:
     * LTP-pass autocorr (W) + pitch solve (N) of job t,  LPC-pass autocorr + solve of job t-1,  residual_cost + pricing of
     * job t-2,  block assembly of job t-3,  collect job t-4 -- W never waits for the pitch solve.  Needs depth + 1 buffer sets.
     * `base`: the jobs before it are complete; a job whose near-ties the host libm decides differently from the device
     * (arbitrate) sends the loop back to it. */
    const uint32_t ltp_skew = (par.ltp_order > 0 && split_ltp_stage) ? 1u : 0u;
    const uint32_t depth = 3 + ltp_skew;
    /* The host may run further ahead than the stages' skew asks for: a job is collected `lag` iterations after it was begun, and
     * every buffer set beyond depth + 1 is one more job staged and uploaded while the device still works on older ones (host
     * input: staging 0.28 ms + upload 0.3 ms per 4 M-sample job on top of the 1.5 ms a job takes from its first kernel to its
     * last byte; with lag = depth the device waited for input about a tenth of the time). */
    const uint32_t lag = depth + std::min<uint32_t>(kRunAhead, (kSlots > depth + 1u) ? kSlots - 1u - depth : 0u);
    uint32_t base = 0, restarts = 0;
    auto in_flight = [&](uint32_t t, uint32_t back) { return t >= back && t - back < njobs && t - back >= base; };
    for (uint32_t t = 0; t < njobs + lag;) {
        const auto t_enq = Clock::now();
        if (in_flight(t, 0)) {
            if (!begin(t) || !run_stage(job_slot(t), ST_A, ltp_skew ? 1 : 0)) return fail(SRLA_APIRESULT_NG);
        }
        /* (the block assembly first: on stream N it must not queue behind this iteration's solve and pricing, which wait for
         * wide kernels that have only just been enqueued) */
        if (in_flight(t, 2 + ltp_skew)) {
            Slot &s = job_slot(t - 2 - ltp_skew);
            if (!run_stage(s, ST_E)) return fail(SRLA_APIRESULT_NG);
        }
        if (in_flight(t, ltp_skew)) {
            Slot &s = job_slot(t - ltp_skew);
            if ((ltp_skew && !run_stage(s, ST_A, 2)) || !run_stage(s, ST_B)) return fail(SRLA_APIRESULT_NG);
        }
        if (in_flight(t, 1 + ltp_skew)) {
            Slot &s = job_slot(t - 1 - ltp_skew);
            if (!run_stage(s, ST_C) || !run_stage(s, ST_D)) return fail(SRLA_APIRESULT_NG);
        }
        if (single && chain.active && chain.early) {
            /* the host prepares the chain jobs while the device works on the first regular job */
            if (!chain.begun) {
                const auto tc = Clock::now();
                if (!chain_begin(chain_seed_off, chain_seed_n)) return fail(SRLA_APIRESULT_NG);
                if (chain_trace) fprintf(stderr, "[chain] begin %.3f ms (%zu calls)\n", ms_since(tc), chain_calls.size());
            }
            if (!chain.ad_done && (t == njobs || chain_search_done())) {
                const auto tc = Clock::now();
                if (!chain_encode_ad()) return fail(SRLA_APIRESULT_NG);
                if (chain_trace) fprintf(stderr, "[chain] encode_ad %.3f ms (%zu calls)\n", ms_since(tc), chain_calls.size());
            }
            if (t == njobs + depth - 2 && !chain_encode_e()) return fail(SRLA_APIRESULT_NG);   /* right behind the last job's block assembly */
        }
        stats.h2d_ms += ms_since(t_enq);       /* host time spent staging and enqueueing */
        if (timeline) tl_printf("[timeline] host: iteration %u enqueued at %.3f ms\n", t, ms_since(t0));
        if (t < lag || t - lag < base) { t++; continue; }
        const uint32_t k = t - lag;
        Slot &s = job_slot(k);
        if (tracked && k + 1 == njobs && !tail.copied && !keep_tail(sx[0], search)) return fail(SRLA_APIRESULT_NG);   /* (everything is enqueued: the host would only wait) */
        if (!wait_job(s)) return fail(SRLA_APIRESULT_NG);
        if (timeline) tl_printf("[timeline] host: job %u collected at %.3f ms\n", k, ms_since(t0));
        if (s.h_info.as<SrlaJobInfo>()->num_tie_items != 0) {
            const int m = arbitrate(s, k);
            if (m < 0) return fail(SRLA_APIRESULT_NG);
            if (m > 0) {
                /* the host libm decided otherwise: everything from this job on is enqueued again (the jobs in flight behind it
                 * were placed behind its bytes) */
                if (++restarts > 16) { fprintf(stderr, "[srla-mi355x] internal error: near-tie arbitration did not settle\n"); return fail(SRLA_APIRESULT_NG); }
                drain();
                for (StreamCtx &st : sx) st.pass_started = false;
                stats.num_restarts++;
                base = k; t = k;
                continue;
            }
        }
        const SRLAApiResult rc = finish_job(s);
        if (timeline) tl_printf("[timeline] host: job %u finished at %.3f ms\n", k, ms_since(t0));
        if (rc != SRLA_APIRESULT_OK && (single || rc != SRLA_APIRESULT_INSUFFICIENT_BUFFER)) {
            if (rc == SRLA_APIRESULT_INSUFFICIENT_BUFFER && sx[0].lshift_spec) { drain(); sx[0].or_dev_end = 0; /* not every job was staged: the device's OR is partial, the host looks at the whole stream */ break; }   /* perhaps only because the shift was guessed wrong: see below */
            return fail(rc);
        }
        if (!single) {
            /* streams whose regular windows end in this job and that have a chain-mode window */
            for (size_t g = 0; g < s.job.segs.size(); g++) {
                const SegPlan &sp = s.job.segs[g];
                StreamCtx &st = sx[sp.stream];
                if (st.chain_n == 0 || sp.s0 + sp.ns != st.body || st.rc != SRLA_APIRESULT_OK) continue;
                const SRLAApiResult crc = chain_sync(sp.stream, &s, g);
                if (crc != SRLA_APIRESULT_OK && crc != SRLA_APIRESULT_INSUFFICIENT_BUFFER) return fail(crc);
            }
        }
        t++;
    }
    if (single && chain.active && sx[0].rc == SRLA_APIRESULT_OK) {
        if (!chain.early) {
            /* the last block encoded before the window: its final call is what the window's only candidate inherits from */
            uint32_t seed_off = 0, seed_n = 0;
            Slot &ls = job_slot(njobs - 1);
            if (!last_block_of(this, ls, ls.job.segs.size() - 1, &seed_off, &seed_n)) return fail(SRLA_APIRESULT_NG);
            if (!chain_begin(seed_off, seed_n) || !chain_encode_ad() || !chain_encode_e()) return fail(SRLA_APIRESULT_NG);
        }
        const SRLAApiResult rc = chain_collect();
        if (rc != SRLA_APIRESULT_OK && !(rc == SRLA_APIRESULT_INSUFFICIENT_BUFFER && sx[0].lshift_spec)) return fail(rc);
    }
    chain.active = false;
    if (!single) {
        for (uint32_t i = 0; i < nst; i++)
            if (sx[i].chain_n != 0 && sx[i].body == 0 && sx[i].rc == SRLA_APIRESULT_OK) {
                const SRLAApiResult crc = chain_sync(i, nullptr, 0);
                if (crc != SRLA_APIRESULT_OK && crc != SRLA_APIRESULT_INSUFFICIENT_BUFFER) return fail(crc);
            }
    }
    /* headers; speculated offset shifts: finish the OR of the stream and, in the rare case that the shift the stream was
     * encoded with is not the one the whole stream has, encode it again */
    SRLAApiResult worst = SRLA_APIRESULT_OK;
    for (uint32_t i = 0; i < nst; i++) {
        StreamCtx &st = sx[i];
        if (st.with_header && st.lshift_spec && !st.lshift_final) {
            if (st.or_on_device && st.or_dev_end >= st.num_samples) {
                /* every job has been collected, so every reduction launch on the upload stream is complete */
                uint32_t m = 0;
                if (hipStreamSynchronize(upload) != hipSuccess || !d2h(&m, d_oracc.as<uint32_t>() + 2u * i, 4))
                    return fail(SRLA_APIRESULT_NG);
                st.or_mask |= m;
                st.or_covered = st.num_samples;
            } else if (st.or_on_device) st.or_covered = 0;      /* the stream ended early: the host looks at all of it */
            if (st.or_covered < st.num_samples) {
                const uint32_t o0 = st.or_covered, len = st.num_samples - o0;
                const uint32_t chunk = 1u << 20, per_ch = (len + chunk - 1) / chunk;
                std::atomic<uint32_t> acc{ 0 };
                pool->parallel_for(per_ch * nch, [&](uint32_t q) {
                    const uint32_t ch = q / per_ch, o = (q % per_ch) * chunk;
                    const uint32_t m = st.pcm ? pcm_channel(st.pcm, st.pcm_bytes, nch, ch, (size_t)o0 + o, std::min(chunk, len - o), nullptr)
                                              : or_reduce(st.host_in[ch] + o0 + o, std::min(chunk, len - o));
                    acc.fetch_or(m, std::memory_order_relaxed);
                });
                st.or_mask |= acc.load();
                st.or_covered = st.num_samples;
            }
            const uint32_t true_shift = shift_of(st.or_mask);
            st.lshift_final = true;
            if (true_shift != st.lshift) {
                drain();
                std::vector<StreamCtx> saved;
                saved.swap(sx);
                StreamCtx again = saved[i];
                again.lshift = true_shift; again.lshift_final = true;
                sx.push_back(again);
                /* the nested call is this stream's alone: it must not count as a call of its own in the statistics, nor leave
                 * its shift in the handle as if the whole call had been a single stream */
                const uint32_t handle_lshift = offset_lshift;
                const double total_before = stats.total_ms;
                const SRLAApiResult nrc = encode_streams(search);
                StreamCtx result = sx[0];
                sx.swap(saved);
                sx[i] = result;
                stats.total_ms = total_before;
                if (!single) offset_lshift = handle_lshift;
                if (nrc != SRLA_APIRESULT_OK && nrc != SRLA_APIRESULT_INSUFFICIENT_BUFFER) return fail(nrc);   /* launch / allocation / HIP error */
                if (nrc != SRLA_APIRESULT_OK && sx[i].rc == SRLA_APIRESULT_OK) sx[i].rc = nrc;
                if (sx[i].rc != SRLA_APIRESULT_OK) worst = sx[i].rc;
                continue;               /* the nested call wrote the header */
            }
        }
        if (st.rc != SRLA_APIRESULT_OK) { worst = st.rc; continue; }
        if (!write_header(st)) return fail(SRLA_APIRESULT_NG);
    }
    if (single && sx[0].with_header && sx[0].rc == SRLA_APIRESULT_OK) offset_lshift = sx[0].lshift;   /* encoder->header of the reference */
    if (timeline) {
        /* (what the guards above do when the call leaves, here under the clock) */
        tl_printf("[timeline] host: streams complete at %.3f ms\n", ms_since(t0));
        if (dma_stream && dma_used) { (void)hipStreamSynchronize(dma_stream); dma_used = false; }
        tl_printf("[timeline] host: copies complete at %.3f ms\n", ms_since(t0));
        for (const void *p : pins.held) host_pin_release(p);
        pins.held.clear();
    }
    if (want_block_price && tracked && worst == SRLA_APIRESULT_OK && njobs > 0) {
        /* SRLAEncoder_ComputeBlockSize through the regular pipeline: the search's price of the block (host_chain.cpp: history_window) */
        Slot &ps = job_slot(njobs - 1);
        SrlaBlockRecord rec;
        if (ps.job.windows.empty() || !d2h(&rec, ps.d_blocks.as<SrlaBlockRecord>() + ps.job.windows[0].block_base, sizeof(rec)) || !rec.valid) return fail(SRLA_APIRESULT_NG);
        block_price = rec.price;
    }
    if (tracked && worst != SRLA_APIRESULT_OK) { drop_pending(); hist_exact = 0; hist_fresh = false; }             /* (a call that failed on the way) */
    else if (tracked && tail.copied) {
        /* what a later call on this handle may have to know (host_impl.h, Capture); the shift is final only now */
        if (sx[0].d_in && hipStreamSynchronize(upload) != hipSuccess) return fail(SRLA_APIRESULT_NG);
        tail.c.par = par; tail.c.lshift = sx[0].lshift; tail.c.search = search;
        tail.c.raw_below_shift = sx[0].raw_below_shift;
        if (!push_capture()) return fail(SRLA_APIRESULT_NG);
    }   /* (a silent stream: no call of the reference's calculator, nothing to keep) */
    stats.total_ms += ms_since(t0);
    if (timeline) { tl_printf("[timeline] call returned at %.3f ms\n", ms_since(t0)); fputs(tl_log.c_str(), stderr); tl_log.clear(); }
    return worst;
}

/* The reference decides "silent" on the samples as they come (srla_encoder.c:783-791), the items' flags on the samples after the
 * offset shift: the same thing unless a block call's samples have bits below a shift that an earlier EncodeWhole left in the handle. */
void Impl::mark_raw_silence(Job &job)
{
    const StreamCtx &st = sx[job.segs[0].stream];
    if (st.host_in == nullptr) return;
    const SegPlan &sp = job.segs[0];
    for (SrlaCandDesc &cd : job.cands) {
        const uint32_t off = sp.s0 + (cd.sample_off - sp.base);
        bool silent = true;
        for (uint32_t ch = 0; ch < par.num_channels && silent; ch++) {
            const int32_t *p = st.host_in[ch] + off;
            for (uint32_t i = 0; i < cd.n; i++) if (p[i] != 0) { silent = false; break; }
        }
        cd.raw_silence = silent ? 1u : 2u;
    }
    job.uploaded = false; job.key = 0;          /* (the table holds what the samples are: not one to be found again by shape) */
}

bool Impl::d2h(void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return true;
    if (!h_bounce.ensure(bytes)) return false;
    if (hipMemcpy(h_bounce.p, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
    memcpy(dst, h_bounce.p, bytes);
    return true;
}

bool Impl::d2h_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height)
{
    if (width == 0 || height == 0) return true;
    if (!h_bounce.ensure(width * height)) return false;
    if (hipMemcpy2D(h_bounce.p, width, src, spitch, width, height, hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (size_t r = 0; r < height; r++) memcpy(static_cast<uint8_t *>(dst) + r * dpitch, h_bounce.as<uint8_t>() + r * width, width);
    return true;
}

/* The samples a capture of the running call needs: the whole call when it is one window; of a stream of several windows the last
 * audible window and the one before it.  Also what the capture is certain to rewrite and cannot write beyond (host_impl.h). */
bool Impl::keep_tail(const StreamCtx &st, bool search)
{
    const uint32_t window_len = search ? par.num_lookahead_samples : par.max_num_samples_per_block, nch = par.num_channels;
    const uint32_t minb = par.min_num_samples_per_block, maxb = par.max_num_samples_per_block;
    const uint32_t nwin = (st.num_samples + window_len - 1) / window_len;
    uint32_t start = (nwin >= 2 ? nwin - 2 : 0) * window_len, n = st.num_samples - start;
    tail.copied = false; tail.silent_stream = false;
    if (st.host_in == nullptr && st.d_in == nullptr) return true;
    /* (planes in device memory: the last eight windows come over, and push_capture looks back over their digital silence) */
    tail.from_device = st.host_in == nullptr;
    tail.window_len = window_len;
    if (tail.from_device) { start = (nwin > 8u ? nwin - 8u : 0u) * window_len; n = st.num_samples - start; }
    tail.whole_stream = start == 0;
    if (st.host_in) {
        /* digital silence at the end: its blocks are not analysed (srla_encoder.c:766-796), so what the buffer holds is what the
         * last AUDIBLE window and the one before it left -- those two are kept (looking back over at most 64 windows) */
        const uint32_t lo = (nwin > 64u) ? (nwin - 64u) * window_len : 0u;
        uint32_t audible = lo;                                  /* one past the last non-zero sample */
        for (uint32_t ch = 0; ch < nch; ch++) {
            const int32_t *p = st.host_in[ch];
            uint32_t i = st.num_samples;
            while (i > audible && p[i - 1] == 0) i--;
            audible = std::max(audible, i);
        }
        if (audible == 0) { tail.silent_stream = true; return true; }      /* nothing was analysed: the buffer, and what is kept of the calls before, stay */
        if (audible > lo) {
            const uint32_t wa = (audible - 1) / window_len;
            start = (wa >= 1 ? wa - 1 : 0) * window_len;
            n = std::min<uint64_t>(st.num_samples, (uint64_t)(wa + 1) * window_len) - start;
        }
    }
    Capture &c = tail.c;
    if (!c.smp.ensure((size_t)nch * n * 4)) return false;
    for (uint32_t ch = 0; ch < nch; ch++) {
        int32_t *dst = c.smp.as<int32_t>() + (size_t)ch * n;
        if (st.host_in) memcpy(dst, st.host_in[ch] + start, (size_t)n * 4);
        else if (hipMemcpyAsync(dst, st.d_in + (size_t)ch * st.d_stride + start, (size_t)n * 4, hipMemcpyDeviceToHost, upload) != hipSuccess) return false;
    }
    c.n = n; c.nch = nch; c.multi = nwin >= 2;
    /* the longest candidate a window of the call holds, and whether one of that length is certain to be analysed (longer than the
     * predictor order, srla_encoder.c:766-796, and not silent) -- looked at in the FIRST kept window, whose transforms every later
     * one of the call can only overwrite from word 0 up */
    const uint32_t w0 = std::min(window_len, n);
    const uint32_t cap_len = search ? std::max(minb, (maxb / minb) * minb) : maxb;       /* candidates are whole minimum blocks, clipped at the window's end */
    const uint32_t longest = std::min(cap_len, w0);
    c.extent = geoms[geom_for(c.multi ? std::min(cap_len, window_len) : longest)].nfft;
    c.rewrites = 0;
    tail.longest = longest;                     /* (push_capture looks at the kept samples once they are all there) */
    tail.copied = true;
    return true;
}

void Impl::drop_pending()
{
    for (Capture *c : pending) spare.push_back(c);
    pending.clear();
}

/* the running call's capture joins the pending ones; older ones that cannot write beyond what it is certain to rewrite go */
bool Impl::push_capture()
{
    Capture *c = nullptr;
    if (!spare.empty()) { c = spare.back(); spare.pop_back(); } else c = new Capture();
    std::swap(c->smp, tail.c.smp);
    c->par = tail.c.par; c->lshift = tail.c.lshift; c->n = tail.c.n; c->nch = tail.c.nch;
    c->extent = tail.c.extent; c->rewrites = 0; c->search = tail.c.search; c->multi = tail.c.multi;
    c->raw_below_shift = tail.c.raw_below_shift;
    if (tail.from_device) {
        /* digital silence at the end, as keep_tail does it for host planes: the last audible window and the one before it stay */
        const uint32_t wl = tail.window_len, n0 = c->n;
        uint32_t audible = 0;
        for (uint32_t ch = 0; ch < c->nch; ch++) {
            const int32_t *p = c->smp.as<int32_t>() + (size_t)ch * n0;
            uint32_t i = n0;
            while (i > audible && p[i - 1] == 0) i--;
            audible = std::max(audible, i);
        }
        if (audible == 0 && tail.whole_stream) { spare.push_back(c); tail.copied = false; return true; }     /* a silent stream: nothing to keep */
        if (audible != 0) {
            const uint32_t wa = (audible - 1) / wl, ns = (wa >= 1 ? wa - 1 : 0) * wl;
            const uint32_t nn = (uint32_t)std::min<uint64_t>(n0, (uint64_t)(wa + 1) * wl) - ns;
            if (ns != 0 || nn != n0) {
                for (uint32_t ch = 0; ch < c->nch; ch++)
                    memmove(c->smp.as<int32_t>() + (size_t)ch * nn, c->smp.as<int32_t>() + (size_t)ch * n0 + ns, (size_t)nn * 4);
                c->n = nn;
            }
        }
    }
    /* is a candidate of the longest kind certain to be analysed -- longer than the predictor order (srla_encoder.c:777-779) and not
     * silent (:783-791)?  The first one of the first kept window: its transform rewrites every word below its length */
    if (tail.longest > srla::kPresetOrder[c->par.preset] && tail.longest <= c->n) {
        bool audible = false;
        for (uint32_t ch = 0; ch < c->nch && !audible; ch++) {
            const int32_t *p = c->smp.as<int32_t>() + (size_t)ch * c->n;
            for (uint32_t i = 0; i < tail.longest; i++) if (p[i] != 0) { audible = true; break; }
        }
        if (audible) c->rewrites = geoms[geom_for(tail.longest)].nfft;
    }
    if (c->rewrites != 0)
        for (size_t k = pending.size(); k-- > 0;)
            if (pending[k]->extent <= c->rewrites) { spare.push_back(pending[k]); pending.erase(pending.begin() + (long)k); }
    pending.push_back(c);
    tail.copied = false;
    if (pending.size() >= kMaxPending) {
        /* (captures that hide nothing of each other pile up -- calls of growing silence, say: settle them now) */
        std::vector<StreamCtx> mine;
        mine.swap(sx);
        const bool ok = replay_pending();
        sx.swap(mine);
        return ok;
    }
    return true;
}

bool Impl::replay_pending()
{
    std::vector<Capture *> todo;
    todo.swap(pending);
    bool ok = true;
    for (Capture *c : todo) { if (ok) ok = replay_one(*c); spare.push_back(c); }
    if (!ok) hist_exact = 0;
    return ok;
}

/* One captured call once more, in history mode, under the parameters of that call, the bytes discarded: afterwards the handle's
 * buffer holds what the reference's held when that call returned (as far as hist_exact says). */
bool Impl::replay_one(Capture &c)
{
    const auto t0 = Clock::now();
    const SRLAEncodeParameter keep_par = par;
    const uint32_t keep_shift = offset_lshift, keep_warned = warned_reasons;
    const SRLAMI355XStats keep_stats = stats;
    const bool keep_price = want_block_price;
    want_block_price = false;
    par = c.par; param_generation++;
    /* (what the stream's earlier windows left below the kept windows' reach is not known -- on a handle that has run nothing but
     * regular calls, too: its buffer is then NOT the fresh handle's zeros any more, which history_encode would otherwise assume) */
    if (c.multi) { hist_exact = 0; hist_fresh = false; }
    const uint32_t nch = c.nch, n = c.n;
    std::vector<const int32_t *> planes(nch);
    for (uint32_t ch = 0; ch < nch; ch++) planes[ch] = c.smp.as<int32_t>() + (size_t)ch * n;
    replay_out.resize((size_t)nch * n * 4 + 64u * (n / std::max<uint32_t>(1u, par.min_num_samples_per_block) + 2u) + 4096u);
    StreamCtx st;
    st.host_in = planes.data(); st.num_samples = n;
    st.data = replay_out.data(); st.data_size = (uint32_t)std::min<size_t>(replay_out.size(), 0xFFFFFFFFu);
    st.with_header = false; st.reference_call = true;
    st.lshift = c.lshift; st.lshift_final = true;
    st.raw_below_shift = c.raw_below_shift;     /* (a block call on samples with bits below the handle's shift: silence is decided on the raw samples, as in the call itself) */
    sx.clear();
    sx.push_back(st);
    replaying = true;
    const SRLAApiResult rc = encode_streams(c.search);
    replaying = false;
    par = keep_par; param_generation++;
    offset_lshift = keep_shift; warned_reasons = keep_warned;
    want_block_price = keep_price;
    stats = keep_stats;
    stats.history_ms += ms_since(t0);
    if (rc != SRLA_APIRESULT_OK) { hist_exact = 0; fprintf(stderr, "[srla-mi355x] internal error: the replay of a captured call failed (%d)\n", (int)rc); return false; }
    return true;
}
