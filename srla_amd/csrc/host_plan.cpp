/*
 * host_plan.cpp -- what the host prepares for a job: host-libm constants per block length, the window / candidate /
 * item tables of the block-division search (srla_encoder.c:336-389) and the job parameters.
 */
#include "host_impl.h"

#include <algorithm>
#include <math.h>
#include <vector>
#include <stdlib.h>
#include <string.h>

namespace srla {
const uint32_t kPresetOrder[SRLA_NUM_PARAMETER_PRESETS] = { 0, 8, 16, 32, 64, 128, 255 };
}

uint32_t Impl::geom_for(uint32_t n)
{
    auto it = geom_index.find(n);
    if (it != geom_index.end()) return it->second;
    SrlaGeom g;
    srla::fill_geom(n, &g);
    auto tw = tw_index.find(g.nfft);
    if (tw == tw_index.end()) {
        const uint32_t off = (uint32_t)(tw_host.size() / 2);
        tw_host.resize(tw_host.size() + 2 * (size_t)srla::twiddle_count(g.nfft));
        srla::build_twiddles(g.nfft, tw_host.data() + 2 * (size_t)off);
        tw = tw_index.emplace(g.nfft, off).first;
        tw_dirty = true;
    }
    g.tw_off = tw->second;
    const uint32_t idx = (uint32_t)geoms.size();
    geoms.push_back(g);
    geom_index.emplace(n, idx);
    geom_dirty = true;
    return idx;
}

bool Impl::sync_tables()
{
    if (welch_bps != par.bits_per_sample) {
        /* the Welch weights of the four block lengths that fill a transform of the LDS classes (lpc.c:256-266), with the products
         * in the kernel's own order -- (divisor 2^-(bps-1) smpl) (n - 1 - smpl): scaling by a power of two commutes with every
         * rounding -- so that a thread multiplies its sample by a table entry instead of forming five operations per sample */
        for (auto &st : streams) if (st) HIP_OK(hipStreamSynchronize(st));
        std::vector<double> tab(SRLA_WELCH_TAB_WORDS);
        const double norm_bps = ldexp(1.0, -(int)(par.bits_per_sample - 1));
        for (uint32_t n = 1024; n <= 8192; n <<= 1) {
            SrlaGeom g;
            srla::fill_geom(n, &g);
            const double div_scaled = g.welch_divisor * norm_bps, d_nm1 = (double)(n - 1u);
            double *t = tab.data() + (n / 2 - 512);
            for (uint32_t e = 0; e < n / 2; e++) { const double de = (double)e, dr = d_nm1 - de; t[e] = div_scaled * de * dr; }
        }
        if (!d_welch.ensure(tab.size() * sizeof(double))) return false;
        HIP_OK(hipMemcpy(d_welch.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        welch_bps = par.bits_per_sample;
    }
    if (!tw_dirty && !geom_dirty) return true;
    for (auto &st : streams) if (st) HIP_OK(hipStreamSynchronize(st));
    if (tw_dirty) {
        if (!d_tw.ensure(tw_host.size() * sizeof(double))) return false;
        HIP_OK(hipMemcpy(d_tw.p, tw_host.data(), tw_host.size() * sizeof(double), hipMemcpyHostToDevice));
        tw_dirty = false;
    }
    if (geom_dirty) {
        if (!d_geoms.ensure(geoms.size() * sizeof(SrlaGeom))) return false;
        HIP_OK(hipMemcpy(d_geoms.p, geoms.data(), geoms.size() * sizeof(SrlaGeom), hipMemcpyHostToDevice));
        geom_dirty = false;
    }
    return true;
}

SrlaLdsPlan Impl::lds_plan(uint32_t nfft) const
{
    uint32_t mp = 0;
    while ((1u << (mp + 1)) <= nfft && mp < SRLA_MAX_PORDER) mp++;
    const uint32_t sig_bytes = 4 * (nfft + SRLA_FIR_PAD), means_bytes = 8 * (2u << mp);
    auto al = [](uint32_t v) { return (v + 15u) & ~15u; };
    SrlaLdsPlan p{};
    uint32_t off = 0;
    p.y_off = off; off += al(sig_bytes);
    p.fft_off = off;
    p.lev_off = 0;
    p.means_off = off; off += al(means_bytes);
    p.small_off = off; off += srla_kernel_small_c_bytes();
    p.total = off;
    /* the fast path of the same kernel carves the block differently: make sure it fits too */
    for (uint32_t fl = 1; fl <= 8 && 1024u * fl <= nfft; fl++) p.total = std::max(p.total, srla_kernel_fast_lds_bytes(fl, par.ltp_order, par.bits_per_sample));
    return p;
}

static uint64_t fnv64(uint64_t h, uint64_t v)
{
    for (int i = 0; i < 8; i++) { h ^= (v >> (8 * i)) & 0xFFu; h *= 1099511628211ull; }
    return h;
}

/* Tables of SearchOptimalBlockPartitions (srla_encoder.c:336-389) for the windows of the plan's segments; all offsets are
 * relative to the job's input planes, so jobs of equal shape share them (key). */
void Impl::build_job(Job &job, const JobPlan &plan, const std::vector<uint32_t> &lshift, bool search, const std::vector<uint32_t> *lens)
{
    const uint32_t minb = par.min_num_samples_per_block, maxb = par.max_num_samples_per_block;
    const uint32_t window_len = search ? par.num_lookahead_samples : maxb;
    const uint32_t nv = num_variants(), pmax = preset_order();
    uint64_t key = 1469598103934665603ull;
    key = fnv64(key, ((uint64_t)param_generation << 2) | 2u | (search ? 1u : 0u));
    key = fnv64(key, plan.total);
    for (size_t k = 0; k < plan.segs.size(); k++) {
        key = fnv64(key, ((uint64_t)plan.segs[k].ns << 32) | plan.segs[k].base);
        key = fnv64(key, lshift[k]);
    }
    if (lens) key = 0;                 /* chain-mode jobs are never reused */
    if (!lens && job.key == key && key != 0 && job.segs.size() == plan.segs.size()) {
        /* same shape: only the identity of the segments' streams may differ */
        for (size_t k = 0; k < plan.segs.size(); k++) { job.segs[k].stream = plan.segs[k].stream; job.segs[k].s0 = plan.segs[k].s0; }
        return;
    }
    if (lens) search = false;
    job.key = key; job.uploaded = false;
    job.segs = plan.segs; job.seg_lshift = lshift; job.total = plan.total;
    job.windows.clear(); job.cands.clear(); job.items.clear(); job.groups.clear(); job.class_index.clear(); job.svr_rows.clear();
    job.seg_first_window.clear();
    job.num_slots = 0; job.res_elems = 0; job.analyzed_samples = 0; job.max_nodes = 2; job.max_window_cands = 1;
    job.keep_residuals = true;                  /* srla_pack_blocks reads the chosen blocks' residuals where srla_residual_cost left them */

    struct Pending { uint32_t cand; uint32_t nfft; uint32_t seg; };
    std::vector<Pending> analysed;
    for (uint32_t sg = 0; sg < plan.segs.size(); sg++) {
        const SegPlan &sp = plan.segs[sg];
        job.seg_first_window.push_back((uint32_t)job.windows.size());
        for (uint32_t pos = 0, wi = 0; pos < sp.ns; wi++) {
            const uint32_t wn = lens ? (*lens)[wi] : std::min(window_len, sp.ns - pos);
            SrlaWindowDesc wd{};
            wd.sample_off = sp.base + pos; wd.n = wn;
            wd.cand_base = (uint32_t)job.cands.size();
            wd.num_nodes = search ? ((wn + minb - 1) / minb + 1) : 2;
            wd.block_base = job.num_slots;
            wd.seg = sg;
            job.num_slots += wd.num_nodes - 1;
            job.max_nodes = std::max(job.max_nodes, wd.num_nodes);
            const uint32_t w = (uint32_t)job.windows.size();
            auto add_cand = [&](uint32_t i, uint32_t j, uint32_t off, uint32_t n) {
                SrlaCandDesc cd{};
                cd.window = w; cd.node_i = i; cd.node_j = j; cd.sample_off = sp.base + pos + off; cd.n = n;
                cd.item_base = 0xFFFFFFFFu;
                if (n > pmax) analysed.push_back({ (uint32_t)job.cands.size(), geoms[geom_for(n)].nfft, sg });
                job.cands.push_back(cd);
            };
            if (!search) add_cand(0, 1, 0, wn);
            else {
                for (uint32_t i = 0; i < wd.num_nodes; i++)
                    for (uint32_t j = i + 1; j < wd.num_nodes; j++) {
                        uint32_t len = (j - i) * minb;
                        if (len > maxb) continue;
                        const uint32_t off = i * minb;
                        len = std::min(len, wn - off);
                        add_cand(i, j, off, len);
                    }
            }
            wd.num_cands = (uint32_t)job.cands.size() - wd.cand_base;
            job.max_window_cands = std::max(job.max_window_cands, wd.num_cands);
            job.windows.push_back(wd);
            pos += wn;
        }
    }
    job.seg_first_window.push_back((uint32_t)job.windows.size());
    /* one launch analyses every item of the job: the LDS plan and the FFT register class are
     * those of the largest FFT present (smaller items simply leave part of them idle) */
    uint32_t max_nfft = 0;
    for (const Pending &p : analysed) max_nfft = std::max(max_nfft, p.nfft);
    if (!analysed.empty()) {
        Group g{};
        g.nfft = max_nfft;
        g.first = 0;
        g.rclass = (int)std::min(4u, std::max(1u, g.nfft / 2048u));     /* larger items go to srla_residual_cost_big */
        g.plan = lds_plan(std::min(g.nfft, 8192u));
        job.items.reserve(analysed.size() * nv);
        for (const Pending &p : analysed) {
            SrlaCandDesc &cd = job.cands[p.cand];
            cd.item_base = (uint32_t)job.items.size();
            const uint32_t gi = geom_for(cd.n);
            for (uint32_t v = 0; v < nv; v++) {
                SrlaItemDesc it{};
                it.sample_off = cd.sample_off; it.n = cd.n; it.variant = v;
                it.geom = gi;
                it.res_off = job.res_elems;
                it.forced_order = -1;
                it.lshift = lshift[p.seg];
                it.forced_ltp = 0;
                it.seg = p.seg;
                if (job.keep_residuals || cd.n > 8192u) job.res_elems += (cd.n + 3u) & ~3u;   /* see SrlaJobParams::keep_residuals */
                job.analyzed_samples += cd.n;
                job.items.push_back(it);
            }
        }
        g.count = (uint32_t)job.items.size();
        {
            /* LDS of the launch: when every item takes the register path of srla_residual_cost (blocks of 1024 * FL samples)
             * only that path's needs count -- the generic path's partition-mean tree would cost a workgroup per CU */
            bool all_fast = true;
            uint32_t fast_bytes = 0;
            for (const SrlaItemDesc &it : job.items) {
                const uint32_t fl = it.n >> 10;
                if ((it.n & 1023u) != 0 || fl < 1 || fl > 8 || fl > 2u * (uint32_t)g.rclass) { all_fast = false; break; }
                fast_bytes = std::max(fast_bytes, srla_kernel_fast_lds_bytes(fl, par.ltp_order, par.bits_per_sample));
            }
            if (all_fast) g.plan.total = fast_bytes;
            /* a job with blocks above 4096 samples: the smaller items in a launch of their own (srla_residual_cost<2>: 92 registers, five
             * wavefronts per SIMD, against 228 and two for the 8192-sample form), with their own LDS size */
            bool any_small = false, any_large = false;
            for (const SrlaItemDesc &it : job.items) { if (it.n <= 4096u) any_small = true; else if (it.n <= 8192u) any_large = true; }
            g.split = g.rclass == 4 && any_small && any_large;
            if (g.split) {
                g.plan_small = lds_plan(4096u);
                bool small_fast = true;
                uint32_t small_bytes = 0;
                for (const SrlaItemDesc &it : job.items) {
                    if (it.n > 4096u) continue;
                    const uint32_t fl = it.n >> 10;
                    if ((it.n & 1023u) != 0 || fl < 1) { small_fast = false; break; }
                    small_bytes = std::max(small_bytes, srla_kernel_fast_lds_bytes(fl, par.ltp_order, par.bits_per_sample));
                }
                if (small_fast) g.plan_small.total = small_bytes;
            }
        }
        job.groups.push_back(g);
    }
    job.big_items.clear(); job.big_max_n = 0;
    for (uint32_t i = 0; i < job.items.size(); i++)
        if (job.items[i].n > 8192u) { job.big_items.push_back(i); job.big_max_n = std::max(job.big_max_n, job.items[i].n); }
    job.class_index.clear();
    job.class_index.reserve(job.items.size());
    for (int c = 0; c < 8; c++) {
        job.class_first[c] = (uint32_t)job.class_index.size();
        for (uint32_t i = 0; i < job.items.size(); i++) {
            const SrlaItemDesc &it = job.items[i];
            const SrlaGeom &gm = geoms[it.geom];
            const uint32_t nfft = gm.nfft;
            const int cls = (nfft < 1024u) ? 0 : (nfft == 1024u) ? 6 : ((nfft <= 2048u) ? 1 : ((nfft <= 4096u) ? 2 : ((nfft <= 8192u) ? 3 : ((nfft <= 16384u) ? 4 : ((nfft <= 32768u) ? 5 : 7)))));
            if (cls == c) {
                SrlaAutocorrItem ai{};
                ai.item = i; ai.sample_off = it.sample_off; ai.n = it.n; ai.variant = it.variant;
                ai.nfft = gm.nfft; ai.tw_off = gm.tw_off; ai.welch_divisor = gm.welch_divisor; ai.acorr_norm = gm.acorr_norm;
                ai.lshift = it.lshift;
                job.class_index.push_back(ai);
            }
        }
        job.class_count[c] = (uint32_t)job.class_index.size() - job.class_first[c];
    }
}

SrlaJobParams Impl::job_params(const Job &job, uint32_t channel_stride, bool lshift_on_device) const
{
    SrlaJobParams jp{};
    jp.num_channels = par.num_channels;
    jp.bits_per_sample = par.bits_per_sample;
    jp.num_segs = (uint32_t)job.segs.size();
    jp.max_order = preset_order();
    jp.order_fixed = (par.preset == 0) ? 1u : 0u;
    jp.ltp_order = par.ltp_order;
    jp.num_samples = job.total;
    jp.channel_stride = channel_stride;
    jp.max_block = par.max_num_samples_per_block;
    jp.min_block = par.min_num_samples_per_block;
    jp.num_items = (uint32_t)job.items.size();
    jp.num_cands = (uint32_t)job.cands.size();
    jp.num_windows = (uint32_t)job.windows.size();
    jp.lshift_dev = lshift_on_device ? (d_or.as<uint32_t>() + 1) : nullptr;
    jp.welch_tab = (welch_bps == par.bits_per_sample && welch_table) ? d_welch.as<double>() : nullptr;
    jp.crowded = call_crowded ? 1u : 0u;
    jp.keep_residuals = job.keep_residuals ? (keep_residuals ? 2u : 1u) : 0u;
    jp.tie_rel = tie_rel; jp.tie_ltp = tie_ltp; jp.tie_logscale = tie_logscale; jp.tie_ltpbias = tie_ltpbias;
    return jp;
}

/* windows per job: bounded by scratch memory (~1.5 GB of residual scratch per slot) */
uint32_t Impl::windows_per_job(bool search) const
{
    const uint32_t window_len = search ? par.num_lookahead_samples : par.max_num_samples_per_block;
    uint64_t per_window = (uint64_t)window_len * num_variants() * 4;
    if (search) {
        const uint32_t ratio = par.max_num_samples_per_block / par.min_num_samples_per_block;
        per_window *= ratio;
    }
    uint64_t w = (1536ull << 20) / std::max<uint64_t>(per_window, 1);
    if (search) {
        /* ... and by the item tables (1.4 KB of result record and up to 2 KB of lags per item): about a million items per job -- only
         * parameters with hundreds of minimum blocks per window get there (`-V 6`: 57 000 items per window) */
        const uint64_t nodes = par.num_lookahead_samples / par.min_num_samples_per_block;
        const uint64_t ratio = par.max_num_samples_per_block / par.min_num_samples_per_block;
        w = std::min<uint64_t>(w, std::max<uint64_t>(1, (1ull << 20) / std::max<uint64_t>(1, nodes * ratio * num_variants())));
    }
    const uint64_t cap_samples = job_samples;  /* several jobs per stream so that the GPU and the host pack overlap */
    w = std::min<uint64_t>(w, std::max<uint64_t>(1, cap_samples / window_len));
    return (uint32_t)std::max<uint64_t>(1, w);
}

/* The jobs of a call.  One stream: full jobs rotate through the kSlots buffer sets; what is left at the end of the stream
 * is cut once more so that the LAST job is small -- after it nothing else runs on the wide stream, so its pricing, block
 * assembly and stream-out are pure latency (0.27 ms for a full job, 8 % of a 600 s stream's time); the two tail jobs have
 * buffer sets of their own, so that repeated calls of equal length keep finding their descriptor tables cached.
 * Several streams: whole jobs first, then the remainders packed greedily in stream order (a remainder is cut at a window
 * boundary when the job is full); segments start on multiples of 16 samples of the job's planes (aligned 16-byte loads,
 * 32-byte aligned staging stores). */
void Impl::plan_jobs(std::vector<JobPlan> &plan, bool search)
{
    plan.clear();
    planned_pieces = false;
    const uint32_t window_len = search ? par.num_lookahead_samples : par.max_num_samples_per_block;
    const uint64_t job_len = (uint64_t)windows_per_job(search) * window_len;
    auto al16 = [](uint64_t v) { return (uint32_t)((v + 15u) & ~(uint64_t)15u); };
    if (sx.size() == 1) {
        const uint32_t body = sx[0].body;
        uint64_t nfull = body / job_len, rest = body - nfull * job_len;
        if (rest == 0 && nfull > 0) { nfull--; rest = job_len; }
        /* A stream of one job and a bit (up to `mid_jobs` jobs): as a whole job plus tail it starts with 0.55 ms of staging and
         * upload and ends with a full job's chain of stages; in pieces like a short stream the device starts after one piece's
         * staging and the last chain is a piece's.  Measured: 120 s of stereo 3 300 -> 3 560 Msamples/s; with TWO whole jobs
         * (200 s) the pieces lost in round 4 (4 500 -> 4 040: the host's staging of piece after piece set the pace) and win since the
         * pieces run on streams of their own (round 5: 4 630 -> 4 860); with three (300 s) they lose 2 %: hence 2. */
        /* (only where the pieces branch below takes the stream: otherwise it would fall through to ONE job of up to twice the
         * planned job length, which the buffer and LDS planning do not assume) */
        /* (a stream with a history-dependent last window keeps the cross-stream pipeline, Slot::piece: one whole job at most) */
        const uint32_t mid = (sx[0].chain_n != 0) ? std::min<uint32_t>(mid_jobs, 1u) : mid_jobs;
        if (nfull > 0 && nfull <= mid && job_len >= 8 * (uint64_t)window_len && body >= 2 * (uint64_t)short_min) { nfull = 0; rest = body; }
        auto one = [&](uint32_t s0, uint32_t ns, uint32_t slot_index) {
            JobPlan jp; jp.segs.push_back({ 0u, s0, ns, 0u }); jp.total = al16(ns); jp.slot = slot_index; plan.push_back(jp);
        };
        for (uint64_t k = 0; k < nfull; k++) one((uint32_t)(k * job_len), (uint32_t)job_len, (uint32_t)(k % kSlots));
        const uint64_t small = (uint64_t)std::max<uint32_t>(1u, 262144u / window_len) * window_len;
        const uint32_t tail0 = (uint32_t)(nfull * job_len);
        if (nfull > 0 && rest > 2 * small) {
            const uint32_t first = (uint32_t)(((rest - small) / window_len) * window_len);
            one(tail0, first, kSlots);
            one(tail0 + first, (uint32_t)(rest - first), kSlots + 1);
        } else if (nfull == 0 && rest >= 2 * (uint64_t)short_min && job_len >= 8 * (uint64_t)window_len) {
            /* a stream shorter than one job: as one job its five stages would run one after the other on an otherwise idle
             * GPU; in a few pieces they overlap (60 s of stereo: 1.15 -> 1.03 ms).  About three pieces: more cost more host
             * time per piece than their overlap returns (60 s in 6 pieces: 2 670 instead of 3 100 Msamples/s) */
            const uint64_t part = job_len / std::max<uint32_t>(2u, short_div);
            uint64_t pieces = (rest + part - 1) / part;
            pieces = std::max<uint64_t>(pieces, std::min<uint64_t>(3u, rest / short_min));
            const uint64_t piece = (((rest + pieces - 1) / pieces + window_len - 1) / window_len) * window_len;
            uint32_t k = 0;
            /* (the last piece, of a length of its own, in the first tail job's buffer set: where the pieces outnumber the rotating
             * sets it would otherwise take turns with a whole piece's table in one of them, and both would be built again every call) */
            for (uint64_t s0 = 0; s0 < rest; s0 += piece, k++)
                one((uint32_t)s0, (uint32_t)std::min<uint64_t>(piece, rest - s0), (s0 + piece >= rest && pieces > kSlots) ? kSlots : k % kSlots);
            planned_pieces = true;
        } else if (rest > 0) {
            one(tail0, (uint32_t)rest, nfull > 0 ? kSlots : 0u);
        }
        return;
    }
    /* First the parts of the streams that fill whole jobs -- one segment of job_len samples each: jobs of ONE shape, so after
     * the first few every job finds its descriptor tables built and uploaded --, then what is left of every stream, packed
     * together.  A stream's segments stay in order, which is all its device-resident output position needs. */
    for (uint32_t i = 0; i < sx.size(); i++) {
        const uint64_t nfull = sx[i].body / job_len;
        for (uint64_t k = 0; k < nfull; k++) {
            JobPlan jp;
            jp.segs.push_back({ i, (uint32_t)(k * job_len), (uint32_t)job_len, 0u });
            jp.total = al16(job_len);
            jp.slot = (uint32_t)(plan.size() % kSlots);
            plan.push_back(jp);
        }
    }
    /* The jobs of remainders rotate through buffer sets of their own (kSlots .. 2 kSlots - 1, where there is room for them below the chain-mode sets): in the
     * whole jobs' sets each of them threw out the ONE table every whole job shares, and the next call's first jobs -- the very jobs
     * the idle device waits for -- built theirs again (1.5 - 2.3 ms of host time each at -V 2 -P 3; C5 1 916 -> 1 977, nine unequal files
     * 1 700 -> 1 776 Msamples/s: profiles/r05/ab_host_path.txt). */
    /* (as many of them as there are rotating sets: a set is then taken again kSlots jobs later, which is what the pipeline's
     * run-ahead -- lag + 1 <= kSlots, host_pipeline.cpp -- needs for ANY kSlots; with a fixed five, SRLA_MI355X_SLOTS=6 let remainder
     * job r + 5 restage the set of job r while r was still in flight.  Where they do not fit below the chain-mode sets the
     * remainders take turns with the whole jobs as before.) */
    const uint32_t rem_sets = (kSlots + kSlots <= kChainSlot) ? kSlots : 0u;
    uint32_t rem_index = 0;
    JobPlan cur;
    auto close = [&]() {
        if (cur.segs.empty()) return;
        cur.slot = rem_sets ? kSlots + (rem_index++ % rem_sets) : (uint32_t)(plan.size() % kSlots);
        plan.push_back(cur);
        cur = JobPlan();
    };
    for (uint32_t i = 0; i < sx.size(); i++) {
        const uint64_t nfull = sx[i].body / job_len;
        uint32_t s0 = (uint32_t)(nfull * job_len), remaining = sx[i].body - s0;
        while (remaining > 0) {
            const uint64_t room = (job_len > cur.total) ? job_len - cur.total : 0;
            uint32_t take = remaining;
            if (take > room) take = (uint32_t)((room / window_len) * window_len);
            if (take == 0) { close(); continue; }
            cur.segs.push_back({ i, s0, take, cur.total });
            cur.total = al16((uint64_t)cur.total + take);
            s0 += take; remaining -= take;
            if (cur.total >= job_len) close();
        }
    }
    close();
}
