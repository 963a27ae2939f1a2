/*
 * host_chain.cpp -- chain mode: the history-dependent last window of a stream (see host_impl.h, struct ChainCall).
 */
#include "host_impl.h"

#include <algorithm>
#include <stdlib.h>
#include <string.h>

void Impl::chain_append(uint32_t jobidx, const Job &job, const std::function<bool(uint32_t, uint32_t)> &silent)
{
    const uint32_t nch = par.num_channels, nv = num_variants();
    std::vector<uint32_t> pass1_round(job.items.size(), 0);
    std::vector<uint8_t> pass1_tainted(job.items.size(), 0);
    for (const SrlaCandDesc &cd : job.cands) {
        if (cd.item_base == 0xFFFFFFFFu || silent(cd.sample_off, cd.n)) continue;   /* RAW by length / SILENT: no analysis (srla_encoder.c:766-796) */
        for (uint32_t k = 0; k < nv; k++) {
            const uint32_t v = (nch >= 2) ? ((k < 2) ? nch + k : k - 2) : k;        /* M, S, then the channels (srla_encoder.c:1229-1275) */
            const uint32_t item = cd.item_base + v;
            for (int pass = (par.ltp_order > 0) ? 1 : 0; pass >= 0; pass--) {
                ChainCall c{};
                c.job = jobidx; c.item = item; c.pass = (uint32_t)pass; c.n = cd.n; c.nfft = geoms[job.items[item].geom].nfft;
                c.src = -1; c.dump = (uint32_t)chain_pool_used; chain_pool_used += c.nfft;
                uint32_t round = (pass == 0 && par.ltp_order > 0) ? pass1_round[item] : 0;
                /* (the item's LPC-lag call analyses what its long-term predictor left: it is no better known than that) */
                c.tainted = (pass == 0 && par.ltp_order > 0) ? (pass1_tainted[item] != 0) : false;
                /* a word read from the buffer itself (chain_calls[0] in history mode) is known as far as buf_exact reaches */
                auto unknown = [&](const ChainCall &sc, uint32_t upto) {
                    return sc.job == 0xFFFFFFFFu ? (upto > buf_exact) : sc.tainted;
                };
                if (c.n & 1u) {
                    const uint32_t mid = c.n >> 1;
                    for (int64_t j = (int64_t)chain_calls.size() - 1; j >= 0; j--)
                        if (chain_calls[(size_t)j].nfft > mid) { c.src = (int32_t)j; break; }
                    if (c.src >= 0 && unknown(chain_calls[(size_t)c.src], mid + 1u)) c.tainted = true;
                    if (c.src >= 0 && chain_calls[(size_t)c.src].job == jobidx) {
                        /* inside a round the LTP-lag launches come first, then the pitch solve, then the LPC-lag launches
                         * (and, with SVR on, the solve chain and the refinement: pass 2) */
                        const ChainCall &sc = chain_calls[(size_t)c.src];
                        const uint32_t need = (pass == 0 && sc.pass == 1) ? sc.round : sc.round + 1;
                        round = std::max(round, need);
                    }
                }
                if (pass == 1 && c.nfft < SRLA_LTP_LAGS) {
                    /* lpc.c:371-373 copies 263 lags out of a shorter FFT buffer: the words from nfft on are those
                     * of the latest earlier calls that reached them (zero when none did) */
                    c.lags = (uint32_t)chain_tab.size() + 1u;
                    const size_t base = chain_tab.size();
                    chain_tab.resize(base + (SRLA_LTP_LAGS - c.nfft), 0u);
                    uint32_t lo = c.nfft;
                    for (int64_t j = (int64_t)chain_calls.size() - 1; j >= 0 && lo < SRLA_LTP_LAGS; j--) {
                        const ChainCall &sc = chain_calls[(size_t)j];
                        if (sc.nfft <= lo) continue;
                        const uint32_t hi = std::min<uint32_t>(sc.nfft, SRLA_LTP_LAGS);
                        for (uint32_t i = lo; i < hi; i++) chain_tab[base + (i - c.nfft)] = sc.dump + i + 1u;
                        if (unknown(sc, hi)) c.tainted = true;
                        lo = hi;
                        if (sc.job == jobidx) round = std::max(round, sc.round + 1);
                    }
                }
                c.round = round;
                if (pass == 1) { pass1_round[item] = round; pass1_tainted[item] = c.tainted ? 1 : 0; }
                if (c.tainted) call_tainted = true;
                chain_calls.push_back(c);
                if (pass == 0 && chain_svr()) {
                    /* LPCCalculator_CalculateLPCCoefficientsSVR keeps its `residual` in the same persistent buffer
                     * (lpc.c:1047): after the item's analysis the first n words are the refinement's (or, where it does not
                     * run, still the transform's: the kernel copies them) -- a writer of n words in the sequence of calls */
                    ChainCall v{};
                    v.job = jobidx; v.item = item; v.pass = 2; v.n = cd.n; v.nfft = cd.n; v.src = (int32_t)chain_calls.size() - 1;
                    v.dump = (uint32_t)chain_pool_used; chain_pool_used += (cd.n + 1u) & ~1u;
                    v.round = round; v.tainted = c.tainted;
                    chain_calls.push_back(v);
                }
            }
        }
    }
}

void Impl::chain_build(uint32_t jobidx, Job &job, ChainJob &cj)
{
    struct Entry { uint32_t round, pass, cls; SrlaAutocorrItem ai; };
    std::vector<Entry> entries;
    const bool ltp = par.ltp_order > 0;
    std::vector<uint8_t> seen(job.items.size(), 0);
    auto make = [&](uint32_t item) {
        const SrlaItemDesc &it = job.items[item];
        const SrlaGeom &gm = geoms[it.geom];
        SrlaAutocorrItem ai{};
        ai.item = item; ai.sample_off = it.sample_off; ai.n = it.n; ai.variant = it.variant;
        ai.nfft = gm.nfft; ai.tw_off = gm.tw_off; ai.welch_divisor = gm.welch_divisor; ai.acorr_norm = gm.acorr_norm;
        ai.lshift = it.lshift;
        return ai;
    };
    auto cls_of = [](uint32_t nfft) { return (nfft <= 1024u) ? 0u : ((nfft <= 2048u) ? 1u : ((nfft <= 4096u) ? 2u : ((nfft <= 8192u) ? 3u : ((nfft <= 16384u) ? 4u : ((nfft <= 32768u) ? 5u : 6u))))); };
    cj.select.assign(std::max<size_t>(1, job.items.size()), 0xFFFFFFFFu);
    cj.select_b.assign(std::max<size_t>(1, job.items.size()), 0u);
    cj.rounds = 1;
    for (SrlaItemDesc &it : job.items) { it.svr_dump = 0; it.svr_under = 0; }
    for (const ChainCall &c : chain_calls) {
        if (c.job != jobidx) continue;
        if (c.pass == 2) {
            /* the refinement's place in the sequence: where the item leaves its n words, and what lies under them */
            job.items[c.item].svr_dump = c.dump + 1u;
            job.items[c.item].svr_under = chain_calls[(size_t)c.src].dump + 1u;
            cj.select_b[c.item] = c.round;
            continue;
        }
        SrlaAutocorrItem ai = make(c.item);
        ai.chain_dump = c.dump + 1u;
        ai.chain_lags = c.lags;
        if (c.src >= 0) ai.chain_src = chain_calls[(size_t)c.src].dump + (c.n >> 1) + 1u;
        entries.push_back({ c.round, c.pass, cls_of(ai.nfft), ai });
        if (c.pass == 1 || !ltp) cj.select[c.item] = c.round;
        seen[c.item] = 1;
        cj.rounds = std::max(cj.rounds, c.round + 1);
    }
    /* items of silent blocks: no call of the reference, but their records are still initialised by the kernel */
    for (uint32_t i = 0; i < job.items.size(); i++)
        if (!seen[i]) {
            const SrlaAutocorrItem ai = make(i);
            for (int pass = ltp ? 1 : 0; pass >= 0; pass--) entries.push_back({ 0u, (uint32_t)pass, cls_of(ai.nfft), ai });
            cj.select[i] = 0;
        }
    /* one launch per round and pass, instantiated for the longest FFT among its items (shorter ones leave part of
     * the workgroup idle): the launches are few and dependent, their number is what costs */
    std::stable_sort(entries.begin(), entries.end(), [](const Entry &a, const Entry &b) {
        if (a.round != b.round) return a.round < b.round;
        return a.pass > b.pass;
    });
    cj.list.clear(); cj.launches.clear();
    for (const Entry &e : entries) {
        if (cj.launches.empty() || cj.launches.back().round != e.round || cj.launches.back().pass != e.pass)
            cj.launches.push_back({ e.round, e.pass, e.cls, (uint32_t)cj.list.size(), 0u });
        cj.launches.back().count++;
        cj.launches.back().cls = std::max(cj.launches.back().cls, e.cls);
        cj.list.push_back(e.ai);
    }
}

bool Impl::chain_stage_a(Slot &s, uint32_t jobidx, const ChainJob &cj)
{
    /* many small dependent launches: on a stream of their own, so that the regular jobs' wide kernels do not queue
     * behind them */
    hipStream_t W = s.own_stream;
    s.ties_gathered = false;
    const SrlaJobParams &jp = s.jp;
    if (!d_chain_list[jobidx].ensure(std::max<size_t>(1, cj.list.size()) * sizeof(SrlaAutocorrItem))) return false;
    if (!d_chain_select[jobidx].ensure(cj.select.size() * 4)) return false;
    const bool svr_rounds = chain_svr() && !s.job.groups.empty() && jp.max_order > 0;
    if (svr_rounds && !d_chain_select_b[jobidx].ensure(cj.select_b.size() * 4)) return false;
    {
        /* the job's tables go up in ONE asynchronous copy each, out of a pinned staging area of the job's own, on the stream their
         * readers run on (four blocking copies out of pageable memory cost 0.11 ms per stage -- a third of a block call).  The area
         * is written again only after the job's events have been waited for (a phase ends with a host round trip). */
        const size_t list_b = cj.list.size() * sizeof(SrlaAutocorrItem), sel_b = cj.select.size() * 4, selb_b = svr_rounds ? cj.select_b.size() * 4 : 0;
        const size_t tab_new = chain_tab.size() - std::min(chain_tab.size(), chain_tab_uploaded), tab_b = tab_new * 4;
        if (chain_tab.size() * 4 > d_chain_tab.cap) return false;
        PinBuf &h = h_chain_up[jobidx];
        if (!h.ensure(list_b + sel_b + selb_b + tab_b + 64)) return false;
        uint8_t *hp = h.as<uint8_t>();
        if (list_b) { memcpy(hp, cj.list.data(), list_b); HIP_OK(hipMemcpyAsync(d_chain_list[jobidx].p, hp, list_b, hipMemcpyHostToDevice, W)); }
        hp += list_b;
        memcpy(hp, cj.select.data(), sel_b);
        HIP_OK(hipMemcpyAsync(d_chain_select[jobidx].p, hp, sel_b, hipMemcpyHostToDevice, W));
        hp += sel_b;
        if (selb_b) { memcpy(hp, cj.select_b.data(), selb_b); HIP_OK(hipMemcpyAsync(d_chain_select_b[jobidx].p, hp, selb_b, hipMemcpyHostToDevice, W)); }
        hp += selb_b;
        if (tab_b) {
            /* only the new entries: kernels of the jobs before may still be reading theirs */
            memcpy(hp, chain_tab.data() + chain_tab_uploaded, tab_b);
            HIP_OK(hipMemcpyAsync(d_chain_tab.as<uint32_t>() + chain_tab_uploaded, hp, tab_b, hipMemcpyHostToDevice, W));
            chain_tab_uploaded = chain_tab.size();
        }
    }
    /* prepare_job put the descriptor uploads on the wide stream: everything after this stage must see them */
    HIP_OK(hipEventRecord(s.t0[ST_A], streams[0]));
    HIP_OK(hipStreamWaitEvent(W, s.t0[ST_A], 0));
    if (jp.lshift_dev != nullptr) HIP_OK(hipStreamWaitEvent(W, ev_or, 0));
    if (s.used_h2d) HIP_OK(hipStreamWaitEvent(W, s.ev_in, 0));
    static const int kClass[4] = { 0, 1, 2, 4 };
    int rc = 0;
    size_t li = 0;
    s.b_done = false;
    for (uint32_t r = 0; r < cj.rounds; r++) {
        for (int pass = (par.ltp_order > 0) ? 1 : 0; pass >= 0; pass--) {
            bool any = false;
            for (; li < cj.launches.size() && cj.launches[li].round == r && cj.launches[li].pass == (uint32_t)pass; li++) {
                const ChainLaunch &l = cj.launches[li];
                if (l.cls >= 4)
                    rc |= srla_launch_autocorr_big(W, &jp, s.in_cur, d_tw.p, (uint32_t)pass, s.d_results.as<SrlaItemResult>(), s.d_lags.as<double>(), nullptr,
                                                   d_chain_list[jobidx].as<SrlaAutocorrItem>() + l.first, l.count, l.cls == 4 ? 16384u : (l.cls == 5 ? 32768u : 65536u), nullptr, nullptr,
                                                   d_chain_pool.as<double>(), d_chain_tab.as<uint32_t>(), s.d_big_scratch.p, SRLA_BIG_GROUPS);
                else
                rc |= srla_launch_autocorr(W, kClass[l.cls], &jp, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), d_tw.p,
                                           (uint32_t)pass, s.d_results.as<SrlaItemResult>(), s.d_lags.as<double>(), nullptr,
                                           d_chain_list[jobidx].as<SrlaAutocorrItem>() + l.first, l.count, nullptr, nullptr,
                                           d_chain_pool.as<double>(), d_chain_tab.as<uint32_t>(), 0);
                any = true;
            }
            if (pass == 1 && any)
                rc |= srla_launch_pitch_solve(W, &jp, s.d_items.as<SrlaItemDesc>(), s.d_lags.as<double>(), s.d_results.as<SrlaItemResult>(), nullptr, nullptr,
                                              d_chain_select[jobidx].as<uint32_t>(), r, s.d_ties.as<uint32_t>(), s.d_tie_data.as<double>());
        }
        if (svr_rounds) {
            /* SVR on: what a later call inherits may be this round's refinements (lpc.c:1047), so the solve chain of the round's
             * items runs here, between the rounds of autocorrelations, instead of once for the whole job */
            const SrlaSvrExtra ex = { s.d_ties.as<uint32_t>(), s.job.svr_rows.empty() ? nullptr : s.d_svr_rows.as<double>(), d_chain_pool.as<double>(),
                                      d_chain_select_b[jobidx].as<uint32_t>(), r };
            rc |= srla_launch_lpc_solve(W, &jp, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), s.d_lags.as<double>(), s.d_err.as<double>(),
                                        d_huff.as<uint8_t>(), s.d_results.as<SrlaItemResult>(), nullptr, s.d_ties.as<uint32_t>(), nullptr, nullptr,
                                        s.in_cur, s.d_coef_ws.as<double>(), par.num_svr_filter_learning_iteration,
                                        std::min<uint32_t>(par.max_num_samples_per_block, 8192u), d_svr_scratch_chain.p, kSvrGroups, s.d_gamma.as<double>(), &ex);
        }
    }
    s.b_done = svr_rounds;
    HIP_OK(hipEventRecord(s.t1[ST_A], W));
    if (rc != 0) { fprintf(stderr, "[srla-mi355x] kernel launch failed in a chain stage\n"); return false; }
    return true;
}

bool Impl::chain_silent(const std::vector<int32_t> &v, uint32_t total, uint32_t off, uint32_t n) const
{
    for (uint32_t ch = 0; ch < par.num_channels; ch++) {
        const int32_t *p = v.data() + (size_t)ch * total + off;
        for (uint32_t i = 0; i < n; i++) if (p[i] != 0) return false;
    }
    return true;
}

void Impl::chain_slot_defaults(Slot &s)
{
    s.own_stream = streams[1];   /* the narrow stream: a stream of their own ended up sharing a hardware queue with the wide one and waited for the whole stream (measured) */
    s.timed = false; s.out_boost = 1; s.emits = false; s.merge_cb = false;
}

static JobPlan one_segment(uint32_t stream, uint32_t s0, uint32_t ns)
{
    JobPlan p;
    p.segs.push_back({ stream, s0, ns, 0u });
    p.total = (ns + 15u) & ~15u;
    return p;
}

/* samples -> device, offset shift settled, tables built (stage_input before build_job: the tables carry the shift) */
bool Impl::chain_make_job(Slot &s, uint32_t s0, uint32_t ns, bool search, const std::vector<uint32_t> *lens)
{
    const JobPlan p = one_segment(chain.stream, s0, ns);
    if (!stage_input(s, p)) return false;
    std::vector<uint32_t> lsh;
    settle_lshift(p, lsh);
    build_job(s.job, p, lsh, search, lens);
    if (sx[chain.stream].raw_below_shift) mark_raw_silence(s.job);
    return true;
}

/* The pool's head IS the handle's buffer after a tracked history-mode call (no copy per call); whoever is about to write there --
 * chain mode lays its calls out from word 0, a batch's streams start from zeros, a growing pool moves -- puts it into d_hist first. */
bool Impl::hist_leave_pool()
{
    if (!pool_holds_hist) return true;
    pool_holds_hist = false;
    if (!d_hist.ensure((size_t)kHistoryWords * sizeof(double))) return false;
    if (hipMemcpyAsync(d_hist.p, d_chain_pool.p, (size_t)kHistoryWords * sizeof(double), hipMemcpyDeviceToDevice, streams[1]) != hipSuccess) return false;
    return hipStreamSynchronize(streams[1]) == hipSuccess;     /* (rare: the writers that follow are on other streams or free the pool) */
}

bool Impl::chain_begin(uint32_t seed_off, uint32_t seed_n)
{
    ChainRun &c = chain;
    if (!hist_leave_pool()) return false;
    const StreamCtx &st = sx[c.stream];
    const uint32_t nch = par.num_channels, nv = num_variants(), passes = par.ltp_order > 0 ? 2u : 1u;
    /* which blocks are all zero decides which calls exist: look at the samples */
    auto fetch = [&](uint32_t off, uint32_t n, std::vector<int32_t> &dst) -> bool {
        dst.resize((size_t)nch * n);
        for (uint32_t ch = 0; ch < nch; ch++) {
            if (st.pcm) (void)pcm_channel(st.pcm, st.pcm_bytes, nch, ch, off, n, dst.data() + (size_t)ch * n);
            else if (st.host_in) memcpy(dst.data() + (size_t)ch * n, st.host_in[ch] + off, (size_t)n * 4);
            else if (!d2h(dst.data() + (size_t)ch * n, st.d_in + (size_t)ch * st.d_stride + off, (size_t)n * 4)) return false;
        }
        return true;
    };
    c.seed_n = seed_n; c.seed_off = seed_off;
    if (!fetch(c.tail_start, c.tail_n, c.tail_smp) || (seed_n && !fetch(seed_off, seed_n, c.seed_smp))) return false;
    const std::function<bool(uint32_t, uint32_t)> silent_tail = [&](uint32_t off, uint32_t n) { return chain_silent(c.tail_smp, c.tail_n, off, n); };
    const std::function<bool(uint32_t, uint32_t)> silent_seed = [&](uint32_t off, uint32_t n) { return chain_silent(c.seed_smp, c.seed_n, off, n); };
    chain_calls.clear();
    chain_pool_used = 0;
    chain_tab.clear();
    chain_tab_uploaded = 0;
    Slot &q = slot[kChainSlot], &sj = slot[kChainSlot + 1], &e = slot[kChainSlot + 2];
    chain_slot_defaults(q); chain_slot_defaults(sj); chain_slot_defaults(e);
    if (seed_n) {
        const std::vector<uint32_t> lens{ seed_n };
        if (!chain_make_job(q, seed_off, seed_n, false, &lens)) return false;
        (void)apply_overrides(q.job, kChainJobKey + 0);
        chain_append(0, q.job, silent_seed);
    }
    if (c.search) {
        if (!chain_make_job(sj, c.tail_start, c.tail_n, true, nullptr)) return false;
        sj.job.key = 0;              /* never mistaken for a regular job's tables */
        (void)apply_overrides(sj.job, kChainJobKey + 1);
        chain_append(1, sj.job, silent_tail);
    } else {
        const std::vector<uint32_t> lens{ c.tail_n };
        if (!chain_make_job(e, c.tail_start, c.tail_n, false, &lens)) return false;
        (void)apply_overrides(e.job, kChainJobKey + 2);
        chain_append(2, e.job, silent_tail);
    }
    {
        /* the encode job's calls are not known yet when searching: its blocks tile the window, an FFT is shorter
         * than twice its block (or the smallest FFT size) */
        const uint32_t max_parts = c.search ? (c.tail_n + par.min_num_samples_per_block - 1) / par.min_num_samples_per_block : 0u;
        const uint64_t bound = chain_pool_used + (uint64_t)nv * (passes + (chain_svr() ? 1u : 0u)) * (2ull * c.tail_n + 64ull * max_parts);
        if (!d_chain_pool.ensure(bound * sizeof(double))) return false;
        const size_t tab_bound = chain_tab.size() + (size_t)std::max(1u, max_parts) * nv * SRLA_LTP_LAGS;
        if (!d_chain_tab.ensure(tab_bound * 4)) return false;
    }
    if (seed_n) {
        chain_build(0, q.job, c.cq);
        q.job.uploaded = false;
        if (!prepare_job(q, false) || !chain_stage_a(q, 0, c.cq)) return false;
    }
    if (c.search) {
        chain_build(1, sj.job, c.cs);
        sj.job.uploaded = false;
        if (!prepare_job(sj, false) || !chain_stage_a(sj, 1, c.cs)) return false;
        for (int st2 = ST_B; st2 <= ST_D; st2++) if (!run_stage(sj, st2)) return false;
    }
    c.begun = true;
    return true;
}

/* Near-ties of the seed and search jobs (they never reach the block assembly, which is where a regular job's tie count is
 * reported): decided with the host libm once the search job has been priced; a job the host decides differently is run again. */
bool Impl::chain_settle_ties()
{
    ChainRun &c = chain;
    Slot &q = slot[kChainSlot], &sj = slot[kChainSlot + 1];
    for (int attempt = 0; attempt < 6; attempt++) {
        if (c.seed_n && hipEventSynchronize(q.t1[ST_A]) != hipSuccess) return false;
        const int mq = c.seed_n ? arbitrate(q, kChainJobKey + 0) : 0;
        const int ms = c.search ? arbitrate(sj, kChainJobKey + 1) : 0;
        if (mq < 0 || ms < 0) return false;
        if (mq == 0 && ms == 0) return true;
        stats.num_restarts++;
        if (mq > 0) {
            (void)apply_overrides(q.job, kChainJobKey + 0);
            q.job.uploaded = false;
            if (!prepare_job(q, false) || !chain_stage_a(q, 0, c.cq)) return false;
        }
        if (c.search) {
            /* the window's calls inherit from the seed's: the search job follows in any case */
            (void)apply_overrides(sj.job, kChainJobKey + 1);
            sj.job.uploaded = false;
            if (!prepare_job(sj, false) || !chain_stage_a(sj, 1, c.cs)) return false;
            for (int st2 = ST_B; st2 <= ST_D; st2++) if (!run_stage(sj, st2)) return false;
            if (hipEventSynchronize(sj.t1[ST_D]) != hipSuccess) return false;
        }
    }
    fprintf(stderr, "[srla-mi355x] internal error: near-tie arbitration of the chain-mode window did not settle\n");
    return false;
}

/* the encode job up to its pricing */
bool Impl::chain_encode_ad()
{
    ChainRun &c = chain;
    Slot &q = slot[kChainSlot], &sj = slot[kChainSlot + 1], &e = slot[kChainSlot + 2];
    if (c.search) {
        const auto tw = Clock::now();
        if (hipEventSynchronize(sj.t1[ST_D]) != hipSuccess) return false;
        if (chain_trace) fprintf(stderr, "[chain] waited %.3f ms for the search job (%u rounds)\n", ms_since(tw), c.cs.rounds);
    }
    if (!chain_settle_ties()) return false;
    if (c.search) {
        const SrlaWindowDesc &wd = sj.job.windows[0];
        std::vector<SrlaBlockRecord> recs(wd.num_nodes - 1);
        if (!d2h(recs.data(), sj.d_blocks.as<SrlaBlockRecord>() + wd.block_base, recs.size() * sizeof(SrlaBlockRecord)))
            return false;
        std::vector<uint32_t> lens;
        uint32_t covered = 0;
        for (const SrlaBlockRecord &r : recs) if (r.valid) { lens.push_back(r.n); covered += r.n; }
        if (covered != c.tail_n) { fprintf(stderr, "[srla-mi355x] internal error: the tail window's partitions cover %u of %u samples\n", covered, c.tail_n); return false; }
        sj.busy = false;
        const std::function<bool(uint32_t, uint32_t)> silent_tail = [&](uint32_t off, uint32_t n) { return chain_silent(c.tail_smp, c.tail_n, off, n); };
        if (!chain_make_job(e, c.tail_start, c.tail_n, false, &lens)) return false;
        (void)apply_overrides(e.job, kChainJobKey + 2);
        chain_append(2, e.job, silent_tail);
        if (chain_pool_used * sizeof(double) > d_chain_pool.cap) return false;
    }
    q.busy = false;
    chain_build(2, e.job, c.ce);
    e.out_boost = kTailBoost; e.emits = true; e.merge_cb = true;
    e.job.uploaded = false;
    if (!prepare_job(e, false) || !chain_stage_a(e, 2, c.ce)) return false;
    for (int st2 = ST_B; st2 <= ST_D; st2++) if (!run_stage(e, st2)) return false;
    c.ad_done = true;
    return true;
}

bool Impl::chain_encode_e() { return run_stage(slot[kChainSlot + 2], ST_E); }

/* has the search job been priced (so that the encode job can be enqueued without waiting)? */
bool Impl::chain_search_done()
{
    if (!chain.begun) return false;
    if (!chain.search) return true;
    const hipError_t e = hipEventQuery(slot[kChainSlot + 1].t1[ST_D]);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess;
}

/* waits for the encode job, settles its near-ties (running it again where the host libm decides otherwise) and collects it */
SRLAApiResult Impl::chain_collect()
{
    Slot &e = slot[kChainSlot + 2];
    for (int attempt = 0;; attempt++) {
        if (!wait_job(e)) return SRLA_APIRESULT_NG;
        if (e.h_info.as<SrlaJobInfo>()->num_tie_items == 0) break;
        const int m = arbitrate(e, kChainJobKey + 2);
        if (m < 0) return SRLA_APIRESULT_NG;
        if (m == 0) break;
        if (attempt >= 6) { fprintf(stderr, "[srla-mi355x] internal error: near-tie arbitration of the chain-mode window did not settle\n"); return SRLA_APIRESULT_NG; }
        stats.num_restarts++;
        /* once more, from the place the host knows the stream has reached (every job before this one has been collected) */
        sx[chain.stream].pass_started = false;
        (void)apply_overrides(e.job, kChainJobKey + 2);
        e.job.uploaded = false;
        if (!prepare_job(e, false) || !chain_stage_a(e, 2, chain.ce)) return SRLA_APIRESULT_NG;
        for (int st2 = ST_B; st2 <= ST_E; st2++) if (!run_stage(e, st2)) return SRLA_APIRESULT_NG;
    }
    return finish_job(e);
}

/* ================================================================================================================
 * History mode (host_impl.h): every window of the stream in the reference's own call order.
 * ============================================================================================================== */

bool Impl::history_regime(bool search) const
{
    if (no_chain) return false;
    const uint32_t grid = search ? par.min_num_samples_per_block : par.max_num_samples_per_block;
    if (grid & 1u) return true;                                   /* odd blocks anywhere: lpc.c:260-264 */
    if (par.ltp_order > 0 && grid <= 256u) return true;           /* blocks shorter than the 263 LTP lags anywhere: lpc.c:371-373 */
    return false;
}

/* a new phase: its calls see the buffer (chain_calls[0]) and each other; the pool behind the buffer is theirs */
void Impl::history_phase_reset()
{
    chain_calls.resize(1);
    chain_pool_used = kHistoryWords;
    chain_tab.clear();
    chain_tab_uploaded = 0;
}

/* folds the calls of job `jobidx` into the buffer: word i <- the last call whose transform was longer than i */
bool Impl::history_commit(uint32_t jobidx, hipStream_t stream)
{
    /* from the last call backwards: a call owns the words between what later calls cover and its own extent (a transform's
     * length, or -- SVR on -- the n words of a refinement's residual) */
    uint32_t lo[SRLA_COMMIT_SEGS], hi[SRLA_COMMIT_SEGS], src[SRLA_COMMIT_SEGS], nseg = 0, covered = 0;
    /* the leading words that are known afterwards: the extents of known calls, from word 0 up to the first extent of a call that
     * read an unknown word; beyond all extents the buffer is what it was */
    uint32_t exact = 0;
    bool all_known = true;
    for (size_t j = chain_calls.size(); j-- > 1;) {
        const ChainCall &c = chain_calls[j];
        if (c.job != jobidx || c.nfft <= covered) continue;
        if (nseg == SRLA_COMMIT_SEGS) { fprintf(stderr, "[srla-mi355x] internal error: too many extents in a history phase\n"); return false; }
        lo[nseg] = covered; hi[nseg] = c.nfft; src[nseg] = c.dump; nseg++;
        if (all_known && !c.tainted) exact = c.nfft; else all_known = false;
        covered = c.nfft;
    }
    if (nseg == 0) return true;                                   /* a silent / RAW window: no call, the buffer stays */
    buf_exact = all_known ? std::max(covered, buf_exact) : exact;
    return srla_launch_chain_commit(stream, d_chain_pool.as<double>(), lo, hi, src, nseg) == 0;
}

SRLAApiResult Impl::history_window(uint32_t stream, uint32_t pos, uint32_t n, bool search)
{
    const auto tw0 = Clock::now();
    double t_prep1 = 0, t_dev1 = 0, t_read = 0, t_prep2 = 0, t_dev2 = 0;
    ChainRun &c = chain;
    StreamCtx &st = sx[stream];
    const uint32_t nch = par.num_channels;
    c.active = true; c.stream = stream; c.tail_start = pos; c.tail_n = n; c.search = search; c.seed_n = 0;
    /* which blocks are all zero decides which calls exist (srla_encoder.c:766-796): look at the samples */
    c.tail_smp.resize((size_t)nch * n);
    for (uint32_t ch = 0; ch < nch; ch++) {
        int32_t *dst = c.tail_smp.data() + (size_t)ch * n;
        if (st.pcm) (void)pcm_channel(st.pcm, st.pcm_bytes, nch, ch, pos, n, dst);
        else if (st.host_in) memcpy(dst, st.host_in[ch] + pos, (size_t)n * 4);
        else if (!d2h(dst, st.d_in + (size_t)ch * st.d_stride + pos, (size_t)n * 4)) return SRLA_APIRESULT_NG;
    }
    const std::function<bool(uint32_t, uint32_t)> silent = [&](uint32_t off, uint32_t len) { return chain_silent(c.tail_smp, c.tail_n, off, len); };
    Slot &sj = slot[kChainSlot + 1], &e = slot[kChainSlot + 2];
    chain_slot_defaults(sj); chain_slot_defaults(e);
    hipStream_t hs = sj.own_stream;
    /* the host's decisions about the window before this one are not about this one */
    overrides.erase(overrides.lower_bound(override_key(kChainJobKey, 0)), overrides.end());

    std::vector<uint32_t> lens;
    if (search) {
        /* SRLAEncoder_SearchOptimalBlockPartitions (srla_encoder.c:310-424): every candidate, then Dijkstra */
        history_phase_reset();
        if (!chain_make_job(sj, pos, n, true, nullptr)) return SRLA_APIRESULT_NG;
        const double t_made = ms_since(tw0);
        sj.job.key = 0;
        chain_append(1, sj.job, silent);
        if (chain_pool_used * sizeof(double) > d_chain_pool.cap || chain_tab.size() * 4 > d_chain_tab.cap) {
            fprintf(stderr, "[srla-mi355x] internal error: history pool too small (%llu words)\n", (unsigned long long)chain_pool_used);
            return SRLA_APIRESULT_NG;
        }
        chain_build(1, sj.job, c.cs);
        const double t_built = ms_since(tw0);
        for (int attempt = 0;; attempt++) {
            (void)apply_overrides(sj.job, kChainJobKey + 1);
            sj.job.uploaded = false;
            chain_tab_uploaded = 0;
            if (!prepare_job(sj, false)) return SRLA_APIRESULT_NG;
            const double t_prepared = ms_since(tw0);
            if (!chain_stage_a(sj, 1, c.cs)) return SRLA_APIRESULT_NG;
            const double t_a = ms_since(tw0);
            for (int st2 = ST_B; st2 <= ST_D; st2++) if (!run_stage(sj, st2)) return SRLA_APIRESULT_NG;
            {
                /* the partition comes back behind the pricing, on its stream: one wait for both */
                const SrlaWindowDesc &wd0 = sj.job.windows[0];
                const size_t bytes = (size_t)(wd0.num_nodes - 1) * sizeof(SrlaBlockRecord);
                if (!h_chain_recs.ensure(bytes)) return SRLA_APIRESULT_NG;
                if (hipMemcpyAsync(h_chain_recs.p, sj.d_blocks.as<SrlaBlockRecord>() + wd0.block_base, bytes, hipMemcpyDeviceToHost, hs) != hipSuccess) return SRLA_APIRESULT_NG;
            }
            t_prep1 = ms_since(tw0);
            if (chain_trace) fprintf(stderr, "[history]   search prep: samples + tables %.3f, call list %.3f, uploads %.3f, stage A enqueued %.3f, B-D enqueued %.3f ms\n",
                                     t_made, t_built, t_prepared, t_a, t_prep1);
            if (hipStreamSynchronize(hs) != hipSuccess) return SRLA_APIRESULT_NG;
            t_dev1 = ms_since(tw0);
            const int m = arbitrate(sj, kChainJobKey + 1);
            if (m < 0) return SRLA_APIRESULT_NG;
            if (m == 0) break;
            if (attempt >= 6) { fprintf(stderr, "[srla-mi355x] internal error: near-tie arbitration of a history-mode window did not settle\n"); return SRLA_APIRESULT_NG; }
            stats.num_restarts++;
        }
        sj.busy = false;
        if (!history_commit(1, hs)) return SRLA_APIRESULT_NG;
        const SrlaWindowDesc &wd = sj.job.windows[0];
        const SrlaBlockRecord *recs = h_chain_recs.as<SrlaBlockRecord>();
        uint32_t covered = 0;
        for (uint32_t k = 0; k + 1 < wd.num_nodes; k++) if (recs[k].valid) { lens.push_back(recs[k].n); covered += recs[k].n; }
        if (covered != n) { fprintf(stderr, "[srla-mi355x] internal error: a window's partitions cover %u of %u samples\n", covered, n); return SRLA_APIRESULT_NG; }
        t_read = ms_since(tw0);
    } else lens.push_back(n);

    /* SRLAEncoder_EncodeBlock for every block of the partition (srla_encoder.c:1676-1692): analysed again, where they now stand */
    history_phase_reset();
    if (!chain_make_job(e, pos, n, false, &lens)) return SRLA_APIRESULT_NG;
    chain_append(2, e.job, silent);
    if (chain_pool_used * sizeof(double) > d_chain_pool.cap || chain_tab.size() * 4 > d_chain_tab.cap) {
        fprintf(stderr, "[srla-mi355x] internal error: history pool too small (%llu words)\n", (unsigned long long)chain_pool_used);
        return SRLA_APIRESULT_NG;
    }
    chain_build(2, e.job, c.ce);
    e.emits = true; e.merge_cb = true; e.out_boost = 1;
    for (int attempt = 0;; attempt++) {
        (void)apply_overrides(e.job, kChainJobKey + 2);
        e.job.uploaded = false;
        chain_tab_uploaded = 0;
        st.pass_started = false;                                 /* the window goes where the host knows the stream has reached */
        if (!prepare_job(e, false) || !chain_stage_a(e, 2, c.ce)) return SRLA_APIRESULT_NG;
        for (int st2 = ST_B; st2 <= ST_E; st2++) if (!run_stage(e, st2)) return SRLA_APIRESULT_NG;
        t_prep2 = ms_since(tw0);
        if (!wait_job(e)) return SRLA_APIRESULT_NG;
        t_dev2 = ms_since(tw0);
        if (e.h_info.as<SrlaJobInfo>()->num_tie_items == 0) break;
        const int m = arbitrate(e, kChainJobKey + 2);
        if (m < 0) return SRLA_APIRESULT_NG;
        if (m == 0) break;
        if (attempt >= 6) { fprintf(stderr, "[srla-mi355x] internal error: near-tie arbitration of a history-mode window did not settle\n"); return SRLA_APIRESULT_NG; }
        stats.num_restarts++;
    }
    if (!history_commit(2, hs)) return SRLA_APIRESULT_NG;
    stats.num_history_windows++;
    if (chain_trace)
        fprintf(stderr, "[history] window of %u: search enqueued %.3f, priced %.3f, partition read %.3f, encode enqueued %.3f, collected %.3f ms (%u + %u rounds)\n",
                n, t_prep1, t_dev1, t_read, t_prep2, t_dev2, search ? c.cs.rounds : 0u, c.ce.rounds);
    if (want_block_price) {
        /* SRLAEncoder_ComputeBlockSize: what the SEARCH pays for the block (with more than two channels the reference prices the
         * first two only, srla_encoder.c:1287-1301), which the pricing kernel left in the block's record */
        SrlaBlockRecord rec;
        if (!d2h(&rec, e.d_blocks.as<SrlaBlockRecord>() + e.job.windows[0].block_base, sizeof(rec)) || !rec.valid)
            return SRLA_APIRESULT_NG;
        block_price = rec.price;
    }
    return finish_job(e);
}

SRLAApiResult Impl::history_encode(bool search)
{
    const auto t0 = Clock::now();
    const uint32_t minb = par.min_num_samples_per_block, maxb = par.max_num_samples_per_block;
    const uint32_t window_len = search ? par.num_lookahead_samples : maxb;
    const uint32_t nv = num_variants(), passes = par.ltp_order > 0 ? 2u : 1u;
    auto drain = [&]() { for (auto &q : streams) if (q) (void)hipStreamSynchronize(q); if (upload) (void)hipStreamSynchronize(upload); };
    {
        /* the pool: the buffer + room for the complete FFT buffer of every call of the larger phase (the search phase: its
         * candidates include every block a partition can hold); sized once -- growing it would lose the buffer */
        uint64_t words = 0, calls = 0;
        auto pow2 = [](uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; };
        if (!search) { words = pow2(maxb); calls = 1; }
        else {
            const uint32_t nodes = window_len / minb + 1;
            for (uint32_t i = 0; i < nodes; i++)
                for (uint32_t j = i + 1; j < nodes; j++) {
                    uint32_t len = (j - i) * minb;
                    if (len > maxb) break;
                    words += pow2(std::min(len, window_len - i * minb)); calls++;
                }
        }
        const size_t want = (kHistoryWords + (uint64_t)nv * (passes + (chain_svr() ? 1u : 0u)) * (words + 2u * calls) + 1024u) * sizeof(double);
        drain();
        if (want > d_chain_pool.cap && (!hist_leave_pool() || !d_chain_pool.ensure(want))) return SRLA_APIRESULT_NG;
        if (!d_chain_tab.ensure(((size_t)calls * nv * SRLA_LTP_LAGS + 1024u) * 4)) return SRLA_APIRESULT_NG;
    }
    SRLAApiResult worst = SRLA_APIRESULT_OK;
    chain.active = false;
    for (uint32_t i = 0; i < sx.size(); i++) {
        StreamCtx &st = sx[i];
        /* the buffer as this handle's calls left it (one of the reference's entry points: host_impl.h, d_hist); a fresh handle's --
         * and, for the streams of a batch, each stream's -- starts as zero pages (the `srla` tool creates its encoder per file) */
        const bool tracked = sx.size() == 1 && st.reference_call;
        if (tracked && !hist_fresh && pool_holds_hist) {
            /* the pool's head is the buffer as the handle's last call left it */
        } else if (tracked && !hist_fresh && d_hist.p != nullptr) {
            if (hipMemcpyAsync(d_chain_pool.p, d_hist.p, (size_t)kHistoryWords * sizeof(double), hipMemcpyDeviceToDevice, streams[1]) != hipSuccess) return SRLA_APIRESULT_NG;
        } else {
            if (!hist_leave_pool()) return SRLA_APIRESULT_NG;
            if (hipMemsetAsync(d_chain_pool.p, 0, (size_t)kHistoryWords * sizeof(double), streams[1]) != hipSuccess) return SRLA_APIRESULT_NG;
        }
        pool_holds_hist = false;
        buf_exact = (tracked && !hist_fresh) ? hist_exact : kHistoryWords;
        if (tracked) hist_exact = 0;                              /* (until the call has come through: the ways out below) */
        chain_calls.clear();
        ChainCall buffer{};
        buffer.job = 0xFFFFFFFFu; buffer.nfft = kHistoryWords; buffer.src = -1; buffer.dump = 0;
        chain_calls.push_back(buffer);
        for (uint32_t pos = 0; pos < st.num_samples && st.rc == SRLA_APIRESULT_OK; pos += window_len) {
            const SRLAApiResult rc = history_window(i, pos, std::min(window_len, st.num_samples - pos), search);
            if (rc == SRLA_APIRESULT_INSUFFICIENT_BUFFER) st.rc = rc;
            else if (rc != SRLA_APIRESULT_OK) { drain(); for (auto &sl : slot) sl.busy = false; chain.active = false; return rc; }
        }
        chain.active = false;
        if (tracked) {
            /* what the next call on this handle finds: the pool's head, where it stands */
            pool_holds_hist = true;
            hist_fresh = false;
            hist_exact = (st.rc == SRLA_APIRESULT_OK) ? buf_exact : 0u;       /* (a call that ran out of room stopped somewhere inside a window) */
        }
        if (st.rc != SRLA_APIRESULT_OK) { worst = st.rc; continue; }
        if (!write_header(st)) return SRLA_APIRESULT_NG;
    }
    drain();
    for (auto &sl : slot) sl.busy = false;
    if (sx.size() == 1 && sx[0].with_header && sx[0].rc == SRLA_APIRESULT_OK) offset_lshift = sx[0].lshift;
    stats.history_ms += ms_since(t0);
    return worst;
}
