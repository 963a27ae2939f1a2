/*
 * host_plan.cpp -- what the host prepares for a job: host-libm constants per block length, the window / candidate /
 * item tables of the block-division search (srla_encoder.c:336-389) and the job parameters.
 */
#include "host_impl.h"

#include <algorithm>
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
    p.fft_off = off; if (par.ltp_order > 0) off += al(sig_bytes);
    p.lev_off = 0;
    p.means_off = off; off += al(means_bytes);
    p.small_off = off; off += srla_kernel_small_c_bytes();
    p.total = off;
    /* the fast path of the same kernel carves the block differently: make sure it fits too */
    for (uint32_t fl = 1; fl <= 8 && 1024u * fl <= nfft; fl++) p.total = std::max(p.total, srla_kernel_fast_lds_bytes(fl));
    return p;
}

void Impl::build_job(Job &job, uint32_t s0, uint32_t ns, bool search, const std::vector<uint32_t> *lens)
{
    const uint32_t minb = par.min_num_samples_per_block, maxb = par.max_num_samples_per_block;
    const uint32_t window_len = search ? par.num_lookahead_samples : maxb;
    const uint32_t nv = num_variants(), pmax = preset_order();
    /* all tables are relative to the job's first sample, so jobs of equal length share them */
    const uint64_t key = lens ? 1ull : (((uint64_t)ns << 24) ^ ((uint64_t)param_generation << 1) ^ (search ? 1u : 0u) ^ 0x8000000000000000ull);
    job.s0 = s0;
    if (!lens && job.key == key && job.ns == ns) return;
    if (lens) search = false;
    job.key = key; job.uploaded = false;
    job.ns = ns;
    job.windows.clear(); job.cands.clear(); job.items.clear(); job.groups.clear(); job.class_index.clear();
    job.num_slots = 0; job.res_elems = 0; job.analyzed_samples = 0;

    struct Pending { uint32_t cand; uint32_t nfft; };
    std::vector<Pending> analysed;
    for (uint32_t pos = 0, wi = 0; pos < ns; wi++) {
        const uint32_t wn = lens ? (*lens)[wi] : std::min(window_len, ns - pos);
        SrlaWindowDesc wd{};
        wd.sample_off = pos; wd.n = wn;
        wd.cand_base = (uint32_t)job.cands.size();
        wd.num_nodes = search ? ((wn + minb - 1) / minb + 1) : 2;
        wd.block_base = job.num_slots;
        job.num_slots += wd.num_nodes - 1;
        const uint32_t w = (uint32_t)job.windows.size();
        auto add_cand = [&](uint32_t i, uint32_t j, uint32_t off, uint32_t n) {
            SrlaCandDesc cd{};
            cd.window = w; cd.node_i = i; cd.node_j = j; cd.sample_off = pos + off; cd.n = n;
            cd.item_base = 0xFFFFFFFFu;
            if (n > pmax) analysed.push_back({ (uint32_t)job.cands.size(), geoms[geom_for(n)].nfft });
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
        job.windows.push_back(wd);
        pos += wn;
    }
    /* one launch analyses every item of the job: the LDS plan and the FFT register class are
     * those of the largest FFT present (smaller items simply leave part of them idle) */
    uint32_t max_nfft = 0;
    for (const Pending &p : analysed) max_nfft = std::max(max_nfft, p.nfft);
    if (!analysed.empty()) {
        Group g{};
        g.nfft = max_nfft;
        g.first = 0;
        g.rclass = (int)std::max(1u, g.nfft / 2048u);
        g.plan = lds_plan(g.nfft);
        for (const Pending &p : analysed) {
            SrlaCandDesc &cd = job.cands[p.cand];
            cd.item_base = (uint32_t)job.items.size();
            for (uint32_t v = 0; v < nv; v++) {
                SrlaItemDesc it{};
                it.sample_off = cd.sample_off; it.n = cd.n; it.variant = v;
                it.geom = geom_for(cd.n);
                it.res_off = job.res_elems;
                it.forced_order = -1;
                job.res_elems += (cd.n + 3u) & ~3u;
                job.analyzed_samples += cd.n;
                job.items.push_back(it);
            }
        }
        g.count = (uint32_t)job.items.size();
        job.groups.push_back(g);
    }
    job.class_index.clear();
    for (int c = 0; c < 4; c++) {
        job.class_first[c] = (uint32_t)job.class_index.size();
        for (uint32_t i = 0; i < job.items.size(); i++) {
            const uint32_t nfft = geoms[job.items[i].geom].nfft;
            const int cls = (nfft <= 1024u) ? 0 : ((nfft <= 2048u) ? 1 : ((nfft <= 4096u) ? 2 : 3));
            if (cls == c) {
                const SrlaItemDesc &it = job.items[i];
                const SrlaGeom &gm = geoms[it.geom];
                SrlaAutocorrItem ai{};
                ai.item = i; ai.sample_off = it.sample_off; ai.n = it.n; ai.variant = it.variant;
                ai.nfft = gm.nfft; ai.tw_off = gm.tw_off; ai.welch_divisor = gm.welch_divisor; ai.acorr_norm = gm.acorr_norm;
                job.class_index.push_back(ai);
            }
        }
        job.class_count[c] = (uint32_t)job.class_index.size() - job.class_first[c];
    }
}

SrlaJobParams Impl::job_params(const Job &job, uint32_t channel_stride) const
{
    SrlaJobParams jp{};
    jp.num_channels = par.num_channels;
    jp.bits_per_sample = par.bits_per_sample;
    jp.offset_lshift = offset_lshift;
    jp.max_order = preset_order();
    jp.order_fixed = (par.preset == 0) ? 1u : 0u;
    jp.ltp_order = par.ltp_order;
    jp.num_samples = job.ns;
    jp.channel_stride = channel_stride;
    jp.max_block = par.max_num_samples_per_block;
    jp.min_block = par.min_num_samples_per_block;
    jp.num_items = (uint32_t)job.items.size();
    jp.num_cands = (uint32_t)job.cands.size();
    jp.num_windows = (uint32_t)job.windows.size();
#ifdef SRLA_DIAG_STOP
    { static const char *e = getenv("SRLA_MI355X_K3_STOP"); jp.out_stride = e ? (uint32_t)atoi(e) : 0u; }   /* kernel timing experiments only */
#endif
    jp.lshift_dev = lshift_on_device ? (d_or.as<uint32_t>() + 1) : nullptr;
    return jp;
}

uint32_t Impl::windows_per_job(bool search) const
{
    const uint32_t window_len = search ? par.num_lookahead_samples : par.max_num_samples_per_block;
    uint64_t per_window = (uint64_t)window_len * num_variants() * 4;
    if (search) {
        const uint32_t ratio = par.max_num_samples_per_block / par.min_num_samples_per_block;
        per_window *= ratio;
    }
    uint64_t w = (1536ull << 20) / std::max<uint64_t>(per_window, 1);
    const uint64_t cap_samples = job_samples;  /* several jobs per stream so that the GPU and the host pack overlap */
    w = std::min<uint64_t>(w, std::max<uint64_t>(1, cap_samples / window_len));
    return (uint32_t)std::max<uint64_t>(1, w);
}
