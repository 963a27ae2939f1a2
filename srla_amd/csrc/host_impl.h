/*
 * host_impl.h -- the encoder handle behind the SRLAEncoder_* C ABI (include/srla_mi355x.h): its state and the
 * functions of the host runtime, which are defined in
 *   host_plan.cpp      tables of the block-division search for a range of windows (jobs), host-libm constants
 *   host_pipeline.cpp  device set-up, staging of host input, the staged execution of jobs, the stream loop
 *   host_chain.cpp     chain mode: the history-dependent last window of a stream
 *   host_ties.cpp      host-libm arbitration of decisions the device flagged as near-ties
 *   host_api.cpp       the C ABI itself
 *
 * What runs where:
 *   host    argument checking exactly as the reference API, stream header, splitting the stream
 *           into look-ahead windows, the candidate/item tables of the block-division search,
 *           host-libm constant tables, staging of host input, enqueueing, collecting finished jobs.
 *   device  everything between samples and finished stream bytes: kernels.hip.
 *
 * A stream is processed as a sequence of jobs (ranges of whole windows, ~4 M samples).  The stages of consecutive
 * jobs are enqueued skewed on three streams (software pipeline, see run_stage / encode_stream) with up to four jobs
 * in flight; the host thread only enqueues and waits for one event per job.  Windows carry no state
 * from one to the next, so jobs are independent; only the offset left shift is a
 * whole-stream quantity (device-resident for device input, speculated for host input).
 *
 * There is no CPU fallback: if no HIP device can be initialised every Encode* / ComputeBlockSize
 * call fails with SRLA_APIRESULT_NG and a message on stderr.
 */
#ifndef SRLA_HOST_IMPL_H
#define SRLA_HOST_IMPL_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <functional>
#include <map>
#include <stdarg.h>
#include <string>
#include <vector>

#include "../../include/srla_mi355x.h"
#include "device_layout.h"
#include "host_pack.h"
#include "host_support.h"
#include "host_tables.h"
#include "kernels.h"

#define SRLA_HANDLE_MAGIC 0x53524C41u /* 'SRLA' */
#define SRLA_MAX_FFT      8192u       /* largest block the LDS-resident FFT handles */

struct SRLAEncoder {
    uint32_t magic;
    uint8_t alloced_by_own;
    void *work;
    struct Impl *impl;
};

namespace srla {

/* max LPC order per preset, libs/srla_internal/src/srla_internal.c:30-38 */
extern const uint32_t kPresetOrder[SRLA_NUM_PARAMETER_PRESETS];
extern int g_device_index;            /* SRLAMI355X_SetDevice: the device of handles created afterwards */

/* ---- one job: a range of whole windows -------------------------------------------------- */
struct Group {
    uint32_t nfft, first, count;
    int rclass;
    SrlaLdsPlan plan;
};

struct Job {
    uint32_t s0 = 0, ns = 0;          /* sample range inside the stream */
    std::vector<SrlaWindowDesc> windows;
    std::vector<SrlaCandDesc> cands;
    std::vector<SrlaItemDesc> items;
    std::vector<Group> groups;
    uint32_t num_slots = 0;
    uint64_t res_elems = 0;
    uint64_t analyzed_samples = 0;
    std::vector<SrlaAutocorrItem> class_index; /* the items grouped by FFT-size class (srla_autocorr launches per class) */
    uint32_t class_first[4] = {}, class_count[4] = {};   /* N' <= 1024, 2048, 4096, 8192 */
    uint64_t key = 0;                 /* geometry signature: equal keys => identical descriptor tables */
    bool uploaded = false;            /* the slot's device copies match the tables above */
};

struct Slot {
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;    /* chain-mode jobs: every stage but the block assembly runs here */
    hipEvent_t t0[6] = {}, t1[6] = {};   /* start / end of the stages of the job (see Impl::run_stage) */
    hipEvent_t ev_in = nullptr;          /* the job's samples have arrived in d_input (host-input calls) */
    const int32_t *in_cur = nullptr;     /* device input of the current job */
    uint32_t stride_cur = 0;
    SrlaJobParams jp{};
    bool want_dbg = false;
    bool timed = false;                  /* this job records start events for every stage (one job in four) */
    /* where this job's blocks go (set when the job is begun, used by the pack stage) */
    uint8_t *out_direct = nullptr;       /* device-visible caller buffer, or nullptr: stage through h_stream */
    uint32_t out_first = 1, out_init_pos = 0, out_limit = 0xFFFFFFFFu;
    uint32_t out_boost = 1;              /* stream-out workgroup multiplier (the last jobs of a stream drain faster) */
    DevBuf d_input16;                    /* host input of at most 16 bits crosses PCIe as int16 and is widened into d_input */
    DevBuf d_input, d_items, d_cands, d_windows, d_results, d_res_ws, d_blocks, d_block_off, d_ctl, d_scratch, d_dbg, d_lags, d_err, d_class_index, d_stream;
    PinBuf h_in, h_stream, h_info;       /* h_info: SrlaJobInfo followed by the per-window byte counts */
    Job job;
    bool busy = false;
    bool used_h2d = false;
};

}  // namespace srla

using namespace srla;

struct Impl {
    SRLAEncoderConfig cfg{};
    SRLAEncodeParameter par{};
    bool set_parameter = false;
    uint32_t param_generation = 0;    /* bumped by SetEncodeParameter: invalidates cached job tables */
    uint32_t offset_lshift = 0;       /* encoder->header.offset_lshift of the reference */
    uint32_t pack_threads = 0;

    int device = 0;                   /* HIP device of this handle (SRLAMI355X_SetDevice at the time of Create) */
    bool dev_ready = false, dev_failed = false;
    static constexpr uint32_t kMaxSlots = 11;         /* rotating + 2 tail + 3 chain-mode job buffer sets */
    static constexpr uint32_t kStreams = 3;   /* more streams than HW queues serialise badly (measured) */
    hipStream_t streams[kStreams] = {};
    hipStream_t chain_stream = nullptr; /* autocorrelation rounds of chain mode */
    hipStream_t upload = nullptr;      /* H2D of host-input jobs: a DMA queue of its own, so uploads never wait behind kernels */
    hipEvent_t ev_or = nullptr;       /* offset-shift reduction done */
    hipEvent_t ev_ref = nullptr;      /* SRLA_MI355X_TIMELINE: start of the stream on the wide stream */
    bool timeline = false;
    std::string tl_log;               /* printed when the stream is done: writing to stderr on the way distorts what is measured */
    void tl_printf(const char *fmt, ...) __attribute__((format(printf, 2, 3)));
    bool lshift_on_device = false;
    PinBuf h_or;
    uint32_t kSlots = 4;              /* job buffer sets (SRLA_MI355X_SLOTS); slot i runs on stream i % kStreams */
    uint64_t job_samples = 4ull << 20; /* samples per job (SRLA_MI355X_JOB_SAMPLES): fixed per-job latencies (serial solve chain, launch gaps) favour large jobs; measured best for long streams, and never worse than smaller ones for short streams */
    Slot slot[kMaxSlots];
    DevBuf d_tw, d_geoms, d_thr, d_huff, d_huffcode, d_pos, d_or;
    bool timing = true;               /* stage timing events (SRLA_MI355X_NO_TIMING drops them) */
    uint32_t tail_boost = 4, tail_boost_jobs = 3;   /* SRLA_MI355X_TAIL_BOOST="wgs,jobs" */
    uint32_t timing_stride = 4;       /* every n-th job carries start events on all stages (SRLA_MI355X_TIMING_STRIDE) */
    bool in_pinned = false;           /* this call's input planes are pinned host memory */
    /* Host input without a callback: the stream is encoded assuming offset shift 0 while the staging copies gather the
     * OR of all samples; only if that OR has trailing zeros (rare for audio) the stream is encoded again with the
     * right shift.  Saves a separate pass over the input before the first kernel can start. */
    bool spec_or_active = false, spec_guessed = false;
    std::atomic<uint32_t> spec_or{ 0 };
    int forced_lshift = -1;           /* >= 0: the shift is known (second attempt) */
    bool no_speculation = false;      /* SRLA_MI355X_NO_SPECULATION */
    bool force_staging = false;       /* SRLA_MI355X_STAGING: never write the caller's buffer from the device */
    bool no_pack16 = false;           /* SRLA_MI355X_NO_PACK16: host input always crosses PCIe as int32 */
    std::map<uint32_t, uint32_t> tw_index;   /* nfft -> offset (double2) */
    std::vector<double> tw_host;
    bool tw_dirty = false;
    std::map<uint32_t, uint32_t> geom_index; /* n -> index */
    std::vector<SrlaGeom> geoms;
    bool geom_dirty = false;
    Pool *pool = nullptr;
    SRLAMI355XStats stats{};

    ~Impl();

    uint32_t preset_order() const { return srla::kPresetOrder[par.preset]; }
    uint32_t num_variants() const { return par.num_channels + (par.num_channels >= 2 ? 2u : 0u); }
    bool search_enabled() const { return par.min_num_samples_per_block != par.max_num_samples_per_block; }

    /* ---- host_pipeline.cpp ---- */
    bool init_device();
    /* ---- host_plan.cpp ---- */
    uint32_t geom_for(uint32_t n);
    bool sync_tables();               /* tables are shared by all slots: waits for everything in flight before re-allocating them */
    SrlaLdsPlan lds_plan(uint32_t nfft) const;   /* LDS carve-up of srla_residual_cost for the largest FFT size of a job */
    /* Candidate table of SearchOptimalBlockPartitions (srla_encoder.c:336-389) for the windows
     * [first sample s0, s0+ns) of a stream; ns ends on a window boundary or at the stream end.
     * `lens` (chain mode): the job's windows are these blocks, one candidate each, instead of the regular tiling */
    void build_job(Job &job, uint32_t s0, uint32_t ns, bool search, const std::vector<uint32_t> *lens = nullptr);
    SrlaJobParams job_params(const Job &job, uint32_t channel_stride) const;
    uint32_t windows_per_job(bool search) const;   /* bounded by scratch memory (~1.5 GB of residual scratch per slot) */

    /* ---- staged execution of one job (host_pipeline.cpp) --------------------------------------------------
     * Three streams: W carries the wide kernels (autocorr, residual_cost), N the narrow ones
     * (Levinson / order / quantiser, pricing), C the block assembly and the stream-out.  A job's stages are chained
     * with events; encode_stream enqueues the stages of consecutive jobs skewed (software pipeline), so that W always
     * has a wide kernel to run while N works through the serial stages of the neighbouring job. */
    enum { ST_A = 0, ST_B, ST_C, ST_D, ST_E, NUM_ST };
    /* d_in: device pointer to channel 0 of the job's first sample, or nullptr to upload host_in (planar pointers,
     * absolute stream positions) */
    bool prepare_job(Slot &s, const int32_t *d_in, uint32_t d_stride, const int32_t *const *host_in, bool want_dbg);
    bool run_stage(Slot &s, int st);
    bool launch_job(Slot &s, const int32_t *d_in, uint32_t d_stride, const int32_t *const *host_in, bool want_dbg);   /* all stages back to back */
    bool wait_job(Slot &s);
    /* A finished job: check the device's verdict, move the bytes to `data + write_off` unless the device wrote
     * them there itself, and report the per-window sizes. */
    SRLAApiResult finish_job(Slot &s, uint8_t *data, uint32_t write_off, uint32_t *written, const uint32_t **window_bytes);
    srla::StreamInfo stream_info(uint32_t num_samples) const;
    /* The body shared by EncodeWhole (host input) and EncodeWholeDevice. */
    SRLAApiResult encode_stream(const int32_t *const *host_in, const int32_t *d_in, uint32_t d_stride,
                                uint32_t num_samples, uint8_t *data, uint32_t data_size, uint32_t *output_size,
                                SRLAEncoder_EncodeBlockCallback cb, bool with_header, bool search);

    /* ---- chain mode: the odd-length tail window ------------------------------------------------------------
     * The reference's Welch window never writes the middle word of an odd-length block (lpc.c:260-264), so that
     * word of its persistent FFT buffer (lpc.c:58,211) still holds what the previous autocorrelation call left
     * there: the analysis of an odd block depends on the calls before it.  With an even minimum block size only
     * the blocks that end at the end of the stream can be odd, so only the last window of an odd-length stream is
     * affected.  That window is encoded in "chain mode": the host lists the reference's autocorrelation calls in
     * its order (search: every candidate block, for each M, S, then the channels, LTP lags before LPC lags,
     * srla_encoder.c:310-424,1208-1334; then the chosen partitions once more, :1646-1698), gives every call a place
     * in a device pool where it leaves its complete FFT buffer, and points every odd call at the word it inherits:
     * index n/2 of the latest earlier call whose FFT was longer than n/2.  Calls are launched in rounds so that a
     * call runs after its source; even calls need no source and all run in round 0.  The calls before the window
     * matter only through the last block encoded before it (the "seed" job).  A fresh handle starts from zeros,
     * as the `srla` tool's freshly mapped buffer does. */
    struct ChainCall { uint32_t job, item, pass, n, nfft, round; int32_t src; uint32_t dump, lags; };
    struct ChainLaunch { uint32_t round, pass, cls, first, count; };
    struct ChainJob {
        std::vector<SrlaAutocorrItem> list;
        std::vector<ChainLaunch> launches;
        std::vector<uint32_t> select;     /* per item: the round of its LTP-lag call */
        uint32_t rounds = 0;
    };
    std::vector<ChainCall> chain_calls;
    uint64_t chain_pool_used = 0;
    std::vector<uint32_t> chain_tab;      /* gather table for the LTP lags beyond a short FFT (SrlaAutocorrItem::chain_lags) */
    size_t chain_tab_uploaded = 0;
    DevBuf d_chain_pool, d_chain_tab, d_chain_list[3], d_chain_select[3];
    /* the reference's calls for the candidates of `job`, appended in its order; silent(off, n): the block is all zero */
    void chain_append(uint32_t jobidx, const Job &job, const std::function<bool(uint32_t, uint32_t)> &silent);
    void chain_build(uint32_t jobidx, const Job &job, ChainJob &cj);
    bool chain_stage_a(Slot &s, uint32_t jobidx, const ChainJob &cj);   /* stage A of a chain job: the autocorrelation launches round by round */
    /* The last window [tail_start, tail_start + tail_n) of a stream in chain mode, in three steps so that it overlaps
     * the regular jobs: chain_begin (seed + search job; needs nothing from the jobs before unless the seed does),
     * chain_encode_ad (reads the search result, enqueues the encode job up to its pricing), chain_encode_e (block
     * assembly, after the last regular job's).  seed_n > 0: the block [seed_off, seed_off + seed_n) is the last one
     * encoded before the window. */
    struct ChainRun {
        bool active = false, begun = false, early = false, ad_done = false;
        uint32_t tail_start = 0, tail_n = 0;
        bool search = false;
        const int32_t *const *host_in = nullptr;
        const int32_t *d_in = nullptr;
        uint32_t d_stride = 0;
        std::vector<int32_t> tail_smp, seed_smp;
        uint32_t seed_n = 0;
        ChainJob cq, cs, ce;
    } chain;
    static constexpr uint32_t kChainSlot = kMaxSlots - 3;   /* seed, search, encode */
    bool chain_silent(const std::vector<int32_t> &v, uint32_t total, uint32_t off, uint32_t n) const;
    void chain_slot_defaults(Slot &s);
    bool chain_begin(uint32_t seed_off, uint32_t seed_n);
    /* the encode job up to its pricing; `first_job`: nothing was encoded before the window */
    bool chain_encode_ad(uint8_t *out_direct, uint32_t init_pos, uint32_t data_size, bool first_job);
    bool chain_encode_e();
    bool chain_search_done();         /* has the search job been priced (so that the encode job can be enqueued without waiting)? */
};

#endif /* SRLA_HOST_IMPL_H */
