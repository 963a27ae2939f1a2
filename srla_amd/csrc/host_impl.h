/*
 * host_impl.h -- the encoder handle behind the SRLAEncoder_* C ABI (include/srla_mi355x.h): its state and the
 * functions of the host runtime, which are defined in
 *   host_plan.cpp      tables of the block-division search for a range of windows (jobs), host-libm constants
 *   host_pipeline.cpp  device set-up, staging of host input, the staged execution of jobs, the loop over the jobs of a call
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
 * A call encodes one stream (SRLAEncoder_EncodeWhole) or many (SRLAMI355X_EncodeBatch).  The streams are cut into jobs:
 * ranges of whole look-ahead windows, ~4 M samples, of one stream or of several short ones (a job's run of windows of
 * one stream is a "segment").  The stages of consecutive jobs are enqueued skewed on three HIP streams (software
 * pipeline, see run_stage / encode_streams) with up to four jobs in flight; the host thread only enqueues and waits for
 * one event per job.  Windows carry no state from one to the next, so jobs are independent; only the
 * offset left shift is a whole-stream quantity (device-resident for device input, speculated for host input).
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
#define SRLA_MAX_FFT      65536u      /* largest transform: blocks of up to 65 535 samples, what a block header can say (srla_encoder.c:1583-1595);
                                       * above 8192 samples: the global-memory slow paths of kernels.hip */
#define SRLA_BIG_GROUPS   384u        /* persistent workgroups (scratch regions) of srla_autocorr_big */

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
    /* rclass 4 (blocks above 4096 samples in the job): srla_residual_cost in two launches -- the items of at most 4096 samples with
     * the 92-register form (plan_small), the others with the 8192-sample form (plan) */
    bool split = false;
    SrlaLdsPlan plan_small;
};

/* samples [s0, s0 + ns) of stream `stream` of the call, standing at sample `base` of the job's input planes */
struct SegPlan {
    uint32_t stream, s0, ns, base;
};
struct JobPlan {
    std::vector<SegPlan> segs;
    uint32_t total = 0;               /* samples per channel plane of the job's input (segments + alignment gaps) */
    uint32_t slot = 0;
};

struct Job {
    std::vector<SegPlan> segs;        /* what the tables below were built for */
    std::vector<uint32_t> seg_lshift;
    uint32_t total = 0;
    std::vector<SrlaWindowDesc> windows;
    std::vector<SrlaCandDesc> cands;
    std::vector<SrlaItemDesc> items;
    std::vector<Group> groups;
    std::vector<uint32_t> seg_first_window;   /* per segment (+ one past the end) */
    uint32_t num_slots = 0;
    uint32_t max_nodes = 2, max_window_cands = 1;   /* of a window of the job (srla_price_windows: LDS size, global fallback) */
    uint64_t res_elems = 0;
    bool keep_residuals = false;
    uint64_t analyzed_samples = 0;
    std::vector<SrlaAutocorrItem> class_index; /* the items grouped by FFT-size class (srla_autocorr launches per class) */
    uint32_t class_first[8] = {}, class_count[8] = {};   /* N' < 1024, 2048, 4096, 8192, 16384, 32768; N' = 1024; N' = 65536 */
    std::vector<uint32_t> big_items;  /* items of more than 8192 samples (srla_residual_cost_big) */
    std::vector<double> svr_rows;     /* rows of 256: SVR-refined predictors the host arbitrated (SrlaItemDesc::forced_svr) */
    uint32_t big_max_n = 0;
    uint64_t key = 0;                 /* geometry signature: equal keys => identical descriptor tables */
    bool uploaded = false;            /* the slot's device copies match the tables above */
};

struct Slot {
    hipStream_t own_stream = nullptr;    /* chain-mode jobs: every stage but the block assembly runs here */
    hipEvent_t t0[6] = {}, t1[6] = {};   /* start / end of the stages of the job (see Impl::run_stage) */
    hipEvent_t ev_a1 = nullptr, ev_p0 = nullptr, ev_p = nullptr, ev_a0 = nullptr;
    hipEvent_t ev_pk = nullptr;          /* block assembly done (stream N): hand-over to srla_stream_out on stream C */   /* stage A of an LTP job in two parts (run_stage): end of the
                                          * LTP-pass autocorrelation, start / end of the pitch solve, start of the LPC-pass autocorrelation */
    hipEvent_t ev_in = nullptr;          /* the job's samples have arrived in d_input (host-input calls) */
    const int32_t *in_cur = nullptr;     /* device input of the current job */
    uint32_t stride_cur = 0;
    SrlaJobParams jp{};
    bool want_dbg = false;
    bool piece = false;                  /* one of the two or three jobs of a short call: stages A - D on a stream of its own, no end events between them */
    bool solo = false;                   /* the only job of a call without a chain-mode window: every stage, the block assembly too, on own_stream (run_stage) */
    bool split_a = false;                /* stage A was enqueued in two parts (run_stage) */
    bool c_start = false;                /* srla_residual_cost's launch carries a start event (timed jobs; every job of a call of two or three) */
    bool timed = false;                  /* this job records start events for every stage (one job in four) */
    bool last_job = false;               /* one of the last jobs of the call's plan (Impl::kDmaTailJobs) */
    bool use_dma = false;                /* this job's bytes leave by a host-issued copy when it is collected (Impl::dma_out) */
    bool dma_pending = false;            /* ev_dma marks the end of the last copies out of this slot's staging buffer */
    hipEvent_t ev_dma = nullptr;
    uint32_t out_boost = 1;              /* stream-out workgroup multiplier (the last jobs of a stream drain faster) */
    DevBuf d_pcm;                        /* PCM input: the job's frames as uploaded, de-interleaved into d_input by srla_deinterleave */
    DevBuf d_input16;                    /* host input of at most 16 bits crosses PCIe as int16 and is widened into d_input */
    DevBuf d_price_ws;                   /* srla_price_windows: two words per candidate, for windows whose candidates do not fit LDS */
    DevBuf d_input, d_items, d_cands, d_windows, d_results, d_res_ws, d_blocks, d_block_off, d_scratch, d_dbg, d_lags, d_err, d_gamma, d_class_index, d_stream;
    DevBuf d_segs, d_seg_ctl;            /* SrlaSegDesc per segment; device-side segment records of srla_block_offsets */
    DevBuf d_coef_ws;                    /* SVR refinement: 64 doubles per item, the predictor between solve and quantiser */
    DevBuf d_big_sig;                    /* srla_residual_cost_big: the signal of blocks above 32768 samples (it no longer fits LDS) */
    DevBuf d_big_scratch, d_big_items;   /* blocks above 8192 samples: FFT scratch in global memory, indices of the big items */
    DevBuf d_svr_rows;                   /* Job::svr_rows on the device */
    DevBuf d_ties, d_tie_data;           /* near-tie list of the job (count + entries), 8 doubles per entry for LTP entries */
    PinBuf h_in, h_stream, h_info;       /* h_info: SrlaJobInfo, the per-window byte counts, SrlaSegInfo per segment */
    PinBuf h_segs;                       /* host copy of the segment table (uploaded per job) */
    PinBuf h_ties;                       /* SrlaTieGather::out */
    bool ties_gathered = false;          /* the job went through the block assembly: its near-ties' numbers stand in h_ties */
    Job job;
    bool busy = false;
    bool used_h2d = false;
    bool b_done = false;                 /* chain mode with SVR on: the solve chain already ran, round by round, inside stage A */
    bool emits = true;                   /* the job runs the block assembly (chain mode's seed and search jobs do not) */
    bool merge_cb = false;               /* one encode callback for the whole segment (chain mode's encode job: its windows are the partitions of ONE look-ahead window) */
    const uint32_t *window_bytes() const { return reinterpret_cast<const uint32_t *>(h_info.as<SrlaJobInfo>() + 1); }
    const SrlaSegInfo *seg_info() const { return reinterpret_cast<const SrlaSegInfo *>(window_bytes() + job.windows.size()); }
};

/* One stream of a call. */
struct StreamCtx {
    const int32_t *const *host_in = nullptr;   /* planar host planes, or */
    const int32_t *d_in = nullptr;             /* planar device planes, d_stride elements apart, or */
    const uint8_t *pcm = nullptr;              /* interleaved little-endian PCM frames in page-locked host memory (EncodeBatchPcm) */
    uint32_t pcm_bytes = 0;                    /* bytes per sample of those frames (1: unsigned + 128, 2, 3, 4: signed) */
    uint32_t d_stride = 0;
    uint32_t num_samples = 0;
    uint8_t *data = nullptr;                   /* the stream's output buffer */
    uint32_t data_size = 0;
    uint8_t *out_direct = nullptr;             /* device-visible address of `data` (pinned / registered / device memory), or null */
    bool out_in_hbm = false;
    bool in_pinned = false;                    /* the input planes are pinned host memory: DMA reads them without staging */
    bool with_header = true;                   /* EncodeWhole / EncodeBatch: header + offset shift; block calls: neither */
    bool raw_below_shift = false;              /* a block call whose samples have bits below the handle's offset shift: silence is decided on the raw samples (SrlaCandDesc::raw_silence) */
    bool reference_call = false;               /* one of the reference's own entry points on this handle: the call reads and leaves the
                                                * handle's persistent FFT buffer as the reference's would (Impl::d_hist) */
    SRLAEncoder_EncodeBlockCallback cb = nullptr;
    /* offset left shift (srla_utility.c:177): the OR of ALL samples decides it.  Host input is encoded while the staging
     * copies are still gathering that OR: a stream whose first job does not hold all of it starts with the shift of what
     * has been seen (lshift_spec) and is encoded again in the rare case that the rest of the stream lowers it */
    uint32_t lshift = 0;
    bool lshift_final = false, lshift_spec = false, lshift_on_device = false;
    uint32_t or_mask = 0, or_covered = 0;      /* OR of samples [0, or_covered) */
    uint32_t or_dev_end = 0;                   /* samples [0, or_dev_end) have gone through that reduction */
    bool or_on_device = false;                 /* pinned input: the OR of what the jobs upload is gathered by srla_or_reduce into
                                                * d_oracc[stream] (the host looked at a short prefix only) and read back at the end */
    /* progress as the host knows it (jobs collected) */
    uint32_t write_off = 0, progress = 0;
    bool pass_started = false;                 /* a segment of the stream has been enqueued in the current pass: later ones continue at the device's running offset */
    uint32_t body = 0, chain_n = 0;            /* regular windows / history-dependent last window (chain mode) */
    SRLAApiResult rc = SRLA_APIRESULT_OK;
};

}  // namespace srla

using namespace srla;

struct Impl {
    SRLAEncoderConfig cfg{};
    SRLAEncodeParameter par{};
    bool set_parameter = false;
    uint32_t param_generation = 0;    /* bumped by SetEncodeParameter: invalidates cached job tables */
    /* Pageable input planes / output buffers are page-locked in place for the duration of a call (hipHostRegister) and then
     * read by DMA / written by the device where they lie: no host thread touches a sample.  -1: decided per call -- output
     * buffers always, input planes when the pool has too few threads to stage at the GPU's pace (N ranks sharing a CPU quota);
     * 0 / 1: never / always (SRLA_MI355X_PIN_INPLACE) */
    int pin_inplace = -1;
    bool hybrid_inplace = true;         /* SRLA_MI355X_HYBRID=0: planes locked in place cross the link as int32, all of them (stage_input) */
    bool pin_too_slow = false;          /* registration measured slower than staging would be (no huge pages): not tried again */
    static constexpr uint32_t kRunAhead = 8;      /* jobs the host may be ahead of the stage skew (bounded by the buffer sets) */
    /* Output by host-issued copies (the default where it applies: a call of more than three jobs whose streams' buffers the
     * device can reach -- pinned, registered or locked in place for the call, or device memory -- and that has no encode
     * callback): the assembly stage of a job ends with srla_pack_blocks, and when the host collects the job it has the segments'
     * bytes copied from the job's staging buffer in HBM to their place with hipMemcpyAsync on a stream that carries nothing
     * else (on this system the runtime's own copy kernel, 256 workgroups, 6.5 MB in 0.12-0.19 ms beside the other kernels).
     * With srla_stream_out behind it the assembly stream took 0.5 ms per 4 M-sample job -- 0.3 ms of them two workgroups
     * pacing themselves on PCIe stores -- against the 0.45 ms of wide kernels, and set the pace of a long stream (kernel
     * trace: stream C back to back).  SRLA_MI355X_DMA_OUT=0: the copy-out kernel everywhere. */
    bool dma_out = true;
    hipStream_t dma_stream = nullptr;
    bool call_crowded = false;          /* this call: more than three jobs, so a job's narrow kernels run beside other jobs' wide ones (SrlaJobParams::crowded) */
    bool planned_pieces = false;      /* plan_jobs cut the call's one short stream into pieces (Slot::piece) */
    bool call_solo = false;           /* the call is ONE job and no chain-mode window (Slot::solo) */
    bool call_dma = false;              /* this call: see above */
    bool dma_used = false;              /* copies may be in flight on dma_stream */
    uint32_t short_min = 786432;        /* SRLA_MI355X_SHORT_MIN: ... and no piece shorter than this many samples */
    uint32_t mid_jobs = 2;              /* SRLA_MI355X_MID_JOBS: a stream of up to this many whole jobs (and a rest) is cut into pieces like a short one (0: only streams shorter than a job) */
    uint32_t short_div = 3;             /* SRLA_MI355X_SHORT_DIV: a stream shorter than one job is cut into pieces of a job / this (4 until the pieces ran on streams of their own: 120 s 3 860 - 4 050 -> 4 400) */
    bool split_ltp_stage = true;        /* SRLA_MI355X_NO_LTP_SKEW: stage A of LTP jobs in one piece on W, as before */
    bool keep_residuals = false;      /* SRLAMI355X_ProbeBlock with a residual buffer: srla_residual_cost stores what it prices */
    uint32_t offset_lshift = 0;       /* encoder->header.offset_lshift of the reference: set by EncodeWhole, used by the block calls */
    uint32_t pack_threads = 0;

    int device = 0;                   /* HIP device of this handle (SRLAMI355X_SetDevice at the time of Create) */
    bool dev_ready = false, dev_failed = false;
    /* Buffer sets: [0, kSlots) rotate through a call's whole jobs; [kSlots, kSlots + 2) serve the tail jobs of ONE stream and
     * [kSlots, 2 kSlots) the remainder jobs of a call of many (host_plan.cpp: only where 2 kSlots <= kChainSlot, i.e. kSlots = 5);
     * [kChainSlot, kMaxSlots) are chain mode's seed / search / encode jobs.  kSlots = 5 .. 9 (SRLA_MI355X_SLOTS). */
    static constexpr uint32_t kMaxSlots = 14;
    static constexpr uint32_t kMaxRotating = kMaxSlots - 5;   /* 9: leaves the two tail sets and chain mode's three */
    static constexpr uint32_t kStreams = 3;   /* more streams than HW queues serialise badly (measured) */
    hipStream_t streams[kStreams] = {};
    hipStream_t chain_stream = nullptr; /* autocorrelation rounds of chain mode */
    hipStream_t upload = nullptr;      /* H2D of host-input jobs: a DMA queue of its own, so uploads never wait behind kernels */
    hipEvent_t ev_or = nullptr;       /* offset-shift reduction done */
    hipEvent_t ev_ref = nullptr;      /* SRLA_MI355X_TIMELINE: start of the call on the wide stream */
    bool timeline = false;
    std::string tl_log;               /* printed when the call is done: writing to stderr on the way distorts what is measured */
    void tl_printf(const char *fmt, ...) __attribute__((format(printf, 2, 3)));
    PinBuf h_or;
    uint32_t kSlots = 5;              /* job buffer sets (SRLA_MI355X_SLOTS) */
    uint64_t job_samples = 4ull << 20; /* samples per job (SRLA_MI355X_JOB_SAMPLES): fixed per-job latencies (serial solve chain, launch gaps) favour large jobs; measured best for long streams, and never worse than smaller ones for short streams */
    Slot slot[kMaxSlots];
    DevBuf d_tw, d_geoms, d_thr, d_huff, d_huffcode, d_pos, d_or, d_oracc;
    DevBuf d_welch;                    /* SrlaJobParams::welch_tab, built for welch_bps bits per sample (sync_tables) */
    uint32_t welch_bps = 0;
    bool direct_tail = true;           /* SRLA_MI355X_DIRECT_TAIL=0: a call's last job leaves through srla_stream_out like the others (round 5) */
    bool welch_table = true;           /* SRLA_MI355X_WELCH_TABLE=0: the window's weights formed per sample in the kernel (round 5) */
    DevBuf d_svr_scratch_chain;        /* the same for the chain-mode jobs: they run on the narrow stream BESIDE a regular job that keeps to the
                                        * wide stream (a call of one job, Slot::own_stream), so the two refinements may be in flight at once --
                                        * one shared region let them overwrite each other's matrices (tools/gpu_sweep.py, seed 48 case 60, round 4) */
    DevBuf d_svr_scratch;              /* srla_svr_refine_big (orders above 64, blocks above 8192 samples): kSvrGroups regions */
    static constexpr uint32_t kSvrGroups = 256;
    bool timing = true;               /* stage timing events (SRLA_MI355X_NO_TIMING drops them) */
    /* measured settings (profiles/r03, r04), constants since round 5 */
    static constexpr uint32_t kPinMinMB = 32;         /* streams of fewer MB of samples are never page-locked in place (staging them costs less than the registration) */
    static constexpr uint32_t kPairMaxItems = 6144;   /* a small job's 2048- and 4096-point autocorrelation classes go in one launch up to this many items (every job's: the autocorr stage 0.195 -> 0.21 ms per job at M, round 6, profiles/r06/ab_launch_shapes.txt) */
    static constexpr uint32_t kPoolLingerUs = 600;    /* how long the pool's workers keep looking for the next round of a short call before they sleep */
    static constexpr uint32_t kDmaTailJobs = 1;       /* the call's last n jobs leave by srla_stream_out even where the others leave by host-issued copies */
    static constexpr uint32_t kTailBoost = 4, kTailBoostJobs = 3;   /* stream-out workgroup multiplier of the call's last jobs */
    bool spin_collect = false;          /* this call: at most three jobs (its last event is polled, not slept on) */
    uint32_t timing_stride = 4;       /* every n-th job carries start events on all stages (SRLA_MI355X_TIMING_STRIDE) */
    uint32_t short_call_jobs = 0;       /* calls of ONE job so far (which of them are timed: encode_streams) */
    void read_environment();          /* host_tuning.cpp: the one place that reads the environment */
    bool no_chain = false;            /* SRLA_MI355X_NO_CHAIN */
    bool chain_trace = false;         /* SRLA_MI355X_CHAIN_TRACE */
    uint32_t env_pack_threads = 0;    /* SRLA_MI355X_PACK_THREADS (0: not set) */
    bool no_speculation = false;      /* SRLA_MI355X_NO_SPECULATION: the OR of a stream is always gathered before its first job */
    bool force_staging = false;       /* SRLA_MI355X_STAGING: never write the caller's buffer from the device */
    bool no_pack16 = false;           /* SRLA_MI355X_NO_PACK16: host input always crosses PCIe as int32 */
    /* near-tie detection and its test hooks (SrlaJobParams; SRLA_MI355X_TIE_TEST="rel,ltp,logscale,ltpbias") */
    double tie_rel = 1e-9, tie_ltp = 1e-9, tie_logscale = 1.0, tie_ltpbias = 0.0;
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
    /* Tables of SearchOptimalBlockPartitions (srla_encoder.c:336-389) for the windows of the plan's segments (each a range
     * of whole windows of one stream, or the end of it); lshift[k]: offset left shift of segment k's stream.
     * `lens` (chain mode, one segment): the job's windows are these blocks, one candidate each, instead of the regular tiling */
    void build_job(Job &job, const JobPlan &plan, const std::vector<uint32_t> &lshift, bool search, const std::vector<uint32_t> *lens = nullptr);
    SrlaJobParams job_params(const Job &job, uint32_t channel_stride, bool lshift_on_device) const;
    uint32_t windows_per_job(bool search) const;   /* bounded by scratch memory (~1.5 GB of residual scratch per slot) */
    /* the jobs of a call: streams cut at window boundaries, short streams sharing jobs */
    void plan_jobs(std::vector<JobPlan> &plan, bool search);

    /* ---- the streams of the current call (host_pipeline.cpp) ---- */
    std::vector<StreamCtx> sx;
    /* Decisions of the host libm that differ from the device's (host_ties.cpp), by (job of the call, item): the job is
     * analysed again with them.  Chain-mode jobs use the job numbers kChainJobKey + 0..2. */
    struct Override { int32_t forced_order = -1; uint32_t forced_ltp = 0; std::vector<double> svr_row; };
    std::map<uint64_t, Override> overrides;
    static constexpr uint32_t kChainJobKey = 0xFFFFFFF0u;
    static uint64_t override_key(uint32_t job, uint32_t item) { return ((uint64_t)job << 32) | item; }
    bool apply_overrides(Job &job, uint32_t jobkey);   /* patches the job's item table; true if anything was patched */
    /* Looks at the items the finished job flagged as near-ties, decides them with the host libm and records an override
     * where the device decided otherwise.  Returns the number of new overrides, -1 on error. */
    int arbitrate(Slot &s, uint32_t jobkey);
    /* the SVR refinement of one flagged item redone with the host libm; 1: the device's predictor differs (override recorded),
     * 0: confirmed, -1: error */
    int arbitrate_svr(Slot &s, uint32_t jobkey, uint32_t item);

    /* ---- staged execution of one job (host_pipeline.cpp) --------------------------------------------------
     * Three HIP streams: W carries the wide kernels (autocorr, residual_cost), N the narrow ones
     * (Levinson / order / quantiser, pricing), C the block assembly and the stream-out.  A job's stages are chained
     * with events; encode_streams enqueues the stages of consecutive jobs skewed (software pipeline), so that W always
     * has a wide kernel to run while N works through the serial stages of the neighbouring job. */
    enum { ST_A = 0, ST_B, ST_C, ST_D, ST_E, NUM_ST };
    /* brings the samples of the plan's segments to the device (staging copy + H2D for host input; device input is used
     * where it lies) and gathers the OR of what it moves */
    bool stage_input(Slot &s, const JobPlan &plan);
    /* settles the offset shift of the streams of the plan (final, or speculative: see StreamCtx) */
    void settle_lshift(const JobPlan &plan, std::vector<uint32_t> &lshift);
    /* buffers, table upload, segment table; `init_pos_of(stream)`: where a stream's first segment of this pass starts */
    bool prepare_job(Slot &s, bool want_dbg);
    /* part (stage A of a job with the long-term predictor only): 0 = the whole stage on W; 1 = the LTP-pass autocorrelation
     * on W and the pitch solve behind it on N; 2 = the LPC-pass autocorrelation on W behind the pitch solve.  The job loop
     * enqueues part 2 one iteration after part 1, so that W runs the neighbouring jobs' wide kernels while the pitch solve
     * (a latency chain of a few wavefronts, 0.6 ms per job at -P 3) works on N. */
    bool run_stage(Slot &s, int st, int part = 0);
    bool wait_job(Slot &s);
    /* one job from plan to finished bytes, synchronously (block calls, probes); arbitrates near-ties */
    bool run_job_sync(Slot &s, const JobPlan &plan, bool search, bool want_dbg, uint32_t jobkey);
    /* A finished job: checks the device's verdict and, segment by segment, moves the bytes into the stream's buffer unless
     * the device wrote them there itself, delivers the callbacks and advances the stream's progress. */
    SRLAApiResult finish_job(Slot &s);
    srla::StreamInfo stream_info(const StreamCtx &st) const;
    bool write_header(StreamCtx &st);
    /* The body shared by EncodeWhole, EncodeWholeDevice, EncodeBatch and the block calls: encodes the streams in `sx`. */
    SRLAApiResult encode_streams(bool search);
    /* classifies the stream's buffers (pinned? device memory?) */
    void classify_buffers(StreamCtx &st, std::vector<const void *> &held);

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
    struct ChainCall { uint32_t job, item, pass, n, nfft, round; int32_t src; uint32_t dump, lags; bool tainted; };
    struct ChainLaunch { uint32_t round, pass, cls, first, count; };
    struct ChainJob {
        std::vector<SrlaAutocorrItem> list;
        std::vector<ChainLaunch> launches;
        std::vector<uint32_t> select;     /* per item: the round of its LTP-lag call */
        std::vector<uint32_t> select_b;   /* SVR on: per item the round of its LPC-lag call = the round its solve chain and refinement run in */
        uint32_t rounds = 0;
    };
    std::vector<ChainCall> chain_calls;
    uint64_t chain_pool_used = 0;
    std::vector<uint32_t> chain_tab;      /* gather table for the LTP lags beyond a short FFT (SrlaAutocorrItem::chain_lags) */
    size_t chain_tab_uploaded = 0;
    DevBuf d_chain_pool, d_chain_tab, d_chain_list[3], d_chain_select[3], d_chain_select_b[3];
    PinBuf h_chain_up[3];                 /* staging of a chain job's tables (chain_stage_a) */
    PinBuf h_chain_recs;                  /* a history-mode window's partition on its way back */
    /* the reference's calls for the candidates of `job`, appended in its order; silent(off, n): the block is all zero */
    void chain_append(uint32_t jobidx, const Job &job, const std::function<bool(uint32_t, uint32_t)> &silent);
    void chain_build(uint32_t jobidx, Job &job, ChainJob &cj);
    /* the refinement writes the reference's buffer too -- where it runs at all: with preset 0 the chosen order is 0 and
     * srla_encoder.c:1084 never calls it, so there is no such writer in the sequence of calls */
    bool chain_svr() const { return par.num_svr_filter_learning_iteration > 0 && preset_order() > 0; }
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
        uint32_t stream = 0;              /* index into sx */
        std::vector<int32_t> tail_smp, seed_smp;
        uint32_t seed_n = 0, seed_off = 0;
        ChainJob cq, cs, ce;
    } chain;
    static constexpr uint32_t kChainSlot = kMaxSlots - 3;   /* seed, search, encode */
    static_assert(kMaxRotating + 2 <= kChainSlot && 5 + 5 <= kChainSlot, "the tail sets (any kSlots) and the remainder sets (kSlots = 5) lie below the chain-mode sets");
    bool chain_silent(const std::vector<int32_t> &v, uint32_t total, uint32_t off, uint32_t n) const;
    void chain_slot_defaults(Slot &s);
    bool chain_make_job(Slot &s, uint32_t s0, uint32_t ns, bool search, const std::vector<uint32_t> *lens);
    bool chain_begin(uint32_t seed_off, uint32_t seed_n);
    bool chain_settle_ties();
    SRLAApiResult chain_collect();
    /* the encode job up to its pricing */
    bool chain_encode_ad();
    bool chain_encode_e();
    bool chain_search_done();         /* has the search job been priced (so that the encode job can be enqueued without waiting)? */

    /* ---- history mode: every window in the reference's own call order -------------------------------------------
     * Chain mode covers streams whose ONLY history-dependent window is the last one.  Two parameter regimes make blocks
     * anywhere in the stream depend on the calls before them:
     *   an odd minimum block (every odd block inherits its middle word, lpc.c:260-264), and
     *   the long-term predictor with blocks of at most 256 samples (263 lags copied out of a shorter FFT, lpc.c:371-373).
     * Such streams are encoded window by window, each window as the reference does it -- search phase (every candidate),
     * Dijkstra, encode phase (the chosen partition analysed again, srla_encoder.c:1646-1698) -- with the chain-mode
     * machinery, and the state the reference carries from call to call, its persistent FFT buffer (lpc.c:58,211), is kept
     * on the device: the first kHistoryWords doubles of the chain pool ARE that buffer as the next phase finds it
     * (srla_chain_commit folds a phase's calls into it).  chain_calls[0] then stands for "whatever the buffer holds":
     * a call that reaches back beyond its own phase reads the buffer.  Slow (two small jobs and two host round trips per
     * window) and exact; the windows of a stream are a dependency chain in these regimes, in the reference as here. */
    uint32_t chain_tail(uint32_t num_samples, bool search) const;   /* the history-dependent last window outside the history regimes (0: none) */
    /* bit-identity that cannot be promised (SRLAMI355X_NONIDENTICAL_*): said on stderr, counted in the statistics */
    uint32_t nonidentical_reasons(uint32_t num_samples) const;
    void note_nonidentical(uint32_t num_samples);
    void note_reasons(uint32_t reasons);
    static std::string nonidentical_text(uint32_t reasons);
    uint32_t warned_reasons = 0;
    static constexpr uint32_t kHistoryWords = 65536;
    /* ---- the buffer from call to call ------------------------------------------------------------------------------
     * The reference keeps ONE calculator per encoder (srla_encoder.c: encoder->lpcc), so the buffer outlives a call: what a
     * handle encoded before can decide a later call's history-dependent blocks (the same stream behind another one under
     * `-B 4095 -V 0` differs from a fresh handle's in 5 of 7 cases: profiles/r04/handle_reuse_probe.txt).  d_hist is that
     * buffer as the last call of one of the reference's entry points left it; hist_exact counts the leading words of it that are
     * known to equal the reference's.  Calls that run in history mode (every call of the history regimes, and every call of at
     * most one window) start from it and leave it exactly; a regular call of several windows does not track it (hist_exact = 0
     * afterwards: its stream's own blocks never reach back beyond their own window and the one before).  A call whose
     * history-dependent read lands in words that are not known is counted (SRLAMI355X_NONIDENTICAL_HANDLE_HISTORY). */
    DevBuf d_hist;
    bool pool_holds_hist = false;         /* the buffer stands in the chain pool's head (after a tracked history-mode call) instead of in d_hist */
    bool hist_leave_pool();
    bool hist_fresh = true;               /* nothing has been encoded on this handle: the buffer is all zero (not even allocated) */
    uint32_t hist_exact = kHistoryWords;  /* words [0, hist_exact) of d_hist are the reference's */
    uint32_t buf_exact = kHistoryWords;   /* the same for the pool's head while a history-mode call runs */
    bool call_tainted = false;            /* a call of the running API call read a word that is not known */
    /* ... and behind calls that ran through the regular pipeline (which keeps no buffer): a call that no earlier call can reach
     * into -- a stream of several windows; a call of at most one window without an odd-length or short long-term-predictor block --
     * is encoded by the regular pipeline and leaves a CAPTURE: its samples (of a stream of several windows the last audible window
     * and the one before it: every full window's search analyses a candidate of the maximum block, whose transform rewrites every
     * word a later call can read; nothing in a regular regime reaches back further), its parameters, its shift, copied into pinned
     * host memory while the call waits for its last job.  When a later call on the handle is about to READ the buffer -- a
     * history-mode call -- the pending captures are first encoded once more, oldest first, in history mode under their own
     * parameters, bytes discarded (replay_pending), which leaves the buffer as the reference's calls left it.  A capture whose
     * largest transform is certain to run (`rewrites`) drops the older captures that cannot write beyond it (`extent`): a stream
     * handed over block by block keeps ONE pending capture.  Calls that never read the buffer pay a copy of at most two windows. */
    struct Capture {
        SRLAEncodeParameter par{};
        uint32_t lshift = 0, n = 0, nch = 0;
        uint32_t extent = 0;              /* no word at or beyond it is written by the call */
        uint32_t rewrites = 0;            /* every word below it is (0: not certain) */
        bool search = false;              /* EncodeOptimalPartitionedBlock / EncodeWhole with a block division search; else EncodeBlock's way */
        bool multi = false;               /* the tail of a stream of several windows: what its earlier windows left below `extent` is not kept */
        bool raw_below_shift = false;     /* a block call on samples with bits below the handle's offset shift (StreamCtx::raw_below_shift) */
        PinBuf smp;                       /* nch planes of n samples */
    };
    struct TailState { bool copied = false, silent_stream = false, from_device = false, whole_stream = false; uint32_t longest = 0, window_len = 0; Capture c; } tail;    /* the running call's */
    std::vector<Capture *> pending, spare;
    static constexpr size_t kMaxPending = 8;
    bool push_capture();                  /* tail.c -> pending */
    void drop_pending();
    /* Device -> host read-backs of a few bytes to a few MB (near-tie numbers, partitions, probe records) go through a page-locked
     * bounce buffer: a copy straight into pageable memory is, on this platform, a GPU write into the caller's pages, and the test
     * process saw it fail ("Memory access fault ... Write access to a read-only page", a heap page still marked copy-on-write after
     * an earlier fork of the process) once in several runs of the whole suite. */
    PinBuf h_bounce;
    bool d2h(void *dst, const void *src, size_t bytes);
    bool d2h_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height);
    bool want_block_price = false;        /* SRLAEncoder_ComputeBlockSize with more than two channels: the search's price of the block */
    uint32_t block_price = 0;
    bool replaying = false;
    std::vector<uint8_t> replay_out;
    void mark_raw_silence(Job &job);      /* SrlaCandDesc::raw_silence of a job of one stream whose samples do not obey its shift */
    bool keep_tail(const StreamCtx &st, bool search);   /* the samples a capture needs (before the call's last wait) */
    bool replay_pending();
    bool replay_one(Capture &c);
    bool history_regime(bool search) const;
    void history_phase_reset();
    bool history_commit(uint32_t jobidx, hipStream_t stream);
    SRLAApiResult history_window(uint32_t stream, uint32_t pos, uint32_t n, bool search);
    SRLAApiResult history_encode(bool search);
};

#endif /* SRLA_HOST_IMPL_H */
