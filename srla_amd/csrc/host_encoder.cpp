/*
 * host_encoder.cpp -- host runtime behind the SRLAEncoder_* C ABI (include/srla_mi355x.h).
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
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/srla_mi355x.h"
#include "device_layout.h"
#include "host_pack.h"
#include "host_tables.h"
#include "kernels.h"

#define SRLA_HANDLE_MAGIC 0x53524C41u /* 'SRLA' */
#define SRLA_MAX_FFT      8192u       /* largest block the LDS-resident FFT handles */

static_assert(sizeof(SrlaItemResult) == SRLAMI355X_ITEM_RECORD_BYTES, "record size");
static_assert(SRLA_DBG_STRIDE == SRLAMI355X_DEBUG_DOUBLES, "debug stride");

struct SRLAEncoder {
    uint32_t magic;
    uint8_t alloced_by_own;
    void *work;
    struct Impl *impl;
};

namespace {

using Clock = std::chrono::steady_clock;
inline double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

int g_device_index = 0;

/* max LPC order per preset, libs/srla_internal/src/srla_internal.c:30-38 */
const uint32_t kPresetOrder[SRLA_NUM_PARAMETER_PRESETS] = { 0, 8, 16, 32, 64, 128, 255 };

#define HIP_OK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            fprintf(stderr, "[srla-mi355x] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e__), \
                    __FILE__, __LINE__);                                                           \
            return false;                                                                          \
        }                                                                                          \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes)
    {
        if (bytes <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        HIP_OK(hipMalloc(&p, want));
        cap = want;
        return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes)
    {
        if (bytes <= cap) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        HIP_OK(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

/* ---- staging copy: pageable planes -> pinned buffer, with the OR of the samples as a by-product ---------
 * The pinned buffer is only read by the DMA engine afterwards, so the stores bypass the cache (no read-for-ownership
 * traffic): about 1.5x the throughput of memcpy for this pattern. */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static uint32_t copy_or_avx2(int32_t *dst, const int32_t *src, size_t n)
{
    uint32_t m = 0;
    size_t k = 0;
    while (k < n && (reinterpret_cast<uintptr_t>(dst + k) & 31u)) { const int32_t x = src[k]; dst[k] = x; m |= (uint32_t)x; k++; }
    __m256i acc = _mm256_setzero_si256();
    for (; k + 32 <= n; k += 32) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k + 8));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k + 16));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k + 24));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + k), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + k + 8), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + k + 16), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + k + 24), d);
        acc = _mm256_or_si256(acc, _mm256_or_si256(_mm256_or_si256(a, b), _mm256_or_si256(c, d)));
    }
    alignas(32) uint32_t lanes[8];
    _mm256_store_si256(reinterpret_cast<__m256i *>(lanes), acc);
    for (int i = 0; i < 8; i++) m |= lanes[i];
    for (; k < n; k++) { const int32_t x = src[k]; dst[k] = x; m |= (uint32_t)x; }
    _mm_sfence();
    return m;
}
#endif
/* The same for streams of at most 16 bits: the staging copy packs the samples to int16, which halves what crosses PCIe
 * (the upload of a 600 s stream, 230 MB as int32, took as long as its whole encode); the device widens them again.
 * *wide gets a non-zero value if a sample does not fit (the caller then stages that job as int32). */
#if defined(__x86_64__)
__attribute__((target("avx2"))) static uint32_t pack16_or_avx2(int16_t *dst, const int32_t *src, size_t n, uint32_t *wide)
{
    uint32_t m = 0, w = 0;
    size_t k = 0;
    while (k < n && (reinterpret_cast<uintptr_t>(dst + k) & 31u)) {
        const int32_t x = src[k]; dst[k] = (int16_t)x; m |= (uint32_t)x; w |= ((uint32_t)x + 32768u) & 0xFFFF0000u; k++;
    }
    __m256i acc = _mm256_setzero_si256(), accw = _mm256_setzero_si256();
    const __m256i bias = _mm256_set1_epi32(32768);
    for (; k + 32 <= n; k += 32) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k + 8));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k + 16));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + k + 24));
        /* packs works inside the 128-bit halves: put the quarters back in order */
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + k), _mm256_permute4x64_epi64(_mm256_packs_epi32(a, b), 0xD8));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + k + 16), _mm256_permute4x64_epi64(_mm256_packs_epi32(c, d), 0xD8));
        acc = _mm256_or_si256(acc, _mm256_or_si256(_mm256_or_si256(a, b), _mm256_or_si256(c, d)));
        accw = _mm256_or_si256(accw, _mm256_or_si256(_mm256_or_si256(_mm256_add_epi32(a, bias), _mm256_add_epi32(b, bias)),
                                                     _mm256_or_si256(_mm256_add_epi32(c, bias), _mm256_add_epi32(d, bias))));
    }
    alignas(32) uint32_t lanes[8];
    _mm256_store_si256(reinterpret_cast<__m256i *>(lanes), acc);
    for (int i = 0; i < 8; i++) m |= lanes[i];
    _mm256_store_si256(reinterpret_cast<__m256i *>(lanes), accw);
    for (int i = 0; i < 8; i++) w |= lanes[i] & 0xFFFF0000u;
    for (; k < n; k++) { const int32_t x = src[k]; dst[k] = (int16_t)x; m |= (uint32_t)x; w |= ((uint32_t)x + 32768u) & 0xFFFF0000u; }
    _mm_sfence();
    *wide = w;
    return m;
}
#endif
static uint32_t pack16_or(int16_t *dst, const int32_t *src, size_t n, uint32_t *wide)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return pack16_or_avx2(dst, src, n, wide);
#endif
    uint32_t m = 0, w = 0;
    for (size_t k = 0; k < n; k++) { const int32_t x = src[k]; dst[k] = (int16_t)x; m |= (uint32_t)x; w |= ((uint32_t)x + 32768u) & 0xFFFF0000u; }
    *wide = w;
    return m;
}

static uint32_t copy_or(int32_t *dst, const int32_t *src, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return copy_or_avx2(dst, src, n);
#endif
    uint32_t m = 0;
    for (size_t k = 0; k < n; k++) { const int32_t x = src[k]; dst[k] = x; m |= (uint32_t)x; }
    return m;
}

/* ---- a tiny persistent thread pool for the bit pack ------------------------------------- */
class Pool {
public:
    explicit Pool(unsigned n) : stop_(false), pending_(0)
    {
        for (unsigned i = 0; i + 1 < n; i++) workers_.emplace_back([this] { loop(); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    /* runs fn(i) for i in [0, count), the caller participates */
    void parallel_for(uint32_t count, const std::function<void(uint32_t)> &fn)
    {
        if (count == 0) return;
        if (workers_.empty() || count == 1) { for (uint32_t i = 0; i < count; i++) fn(i); return; }
        {
            std::lock_guard<std::mutex> l(m_);
            fn_ = &fn; next_.store(0); count_ = count; pending_ = (unsigned)workers_.size(); gen_++;
            gen_atomic_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
        run_chunk();
        std::unique_lock<std::mutex> l(m_);
        done_cv_.wait(l, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
    unsigned size() const { return (unsigned)workers_.size() + 1; }

private:
    void run_chunk()
    {
        for (;;) {
            const uint32_t i = next_.fetch_add(1);
            if (i >= count_) break;
            (*fn_)(i);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                /* spin briefly before sleeping: pack rounds arrive every few hundred microseconds */
                for (int spin = 0; spin < 200 && gen_atomic_.load(std::memory_order_acquire) == seen && !stop_; spin++) {
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                }
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            run_chunk();
            {
                std::lock_guard<std::mutex> l(m_);
                if (--pending_ == 0) done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    bool stop_;
    const std::function<void(uint32_t)> *fn_ = nullptr;
    std::atomic<uint32_t> next_{ 0 };
    uint32_t count_ = 0;
    unsigned pending_;
    uint64_t gen_ = 0;
    std::atomic<uint64_t> gen_atomic_{ 0 };
};

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

}  // namespace

struct Impl {
    SRLAEncoderConfig cfg{};
    SRLAEncodeParameter par{};
    bool set_parameter = false;
    uint32_t param_generation = 0;    /* bumped by SetEncodeParameter: invalidates cached job tables */
    uint32_t offset_lshift = 0;       /* encoder->header.offset_lshift of the reference */
    uint32_t pack_threads = 0;

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
    void tl_printf(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
    {
        char buf[640];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        tl_log += buf;
    }
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

    ~Impl()
    {
        delete pool;
        if (dev_ready) {
            (void)hipSetDevice(g_device_index);
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

    uint32_t preset_order() const { return kPresetOrder[par.preset]; }
    uint32_t num_variants() const { return par.num_channels + (par.num_channels >= 2 ? 2u : 0u); }
    bool search_enabled() const { return par.min_num_samples_per_block != par.max_num_samples_per_block; }

    bool init_device()
    {
        if (dev_ready) return true;
        if (dev_failed) return false;
        dev_failed = true;
        if (const char *e = getenv("SRLA_MI355X_SLOTS")) { const int v = atoi(e); if (v >= 2 && v + 5 <= (int)kMaxSlots) kSlots = (uint32_t)v; }   /* + 2 slots for the tail jobs, + 3 for chain mode */
        if (const char *e = getenv("SRLA_MI355X_JOB_SAMPLES")) { const long long v = atoll(e); if (v >= 65536) job_samples = (uint64_t)v; }
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            fprintf(stderr, "[srla-mi355x] no HIP device available: the MI355X encode path cannot run "
                            "(there is no CPU fallback)\n");
            return false;
        }
        HIP_OK(hipSetDevice(g_device_index));
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

    uint32_t geom_for(uint32_t n)
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

    /* tables are shared by both slots: wait for everything in flight before re-allocating them */
    bool sync_tables()
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

    /* LDS carve-up of srla_residual_cost for the largest FFT size of a job */
    SrlaLdsPlan lds_plan(uint32_t nfft) const
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

    /* Candidate table of SearchOptimalBlockPartitions (srla_encoder.c:336-389) for the windows
     * [first sample s0, s0+ns) of a stream; ns ends on a window boundary or at the stream end. */
    /* `lens` (chain mode): the job's windows are these blocks, one candidate each, instead of the regular tiling */
    void build_job(Job &job, uint32_t s0, uint32_t ns, bool search, const std::vector<uint32_t> *lens = nullptr)
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

    SrlaJobParams job_params(const Job &job, uint32_t channel_stride) const
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
        { static const char *e = getenv("SRLA_MI355X_K3_STOP"); jp.out_stride = e ? (uint32_t)atoi(e) : 0u; }   /* diagnostics */
        jp.lshift_dev = lshift_on_device ? (d_or.as<uint32_t>() + 1) : nullptr;
        return jp;
    }

    /* Enqueue one job on its slot's stream.  d_in: device pointer to channel 0 of the job's first
     * sample, or nullptr to upload host_in (planar pointers, absolute stream positions). */
    /* ---- staged execution of one job --------------------------------------------------------------------
     * Three streams: W carries the wide kernels (autocorr, residual_cost, pack_blocks), N the narrow ones
     * (Levinson / order / quantiser, pricing), C the D2H copies.  A job's stages are chained with events;
     * encode_stream enqueues the stages of consecutive jobs skewed (software pipeline), so that W always has
     * a wide kernel to run while N works through the serial stages of the neighbouring job. */
    enum { ST_A = 0, ST_B, ST_C, ST_D, ST_E, NUM_ST };

    bool prepare_job(Slot &s, const int32_t *d_in, uint32_t d_stride, const int32_t *const *host_in, bool want_dbg)
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

    bool run_stage(Slot &s, int st)
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

    /* all stages of one job back to back (single-block calls, probes) */
    bool launch_job(Slot &s, const int32_t *d_in, uint32_t d_stride, const int32_t *const *host_in, bool want_dbg)
    {
        in_pinned = false;
        if (!prepare_job(s, d_in, d_stride, host_in, want_dbg)) return false;
        for (int st = 0; st < NUM_ST; st++) if (!run_stage(s, st)) return false;
        return true;
    }

    bool wait_job(Slot &s)
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
    void chain_append(uint32_t jobidx, const Job &job, const std::function<bool(uint32_t, uint32_t)> &silent)
    {
        const uint32_t nch = par.num_channels, nv = num_variants();
        std::vector<uint32_t> pass1_round(job.items.size(), 0);
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
                    if (c.n & 1u) {
                        const uint32_t mid = c.n >> 1;
                        for (int64_t j = (int64_t)chain_calls.size() - 1; j >= 0; j--)
                            if (chain_calls[(size_t)j].nfft > mid) { c.src = (int32_t)j; break; }
                        if (c.src >= 0 && chain_calls[(size_t)c.src].job == jobidx) {
                            /* inside a round the LTP-lag launches come first, then the pitch solve, then the LPC-lag launches */
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
                            lo = hi;
                            if (sc.job == jobidx) round = std::max(round, sc.round + 1);
                        }
                    }
                    c.round = round;
                    if (pass == 1) pass1_round[item] = round;
                    chain_calls.push_back(c);
                }
            }
        }
    }

    void chain_build(uint32_t jobidx, const Job &job, ChainJob &cj)
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
            return ai;
        };
        auto cls_of = [](uint32_t nfft) { return (nfft <= 1024u) ? 0u : ((nfft <= 2048u) ? 1u : ((nfft <= 4096u) ? 2u : 3u)); };
        cj.select.assign(std::max<size_t>(1, job.items.size()), 0xFFFFFFFFu);
        cj.rounds = 1;
        for (const ChainCall &c : chain_calls) {
            if (c.job != jobidx) continue;
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

    /* stage A of a chain job: the autocorrelation launches round by round */
    bool chain_stage_a(Slot &s, uint32_t jobidx, const ChainJob &cj)
    {
        /* many small dependent launches: on a stream of their own, so that the regular jobs' wide kernels do not queue
         * behind them */
        hipStream_t W = s.own_stream;
        const SrlaJobParams &jp = s.jp;
        if (!d_chain_list[jobidx].ensure(std::max<size_t>(1, cj.list.size()) * sizeof(SrlaAutocorrItem))) return false;
        if (!d_chain_select[jobidx].ensure(cj.select.size() * 4)) return false;
        if (!cj.list.empty()) HIP_OK(hipMemcpy(d_chain_list[jobidx].p, cj.list.data(), cj.list.size() * sizeof(SrlaAutocorrItem), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_chain_select[jobidx].p, cj.select.data(), cj.select.size() * 4, hipMemcpyHostToDevice));
        if (chain_tab.size() > chain_tab_uploaded) {
            /* only the new entries: kernels of the jobs before may still be reading theirs */
            if (chain_tab.size() * 4 > d_chain_tab.cap) return false;
            HIP_OK(hipMemcpy(d_chain_tab.as<uint32_t>() + chain_tab_uploaded, chain_tab.data() + chain_tab_uploaded,
                             (chain_tab.size() - chain_tab_uploaded) * 4, hipMemcpyHostToDevice));
            chain_tab_uploaded = chain_tab.size();
        }
        /* prepare_job put the descriptor uploads on the wide stream: everything after this stage must see them */
        HIP_OK(hipEventRecord(s.t0[ST_A], streams[0]));
        HIP_OK(hipStreamWaitEvent(W, s.t0[ST_A], 0));
        if (lshift_on_device) HIP_OK(hipStreamWaitEvent(W, ev_or, 0));
        if (s.used_h2d) HIP_OK(hipStreamWaitEvent(W, s.ev_in, 0));
        static const int kClass[4] = { 0, 1, 2, 4 };
        int rc = 0;
        size_t li = 0;
        for (uint32_t r = 0; r < cj.rounds; r++)
            for (int pass = (par.ltp_order > 0) ? 1 : 0; pass >= 0; pass--) {
                bool any = false;
                for (; li < cj.launches.size() && cj.launches[li].round == r && cj.launches[li].pass == (uint32_t)pass; li++) {
                    const ChainLaunch &l = cj.launches[li];
                    rc |= srla_launch_autocorr(W, kClass[l.cls], &jp, s.in_cur, s.d_items.as<SrlaItemDesc>(), d_geoms.as<SrlaGeom>(), d_tw.p,
                                               (uint32_t)pass, s.d_results.as<SrlaItemResult>(), s.d_lags.as<double>(), nullptr,
                                               d_chain_list[jobidx].as<SrlaAutocorrItem>() + l.first, l.count, nullptr, nullptr,
                                               d_chain_pool.as<double>(), d_chain_tab.as<uint32_t>());
                    any = true;
                }
                if (pass == 1 && any)
                    rc |= srla_launch_pitch_solve(W, &jp, s.d_lags.as<double>(), s.d_results.as<SrlaItemResult>(), nullptr, nullptr,
                                                  d_chain_select[jobidx].as<uint32_t>(), r);
            }
        HIP_OK(hipEventRecord(s.t1[ST_A], W));
        if (rc != 0) { fprintf(stderr, "[srla-mi355x] kernel launch failed in a chain stage\n"); return false; }
        return true;
    }

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

    bool chain_silent(const std::vector<int32_t> &v, uint32_t total, uint32_t off, uint32_t n) const
    {
        for (uint32_t ch = 0; ch < par.num_channels; ch++) {
            const int32_t *p = v.data() + (size_t)ch * total + off;
            for (uint32_t i = 0; i < n; i++) if (p[i] != 0) return false;
        }
        return true;
    }

    void chain_slot_defaults(Slot &s)
    {
        s.own_stream = streams[1];   /* the narrow stream: a stream of their own ended up sharing a hardware queue with the wide one and waited for the whole stream (measured) */
        s.out_direct = nullptr; s.out_first = 1; s.out_init_pos = 0; s.out_limit = 0xFFFFFFFFu; s.timed = false; s.out_boost = 1;
    }

    bool chain_begin(uint32_t seed_off, uint32_t seed_n)
    {
        ChainRun &c = chain;
        const uint32_t nch = par.num_channels, nv = num_variants(), passes = par.ltp_order > 0 ? 2u : 1u;
        /* which blocks are all zero decides which calls exist: look at the samples */
        auto fetch = [&](uint32_t off, uint32_t n, std::vector<int32_t> &dst) -> bool {
            dst.resize((size_t)nch * n);
            for (uint32_t ch = 0; ch < nch; ch++) {
                if (c.host_in) memcpy(dst.data() + (size_t)ch * n, c.host_in[ch] + off, (size_t)n * 4);
                else if (hipMemcpy(dst.data() + (size_t)ch * n, c.d_in + (size_t)ch * c.d_stride + off, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
            }
            return true;
        };
        c.seed_n = seed_n;
        if (!fetch(c.tail_start, c.tail_n, c.tail_smp) || (seed_n && !fetch(seed_off, seed_n, c.seed_smp))) return false;
        const std::function<bool(uint32_t, uint32_t)> silent_tail = [&](uint32_t off, uint32_t n) { return chain_silent(c.tail_smp, c.tail_n, off, n); };
        const std::function<bool(uint32_t, uint32_t)> silent_seed = [&](uint32_t off, uint32_t n) { return chain_silent(c.seed_smp, c.seed_n, off, n); };
        chain_calls.clear();
        chain_pool_used = 0;
        chain_tab.clear();
        chain_tab_uploaded = 0;
        Slot &q = slot[kChainSlot], &sj = slot[kChainSlot + 1], &e = slot[kChainSlot + 2];
        if (seed_n) {
            const std::vector<uint32_t> lens{ seed_n };
            build_job(q.job, seed_off, seed_n, false, &lens);
            chain_append(0, q.job, silent_seed);
        }
        if (c.search) {
            build_job(sj.job, c.tail_start, c.tail_n, true);
            chain_append(1, sj.job, silent_tail);
        } else {
            const std::vector<uint32_t> lens{ c.tail_n };
            build_job(e.job, c.tail_start, c.tail_n, false, &lens);
            chain_append(2, e.job, silent_tail);
        }
        {
            /* the encode job's calls are not known yet when searching: its blocks tile the window, an FFT is shorter
             * than twice its block (or the smallest FFT size) */
            const uint32_t max_parts = c.search ? (c.tail_n + par.min_num_samples_per_block - 1) / par.min_num_samples_per_block : 0u;
            const uint64_t bound = chain_pool_used + (uint64_t)nv * passes * (2ull * c.tail_n + 64ull * max_parts);
            if (!d_chain_pool.ensure(bound * sizeof(double))) return false;
            const size_t tab_bound = chain_tab.size() + (size_t)std::max(1u, max_parts) * nv * SRLA_LTP_LAGS;
            if (!d_chain_tab.ensure(tab_bound * 4)) return false;
        }
        if (seed_n) {
            chain_build(0, q.job, c.cq);
            chain_slot_defaults(q);
            if (!prepare_job(q, c.d_in ? c.d_in + seed_off : nullptr, c.d_stride, c.host_in, false) || !chain_stage_a(q, 0, c.cq)) return false;
        }
        if (c.search) {
            chain_build(1, sj.job, c.cs);
            chain_slot_defaults(sj);
            if (!prepare_job(sj, c.d_in ? c.d_in + c.tail_start : nullptr, c.d_stride, c.host_in, false) || !chain_stage_a(sj, 1, c.cs)) return false;
            for (int st = ST_B; st <= ST_D; st++) if (!run_stage(sj, st)) return false;
        }
        c.begun = true;
        return true;
    }

    /* the encode job up to its pricing; `first_job`: nothing was encoded before the window */
    bool chain_encode_ad(uint8_t *out_direct, uint32_t init_pos, uint32_t data_size, bool first_job)
    {
        ChainRun &c = chain;
        Slot &q = slot[kChainSlot], &sj = slot[kChainSlot + 1], &e = slot[kChainSlot + 2];
        if (c.search) {
            const auto tw = Clock::now();
            if (hipEventSynchronize(sj.t1[ST_D]) != hipSuccess) return false;
            static const bool trace = getenv("SRLA_MI355X_CHAIN_TRACE") != nullptr;
            if (trace) fprintf(stderr, "[chain] waited %.3f ms for the search job (%u rounds)\n", ms_since(tw), c.cs.rounds);
            const SrlaWindowDesc &wd = sj.job.windows[0];
            std::vector<SrlaBlockRecord> recs(wd.num_nodes - 1);
            if (hipMemcpy(recs.data(), sj.d_blocks.as<SrlaBlockRecord>() + wd.block_base, recs.size() * sizeof(SrlaBlockRecord), hipMemcpyDeviceToHost) != hipSuccess)
                return false;
            std::vector<uint32_t> lens;
            uint32_t covered = 0;
            for (const SrlaBlockRecord &r : recs) if (r.valid) { lens.push_back(r.n); covered += r.n; }
            if (covered != c.tail_n) { fprintf(stderr, "[srla-mi355x] internal error: the tail window's partitions cover %u of %u samples\n", covered, c.tail_n); return false; }
            sj.busy = false;
            const std::function<bool(uint32_t, uint32_t)> silent_tail = [&](uint32_t off, uint32_t n) { return chain_silent(c.tail_smp, c.tail_n, off, n); };
            build_job(e.job, c.tail_start, c.tail_n, false, &lens);
            chain_append(2, e.job, silent_tail);
            if (chain_pool_used * sizeof(double) > d_chain_pool.cap) return false;
        }
        q.busy = false;
        chain_build(2, e.job, c.ce);
        chain_slot_defaults(e);
        e.out_direct = out_direct; e.out_first = first_job ? 1u : 0u; e.out_init_pos = init_pos; e.out_limit = data_size; e.out_boost = tail_boost;
        if (!prepare_job(e, c.d_in ? c.d_in + c.tail_start : nullptr, c.d_stride, c.host_in, false) || !chain_stage_a(e, 2, c.ce)) return false;
        for (int st = ST_B; st <= ST_D; st++) if (!run_stage(e, st)) return false;
        c.ad_done = true;
        return true;
    }

    bool chain_encode_e() { return run_stage(slot[kChainSlot + 2], ST_E); }

    /* has the search job been priced (so that the encode job can be enqueued without waiting)? */
    bool chain_search_done()
    {
        if (!chain.begun) return false;
        if (!chain.search) return true;
        const hipError_t e = hipEventQuery(slot[kChainSlot + 1].t1[ST_D]);
        if (e != hipSuccess) (void)hipGetLastError();
        return e == hipSuccess;
    }

    /* A finished job: check the device's verdict, move the bytes to `data + write_off` unless the device wrote
     * them there itself, and report the per-window sizes. */
    SRLAApiResult finish_job(Slot &s, uint8_t *data, uint32_t write_off, uint32_t *written, const uint32_t **window_bytes)
    {
        const auto t0 = Clock::now();
        const SrlaJobInfo info = *s.h_info.as<SrlaJobInfo>();
        static const bool diag = getenv("SRLA_MI355X_K3_STOP") != nullptr;   /* timing experiments: the stream is garbage */
        if (diag) { *written = 0; *window_bytes = reinterpret_cast<const uint32_t *>(s.h_info.as<SrlaJobInfo>() + 1); return SRLA_APIRESULT_OK; }
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

    srla::StreamInfo stream_info(uint32_t num_samples) const
    {
        srla::StreamInfo si;
        si.num_channels = par.num_channels; si.bits_per_sample = par.bits_per_sample;
        si.sampling_rate = par.sampling_rate; si.num_samples = num_samples; si.offset_lshift = offset_lshift;
        si.max_block = par.max_num_samples_per_block; si.preset = par.preset; si.ltp_order = par.ltp_order;
        return si;
    }

    /* windows per job: bounded by scratch memory (~1.5 GB of residual scratch per slot) */
    uint32_t windows_per_job(bool search) const
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

    /* The body shared by EncodeWhole (host input) and EncodeWholeDevice. */
    SRLAApiResult encode_stream(const int32_t *const *host_in, const int32_t *d_in, uint32_t d_stride,
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
};

/* ============================================================================================
 * C ABI
 * ========================================================================================== */
namespace {

int32_t work_size_of(const SRLAEncoderConfig *config)
{
    /* validity rules of srla_encoder.c:468-496 */
    if (config == NULL) return -1;
    if (config->max_num_samples_per_block == 0 || config->min_num_samples_per_block == 0
        || config->max_num_lookahead_samples == 0 || config->max_num_channels == 0) return -1;
    if (config->max_num_parameters > config->max_num_samples_per_block) return -1;
    if (config->min_num_samples_per_block > config->max_num_samples_per_block) return -1;
    if (config->max_num_lookahead_samples < config->max_num_samples_per_block) return -1;
    return (int32_t)(sizeof(SRLAEncoder) + 16);
}

Impl *impl_of(SRLAEncoder *e) { return (e && e->magic == SRLA_HANDLE_MAGIC) ? e->impl : nullptr; }

}  // namespace

extern "C" {

const char *SRLAMI355X_Version(void) { return "srla-mi355x 0.1 (gfx950 HIP; SRLA codec 18 / format 10)"; }

int SRLAMI355X_SetDevice(int device_index)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_index < 0 || device_index >= count) return -1;
    g_device_index = device_index;
    return (hipSetDevice(device_index) == hipSuccess) ? 0 : -1;
}

SRLAApiResult SRLAEncoder_EncodeHeader(const struct SRLAHeader *header, uint8_t *data, uint32_t data_size)
{
    /* srla_encoder.c:85-165 */
    if (header == NULL || data == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (data_size < SRLA_HEADER_SIZE) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (header->num_channels == 0 || header->num_samples == 0 || header->sampling_rate == 0
        || header->bits_per_sample == 0 || header->offset_lshift >= 32 || header->max_num_samples_per_block == 0
        || header->preset >= SRLA_NUM_PARAMETER_PRESETS) return SRLA_APIRESULT_INVALID_FORMAT;
    srla::StreamInfo si;
    si.num_channels = header->num_channels; si.bits_per_sample = header->bits_per_sample;
    si.sampling_rate = header->sampling_rate; si.num_samples = header->num_samples;
    si.offset_lshift = header->offset_lshift; si.max_block = header->max_num_samples_per_block;
    si.preset = header->preset; si.ltp_order = 0;
    srla::write_stream_header(si, data);
    return SRLA_APIRESULT_OK;
}

int32_t SRLAEncoder_CalculateWorkSize(const struct SRLAEncoderConfig *config) { return work_size_of(config); }

struct SRLAEncoder *SRLAEncoder_Create(const struct SRLAEncoderConfig *config, void *work, int32_t work_size)
{
    /* srla_encoder.c:549-694 */
    uint8_t own = 0;
    if (work == NULL && work_size == 0) {
        if ((work_size = work_size_of(config)) < 0) return NULL;
        work = malloc((size_t)work_size);
        own = 1;
    }
    if (config == NULL || work == NULL || work_size < work_size_of(config) || work_size_of(config) < 0) {
        if (own) free(work);
        return NULL;
    }
    SRLAEncoder *e = reinterpret_cast<SRLAEncoder *>(((uintptr_t)work + 15u) & ~(uintptr_t)15u);
    e->magic = SRLA_HANDLE_MAGIC;
    e->alloced_by_own = own;
    e->work = work;
    e->impl = new Impl();
    e->impl->cfg = *config;
    return e;
}

void SRLAEncoder_Destroy(struct SRLAEncoder *encoder)
{
    if (encoder == NULL || encoder->magic != SRLA_HANDLE_MAGIC) return;
    delete encoder->impl;
    encoder->impl = nullptr;
    encoder->magic = 0;
    if (encoder->alloced_by_own) free(encoder->work);
}

SRLAApiResult SRLAEncoder_SetEncodeParameter(struct SRLAEncoder *encoder, const struct SRLAEncodeParameter *p)
{
    /* srla_encoder.c:710-763 */
    Impl *im = impl_of(encoder);
    if (im == nullptr || p == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (p->num_channels == 0 || p->bits_per_sample == 0 || p->sampling_rate == 0
        || p->preset >= SRLA_NUM_PARAMETER_PRESETS) return SRLA_APIRESULT_INVALID_FORMAT;
    if (p->min_num_samples_per_block == 0) return SRLA_APIRESULT_INVALID_FORMAT; /* the reference divides by it */
    if (p->min_num_samples_per_block > p->max_num_samples_per_block
        || p->num_lookahead_samples < p->max_num_samples_per_block
        || (p->num_lookahead_samples % p->min_num_samples_per_block) != 0
        || (p->ltp_order > 0 && (p->ltp_order % 2) == 0) || p->ltp_order > SRLA_MAX_LTP_ORDER)
        return SRLA_APIRESULT_INVALID_FORMAT;
    if (im->cfg.max_num_samples_per_block < p->max_num_samples_per_block
        || im->cfg.min_num_samples_per_block > p->min_num_samples_per_block
        || im->cfg.max_num_lookahead_samples < p->num_lookahead_samples
        || im->cfg.max_num_channels < p->num_channels) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    /* limits of this implementation (documented in DESIGN.md) */
    if (p->num_channels > SRLA_MAX_CH) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (p->max_num_samples_per_block > SRLA_MAX_FFT) {
        fprintf(stderr, "[srla-mi355x] max block size %u exceeds the LDS-resident FFT limit of %u samples\n",
                p->max_num_samples_per_block, SRLA_MAX_FFT);
        return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    }
    if (p->num_lookahead_samples / p->min_num_samples_per_block + 1 > SRLA_MAX_NODES) {
        fprintf(stderr, "[srla-mi355x] look-ahead / min block = %u exceeds %u search nodes\n",
                p->num_lookahead_samples / p->min_num_samples_per_block, SRLA_MAX_NODES - 1);
        return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    }
    if (p->bits_per_sample != 8 && p->bits_per_sample != 16 && p->bits_per_sample != 24) return SRLA_APIRESULT_INVALID_FORMAT;
    if (p->num_svr_filter_learning_iteration != 0) {
        fprintf(stderr, "[srla-mi355x] SVR coefficient refinement (--svr-filter-learning-iteration) is outside the "
                        "accelerated path and is not implemented\n");
        return SRLA_APIRESULT_NG;
    }
    im->par = *p;
    im->param_generation++;
    im->offset_lshift = 0;
    im->set_parameter = true;
    return SRLA_APIRESULT_OK;
}

static SRLAApiResult single_window(Impl *im, const int32_t *const *input, uint32_t num_samples, bool search,
                                   uint8_t *data, uint32_t data_size, uint32_t *output_size, bool size_only)
{
    if (!im->init_device()) return SRLA_APIRESULT_NG;
    if (!size_only)
        return im->encode_stream(input, nullptr, 0, num_samples, data, data_size, output_size, nullptr, false, search);
    /* ComputeBlockSize: run the job, read the block record, skip the pack */
    if ((num_samples & 1u) || (im->par.ltp_order > 0 && num_samples <= 256u)) {
        /* a history-dependent block (chain mode): its size is that of the block EncodeBlock would write */
        std::vector<uint8_t> tmp((size_t)num_samples * im->par.num_channels * (im->par.bits_per_sample / 8) + 64);
        return im->encode_stream(input, nullptr, 0, num_samples, tmp.data(), (uint32_t)tmp.size(), output_size, nullptr, false, search);
    }
    Slot &s = im->slot[0];
    im->build_job(s.job, 0, num_samples, false);
    s.out_direct = nullptr; s.out_first = 1; s.out_init_pos = 0; s.out_limit = 0xFFFFFFFFu; s.timed = im->timing; s.out_boost = 1;
    if (!im->launch_job(s, nullptr, 0, input, false) || !im->wait_job(s)) return SRLA_APIRESULT_NG;
    const SrlaJobInfo *info = s.h_info.as<SrlaJobInfo>();
    if (info->error != 0 || info->num_blocks != 1) return SRLA_APIRESULT_NG;
    *output_size = info->total_bytes;
    return SRLA_APIRESULT_OK;
}

SRLAApiResult SRLAEncoder_ComputeBlockSize(struct SRLAEncoder *encoder, const int32_t *const *input,
                                           uint32_t num_samples, uint32_t *output_size)
{
    /* srla_encoder.c:1477-1546 */
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || num_samples == 0 || output_size == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples > im->par.max_num_samples_per_block) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    return single_window(im, input, num_samples, false, nullptr, 0, output_size, true);
}

SRLAApiResult SRLAEncoder_EncodeBlock(struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
                                      uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
    /* srla_encoder.c:1549-1643 */
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || num_samples == 0 || data == NULL || data_size == 0 || output_size == NULL)
        return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples > im->par.max_num_samples_per_block) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    return single_window(im, input, num_samples, false, data, data_size, output_size, false);
}

SRLAApiResult SRLAEncoder_EncodeOptimalPartitionedBlock(struct SRLAEncoder *encoder, const int32_t *const *input,
                                                        uint32_t num_samples, uint8_t *data, uint32_t data_size,
                                                        uint32_t *output_size)
{
    /* srla_encoder.c:1646-1698 */
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || data == NULL || output_size == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples == 0 || num_samples > im->par.num_lookahead_samples) return SRLA_APIRESULT_NG;
    return single_window(im, input, num_samples, true, data, data_size, output_size, false);
}

SRLAApiResult SRLAEncoder_EncodeWhole(struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
                                      uint8_t *data, uint32_t data_size, uint32_t *output_size,
                                      SRLAEncoder_EncodeBlockCallback encode_callback)
{
    /* srla_encoder.c:1701-1788 */
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || data == NULL || output_size == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (data_size < SRLA_HEADER_SIZE) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (num_samples == 0) return SRLA_APIRESULT_INVALID_FORMAT;
    if (!im->init_device()) return SRLA_APIRESULT_NG;
    return im->encode_stream(input, nullptr, 0, num_samples, data, data_size, output_size, encode_callback, true,
                             im->search_enabled());
}

SRLAApiResult SRLAMI355X_EncodeWholeDevice(struct SRLAEncoder *encoder, const int32_t *d_input, uint32_t channel_stride,
                                           uint32_t num_samples, uint8_t *data, uint32_t data_size, uint32_t *output_size,
                                           SRLAEncoder_EncodeBlockCallback encode_callback)
{
    Impl *im = impl_of(encoder);
    if (im == nullptr || d_input == NULL || data == NULL || output_size == NULL || channel_stride < num_samples)
        return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (data_size < SRLA_HEADER_SIZE) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (num_samples == 0) return SRLA_APIRESULT_INVALID_FORMAT;
    if (!im->init_device()) return SRLA_APIRESULT_NG;
    return im->encode_stream(nullptr, d_input, channel_stride, num_samples, data, data_size, output_size,
                             encode_callback, true, im->search_enabled());
}

void SRLAMI355X_SetPackThreads(struct SRLAEncoder *encoder, uint32_t num_threads)
{
    Impl *im = impl_of(encoder);
    if (!im) return;
    im->pack_threads = num_threads;
    if (im->pool) { delete im->pool; im->pool = new Pool(num_threads ? num_threads : 1); }
}

void SRLAMI355X_GetStats(struct SRLAEncoder *encoder, struct SRLAMI355XStats *stats, int reset)
{
    Impl *im = impl_of(encoder);
    if (!im) return;
    if (stats) *stats = im->stats;
    if (reset) memset(&im->stats, 0, sizeof(im->stats));
}

SRLAApiResult SRLAMI355X_ProbeBlock(struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
                                    void *records, int32_t *residuals, double *debug)
{
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || num_samples == 0) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples > im->par.max_num_samples_per_block) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (num_samples <= im->preset_order()) return SRLA_APIRESULT_INVALID_ARGUMENT; /* RAW by length: nothing to analyse */
    if (!im->init_device()) return SRLA_APIRESULT_NG;
    Slot &s = im->slot[0];
    im->build_job(s.job, 0, num_samples, false);
    s.out_direct = nullptr; s.out_first = 1; s.out_init_pos = 0; s.out_limit = 0xFFFFFFFFu; s.timed = im->timing; s.out_boost = 1;
    if (!im->launch_job(s, nullptr, 0, input, true) || !im->wait_job(s)) return SRLA_APIRESULT_NG;
    const uint32_t nv = im->num_variants();
    if (records && hipMemcpy(records, s.d_results.p, (size_t)nv * sizeof(SrlaItemResult), hipMemcpyDeviceToHost) != hipSuccess)
        return SRLA_APIRESULT_NG;
    if (debug && hipMemcpy(debug, s.d_dbg.p, (size_t)nv * SRLA_DBG_STRIDE * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return SRLA_APIRESULT_NG;
    if (residuals) {
        for (uint32_t v = 0; v < nv; v++)
            if (hipMemcpy(residuals + (size_t)v * num_samples, s.d_res_ws.as<int32_t>() + s.job.items[v].res_off,
                          (size_t)num_samples * 4, hipMemcpyDeviceToHost) != hipSuccess) return SRLA_APIRESULT_NG;
    }
    return SRLA_APIRESULT_OK;
}

}  /* extern "C" */
