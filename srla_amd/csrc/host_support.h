/*
 * host_support.h -- small host-side building blocks of the runtime: HIP error macro, growable device / pinned
 * buffers, the staging copies (pageable planes -> pinned memory, with the OR of the samples as a by-product) and
 * the persistent thread pool that runs them.
 */
#ifndef SRLA_HOST_SUPPORT_H
#define SRLA_HOST_SUPPORT_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <thread>
#include <vector>

namespace srla {

using Clock = std::chrono::steady_clock;
inline double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

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

/* Temporary page-locking of a caller's buffer for the duration of a call (hipHostRegister), process-wide and reference
 * counted: two handles working on the same buffer at the same time share one registration, and it is dropped when the
 * last of them is done.  acquire() fails (false) where registration does -- the caller then stages as usual. */
bool host_pin_acquire(const void *p, size_t bytes, double *us_per_mb);
/* A pointer that HIP reports as page-locked host memory may lie in a range THIS registry locked for another handle's call:
 * takes a reference on that registration (so it outlives the other call) and returns its key for host_pin_release; nullptr
 * when the memory is not the registry's (the caller's own hipHostMalloc / hipHostRegister: theirs to keep alive). */
const void *host_pin_addref(const void *p);
void host_pin_release(const void *p);

/* staging copies (host_support.cpp): return the OR of the samples they move */
uint32_t copy_or(int32_t *dst, const int32_t *src, size_t n);
/* the same, packing to int16 (streams of at most 16 bits per sample cross PCIe at half the size; the device widens them
 * again); *wide gets a non-zero value if a sample does not fit (the caller then stages that job as int32) */
uint32_t pack16_or(int16_t *dst, const int32_t *src, size_t n, uint32_t *wide);
/* interleaved little-endian PCM frames -> one channel's samples (libs/wav/src/wav.c: 8-bit unsigned + 128, else signed); returns their OR */
uint32_t pcm_channel(const uint8_t *frames, uint32_t bytes_per_sample, uint32_t num_channels, uint32_t ch, size_t first, size_t n, int32_t *dst /* may be null */);
/* OR of n samples (no copy) */
uint32_t or_reduce(const int32_t *src, size_t n);

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
            fn_ = &fn; next_.store(0); count_ = count; pending_.store((unsigned)workers_.size(), std::memory_order_relaxed); gen_++;
            gen_atomic_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
        run_chunk();
        /* the stragglers are microseconds away: look before sleeping (a futex wake-up is 10-20 us) */
        for (int spin = 0; spin < 4000 && pending_.load(std::memory_order_acquire) != 0; spin++) cpu_pause();
        if (pending_.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> l(m_);
            done_cv_.wait(l, [this] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        fn_ = nullptr;
    }
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    /* How long a worker keeps looking for the next round before it sleeps.  Rounds of a long call arrive every few hundred
     * microseconds and a sleeping worker is fine for them (0: a few microseconds of looking); a call of one job has two rounds --
     * staging at its start, the copy-out at its end, 0.3 ms apart -- and a worker that has to be woken for the second one costs a
     * tenth of the call's time. */
    void set_linger_us(uint32_t us) { linger_us_.store(us, std::memory_order_relaxed); }

private:
    static void cpu_pause()
    {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
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
                /* look briefly before sleeping: pack rounds arrive every few hundred microseconds */
                for (int spin = 0; spin < 200 && gen_atomic_.load(std::memory_order_acquire) == seen && !stop_; spin++) cpu_pause();
                const uint32_t linger = linger_us_.load(std::memory_order_relaxed);
                if (linger != 0 && gen_atomic_.load(std::memory_order_acquire) == seen) {
                    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(linger);
                    while (gen_atomic_.load(std::memory_order_acquire) == seen && !stop_) {
                        for (int k = 0; k < 32; k++) cpu_pause();
                        if (std::chrono::steady_clock::now() >= until) break;
                    }
                }
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            run_chunk();
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> l(m_);
                done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    std::atomic<bool> stop_;     /* read inside the unlocked spin loops */
    const std::function<void(uint32_t)> *fn_ = nullptr;
    std::atomic<uint32_t> next_{ 0 };
    uint32_t count_ = 0;
    std::atomic<unsigned> pending_;
    std::atomic<uint32_t> linger_us_{ 0 };
    uint64_t gen_ = 0;
    std::atomic<uint64_t> gen_atomic_{ 0 };
};

}  // namespace srla
#endif
