/*
 * host_support.cpp -- the staging copies of host_support.h.
 */
#include "host_support.h"

#include <algorithm>
#include <map>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace srla {

/* ---- staging copy: pageable planes -> pinned buffer, with the OR of the samples as a by-product ---------
 * The pinned buffer is only read by the DMA engine afterwards, so the stores bypass the cache (no read-for-ownership
 * traffic): about 1.5x the throughput of memcpy for this pattern. */
#if defined(__x86_64__)
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

uint32_t pack16_or(int16_t *dst, const int32_t *src, size_t n, uint32_t *wide)
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

uint32_t copy_or(int32_t *dst, const int32_t *src, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return copy_or_avx2(dst, src, n);
#endif
    uint32_t m = 0;
    for (size_t k = 0; k < n; k++) { const int32_t x = src[k]; dst[k] = x; m |= (uint32_t)x; }
    return m;
}

uint32_t or_reduce(const int32_t *src, size_t n)
{
    uint32_t m = 0;
    for (size_t k = 0; k < n; k++) m |= (uint32_t)src[k];
    return m;
}


uint32_t pcm_channel(const uint8_t *frames, uint32_t B, uint32_t nch, uint32_t ch, size_t first, size_t n, int32_t *dst)
{
    const size_t frame = (size_t)B * nch;
    const uint8_t *p = frames + first * frame + (size_t)ch * B;
    uint32_t m = 0;
    for (size_t i = 0; i < n; i++, p += frame) {
        int32_t v;
        if (B == 1) v = (int32_t)p[0] - 128;
        else if (B == 2) v = (int16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
        else if (B == 3) v = ((int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24))) >> 8;
        else v = (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
        if (dst) dst[i] = v;
        m |= (uint32_t)v;
    }
    return m;
}

namespace {
struct PinEntry { size_t bytes; uint32_t refs; };
std::mutex g_pin_mutex;
std::map<const void *, PinEntry> g_pins;
}

bool host_pin_acquire(const void *p, size_t bytes, double *us_per_mb)
{
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    auto it = g_pins.find(p);
    if (it != g_pins.end()) {
        if (it->second.bytes < bytes) return false;
        it->second.refs++;
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (us_per_mb) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        *us_per_mb = us / std::max(1.0, (double)bytes / 1048576.0);
    }
    g_pins.emplace(p, PinEntry{ bytes, 1u });
    return true;
}

const void *host_pin_addref(const void *p)
{
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    auto it = g_pins.upper_bound(p);
    if (it == g_pins.begin()) return nullptr;
    --it;
    const char *base = static_cast<const char *>(it->first);
    if (static_cast<const char *>(p) >= base + it->second.bytes) return nullptr;
    it->second.refs++;
    return it->first;
}

void host_pin_release(const void *p)
{
    std::lock_guard<std::mutex> lock(g_pin_mutex);
    auto it = g_pins.find(p);
    if (it == g_pins.end()) return;
    if (--it->second.refs == 0) {
        if (hipHostUnregister(const_cast<void *>(p)) != hipSuccess) (void)hipGetLastError();
        g_pins.erase(it);
    }
}

}  // namespace srla
