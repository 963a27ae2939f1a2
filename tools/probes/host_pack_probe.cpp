/* host packing loop of stage_input (host_support.cpp: pack16_or), variants: software prefetch distance, AVX-512 narrowing.
 * hipcc -O2 -std=c++17 -pthread tools/probes/host_pack_probe.cpp -o /tmp/host_pack_probe && /tmp/host_pack_probe THREADS [dst: malloc|pinned] [src: default|nohuge|huge] */
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <vector>
#include <cstring>
#include <sys/mman.h>
template <int PF, bool AVX512>
__attribute__((target("avx2,avx512f,avx512bw"))) static uint32_t pack(int16_t *dst, const int32_t *src, size_t n, uint32_t *wide)
{
    uint32_t m = 0, w = 0; size_t k = 0;
    if (AVX512) {
        __m512i acc = _mm512_setzero_si512(), accw = _mm512_setzero_si512();
        const __m512i bias = _mm512_set1_epi32(32768);
        for (; k + 64 <= n; k += 64) {
            if (PF) { _mm_prefetch((const char *)(src + k) + PF, _MM_HINT_T0); _mm_prefetch((const char *)(src + k) + PF + 64, _MM_HINT_T0); _mm_prefetch((const char *)(src + k) + PF + 128, _MM_HINT_T0); _mm_prefetch((const char *)(src + k) + PF + 192, _MM_HINT_T0); }
            const __m512i a = _mm512_loadu_si512(src + k), b = _mm512_loadu_si512(src + k + 16), c = _mm512_loadu_si512(src + k + 32), d = _mm512_loadu_si512(src + k + 48);
            _mm256_stream_si256((__m256i *)(dst + k), _mm512_cvtepi32_epi16(a));
            _mm256_stream_si256((__m256i *)(dst + k + 16), _mm512_cvtepi32_epi16(b));
            _mm256_stream_si256((__m256i *)(dst + k + 32), _mm512_cvtepi32_epi16(c));
            _mm256_stream_si256((__m256i *)(dst + k + 48), _mm512_cvtepi32_epi16(d));
            acc = _mm512_or_si512(acc, _mm512_or_si512(_mm512_or_si512(a, b), _mm512_or_si512(c, d)));
            accw = _mm512_or_si512(accw, _mm512_or_si512(_mm512_or_si512(_mm512_add_epi32(a, bias), _mm512_add_epi32(b, bias)), _mm512_or_si512(_mm512_add_epi32(c, bias), _mm512_add_epi32(d, bias))));
        }
        m = _mm512_reduce_or_epi32(acc); w = _mm512_reduce_or_epi32(accw) & 0xFFFF0000u;
    } else {
    __m256i acc = _mm256_setzero_si256(), accw = _mm256_setzero_si256();
    const __m256i bias = _mm256_set1_epi32(32768);
    for (; k + 32 <= n; k += 32) {
        if (PF) { _mm_prefetch((const char *)(src + k) + PF, _MM_HINT_T0); _mm_prefetch((const char *)(src + k) + PF + 64, _MM_HINT_T0); }
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + k));
        const __m256i b = _mm256_loadu_si256((const __m256i *)(src + k + 8));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + k + 16));
        const __m256i d = _mm256_loadu_si256((const __m256i *)(src + k + 24));
        _mm256_stream_si256((__m256i *)(dst + k), _mm256_permute4x64_epi64(_mm256_packs_epi32(a, b), 0xD8));
        _mm256_stream_si256((__m256i *)(dst + k + 16), _mm256_permute4x64_epi64(_mm256_packs_epi32(c, d), 0xD8));
        acc = _mm256_or_si256(acc, _mm256_or_si256(_mm256_or_si256(a, b), _mm256_or_si256(c, d)));
        accw = _mm256_or_si256(accw, _mm256_or_si256(_mm256_or_si256(_mm256_add_epi32(a, bias), _mm256_add_epi32(b, bias)), _mm256_or_si256(_mm256_add_epi32(c, bias), _mm256_add_epi32(d, bias))));
    }
    alignas(32) uint32_t lanes[8];
    _mm256_store_si256((__m256i *)lanes, acc); for (int i = 0; i < 8; i++) m |= lanes[i];
    _mm256_store_si256((__m256i *)lanes, accw); for (int i = 0; i < 8; i++) w |= lanes[i] & 0xFFFF0000u;
    }
    _mm_sfence(); *wide = w; return m;
}
typedef uint32_t (*Fn)(int16_t *, const int32_t *, size_t, uint32_t *);
int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 1;
    const size_t N = 57600000;  /* 600 s stereo */
    const char *dmode = argc > 2 ? argv[2] : "malloc", *smode = argc > 3 ? argv[3] : "default";
    int32_t *src = (int32_t *)aligned_alloc(2u << 20, (N * 4 + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1));
    if (!strcmp(smode, "nohuge")) madvise(src, N * 4, MADV_NOHUGEPAGE);
    if (!strcmp(smode, "huge")) madvise(src, N * 4, MADV_HUGEPAGE);
    int16_t *dst = nullptr;
    if (!strcmp(dmode, "pinned")) { if (hipHostMalloc((void **)&dst, N * 2, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; } }
    else dst = (int16_t *)aligned_alloc(4096, N * 2);
    printf("dst %s, src %s\n", dmode, smode);
    for (size_t i = 0; i < N; i++) src[i] = (int32_t)((i * 2654435761u) >> 17) - 16384;
    memset(dst, 0, N * 2);
    struct { const char *name; Fn f; } v[] = { {"avx2", pack<0, false>}, {"avx2+pf4k", pack<4096, false>}, {"avx512", pack<0, true>}, {"avx512+pf4k", pack<4096, true>} };
    for (int round = 0; round < 2; round++)
    for (auto &x : v) {
        double best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            const size_t chunk = 256u << 10; const size_t ntask = N / chunk;
            std::atomic<size_t> next{0};
            for (int t = 0; t < T; t++) th.emplace_back([&] { uint32_t w; for (;;) { size_t i = next.fetch_add(1); if (i >= ntask) break; x.f(dst + i * chunk, src + i * chunk, chunk, &w); } });
            for (auto &t : th) t.join();
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt < best) best = dt;
        }
        printf("%-14s T=%d  %.2f ms  read %.1f GB/s\n", x.name, T, best * 1e3, N * 4 / best / 1e9);
    }
}
