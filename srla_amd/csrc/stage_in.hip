/*
 * stage_in.hip -- what brings a job's samples into shape on the device: srla_widen16 (int16 staging -> int32 planes),
 * srla_deinterleave (WAV data chunks -> planes), srla_or_reduce / srla_mask_to_shift (the offset left shift of a device-resident
 * stream), srla_chain_commit (history mode: the reference's persistent buffer); and the launchers' tuning record.  DESIGN.md 3.6, 4.
 */
#include "kernels_common.h"

SrlaLaunchTuning g_srla_tune = {};
extern "C" void srla_set_launch_tuning(const SrlaLaunchTuning *t) { if (t) g_srla_tune = *t; }

/* ------------------------------------------------------------------- offset left shift ---- */
/* OR of every sample of every channel (srla_utility.c:177-203); grid.y = channel */
__global__ __launch_bounds__(NT) void srla_or_reduce(const int32_t *__restrict__ in, size_t channel_stride, size_t count,
                                                     uint32_t *__restrict__ out)
{
    /* every workgroup streams ONE contiguous slice of the channel (16 KB per step, four 16-byte loads in flight
     * per thread): contiguous slices keep DRAM pages and the TLB busy with useful data */
    const int32_t *p = in + (size_t)blockIdx.y * channel_stride;
    const size_t per = (((count + gridDim.x - 1) / gridDim.x) + (NT * 16 - 1)) / (NT * 16) * (NT * 16);
    const size_t lo = (size_t)blockIdx.x * per;
    size_t hi = lo + per;
    if (hi > count) hi = count;
    uint32_t m = 0;
    if (lo < hi) {
        const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
        size_t i = lo + (size_t)threadIdx.x * 4;
        if (aligned) {
            for (; i + 3 * NT * 4 + 4 <= hi; i += 4 * NT * 4) {
                const int4 a = *reinterpret_cast<const int4 *>(p + i);
                const int4 b = *reinterpret_cast<const int4 *>(p + i + NT * 4);
                const int4 c = *reinterpret_cast<const int4 *>(p + i + 2 * NT * 4);
                const int4 d = *reinterpret_cast<const int4 *>(p + i + 3 * NT * 4);
                m |= (uint32_t)(a.x | a.y | a.z | a.w) | (uint32_t)(b.x | b.y | b.z | b.w) | (uint32_t)(c.x | c.y | c.z | c.w) | (uint32_t)(d.x | d.y | d.z | d.w);
            }
        }
        for (; i < hi; i += NT * 4)
            for (size_t k = i; k < hi && k < i + 4; k++) m |= (uint32_t)p[k];
    }
    /* one atomic per workgroup, and only if it would add bits: same-address atomics serialise in L2 (32 K of them
     * cost more than streaming the 230 MB) */
    __shared__ uint32_t s_m[NWAVES];
    for (int off = 32; off > 0; off >>= 1) m |= __shfl_down(m, off, WAVE);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t all = 0;
        for (int w = 0; w < NWAVES; w++) all |= s_m[w];
        const uint32_t seen = __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (all & ~seen) atomicOr(out, all);
    }
}

/* out[1] = trailing zero count of out[0] (0 when the stream is all zero) */
__global__ void srla_mask_to_shift(uint32_t *__restrict__ out)
{
    const uint32_t mask = out[0];
    out[1] = mask ? (uint32_t)(__ffs((int)mask) - 1) : 0u;
}

/* ---------------------------------------------------------------------- history mode ---- */
/* The reference's persistent FFT buffer (lpc.c:58,211) as it stands after a phase of calls, kept in the first `top` words
 * of the chain pool: word i becomes what the LAST call of the phase whose transform was longer than i left there (every call
 * left its complete buffer at its own place in the pool); words no call of the phase reached keep what they held.
 * The host says which call owns which words (segments). */
struct SrlaCommitTable { uint32_t lo[SRLA_COMMIT_SEGS], hi[SRLA_COMMIT_SEGS], src[SRLA_COMMIT_SEGS], n; };
__global__ __launch_bounds__(256) void srla_chain_commit(double *__restrict__ pool, SrlaCommitTable tab, uint32_t top)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= top) return;
    for (uint32_t k = 0; k < tab.n; k++)
        if (i >= tab.lo[k] && i < tab.hi[k]) { pool[i] = pool[(size_t)tab.src[k] + i]; return; }
}

extern "C" int srla_launch_chain_commit(hipStream_t stream, double *pool, const uint32_t *lo, const uint32_t *hi, const uint32_t *src, uint32_t nseg)
{
    SrlaCommitTable tab;
    uint32_t top = 0;
    if (nseg > SRLA_COMMIT_SEGS) return -1;
    for (uint32_t k = 0; k < nseg; k++) { tab.lo[k] = lo[k]; tab.hi[k] = hi[k]; tab.src[k] = src[k]; top = (hi[k] > top) ? hi[k] : top; }
    tab.n = nseg;
    if (top == 0) return 0;
    hipLaunchKernelGGL(srla_chain_commit, dim3((top + 255u) / 256u), dim3(256), 0, stream, pool, tab, top);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ __launch_bounds__(NT) void srla_widen16(const int16_t *__restrict__ src, size_t stride16, int32_t *__restrict__ dst, uint32_t n)
{
    /* four samples per thread: one 8-byte load, four consecutive stores (the int32 planes need not be 16-byte aligned) */
    const uint32_t ch = blockIdx.y;
    const size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 4u;
    if (i >= n) return;
    const short4 v = *reinterpret_cast<const short4 *>(src + (size_t)ch * stride16 + i);   /* planes are padded to 16 samples */
    int32_t *d = dst + (size_t)ch * n + i;
    d[0] = v.x;
    if (i + 1 < n) d[1] = v.y;
    if (i + 2 < n) d[2] = v.z;
    if (i + 3 < n) d[3] = v.w;
}

extern "C" int srla_launch_widen16(hipStream_t stream, const int16_t *src, size_t stride16, int32_t *dst, uint32_t n, uint32_t num_channels)
{
    if (n == 0 || num_channels == 0) return 0;
    const uint32_t per = NT * 4u;
    hipLaunchKernelGGL(srla_widen16, dim3((n + per - 1) / per, num_channels), dim3(NT), 0, stream, src, stride16, dst, n);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

/* Interleaved little-endian PCM frames (what a WAV data chunk holds, uploaded as they are) -> planar int32: the conversion of
 * libs/wav/src/wav.c (8-bit samples are unsigned with offset 128, the others signed), four frames per thread, one
 * 16-byte store per channel.  dst points at the first sample of the segment in plane 0 (16-byte aligned), planes `stride` apart. */
template <int B>
__global__ __launch_bounds__(NT) void srla_deinterleave(const uint8_t *__restrict__ src, uint32_t num_channels, uint32_t count,
                                                        int32_t *__restrict__ dst, size_t stride)
{
    const uint32_t f0 = 4u * (blockIdx.x * NT + threadIdx.x);
    if (f0 >= count) return;
    const uint32_t frame = (uint32_t)B * num_channels;
    for (uint32_t ch = 0; ch < num_channels; ch++) {
        int32_t v[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (f0 + (uint32_t)i >= count) break;
            const uint8_t *p = src + (size_t)(f0 + (uint32_t)i) * frame + (size_t)ch * B;
            if (B == 1) v[i] = (int32_t)p[0] - 128;
            else if (B == 2) v[i] = (int16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
            else if (B == 3) v[i] = ((int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24))) >> 8;
            else v[i] = (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
        }
        int32_t *o = dst + (size_t)ch * stride + f0;
        if (f0 + 4u <= count) *reinterpret_cast<int4 *>(o) = make_int4(v[0], v[1], v[2], v[3]);
        else for (uint32_t i = 0; f0 + i < count; i++) o[i] = v[i];
    }
}

extern "C" int srla_launch_deinterleave(hipStream_t stream, const void *src, uint32_t bytes_per_sample, uint32_t num_channels,
                                        uint32_t count, int32_t *dst, size_t stride)
{
    if (count == 0) return 0;
    const dim3 grid((count + 4u * NT - 1u) / (4u * NT)), blk(NT);
    const uint8_t *s8 = (const uint8_t *)src;
    switch (bytes_per_sample) {
    case 1: hipLaunchKernelGGL(srla_deinterleave<1>, grid, blk, 0, stream, s8, num_channels, count, dst, stride); break;
    case 2: hipLaunchKernelGGL(srla_deinterleave<2>, grid, blk, 0, stream, s8, num_channels, count, dst, stride); break;
    case 3: hipLaunchKernelGGL(srla_deinterleave<3>, grid, blk, 0, stream, s8, num_channels, count, dst, stride); break;
    case 4: hipLaunchKernelGGL(srla_deinterleave<4>, grid, blk, 0, stream, s8, num_channels, count, dst, stride); break;
    default: return -1;
    }
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

/* the same without the conversion: *out |= OR of the samples (several launches accumulate into one word) */
extern "C" int srla_launch_or_accumulate(hipStream_t stream, const int32_t *in, size_t channel_stride, size_t count,
                                         uint32_t num_channels, uint32_t *out)
{
    size_t blocks = (count + NT * 64 - 1) / (NT * 64);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(srla_or_reduce, dim3((uint32_t)blocks, num_channels), dim3(NT), 0, stream, in, channel_stride, count, out);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" int srla_launch_or_reduce(hipStream_t stream, const int32_t *in, size_t channel_stride, size_t count,
                                     uint32_t num_channels, uint32_t *out)
{
    /* out[0] must be zero on entry; on completion out[0] = OR mask, out[1] = offset left shift */
    size_t blocks = (count + NT * 64 - 1) / (NT * 64);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(srla_or_reduce, dim3((uint32_t)blocks, num_channels), dim3(NT), 0, stream, in, channel_stride, count, out);
    hipLaunchKernelGGL(srla_mask_to_shift, dim3(1), dim3(1), 0, stream, out);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}
