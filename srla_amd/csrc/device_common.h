/*
 * device_common.h -- small device helpers shared by the kernel files (kernels.hip, autocorr_wave.hip): complex arithmetic with
 * the reference's roundings, DPP wave reductions, the XCD-aware item mapping and the sample loaders.
 * Every file that includes this is compiled with -ffp-contract=off and pins it again below.
 */
#ifndef SRLA_DEVICE_COMMON_H
#define SRLA_DEVICE_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"

#pragma clang fp contract(off)

#define NT 256
#define WAVE 64
#define NWAVES (NT / WAVE)
#define FIR_PAD 256   /* ints of zero padding in front of the signal in LDS (>= max order rounded to 4) */

typedef double2 cplx;

/* ---------------------------------------------------------------- small device helpers --- */
__device__ __forceinline__ uint32_t zigzag32(int32_t s) { return ((uint32_t)s << 1) ^ (uint32_t)(-(int32_t)(s < 0)); }

__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx c_mul(cplx a, cplx b)
{
    /* fft.c:57-63: two roundings per product term, no fused multiply-add */
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

/* (int32_t)d as the reference's build computes it: cvttsd2si yields the "integer indefinite" 0x80000000 for every value outside
 * int32 and for NaN, where the device's conversion saturates (INT_MAX for large positive values).  The reference converts unbounded
 * doubles in one place that matters -- the long-term predictor's taps, (int32_t)Round(coef * 32) before the clip to [-32, 31]
 * (srla_encoder.c:1031-1037): a block whose lag 0 is rounding noise (an all-zero variant of an odd-length block, whose only input is
 * the word the Welch window leaves untouched, lpc.c:260-264) gets taps of 1e30 and more, and the reference's stream then carries -32
 * for the POSITIVE ones too. */
__device__ __forceinline__ int32_t cvt_i32_as_x86(double d)
{
    return (d >= -2147483648.0 && d < 2147483648.0) ? (int32_t)d : (int32_t)0x80000000;
}

__device__ __forceinline__ double round_half_away(double d)
{
    /* srla_utility.c:22-25 */
    return (d >= 0.0) ? floor(d + 0.5) : -floor(-d + 0.5);
}

/* Wave-wide reductions on the DPP path (no LDS traffic): inclusive scan inside each row of 16 lanes
 * (row_shr 1,2,4,8), then row 0 -> 1 and 2 -> 3 (row_bcast15), then lane 31 -> rows 2,3 (row_bcast31);
 * lane 63 ends up with the total, which is returned in every lane.  All 64 lanes must be active. */
#define SRLA_DPP_STEP(T, v, OP, IDENT, CTRL, ROWMASK) \
    v = OP(v, (T)__builtin_amdgcn_update_dpp((int)(IDENT), (int)(v), CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ uint32_t u32_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t u32_max(uint32_t a, uint32_t b) { return (a > b) ? a : b; }
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    SRLA_DPP_STEP(uint32_t, v, u32_add, 0, 0x111, 0xf); SRLA_DPP_STEP(uint32_t, v, u32_add, 0, 0x112, 0xf);
    SRLA_DPP_STEP(uint32_t, v, u32_add, 0, 0x114, 0xf); SRLA_DPP_STEP(uint32_t, v, u32_add, 0, 0x118, 0xf);
    SRLA_DPP_STEP(uint32_t, v, u32_add, 0, 0x142, 0xa); SRLA_DPP_STEP(uint32_t, v, u32_add, 0, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    SRLA_DPP_STEP(uint32_t, v, u32_max, 0, 0x111, 0xf); SRLA_DPP_STEP(uint32_t, v, u32_max, 0, 0x112, 0xf);
    SRLA_DPP_STEP(uint32_t, v, u32_max, 0, 0x114, 0xf); SRLA_DPP_STEP(uint32_t, v, u32_max, 0, 0x118, 0xf);
    SRLA_DPP_STEP(uint32_t, v, u32_max, 0, 0x142, 0xa); SRLA_DPP_STEP(uint32_t, v, u32_max, 0, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#define SRLA_DPP64(CTRL, ROWMASK)                                                                                  \
    {                                                                                                              \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWMASK, 0xf, false);  \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((unsigned long long)v >> 32), CTRL, ROWMASK, 0xf, false); \
        v += (long long)(((unsigned long long)hi << 32) | lo);                                                     \
    }
    SRLA_DPP64(0x111, 0xf) SRLA_DPP64(0x112, 0xf) SRLA_DPP64(0x114, 0xf) SRLA_DPP64(0x118, 0xf) SRLA_DPP64(0x142, 0xa) SRLA_DPP64(0x143, 0xc)
#undef SRLA_DPP64
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)v >> 32), 63);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ uint32_t xcd_position(uint32_t b, uint32_t count)
{
    const uint32_t chunk = (count + 7u) >> 3;
    return (b & 7u) * chunk + (b >> 3);
}

/* what the sample loaders need from the job parameters (kept in registers; the by-value kernel argument is never modified) */
struct InputView {
    uint32_t nch, stride, sh;
};
__device__ __forceinline__ InputView input_view(const SrlaJobParams &jp, uint32_t item_lshift)
{
    InputView v;
    v.nch = jp.num_channels;
    v.stride = jp.channel_stride;
    v.sh = jp.lshift_dev ? *jp.lshift_dev : item_lshift;
    return v;
}
__device__ __forceinline__ InputView input_view(const SrlaJobParams &jp, uint32_t item_lshift, const int32_t *) { return input_view(jp, item_lshift); }

/* variant sample i of the job input (srla_encoder.c:1229-1253, srla_utility.c:91-103) */
__device__ __forceinline__ int32_t load_variant(const int32_t *__restrict__ in, const InputView &jp,
                                                uint32_t variant, uint32_t idx)
{
    const uint32_t sh = jp.sh;
    if (variant < jp.nch) return in[(size_t)variant * jp.stride + idx] >> sh;
    const int32_t l = in[idx] >> sh;
    const int32_t r = in[(size_t)jp.stride + idx] >> sh;
    const int32_t s = (int32_t)((uint32_t)r - (uint32_t)l);
    if (variant == jp.nch + 1) return s;
    return (int32_t)((uint32_t)l + (uint32_t)(s >> 1));
}

/* four consecutive variant samples starting at i4 (zeros beyond n); 16-byte loads when possible */
__device__ __forceinline__ void load_chunk(const int32_t *__restrict__ in, const InputView &jp, uint32_t variant,
                                           uint32_t i4, uint32_t n, bool aligned, int32_t out[4])
{
    if (aligned && i4 + 4 <= n) {
        const uint32_t sh = jp.sh;
        if (variant < jp.nch) {
            const int4 a = *reinterpret_cast<const int4 *>(in + (size_t)variant * jp.stride + i4);
            out[0] = a.x >> sh; out[1] = a.y >> sh; out[2] = a.z >> sh; out[3] = a.w >> sh;
        } else {
            const int4 a = *reinterpret_cast<const int4 *>(in + i4);
            const int4 b = *reinterpret_cast<const int4 *>(in + (size_t)jp.stride + i4);
            const int32_t l[4] = { a.x >> sh, a.y >> sh, a.z >> sh, a.w >> sh };
            const int32_t r[4] = { b.x >> sh, b.y >> sh, b.z >> sh, b.w >> sh };
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int32_t s = (int32_t)((uint32_t)r[i] - (uint32_t)l[i]);
                out[i] = (variant == jp.nch + 1) ? s : (int32_t)((uint32_t)l[i] + (uint32_t)(s >> 1));
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = (i4 + i < n) ? load_variant(in, jp, variant, i4 + i) : 0;
    }
}

__device__ __forceinline__ bool input_aligned(const int32_t *in, const InputView &jp)
{
    return ((reinterpret_cast<uintptr_t>(in) & 15u) == 0) && ((jp.stride & 3u) == 0);
}

#endif /* SRLA_DEVICE_COMMON_H */
