/*
 * autocorr_wave.hip -- srla_autocorr_w<T>: the autocorrelation of one item (lpc.c:330-376 with everything in front of it:
 * pre-emphasis tap and filter srla_utility.c:214-254,342, long-term predictor srla_lpc_predict.c:267-294, Welch window
 * lpc.c:256-266) by T = nfft / 32 lanes, for real FFTs of 1024, 2048, 4096 and 8192 points (T = 32, 64, 128, 256: two items
 * per wavefront, one, one item per two / four wavefronts).
 *
 * The complex transform of m = 16 T points (fft.c:71-136) lives in REGISTERS, 16 complex values per lane, and is done in
 * place: the same radix-4 / radix-2 butterflies on the same operands as the reference's Stockham transform -- hence the same
 * bits (tools/fft_schedule_model.py checks the schedule in exact arithmetic on the CPU) -- but grouped so that two stages run
 * back to back on values a lane already holds:
 *
 *   pass 1   stages 1-2.  A lane holds one residue class modulo n2 = m / 16: slot = class + n2 j, j < 16.
 *   T1       through LDS, in slot order: after two in-place stages the transform has split into 16 contiguous blocks of n2.
 *   pass 2   stages 3-4.  A lane holds the "unit" (B, c): slots B n2 + 4 C a + C b + c, a, b < 4 (C = m / 256).
 *   T2       through LDS.
 *   pass 3   what is left (C points per unit: radix 2, 4, 4 then 2, 4 then 4).  A lane holds the units v = t + T g of C slots.
 *   T3       through LDS: X[k] stands at the digit-reversed slot; a lane takes the bins i of its class and their partners m - i
 *            and does the symmetry pass of the real transform (fft.c:164-183), |X|^2 (lpc.c:357-365) and the symmetry pass of
 *            the inverse for its own bins.
 *   inverse  passes 1-3 again (T4, T5 in between), pruned to the outputs that are read: lag i is component i & 1 of output i / 2.
 *
 * Five trips through LDS per item instead of one per stage and direction (about 25), and the butterflies of the in-register
 * stages need no address arithmetic.  Items of at most 64 lanes need no barrier at all: a wavefront's LDS operations execute
 * in order, so its lanes exchange data through their LDS region with nothing but the instruction order between them.
 * One item's LDS region is m complex slots (16 m bytes = 256 bytes per lane).
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "device_layout.h"
#include "kernels.h"

#pragma clang fp contract(off)

#include "device_common.h"

#ifndef SRLA_W_WAVES
#define SRLA_W_WAVES 2
#endif

namespace {

/* The lanes of an item talk through LDS without a barrier: the hardware keeps one wavefront's DS instructions in order; the
 * compiler is told not to move LDS accesses across this point. */
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ __forceinline__ int32_t lane_read(int32_t v, uint32_t src_lane) { return __builtin_amdgcn_ds_bpermute((int)(src_lane << 2), v); }

/* base-4 digit reversal of an 8-bit unit index */
__device__ __forceinline__ uint32_t rev4(uint32_t v) { return ((v >> 6) & 3u) | (((v >> 4) & 3u) << 2) | (((v >> 2) & 3u) << 4) | ((v & 3u) << 6); }
/* output digit of position c inside a pass-3 unit of C points */
template <int C> __device__ __forceinline__ constexpr uint32_t wrev(uint32_t c) { return (C <= 4) ? c : ((C == 8) ? ((c >> 1) + 4u * (c & 1u)) : ((c >> 2) + 4u * (c & 3u))); }

/* radix-4 butterfly in place (fft.c:98-110): a, b, c, d = inputs p, p + n/4, p + n/2, p + 3n/4 of a sub-transform; the
 * outputs of branch k replace input k. */
template <int FLAG>
__device__ __forceinline__ void bf4(cplx &a, cplx &b, cplx &c, cplx &d, const cplx w1, const cplx w2, const cplx w3)
{
    const cplx apc = c_add(a, c), amc = c_sub(a, c), bpd = c_add(b, d), bmd = c_sub(b, d);
    const cplx jbmd = (FLAG < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
    a = c_add(apc, bpd);
    b = c_mul(w1, c_sub(amc, jbmd));
    c = c_mul(w2, c_sub(apc, bpd));
    d = c_mul(w3, c_add(amc, jbmd));
}
/* the same with only the wanted branches formed (pruned inverse) */
template <int FLAG>
__device__ __forceinline__ void bf4_pruned(cplx &a, cplx &b, cplx &c, cplx &d, const cplx w1, const cplx w2, const cplx w3,
                                           const bool k1, const bool k2, const bool k3)
{
    const cplx apc = c_add(a, c), amc = c_sub(a, c), bpd = c_add(b, d), bmd = c_sub(b, d);
    const cplx jbmd = (FLAG < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
    a = c_add(apc, bpd);
    if (k1) b = c_mul(w1, c_sub(amc, jbmd));
    if (k2) c = c_mul(w2, c_sub(apc, bpd));
    if (k3) d = c_mul(w3, c_add(amc, jbmd));
}

/* Twiddle tables of one direction (host_tables.cpp): per stage of sub-size n > 2 three runs of n / 4 entries (w, w^2, w^3). */
template <int M> struct TwOff {
    static constexpr uint32_t s1 = 0u;                       /* n = M      */
    static constexpr uint32_t s2 = s1 + 3u * (M / 4);        /* n = M / 4  */
    static constexpr uint32_t s3 = s2 + 3u * (M / 16);       /* n = M / 16 */
    static constexpr uint32_t s4 = s3 + 3u * (M / 64);       /* n = M / 64 */
    static constexpr uint32_t s5 = s4 + 3u * (M / 256);      /* n = M / 256 (> 2) */
    static constexpr uint32_t s6 = s5 + 3u * (M / 1024);     /* n = M / 1024 (> 2) */
    static constexpr uint32_t total()
    {
        uint32_t t = 0;
        for (uint32_t n = M; n > 2; n >>= 2) t += 3u * (n >> 2);
        return t;
    }
};

/* Where slot i of a transposition stands in the item's LDS region.  Every transposition rewrites the whole region, so each
 * may have its own permutation: chosen (tools/lds_conflicts.py) so that the 16-byte stores (served in groups of 8 consecutive
 * lanes over 32 banks) and loads (groups of 16 lanes over 64 banks) of writer and reader both spread over the banks. */
template <int T> __device__ __forceinline__ uint32_t sw1(uint32_t i) { return (T >= 256) ? i : (i ^ ((i >> 4) & 15u)); }
template <int T> __device__ __forceinline__ uint32_t sw2(uint32_t i) { return i ^ ((i >> 4) & ((T >= 256) ? 15u : 7u)); }
template <int T> __device__ __forceinline__ uint32_t sw3(uint32_t i) { return i ^ ((i >> 4) & 1u) ^ ((i >> 5) & 7u); }

/* T > 64: the lanes of an item span wavefronts, which meet at a workgroup barrier; otherwise instruction order is enough */
template <int T> __device__ __forceinline__ void item_sync() { if (T > 64) __syncthreads(); else WSYNC(); __builtin_amdgcn_sched_barrier(0); }
/* The kernel is a few very long straight-line stretches; left alone, the scheduler hoists every table load of a stretch to its
 * top and runs out of registers.  Nothing is moved across these marks. */
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

/* Table loads are issued one butterfly (group) ahead of their use, by hand: PIN makes a lane index opaque at a point of the
 * program, so the loads addressed through it cannot be hoisted further up (left alone, the scheduler moves every load of these
 * long straight-line stretches to their top and runs out of registers); the loads of the following group stand right behind
 * the pin of the current one, so they are in flight while it is computed. */
#define PIN(v) asm volatile("" : "+v"(v))
/* the same, tied behind the computation that produced `val` (volatile statements alone keep their order among themselves only:
 * nothing would stop all of them, and the loads behind them, from moving in front of the arithmetic) */
#define PIN_AFTER(v, val) asm volatile("" : "+v"(v), "+v"(val))
struct Tw3 { cplx w1, w2, w3; };
__device__ __forceinline__ Tw3 load_tw3(const cplx *__restrict__ tab, const uint32_t run, const uint32_t p)
{
    Tw3 w;
    w.w1 = tab[p]; w.w2 = tab[run + p]; w.w3 = tab[2u * run + p];
    return w;
}

/* passes 1-3 of the transform on x[16] (see the head of the file).  L: the item's LDS region (m slots).  t: the lane inside the
 * item.  cl: the lane's residue class of pass 1.  PRUNE: only outputs k < need are wanted.  On return x[C g + c] holds output
 * k = rev4(v) + 256 wrev(c) of unit v = t + T g. */
template <int T, int FLAG, bool PRUNE>
__device__ __forceinline__ void transform_passes(cplx (&x)[16], cplx *__restrict__ L, const cplx *__restrict__ tw, const uint32_t t,
                                                 const uint32_t cl, const uint32_t need)
{
    constexpr int M = 16 * T, N2 = T, C = T / 16;
    typedef TwOff<M> TO;
    /* ---- pass 1: x[j] <-> slot cl + N2 j */
    {
        uint32_t pc = cl;
        PIN(pc);
        Tw3 w = load_tw3(tw + TO::s1, M / 4, pc);
#pragma unroll
        for (int j0 = 0; j0 < 4; j0++) {
            uint32_t pn = cl;
            if (j0 == 0) PIN(pn); else PIN_AFTER(pn, x[j0 - 1].x);
            /* the next butterfly's entries (after the last one: those of stage 2) */
            const Tw3 wn = (j0 < 3) ? load_tw3(tw + TO::s1, M / 4, pn + (uint32_t)(N2 * (j0 + 1))) : load_tw3(tw + TO::s2, M / 16, pn);
            bf4<FLAG>(x[j0], x[j0 + 4], x[j0 + 8], x[j0 + 12], w.w1, w.w2, w.w3);
            w = wn;
        }
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) bf4<FLAG>(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3], w.w1, w.w2, w.w3);
    }
    /* ---- T1 */
    item_sync<T>();
#pragma unroll
    for (int j = 0; j < 16; j++) L[sw1<T>((uint32_t)(j * N2) + cl)] = x[j];
    item_sync<T>();
    /* ---- pass 2: unit u = t = B C + c; x[4 a + b] <-> slot B N2 + 4 C a + C b + c */
    const uint32_t B = t / (uint32_t)C, c = t % (uint32_t)C;
    const uint32_t P = (B >> 2) | ((B & 3u) << 2);                 /* the outputs of block B are those with k = P modulo 16 */
    const uint32_t ubase = B * (uint32_t)N2 + c;
    const bool ulive = !PRUNE || P < need;
    if (ulive) {
        uint32_t pc = c;
        PIN(pc);
        Tw3 w = load_tw3(tw + TO::s3, M / 64, pc);
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) x[4 * a + b] = L[sw1<T>(ubase + (uint32_t)(4 * C * a + C * b))];
        /* stage 3 (n = N2): butterflies p = C b + c on inputs a = 0..3 */
        const bool a1 = !PRUNE || P + 16u < need, a2 = !PRUNE || P + 32u < need, a3 = !PRUNE || P + 48u < need;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t pn = c;
            if (b == 0) PIN(pn); else PIN_AFTER(pn, x[b - 1].x);
            /* the next butterfly's entries (after the last one: those of stage 4, butterfly p = c of every sub-block) */
            const Tw3 wn = (b < 3) ? load_tw3(tw + TO::s3, M / 64, pn + (uint32_t)(C * (b + 1))) : load_tw3(tw + TO::s4, M / 256, pn);
            if (PRUNE) bf4_pruned<FLAG>(x[b], x[4 + b], x[8 + b], x[12 + b], w.w1, w.w2, w.w3, a1, a2, a3);
            else bf4<FLAG>(x[b], x[4 + b], x[8 + b], x[12 + b], w.w1, w.w2, w.w3);
            w = wn;
        }
        /* stage 4 (n = 4 C) inside sub-block a: butterfly p = c on inputs b = 0..3 */
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const uint32_t Pa = P + 16u * (uint32_t)a;
            if (!PRUNE) bf4<FLAG>(x[4 * a], x[4 * a + 1], x[4 * a + 2], x[4 * a + 3], w.w1, w.w2, w.w3);
            else if (Pa < need)
                bf4_pruned<FLAG>(x[4 * a], x[4 * a + 1], x[4 * a + 2], x[4 * a + 3], w.w1, w.w2, w.w3, Pa + 64u < need, Pa + 128u < need, Pa + 192u < need);
        }
    }
    /* ---- T2 */
    item_sync<T>();
    if (ulive) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (!PRUNE || P + 16u * (uint32_t)a + 64u * (uint32_t)b < need) L[sw2<T>(ubase + (uint32_t)(4 * C * a + C * b))] = x[4 * a + b];
    }
    item_sync<T>();
    /* ---- pass 3: unit g: v = t + T g; x[C g + c] <-> slot C v + c.  Output k = rev4(v) + 256 wrev(c).  (The tables of these
     * last stages are the same for every lane: scalar loads.) */
#pragma unroll
    for (int g = 0; g < 16 / C; g++) {
        const uint32_t v = t + (uint32_t)(T * g);
        if (!PRUNE || rev4(v) < need) {
            cplx *r = &x[C * g];
#pragma unroll
            for (int cc = 0; cc < C; cc++) r[cc] = L[sw2<T>(v * (uint32_t)C + (uint32_t)cc)];
        }
    }
#pragma unroll
    for (int g = 0; g < 16 / C; g++) {
        const uint32_t v = t + (uint32_t)(T * g);
        if (!PRUNE || rev4(v) < need) {
            cplx *r = &x[C * g];
            /* with PRUNE every wanted output has k < 256, i.e. is position 0 of its unit: only branch 0 is formed */
            if (C == 2) {
                const cplx a = r[0], b = r[1];
                r[0] = c_add(a, b);
                if (!PRUNE) r[1] = c_sub(a, b);
            } else if (C == 4) {
                const cplx w1 = tw[TO::s5], w2 = tw[TO::s5 + 1], w3 = tw[TO::s5 + 2];
                if (PRUNE) bf4_pruned<FLAG>(r[0], r[1], r[2], r[3], w1, w2, w3, false, false, false);
                else bf4<FLAG>(r[0], r[1], r[2], r[3], w1, w2, w3);
            } else if (C == 8) {
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const cplx w1 = tw[TO::s5 + p], w2 = tw[TO::s5 + 2 + p], w3 = tw[TO::s5 + 4 + p];
                    if (PRUNE) bf4_pruned<FLAG>(r[p], r[p + 2], r[p + 4], r[p + 6], w1, w2, w3, false, false, false);
                    else bf4<FLAG>(r[p], r[p + 2], r[p + 4], r[p + 6], w1, w2, w3);
                }
#pragma unroll
                for (int k = 0; k < (PRUNE ? 1 : 4); k++) {
                    const cplx a = r[2 * k], b = r[2 * k + 1];
                    r[2 * k] = c_add(a, b);
                    if (!PRUNE) r[2 * k + 1] = c_sub(a, b);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const cplx w1 = tw[TO::s5 + p], w2 = tw[TO::s5 + 4 + p], w3 = tw[TO::s5 + 8 + p];
                    if (PRUNE) bf4_pruned<FLAG>(r[p], r[p + 4], r[p + 8], r[p + 12], w1, w2, w3, false, false, false);
                    else bf4<FLAG>(r[p], r[p + 4], r[p + 8], r[p + 12], w1, w2, w3);
                }
                const cplx w1 = tw[TO::s6], w2 = tw[TO::s6 + 1], w3 = tw[TO::s6 + 2];
#pragma unroll
                for (int k = 0; k < (PRUNE ? 1 : 4); k++) {
                    if (PRUNE) bf4_pruned<FLAG>(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3], w1, w2, w3, false, false, false);
                    else bf4<FLAG>(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3], w1, w2, w3);
                }
            }
        }
    }
}

/* One bin of the spectrum pass.  The pair (i, m - i), 1 <= i <= m / 2, with za = X[i], zb = X[m - i]: symmetry pass of the
 * forward real transform (fft.c:164-183, flag -1), power of both bins (lpc.c:357-365), symmetry pass of the inverse (flag +1)
 * -- the arithmetic of spectrum_power_pass (kernels.hip).  Returns what the pass leaves in bin i (SECOND = false) or in bin
 * m - i (SECOND = true).  self: i = m / 2 pairs with itself, and the reference's second pair of stores wins in both passes. */
template <bool SECOND>
__device__ __forceinline__ cplx spectrum_bin(const cplx za, const cplx zb, const cplx wf, const cplx wi, const bool self)
{
    double p1, p3;
    {
        const double c2 = -0.5;
        const double x1 = za.x, x2 = za.y, x3 = zb.x, x4 = zb.y;
        const double wr = wf.x, wim = wf.y;
        const double h1r = 0.5 * (x1 + x3);
        const double h1i = 0.5 * (x2 - x4);
        const double h2r = -c2 * (x2 + x4);
        const double h2i = c2 * (x1 - x3);
        const double y1 = h1r + (wr * h2r) - (wim * h2i);
        const double y2 = h1i + (wr * h2i) + (wim * h2r);
        const double y3 = h1r - (wr * h2r) + (wim * h2i);
        const double y4 = -h1i + (wr * h2i) + (wim * h2r);
        p1 = y1 * y1 + y2 * y2;
        p3 = y3 * y3 + y4 * y4;
        if (self) p1 = p3;
    }
    const double c2 = 0.5;
    const double x1 = p1, x2 = 0.0, x3 = p3, x4 = 0.0;
    const double wr = wi.x, wim = wi.y;
    const double h1r = 0.5 * (x1 + x3);
    const double h1i = 0.5 * (x2 - x4);
    const double h2r = -c2 * (x2 + x4);
    const double h2i = c2 * (x1 - x3);
    if (!SECOND) return make_double2(h1r + (wr * h2r) - (wim * h2i), h1i + (wr * h2i) + (wim * h2r));
    return make_double2(h1r - (wr * h2r) + (wim * h2i), -h1i + (wr * h2i) + (wim * h2r));
}

__device__ __forceinline__ long long group_sum_i64(long long v, const int T)
{
    for (int d = 1; d < T; d <<= 1) {
        const uint32_t src = (threadIdx.x & 63u) ^ (uint32_t)d;
        const uint32_t lo = (uint32_t)lane_read((int32_t)(uint32_t)v, src), hi = (uint32_t)lane_read((int32_t)(uint32_t)((unsigned long long)v >> 32), src);
        v += (long long)(((unsigned long long)hi << 32) | lo);
    }
    return v;
}
__device__ __forceinline__ uint32_t group_max_u32(uint32_t v, const int T)
{
    for (int d = 1; d < T; d <<= 1) {
        const uint32_t o = (uint32_t)lane_read((int32_t)v, (threadIdx.x & 63u) ^ (uint32_t)d);
        v = (o > v) ? o : v;
    }
    return v;
}

/* two consecutive variant samples starting at the even index i (zeros from n on) */
__device__ __forceinline__ void load_pair(const int32_t *__restrict__ in, const InputView &iv, const uint32_t variant, const uint32_t i,
                                          const uint32_t n, const bool aligned8, int32_t &s0, int32_t &s1)
{
    if (aligned8 && i + 2u <= n) {
        const uint32_t sh = iv.sh;
        if (variant < iv.nch) {
            const int2 a = *reinterpret_cast<const int2 *>(in + (size_t)variant * iv.stride + i);
            s0 = a.x >> sh; s1 = a.y >> sh;
        } else {
            const int2 a = *reinterpret_cast<const int2 *>(in + i);
            const int2 b = *reinterpret_cast<const int2 *>(in + (size_t)iv.stride + i);
            const int32_t l0 = a.x >> sh, l1 = a.y >> sh, r0 = b.x >> sh, r1 = b.y >> sh;
            const int32_t d0 = (int32_t)((uint32_t)r0 - (uint32_t)l0), d1 = (int32_t)((uint32_t)r1 - (uint32_t)l1);
            if (variant == iv.nch + 1u) { s0 = d0; s1 = d1; }
            else { s0 = (int32_t)((uint32_t)l0 + (uint32_t)(d0 >> 1)); s1 = (int32_t)((uint32_t)l1 + (uint32_t)(d1 >> 1)); }
        }
    } else {
        s0 = (i < n) ? load_variant(in, iv, variant, i) : 0;
        s1 = (i + 1u < n) ? load_variant(in, iv, variant, i + 1u) : 0;
    }
}

/* scratch for the reductions of items that span wavefronts (T > 64) */
struct WaveSums {
    long long r0[4], r1[4];
    uint32_t amax[4];
    int32_t coef;
    uint32_t pad[3];
};

}  // namespace

template <int T>
__global__ __launch_bounds__((T > 64) ? T : 64) __attribute__((amdgpu_waves_per_eu(SRLA_W_WAVES, SRLA_W_WAVES))) void srla_autocorr_w(
    SrlaJobParams jp, const int32_t *__restrict__ input, const cplx *__restrict__ twiddles, uint32_t pass,
    SrlaItemResult *__restrict__ results, double *__restrict__ lags_ws, double *__restrict__ dbg,
    const SrlaAutocorrItem *__restrict__ class_items, uint32_t count)
{
    constexpr int M = 16 * T, N2 = T, C = T / 16, IPW = (T >= 64) ? 1 : 64 / T, NW = (T > 64) ? T / 64 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t grp = (T >= 64) ? 0u : lane / (uint32_t)T, t = (T >= 64) ? tid : lane % (uint32_t)T;
    cplx *L = reinterpret_cast<cplx *>(lds_raw) + grp * (uint32_t)M;
    WaveSums *ws = reinterpret_cast<WaveSums *>(lds_raw);      /* T > 64: scratch of the first reductions, before the region's first real use */

    const uint32_t nwg = (count + (uint32_t)IPW - 1u) / (uint32_t)IPW;
    const uint32_t wpos = xcd_position(blockIdx.x, nwg);
    const uint32_t pos = wpos * (uint32_t)IPW + grp;
    if (wpos >= nwg || pos >= count) return;              /* (the lanes of an item leave together) */
    const SrlaAutocorrItem it = class_items[pos];
    const InputView iv = input_view(jp, it.lshift);
    const uint32_t item_idx = it.item, n = it.n, bps = jp.bits_per_sample;
    const int32_t *in = input + it.sample_off;
    const bool aligned8 = ((reinterpret_cast<uintptr_t>(in) & 7u) == 0) && ((iv.stride & 1u) == 0);
    const bool first_pass = (pass == 1) || (jp.ltp_order == 0);   /* the pass that owns the pre-emphasis tap */
    SrlaItemResult *out = &results[item_idx];
    const uint32_t lane0 = grp * (uint32_t)T;              /* the item's first lane (T <= 64) */

    /* the lane's class is t: s0 / s1[j] = samples 2 slot, 2 slot + 1 of slot t + N2 j; pv[j] = the sample in front of them (the
     * very first sample has itself in front, srla_utility.c:342) */
    int32_t s0[16], s1[16], pv[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t e = 2u * (t + (uint32_t)(N2 * j));
        load_pair(in, iv, it.variant, e, n, aligned8, s0[j], s1[j]);
        pv[j] = (e == 0 || e >= n) ? s0[j] : load_variant(in, iv, it.variant, e - 1u);
    }
    SCHED_FENCE();

    int32_t coef;
    if (first_pass) {
        /* exact integer correlations r0 = sum x^2, r1 = sum x[i] x[i+1] (srla_utility.c:226-240) */
        long long r0 = 0, r1 = 0;
        uint32_t absmax = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t e = 2u * (t + (uint32_t)(N2 * j));
            const long long a = s0[j], b = s1[j], c = (e + 2u < n) ? (long long)load_variant(in, iv, it.variant, e + 2u) : 0ll;
            r0 += a * a + b * b;
            r1 += a * b + b * c;
            const uint32_t ua = (a < 0) ? (uint32_t)(-a) : (uint32_t)a, ub = (b < 0) ? (uint32_t)(-b) : (uint32_t)b;
            absmax = (ua > absmax) ? ua : absmax;
            absmax = (ub > absmax) ? ub : absmax;
        }
        long long sum0, sum1;
        uint32_t am;
        if (T > 64) {
            const long long w0 = wave_sum_i64(r0), w1 = wave_sum_i64(r1);
            const uint32_t wm = wave_max_u32(absmax);
            if (lane == 0) { ws->r0[tid >> 6] = w0; ws->r1[tid >> 6] = w1; ws->amax[tid >> 6] = wm; }
            __syncthreads();
            sum0 = 0; sum1 = 0; am = 0;
            for (int w = 0; w < NW; w++) { sum0 += ws->r0[w]; sum1 += ws->r1[w]; am = (ws->amax[w] > am) ? ws->amax[w] : am; }
        } else {
            sum0 = group_sum_i64(r0, T); sum1 = group_sum_i64(r1, T); am = group_max_u32(absmax, T);
        }
        uint32_t flags = (n & 1u) ? SRLA_ITEM_ODD_LENGTH : 0u;
        if (am == 0) flags |= SRLA_ITEM_INPUT_ZERO;
        const bool exact = am < (1u << 23) && sum0 < (1LL << 53);
        int32_t c = 0;
        if (exact) {
            /* every partial sum of the reference's double accumulation is an exactly representable integer, so the summation
             * order does not matter */
            const double d0 = (double)sum0, d1 = (double)sum1;
            if (!(d0 < 1e-6)) {
                c = (int32_t)round_half_away((d1 / d0) * 16.0);
                c = (c < -16) ? -16 : ((c > 15) ? 15 : c);
            }
        } else if (t == 0) {
            /* literally (the rounding depends on the order): one lane, rare */
            double curr = load_variant(in, iv, it.variant, 0), succ = (n > 1) ? load_variant(in, iv, it.variant, 1) : 0.0;
            double d0 = 0.0, d1 = 0.0;
            for (uint32_t i = 0; i + 2 < n; i++) {
                const double nn = load_variant(in, iv, it.variant, i + 2);
                d0 += curr * curr; d1 += curr * succ; curr = succ; succ = nn;
            }
            d0 += curr * curr; d1 += curr * succ; curr = succ; d0 += curr * curr;
            if (!(d0 < 1e-6)) {
                c = (int32_t)round_half_away((d1 / d0) * 16.0);
                c = (c < -16) ? -16 : ((c > 15) ? 15 : c);
            }
        }
        if (T > 64) {
            if (!exact) {                                     /* (uniform over the workgroup) */
                if (t == 0) ws->coef = c;
                __syncthreads();
                c = ws->coef;
            }
            coef = c;
        } else {
            const int32_t cb = lane_read(c, lane0);
            coef = exact ? c : cb;
        }
        if (t == 0) {
            /* this pass initialises the item record */
            out->preemph_prev = s0[0];
            out->preemph_coef = coef;
            out->lpc_order = 0; out->lpc_rshift = 0; out->use_sum = 0; out->ltp_period = 0;
            out->ltp_coef[0] = 0; out->ltp_coef[1] = 0; out->ltp_coef[2] = 0;
            out->code_length = 0; out->res_code_type = 0; out->res_porder = 0; out->res_bits = 0;
            out->flags = flags; out->pad[0] = 0; out->pad[1] = 0;
        }
    } else {
        coef = out->preemph_coef;
        /* No pitch found: the LPC analysis sees the very signal the LTP analysis saw, and its lags are the first of the 263
         * already stored by that pass. */
        if (out->ltp_period == 0 && dbg == nullptr) return;
    }
    if (pass == 0 && jp.max_order == 0) return;              /* preset 0: fixed order 0, no LPC analysis needed */

    /* pre-emphasis in registers: y[i] = x[i] - ((x[i-1] * coef) >> 4) (srla_utility.c:342) */
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int32_t a = s0[q], b = s1[q];
        s0[q] = (int32_t)((uint32_t)a - (uint32_t)((int32_t)((uint32_t)pv[q] * (uint32_t)coef) >> 4));
        s1[q] = (int32_t)((uint32_t)b - (uint32_t)((int32_t)((uint32_t)a * (uint32_t)coef) >> 4));
    }

    if (pass == 0 && jp.ltp_order > 0) {
        const uint32_t period = out->ltp_period;
        if (period > 0) {                                     /* (uniform over the item) */
            /* long-term predictor (srla_lpc_predict.c:267-294): the pre-emphasised signal through the item's LDS region */
            int32_t *yl = reinterpret_cast<int32_t *>(L);
            const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
            item_sync<T>();                                   /* (the reduction scratch lies in the same region) */
            const int32_t c0 = out->ltp_coef[0], c1 = out->ltp_coef[1], c2 = out->ltp_coef[2];
#pragma unroll
            for (int j = 0; j < 16; j++) *reinterpret_cast<int2 *>(yl + 2u * (t + (uint32_t)(N2 * j))) = make_int2(s0[j], s1[j]);
            item_sync<T>();
            const uint32_t first = period + half_order + 1u;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t e = 2u * (t + (uint32_t)(N2 * j));
                if (e + 1u >= first && e < n) {
                    /* the taps of sample e start at e - period - half_order, those of e + 1 one further */
                    const uint32_t base = e + 1u - first;
                    const int32_t y0 = (e >= first) ? yl[base] : 0, y1 = yl[base + 1u];
                    const int32_t y2 = (taps == 3) ? yl[base + 2u] : 0, y3 = (taps == 3) ? yl[base + 3u] : 0;
                    if (e >= first) {
                        uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)y0;
                        if (taps == 3) acc += (uint32_t)c1 * (uint32_t)y1 + (uint32_t)c2 * (uint32_t)y2;
                        s0[j] = (int32_t)((uint32_t)s0[j] - (uint32_t)((int32_t)acc >> 5));
                    }
                    if (e + 1u < n) {
                        uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)y1;
                        if (taps == 3) acc += (uint32_t)c1 * (uint32_t)y2 + (uint32_t)c2 * (uint32_t)y3;
                        s1[j] = (int32_t)((uint32_t)s1[j] - (uint32_t)((int32_t)acc >> 5));
                    }
                }
            }
        }
    }

    /* Welch window (lpc.c:256-266) on the [-1, 1) normalised signal, zero padded: weight(e) = (divisor * smpl) * (n - 1 - smpl)
     * with smpl = e in the first half and n - 1 - e in the second; the middle sample of an odd block is left alone (zero: outside
     * chain mode).  e and n - 1 - e are formed as doubles by exact additions from one conversion per lane. */
    cplx x[16];
    {
        const double norm_bps = __builtin_ldexp(1.0, -(int)(bps - 1));
        const uint32_t half = n >> 1;
        const double d_nm1 = (double)(n - 1u), d2t = (double)(2u * t);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t e0 = 2u * (t + (uint32_t)(N2 * j));
            const double de0 = d2t + (double)(2 * N2 * j);                 /* (double)e0, exact */
            double w[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint32_t e = e0 + (uint32_t)i;
                const double de = de0 + (double)i, dr = d_nm1 - de;
                const int32_t y = i ? s1[j] : s0[j];
                double val = 0.0;
                if (e < n && (e < half || e >= n - half)) {
                    const bool firsth = e < half;
                    const double a = firsth ? de : dr, b = firsth ? dr : de;
                    const double in_d = (double)y * norm_bps;
                    const double wt = it.welch_divisor * a * b;
                    val = in_d * wt;
                }
                w[i] = val;
            }
            x[j] = make_double2(w[0], w[1]);
        }
    }
    SCHED_FENCE();

    const uint32_t num_lags = (pass == 1) ? SRLA_LTP_LAGS : (jp.max_order + 1);
    const uint32_t need = (num_lags + 1u) >> 1;
    const cplx *twbase = twiddles + it.tw_off;
    constexpr uint32_t ct = TwOff<M>::total();
    const cplx *tw_fwd = twbase, *tw_inv = twbase + ct, *rtw_fwd = twbase + 2u * ct, *rtw_inv = rtw_fwd + (uint32_t)(M / 2);

    transform_passes<T, -1, false>(x, L, tw_fwd, t, t, (uint32_t)M);

    /* ---- T3: X[k], k = rev4(v) + 256 wrev(c), to slot k; every lane then takes the bins i = t + N2 j of its class and their
     * partners m - i, and leaves in x[j] what the spectrum pass leaves in bin i */
    item_sync<T>();
#pragma unroll
    for (int g = 0; g < 16 / C; g++) {
        const uint32_t k0 = rev4(t + (uint32_t)(T * g));
#pragma unroll
        for (int c = 0; c < C; c++) L[sw3<T>(k0 + 256u * wrev<C>((uint32_t)c))] = x[C * g + c];
    }
    item_sync<T>();
    {
        /* bin by bin, the values and table entries of the next one in flight meanwhile (PIN as in transform_passes) */
        struct Bin { cplx own, par, wf, wi; };
        auto fetch = [&](const int j, const uint32_t tt) {
            const uint32_t i = tt + (uint32_t)(N2 * j);
            /* j < 8: i <= m / 2 - 1, the pair is (i, m - i) (i = 0, the DC bin of lane 0, is redone below: any in-range table
             * entry serves); else the pair is (m - i, i) */
            const uint32_t ip = (j < 8) ? ((i == 0) ? 1u : i) : (uint32_t)M - i;
            Bin b;
            b.own = L[sw3<T>(i)];
            b.par = L[sw3<T>(((uint32_t)M - i) & (uint32_t)(M - 1))];
            b.wf = rtw_fwd[ip - 1u];
            b.wi = rtw_inv[ip - 1u];
            return b;
        };
        uint32_t tp = t;
        PIN(tp);
        Bin cur = fetch(0, tp);
        const cplx dc = cur.own;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t tn = t;
            if (j == 0) PIN(tn); else PIN_AFTER(tn, x[j - 1].x);
            Bin nxt = cur;
            if (j + 1 < 16) nxt = fetch(j + 1, tn);
            /* i = m / 2 (lane 0, j = 8) pairs with itself */
            if (j < 8) x[j] = spectrum_bin<false>(cur.own, cur.par, cur.wf, cur.wi, false);
            else x[j] = spectrum_bin<true>(cur.par, cur.own, cur.wf, cur.wi, (j == 8) && t == 0);
            cur = nxt;
        }
        if (t == 0) {
            /* DC / Nyquist bin: x0 = re + im, x1 = re - im, squared (fft.c:187-191, lpc.c:358-359); then the inverse's
             * 0.5 (x0 + x1), 0.5 (x0 - x1) */
            const double a = dc.x + dc.y, b = dc.x - dc.y;
            const double pa = a * a, pb = b * b;
            x[0] = make_double2(0.5 * (pa + pb), 0.5 * (pa - pb));
        }
    }

    transform_passes<T, 1, true>(x, L, tw_inv, t, t, need);

    /* lags: output k (position 0 of unit v = t + T g, k = rev4(v)) holds the unscaled lags 2k and 2k + 1 (lpc.c:367-375) */
    const size_t stride = jp.num_items;
#pragma unroll
    for (int g = 0; g < 16 / C; g++) {
        const uint32_t k = rev4(t + (uint32_t)(T * g));
        if (k < need) {
            const cplx z = x[C * g];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint32_t li = 2u * k + (uint32_t)i;
                if (li < num_lags) {
                    const double lag = (i ? z.y : z.x) * it.acorr_norm;
                    lags_ws[(size_t)li * stride + item_idx] = lag;
                    if (dbg) dbg[(size_t)item_idx * SRLA_DBG_STRIDE + ((pass == 1) ? SRLA_DBG_LTPLAGS : SRLA_DBG_LAGS) + li] = lag;
                }
            }
        }
    }
}

/* items of ONE transform size: nfft = 1024, 2048, 4096 or 8192 points (T = nfft / 32 lanes each); not for chain mode
 * (history-dependent blocks keep to srla_autocorr, which models the reference's persistent buffer) */
extern "C" int srla_launch_autocorr_wave(hipStream_t stream, uint32_t nfft, const SrlaJobParams *jp, const int32_t *input, const void *twiddles,
                                         uint32_t pass, SrlaItemResult *results, double *lags_ws, double *dbg,
                                         const SrlaAutocorrItem *class_items, uint32_t count, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (count == 0) return 0;
#define LAUNCH(TT)                                                                                                          \
    do {                                                                                                                    \
        static bool done_ = false;                                                                                          \
        if (!done_) { (void)hipFuncSetAttribute((const void *)srla_autocorr_w<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done_ = true; } \
        const uint32_t ipw = (TT >= 64) ? 1u : 64u / TT, nwg = (count + ipw - 1u) / ipw, threads = (TT > 64) ? TT : 64u;    \
        const uint32_t lds = ipw * 16u * TT * 16u;                                                                                                         \
        hipExtLaunchKernelGGL((srla_autocorr_w<TT>), dim3(8u * ((nwg + 7u) >> 3)), dim3(threads), lds, stream, ev_start, ev_stop, 0, *jp, input, \
                              (const cplx *)twiddles, pass, results, lags_ws, dbg, class_items, count);                     \
    } while (0)
    switch (nfft) {
    case 1024: LAUNCH(32); break;
    case 2048: LAUNCH(64); break;
    case 4096: LAUNCH(128); break;
    case 8192: LAUNCH(256); break;
    default: return -1;
    }
#undef LAUNCH
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}
