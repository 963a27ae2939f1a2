/*
 * kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the SRLA encode hot path.
 *
 * The per-item analysis of ComputeCoefficientsPerChannel (srla_encoder.c:966-1205) is a pipeline of
 * kernels chosen by the SHAPE of each stage's parallelism, not by the reference's call graph:
 *
 *   srla_autocorr<R, T>     one workgroup per item, one launch per FFT-size class.  Samples are loaded once with
 *                           16-byte loads and stay in registers: exact integer correlations -> pre-emphasis tap ->
 *                           pre-emphasis (-> long-term predictor) -> Welch window -> real FFT in one padded LDS
 *                           buffer (fp64, operation order of libs/fft) -> forward symmetry pass, |X|^2 and inverse
 *                           symmetry pass fused -> inverse FFT pruned to the lags that are read -> lags.
 *   srla_pitch_solve        (LTP only) ONE LANE per item: the sequential pitch scan of lpc.c:1473-1555 (lags staged
 *                           in LDS) and the 3x3 Cholesky solve.
 *   srla_lpc_errvars(_lean) / srla_lpc_recursion  ONE LANE per item: Levinson-Durbin with the gamma dot product summed in index
 *                           order (lpc.c:417-438) -- inherently serial per item, so 64 recursions run side by
 *                           side in a wavefront (coefficients in registers for the preset orders, LDS otherwise).
 *   srla_order_select       one wave per item, lane = order: code-length estimate and its first strict minimum.
 *   srla_lpc_quantize(_regs)   one lane per item: predictor of the chosen order, 8-bit quantiser, tap cost.
 *   srla_residual_cost<R>   one workgroup per item: pre-emphasis (+LTP), the wrap-around int32 FIR on packed int16 / int8
 *                           planes (v_dot2_i32_i16 / v_dot4_i32_i8), residual to HBM (uint16 where it fits), partitioned
 *                           (recursive) Rice code-length search.
 *   srla_price_windows      stereo decision + block sizes + shortest path, one wave per window.
 *   srla_block_offsets      byte offset of every chosen block, the job's place in the stream, per-window sizes.
 *   srla_pack_blocks        one workgroup per chosen block: the COMPLETE block (header, payload fields, Huffman
 *                           coded taps, Rice coded residuals, Fletcher-16) assembled in LDS.
 *   srla_stream_out         the job's finished bytes, device buffer -> host memory (a few workgroups, PCIe paced) where the
 *                           host does not copy them itself when it collects the job (host_pipeline.cpp, Impl::dma_out).
 *   srla_or_reduce          whole-stream OR for the offset left shift.
 *
 * No MFMA: integer/fp64 butterflies and reductions, not a dense contraction.  All fp64 arithmetic
 * must round exactly like the C90 reference: compiled with -ffp-contract=off and pinned below.
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <float.h>
#include <algorithm>

#include "device_layout.h"
#include "kernels.h"

#pragma clang fp contract(off)

/* The narrow kernels of stream N (a few hundred latency-bound wavefronts: serial fp64 recursions, the window pricing) share their
 * SIMDs with the wide kernels' wavefronts, which issue VALU instructions back to back: at the default priority a recursion's
 * next instruction waits its turn behind them and the solve stage of a job took 0.35-0.53 ms beside them against 0.1 ms alone --
 * longer than the wide stream had work for, so srla_residual_cost of the job waited for it (timeline).  Raised
 * wave priority lets the few instructions they have go first; they are too few to slow the wide kernels down. */
#ifdef SRLA_NO_NARROW_PRIORITY
#define NARROW_KERNEL_PRIORITY() do { } while (0)
#else
#define NARROW_KERNEL_PRIORITY() __builtin_amdgcn_s_setprio(3)
#endif

#include "device_common.h"

/* -DSRLA_DIAG_PHASES (tools/r05_phases.sh; never in the shipped library): where the wavefronts of the two wide kernels spend their
 * time IN FLIGHT -- every wavefront stamps the shader clock at phase boundaries and adds the differences to a per-workgroup row of
 * srla_diag_phase[kernel][row][phase]; SRLAMI355X_DiagPhases copies the table out and clears it. */
#ifdef SRLA_DIAG_PHASES
__device__ unsigned long long srla_diag_phase[2][1024][16];
#define PHASE_INIT() unsigned long long ph_t_ = __builtin_amdgcn_s_memtime()
#define PHASE_PARAM , unsigned long long &ph_t_
#define PHASE_ARG , ph_t_
#define PHASE(KERNEL, K) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63u) == 0) atomicAdd(&srla_diag_phase[KERNEL][blockIdx.x & 1023u][K], t_ - ph_t_); ph_t_ = t_; } while (0)
extern "C" int SRLAMI355X_DiagPhases(unsigned long long *out /* [2][16] */)
{
    static unsigned long long host[2][1024][16];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(srla_diag_phase), sizeof host) != hipSuccess) return -1;
    for (int k = 0; k < 2; k++) for (int p = 0; p < 16; p++) { unsigned long long t = 0; for (int r = 0; r < 1024; r++) t += host[k][r][p]; out[16 * k + p] = t; }
    memset(host, 0, sizeof host);
    return hipMemcpyToSymbol(HIP_SYMBOL(srla_diag_phase), host, sizeof host) == hipSuccess ? 0 : -1;
}
#else
#define PHASE_INIT() do { } while (0)
#define PHASE(KERNEL, K) do { } while (0)
#define PHASE_PARAM
#define PHASE_ARG
#endif

/* ------------------------------------------------------------------------------ FFT ------ */
/* complex FFT of m points held interleaved in LDS: radix-4 decimation in frequency with the
 * reference's (Stockham) butterfly arithmetic (fft.c:71-136).  Butterfly inputs are staged in
 * registers, so one LDS buffer suffices (two barriers per stage); the stage's twiddle is fetched
 * together with the inputs so its latency overlaps the LDS reads.
 * LDS slots are not padded: one pad slot per 16 (which takes the first two stages' strided stores off the same
 * banks) was measured 2 % SLOWER than plain indexing -- the extra address arithmetic costs more than the conflicts.
 * PRUNE: only the first `need` complex outputs of the transform will be read (the inverse transform feeds a few
 * dozen lags).  Output k of butterfly (p, q) of the stage with stride s is read by a needed butterfly of a later
 * stage iff q + s k < need, so butterflies with q >= need are skipped and outputs with s k >= need are neither
 * multiplied by their twiddle nor stored. */
/* Slot of complex element c.  SWZ (the fused-pass transform below): the low three bits are XORed with bits 4-6, so that
 * the 16 consecutive elements one thread stores in the first fused pass land in different 16-byte columns than those of
 * its seven neighbours in the store group (a ds_write_b128 is served in groups of 8 lanes over 32 banks), while aligned
 * runs of 8 elements stay runs of 8 (the contiguous reads of every pass remain conflict-free). */
template <bool SWZ>
__device__ __forceinline__ uint32_t cidx(uint32_t c) { return SWZ ? (c ^ ((c >> 4) & 7u)) : c; }

template <int R, int NTK, bool PRUNE>
__device__ void fft_complex_lds(cplx *x, uint32_t m, int flag, const cplx *__restrict__ tw, uint32_t need)
{
    /* tw: per stage (sub-size n) three tables of n/4 entries each: w^p, w^2p, w^3p -- the host builds them with
     * the reference's own products (w2 = w1*w1, w3 = w1*w2, fft.c:95-96), so the values are identical */
    const uint32_t tid = threadIdx.x;
    uint32_t n = m, s = 1, log2s = 0;
    const uint32_t nb = m >> 2;
    /* Index arithmetic of a stage, with q + s p = bf (p = bf >> log2 s, q = bf & (s - 1)):
     *   inputs   q + s (p + k n/4)  = bf + k m/4          (s n = m)
     *   outputs  q + s (4 p + k)    = (4 bf - 3 q) + k s
     * i.e. each butterfly needs two bases and two uniform strides instead of eight computed addresses. */
    const uint32_t m4 = m >> 2;
    while (n > 2) {
        const uint32_t n1 = n >> 2;
        /* uniform: which outputs can matter at all */
        const bool k1 = !PRUNE || s < need, k2 = !PRUNE || 2 * s < need, k3 = !PRUNE || 3 * s < need;
        cplx a[R], b[R], c[R], d[R], w1[R], w2[R], w3[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t bf = tid + (uint32_t)r * NTK;
            const uint32_t q = bf & (s - 1);
            if (bf < nb && (!PRUNE || q < need)) {
                const uint32_t p = bf >> log2s;
                if (k1) w1[r] = tw[p];
                if (k2) w2[r] = tw[n1 + p];
                if (k3) w3[r] = tw[2 * n1 + p];
                a[r] = x[bf]; b[r] = x[bf + m4]; c[r] = x[bf + 2 * m4]; d[r] = x[bf + 3 * m4];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t bf = tid + (uint32_t)r * NTK;
            const uint32_t q = bf & (s - 1);
            if (bf < nb && (!PRUNE || q < need)) {
                const cplx apc = c_add(a[r], c[r]), amc = c_sub(a[r], c[r]), bpd = c_add(b[r], d[r]);
                const cplx bmd = c_sub(b[r], d[r]);
                /* (0, -flag) * (b - d): the reference evaluates 0*re - (-flag)*im and 0*im + (-flag)*re
                 * (fft.c:57-63, 104); for finite data that is exactly (flag*im, -flag*re) up to the sign of a zero */
                const cplx jbmd = (flag < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
                const uint32_t wb = 4u * bf - 3u * q;
                x[wb] = c_add(apc, bpd);
                if (k1) x[wb + s] = c_mul(w1[r], c_sub(amc, jbmd));
                if (k2) x[wb + 2 * s] = c_mul(w2[r], c_sub(apc, bpd));
                if (k3) x[wb + 3 * s] = c_mul(w3[r], c_add(amc, jbmd));
            }
        }
        __syncthreads();
        tw += 3 * n1;
        n >>= 2;
        s <<= 2;
        log2s += 2;
    }
    if (n == 2) {
        cplx a[2 * R], b[2 * R];
#pragma unroll
        for (int r = 0; r < 2 * R; r++) {
            const uint32_t q = tid + (uint32_t)r * NTK;
            if (q < s && (!PRUNE || q < need)) { a[r] = x[q]; b[r] = x[q + s]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2 * R; r++) {
            const uint32_t q = tid + (uint32_t)r * NTK;
            if (q < s && (!PRUNE || q < need)) {
                x[q] = c_add(a[r], b[r]);
                if (!PRUNE || q + s < need) x[q + s] = c_sub(a[r], b[r]);
            }
        }
        __syncthreads();
    }
}

/* The same transform with the length M and the direction FLAG known at compile time: the stage loop unrolls, so every stage's
 * sub-size, stride and table offset are constants -- p and q of a butterfly are a shift and a mask by immediates, the four inputs
 * and four outputs stand at one computed LDS address plus immediate offsets, the three table entries at one computed address
 * plus immediates, and the direction costs no selects.  (Half of srla_autocorr's VALU instructions were this bookkeeping, not
 * fp64 arithmetic: SQ_INSTS_VALU_*_F64 / SQ_INSTS_VALU = 0.49.)  Same butterflies, same operands, same bits. */
template <int R, int NTK, bool PRUNE, int M, int FLAG, bool FIRSTREG = false /* the first stage has been done from registers (fft_first_stage_regs): start at the second */>
__device__ __forceinline__ void fft_complex_lds_ct(cplx *x, const cplx *__restrict__ tw, const uint32_t need)
{
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t nb = M >> 2, m4 = M >> 2;
    constexpr int NST = (M >= 4096) ? 6 : ((M >= 1024) ? 5 : ((M >= 256) ? 4 : ((M >= 64) ? 3 : ((M >= 16) ? 2 : 1))));   /* radix-4 stages: n = M, M/4, ... > 2 */
#ifdef SRLA_FFT_NO_SWZ12
    constexpr bool SWZ12 = false;
#else
    constexpr bool SWZ12 = M >= 128;      /* the first stage's outputs permuted in LDS, see below */
#endif
    uint32_t twoff = 0;
#pragma unroll
    for (int st = 0; st < NST; st++) {
        const uint32_t n = (uint32_t)M >> (2 * st), s = 1u << (2 * st), log2s = 2u * (uint32_t)st;
        if (n <= 2) break;
        const uint32_t n1 = n >> 2;
        if (FIRSTREG && st == 0) { twoff += 3 * n1; continue; }
        const bool k1 = !PRUNE || s < need, k2 = !PRUNE || 2 * s < need, k3 = !PRUNE || 3 * s < need;
        cplx a[R], b[R], c[R], d[R], w1[R], w2[R], w3[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t bf = tid + (uint32_t)r * NTK;
            const uint32_t q = bf & (s - 1);
            if (bf < nb && (!PRUNE || q < need)) {
                const uint32_t p = bf >> log2s;
                const cplx *t = tw + twoff + p;
                if (k1) w1[r] = t[0];
                if (k2) w2[r] = t[n1];
                if (k3) w3[r] = t[2 * n1];
                /* SWZ12, second stage: the first stage left its outputs in the permuted order (below) */
                const cplx *xi = x + ((FIRSTREG && st == 1) ? (bf ^ ((bf >> 3) & 7u)) : ((SWZ12 && st == 1) ? (bf ^ ((bf >> 3) & 3u)) : bf));
                a[r] = xi[0]; b[r] = xi[m4]; c[r] = xi[2 * m4]; d[r] = xi[3 * m4];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t bf = tid + (uint32_t)r * NTK;
            const uint32_t q = bf & (s - 1);
            if (bf < nb && (!PRUNE || q < need)) {
                const cplx apc = c_add(a[r], c[r]), amc = c_sub(a[r], c[r]), bpd = c_add(b[r], d[r]);
                const cplx bmd = c_sub(b[r], d[r]);
                const cplx jbmd = (FLAG < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
                if (SWZ12 && !FIRSTREG && st == 0) {
                    /* First stage: a thread's four outputs are the consecutive elements 4 bf .. 4 bf + 3, so the eight lanes a
                     * ds_write_b128 is served in store 64 bytes apart: output k of every second lane lands in the same 16-byte
                     * column of the eight (4-way conflicts on all four stores).  Element e therefore goes to e ^ ((e >> 3) & 3):
                     * the four elements of a thread are permuted among themselves, differently in the four lanes that share a
                     * column, and the eight stores land in eight columns.  The second stage reads elements bf + k m/4, lanes of a
                     * quad permuted within the quad -- a ds_read_b128 is served in groups made of whole quads, so it stays
                     * conflict-free. */
                    const uint32_t kx = (bf >> 1) & 3u;
                    cplx *xo = x + 4u * bf;
                    xo[kx] = c_add(apc, bpd);
                    if (k1) xo[1u ^ kx] = c_mul(w1[r], c_sub(amc, jbmd));
                    if (k2) xo[2u ^ kx] = c_mul(w2[r], c_sub(apc, bpd));
                    if (k3) xo[3u ^ kx] = c_mul(w3[r], c_add(amc, jbmd));
                } else {
                cplx *xo = x + (4u * bf - 3u * q);
                xo[0] = c_add(apc, bpd);
                if (k1) xo[s] = c_mul(w1[r], c_sub(amc, jbmd));
                if (k2) xo[2 * s] = c_mul(w2[r], c_sub(apc, bpd));
                if (k3) xo[3 * s] = c_mul(w3[r], c_add(amc, jbmd));
                }
            }
        }
        __syncthreads();
        twoff += 3 * n1;
    }
    constexpr uint32_t last_n = (uint32_t)M >> (2 * NST);       /* 2 when log2 M is odd, else 1 */
    if (last_n == 2) {
        constexpr uint32_t s = (uint32_t)M >> 1;
        cplx a[2 * R], b[2 * R];
#pragma unroll
        for (int r = 0; r < 2 * R; r++) {
            const uint32_t q = tid + (uint32_t)r * NTK;
            if (q < s && (!PRUNE || q < need)) { a[r] = x[q]; b[r] = x[q + s]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2 * R; r++) {
            const uint32_t q = tid + (uint32_t)r * NTK;
            if (q < s && (!PRUNE || q < need)) {
                x[q] = c_add(a[r], b[r]);
                if (!PRUNE || q + s < need) x[q + s] = c_sub(a[r], b[r]);
            }
        }
        __syncthreads();
    }
}

/* The first radix-4 stage of the forward transform fed from registers.  A thread of the 4096- and 8192-point classes loads the
 * sample chunks 4 tid + c nfft/4, c = 0..3 -- complex elements 2 tid + c m/4 and 2 tid + 1 + c m/4: exactly the four inputs of
 * butterflies 2 tid and 2 tid + 1 (inputs bf + k m/4).  With that assignment the windowed signal never passes through LDS (a
 * store and a load of the whole buffer and one barrier less per item).  The butterflies are fft_complex_lds_ct's (the same
 * operands in the same order); their eight consecutive outputs 8 tid .. 8 tid + 7 go to e ^ ((e >> 3) & 7) -- the eight lanes a
 * 16-byte store is served in land in eight columns -- and the second stage reads bf + k m/4 at bf ^ ((bf >> 3) & 7): lanes
 * permuted within their block of eight by at most their block's index, which leaves every lane group of a 16-byte load (whole
 * quads of four blocks) on sixteen columns (checked in tests/test_kernel_models.py). */
template <int M>
__device__ __forceinline__ void fft_first_stage_regs(cplx *x, const double (&w)[4][4], const cplx *__restrict__ tw)
{
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t n1 = (uint32_t)M >> 2;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t bf = 2u * tid + (uint32_t)h;
        const cplx *t = tw + bf;
        const cplx w1 = t[0], w2 = t[n1], w3 = t[2 * n1];
        const cplx a = make_double2(w[0][2 * h], w[0][2 * h + 1]), b = make_double2(w[1][2 * h], w[1][2 * h + 1]);
        const cplx c = make_double2(w[2][2 * h], w[2][2 * h + 1]), d = make_double2(w[3][2 * h], w[3][2 * h + 1]);
        const cplx apc = c_add(a, c), amc = c_sub(a, c), bpd = c_add(b, d), bmd = c_sub(b, d);
        const cplx jbmd = make_double2(-bmd.y, bmd.x);                                /* forward: flag = -1 */
        cplx *xo = x + 8u * tid;
        const uint32_t sw = tid & 7u, j0 = 4u * (uint32_t)h;
        xo[(j0 + 0u) ^ sw] = c_add(apc, bpd);
        xo[(j0 + 1u) ^ sw] = c_mul(w1, c_sub(amc, jbmd));
        xo[(j0 + 2u) ^ sw] = c_mul(w2, c_sub(apc, bpd));
        xo[(j0 + 3u) ^ sw] = c_mul(w3, c_add(amc, jbmd));
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t complex_table_len(uint32_t m)
{
    uint32_t t = 0;
    for (uint32_t n = m; n > 2; n >>= 2) t += 3 * (n >> 2);
    return t;
}

/* ---- The transform with wave-private stages (round 5; M <= 2048 complex points on at most four wavefronts) ------------------
 * After the first radix-4 stage the Stockham data splits into four independent sub-transforms by index mod 4: a later stage
 * with stride s (a multiple of 4) reads q + s (p + k n/4) and writes q + s (4 p + k) (fft.c:71-128), so q mod 4 never changes.
 * The four residues are therefore kept as four contiguous REGIONS of the LDS buffer -- element e stands at region e & 3,
 * position e >> 2 -- and every wavefront owns whole regions: inside a region, butterfly bf = 4 j + rho of the stage with
 * stride s = 4 s' is butterfly j of an ordinary stage with stride s' on M / 4 points (inputs j + k M/16, outputs
 * 4 j - 3 (j & (s' - 1)) + k s', table entry j >> log2 s' of the SAME table: p is the same number).  A wavefront's LDS
 * operations execute in order, so stages 2 .. last of either direction need no workgroup barrier at all: each wavefront
 * runs through its regions at its own pace.  What is left of the barriers: one behind the first stage (which is in place:
 * butterfly bf reads the elements bf + k M/4 and leaves output k = element 4 bf + k at position bf of region k -- the same
 * four slots), one in front of the spectrum pass, one inside it (it reads the regions and writes the inverse's input order),
 * one behind it, one behind the inverse's first stage, one in front of the lag stores.
 * Same butterflies, same operands, same operation order: the same bits.
 *
 * fft_regions: stages 2 .. last (and the closing radix-2 stage) of the M-point transform on the region layout.
 *   INSWZ: the regions' positions come permuted by fft_swz (the inverse: its first stage ran in place on the spectrum pass's
 *          permuted output, see spectrum_power_pass_regions).
 * Lane -> butterfly: local index u = lane + 64 r; the wavefront's RPW = 4 / (NTK / 64) regions have BPR = M / 16 butterflies
 * each per stage: region = wave RPW + u / BPR, j = u % BPR. */
__device__ __forceinline__ uint32_t fft_swz(uint32_t e) { return e ^ ((e >> 3) & 3u); }

template <int R, int NTK, int M, int FLAG, bool PRUNE, bool INSWZ>
__device__ __forceinline__ void fft_regions(cplx *x, const cplx *__restrict__ tw, const uint32_t need)
{
    constexpr int NW = NTK / 64, RPW = 4 / NW;
    constexpr uint32_t BPR = (uint32_t)M >> 4, QM = (uint32_t)M >> 2, SUB4 = (uint32_t)M >> 4;   /* butterflies per region and stage; region size; quarter of a region */
    static_assert(NW >= 1 && NW <= 4 && RPW * (int)BPR == 64 * R, "every wavefront owns whole regions");
    constexpr int NST = (M >= 4096) ? 6 : ((M >= 1024) ? 5 : ((M >= 256) ? 4 : ((M >= 64) ? 3 : ((M >= 16) ? 2 : 1))));
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t twoff = 3u * ((uint32_t)M >> 2);                 /* behind the first stage's tables */
#pragma unroll
    for (int st = 1; st < NST; st++) {
        const uint32_t n = (uint32_t)M >> (2 * st), s = 1u << (2 * st), sp = s >> 2, log2sp = 2u * (uint32_t)(st - 1);
        if (n <= 2) break;
        const uint32_t n1 = n >> 2;
        const bool k1 = !PRUNE || s < need, k2 = !PRUNE || 2 * s < need, k3 = !PRUNE || 3 * s < need;
        cplx a[R], b[R], c[R], d[R], w1[R], w2[R], w3[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t u = lane + 64u * (uint32_t)r;
            const uint32_t rho = wave * (uint32_t)RPW + u / BPR, j = u % BPR;
            const uint32_t qp = j & (sp - 1);
            if (!PRUNE || 4u * qp + rho < need) {
                const uint32_t p = j >> log2sp;
                const cplx *t = tw + twoff + p;
                if (k1) w1[r] = t[0];
                if (k2) w2[r] = t[n1];
                if (k3) w3[r] = t[2 * n1];
                const cplx *xi = x + rho * QM + (((INSWZ && st == 1) || st == 2) ? fft_swz(j) : j);
                a[r] = xi[0]; b[r] = xi[SUB4]; c[r] = xi[2 * SUB4]; d[r] = xi[3 * SUB4];
            }
        }
        /* (no barrier: the wavefront's own loads above are executed before its stores below) */
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t u = lane + 64u * (uint32_t)r;
            const uint32_t rho = wave * (uint32_t)RPW + u / BPR, j = u % BPR;
            const uint32_t qp = j & (sp - 1);
            if (!PRUNE || 4u * qp + rho < need) {
                const cplx apc = c_add(a[r], c[r]), amc = c_sub(a[r], c[r]), bpd = c_add(b[r], d[r]);
                const cplx bmd = c_sub(b[r], d[r]);
                const cplx jbmd = (FLAG < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
                if (st == 1) {
                    /* the region's first stage: a thread's four outputs are the consecutive positions 4 j .. 4 j + 3; permuted
                     * among themselves as in fft_complex_lds_ct (position e goes to fft_swz(e)), the next stage reads at fft_swz */
                    const uint32_t kx = (j >> 1) & 3u;
                    cplx *xo = x + rho * QM + 4u * j;
                    xo[kx] = c_add(apc, bpd);
                    if (k1) xo[1u ^ kx] = c_mul(w1[r], c_sub(amc, jbmd));
                    if (k2) xo[2u ^ kx] = c_mul(w2[r], c_sub(apc, bpd));
                    if (k3) xo[3u ^ kx] = c_mul(w3[r], c_add(amc, jbmd));
                } else {
                    cplx *xo = x + rho * QM + (4u * j - 3u * qp);
                    xo[0] = c_add(apc, bpd);
                    if (k1) xo[sp] = c_mul(w1[r], c_sub(amc, jbmd));
                    if (k2) xo[2 * sp] = c_mul(w2[r], c_sub(apc, bpd));
                    if (k3) xo[3 * sp] = c_mul(w3[r], c_add(amc, jbmd));
                }
            }
        }
        asm volatile("" ::: "memory");
        twoff += 3 * n1;
    }
    constexpr uint32_t last_n = (uint32_t)M >> (2 * NST);       /* 2 when log2 M is odd, else 1 */
    if (last_n == 2) {
        /* the radix-2 stage (stride M / 2): the pairs (q', q' + M / 8) of every region */
        constexpr uint32_t HP = (uint32_t)M >> 3, s = (uint32_t)M >> 1;
        cplx a[2 * R], b[2 * R];
#pragma unroll
        for (int r = 0; r < 2 * R; r++) {
            const uint32_t u = lane + 64u * (uint32_t)r;
            const uint32_t rho = wave * (uint32_t)RPW + u / HP, qp = u % HP;
            if (!PRUNE || 4u * qp + rho < need) { a[r] = x[rho * QM + qp]; b[r] = x[rho * QM + qp + HP]; }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 2 * R; r++) {
            const uint32_t u = lane + 64u * (uint32_t)r;
            const uint32_t rho = wave * (uint32_t)RPW + u / HP, qp = u % HP;
            if (!PRUNE || 4u * qp + rho < need) {
                x[rho * QM + qp] = c_add(a[r], b[r]);
                if (!PRUNE || 4u * qp + rho + s < need) x[rho * QM + qp + HP] = c_sub(a[r], b[r]);
            }
        }
        asm volatile("" ::: "memory");
    }
}

/* The first radix-4 stage on the region layout, in place: butterfly bf reads the elements bf + k M/4 (natural order, or -- SWZ,
 * the inverse -- at fft_swz of their index: what spectrum_power_pass_regions leaves) and puts output k where input k stood,
 * which is position bf (or fft_swz(bf)) of region k.  No thread touches another's slots, so there is no barrier between its loads
 * and stores; one barrier behind. */
template <int R, int NTK, int M, int FLAG, bool SWZ>
__device__ __forceinline__ void fft_first_stage_regions(cplx *x, const cplx *__restrict__ tw)
{
    constexpr uint32_t QM = (uint32_t)M >> 2;
    static_assert(R * NTK == (int)QM, "one first-stage butterfly per thread and r");
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t bf = threadIdx.x + (uint32_t)r * NTK;
        const cplx *t = tw + bf;
        const cplx w1 = t[0], w2 = t[QM], w3 = t[2 * QM];
        cplx *xi = x + (SWZ ? fft_swz(bf) : bf);
        const cplx a = xi[0], b = xi[QM], c = xi[2 * QM], d = xi[3 * QM];
        const cplx apc = c_add(a, c), amc = c_sub(a, c), bpd = c_add(b, d), bmd = c_sub(b, d);
        const cplx jbmd = (FLAG < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
        xi[0] = c_add(apc, bpd);
        xi[QM] = c_mul(w1, c_sub(amc, jbmd));
        xi[2 * QM] = c_mul(w2, c_sub(apc, bpd));
        xi[3 * QM] = c_mul(w3, c_add(amc, jbmd));
    }
    __syncthreads();
}

/* The same fed from registers (fft_first_stage_regs' assignment: thread tid holds the inputs of butterflies 2 tid and 2 tid + 1 of
 * the forward transform): the two outputs k go to positions 2 tid, 2 tid + 1 of region k -- 32 consecutive bytes per lane. */
template <int M>
__device__ __forceinline__ void fft_first_stage_regs_regions(cplx *x, const double (&w)[4][4], const cplx *__restrict__ tw)
{
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t QM = (uint32_t)M >> 2;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t bf = 2u * tid + (uint32_t)h;
        const cplx *t = tw + bf;
        const cplx w1 = t[0], w2 = t[QM], w3 = t[2 * QM];
        const cplx a = make_double2(w[0][2 * h], w[0][2 * h + 1]), b = make_double2(w[1][2 * h], w[1][2 * h + 1]);
        const cplx c = make_double2(w[2][2 * h], w[2][2 * h + 1]), d = make_double2(w[3][2 * h], w[3][2 * h + 1]);
        const cplx apc = c_add(a, c), amc = c_sub(a, c), bpd = c_add(b, d), bmd = c_sub(b, d);
        const cplx jbmd = make_double2(-bmd.y, bmd.x);                                /* forward: flag = -1 */
        cplx *xo = x + bf;
        xo[0] = c_add(apc, bpd);
        xo[QM] = c_mul(w1, c_sub(amc, jbmd));
        xo[2 * QM] = c_mul(w2, c_sub(apc, bpd));
        xo[3 * QM] = c_mul(w3, c_add(amc, jbmd));
    }
    __syncthreads();
}

/* One radix-4 butterfly with the reference's arithmetic (fft.c:98-110), as in fft_complex_lds.  y1..y3 are only formed
 * when wanted (pruned inverse). */
__device__ __forceinline__ void butterfly4(const cplx a, const cplx b, const cplx c, const cplx d, const int flag,
                                           const cplx w1, const cplx w2, const cplx w3, const bool k1, const bool k2, const bool k3,
                                           cplx &y0, cplx &y1, cplx &y2, cplx &y3)
{
    const cplx apc = c_add(a, c), amc = c_sub(a, c), bpd = c_add(b, d), bmd = c_sub(b, d);
    const cplx jbmd = (flag < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
    y0 = c_add(apc, bpd);
    if (k1) y1 = c_mul(w1, c_sub(amc, jbmd));
    if (k2) y2 = c_mul(w2, c_sub(apc, bpd));
    if (k3) y3 = c_mul(w3, c_add(amc, jbmd));
}

/* lanes 32-63 of a <-> lanes 0-31 of b (v_permlane32_swap_b32, gfx950), one dword at a time */
__device__ __forceinline__ void swap_halves(double &a, double &b)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const u32x2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi.x, (int)lo.x);
    b = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void swap_halves(cplx &a, cplx &b) { swap_halves(a.x, b.x); swap_halves(a.y, b.y); }

/* The same transform as fft_complex_lds -- the same butterflies on the same operands, hence the same bits -- with two
 * radix-4 stages per LDS round trip.  The stage with sub-size n and stride s splits into s independent transforms of n
 * points (fixed q = position mod s).  A "unit" u = q + s p' is the 16 elements q + s (p' + i n/16), i = k' + 4 k, of one
 * of them (= x[u + i m/16]: contiguous over the lanes): four butterflies p = p' + k' n/16 of this stage and, on their
 * outputs, the four butterflies (q + s k, p') of the next stage, results at q + s (k + 4 k'' + 16 p').
 * TWO lanes share a unit: lane l < 32 of a wavefront does the butterflies k' = 0, 2 of this stage and lane l + 32 does
 * k' = 1, 3; two v_permlane32_swap per complex value then leave, in both lanes and in the same registers, the four
 * operands (k' = 0..3) of the next stage's butterflies k = 0, 1 (lower lane) and k = 2, 3 (upper lane) -- no selects, no
 * LDS.  A whole unit per thread would halve the wavefronts per item, and this kernel lives on occupancy (measured:
 * 8 instead of 16 wavefronts per CU costs the one-stage version 88 %, the one-thread-per-unit version of this one 33 %).
 * What remains after the fused passes is 8, 4 or 2 points per transform: one more trip (radix-4 followed by radix-2 in
 * registers, a radix-4 stage, or the radix-2 stage).  m = 2048: three round trips instead of six; the stores, which bound
 * the one-stage version (ds_write_b128: 13 cycles per wave-instruction, MI355X_MICROARCH.md), halve.
 * Needs NTK = m / 8.  Pruning as in fft_complex_lds, with the exact per-element test: an element at position pos of a
 * stage whose outputs have stride s is wanted iff pos mod 4 s < need. */
template <int NTK, bool PRUNE>
__device__ void fft_complex_lds16(cplx *x, const uint32_t m, const int flag, const cplx *__restrict__ tw, const uint32_t need)
{
    const uint32_t tid = threadIdx.x, role = (tid >> 5) & 1u;
    const uint32_t u = ((tid >> 6) << 5) | (tid & 31u);
    uint32_t n = m, s = 1, log2s = 0;
    const uint32_t m16 = m >> 4;
    while (n >= 16) {
        const uint32_t n1 = n >> 2, n2 = n >> 4;
        const cplx *twb = tw + 3 * n1;                         /* tables of the next stage (sub-size n / 4) */
        const uint32_t q = u & (s - 1), pp = u >> log2s;
        const bool active = u < m16 && (!PRUNE || q < need);
        const bool a1 = !PRUNE || q + s < need, a2 = !PRUNE || q + 2 * s < need, a3 = !PRUNE || q + 3 * s < need;
        cplx ya[2][4];
        if (active) {
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const uint32_t kp = role + 2u * (uint32_t)kk, p = pp + kp * n2;
                cplx w1, w2, w3;
                if (a1) w1 = tw[p];
                if (a2) w2 = tw[n1 + p];
                if (a3) w3 = tw[2 * n1 + p];
                const uint32_t ib = u + kp * m16;
                const cplx i0 = x[cidx<true>(ib)], i1 = x[cidx<true>(ib + 4 * m16)], i2 = x[cidx<true>(ib + 8 * m16)],
                           i3 = x[cidx<true>(ib + 12 * m16)];
                butterfly4(i0, i1, i2, i3, flag, w1, w2, w3, a1, a2, a3, ya[kk][0], ya[kk][1], ya[kk][2], ya[kk][3]);
            }
        }
        /* every lane: after this ya[kk][j] holds output k = 2 role + j of butterfly k' = 2 kk, ya[kk][j + 2] that of k' = 2 kk + 1 */
#pragma unroll
        for (int kk = 0; kk < 2; kk++) { swap_halves(ya[kk][0], ya[kk][2]); swap_halves(ya[kk][1], ya[kk][3]); }
        cplx out[2][4];
        if (active) {
            cplx wb1, wb2, wb3;
            if (!PRUNE || q + 4 * s < need) { wb1 = twb[pp]; wb2 = twb[n2 + pp]; wb3 = twb[2 * n2 + pp]; }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t qk = q + (2u * role + (uint32_t)j) * s;       /* the next stage's q */
                if (!PRUNE || qk < need) {
                    const bool b1 = !PRUNE || qk + 4 * s < need, b2 = !PRUNE || qk + 8 * s < need, b3 = !PRUNE || qk + 12 * s < need;
                    butterfly4(ya[0][j], ya[0][j + 2], ya[1][j], ya[1][j + 2], flag, wb1, wb2, wb3, b1, b2, b3,
                               out[j][0], out[j][1], out[j][2], out[j][3]);
                }
            }
        }
        __syncthreads();                                       /* only the stores wait for every thread's reads */
        if (active) {
            const uint32_t ob = q + ((16u * pp) << log2s);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t k = 2u * role + (uint32_t)j, qk = q + k * s;
                if (!PRUNE || qk < need) {
                    const bool b1 = !PRUNE || qk + 4 * s < need, b2 = !PRUNE || qk + 8 * s < need, b3 = !PRUNE || qk + 12 * s < need;
                    const uint32_t o = ob + (k << log2s);
                    x[cidx<true>(o)] = out[j][0];
                    if (b1) x[cidx<true>(o + 4 * s)] = out[j][1];
                    if (b2) x[cidx<true>(o + 8 * s)] = out[j][2];
                    if (b3) x[cidx<true>(o + 12 * s)] = out[j][3];
                }
            }
        }
        __syncthreads();
        tw += 3 * n1 + 3 * n2;
        n >>= 4;
        s <<= 4;
        log2s += 4;
    }
    if (n == 8) {
        /* radix-4 stage (two butterflies p = 0, 1) and the radix-2 stage on its outputs: s = m / 8 transforms of 8 points,
         * one per thread */
        const uint32_t q = tid;
        const bool active = q < s && (!PRUNE || q < need);
        cplx ya[2][4];
        const bool a1 = !PRUNE || q + s < need, a2 = !PRUNE || q + 2 * s < need, a3 = !PRUNE || q + 3 * s < need;
        if (active) {
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const cplx i0 = x[cidx<true>(q + (uint32_t)p * s)], i1 = x[cidx<true>(q + (uint32_t)(p + 2) * s)],
                           i2 = x[cidx<true>(q + (uint32_t)(p + 4) * s)], i3 = x[cidx<true>(q + (uint32_t)(p + 6) * s)];
                butterfly4(i0, i1, i2, i3, flag, tw[p], tw[2 + p], tw[4 + p], a1, a2, a3, ya[p][0], ya[p][1], ya[p][2], ya[p][3]);
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t q2 = q + (uint32_t)k * s;
                if (!PRUNE || q2 < need) {
                    x[cidx<true>(q2)] = c_add(ya[0][k], ya[1][k]);
                    if (!PRUNE || q2 + 4 * s < need) x[cidx<true>(q2 + 4 * s)] = c_sub(ya[0][k], ya[1][k]);
                }
            }
        }
        __syncthreads();
    } else if (n == 4) {
        /* one radix-4 stage, p = 0: s = m / 4 butterflies, two per thread */
        cplx y[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const uint32_t q = tid + (uint32_t)r * NTK;
            if (q < s && (!PRUNE || q < need)) {
                const bool a1 = !PRUNE || q + s < need, a2 = !PRUNE || q + 2 * s < need, a3 = !PRUNE || q + 3 * s < need;
                butterfly4(x[cidx<true>(q)], x[cidx<true>(q + s)], x[cidx<true>(q + 2 * s)], x[cidx<true>(q + 3 * s)], flag,
                           tw[0], tw[1], tw[2], a1, a2, a3, y[r][0], y[r][1], y[r][2], y[r][3]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const uint32_t q = tid + (uint32_t)r * NTK;
            if (q < s && (!PRUNE || q < need)) {
                x[cidx<true>(q)] = y[r][0];
                if (!PRUNE || q + s < need) x[cidx<true>(q + s)] = y[r][1];
                if (!PRUNE || q + 2 * s < need) x[cidx<true>(q + 2 * s)] = y[r][2];
                if (!PRUNE || q + 3 * s < need) x[cidx<true>(q + 3 * s)] = y[r][3];
            }
        }
        __syncthreads();
    } else if (n == 2) {
        cplx a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t q = tid + (uint32_t)r * NTK;
            if (q < s && (!PRUNE || q < need)) { a[r] = x[cidx<true>(q)]; b[r] = x[cidx<true>(q + s)]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t q = tid + (uint32_t)r * NTK;
            if (q < s && (!PRUNE || q < need)) {
                x[cidx<true>(q)] = c_add(a[r], b[r]);
                if (!PRUNE || q + s < need) x[cidx<true>(q + s)] = c_sub(a[r], b[r]);
            }
        }
        __syncthreads();
    }
}

/* Between the two transforms, one pass over the spectrum: the symmetry pass of the forward real FFT
 * (fft.c:164-183) for the pair (i, N/2 - i), the power spectrum of both bins (lpc.c:357-365), and the
 * symmetry pass of the inverse real FFT on the result -- the same thread owns the same pair in all three,
 * so nothing goes back to LDS in between.  rtw_fwd / rtw_inv [i-1] = (wr, wi) for pair i. */
template <int NTK, bool SWZ>
__device__ void spectrum_power_pass(cplx *x, uint32_t nfft, const cplx *__restrict__ rtw_fwd, const cplx *__restrict__ rtw_inv)
{
    const uint32_t quarter = nfft >> 2, m = nfft >> 1;
    if (threadIdx.x == 0) {
        /* DC / Nyquist bin: x0 = re + im, x1 = re - im, squared (fft.c:187-191, lpc.c:358-359); then the inverse's
         * 0.5 (x0 + x1), 0.5 (x0 - x1) */
        const cplx z = x[0];
        const double a = z.x + z.y, b = z.x - z.y;
        const double pa = a * a, pb = b * b;
        x[0] = make_double2(0.5 * (pa + pb), 0.5 * (pa - pb));
    }
    for (uint32_t i = 1 + threadIdx.x; i <= quarter; i += NTK) {
        const bool self = (i == m - i);                       /* the middle bin pairs with itself */
        const uint32_t ia = cidx<SWZ>(i), ib = cidx<SWZ>(m - i);
        double p1, p3;
        {
            const double c2 = -0.5;                           /* flag = -1 */
            const cplx w = rtw_fwd[i - 1];
            const cplx za = x[ia], zb = x[ib];
            const double x1 = za.x, x2 = za.y, x3 = zb.x, x4 = zb.y;
            const double wr = w.x, wi = w.y;
            const double h1r = 0.5 * (x1 + x3);
            const double h1i = 0.5 * (x2 - x4);
            const double h2r = -c2 * (x2 + x4);
            const double h2i = c2 * (x1 - x3);
            const double y1 = h1r + (wr * h2r) - (wi * h2i);
            const double y2 = h1i + (wr * h2i) + (wi * h2r);
            /* for the self-paired middle element the reference's second pair of stores wins */
            const double y3 = h1r - (wr * h2r) + (wi * h2i);
            const double y4 = -h1i + (wr * h2i) + (wi * h2r);
            p1 = y1 * y1 + y2 * y2;
            p3 = y3 * y3 + y4 * y4;
            if (self) p1 = p3;
        }
        {
            const double c2 = 0.5;                            /* flag = +1 */
            const cplx w = rtw_inv[i - 1];
            const double x1 = p1, x2 = 0.0, x3 = p3, x4 = 0.0;
            const double wr = w.x, wi = w.y;
            const double h1r = 0.5 * (x1 + x3);
            const double h1i = 0.5 * (x2 - x4);
            const double h2r = -c2 * (x2 + x4);
            const double h2i = c2 * (x1 - x3);
            const double y1 = h1r + (wr * h2r) - (wi * h2i);
            const double y2 = h1i + (wr * h2i) + (wi * h2r);
            const double y3 = h1r - (wr * h2r) + (wi * h2i);
            const double y4 = -h1i + (wr * h2i) + (wi * h2r);
            if (!self) x[ia] = make_double2(y1, y2);
            x[ib] = make_double2(y3, y4);
        }
    }
    __syncthreads();
}

/* The same pass between the two halves of the region-layout transform (fft_regions): reads the forward transform's output where
 * it stands -- bin e at position e >> 2 of region e & 3 -- and leaves the inverse's input in natural order permuted by fft_swz,
 * which fft_first_stage_regions<SWZ> reads in place.  Pair P = tid + it NTK: residue rho = P / (M/8), j' = P % (M/8) stand for bin
 * i = rho + 4 (j' + (rho == 0)) and its partner M - i = region (4 - rho) & 3, position M/4 - 1 - j': both runs are contiguous over
 * the lanes (no bank conflicts on the loads), and the stores at fft_swz(i) = i ^ ((i >> 3) & 3), i = rho + 4 j', put the eight lanes
 * of a store group on eight 16-byte columns.  The bins a thread writes are not the slots it read: every pair is loaded first, one
 * barrier, then arithmetic and stores (the arithmetic itself is spectrum_power_pass's, operation for operation). */
template <int NTK, int M>
__device__ __forceinline__ void spectrum_power_pass_regions(cplx *x, const cplx *__restrict__ rtw_fwd, const cplx *__restrict__ rtw_inv)
{
    constexpr uint32_t QM = (uint32_t)M >> 2, EIGHTH = (uint32_t)M >> 3;
    constexpr int ITS = (M / 2) / NTK;
    static_assert(ITS * NTK == M / 2, "whole rounds");
    const uint32_t tid = threadIdx.x;
    cplx za[ITS], zb[ITS], wf[ITS], wi_[ITS];
    cplx z0 = make_double2(0.0, 0.0);
    if (tid == 0) z0 = x[0];
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        const uint32_t P = tid + (uint32_t)it * NTK;
        const uint32_t rho = P / EIGHTH, jp = P % EIGHTH;
        const uint32_t i = rho + 4u * (jp + (rho == 0u ? 1u : 0u));
        wf[it] = rtw_fwd[i - 1]; wi_[it] = rtw_inv[i - 1];
        za[it] = x[rho * QM + jp + (rho == 0u ? 1u : 0u)];
        zb[it] = x[((4u - rho) & 3u) * QM + (QM - 1u - jp)];
    }
    __syncthreads();
    if (tid == 0) {
        /* DC / Nyquist bin: x0 = re + im, x1 = re - im, squared (fft.c:187-191, lpc.c:358-359); then the inverse's
         * 0.5 (x0 + x1), 0.5 (x0 - x1) */
        const double a = z0.x + z0.y, b = z0.x - z0.y;
        const double pa = a * a, pb = b * b;
        x[0] = make_double2(0.5 * (pa + pb), 0.5 * (pa - pb));
    }
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        const uint32_t P = tid + (uint32_t)it * NTK;
        const uint32_t rho = P / EIGHTH, jp = P % EIGHTH;
        const uint32_t i = rho + 4u * (jp + (rho == 0u ? 1u : 0u));
        const bool self = (i == (uint32_t)M - i);                       /* the middle bin pairs with itself */
        double p1, p3;
        {
            const double c2 = -0.5;                           /* flag = -1 */
            const cplx w = wf[it];
            const double x1 = za[it].x, x2 = za[it].y, x3 = zb[it].x, x4 = zb[it].y;
            const double wr = w.x, wi = w.y;
            const double h1r = 0.5 * (x1 + x3);
            const double h1i = 0.5 * (x2 - x4);
            const double h2r = -c2 * (x2 + x4);
            const double h2i = c2 * (x1 - x3);
            const double y1 = h1r + (wr * h2r) - (wi * h2i);
            const double y2 = h1i + (wr * h2i) + (wi * h2r);
            const double y3 = h1r - (wr * h2r) + (wi * h2i);
            const double y4 = -h1i + (wr * h2i) + (wi * h2r);
            p1 = y1 * y1 + y2 * y2;
            p3 = y3 * y3 + y4 * y4;
            if (self) p1 = p3;
        }
        {
            const double c2 = 0.5;                            /* flag = +1 */
            const cplx w = wi_[it];
            const double x1 = p1, x2 = 0.0, x3 = p3, x4 = 0.0;
            const double wr = w.x, wi = w.y;
            const double h1r = 0.5 * (x1 + x3);
            const double h1i = 0.5 * (x2 - x4);
            const double h2r = -c2 * (x2 + x4);
            const double h2i = c2 * (x1 - x3);
            const double y1 = h1r + (wr * h2r) - (wi * h2i);
            const double y2 = h1i + (wr * h2i) + (wi * h2r);
            const double y3 = h1r - (wr * h2r) + (wi * h2i);
            const double y4 = -h1i + (wr * h2i) + (wi * h2r);
            if (!self) x[fft_swz(i)] = make_double2(y1, y2);
            x[fft_swz((uint32_t)M - i)] = make_double2(y3, y4);
        }
    }
    __syncthreads();
}

/* circular autocorrelation of the (already windowed, zero padded) signal in buf (lpc.c:330-376): on return
 * complex slot cidx<F16>(i/2) component i&1 holds the unscaled lag i, for i < num_lags */
template <int R, int NTK, bool F16, int NFFT = 0, bool FIRSTREG = false, bool WP = false /* the region layout with wave-private stages: fft_regions */>
__device__ void autocorr_in_place(cplx *buf, uint32_t nfft, const cplx *__restrict__ twbase, uint32_t num_lags PHASE_PARAM)
{
    const uint32_t m = nfft >> 1;
    const uint32_t ct = complex_table_len(m), quarter = nfft >> 2;
    const cplx *tw_fwd = twbase;
    const cplx *tw_inv = twbase + ct;
    const cplx *rtw_fwd = twbase + 2 * ct;
    const cplx *rtw_inv = rtw_fwd + quarter;
    if constexpr (WP) {
        /* the first forward stage has been done (from LDS in place, or from registers): the regions are complete behind its barrier */
        fft_regions<R, NTK, NFFT / 2, -1, false, false>(buf, tw_fwd, m);
        __syncthreads();
        PHASE(0, 4);                                                  /* forward stages 2.. + barrier */
        spectrum_power_pass_regions<NTK, NFFT / 2>(buf, rtw_fwd, rtw_inv);
        PHASE(0, 5);                                                  /* spectrum pass (two barriers) */
        fft_first_stage_regions<R, NTK, NFFT / 2, 1, true>(buf, tw_inv);
        PHASE(0, 6);                                                  /* first inverse stage + barrier */
        fft_regions<R, NTK, NFFT / 2, 1, true, true>(buf, tw_inv, (num_lags + 1) >> 1);
        __syncthreads();
    } else if constexpr (NFFT != 0 && !F16) {
        /* the transform's length known at compile time (a launch of one FFT-size class outside chain mode) */
        fft_complex_lds_ct<R, NTK, false, NFFT / 2, -1, FIRSTREG>(buf, tw_fwd, m);
        spectrum_power_pass<NTK, false>(buf, nfft, rtw_fwd, rtw_inv);
        fft_complex_lds_ct<R, NTK, true, NFFT / 2, 1>(buf, tw_inv, (num_lags + 1) >> 1);
    } else if (F16) {
        fft_complex_lds16<NTK, false>(buf, m, -1, tw_fwd, m);
        spectrum_power_pass<NTK, true>(buf, nfft, rtw_fwd, rtw_inv);
        fft_complex_lds16<NTK, true>(buf, m, 1, tw_inv, (num_lags + 1) >> 1);
    } else {
        fft_complex_lds<R, NTK, false>(buf, m, -1, tw_fwd, m);
        spectrum_power_pass<NTK, false>(buf, nfft, rtw_fwd, rtw_inv);
        fft_complex_lds<R, NTK, true>(buf, m, 1, tw_inv, (num_lags + 1) >> 1);
    }
}

/* ------------------------------------------------------------ order choice (H2: libm) ----- */
/* srla_encoder.c:873-885 */
/* logscale: 1.0 (exact) in production; the tie tests falsify the device's log with it (SrlaJobParams) */
__device__ __forceinline__ double geometric_entropy(double mean_abs, uint32_t bps, double logscale)
{
    const double intmean = mean_abs * (double)(1 << (bps - 1));
    const double rho = 1.0 / (1.0 + intmean);
    const double invrho = 1.0 - rho;
    if (mean_abs < 1e-16) return 0.0;
    return -(invrho * ((log(invrho) * logscale) * 1.4426950408889634) + rho * ((log(rho) * logscale) * 1.4426950408889634)) / rho;
}

/* correctly rounded x^-0.5 for the 3x3 LTP solve (lpc.c:591 uses pow(sum, -0.5)) */
__device__ __forceinline__ double inv_sqrt_cr(double x)
{
    const double s = sqrt(x);
    const double s_lo = __builtin_fma(-s, s, x) / (2.0 * s);          /* sqrt(x) = s + s_lo  */
    const double r = 1.0 / s;
    const double e = __builtin_fma(-s, r, 1.0);                        /* 1 - s*r             */
    return r + r * (e - s_lo * r);
}

/* ================================================================================================
 * K1: srla_autocorr -- pass 0: LPC lags (after the LTP filter when a pitch was found),
 *                      pass 1: LTP lags.
 * ============================================================================================== */
struct SmallA {
    long long lscratch[2 * 8];
    uint32_t uscratch[8];
    int32_t preemph_coef;
    uint32_t flags;
    uint32_t pad[2];
};

/* (Laid OVER the FFT buffer instead of behind it -- a 4096-point item is then 32 768 bytes instead of 32 944 -- it changes nothing:
 * measured in round 4, srla_autocorr 0.212 ms per job either way; LDS is handed out in granules that leave four workgroups per CU.) */
extern "C" uint32_t srla_kernel_small_a_bytes(void) { return (uint32_t)((sizeof(SmallA) + 15) & ~15u); }

/* R: chunks of 8 samples per thread (8 R NTK >= nfft).  F16: the fused-pass transform (NTK = nfft / 32), else one stage
 * per round trip (R butterflies per thread and stage) */
/* the analysis of one item: the body of srla_autocorr (one FFT-size class per launch) and of srla_autocorr_pair (two classes in
 * one launch); `bid`: the workgroup's index within its class */
template <int R, int NTK, bool F16, int NFFT = 0 /* every item of the launch has this FFT size (0: they say themselves) */, bool WP = false /* fft_regions */>
__device__ __forceinline__ void autocorr_item(
    const SrlaJobParams &jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    const SrlaGeom *__restrict__ geoms, const cplx *__restrict__ twiddles, uint32_t fft_bytes, uint32_t pass,
    SrlaItemResult *__restrict__ results, double *__restrict__ lags_ws, double *__restrict__ dbg,
    const SrlaAutocorrItem *__restrict__ class_items, uint32_t count, double *__restrict__ chain_pool,
    const uint32_t *__restrict__ chain_tab, const uint32_t bid)
{
    constexpr int CH = 2 * R;   /* chunks of four samples per thread: covers 8 * R * NTK >= nfft */
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    cplx *buf = (cplx *)lds;
    SmallA *sm = (SmallA *)(lds + fft_bytes);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t pos = xcd_position(bid, count);
    if (pos >= count) return;
    PHASE_INIT();
    const SrlaAutocorrItem it = class_items[pos];                    /* items of one FFT-size class */
    const InputView iv = input_view(jp, it.lshift, input);
    const uint32_t item_idx = it.item;
    const struct { uint32_t nfft, tw_off; double welch_divisor, acorr_norm; } g = { it.nfft, it.tw_off, it.welch_divisor, it.acorr_norm };
    const uint32_t n = it.n, nfft = NFFT ? (uint32_t)NFFT : g.nfft, bps = jp.bits_per_sample;
    const int32_t *in = input + it.sample_off;
    const bool aligned = input_aligned(in, iv);
    const bool first_pass = (pass == 1) || (jp.ltp_order == 0);   /* the pass that owns the pre-emphasis tap */
    SrlaItemResult *out = &results[item_idx];
#ifdef SRLA_DIAG_PHASES
    asm volatile("" :: "s"(n), "s"(iv.sh), "s"(it.sample_off));
    PHASE(0, 9);                                                      /* item record + shift fetched */
#endif

    int32_t v[CH][4];
    int32_t pv[CH], nxv[CH];
    /* The sample before the chunk (pre-emphasis) and the one after it (r1) are the neighbouring lanes' -- lane l - 1 holds samples
     * i4 - 4 .. i4 - 1 of the same chunk round, lane l + 1 samples i4 + 4 .. (zeros beyond n, which is what nxv wants there) -- so
     * they come by DPP wave shifts; only the wavefront's first and last lane fetch theirs, in ONE load that touches two cache lines.
     * (Round 4 issued two single-sample loads per chunk and channel from every lane: 16-byte lane stride, so each of them walked the
     * same sixteen cache lines as the chunk's own 16-byte load -- two thirds of the kernel's L1 work, and the wavefronts spent a
     * quarter of their lifetime waiting for their samples: profiles/r05/phases_*.txt.) */
    int32_t edge[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
        load_chunk(in, iv, it.variant, i4, n, aligned, v[c]);
        const bool need_prev = lane == 0 && i4 != 0 && i4 < n, need_next = lane == 63 && first_pass && i4 + 4 < n;
        edge[c] = (need_prev || need_next) ? load_variant(in, iv, it.variant, need_prev ? i4 - 1 : i4 + 4) : 0;
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
        const int32_t from_below = __builtin_amdgcn_update_dpp(edge[c], v[c][3], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        pv[c] = (i4 == 0 || i4 >= n) ? v[c][0] : from_below;
        nxv[c] = __builtin_amdgcn_update_dpp(edge[c], v[c][0], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    }

#ifdef SRLA_DIAG_STOP
    if (jp.out_stride == 11) { int32_t t = 0; for (int c = 0; c < CH; c++) t ^= v[c][0] ^ v[c][3] ^ pv[c] ^ nxv[c]; if (t == 0x7fffffff) out->pad[1] = 1; return; }
#endif
    asm volatile("" :: "v"(v[0][0]), "v"(v[CH - 1][3]), "v"(pv[CH - 1]), "v"(nxv[CH - 1]));
    PHASE(0, 0);                                                      /* item record fetched, sample loads landed */
    int32_t coef;
    if (first_pass) {
        /* exact integer correlations r0 = sum x^2, r1 = sum x[i] x[i+1] (srla_utility.c:226-240) */
        long long r0 = 0, r1 = 0;
        uint32_t absmax = 0;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int32_t nx = nxv[c];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const long long x = v[c][i];
                const long long y = (i < 3) ? (long long)v[c][i + 1] : (long long)nx;
                r0 += x * x;
                r1 += x * y;
                const uint32_t a = (v[c][i] < 0) ? (uint32_t)(-(int64_t)v[c][i]) : (uint32_t)v[c][i];
                absmax = (a > absmax) ? a : absmax;
            }
        }
        r0 = wave_sum_i64(r0); r1 = wave_sum_i64(r1); absmax = wave_max_u32(absmax);
        if (lane == 0) { sm->lscratch[wave] = r0; sm->lscratch[8 + wave] = r1; sm->uscratch[wave] = absmax; }
        __syncthreads();
        /* every thread finishes the reduction and derives the tap itself (uniform values): no second barrier, no
         * single-lane section the other 255 threads wait for */
        long long s0 = 0, s1 = 0; uint32_t am = 0;
        for (int w = 0; w < NTK / WAVE; w++) { s0 += sm->lscratch[w]; s1 += sm->lscratch[8 + w]; am = (sm->uscratch[w] > am) ? sm->uscratch[w] : am; }
        uint32_t flags = (n & 1u) ? SRLA_ITEM_ODD_LENGTH : 0u;
        if (am == 0) flags |= SRLA_ITEM_INPUT_ZERO;
        if (am < (1u << 23) && s0 < (1LL << 53)) {
            /* every partial sum of the reference's double accumulation is an exactly representable
             * integer, so the summation order does not matter */
            const double d0 = (double)s0, d1 = (double)s1;
            int32_t c = 0;
            if (!(d0 < 1e-6)) {
                c = (int32_t)round_half_away((d1 / d0) * 16.0);
                c = (c < -16) ? -16 : ((c > 15) ? 15 : c);
            }
            coef = c;
        } else {
            /* srla_utility.c:226-240 literally (rounding depends on the order): one lane, rare */
            if (tid == 0) {
                double curr = load_variant(in, iv, it.variant, 0), succ = load_variant(in, iv, it.variant, 1);
                double d0 = 0.0, d1 = 0.0;
                for (uint32_t i = 0; i + 2 < n; i++) {
                    const double nn = load_variant(in, iv, it.variant, i + 2);
                    d0 += curr * curr; d1 += curr * succ; curr = succ; succ = nn;
                }
                d0 += curr * curr; d1 += curr * succ; curr = succ; d0 += curr * curr;
                int32_t c = 0;
                if (!(d0 < 1e-6)) {
                    c = (int32_t)round_half_away((d1 / d0) * 16.0);
                    c = (c < -16) ? -16 : ((c > 15) ? 15 : c);
                }
                sm->preemph_coef = c;
            }
            __syncthreads();
            coef = sm->preemph_coef;
        }
        if (tid == 0) {
            /* this pass initialises the item record */
            out->preemph_prev = v[0][0];      /* thread 0 holds sample 0 (not yet pre-emphasised) */
            out->preemph_coef = coef;
            out->lpc_order = 0; out->lpc_rshift = 0; out->use_sum = 0; out->ltp_period = 0;
            out->ltp_coef[0] = 0; out->ltp_coef[1] = 0; out->ltp_coef[2] = 0;
            out->code_length = 0; out->res_code_type = 0; out->res_porder = 0; out->res_bits = 0;
            out->flags = flags; out->pad[0] = 0; out->pad[1] = 0;
        }
    } else {
        coef = out->preemph_coef;
        /* No pitch found: the LPC analysis sees the very signal the LTP analysis saw, and its lags are the first of the
         * 263 already stored by that pass (same transform, the pruning of the inverse only skips work).  Not in chain
         * mode, where the call itself matters, nor when the lags are also wanted in the debug record. */
        if (out->ltp_period == 0 && chain_pool == nullptr && dbg == nullptr) return;
    }
#ifdef SRLA_DIAG_STOP
    if (jp.out_stride == 12) { if (coef == 0x7fffffff) out->pad[1] = 1; return; }
#endif
    /* Chain mode (the odd-length tail window of a stream, host_encoder.cpp): the launch reproduces one call of
     * the reference on its persistent FFT buffer (lpc.c:58,211).  chain_src - 1 is where the buffer's middle word
     * stands in chain_pool (the Welch window leaves it untouched for odd n, lpc.c:260-264); at chain_dump - 1 the
     * call leaves the complete buffer (all nfft words of the inverse transform) for the calls after it. */
    const bool chain = chain_pool != nullptr;
    if (pass == 0 && jp.max_order == 0 && !chain) return;   /* preset 0: fixed order 0, no LPC analysis needed */
    PHASE(0, 1);                                                      /* tap sums, reduction, pre-emphasis tap */

    /* pre-emphasis in registers: y[i] = x[i] - ((x[i-1] * coef) >> 4), x[-1] = x[0] (srla_utility.c:342) */
    if (jp.bits_per_sample <= 18) {
        /* narrow input: sample (at most 19 bits) times 5-bit tap on the full-rate 24-bit multiplier */
#pragma unroll
        for (int c = 0; c < CH; c++) {
            int32_t prev = pv[c];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int32_t cur = v[c][i];
                v[c][i] = (int32_t)((uint32_t)cur - (uint32_t)(__mul24(prev, coef) >> 4));
                prev = cur;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            int32_t prev = pv[c];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int32_t cur = v[c][i];
                v[c][i] = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)coef) >> 4));
                prev = cur;
            }
        }
    }

    if (pass == 0 && jp.ltp_order > 0) {
        const uint32_t period = out->ltp_period;
        if (period > 0) {
            /* long-term predictor (srla_lpc_predict.c:267-294): stage y in LDS, filter into registers */
            int32_t *ylds = (int32_t *)buf;
            const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
            const int32_t c0 = out->ltp_coef[0], c1 = out->ltp_coef[1], c2 = out->ltp_coef[2];
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
                if (i4 < nfft) *reinterpret_cast<int4 *>(ylds + i4) = make_int4(v[c][0], v[c][1], v[c][2], v[c][3]);
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t s = i4 + i;
                    if (s < n && s >= period + half_order + 1) {
                        const uint32_t base = s - period - half_order;
                        uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)ylds[base];
                        if (taps == 3) acc += (uint32_t)c1 * (uint32_t)ylds[base + 1] + (uint32_t)c2 * (uint32_t)ylds[base + 2];
                        v[c][i] = (int32_t)((uint32_t)v[c][i] - (uint32_t)((int32_t)acc >> 5));
                    }
                }
            }
            __syncthreads();
        }
    }

    /* Welch window (lpc.c:256-266) on the [-1,1) normalised signal, zero padded to nfft */
#if defined(SRLA_DIAG_STOP) || defined(SRLA_FFT_NO_FIRSTREG)
    constexpr bool FIRSTREG = false;
#else
    constexpr bool FIRSTREG = NFFT != 0 && !F16 && R == 2 && 8 * R * NTK == NFFT;   /* fft_first_stage_regs: the windowed chunks stay in registers */
#endif
    double wreg[FIRSTREG ? 4 : 1][4];
    {
        const double norm_bps = __builtin_ldexp(1.0, -(int)(bps - 1));
        const uint32_t half = n >> 1;
        /* weight(e) = (divisor * smpl) * (n - 1 - smpl), smpl = e in the first half and n - 1 - e in the second:
         * both factors are small integers, so they are formed as doubles by exact additions from one conversion per
         * thread instead of two int -> double conversions per sample (quarter-rate instructions) */
        const double d_tid4 = (double)(4u * tid), d_nm1 = (double)(n - 1u);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
            if (i4 < nfft) {
                double w[4];
                const double de0 = d_tid4 + (double)(4 * c * NTK);          /* (double)i4, exact */
                const bool first = i4 + 4u <= half, second = i4 >= n - half && i4 + 4u <= n;
                if (first || second) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const double de = de0 + (double)i, dr = d_nm1 - de;   /* (double)e and (double)(n - 1 - e), exact */
                        const double a = first ? de : dr, b = first ? dr : de;
                        const double in_d = (double)v[c][i] * norm_bps;
                        const double wt = g.welch_divisor * a * b;
                        w[i] = in_d * wt;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t e = i4 + i;
                        double val = 0.0;
                        if (e < n) {
                            uint32_t smpl; bool touched = true;
                            if (e < half) smpl = e;
                            else if (e >= n - half) smpl = n - 1 - e;
                            else { smpl = 0; touched = false; }   /* middle sample of an odd block */
                            if (touched) {
                                const double in_d = (double)v[c][i] * norm_bps;
                                const double wt = g.welch_divisor * (double)smpl * (double)(n - 1 - smpl);
                                val = in_d * wt;
                            } else if (chain && it.chain_src) {
                                val = chain_pool[it.chain_src - 1u];
                            }
                        }
                        w[i] = val;
                    }
                }
                if constexpr (FIRSTREG) {
                    wreg[c][0] = w[0]; wreg[c][1] = w[1]; wreg[c][2] = w[2]; wreg[c][3] = w[3];
                } else {
                    buf[cidx<F16>(i4 >> 1)] = make_double2(w[0], w[1]);
                    buf[cidx<F16>((i4 >> 1) + 1u)] = make_double2(w[2], w[3]);
                }
            }
        }
        if constexpr (!FIRSTREG) __syncthreads();
    }

    PHASE(0, 2);                                                      /* pre-emphasis, (LTP), window */
    const uint32_t num_lags = (pass == 1) ? SRLA_LTP_LAGS : (jp.max_order + 1);
    const bool dump = chain && it.chain_dump;
#ifdef SRLA_DIAG_STOP
    if (jp.out_stride == 13) { if (buf[tid].x == 1.2345e300) out->pad[1] = 1; return; }
    if (jp.out_stride >= 14 && jp.out_stride <= 16) {
        const uint32_t m = nfft >> 1, ct = complex_table_len(m), quarter = nfft >> 2;
        const cplx *twbase = twiddles + g.tw_off;
        if (F16) fft_complex_lds16<NTK, false>(buf, m, -1, twbase, m); else fft_complex_lds<R, NTK, false>(buf, m, -1, twbase, m);
        if (jp.out_stride >= 15) spectrum_power_pass<NTK, F16>(buf, nfft, twbase + 2 * ct, twbase + 2 * ct + quarter);
        if (jp.out_stride >= 16) { if (F16) fft_complex_lds16<NTK, true>(buf, m, 1, twbase + ct, (num_lags + 1) >> 1); else fft_complex_lds<R, NTK, true>(buf, m, 1, twbase + ct, (num_lags + 1) >> 1); }
        if (buf[tid].x == 1.2345e300) out->pad[1] = 1;
        return;
    }
#endif
    static_assert(!WP || (NFFT != 0 && !F16), "the region layout needs the transform's length at compile time");
    if constexpr (WP) {
        if constexpr (FIRSTREG) fft_first_stage_regs_regions<NFFT / 2>(buf, wreg, twiddles + g.tw_off);
        else fft_first_stage_regions<R, NTK, NFFT / 2, -1, false>(buf, twiddles + g.tw_off);
    } else {
        if constexpr (FIRSTREG) fft_first_stage_regs<NFFT / 2>(buf, wreg, twiddles + g.tw_off);
    }
    PHASE(0, 3);                                                      /* first forward stage + its barrier */
    autocorr_in_place<R, NTK, F16, NFFT, FIRSTREG, WP>(buf, nfft, twiddles + g.tw_off, (num_lags < nfft && !dump) ? num_lags : nfft PHASE_ARG);
    PHASE(0, 7);                                                      /* inverse stages 2.. + barrier (4-6: inside autocorr_in_place) */
    /* where complex element e of the result stands */
    auto slot = [&](uint32_t e) -> uint32_t { return WP ? ((e & 3u) * (uint32_t)(NFFT / 8) + (e >> 2)) : cidx<F16>(e); };
    if (dump) {
        double *dst = chain_pool + (it.chain_dump - 1u);
        for (uint32_t i = tid; i < nfft; i += NTK) { const cplx z = buf[slot(i >> 1)]; dst[i] = (i & 1u) ? z.y : z.x; }
    }

    const size_t stride = jp.num_items;
    for (uint32_t i = tid; i < num_lags; i += NTK) {
        double lag = 0.0;
        if (i < nfft) { const cplx z = buf[slot(i >> 1)]; lag = ((i & 1u) ? z.y : z.x) * g.acorr_norm; }
        else if (chain && it.chain_lags) {
            /* the reference copies 263 lags out of a shorter FFT buffer: what earlier calls left there */
            const uint32_t o = chain_tab[it.chain_lags - 1u + (i - nfft)];
            if (o) lag = chain_pool[o - 1u] * g.acorr_norm;
        }
        lags_ws[(size_t)i * stride + item_idx] = lag;
        if (dbg) dbg[(size_t)item_idx * SRLA_DBG_STRIDE + ((pass == 1) ? SRLA_DBG_LTPLAGS : SRLA_DBG_LAGS) + i] = lag;
    }
    PHASE(0, 8);                                                      /* lag stores */
}

template <int R, int NTK, bool F16, int NFFT = 0, bool WP = false>
__global__ __launch_bounds__(NTK) __attribute__((amdgpu_waves_per_eu(4, 8))) void srla_autocorr(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    const SrlaGeom *__restrict__ geoms, const cplx *__restrict__ twiddles, uint32_t fft_bytes, uint32_t pass,
    SrlaItemResult *__restrict__ results, double *__restrict__ lags_ws, double *__restrict__ dbg,
    const SrlaAutocorrItem *__restrict__ class_items, uint32_t count, double *__restrict__ chain_pool,
    const uint32_t *__restrict__ chain_tab)
{
    autocorr_item<R, NTK, F16, NFFT, WP>(jp, input, items, geoms, twiddles, fft_bytes, pass, results, lags_ws, dbg, class_items, count, chain_pool, chain_tab, blockIdx.x);
}

/* The 4096-point and the 2048-point class of a SMALL job in one launch (both run on 256 threads): a short stream's chain of
 * launches is a latency chain on a mostly idle device, and two class launches one after the other cost two launch floors where the
 * items of both fit the device together.  The workgroups of the larger class come first.  Registers and LDS are the larger class's
 * for every workgroup, which would halve the 2048-point items' occupancy in a full job: small jobs only (srla_launch_autocorr_pair). */
template <bool WP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void srla_autocorr_pair(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    const SrlaGeom *__restrict__ geoms, const cplx *__restrict__ twiddles, uint32_t fft_bytes, uint32_t pass,
    SrlaItemResult *__restrict__ results, double *__restrict__ lags_ws, double *__restrict__ dbg,
    const SrlaAutocorrItem *__restrict__ items_4096, uint32_t count_4096, const SrlaAutocorrItem *__restrict__ items_2048, uint32_t count_2048)
{
    const uint32_t g4 = 8u * ((count_4096 + 7u) >> 3);
    if (blockIdx.x < g4)
        autocorr_item<2, 256, false, 4096, WP>(jp, input, items, geoms, twiddles, fft_bytes, pass, results, lags_ws, dbg, items_4096, count_4096, nullptr, nullptr, blockIdx.x);
    else
        autocorr_item<1, 256, false, 2048, WP>(jp, input, items, geoms, twiddles, fft_bytes, pass, results, lags_ws, dbg, items_2048, count_2048, nullptr, nullptr, blockIdx.x - g4);
}

/* ================================================================================================
 * K2p: srla_pitch_solve -- one lane per item (lpc.c:1473-1649, srla_encoder.c:1031-1047)
 * ============================================================================================== */
/* Near-ties (H2): an item whose decision hangs on a libm function the device cannot reproduce bit for bit is appended to
 * the job's tie list -- ties[0] = count, ties[1 + k] = item | kind << 30 (kind 0: LPC order, 1: LTP taps, 2: SVR refinement) -- and, for LTP items, the
 * numbers the host needs to redo the 3x3 solve with its own pow() go to tie_data[8 k ..]. */
__device__ __forceinline__ uint32_t tie_append(uint32_t *__restrict__ ties, uint32_t item, uint32_t kind)
{
    const uint32_t k = atomicAdd(&ties[0], 1u);
    ties[1u + k] = item | (kind << 30);
    return k;
}

/* One step of the pitch scan (lpc.c:1486-1527) as a state machine over the lag index j = 8 .. 263, so that every lane of a
 * wavefront walks the lags in the same order with loads that do not depend on the data.  The reference's loops: from i, the
 * first upward zero crossing `start` (262 when there is none); from start + 1 the first downward one `end` (at most 261, or
 * start + 1 when that is larger); the largest local maximum above zero of [start, end] is a candidate; on with i = end + 1
 * while i < 262 and fewer than 20 candidates.  rm, rc, rn = R(j - 1), R(j), R(j + 1); returns true when a candidate is
 * complete (value *cand_val at *cand_at). */
struct PitchScan {
    uint32_t start, peak_at, ncand;
    double peak;
    bool in_seg, done;
};
__device__ __forceinline__ bool pitch_scan_step(PitchScan &st, const uint32_t j, const double rm, const double rc, const double rn,
                                                uint32_t *cand_at, double *cand_val)
{
    bool cand = false;
    if (!st.done) {
        if (!st.in_seg) {
            /* (j == 262: no crossing in [i, 261], the reference goes on with start = 262, end = 263) */
            if (j >= SRLA_LTP_MAX_PERIOD || (rm < 0.0 && rc > 0.0)) { st.start = j; st.in_seg = true; st.peak = 0.0; st.peak_at = 0; }
        }
        if (st.in_seg) {
            if (rc > rm && rc > rn && rc > st.peak) { st.peak = rc; st.peak_at = j; }
            if (j > st.start && (j >= SRLA_LTP_MAX_PERIOD - 1u || (rc > 0.0 && rn < 0.0))) {
                if (st.peak_at != 0) { cand = true; *cand_at = st.peak_at; *cand_val = st.peak; st.ncand++; }
                st.in_seg = false;
                if (j + 1u >= SRLA_LTP_MAX_PERIOD || st.ncand >= 20u) st.done = true;
            }
        }
    }
    return cand;
}

__global__ __launch_bounds__(WAVE) void srla_pitch_solve(SrlaJobParams jp, const SrlaItemDesc *__restrict__ items,
                                                         const double *__restrict__ lags_ws,
                                                         SrlaItemResult *__restrict__ results,
                                                         const uint32_t *__restrict__ select, uint32_t round,
                                                         uint32_t *__restrict__ ties, double *__restrict__ tie_data)
{
    NARROW_KERNEL_PRIORITY();
    /* One LANE per item, 64 items per wavefront.  The lag table is [lag][item], so the lanes of a wavefront read one lag of
     * their 64 items with one coalesced load, and because the scan visits the lags in a fixed order (pitch_scan_step) the
     * loads run ahead of the arithmetic instead of forming a chain of data-dependent round trips (round 2: the lags of 8 items
     * staged in LDS and scanned by 8 lanes: 0.4 ms per job alone, 0.77 ms in flight at -V 2 -P 3).  Two passes over the lags:
     * the first finds the largest candidate peak, the second the first candidate within 0.9 of it (lpc.c:1540-1546) -- no
     * candidate list, whose dynamic indexing would live in scratch memory. */
    const size_t stride = jp.num_items;
    const uint32_t idx = blockIdx.x * WAVE + threadIdx.x;
    if (idx >= jp.num_items) return;
    if (select != nullptr && select[idx] != round) return;   /* chain mode: only the items whose LTP lags this round produced */
    const double *lg = lags_ws + idx;
    /* words 263 and 264 of the reference's lag buffer are never written: zero (fresh pages) */
    auto R = [&](uint32_t j) -> double { return (j < SRLA_LTP_LAGS) ? lg[(size_t)j * stride] : 0.0; };
    SrlaItemResult *out = &results[idx];
    const double r0 = R(0);
    uint32_t period = 0;
    if (!(fabs(r0) <= (double)FLT_MIN)) {
        double best = 0.0;
        uint32_t ncand = 0;
        for (int pass = 0; pass < 2; pass++) {
            PitchScan st;
            st.start = 0; st.peak_at = 0; st.ncand = 0; st.peak = 0.0; st.in_seg = false; st.done = false;
            double rm = R(SRLA_LTP_MIN_PERIOD - 1u), rc = R(SRLA_LTP_MIN_PERIOD);
            bool found = false;
            /* j = 8 .. 263 in blocks of 8: the eight loads of a block are issued together */
            for (uint32_t j0 = SRLA_LTP_MIN_PERIOD; j0 < SRLA_LTP_MAX_PERIOD + 2u; j0 += 8u) {
                double nx[8];
#pragma unroll
                for (int u = 0; u < 8; u++) nx[u] = R(j0 + 1u + (uint32_t)u);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    uint32_t at = 0; double val = 0.0;
                    if (pitch_scan_step(st, j0 + (uint32_t)u, rm, rc, nx[u], &at, &val)) {
                        if (pass == 0) { if (val > best) best = val; }
                        else if (!found && val >= 0.9 * best) { period = at; found = true; }
                    }
                    rm = rc; rc = nx[u];
                }
                /* (a wavefront leaves the loop when all of its lanes are done) */
                if (st.done || found) break;
            }
            if (pass == 0) {
                ncand = st.ncand;
                if (ncand == 0 || best < 0.1 * r0) break;          /* lpc.c:1530-1537 */
            }
        }
        if (period < (jp.ltp_order / 2) + 1) period = 0;
    }
    uint32_t flags = 0;
    int32_t q[3] = { 0, 0, 0 };
    if (period > 0) {
        const int dim = (int)jp.ltp_order;
        const double rl[3] = { r0 * (1.0 + 1e-5), R(1), R(2) };
        double am[3][3], inv_diag[3], xs[3];
        bool ok = true;
        for (int j = 0; j < dim; j++) for (int k = j; k < dim; k++) am[j][k] = am[k][j] = rl[k - j];
        for (int i2 = 0; i2 < dim && ok; i2++) {
            double sum = am[i2][i2];
            for (int k = i2 - 1; k >= 0; k--) sum -= am[i2][k] * am[i2][k];
            if (sum <= 0.0) { ok = false; break; }
            inv_diag[i2] = inv_sqrt_cr(sum);                     /* lpc.c:591: pow(sum, -0.5) of the platform libm */
            for (int j = i2 + 1; j < dim; j++) {
                sum = am[i2][j];
                for (int k = i2 - 1; k >= 0; k--) sum -= am[i2][k] * am[j][k];
                am[j][i2] = sum * inv_diag[i2];
            }
        }
        if (!ok) { flags |= SRLA_ITEM_LTP_FAIL; period = 0; }
        else {
            double b[3] = { 0.0, 0.0, 0.0 };
            for (int i2 = 0; i2 < dim; i2++) {
                const uint32_t j = period - jp.ltp_order / 2 + (uint32_t)i2;
                b[i2] = (j == 0) ? rl[0] : R(j);
            }
            for (int i2 = 0; i2 < dim; i2++) {
                double sum = b[i2];
                for (int j = i2 - 1; j >= 0; j--) sum -= am[i2][j] * xs[j];
                xs[i2] = sum * inv_diag[i2];
            }
            for (int i2 = dim - 1; i2 >= 0; i2--) {
                double sum = xs[i2];
                for (int j = i2 + 1; j < dim; j++) sum -= am[j][i2] * xs[j];
                xs[i2] = sum * inv_diag[i2];
            }
            for (int i2 = 0; i2 < dim; i2++) {
                const double scaled = xs[i2] * 32.0 + jp.tie_ltpbias;   /* bias: 0.0 in production (tie tests) */
                const double fr = fabs(scaled) + 0.5;
                if (fabs(fr - floor(fr + 0.5)) < jp.tie_ltp && fabs(scaled) < 40.0) flags |= SRLA_ITEM_LTP_TIE;
                int32_t c = cvt_i32_as_x86(round_half_away(scaled));
                c = (c < -32) ? -32 : ((c > 31) ? 31 : c);
                q[i2] = c;
            }
            for (int i2 = 0; i2 < dim / 2; i2++) { const int32_t t = q[i2]; q[i2] = q[dim - 1 - i2]; q[dim - 1 - i2] = t; }
            const uint32_t forced = items[idx].forced_ltp;
            if (forced >> 31) {
                /* the host has redone the solve with its libm (host_ties.cpp) */
                for (int i2 = 0; i2 < 3; i2++) q[i2] = ((int32_t)((forced >> (6 * i2)) << 26)) >> 26;
                flags &= ~SRLA_ITEM_LTP_TIE;
            } else if ((flags & SRLA_ITEM_LTP_TIE) && ties != nullptr) {
                const uint32_t k = tie_append(ties, idx, 1u);
                double *td = tie_data + 8u * (size_t)k;
                td[0] = r0; td[1] = R(1); td[2] = R(2);
                td[3] = R(period - 1); td[4] = R(period); td[5] = R(period + 1);
                td[6] = (double)period;
                td[7] = (double)(((uint32_t)q[0] & 63u) | (((uint32_t)q[1] & 63u) << 6) | (((uint32_t)q[2] & 63u) << 12));
            }
        }
    }
    out->ltp_period = period;
    out->ltp_coef[0] = (period > 0) ? q[0] : 0;
    out->ltp_coef[1] = (period > 0) ? q[1] : 0;
    out->ltp_coef[2] = (period > 0) ? q[2] : 0;
    if (flags) out->flags |= flags;
}

/* ================================================================================================
 * K2: Levinson-Durbin / order choice / quantiser.  Three kernels:
 *   srla_lpc_recursion<L>  one LANE per item: the full recursion (lpc.c:379-441) with the gamma dot product
 *                          summed in index order; a[] and r[] live in LDS column-major ([i][lane]) so the 64
 *                          recursions of a wavefront run without bank conflicts; loops are unrolled so that
 *                          several LDS loads are in flight per dependent add.  Writes the (uncompensated)
 *                          error variance of every order.
 *   srla_order_select      one WAVE per item, lane = order: window compensation (lpc.c:490-497), code-length
 *                          estimate (srla_encoder.c:934-957) and its first strict minimum -- fully parallel.
 *   srla_lpc_quantize<L>   one LANE per item: recursion up to the chosen order, 8-bit quantiser with error
 *                          feedback (lpc.c:1341-1405), tap order reversal (srla_encoder.c:1104-1108), Huffman
 *                          cost plain vs pair-summed (srla_encoder.c:1141-1174).
 * ============================================================================================== */
#define A_(i) a[(size_t)(i) * L + lane]
#define R_(i) r[(size_t)(i) * L + lane]

/* recursion up to `upto` (>= 1); err_out (may be null) receives the error variance of orders 1..upto at
 * err_out[order * stride]; on return A_(1..upto) is the predictor of order `upto` */
template <int L>
__device__ __forceinline__ void levinson_lane(double *a, double *r, uint32_t lane, double r0, uint32_t upto,
                                              double *err_out, size_t stride)
{
    const double a1 = -R_(1) / r0;
    A_(0) = 1.0; A_(1) = a1; A_(2) = 0.0;
    double e = r0 + R_(1) * a1;
    if (err_out) err_out[stride] = e;
    for (uint32_t k = 1; k < upto; k++) {
        /* gamma = sum_{i=0..k} a[i] * r[k+1-i], accumulated in index order (lpc.c:420-423) */
        double gamma = 0.0;
        uint32_t i = 0;
        for (; i + 4 <= k + 1; i += 4) {
            const double x0 = A_(i), x1 = A_(i + 1), x2 = A_(i + 2), x3 = A_(i + 3);
            const double y0 = R_(k + 1 - i), y1 = R_(k - i), y2 = R_(k - 1 - i), y3 = R_(k - 2 - i);
            const double p0 = x0 * y0, p1 = x1 * y1, p2 = x2 * y2, p3 = x3 * y3;
            gamma += p0; gamma += p1; gamma += p2; gamma += p3;
        }
        for (; i < k + 1; i++) gamma += A_(i) * R_(k + 1 - i);
        gamma /= -e;
        e = e * (1.0 - gamma * gamma);
        /* a'[i] = a[i] + gamma * a[k+1-i] for i = 0..k+1, pairwise in place (lpc.c:430-433) */
        uint32_t lo = 0, hi = k + 1;
        for (; lo + 1 < hi - 1; lo += 2, hi -= 2) {
            const double al0 = A_(lo), ah0 = A_(hi), al1 = A_(lo + 1), ah1 = A_(hi - 1);
            A_(lo) = al0 + gamma * ah0; A_(hi) = ah0 + gamma * al0;
            A_(lo + 1) = al1 + gamma * ah1; A_(hi - 1) = ah1 + gamma * al1;
        }
        for (; lo <= hi; lo++, hi--) {
            const double al = A_(lo), ah = A_(hi);
            A_(lo) = al + gamma * ah;
            if (lo != hi) A_(hi) = ah + gamma * al;
            if (hi == 0) break;
        }
        A_(k + 2) = 0.0;
        if (err_out) err_out[(size_t)(k + 1) * stride] = e;
    }
}

template <int L>
__global__ __launch_bounds__(WAVE) void srla_lpc_recursion(SrlaJobParams jp, const double *__restrict__ lags_ws,
                                                           double *__restrict__ err_ws, const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = blockIdx.x * L + lane;
    if (lane >= L || idx >= jp.num_items) return;   /* no barriers below: lanes are independent */
    if (sel != nullptr && sel[idx] != sel_round) return;   /* chain mode with SVR on: the solve chain round by round */
    const uint32_t p = jp.max_order;
    double *a = (double *)lds;                        /* a[i * L + lane], i < p + 2 */
    double *r = a + (size_t)(p + 2) * L;              /* r[i * L + lane], i < p + 1 */
    const size_t stride = jp.num_items;
    for (uint32_t i = 0; i <= p; i++) R_(i) = lags_ws[(size_t)i * stride + idx];
    const double r0 = R_(0) * (1.0 + 1e-5);          /* ridge, lpc.c:483 */
    double *err = err_ws + idx;
    err[0] = r0;
    if (fabs(r0) < (double)FLT_EPSILON) {
        for (uint32_t o = 1; o <= p; o++) err[(size_t)o * stride] = r0;   /* lpc.c:395-405 */
        return;
    }
    levinson_lane<L>(a, r, lane, r0, p, err, stride);
}

__global__ __launch_bounds__(WAVE) void srla_order_select(
    SrlaJobParams jp, const SrlaItemDesc *__restrict__ items, const SrlaGeom *__restrict__ geoms,
    const double *__restrict__ err_ws, SrlaItemResult *__restrict__ results, double *__restrict__ dbg,
    uint32_t *__restrict__ ties, const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    NARROW_KERNEL_PRIORITY();
    /* (neighbouring items share the 64-byte sectors of the [order][item] table: they go to the same XCD, hence the same L2 --
     * dealt round robin over the XCDs every sector was fetched from HBM eight times, 255 MB per launch at -V 2) */
    const uint32_t idx = xcd_position(blockIdx.x, jp.num_items), lane = threadIdx.x;
    if (idx >= jp.num_items) return;
    if (sel != nullptr && sel[idx] != sel_round) return;
    const SrlaItemDesc it = items[idx];
    const double comp = geoms[it.geom].welch_comp;
    const uint32_t p = jp.max_order, n = it.n, bps = jp.bits_per_sample;
    const size_t stride = jp.num_items;
    double *dbg_item = dbg ? dbg + (size_t)idx * SRLA_DBG_STRIDE : nullptr;
    if (dbg_item && lane == 0) dbg_item[SRLA_DBG_ERRVARS] = err_ws[idx] * comp;
    /* first strict minimum == the lowest order among the smallest lengths; lengths that are NaN or not
     * below FLT_MAX are never chosen (srla_encoder.c:938-950) */
    const double kInf = __builtin_inf();
    double best = kInf, second = kInf;
    uint32_t best_order = 0;
    for (uint32_t base = 1; base <= p; base += WAVE) {
        const uint32_t o = base + lane;
        double len = kInf;
        if (o <= p) {
            const double ev = err_ws[(size_t)o * stride + idx] * comp;          /* lpc.c:490-497 */
            const double mabse = 2.0 * sqrt(ev / 2.0);
            double l = geometric_entropy(mabse, bps, jp.tie_logscale) * (double)n;
            l += (double)(8u * o);
            if (dbg_item) { dbg_item[SRLA_DBG_ERRVARS + o] = ev; dbg_item[SRLA_DBG_LENS + o] = l; }
            if (l < (double)FLT_MAX) len = l;
        }
        /* wave reduction of (len, order) in lexicographic order, plus the runner-up length */
        double v = len, v2 = kInf; uint32_t vo = o;
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(v, off, WAVE), ov2 = __shfl_xor(v2, off, WAVE);
            const uint32_t oo = __shfl_xor(vo, off, WAVE);
            const bool other_wins = (ov < v) || (ov == v && oo < vo);
            const double loser = other_wins ? v : ov;
            double s2 = (ov2 < v2) ? ov2 : v2;
            s2 = (loser < s2) ? loser : s2;
            if (other_wins) { v = ov; vo = oo; }
            v2 = s2;
        }
        if (v < best) { second = (best < v2) ? best : v2; best = v; best_order = vo; }
        else { const double c = (v < v2) ? v : v2; second = (c < second) ? c : second; }
    }
    if (lane == 0) {
        uint32_t order = (best < kInf) ? best_order : 0u;
        uint32_t flags = 0;
        if (jp.order_fixed) order = p;
        else if (it.forced_order < 0 && order != 0 && (second - best) <= jp.tie_rel * fabs(best) + 1e-9) {
            flags |= SRLA_ITEM_ORDER_TIE;
            if (ties) (void)tie_append(ties, idx, 0u);
        }
        if (it.forced_order >= 0) order = (uint32_t)it.forced_order;
        results[idx].lpc_order = order;
        if (flags) results[idx].flags |= flags;
    }
}

/* shared tail of the quantiser kernels: cf(i) = tap i of the chosen predictor */
template <typename CF, typename QS, typename QL>
__device__ __forceinline__ void quantize_and_price(uint32_t order, bool silent, CF cf, QS qstore, QL qload,
                                                   const uint8_t *__restrict__ huff_len, SrlaItemResult *out,
                                                   const double band = 0.0, bool *near_boundary = nullptr)
{
    /* band > 0: *near_boundary is set when the outcome hangs on the last bits of a tap -- the largest tap within `band` (relative) of a
     * power of two (the shared shift), or a scaled tap plus the error fed back within `band` of a rounding boundary */
    uint32_t rshift = 0, use_sum = 0, coef_bits = 0;
    if (order > 0) {
        /* 8-bit quantisation with error feedback from the last tap (lpc.c:1341-1405) */
        double maxabs = 0.0;
        if (!silent) for (uint32_t i = 0; i < order; i++) { const double v = fabs(cf(i)); if (maxabs < v) maxabs = v; }
        if (maxabs <= 0.0078125) {
            rshift = 8;
            for (uint32_t i = 0; i < order; i++) qstore(i, 0);
        } else {
            int ndigit;
            const double mant = frexp(maxabs, &ndigit);
            rshift = (uint32_t)(7 - ndigit);
            if (rshift >= 16u) rshift = 15u;
            const double scale = __builtin_ldexp(1.0, (int)rshift);
            bool near = band > 0.0 && (mant - 0.5 < band || 1.0 - mant < band);
            double qerr = 0.0;
            for (int i = (int)order - 1; i >= 0; i--) {
                qerr += cf((uint32_t)i) * scale;
                if (band > 0.0) { const double a = fabs(qerr), fr = a - floor(a); if (fabs(fr - 0.5) < band) near = true; }
                int32_t qq = cvt_i32_as_x86(round_half_away(qerr));
                if (qq >= 128) qq = 127; else if (qq < -128) qq = -128;
                qerr -= (double)qq;
                qstore(order - 1 - (uint32_t)i, qq);        /* reversed: oldest sample first (srla_encoder.c:1104) */
            }
            if (near_boundary != nullptr && near) *near_boundary = true;
        }
        /* Huffman cost plain vs pair-summed (srla_encoder.c:1141-1174) */
        uint32_t plain = 0, summed = 0, overflow = 0;
        int32_t prevq = 0;
        for (uint32_t k = 0; k < order; k++) {
            const int32_t c = qload(k);
            out->lpc_coef[k] = (int8_t)c;
            plain += huff_len[zigzag32(c)];
            if (k == 0) summed += huff_len[zigzag32(c)];
            else {
                const uint32_t z = zigzag32(c + prevq);
                if (z >= 256u) overflow = 1; else summed += huff_len[256 + z];
            }
            prevq = c;
        }
        use_sum = (overflow == 0 && (order == 1 || summed < plain)) ? 1u : 0u;
        coef_bits = use_sum ? summed : plain;
    }
    out->lpc_rshift = rshift;
    out->use_sum = use_sum;
    out->pad[0] = coef_bits;
}

/* The whole solve chain of one item in ONE pass, one LANE per item, everything but the snapshot in registers (orders 8, 16,
 * 32, 64): the recursion (lags r[] and predictor a[] in VGPRs, static indices), and after every step the code-length
 * estimate of that order (it needs only the step's error variance, srla_encoder.c:940-950) -- whenever the estimate
 * improves, i.e. at every order the reference's sequential `minlen > len` scan would adopt, the predictor is copied to LDS.
 * What is in LDS at the end is the predictor of the chosen order: no second recursion, no separate order-selection launch.
 * Then the 8-bit quantiser and the tap cost, with the static Huffman lengths in LDS. */
template <int P>
__global__ __launch_bounds__(WAVE) void srla_lpc_solve_regs(
    SrlaJobParams jp, const SrlaItemDesc *__restrict__ items, const SrlaGeom *__restrict__ geoms,
    const double *__restrict__ lags_ws, double *__restrict__ err_ws, const uint8_t *__restrict__ huff_len,
    SrlaItemResult *__restrict__ results, double *__restrict__ dbg, uint32_t *__restrict__ ties,
    double *__restrict__ coef_ws /* SVR refinement follows: the predictor of the chosen order goes here (row of P per item), unquantised */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int L = WAVE;
    const uint32_t lane = threadIdx.x;
    uint8_t *s_huff = lds;                                   /* 512 bytes: plain, pair-summed code lengths */
    double *snap = (double *)(lds + 512);                    /* snap[i * L + lane], i < P: taps of the best order so far */
    int32_t *q = (int32_t *)(snap + (size_t)P * L);          /* q[i * L + lane]: quantised taps */
    for (uint32_t i = lane; i < 128; i += WAVE) ((uint32_t *)s_huff)[i] = ((const uint32_t *)huff_len)[i];
    __syncthreads();
    const uint32_t idx = blockIdx.x * WAVE + lane;
    if (idx >= jp.num_items) return;                         /* no barriers below: lanes are independent */
    const size_t stride = jp.num_items;
    double r[P + 1];
#pragma unroll
    for (int i = 0; i <= P; i++) r[i] = lags_ws[(size_t)i * stride + idx];
    const SrlaItemDesc it = items[idx];
    const double comp = geoms[it.geom].welch_comp;
    const uint32_t n = it.n, bps = jp.bits_per_sample;
    const int32_t forced = it.forced_order;
    const double r0 = r[0] * (1.0 + 1e-5);                   /* ridge, lpc.c:483 */
    double *err = err_ws + idx;
    err[0] = r0;
    double *dbg_item = dbg ? dbg + (size_t)idx * SRLA_DBG_STRIDE : nullptr;
    if (dbg_item) dbg_item[SRLA_DBG_ERRVARS] = r0 * comp;
    const bool silent = fabs(r0) < (double)FLT_EPSILON;

    /* srla_encoder.c:934-957: minlen = FLT_MAX; for order = 1..p: if (minlen > len) adopt -- literally, plus the runner-up
     * length for the near-tie test */
    double best = (double)FLT_MAX, second = __builtin_inf();
    uint32_t best_order = 0;
    auto consider = [&](uint32_t order, double e) -> bool {
        const double ev = e * comp;                                              /* lpc.c:490-497 */
        const double mabse = 2.0 * sqrt(ev / 2.0);
        double l = geometric_entropy(mabse, bps, jp.tie_logscale) * (double)n;
        l += (double)(8u * order);
        if (dbg_item) { dbg_item[SRLA_DBG_ERRVARS + order] = ev; dbg_item[SRLA_DBG_LENS + order] = l; }
        const bool better = best > l;
        if (better) { second = best; best = l; }
        else if (l < second) second = l;
        const bool take = (forced >= 0) ? (order == (uint32_t)forced) : better;
        if (take) best_order = order;
        return take;
    };

    if (silent) {
        /* lpc.c:395-405: every error variance is r0, every predictor zero */
        for (uint32_t o = 1; o <= (uint32_t)P; o++) { err[(size_t)o * stride] = r0; (void)consider(o, r0); }
    } else {
        double a[P + 2];
        const double a1 = -r[1] / r0;
        a[0] = 1.0; a[1] = a1; a[2] = 0.0;
        double e = r0 + r[1] * a1;
        err[stride] = e;
        if (consider(1u, e)) snap[lane] = a[1];
#pragma unroll
        for (int k = 1; k < P; k++) {
            double gamma = 0.0;
#pragma unroll
            for (int i = 0; i <= k; i++) gamma += a[i] * r[k + 1 - i];          /* index order, lpc.c:420-423 */
            gamma /= -e;
            e = e * (1.0 - gamma * gamma);
#pragma unroll
            for (int i = 0; i <= (k + 1) / 2; i++) {
                const int j = k + 1 - i;
                const double ai = a[i], aj = a[j];
                a[i] = ai + gamma * aj;
                if (i != j) a[j] = aj + gamma * ai;
            }
            a[k + 2] = 0.0;
            err[(size_t)(k + 1) * stride] = e;
            if (consider((uint32_t)(k + 1), e)) {
#pragma unroll
                for (int i = 0; i <= k; i++) snap[(size_t)i * L + lane] = a[1 + i];
            }
        }
    }
    uint32_t order = best_order;
    uint32_t flags = 0;
    if (jp.order_fixed) order = (uint32_t)P;
    else if (forced < 0 && order != 0 && (second - best) <= jp.tie_rel * fabs(best) + 1e-9) {
        flags |= SRLA_ITEM_ORDER_TIE;
        if (ties) (void)tie_append(ties, idx, 0u);
    }
    SrlaItemResult *out = &results[idx];
    out->lpc_order = order;
    if (flags) out->flags |= flags;
    if (coef_ws != nullptr) {
        double *row = coef_ws + (size_t)idx * 64u;             /* SVR_P doubles per item whatever the preset */
        for (uint32_t i = 0; i < order; i++) row[i] = silent ? 0.0 : snap[(size_t)i * L + lane];
        return;
    }
    quantize_and_price(order, silent,
                       [&](uint32_t i) -> double { return snap[(size_t)i * L + lane]; },
                       [&](uint32_t i, int32_t v) { q[(size_t)i * L + lane] = v; },
                       [&](uint32_t i) -> int32_t { return q[(size_t)i * L + lane]; }, s_huff, out);
}

/* The same chain as three launches, each with the parallelism its part has (the default for orders 8 .. 64):
 *   srla_lpc_errvars<P>   one LANE per item, registers: the recursion alone -- error variance of every order (err_ws) and the
 *                         reflection coefficient of every step (gamma_ws; row 0 holds a[1] of order 1).
 *   srla_order_select     one WAVE per item, lane = order: the 64 code-length estimates (a square root, two divisions and two
 *                         logarithms each -- more than half of the one-pass kernel's instructions, and there on the serial
 *                         chain of a single lane) side by side.
 *   srla_lpc_taps<P>      one LANE per item: the predictor of the chosen order rebuilt from the stored reflection coefficients
 *                         (lpc.c:430-433, the update alone: the same products and sums on the same operands, no dot
 *                         products), 8-bit quantiser, tap cost.
 * srla_lpc_solve_regs holds a whole SIMD's registers for 0.11 ms on its own and 0.17-0.24 ms beside the wide kernels (206
 * wavefronts of 262 VGPRs at -V 1); the three together hold far less for far shorter. */
template <int P>
__global__ __launch_bounds__(WAVE) void srla_lpc_errvars(SrlaJobParams jp, const double *__restrict__ lags_ws, double *__restrict__ err_ws,
                                                         double *__restrict__ gamma_ws, const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    NARROW_KERNEL_PRIORITY();
    const uint32_t idx = blockIdx.x * WAVE + threadIdx.x;
    if (idx >= jp.num_items) return;
    if (sel != nullptr && sel[idx] != sel_round) return;
    const size_t stride = jp.num_items;
    double r[P + 1];
#pragma unroll
    for (int i = 0; i <= P; i++) r[i] = lags_ws[(size_t)i * stride + idx];
    const double r0 = r[0] * (1.0 + 1e-5);                   /* ridge, lpc.c:483 */
    double *err = err_ws + idx, *gam = gamma_ws + idx;
    err[0] = r0;
    if (fabs(r0) < (double)FLT_EPSILON) {
        /* lpc.c:395-405: every error variance is r0, every predictor zero */
        for (uint32_t o = 1; o <= (uint32_t)P; o++) { err[(size_t)o * stride] = r0; gam[(size_t)(o - 1) * stride] = 0.0; }
        return;
    }
    double a[P + 2];
    const double a1 = -r[1] / r0;
    a[0] = 1.0; a[1] = a1; a[2] = 0.0;
    double e = r0 + r[1] * a1;
    err[stride] = e;
    gam[0] = a1;
#pragma unroll
    for (int k = 1; k < P; k++) {
        double gamma = 0.0;
#pragma unroll
        for (int i = 0; i <= k; i++) gamma += a[i] * r[k + 1 - i];          /* index order, lpc.c:420-423 */
        gamma /= -e;
        e = e * (1.0 - gamma * gamma);
#pragma unroll
        for (int i = 0; i <= (k + 1) / 2; i++) {
            const int j = k + 1 - i;
            const double ai = a[i], aj = a[j];
            a[i] = ai + gamma * aj;
            if (i != j) a[j] = aj + gamma * ai;
        }
        a[k + 2] = 0.0;
        err[(size_t)(k + 1) * stride] = e;
        gam[(size_t)k * stride] = gamma;
    }
}

/* srla_lpc_errvars with a footprint that fits beside the wide kernels.  The register form of order 64 holds 348 registers per
 * lane -- two thirds of a SIMD's file.  Beside srla_residual_cost (five wavefronts of 96 registers per SIMD, a fresh workgroup
 * taking every slot that frees up) such a wavefront finds no room until the wide launch drains: in a kernel trace of a 600 s
 * encode srla_lpc_errvars<64> took 31 us when it started just ahead of a wide kernel and 230-470 us otherwise, it ended exactly
 * where a wide kernel ended, stream N was busy back to back and srla_residual_cost of the job waited for it (0.5 ms of gaps on
 * stream W per call).  Here the lags r[] and the upper part of the predictor a[] stand in LDS ([index][lane]: conflict-free,
 * addresses are immediates because everything stays unrolled), AREG entries of a[] in registers and L = 32 items share a
 * wavefront: about 120 registers and 23 KB of LDS -- what ONE retiring workgroup of a wide kernel leaves behind.  The same
 * operations in the same order on the same operands (lpc.c:417-438): identical bits. */
template <int P, int L, int AREG>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4, 8))) /* at most 128 registers */ void srla_lpc_errvars_lean(SrlaJobParams jp, const double *__restrict__ lags_ws, double *__restrict__ err_ws,
                                                              double *__restrict__ gamma_ws, const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    NARROW_KERNEL_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = blockIdx.x * L + lane;
    if (lane >= (uint32_t)L || idx >= jp.num_items) return;   /* no barriers below: lanes are independent */
    if (sel != nullptr && sel[idx] != sel_round) return;
    const size_t stride = jp.num_items;
    double *rl = reinterpret_cast<double *>(lds) + lane;      /* r[k] at rl[k * L] */
    double *al = rl + (size_t)(P + 1) * L;                    /* a[i], i >= AREG, at al[(i - AREG) * L] */
#pragma unroll
    for (int i = 0; i <= P; i++) rl[i * L] = lags_ws[(size_t)i * stride + idx];
    const double r0 = rl[0] * (1.0 + 1e-5);                   /* ridge, lpc.c:483 */
    double *err = err_ws + idx, *gam = gamma_ws + idx;
    err[0] = r0;
    if (fabs(r0) < (double)FLT_EPSILON) {
        /* lpc.c:395-405: every error variance is r0, every predictor zero */
        for (uint32_t o = 1; o <= (uint32_t)P; o++) { err[(size_t)o * stride] = r0; gam[(size_t)(o - 1) * stride] = 0.0; }
        return;
    }
    double areg[AREG];
    /* i is a constant wherever these are called (the loops below are fully unrolled), so the choice folds away */
    auto A = [&](int i) -> double { return (i < AREG) ? areg[i < AREG ? i : 0] : al[(i - AREG) * L]; };
    auto setA = [&](int i, double v) { if (i < AREG) areg[i < AREG ? i : 0] = v; else al[(i - AREG) * L] = v; };
    const double r1 = rl[L];
    const double a1 = -r1 / r0;
    setA(0, 1.0); setA(1, a1); setA(2, 0.0);
    double e = r0 + r1 * a1;
    err[stride] = e;
    gam[0] = a1;
#pragma unroll
    for (int k = 1; k < P; k++) {
        /* (the fences keep the scheduler from hoisting a whole step's LDS loads to its top: chunks of eight terms, whose loads
         * are in flight together, stay within the register budget) */
        double gamma = 0.0;
#pragma unroll
        for (int i = 0; i <= k; i++) {
            gamma += A(i) * rl[(k + 1 - i) * L];                               /* index order, lpc.c:420-423 */
            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        gamma /= -e;
        e = e * (1.0 - gamma * gamma);
#pragma unroll
        for (int i = 0; i <= (k + 1) / 2; i++) {
            const int j = k + 1 - i;
            const double ai = A(i), aj = A(j);
            setA(i, ai + gamma * aj);
            if (i != j) setA(j, aj + gamma * ai);
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        setA(k + 2, 0.0);
        __builtin_amdgcn_sched_barrier(0);
        err[(size_t)(k + 1) * stride] = e;
        gam[(size_t)k * stride] = gamma;
    }
}

template <int P>
__global__ __launch_bounds__(WAVE) void srla_lpc_taps(SrlaJobParams jp, const double *__restrict__ err_ws, const double *__restrict__ gamma_ws,
                                                      const uint8_t *__restrict__ huff_len, SrlaItemResult *__restrict__ results,
                                                      double *__restrict__ coef_ws /* SVR refinement follows: the predictor of the chosen order goes here (row of 64 per item), unquantised */,
                                                      const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    NARROW_KERNEL_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int L = WAVE;
    const uint32_t lane = threadIdx.x;
    uint8_t *s_huff = lds;                                   /* 512 bytes: plain, pair-summed code lengths */
    double *snap = (double *)(lds + 512);                    /* snap[i * L + lane], i < P: taps of the chosen order */
    int32_t *q = (int32_t *)(snap + (size_t)P * L);          /* q[i * L + lane]: quantised taps */
    for (uint32_t i = lane; i < 128; i += WAVE) ((uint32_t *)s_huff)[i] = ((const uint32_t *)huff_len)[i];
    __syncthreads();
    const uint32_t idx = blockIdx.x * WAVE + lane;
    if (idx >= jp.num_items) return;                         /* no barriers below: lanes are independent */
    if (sel != nullptr && sel[idx] != sel_round) return;
    const size_t stride = jp.num_items;
    SrlaItemResult *out = &results[idx];
    const uint32_t order = out->lpc_order;
    const bool silent = fabs(err_ws[idx]) < (double)FLT_EPSILON;
    const double *gam = gamma_ws + idx;
    /* every reflection coefficient the wavefront can need, fetched at once (one coalesced load per step, all in flight together:
     * fetched step by step inside the branch below they were 63 dependent round trips, 50 us of the launch) */
    uint32_t top = order;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o2 = (uint32_t)__shfl_xor((int)top, off, WAVE); top = (o2 > top) ? o2 : top; }
    double g[P];
#pragma unroll
    for (int k = 0; k < P; k++) g[k] = ((uint32_t)k < top) ? gam[(size_t)k * stride] : 0.0;
    double a[P + 2];
    a[0] = 1.0; a[1] = g[0]; a[2] = 0.0;
#pragma unroll
    for (int k = 1; k < P; k++) {
        if ((uint32_t)k < order) {                           /* (a wavefront goes as far as the highest order among its items) */
            const double gamma = g[k];
#pragma unroll
            for (int i = 0; i <= (k + 1) / 2; i++) {
                const int j = k + 1 - i;
                const double ai = a[i], aj = a[j];
                a[i] = ai + gamma * aj;
                if (i != j) a[j] = aj + gamma * ai;
            }
        }
        a[k + 2] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < P; i++) snap[(size_t)i * L + lane] = a[1 + i];
    if (coef_ws != nullptr) {
        double *row = coef_ws + (size_t)idx * 64u;             /* SVR_P doubles per item whatever the preset */
        for (uint32_t i = 0; i < order; i++) row[i] = silent ? 0.0 : snap[(size_t)i * L + lane];
        return;
    }
    quantize_and_price(order, silent,
                       [&](uint32_t i) -> double { return snap[(size_t)i * L + lane]; },
                       [&](uint32_t i, int32_t v) { q[(size_t)i * L + lane] = v; },
                       [&](uint32_t i) -> int32_t { return q[(size_t)i * L + lane]; }, s_huff, out);
}

template <int L>
__global__ __launch_bounds__(WAVE) void srla_lpc_quantize(
    SrlaJobParams jp, const double *__restrict__ lags_ws, const uint8_t *__restrict__ huff_len,
    SrlaItemResult *__restrict__ results, double *__restrict__ coef_ws /* SVR refinement follows: the taps of the chosen order, unquantised, rows of 256 */,
    const uint32_t *__restrict__ sel, uint32_t sel_round)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t idx = blockIdx.x * L + lane;
    if (lane >= L || idx >= jp.num_items) return;
    if (sel != nullptr && sel[idx] != sel_round) return;
    const uint32_t p = jp.max_order;
    double *a = (double *)lds;
    double *r = a + (size_t)(p + 2) * L;
    const size_t stride = jp.num_items;
    SrlaItemResult *out = &results[idx];
    const uint32_t order = out->lpc_order;
    uint32_t rshift = 0, use_sum = 0, coef_bits = 0;
    if (order > 0) {
        for (uint32_t i = 0; i <= order; i++) R_(i) = lags_ws[(size_t)i * stride + idx];
        const double r0 = R_(0) * (1.0 + 1e-5);
        const bool silent = fabs(r0) < (double)FLT_EPSILON;
        if (!silent) levinson_lane<L>(a, r, lane, r0, order, nullptr, 0);
        if (coef_ws != nullptr) {
            double *row = coef_ws + (size_t)idx * 256u;
            for (uint32_t i = 0; i < order; i++) row[i] = silent ? 0.0 : A_(1 + i);
            return;
        }
        /* 8-bit quantisation with error feedback from the last tap (lpc.c:1341-1405); q[] reuses r[] */
        int32_t *q = (int32_t *)r;
#define Q_(i) q[(size_t)(i) * (2 * L) + lane]
        double maxabs = 0.0;
        if (!silent) for (uint32_t i = 0; i < order; i++) { const double v = fabs(A_(1 + i)); if (maxabs < v) maxabs = v; }
        if (maxabs <= 0.0078125) {
            rshift = 8;
            for (uint32_t i = 0; i < order; i++) Q_(i) = 0;
        } else {
            int ndigit;
            (void)frexp(maxabs, &ndigit);
            rshift = (uint32_t)(7 - ndigit);
            if (rshift >= 16u) rshift = 15u;
            const double scale = __builtin_ldexp(1.0, (int)rshift);
            double qerr = 0.0;
            for (int i = (int)order - 1; i >= 0; i--) {
                qerr += A_(1 + i) * scale;
                int32_t qq = cvt_i32_as_x86(round_half_away(qerr));
                if (qq >= 128) qq = 127; else if (qq < -128) qq = -128;
                qerr -= (double)qq;
                Q_(order - 1 - i) = qq;          /* reversed: oldest sample first (srla_encoder.c:1104) */
            }
        }
        /* Huffman cost plain vs pair-summed (srla_encoder.c:1141-1174) */
        uint32_t plain = 0, summed = 0, overflow = 0;
        int32_t prevq = 0;
        for (uint32_t k = 0; k < order; k++) {
            const int32_t c = Q_(k);
            out->lpc_coef[k] = (int8_t)c;
            plain += huff_len[zigzag32(c)];
            if (k == 0) summed += huff_len[zigzag32(c)];
            else {
                const uint32_t z = zigzag32(c + prevq);
                if (z >= 256u) overflow = 1; else summed += huff_len[256 + z];
            }
            prevq = c;
        }
        use_sum = (overflow == 0 && (order == 1 || summed < plain)) ? 1u : 0u;
        coef_bits = use_sum ? summed : plain;
#undef Q_
    }
    out->lpc_rshift = rshift;
    out->use_sum = use_sum;
    out->pad[0] = coef_bits;
}
#undef A_
#undef R_

/* ================================================================================================
 * K3: srla_residual_cost -- FIR residual + Rice code-length search, one workgroup per item
 * ============================================================================================== */
struct SmallC {
    int32_t  coefq[FIR_PAD + 8];   /* taps, front padded with zeros to a multiple of four */
    uint32_t level_bits[16];
    uint8_t  ktab[2048];           /* heap layout: level p at [2^p - 1, 2^(p+1) - 1) */
    uint32_t max_u;
    uint32_t pad[3];
};

extern "C" uint32_t srla_kernel_small_c_bytes(void) { return (uint32_t)((sizeof(SmallC) + 15) & ~15u); }

/* ---- fast path of K3 for blocks of 1024 * FL samples (FL = 1..4): every thread owns S = 4 * FL
 * CONTIGUOUS samples = four finest partitions of the 1024-way split, so the whole partition-mean tree
 * (srla_coder.c:366-389) lives in registers: levels 10..8 inside a thread, 7..2 by wave shuffles, 1..0
 * through four LDS words.  The residual never goes back to LDS; only the signal (for the FIR windows of
 * neighbouring threads) and the 2047-byte parameter table do.  LDS layout of the signal: four words of
 * padding after every S samples so that the 16-byte window loads of a wavefront are conflict free. */
static_assert(offsetof(SrlaItemResult, lpc_coef) % 4 == 0 && sizeof(SrlaItemResult) % 4 == 0, "the taps of an item record can be read as aligned words");
#define MF_PADB 256                 /* FIR_MFMA: zero bytes in front of every byte plane (>= the largest order rounded up to 16) */
#define MF_OFFZ 144                 /* ... index of tap 0 in the zero-padded tap string (>= 16 FL + 14 + 15 for FL <= 8) */
#define MF_TZB  544                 /* ... bytes of one copy of it: MF_OFFZ + 64 k-blocks' worth for order 255 (5 at FL <= 4) + a lane's reach */
struct SmallF {
    union {
        int32_t  coefq[FIR_PAD + 8];          /* FIR_MAD24 / FIR_WIDE: taps, front padded with zeros to a multiple of four */
        uint32_t cpack[2][FIR_PAD + 4];       /* FIR_DOT: per group of four taps the four coefficient words of the low plane
                                               * ([0]: int16 pairs) and of the high plane ([1]: int8 quads), then the closing words */
        uint8_t  tz[4][MF_TZB];               /* FIR_MFMA: the zero-padded tap string, four copies shifted by 0..3 bytes (see mfma_fir) */
    };
    uint32_t level_bits[16];
    uint8_t  ktab[2048];
    double   wave_mean[NWAVES];
    double   thr[32];                 /* Rice parameter thresholds (copy of the host table) */
    uint32_t wave_max[NWAVES];
    uint32_t wave_high[NWAVES];       /* FIR_DOT: does any sample of the wavefront leave 16 bits? */
};

/* LDS layout of the fast path's signal: every thread owns S = 4 FL consecutive samples.  For even FL a pad of four
 * words follows every S samples, for odd FL none: the distance between the threads' 16-byte accesses is then 4, 12, 12,
 * 20, 20, 28, 28, 36 words for FL = 1..8 -- never a multiple of 8, which would put every second or fourth lane on the
 * same banks (the 3072-sample class, S = 12, measured 40 % slower per sample than its neighbours with the pad: 16
 * words) -- and odd FL need no index arithmetic at all. */
template <int FL>
__device__ __forceinline__ uint32_t sig_index(int s_plus_pad)
{
    constexpr int S = 4 * FL;
    if constexpr (FL & 1) return (uint32_t)s_plus_pad;
    else return (uint32_t)(s_plus_pad + (s_plus_pad / S) * 4);
}

__device__ __forceinline__ uint32_t rice_param(double mean, uint32_t code_type, const double *thr)
{
    if (code_type == SRLA_CODE_RICE) {
        /* srla_coder.c:262-276 through the host-derived thresholds: k = #{t : mean >= thr[t]}; thr ascends
         * (unreachable entries are +inf), so a 5-step bisection plus one compare counts them */
        uint32_t k = 0;
#pragma unroll
        for (uint32_t step = 16; step > 0; step >>= 1) k += (mean >= thr[k + step - 1]) ? step : 0u;
        k += (k == 31u && mean >= thr[31]) ? 1u : 0u;
        return k;
    }
    const double gp = 0.66794162356 * (1.0 + mean);   /* srla_coder.c:298-311 */
    const uint32_t golomb = (uint32_t)((1.0 > gp) ? 1.0 : gp);
    return 31u - (uint32_t)__clz((int)golomb);
}

__device__ __forceinline__ uint32_t code_cost(uint32_t val, uint32_t k, uint32_t code_type)
{
    if (code_type == SRLA_CODE_RICE) return 1u + k + (val >> k);                 /* srla_coder.c:327-330 */
    /* srla_coder.c:333-347: k + 2 bits up to 2^(k+1), then one more bit per 2^k */
    return (k + 2u) + (__builtin_elementwise_sub_sat(val, 2u << k) >> k);
}

/* the part of code_cost that depends on the sample: sum it, add count * fixed(k) once */
__device__ __forceinline__ uint32_t code_cost_var(uint32_t val, uint32_t k, uint32_t code_type)
{
    /* one formula for both codes: Rice is the recursive code with threshold 0 (the threshold depends on k only, so it
     * leaves the sample loops; a select between the two forms would not) */
    const uint32_t thr = (code_type == SRLA_CODE_RICE) ? 0u : (2u << k);
    return __builtin_elementwise_sub_sat(val, thr) >> k;
}
__device__ __forceinline__ uint32_t code_cost_fixed(uint32_t k, uint32_t code_type)
{
    return (code_type == SRLA_CODE_RICE) ? (1u + k) : (k + 2u);
}

/* acc + a * b on the full-rate 24-bit multiplier (a, b within 24 bits signed) */
__device__ __forceinline__ uint32_t mad24(int32_t a, int32_t b, uint32_t acc)
{
    uint32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
}

/* The FIR of the fast path, three ways (all the reference's wrap-around int32 sum, srla_lpc_predict.c:118-265):
 * FIR_DOT   (input of at most 18 bits, the default): the signal x -- within 24 bits after M/S, pre-emphasis and the LTP -- is split
 *           exactly as x = 2^16 h + l with l = the sign-extended low half and h = (x - l) >> 16 (8 bits), and kept in LDS as an int16
 *           plane and an int8 plane.  sum c x = sum c l + 2^16 sum c h modulo 2^32, with the taps (8-bit) packed to match: the low
 *           plane costs one v_dot2_i32_i16 per TWO taps and sample, the high plane one v_dot4_i32_i8 per FOUR -- and h is zero
 *           wherever the signal stays within 16 bits (ordinary 16-bit audio below full scale), which the workgroup finds out while it
 *           packs the planes and then skips the high pass.  A thread's four outputs of a chunk lie at the four byte phases of the
 *           packed words, so the TAPS are packed in four phases (even / odd for the low plane) and the sample words are used as they
 *           lie; a tap pair that straddles two groups of four taps is closed by one more word after the loop.
 * FIR_MFMA  (round 5; blocks of at most 4096 samples, input of at most 18 bits): the FIR as a Toeplitz product on the MATRIX pipe
 *           (v_mfma_i32_16x16x64_i8), which is otherwise idle and issues beside the VALU.  The signal is split into signed byte
 *           digits x = s0 + 256 s1 + 65536 x2 (s0, s1 in [-128, 127]; x2 = 0 wherever x + 128 stays within 16 bits) kept as three
 *           byte planes; sum c x = sum c s0 + 2^8 sum c s1 + 2^16 sum c x2 modulo 2^32, every partial sum exact in the
 *           accumulators (|sum| <= 64 * 128 * 128 * k-blocks).  One product gives 16 x 16 outputs: column cc = the 64-byte window of
 *           the plane that starts 16 FL cc samples into the wavefront's run (minus the order rounded up to 16: 16-byte aligned
 *           loads, conflict free), row 4 g + i = output 4 FL g + 4 T + i of that window for tile T -- so that lane 16 g + cc ends up
 *           with chunk T of thread 4 cc + g: ONE fixed lane permutation (ds_bpermute) returns every chunk to its owner.  The tap
 *           matrix of a lane is 16 consecutive bytes of the zero-padded tap string at a byte offset that depends on the lane's row;
 *           four copies of the string shifted by 0..3 bytes make it four aligned words.  mfma_fir() below; tools/probes has the
 *           operand-layout probe and the numpy model the index arithmetic was checked with.
 * FIR_MAD24 (what FIR_DOT replaced; -DSRLA_FIR_MAD24): one v_mad_i32_i24 per tap and sample on int32 words.
 * FIR_WIDE  samples beyond 24 bits (24-bit input: M/S, pre-emphasis and the LTP widen it to 28): every tap multiplies
 *           the two 16-bit halves of the sample separately on the 24-bit multiplier (x c = (x >> 16) c 2^16 + (x & 0xffff) c
 *           modulo 2^32), which is still twice as fast as 32-bit multiplies */
#define FIR_MAD24 0
#define FIR_WIDE  1
#define FIR_DOT   2
#define FIR_MFMA  3
#ifdef SRLA_FIR_MAD24
#define SRLA_FIR_NARROW FIR_MAD24
#else
#define SRLA_FIR_NARROW FIR_DOT
#endif
/* LDS of ONE item of the fast path, by block length (fl = n / 1024) and the log2 of the items that share its workgroup: the
 * int32 signal (front padding for the FIR's and the LTP's reach back, pads between the threads' runs: sig_index), or only the
 * two planes that lie over it when neither the long-term predictor nor a FIR form other than FIR_DOT needs the int32 words;
 * then the small structure.  One definition for the kernel and for the host's launch size. */
__host__ __device__ constexpr uint32_t fast_sig_bytes(int fl, int lg, bool planes_only)
{
    const int ch = fl << lg, s = 4 * ch, padw = (ch & 1) ? 0 : 4;
    const int padmin = (FIR_PAD > SRLA_LTP_MAX_PERIOD + 2) ? FIR_PAD : (SRLA_LTP_MAX_PERIOD + 2);
    const int pads = ((padmin + s - 1) / s) * s, padf = ((FIR_PAD + s - 1) / s) * s;
    const uint32_t sig_words = (uint32_t)((pads + 1024 * fl) / s) * (uint32_t)(s + padw) + 8u;
    const uint32_t plane_elems = (uint32_t)((padf + 1024 * fl) / s) * (uint32_t)(s + padw);
    const uint32_t planes = ((plane_elems * 2u + 15u) & ~15u) + plane_elems;
    return ((planes_only ? planes : sig_words * 4u) + 15u) & ~15u;
}
/* Items per workgroup (log2).  -DSRLA_ITEM_GROUPS: blocks of 1024 and (without the long-term predictor) 2048 samples share a
 * workgroup two by two, so that a thread holds 8 or 16 samples instead of 4 or 8 and its fixed work in the Rice search is spread
 * over more of them.  Bit-identical (all GPU tests), and measured SLOWER: srla_residual_cost 0.196 -> 0.218 ms per job at M,
 * 0.503 -> 0.578 at -V 2, 0.725 -> 0.771 at -V 2 -P 3 (the neighbour's workgroup that leaves at once still takes a slot and
 * three dependent loads, and two items in lock step wait for the slower one at every barrier).  Not the default. */
__host__ __device__ constexpr int fast_group_log2(uint32_t fl, uint32_t ltp_order)
{
#ifdef SRLA_ITEM_GROUPS
    return (fl == 1u) ? 1 : ((fl == 2u && ltp_order == 0u) ? 1 : 0);
#else
    return (void)fl, (void)ltp_order, 0;
#endif
}
typedef short srla_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2_i16(uint32_t a, uint32_t b, uint32_t acc)
{
    return (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(srla_short2, a), __builtin_bit_cast(srla_short2, b), (int)acc, false);
}
__device__ __forceinline__ uint32_t dot4_i8(uint32_t a, uint32_t b, uint32_t acc)
{
    return (uint32_t)__builtin_amdgcn_sdot4((int)a, (int)b, (int)acc, false);
}
/* distance, in padded groups of four samples, from a thread's first own group to the group d groups away (d < FL): going
 * back, a pad group lies behind every FL groups (even FL only, see sig_index); d is the same in every lane */
template <int FL>
__device__ __forceinline__ int group_offset(int d)
{
    if constexpr (FL & 1) return d;
    else return (d < 0) ? d - ((FL - 1 - d) / FL) : d;
}

/* LG: log2 of the items that share the workgroup (0, 1, 2).  A thread's fixed work in the Rice search -- a dozen parameters, eleven
 * wave reductions, the parameter table -- does not shrink with its samples, so at four or eight samples per thread (1024- and
 * 2048-sample blocks on 256 threads) it outweighs the per-sample work (cut-short timing: 45 % of a -V 2 launch).  With 2^LG items
 * per workgroup an item has T = 256 >> LG threads of 4 FL << LG samples each; `lds`, `in`, `it`, `out` are the thread's own item's,
 * barriers are the workgroup's (the items run in lock step: every barrier below is reached by all of them). */
template <int FL, int MODE, int LG = 0>
__device__ __forceinline__ void residual_cost_fast(const SrlaJobParams &jp, const InputView &iv, const int32_t *__restrict__ in, const SrlaItemDesc &it,
                                   unsigned char *lds, const double *__restrict__ rice_thresholds,
                                   int32_t *__restrict__ res_ws, SrlaItemResult *__restrict__ out)
{
    constexpr int T = NT >> LG, WPI = T / WAVE;                  /* threads, wavefronts per item */
    constexpr int CH = FL << LG;                                /* chunks of four samples per thread */
    constexpr int S = 4 * CH;                                   /* samples per thread */
    constexpr int PADMIN = (FIR_PAD > SRLA_LTP_MAX_PERIOD + 2) ? FIR_PAD : (SRLA_LTP_MAX_PERIOD + 2);
    constexpr int PADS = ((PADMIN + S - 1) / S) * S;            /* front padding, a multiple of S: covers the FIR's reach back and the LTP's */
    constexpr int PADW = (CH & 1) ? 0 : 4;                      /* see sig_index */
    constexpr uint32_t SIG_WORDS = (uint32_t)((PADS + 1024 * FL) / S) * (S + PADW) + 8;
    constexpr bool WIDE = MODE == FIR_WIDE, MF = MODE == FIR_MFMA, DOT = MODE == FIR_DOT || MF;   /* (DOT: the signal lives as planes; MF: byte planes) */
    constexpr uint32_t MF_PLS = MF_PADB + 1024u * FL;            /* bytes of one byte plane */
    static_assert(!MF || (LG == 0 && FL <= 4), "FIR_MFMA: one item per workgroup, blocks of at most 4096 samples");
    static_assert(!MF || 3u * MF_PLS <= fast_sig_bytes(FL, LG, true), "the byte planes fit where the int16 / int8 planes would lie");
    /* FIR_DOT: the two planes lie over the int32 signal (which then only the LTP uses, before them): PADF zeros + the block, in
     * the same padded element order as the int32 layout */
    constexpr int PADF = ((FIR_PAD + S - 1) / S) * S;
    constexpr uint32_t PLANE_ELEMS = (uint32_t)((PADF + 1024 * FL) / S) * (S + PADW);
    constexpr uint32_t HIGH_OFF = (PLANE_ELEMS * 2 + 15) & ~15u;
    static_assert(HIGH_OFF + PLANE_ELEMS <= SIG_WORDS * 4, "the planes fit the int32 signal's LDS");
    int32_t *sig = (int32_t *)lds;
    static_assert(fast_sig_bytes(FL, LG, false) == ((SIG_WORDS * 4 + 15) & ~15u) && fast_sig_bytes(FL, LG, true) == ((HIGH_OFF + PLANE_ELEMS + 15) & ~15u), "one layout");
    SmallF *sm = (SmallF *)(lds + fast_sig_bytes(FL, LG, DOT && jp.ltp_order == 0));
    const uint32_t tid = threadIdx.x & (uint32_t)(T - 1), lane = tid & 63, wave = tid >> 6;   /* thread, wavefront within the item */
    const uint32_t n = 1024u * FL, bps = jp.bits_per_sample;
    const bool aligned = input_aligned(in, iv);
    const int32_t coef = out->preemph_coef;
    const uint32_t order = out->lpc_order, rshift = out->lpc_rshift, period = out->ltp_period;
    const uint32_t o4 = (order + 3u) & ~3u;
    const uint32_t s_base = (uint32_t)S * tid;
    PHASE_INIT();
    /* (two more values that would otherwise be fetched late, in front of a barrier / of the record's last store: requested here,
     * where their round trips run beside the sample loads) */
    const double thr_mine = (tid >= 32 && tid < 64) ? rice_thresholds[tid - 32] : 0.0;
    const uint32_t tap_bits = (tid == 0) ? out->pad[0] : 0u;       /* tap codes (srla_lpc_taps) */

    /* load + pre-emphasis (srla_utility.c:342) */
    int32_t y[S];
#ifdef SRLA_DIAG_STOP
    const uint32_t load_variant_as = (jp.out_stride >= 21 && jp.out_stride <= 26) ? 0u : it.variant;   /* what would cheaper loads buy? */
#else
    const uint32_t load_variant_as = it.variant;
#endif
#pragma unroll
    for (int c = 0; c < CH; c++) {
        int32_t t4[4];
        load_chunk(in, iv, load_variant_as, s_base + 4 * c, n, aligned, t4);
        y[4 * c] = t4[0]; y[4 * c + 1] = t4[1]; y[4 * c + 2] = t4[2]; y[4 * c + 3] = t4[3];
    }
    /* FIR_DOT: the taps a thread will pack (group tid of four taps and its three neighbours on either side) are requested HERE,
     * behind the sample loads and ahead of everything that waits for them, so that their round trip to the item record (which
     * has to wait for the order) runs beside the samples' instead of standing between the planes and the barrier below */
    uint32_t ctap[7] = { 0, 0, 0, 0, 0, 0, 0 };
    /* FIR_MFMA: the words of the four shifted copies of the zero-padded tap string this thread will store (wavefront s builds copy s,
     * words lane, lane + 64, lane + 128; byte m of copy s = tap m - s - MF_OFFZ) */
    constexpr int TZR = MF ? 3 : 1;                       /* up to 136 words per copy */
    static_assert(!MF || T == 4 * WAVE, "one wavefront per copy");
    uint32_t tzw[TZR] = { 0 };
    const uint32_t mf_p2 = (order + 15u) & ~15u;          /* the order rounded up to 16: how far in front of a window the loads start */
    const uint32_t mf_nkb = MF ? (uint32_t)__builtin_amdgcn_readfirstlane((16u * FL - 1u + mf_p2 + 63u) >> 6) : 0u;   /* k-blocks of 64 */
    const uint32_t mf_ndw = (MF_OFFZ + 64u * mf_nkb + 16u) >> 2;   /* words of a copy that are read */
    if constexpr (MF) {
#pragma unroll
        for (int r = 0; r < TZR; r++) {
            const uint32_t sft = wave, dw = lane + 64u * (uint32_t)r;
            if (dw < mf_ndw) {
                /* bytes k0 .. k0 + 3 of the tap array (zero outside [0, order)): two aligned words of it -- the taps are a few dozen
                 * bytes in one or two cache lines, whatever the lane -- funnel-shifted by the copy's shift, then masked */
                const int k0 = (int)(4u * dw) - (int)sft - MF_OFFZ;
                const int q0 = k0 >> 2, nq = (int)((order + 3u) >> 2);
                const uint32_t *cw = reinterpret_cast<const uint32_t *>(out->lpc_coef);
                const uint32_t lo = (q0 >= 0 && q0 < nq) ? cw[q0] : 0u, hi = (q0 + 1 >= 0 && q0 + 1 < nq) ? cw[q0 + 1] : 0u;
                const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(k0 & 3));
                const int first = (k0 < 0) ? -k0 : 0, last = ((int)order - k0 < 4) ? (int)order - k0 : 4;     /* valid bytes [first, last) */
                uint32_t mask = 0;
                if (last > first && first < 4) mask = (0xFFFFFFFFu >> (8 * (4 - last))) & (0xFFFFFFFFu << (8 * first));
                tzw[r] = w & mask;
            }
        }
    } else if constexpr (DOT) {
        if (tid <= (o4 >> 2)) {
            const int b = 4 * (int)tid;
#pragma unroll
            for (int d = -3; d <= 3; d++) {
                const int k = b + d;
                ctap[3 + d] = (k < (int)(o4 - order) || k >= (int)o4) ? 0u : (uint32_t)(int32_t)out->lpc_coef[k - (int)(o4 - order)];
            }
        }
    }
    asm volatile("" :: "v"(y[0]), "v"(y[S - 1]));
    PHASE(1, 0);                                                      /* sample loads landed */
    {
        /* the sample before the thread's first is the lane below's last (DPP); the wavefront's first lane fetches its own */
        const int32_t edge = (lane == 0 && tid != 0) ? load_variant(in, iv, it.variant, s_base - 1) : y[0];
        int32_t prev = __builtin_amdgcn_update_dpp(edge, y[S - 1], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
#pragma unroll
        for (int i = 0; i < S; i++) {
            const int32_t cur = y[i];
            /* narrow input: the sample (at most 19 bits: S = R - L of 18-bit input) times the 5-bit tap on the full-rate 24-bit
             * multiplier (v_mul_lo_u32 issues at a quarter of the rate) */
            const int32_t prod = WIDE ? (int32_t)((uint32_t)prev * (uint32_t)coef) : __mul24(prev, coef);
            y[i] = (int32_t)((uint32_t)cur - (uint32_t)(prod >> 4));
            prev = cur;
        }
    }
#define PUBLISH_Y()                                                                                              \
    _Pragma("unroll") for (int c = 0; c < CH; c++)                                                               \
        *reinterpret_cast<int4 *>(sig + sig_index<CH>(PADS + (int)s_base + 4 * c)) = make_int4(y[4 * c], y[4 * c + 1], y[4 * c + 2], y[4 * c + 3]);
    /* FIR_DOT: the block as two planes, x = 2^16 h + l (the wavefront notes whether any of its h is not zero) */
    auto publish_planes = [&]() {
        if constexpr (MF) {
            /* signed byte digits: s0 = the low byte, x1 = (x + 128) >> 8 = s1 + 256 x2; as BYTES: plane 0 = x & 0xff, plane 1 = byte 1 of
             * t = x + 128, plane 2 = x2 = (t + 32768) >> 16 -- zero unless t leaves 16 bits, which the wavefront notes */
            int32_t t[S];
            uint32_t high_any = 0;
#pragma unroll
            for (int i = 0; i < S; i++) { t[i] = y[i] + 128; high_any |= (uint32_t)((t[i] + 32768) >> 16); }
            const bool wave_high = __any((int)(high_any != 0));
            uint32_t w0[CH], w1[CH], w2[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) {
                /* one byte permute gives both planes' bytes of two samples: [t0.b0, t1.b0, t0.b1, t1.b1]; byte 0 of t is the low byte of x
                 * with its top bit flipped (x + 128): one XOR per word puts it back */
                const uint32_t q01 = __builtin_amdgcn_perm((uint32_t)t[4 * c + 1], (uint32_t)t[4 * c + 0], 0x05010400u);
                const uint32_t q23 = __builtin_amdgcn_perm((uint32_t)t[4 * c + 3], (uint32_t)t[4 * c + 2], 0x05010400u);
                w0[c] = __builtin_amdgcn_perm(q23, q01, 0x05040100u) ^ 0x80808080u;
                w1[c] = __builtin_amdgcn_perm(q23, q01, 0x07060302u);
                uint32_t hw = 0;
                if (wave_high) {
#pragma unroll
                    for (int i = 0; i < 4; i++) hw |= ((uint32_t)((t[4 * c + i] + 32768) >> 16) & 0xFFu) << (8 * i);
                }
                w2[c] = hw;
            }
            unsigned char *p0 = lds + MF_PADB + s_base;
            if constexpr (CH == 4) {
                *reinterpret_cast<uint4 *>(p0) = make_uint4(w0[0], w0[1], w0[2], w0[3]);
                *reinterpret_cast<uint4 *>(p0 + MF_PLS) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
                *reinterpret_cast<uint4 *>(p0 + 2 * MF_PLS) = make_uint4(w2[0], w2[1], w2[2], w2[3]);
            } else if constexpr (CH == 2) {
                *reinterpret_cast<uint2 *>(p0) = make_uint2(w0[0], w0[1]);
                *reinterpret_cast<uint2 *>(p0 + MF_PLS) = make_uint2(w1[0], w1[1]);
                *reinterpret_cast<uint2 *>(p0 + 2 * MF_PLS) = make_uint2(w2[0], w2[1]);
            } else {
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    reinterpret_cast<uint32_t *>(p0)[c] = w0[c];
                    reinterpret_cast<uint32_t *>(p0 + MF_PLS)[c] = w1[c];
                    reinterpret_cast<uint32_t *>(p0 + 2 * MF_PLS)[c] = w2[c];
                }
            }
            if (lane == 0) {
                sm->wave_high[wave] = wave_high ? 1u : 0u;
                if (WPI < NWAVES && wave == 0) for (int w = WPI; w < NWAVES; w++) sm->wave_high[w] = 0u;
            }
            /* front padding: zero samples */
            for (uint32_t i = tid; i < 3u * (MF_PADB / 4); i += T) reinterpret_cast<uint32_t *>(lds + (i / (MF_PADB / 4)) * MF_PLS)[i % (MF_PADB / 4)] = 0;
            return;
        }
        uint32_t hq[S], high_any = 0;
#pragma unroll
        for (int i = 0; i < S; i++) { hq[i] = (uint32_t)((y[i] + 0x8000) >> 16); high_any |= hq[i]; }
        const bool wave_high = __any((int)(high_any != 0));
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t e = sig_index<CH>(PADF + (int)s_base + 4 * c);
            *reinterpret_cast<uint2 *>(lds + 2 * e) = make_uint2(((uint32_t)y[4 * c] & 0xFFFFu) | ((uint32_t)y[4 * c + 1] << 16),
                                                                 ((uint32_t)y[4 * c + 2] & 0xFFFFu) | ((uint32_t)y[4 * c + 3] << 16));
            uint32_t hw = 0;
            if (wave_high) hw = (hq[4 * c] & 0xFFu) | ((hq[4 * c + 1] & 0xFFu) << 8) | ((hq[4 * c + 2] & 0xFFu) << 16) | (hq[4 * c + 3] << 24);
            *reinterpret_cast<uint32_t *>(lds + HIGH_OFF + e) = hw;
        }
        if (lane == 0) {
            sm->wave_high[wave] = wave_high ? 1u : 0u;
            if (WPI < NWAVES && wave == 0) for (int w = WPI; w < NWAVES; w++) sm->wave_high[w] = 0u;   /* (read four at a time below) */
        }
        constexpr uint32_t FRONT = (uint32_t)(PADF / S) * (S + PADW);     /* front padding, elements */
        for (uint32_t i = tid; i < FRONT / 2; i += T) reinterpret_cast<uint32_t *>(lds)[i] = 0;
        for (uint32_t i = tid; i < FRONT / 4; i += T) reinterpret_cast<uint32_t *>(lds + HIGH_OFF)[i] = 0;
    };
    /* the LTP's two barriers are taken by every item of the workgroup or by none */
    const bool ltp_block = (LG == 0) ? (period > 0) : (jp.ltp_order > 0);
    if (DOT && !ltp_block) publish_planes();
    else {
        PUBLISH_Y();
        for (uint32_t i = tid; i < (uint32_t)(PADS / S) * (S + PADW); i += T) sig[i] = 0;    /* front padding */
    }
    if constexpr (MF) {
#pragma unroll
        for (int r = 0; r < TZR; r++) {
            const uint32_t dw = lane + 64u * (uint32_t)r;
            if (dw < mf_ndw) reinterpret_cast<uint32_t *>(sm->tz[wave])[dw] = tzw[r];
        }
    } else if constexpr (DOT) {
        /* tap k of the zero-padded, reversed filter (k outside [0, o4): zero): ctap[3 + d] = tap 4 tid + d, fetched at the top */
        static_assert(FIR_PAD / 4 < T, "one group of four taps per thread");
        if (tid <= (o4 >> 2)) {
            const int b = 4 * (int)tid;
            const uint32_t (&c)[7] = ctap;
            /* low plane: outputs 0 and 2 of a chunk use (b, b+1) (b+2, b+3), outputs 1 and 3 (b-1, b) (b+1, b+2) */
            sm->cpack[0][b + 0] = (c[3] & 0xFFFFu) | (c[4] << 16);
            sm->cpack[0][b + 1] = (c[5] & 0xFFFFu) | (c[6] << 16);
            sm->cpack[0][b + 2] = (c[2] & 0xFFFFu) | (c[3] << 16);
            sm->cpack[0][b + 3] = (c[4] & 0xFFFFu) | (c[5] << 16);
            /* high plane: output r of a chunk uses taps b - r .. b - r + 3 */
#pragma unroll
            for (int r = 0; r < 4; r++)
                sm->cpack[1][b + r] = (c[3 - r] & 0xFFu) | ((c[4 - r] & 0xFFu) << 8) | ((c[5 - r] & 0xFFu) << 16) | (c[6 - r] << 24);
        }
    } else {
        for (uint32_t k = tid; k < o4; k += T) sm->coefq[k] = (k < o4 - order) ? 0 : (int32_t)out->lpc_coef[k - (o4 - order)];
    }
    if (tid < 16) sm->level_bits[tid] = 0;
    if (tid >= 32 && tid < 64) sm->thr[tid - 32] = thr_mine;
    __syncthreads();
    PHASE(1, 1);                                                      /* pre-emphasis, planes + taps published, barrier */
#ifdef SRLA_DIAG_STOP
    if (jp.out_stride == 1 || jp.out_stride == 22) { if (y[0] == 0x7fffffff) out->pad[1] = 1; return; }   /* kernel timing experiments (make EXTRA=-DSRLA_DIAG_STOP): never in the shipped library */
#endif

    if (ltp_block) {
        if (period > 0) {
        /* long-term predictor, srla_lpc_predict.c:267-294 (in place: read everything, barrier, rewrite) */
        const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
        const int32_t c0 = out->ltp_coef[0], c1 = out->ltp_coef[1], c2 = out->ltp_coef[2];
        /* the thread's S + 2 source samples are consecutive: one division locates the first one in the padded
         * layout (a pad of four words after every S), the others follow by compare-and-step.  PADS >= the largest
         * period + 2, so the first source index is never negative. */
        const uint32_t base0 = (uint32_t)PADS + s_base - period - half_order;
        const uint32_t q0 = base0 / (uint32_t)S, r0 = base0 - q0 * (uint32_t)S;
        int32_t src[S + 2];
#pragma unroll
        for (int i = 0; i < S + 2; i++) {
            const uint32_t step = (r0 + (uint32_t)i >= 2u * S) ? 2u * PADW : ((r0 + (uint32_t)i >= (uint32_t)S) ? (uint32_t)PADW : 0u);
            src[i] = (i < S || taps == 3) ? sig[base0 + (uint32_t)PADW * q0 + (uint32_t)i + step] : 0;
        }
#pragma unroll
        for (int i = 0; i < S; i++) {
            const uint32_t s = s_base + i;
            if (s >= period + half_order + 1) {
                uint32_t acc;
                if constexpr (!WIDE) {
                    /* 6-bit taps, samples within 24 bits (see the FIR below): the full-rate 24-bit multiplier */
                    acc = mad24(c0, src[i], 16u);
                    if (taps == 3) acc = mad24(c2, src[i + 2], mad24(c1, src[i + 1], acc));
                } else {
                    acc = 16u + (uint32_t)c0 * (uint32_t)src[i];
                    if (taps == 3) acc += (uint32_t)c1 * (uint32_t)src[i + 1] + (uint32_t)c2 * (uint32_t)src[i + 2];
                }
                y[i] = (int32_t)((uint32_t)y[i] - (uint32_t)((int32_t)acc >> 5));
            }
        }
        }
        __syncthreads();
        if constexpr (DOT) publish_planes(); else PUBLISH_Y();
        __syncthreads();
    }
#undef PUBLISH_Y

    PHASE(1, 2);                                                      /* LTP */
    /* int32 wrap-around FIR (srla_lpc_predict.c:118-265) */
    uint32_t u[S];
    uint32_t max_u = 0;
    {
        const int32_t half = (int32_t)(1u << ((rshift - 1u) & 31u));
        uint32_t acc[S];
#pragma unroll
        for (int i = 0; i < S; i++) acc[i] = (uint32_t)half;
        int32_t yprev = 0;
        if constexpr (MF) {
            typedef int mf_v4i __attribute__((ext_vector_type(4)));
            const uint32_t cc = lane & 15u, gk = lane >> 4;             /* column / row of the lane's operands, its k-group */
            const uint32_t dpad = mf_p2 - order;
            /* B: 16 bytes of a plane, 16-byte aligned (MF_PADB, the wavefront's base and the rounded order are multiples of 16) */
            const unsigned char *bp = lds + MF_PADB + (uint32_t)(64 * S) * wave + 16u * FL * cc - mf_p2 + 16u * gk;
            /* A: 16 bytes of the tap string from tap index 16 gk - 4 FL (cc >> 2) - (cc & 3) - dpad (+ 64 kb - 4 T): the copy shifted by
             * sft makes that a word address */
            const uint32_t sft = ((cc & 3u) + dpad) & 3u;
            const uint32_t *az = reinterpret_cast<const uint32_t *>(sm->tz[sft]) + ((MF_OFFZ + 16u * gk - 4u * FL * (cc >> 2) - (cc & 3u) - dpad + sft) >> 2);
            const uint4 wh = *reinterpret_cast<const uint4 *>(sm->wave_high);
            const bool high = __builtin_amdgcn_readfirstlane(wh.x | wh.y | wh.z | wh.w) != 0;
            mf_v4i a0[FL], a1[FL];
#pragma unroll
            for (int t = 0; t < FL; t++) { a0[t] = (mf_v4i){ half, half, half, half }; a1[t] = (mf_v4i){ 0, 0, 0, 0 }; }
            for (uint32_t kb = 0; kb < mf_nkb; kb++) {
                const mf_v4i b0 = *reinterpret_cast<const mf_v4i *>(bp + 64u * kb);
                const mf_v4i b1 = *reinterpret_cast<const mf_v4i *>(bp + MF_PLS + 64u * kb);
#pragma unroll
                for (int t = 0; t < FL; t++) {
                    const uint32_t *ap = az + 16u * kb - (uint32_t)t;
                    const mf_v4i a = (mf_v4i){ (int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3] };
                    a0[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b0, a0[t], 0, 0, 0);
                    a1[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b1, a1[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < FL; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) a0[t][i] = (int)((uint32_t)a0[t][i] + ((uint32_t)a1[t][i] << 8));
            if (high) {
                /* the third digit, where a sample left 16 bits (full-scale input): one more pass */
#pragma unroll
                for (int t = 0; t < FL; t++) a1[t] = (mf_v4i){ 0, 0, 0, 0 };
                for (uint32_t kb = 0; kb < mf_nkb; kb++) {
                    const mf_v4i b2 = *reinterpret_cast<const mf_v4i *>(bp + 2 * MF_PLS + 64u * kb);
#pragma unroll
                    for (int t = 0; t < FL; t++) {
                        const uint32_t *ap = az + 16u * kb - (uint32_t)t;
                        const mf_v4i a = (mf_v4i){ (int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3] };
                        a1[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b2, a1[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < FL; t++)
#pragma unroll
                    for (int i = 0; i < 4; i++) a0[t][i] = (int)((uint32_t)a0[t][i] + ((uint32_t)a1[t][i] << 16));
            }
            /* lane 16 g + cc holds chunk t of thread 4 cc + g: every thread fetches its own from lane 16 (lane & 3) + (lane >> 2) */
            const int from = (int)(4u * (16u * (lane & 3u) + (lane >> 2)));
#pragma unroll
            for (int t = 0; t < FL; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[4 * t + i] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, a0[t][i]);
            if (tid != 0 && s_base < order) {
                const unsigned char *pe = lds + MF_PADB + s_base - 1u;
                yprev = (int32_t)*reinterpret_cast<const int8_t *>(pe) + 256 * (int32_t)*reinterpret_cast<const int8_t *>(pe + MF_PLS)
                      + 65536 * (int32_t)*reinterpret_cast<const int8_t *>(pe + 2 * MF_PLS);
            }
        } else if constexpr (DOT) {
            const int ng = (int)__builtin_amdgcn_readfirstlane(o4 >> 2);          /* groups of four taps */
            const uint2 *lgrp = reinterpret_cast<const uint2 *>(lds);               /* low plane: a group = four int16 */
            const uint32_t *hgrp = reinterpret_cast<const uint32_t *>(lds + HIGH_OFF);   /* high plane: a group = four int8 */
            const int own = (int)(sig_index<CH>(PADF + (int)s_base) >> 2);           /* the thread's first own group */
            uint2 cur[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) cur[c] = lgrp[own + group_offset<CH>(c - ng)];
            for (int j = 0; j < ng; j++) {
                const uint4 cf = *reinterpret_cast<const uint4 *>(&sm->cpack[0][4 * j]);
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const uint2 nxt = lgrp[own + group_offset<CH>(c + j + 1 - ng)];
                    acc[4 * c + 0] = dot2_i16(cf.y, cur[c].y, dot2_i16(cf.x, cur[c].x, acc[4 * c + 0]));
                    acc[4 * c + 1] = dot2_i16(cf.w, cur[c].y, dot2_i16(cf.z, cur[c].x, acc[4 * c + 1]));
                    acc[4 * c + 2] = dot2_i16(cf.y, nxt.x, dot2_i16(cf.x, cur[c].y, acc[4 * c + 2]));
                    acc[4 * c + 3] = dot2_i16(cf.w, nxt.x, dot2_i16(cf.z, cur[c].y, acc[4 * c + 3]));
                    cur[c] = nxt;
                }
            }
            {
                /* the last tap of the odd outputs: its partner in the pair is the output's own sample, times zero */
                const uint32_t cl = sm->cpack[0][4 * ng + 2];
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    acc[4 * c + 1] = dot2_i16(cl, cur[c].x, acc[4 * c + 1]);
                    acc[4 * c + 3] = dot2_i16(cl, cur[c].y, acc[4 * c + 3]);
                }
            }
            const uint4 wh = *reinterpret_cast<const uint4 *>(sm->wave_high);
            if (__builtin_amdgcn_readfirstlane(wh.x | wh.y | wh.z | wh.w)) {
                uint32_t ah[S];
#pragma unroll
                for (int i = 0; i < S; i++) ah[i] = 0;
                for (int j = 0; j <= ng; j++) {
                    /* group ng closes the pass: the taps that are left for outputs 1..3 meet the chunk's own samples */
                    const uint4 cf = *reinterpret_cast<const uint4 *>(&sm->cpack[1][4 * j]);
#pragma unroll
                    for (int c = 0; c < CH; c++) {
                        const uint32_t a = hgrp[own + group_offset<CH>(c + j - ng)];
                        ah[4 * c + 0] = dot4_i8(cf.x, a, ah[4 * c + 0]);
                        ah[4 * c + 1] = dot4_i8(cf.y, a, ah[4 * c + 1]);
                        ah[4 * c + 2] = dot4_i8(cf.z, a, ah[4 * c + 2]);
                        ah[4 * c + 3] = dot4_i8(cf.w, a, ah[4 * c + 3]);
                    }
                }
#pragma unroll
                for (int i = 0; i < S; i++) acc[i] += ah[i] << 16;
            }
            if (tid != 0 && s_base < order) {
                const uint32_t e = sig_index<CH>(PADF + (int)s_base - 1);
                yprev = (int32_t)*reinterpret_cast<const int16_t *>(lds + 2 * e) + ((int32_t)*reinterpret_cast<const int8_t *>(lds + HIGH_OFF + e)) * 65536;
            }
        } else {
        int4 cur[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) cur[c] = *reinterpret_cast<const int4 *>(sig + sig_index<CH>(PADS + (int)s_base + 4 * c - (int)o4));
        /* !WIDE: every sample fits in 24 bits (bps <= 18: |x| < 2^(bps-1) per channel, S = R - L doubles it,
         * pre-emphasis doubles again, the LTP at most quadruples) and the taps are 8-bit, so the full-rate 24-bit
         * multiply gives the same low 32 bits as the wrap-around 32-bit product */
        if constexpr (MODE == FIR_MAD24) {
        for (uint32_t kb = 0; kb < o4; kb += 4) {
            const int4 cf = *reinterpret_cast<const int4 *>(&sm->coefq[kb]);
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const int4 nxt = *reinterpret_cast<const int4 *>(sig + sig_index<CH>(PADS + (int)s_base + 4 * c - (int)o4 + (int)kb + 4));
                const int w0 = cur[c].x, w1 = cur[c].y, w2 = cur[c].z, w3 = cur[c].w, w4 = nxt.x, w5 = nxt.y, w6 = nxt.z;
                acc[4 * c + 0] = mad24(cf.w, w3, mad24(cf.z, w2, mad24(cf.y, w1, mad24(cf.x, w0, acc[4 * c + 0]))));
                acc[4 * c + 1] = mad24(cf.w, w4, mad24(cf.z, w3, mad24(cf.y, w2, mad24(cf.x, w1, acc[4 * c + 1]))));
                acc[4 * c + 2] = mad24(cf.w, w5, mad24(cf.z, w4, mad24(cf.y, w3, mad24(cf.x, w2, acc[4 * c + 2]))));
                acc[4 * c + 3] = mad24(cf.w, w6, mad24(cf.z, w5, mad24(cf.y, w4, mad24(cf.x, w3, acc[4 * c + 3]))));
                cur[c] = nxt;
            }
        }
        } else {
        int4 curh[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            curh[c] = make_int4(cur[c].x >> 16, cur[c].y >> 16, cur[c].z >> 16, cur[c].w >> 16);
            cur[c] = make_int4(cur[c].x & 0xFFFF, cur[c].y & 0xFFFF, cur[c].z & 0xFFFF, cur[c].w & 0xFFFF);
        }
        for (uint32_t kb = 0; kb < o4; kb += 4) {
            const int4 cf = *reinterpret_cast<const int4 *>(&sm->coefq[kb]);
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const int4 nx = *reinterpret_cast<const int4 *>(sig + sig_index<CH>(PADS + (int)s_base + 4 * c - (int)o4 + (int)kb + 4));
                const int4 nl = make_int4(nx.x & 0xFFFF, nx.y & 0xFFFF, nx.z & 0xFFFF, nx.w & 0xFFFF);
                const int4 nh = make_int4(nx.x >> 16, nx.y >> 16, nx.z >> 16, nx.w >> 16);
                {
                    const int w0 = cur[c].x, w1 = cur[c].y, w2 = cur[c].z, w3 = cur[c].w, w4 = nl.x, w5 = nl.y, w6 = nl.z;
                    acc[4 * c + 0] = mad24(cf.w, w3, mad24(cf.z, w2, mad24(cf.y, w1, mad24(cf.x, w0, acc[4 * c + 0]))));
                    acc[4 * c + 1] = mad24(cf.w, w4, mad24(cf.z, w3, mad24(cf.y, w2, mad24(cf.x, w1, acc[4 * c + 1]))));
                    acc[4 * c + 2] = mad24(cf.w, w5, mad24(cf.z, w4, mad24(cf.y, w3, mad24(cf.x, w2, acc[4 * c + 2]))));
                    acc[4 * c + 3] = mad24(cf.w, w6, mad24(cf.z, w5, mad24(cf.y, w4, mad24(cf.x, w3, acc[4 * c + 3]))));
                }
                {
                    /* the high halves' four products are summed on their own and enter shifted (one shift-add per group) */
                    const int w0 = curh[c].x, w1 = curh[c].y, w2 = curh[c].z, w3 = curh[c].w, w4 = nh.x, w5 = nh.y, w6 = nh.z;
                    acc[4 * c + 0] += mad24(cf.w, w3, mad24(cf.z, w2, mad24(cf.y, w1, mad24(cf.x, w0, 0u)))) << 16;
                    acc[4 * c + 1] += mad24(cf.w, w4, mad24(cf.z, w3, mad24(cf.y, w2, mad24(cf.x, w1, 0u)))) << 16;
                    acc[4 * c + 2] += mad24(cf.w, w5, mad24(cf.z, w4, mad24(cf.y, w3, mad24(cf.x, w2, 0u)))) << 16;
                    acc[4 * c + 3] += mad24(cf.w, w6, mad24(cf.z, w5, mad24(cf.y, w4, mad24(cf.x, w3, 0u)))) << 16;
                }
                cur[c] = nl; curh[c] = nh;
            }
        }
        }
        yprev = (tid == 0) ? 0 : sig[sig_index<CH>(PADS + (int)s_base - 1)];
        }
        int32_t rr[S];
        /* the first `order` samples of the block are differenced, not predicted (srla_lpc_predict.c:118-265): only the first
         * threads of the first wavefront hold any, every other wavefront takes the plain form without per-sample selects */
        if (__any((int)(order == 0 || s_base < order))) {
#pragma unroll
            for (int i = 0; i < S; i++) {
                const uint32_t s = s_base + i;
                int32_t rv;
                if (order == 0 || s == 0) rv = y[i];
                else if (s < order) rv = (int32_t)((uint32_t)y[i] - (uint32_t)((i == 0) ? yprev : y[i - 1]));
                else rv = (int32_t)((uint32_t)y[i] + (uint32_t)((int32_t)acc[i] >> rshift));
                rr[i] = rv;
            }
        } else {
#pragma unroll
            for (int i = 0; i < S; i++) rr[i] = (int32_t)((uint32_t)y[i] + (uint32_t)((int32_t)acc[i] >> rshift));
        }
#pragma unroll
        for (int i = 0; i < S; i++) {
            u[i] = zigzag32(rr[i]);
            max_u = (u[i] > max_u) ? u[i] : max_u;
        }
    }
#ifdef SRLA_DIAG_STOP
    if (jp.out_stride == 2) { if (max_u == 0x7fffffff) out->pad[1] = 1; return; }   /* kernel timing experiments (make EXTRA=-DSRLA_DIAG_STOP): never in the shipped library */
#endif

    asm volatile("" :: "v"(u[0]), "v"(u[S - 1]), "v"(max_u));
    PHASE(1, 3);                                                      /* FIR, residual, zig-zag */
    /* Partition means: exact integer sums at the finest level, pairwise averages above (srla_coder.c:366-389).  A thread holds
     * Q = 4 << LG finest partitions of FL samples; its levels 10 .. TL = 8 - LG form a heap in registers (node 1: the partition
     * that is the thread, nodes Q .. 2Q-1: level 10), levels TL-1 .. TL-6 come by wave shuffles, and what is left above the
     * wavefront (LS = TL - 6 levels) through LDS across the item's wavefronts. */
    constexpr int Q = 4 << LG, LOGQ = 2 + LG, TL = 8 - LG, LS = TL - 6;
    const bool sums32 = __all((int)(max_u < (1u << 28)));
    /* the means of the thread's finest partitions; then level by level in place: lv[i] = (lv[2i] + lv[2i+1]) / 2 */
    auto finest_means = [&](double *lv) {
        if (sums32) {
            /* the sum of a finest partition (at most 8 values) stays within 32 bits: one add per sample and an exact conversion */
#pragma unroll
            for (int p = 0; p < Q; p++) {
                uint32_t sum = 0;
#pragma unroll
                for (int i = 0; i < FL; i++) sum += u[p * FL + i];
                lv[p] = (double)sum / (double)FL;
            }
        } else {
#pragma unroll
            for (int p = 0; p < Q; p++) {
                unsigned long long sum = 0;
#pragma unroll
                for (int i = 0; i < FL; i++) sum += u[p * FL + i];
                lv[p] = (double)sum / (double)FL;
            }
        }
    };
    double m[TL + 1];                               /* m[l]: mean of the level-l partition this thread lies in, l <= TL */
    {
        double lv[Q];
        finest_means(lv);
#pragma unroll
        for (int w = Q / 2; w >= 1; w >>= 1)
#pragma unroll
            for (int i = 0; i < w; i++) lv[i] = (lv[2 * i] + lv[2 * i + 1]) / 2.0;
        m[TL] = lv[0];
    }
#pragma unroll
    for (int l = TL - 1; l >= LS; l--) {
        const double other = __shfl_xor(m[l + 1], 1 << (TL - 1 - l), WAVE);
        /* (mean[2p] + mean[2p+1]) / 2: the lane holding the even child adds in that order */
        const bool even = ((lane >> (TL - 1 - l)) & 1u) == 0;
        m[l] = even ? (m[l + 1] + other) / 2.0 : (other + m[l + 1]) / 2.0;
    }
    max_u = wave_max_u32(max_u);
    if (lane == 0) { sm->wave_mean[wave] = m[LS]; sm->wave_max[wave] = max_u; }
    __syncthreads();
    PHASE(1, 4);                                                      /* partition means, barrier */
    {
        if constexpr (LS == 2) {
            const double a = sm->wave_mean[0], b = sm->wave_mean[1], c = sm->wave_mean[2], d = sm->wave_mean[3];
            const double m1a = (a + b) / 2.0, m1b = (c + d) / 2.0;
            m[1] = (wave < 2) ? m1a : m1b;
            m[0] = (m1a + m1b) / 2.0;
        } else if constexpr (LS == 1) {
            m[0] = (sm->wave_mean[0] + sm->wave_mean[1]) / 2.0;
        }
        max_u = sm->wave_max[0];
        for (int w = 1; w < WPI; w++) max_u = (sm->wave_max[w] > max_u) ? sm->wave_max[w] : max_u;
    }
    /* The residual goes to the scratch in HBM for srla_pack_blocks -- of EVERY item, chosen or not, which made these stores most
     * of the launch's memory traffic.  What the pack kernel codes is the zig-zag mapped value, and where the block's largest one
     * fits 16 bits (ordinary 16-bit audio) that is what is stored, two bytes per sample, in the first half of the item's region
     * (SRLA_ITEM_RES_U16); else, and for SRLAMI355X_ProbeBlock (keep_residuals == 2), the int32 residual. */
    /* every zig-zag value of the block within 16 bits (the whole item agrees): pairs of them in one register serve the store
     * below and the code-bit pass further down */
    const bool narrow16 = max_u < 65536u;
    uint32_t pk[S / 2];
#pragma unroll
    for (int j = 0; j < S / 2; j++) pk[j] = u[2 * j] | (u[2 * j + 1] << 16);
    if (jp.keep_residuals) {
        if (jp.keep_residuals == 1u && narrow16) {
            uint32_t *r16 = reinterpret_cast<uint32_t *>(res_ws + it.res_off) + (s_base >> 1);
            if constexpr ((CH & 1) == 0) {
#pragma unroll
                for (int c = 0; c < CH / 2; c++) *reinterpret_cast<uint4 *>(r16 + 4 * c) = make_uint4(pk[4 * c], pk[4 * c + 1], pk[4 * c + 2], pk[4 * c + 3]);
            } else {
#pragma unroll
                for (int c = 0; c < CH; c++) *reinterpret_cast<uint2 *>(r16 + 2 * c) = make_uint2(pk[2 * c], pk[2 * c + 1]);
            }
            if (tid == 0) out->flags |= SRLA_ITEM_RES_U16;
        } else {
            if (tid == 0) out->flags &= ~SRLA_ITEM_RES_U16;
            int32_t *res_out = res_ws + it.res_off + s_base;
#pragma unroll
            for (int c = 0; c < CH; c++) {
                int32_t r4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) r4[i] = (int32_t)((u[4 * c + i] >> 1) ^ (0u - (u[4 * c + i] & 1u)));
                *reinterpret_cast<int4 *>(res_out + 4 * c) = make_int4(r4[0], r4[1], r4[2], r4[3]);
            }
        }
    }
    uint32_t code_type;
    if (max_u == 0) code_type = SRLA_CODE_ALLZERO;
    else if (m[0] < 2) code_type = SRLA_CODE_RICE;
    else code_type = SRLA_CODE_RECURSIVE_RICE;
    PHASE(1, 5);                                                      /* residual store */
#ifdef SRLA_DIAG_STOP
    if (jp.out_stride == 3) { if (m[0] == 1.2345) out->pad[1] = 1; return; }   /* kernel timing experiments (make EXTRA=-DSRLA_DIAG_STOP): never in the shipped library */
#endif

    /* (the barriers below are outside the `coded` branches: with several items per workgroup an all-zero item meets them too) */
    const bool coded = code_type != SRLA_CODE_ALLZERO;
    uint32_t kth[2 * Q];       /* parameters of the thread's heap: kth[(1 << d) + j], partition j of level TL + d */
    uint32_t kl[TL + 1];       /* kl[l]: parameter of the level-l partition the thread lies in (kl[TL] = kth[1]) */
    if (coded) {
        /* The thread's heap of means once more, every mean turned into its parameter at once: keeping the 2Q - 1 doubles across
         * the barrier above instead cost the launch its occupancy (spills under the 96-register cap).  The empty asm keeps the
         * compiler from recognising the earlier computation and keeping its values alive after all. */
        {
#pragma unroll
            for (int i = 0; i < S; i++) asm volatile("" : "+v"(u[i]));
            double lv[Q];
            finest_means(lv);
#pragma unroll
            for (int p = 0; p < Q; p++) kth[Q + p] = rice_param(lv[p], code_type, sm->thr);
#pragma unroll
            for (int w = Q / 2; w >= 1; w >>= 1)
#pragma unroll
                for (int i = 0; i < w; i++) {
                    lv[i] = (lv[2 * i] + lv[2 * i + 1]) / 2.0;
                    kth[w + i] = rice_param(lv[i], code_type, sm->thr);
                }
        }
#pragma unroll
        for (int l = 0; l < TL; l++) kl[l] = rice_param(m[l], code_type, sm->thr);
        kl[TL] = kth[1];
        /* publish the table (leaders only), heap layout: level l at [2^l - 1, 2^(l+1) - 1) */
#pragma unroll
        for (int d = 0; d <= LOGQ; d++)
#pragma unroll
            for (int j = 0; j < (1 << d); j++) sm->ktab[((1u << (TL + d)) - 1u) + (tid << d) + (uint32_t)j] = (uint8_t)kth[(1 << d) + j];
#pragma unroll
        for (int l = 0; l < TL; l++)
            if ((tid & ((1u << (TL - l)) - 1u)) == 0) sm->ktab[((1u << l) - 1) + (tid >> (TL - l))] = (uint8_t)kl[l];
    }
    __syncthreads();
    PHASE(1, 6);                                                      /* Rice parameters, table, barrier */
#ifdef SRLA_DIAG_STOP
    if (jp.out_stride == 4) { if (coded && kl[0] + kth[2 * Q - 1] == 0x7fffffff) out->pad[1] = 1; return; }   /* kernel timing experiments (make EXTRA=-DSRLA_DIAG_STOP): never in the shipped library */
#endif
    uint32_t best_porder = 0, best_bits = 0;
    if (coded) {
        uint32_t acc[11];
        /* side information (srla_coder.c:415-427) booked by the first thread of each partition */
#pragma unroll
        for (int d = 0; d <= LOGQ; d++) {
            const uint32_t lbase = (1u << (TL + d)) - 1u;
            uint32_t side = 0;
#pragma unroll
            for (int j = 0; j < (1 << d); j++) {
                const uint32_t part = (tid << d) + (uint32_t)j;
                const uint32_t prevk = (j == 0) ? ((part == 0) ? 0u : sm->ktab[lbase + part - 1]) : kth[(1 << d) + (j > 0 ? j - 1 : 0)];
                side += (part == 0) ? 15u : (zigzag32((int32_t)kth[(1 << d) + j] - (int32_t)prevk) + 1u);
            }
            acc[TL + d] = side;
        }
#pragma unroll
        for (int l = 0; l < TL; l++) {
            uint32_t side = 0;
            if ((tid & ((1u << (TL - l)) - 1u)) == 0) {
                const uint32_t part = tid >> (TL - l);
                side = (part == 0) ? 15u : (zigzag32((int32_t)kl[l] - (int32_t)sm->ktab[((1u << l) - 1) + part - 1]) + 1u);
            }
            acc[l] = side;
        }
        /* Code bits of this thread's samples under every level's parameters.  Levels 0 .. TL price all of the thread's samples
         * with one parameter, level TL + d each of its 2^d parts: 11 evaluations per sample, no tables, no lane divergence (the
         * variable part is a saturating subtract + shift, srla_coder.c:327-347).  Neighbouring coarse levels very often have the
         * same parameter in every lane of the wavefront: then the thread's sum is the one just computed (wave-uniform test). */
        if (narrow16) {
            /* two samples per instruction: saturating subtract, shift and a dot product with (1, 1) that adds both halves
             * to a 32-bit sum (v_pk_sub_u16 clamp, v_pk_lshrrev_b16, v_dot2_u32_u16) -- three instructions per PAIR and
             * level instead of three per sample.  A parameter's threshold 2 << k leaves 16 bits at k = 15: every value
             * is below it then, as below 65535; k >= 16 prices every value at zero quotient bits likewise. */
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const us2 ones = { 1, 1 };
            auto thr16 = [&](uint32_t k) -> uint32_t {
                const uint32_t t2 = (code_type == SRLA_CODE_RICE) ? 0u : (2u << (k & 15u));
                return (k >= 16u) ? 0xFFFFu : ((t2 > 0xFFFFu) ? 0xFFFFu : t2);
            };
            auto pair_cost = [&](uint32_t w, uint32_t thr2, uint32_t sh2, uint32_t sum) -> uint32_t {
                const us2 dd = __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, w), __builtin_bit_cast(us2, thr2));
                return __builtin_amdgcn_udot2(dd >> __builtin_bit_cast(us2, sh2), ones, sum, false);
            };
            uint32_t t = 0;
#pragma unroll
            for (int l = 0; l <= TL; l++) {
                const bool same = (l > 0) && __all((int)(kl[l] == kl[l > 0 ? l - 1 : 0]));
                if (!same) {
                    const uint32_t thr2 = thr16(kl[l]) * 0x10001u, sh2 = (kl[l] & 15u) * 0x10001u;
                    t = (uint32_t)S * code_cost_fixed(kl[l], code_type);
#pragma unroll
                    for (int j = 0; j < S / 2; j++) t = pair_cost(pk[j], thr2, sh2, t);
                }
                acc[l] += t;
            }
#pragma unroll
            for (int d = 1; d <= LOGQ; d++) {
                const int plen = S >> d;                                   /* samples of a part */
                uint32_t td = 0, th[1 << LOGQ], sh[1 << LOGQ];
#pragma unroll
                for (int j = 0; j < (1 << d); j++) {
                    td += (uint32_t)plen * code_cost_fixed(kth[(1 << d) + j], code_type);
                    th[j] = thr16(kth[(1 << d) + j]); sh[j] = kth[(1 << d) + j] & 15u;
                }
#pragma unroll
                for (int j = 0; j < S / 2; j++) {
                    const int q0 = (2 * j) / plen, q1 = (2 * j + 1) / plen;   /* the parts of the pair's two samples */
                    td = pair_cost(pk[j], th[q0] | (th[q1] << 16), sh[q0] | (sh[q1] << 16), td);
                }
                acc[TL + d] += td;
            }
        } else {
            uint32_t t = 0;
#pragma unroll
            for (int l = 0; l <= TL; l++) {
                const bool same = (l > 0) && __all((int)(kl[l] == kl[l > 0 ? l - 1 : 0]));
                if (!same) {
                    t = (uint32_t)S * code_cost_fixed(kl[l], code_type);
#pragma unroll
                    for (int i = 0; i < S; i++) t += code_cost_var(u[i], kl[l], code_type);
                }
                acc[l] += t;
            }
#pragma unroll
            for (int d = 1; d <= LOGQ; d++) {
                const int plen = S >> d;
                uint32_t td = 0;
#pragma unroll
                for (int j = 0; j < (1 << d); j++) td += (uint32_t)plen * code_cost_fixed(kth[(1 << d) + j], code_type);
#pragma unroll
                for (int i = 0; i < S; i++) td += code_cost_var(u[i], kth[(1 << d) + i / plen], code_type);
                acc[TL + d] += td;
            }
        }
#ifdef SRLA_DIAG_STOP
        if (jp.out_stride == 5) { uint32_t tt = 0; for (int l = 0; l <= 10; l++) tt += acc[l]; if (tt == 0x7fffffff) out->pad[1] = 1; }
#endif
#pragma unroll
        for (int l = 0; l <= 10; l++) {
            const uint32_t sum = wave_sum_u32(acc[l]);
            if (lane == 0) atomicAdd(&sm->level_bits[l], sum);
        }
    }
#ifdef SRLA_DIAG_STOP
    if (jp.out_stride == 5) return;   /* kernel timing experiments (make EXTRA=-DSRLA_DIAG_STOP): never in the shipped library */
#endif
    __syncthreads();
    PHASE(1, 7);                                                      /* side information, code bits of 11 levels, reductions, barrier */
    if (coded) {
        best_bits = 0xFFFFFFFFu;
        for (uint32_t l = 0; l <= 10; l++) {
            const uint32_t b = sm->level_bits[l];
            if (b < best_bits) { best_bits = b; best_porder = l; }
        }
        for (uint32_t p = tid; p < (1u << best_porder); p += T) out->kparam[p] = sm->ktab[((1u << best_porder) - 1) + p];
    }
    if (tid == 0) {
        const uint32_t res_bits = best_bits + 2u;
        uint32_t bits = res_bits + (bps + 1u) + 5u + (8u + 4u + 1u) + tap_bits + 1u;   /* srla_encoder.c:1121-1187 */
        if (period > 0) bits += 1u + 8u + jp.ltp_order * 6u;
        out->code_length = bits;
        out->res_code_type = code_type;
        out->res_porder = best_porder;
        out->res_bits = res_bits;
    }
    PHASE(1, 8);                                                      /* arg-min, record */
}

extern "C" uint32_t srla_kernel_fast_lds_bytes(uint32_t fl, uint32_t ltp_order, uint32_t bits_per_sample)
{
    const bool dot = SRLA_FIR_NARROW == FIR_DOT && bits_per_sample <= 18;
    const int lg = dot ? fast_group_log2(fl, ltp_order) : 0;
    const uint32_t item = fast_sig_bytes((int)fl, lg, dot && ltp_order == 0) + (uint32_t)((sizeof(SmallF) + 15) & ~15u);
    return item << lg;
}

/* The partitioned (recursive) Rice parameter search of SRLACoder_ComputeCodeLength (srla_coder.c:349-484) over the zig-zag
 * mapped residual u[0..n) in LDS, and the channel's code length (srla_encoder.c:1121-1187): the tail shared by the
 * LDS paths of srla_residual_cost and srla_residual_cost_big.  NT threads; sm->max_u and sm->level_bits are set up by
 * the caller. */
__device__ __forceinline__ void rice_search_finish(const uint32_t *u, const SrlaGeom &g, double *means, SmallC *sm,
                                                   const double *__restrict__ rice_thresholds, uint32_t bps, uint32_t period,
                                                   uint32_t ltp_order, SrlaItemResult *__restrict__ out)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    /* ---- partitioned (recursive) Rice search, srla_coder.c:349-484 -------------------------------- */
    const uint32_t mp = g.max_porder, nparts = 1u << mp, fl = g.fine_len;
    unsigned long long *sums = (unsigned long long *)(means + (nparts - 1));
    const uint32_t tpp = (nparts >= NT) ? 1u : (NT / nparts);     /* threads per finest partition */
    if (tpp > 1) { for (uint32_t p = tid; p < nparts; p += NT) sums[p] = 0ull; }
    __syncthreads();
    if (tpp == 1) {
        for (uint32_t p = tid; p < nparts; p += NT) {
            unsigned long long s = 0;
            const uint32_t *up = u + p * fl;
            for (uint32_t i = 0; i < fl; i++) s += up[i];
            means[(nparts - 1) + p] = (double)s / (double)fl;     /* exact integer sum, srla_coder.c:373-381 */
        }
    } else {
        const uint32_t p = tid / tpp, j = tid % tpp;
        unsigned long long s = 0;
        const uint32_t *up = u + p * fl;
        for (uint32_t i = j; i < fl; i += tpp) s += up[i];
        atomicAdd(&sums[p], s);
        __syncthreads();
        for (uint32_t q = tid; q < nparts; q += NT) { const unsigned long long t = sums[q]; means[(nparts - 1) + q] = (double)t / (double)fl; }
    }
    __syncthreads();
    const uint32_t max_u_all = sm->max_u;
    uint32_t code_type;
    /* pairwise mean tree (srla_coder.c:385-389): wide levels by the whole workgroup, the narrow top by one wave */
    int lvl = (int)mp - 1;
    for (; lvl >= 0 && (1u << lvl) >= WAVE; lvl--) {
        const uint32_t cnt = 1u << lvl;
        for (uint32_t p = tid; p < cnt; p += NT)
            means[(cnt - 1) + p] = (means[(2 * cnt - 1) + 2 * p] + means[(2 * cnt - 1) + 2 * p + 1]) / 2.0;
        __syncthreads();
    }
    if (tid < WAVE) {
        for (; lvl >= 0; lvl--) {
            const uint32_t cnt = 1u << lvl;
            if (tid < cnt) means[(cnt - 1) + tid] = (means[(2 * cnt - 1) + 2 * tid] + means[(2 * cnt - 1) + 2 * tid + 1]) / 2.0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
    }
    __syncthreads();
    if (max_u_all == 0) code_type = SRLA_CODE_ALLZERO;
    else if (means[0] < 2) code_type = SRLA_CODE_RICE;
    else code_type = SRLA_CODE_RECURSIVE_RICE;

    uint32_t best_porder = 0, best_bits = 0;
    if (code_type != SRLA_CODE_ALLZERO) {
        /* parameter per (level, partition) */
        for (uint32_t e = tid; e < 2 * nparts - 1; e += NT) {
            const double mean = means[e];
            uint32_t k;
            if (code_type == SRLA_CODE_RICE) {
                k = 0;   /* srla_coder.c:262-276 through the host-derived monotone thresholds */
                for (int t = 0; t < 32; t++) k += (mean >= rice_thresholds[t]) ? 1u : 0u;
            } else {
                const double gp = 0.66794162356 * (1.0 + mean);   /* srla_coder.c:298-311 */
                const uint32_t golomb = (uint32_t)((1.0 > gp) ? 1.0 : gp);
                k = 31u - (uint32_t)__clz((int)golomb);
            }
            sm->ktab[e] = (uint8_t)k;
        }
        __syncthreads();
        /* cost of every partition order in one pass over the residual; side information per level:
         * 10 bits of partition order, 5 bits for the first parameter, zig-zag(delta) + 1 per further
         * partition (srla_coder.c:415-427) */
        uint32_t acc[SRLA_MAX_PORDER + 1];
#pragma unroll
        for (int l = 0; l <= SRLA_MAX_PORDER; l++) acc[l] = 0;
        {
            uint32_t p_first, p_step, j_first, j_step;
            if (tpp == 1) { p_first = tid; p_step = NT; j_first = 0; j_step = 1; }
            else { p_first = tid / tpp; p_step = nparts; j_first = tid % tpp; j_step = tpp; }
            for (uint32_t p = p_first; p < nparts; p += p_step) {
                const uint32_t *up = u + p * fl;
                uint32_t kk[SRLA_MAX_PORDER + 1];
#pragma unroll
                for (int l = 0; l <= SRLA_MAX_PORDER; l++) {
                    kk[l] = 0;
                    if ((uint32_t)l <= mp) {
                        const uint32_t pl = p >> (mp - l), e = ((1u << l) - 1) + pl;
                        kk[l] = sm->ktab[e];
                        /* the first thread of the first fine partition of a level-l partition books its side info */
                        if (j_first == 0 && (p & ((1u << (mp - l)) - 1)) == 0)
                            acc[l] += (pl == 0) ? 15u : (zigzag32((int32_t)kk[l] - (int32_t)sm->ktab[e - 1]) + 1u);
                    }
                }
                for (uint32_t i = j_first; i < fl; i += j_step) {
                    const uint32_t val = up[i];
#pragma unroll
                    for (int l = 0; l <= SRLA_MAX_PORDER; l++) {
                        if ((uint32_t)l <= mp) {
                            const uint32_t k = kk[l];
                            if (code_type == SRLA_CODE_RICE) {
                                acc[l] += 1u + k + (val >> k);                      /* srla_coder.c:327-330 */
                            } else {
                                int32_t over = (int32_t)val - (int32_t)(2u << k);  /* srla_coder.c:333-347 */
                                over = (over > 0) ? over : 0;
                                acc[l] += (k + 2u) + ((uint32_t)over >> k);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int l = 0; l <= SRLA_MAX_PORDER; l++) {
            if ((uint32_t)l <= mp) {
                const uint32_t s = wave_sum_u32(acc[l]);
                if (lane == 0) atomicAdd(&sm->level_bits[l], s);
            }
        }
        __syncthreads();
        best_bits = 0xFFFFFFFFu;
        for (uint32_t l = 0; l <= mp; l++) {
            const uint32_t b = sm->level_bits[l];
            if (b < best_bits) { best_bits = b; best_porder = l; }
        }
        for (uint32_t p = tid; p < (1u << best_porder); p += NT) out->kparam[p] = sm->ktab[((1u << best_porder) - 1) + p];
    }
    if (tid == 0) {
        const uint32_t res_bits = best_bits + 2u;
        uint32_t bits = res_bits;                 /* srla_encoder.c:1121-1187 */
        bits += bps + 1u;                         /* pre-emphasis state */
        bits += 5u;                               /* pre-emphasis tap   */
        bits += 8u + 4u + 1u;                     /* order, shift, sum flag */
        bits += out->pad[0];                      /* tap codes (srla_lpc_solve) */
        bits += 1u;                               /* LTP flag */
        if (period > 0) bits += 1u + 8u + ltp_order * 6u;
        out->code_length = bits;
        out->res_code_type = code_type;
        out->res_porder = best_porder;
        out->res_bits = res_bits;
    }
}

#ifndef SRLA_RC_WAVES
#define SRLA_RC_WAVES 5       /* wavefronts per SIMD the forms for blocks of at most 4096 samples are compiled for (92 registers) */
#endif
#ifndef SRLA_RC4_WAVES
#define SRLA_RC4_WAVES 3      /* wavefronts per SIMD the 8192-sample form is compiled for: 168 registers and 17 spilled dwords per lane; 2 (228 registers, no spills) was 9 % slower at -B 8192 -V 2 -P 3, profiles/r04/ab_residual_cost_split.txt */
#endif
template <int R, bool MFMA = false /* blocks of at most 4096 samples, narrow input: the FIR on the matrix pipe (FIR_MFMA) */>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(R >= 4 ? SRLA_RC4_WAVES : SRLA_RC_WAVES, 8))) void srla_residual_cost(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    const SrlaGeom *__restrict__ geoms, SrlaLdsPlan plan, const double *__restrict__ rice_thresholds,
    int32_t *__restrict__ res_ws, SrlaItemResult *__restrict__ results)
{
    constexpr int CH = 2 * R;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    /* as in srla_autocorr: workgroups go to the XCDs round robin, so each XCD is given one contiguous range of items --
     * the dozen items that read the same samples then share one L2 instead of pulling them into all eight */
    const uint32_t block = xcd_position(blockIdx.x, jp.num_items);
    if (block >= jp.num_items) return;
    const SrlaItemDesc itf = items[block];
    if (itf.n > 8192u) return;                       /* srla_residual_cost_big takes these */
    if (jp.rc_hi != 0u && (itf.n <= jp.rc_lo || itf.n > jp.rc_hi)) return;   /* the other launch of the job takes these */
    const InputView iv = input_view(jp, itf.lshift, input);
    {
        /* blocks of 1024 * FL samples take the register / shuffle fast path */
        const uint32_t fl = itf.n >> 10;
        if ((itf.n & 1023u) == 0 && fl >= 1 && fl <= 8 && fl <= (uint32_t)(2 * R)) {
            if (SRLA_FIR_NARROW == FIR_DOT && jp.bits_per_sample <= 18) {
                /* Blocks of 1024 and (without the LTP) 2048 samples share a workgroup two by two: the item with the even index
                 * leads, its neighbour's workgroup leaves at once.  Neighbours are nearly always the variants of one candidate,
                 * hence of one length; where they are not (odd item counts, mono candidates of different lengths) each item
                 * keeps its own workgroup. */
                const int lg = fast_group_log2(fl, jp.ltp_order);
                if (lg > 0) {
                    const uint32_t leader = block & ~((1u << lg) - 1u);
                    bool grouped = leader + (1u << lg) <= jp.num_items;
                    for (uint32_t g = 0; grouped && g < (1u << lg); g++) grouped = items[leader + g].n == itf.n;
                    if (grouped) {
                        if (block != leader) return;
                        /* (lg is 0 or 1.  The item is the wavefront's: a scalar, so that everything read through it -- order,
                         * shift, taps, pointers -- stays in scalar registers and scalar branches as with one item per workgroup) */
                        const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / (uint32_t)(NT >> 1)));
                        const SrlaItemDesc itg = items[leader + g];
                        const InputView ivg = input_view(jp, itg.lshift, input);
                        unsigned char *ldsg = lds + g * (fast_sig_bytes((int)fl, 1, jp.ltp_order == 0) + (uint32_t)((sizeof(SmallF) + 15) & ~15u));
                        if (fl == 1) residual_cost_fast<1, FIR_DOT, 1>(jp, ivg, input + itg.sample_off, itg, ldsg, rice_thresholds, res_ws, &results[leader + g]);
                        else residual_cost_fast<2, FIR_DOT, 1>(jp, ivg, input + itg.sample_off, itg, ldsg, rice_thresholds, res_ws, &results[leader + g]);
                        return;
                    }
                }
            }
            const int32_t *inf = input + itf.sample_off;
            SrlaItemResult *outf = &results[block];
#define FAST(FLV)                                                                                                   \
            do {                                                                                                    \
                if (jp.bits_per_sample <= 18) residual_cost_fast<FLV, (MFMA && FLV <= 4) ? FIR_MFMA : SRLA_FIR_NARROW>(jp, iv, inf, itf, lds, rice_thresholds, res_ws, outf); \
                else residual_cost_fast<FLV, FIR_WIDE>(jp, iv, inf, itf, lds, rice_thresholds, res_ws, outf);       \
                return;                                                                                             \
            } while (0)
            switch (fl) {
            case 1: FAST(1);
            case 2: FAST(2);
            case 3: FAST(3);
            case 4: FAST(4);
            default:
                /* only the 8192-sample class (R = 4) holds these instantiations */
                if constexpr (R >= 4) {
                    switch (fl) {
                    case 5: FAST(5);
                    case 6: FAST(6);
                    case 7: FAST(7);
                    default: FAST(8);
                    }
                }
                return;
            }
#undef FAST
        }
    }
    int32_t *sigA = (int32_t *)(lds + plan.y_off);        /* FIR_PAD zeros, then the signal (the LTP rewrites it in place) */
    double *means = (double *)(lds + plan.means_off);
    SmallC *sm = (SmallC *)(lds + plan.small_off);

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t item_idx = block;
    const SrlaItemDesc it = itf;
    const SrlaGeom g = geoms[it.geom];
    const uint32_t n = it.n, bps = jp.bits_per_sample;
    const int32_t *in = input + it.sample_off;
    const bool aligned = input_aligned(in, iv);
    SrlaItemResult *out = &results[item_idx];
    const int32_t coef = out->preemph_coef;
    const uint32_t order = out->lpc_order, rshift = out->lpc_rshift, period = out->ltp_period;
    const uint32_t o4 = (order + 3u) & ~3u;

    int32_t v[CH][4];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i4 = 4u * (tid + (uint32_t)c * NT);
        load_chunk(in, iv, it.variant, i4, n, aligned, v[c]);
        int32_t prev = (i4 == 0 || i4 >= n) ? v[c][0] : load_variant(in, iv, it.variant, i4 - 1);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int32_t cur = v[c][i];
            v[c][i] = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)coef) >> 4));
            prev = cur;
        }
        if (i4 < g.nfft) *reinterpret_cast<int4 *>(sigA + FIR_PAD + i4) = make_int4(v[c][0], v[c][1], v[c][2], v[c][3]);
    }
    for (uint32_t i = tid; i < FIR_PAD; i += NT) sigA[i] = 0;
    /* taps, zero padded in FRONT so that the tap loop runs in aligned groups of four */
    for (uint32_t k = tid; k < o4; k += NT) sm->coefq[k] = (k < o4 - order) ? 0 : (int32_t)out->lpc_coef[k - (o4 - order)];
    if (tid < 16) sm->level_bits[tid] = 0;
    if (tid == 0) sm->max_u = 0;
    __syncthreads();

    const int32_t *src = sigA + FIR_PAD;
    if (period > 0) {
        /* long-term predictor, srla_lpc_predict.c:267-294 */
        const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
        const int32_t c0 = out->ltp_coef[0], c1 = out->ltp_coef[1], c2 = out->ltp_coef[2];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i4 = 4u * (tid + (uint32_t)c * NT);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t s = i4 + i;
                if (s < n && s >= period + half_order + 1) {
                    const uint32_t base = s - period - half_order;
                    uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)src[base];
                    if (taps == 3) acc += (uint32_t)c1 * (uint32_t)src[base + 1] + (uint32_t)c2 * (uint32_t)src[base + 2];
                    v[c][i] = (int32_t)((uint32_t)v[c][i] - (uint32_t)((int32_t)acc >> 5));
                }
            }
        }
        /* every thread has read its sources: the filtered signal replaces the unfiltered one in place (a second buffer
         * used to cost the whole launch -- the register path included -- a third of its workgroups per CU) */
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i4 = 4u * (tid + (uint32_t)c * NT);
            if (i4 < g.nfft) *reinterpret_cast<int4 *>(sigA + FIR_PAD + i4) = make_int4(v[c][0], v[c][1], v[c][2], v[c][3]);
        }
        __syncthreads();
    }

    /* ---- int32 wrap-around FIR (srla_lpc_predict.c:118-265), four outputs per group, taps in fours ---- */
    uint32_t uz[CH][4];
    {
        const int32_t half = (int32_t)(1u << ((rshift - 1u) & 31u));
        uint32_t acc[CH][4];
        int4 cur[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i4 = 4u * (tid + (uint32_t)c * NT);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[c][i] = (uint32_t)half;
            cur[c] = (i4 < n) ? *reinterpret_cast<const int4 *>(src + (int)i4 - (int)o4) : make_int4(0, 0, 0, 0);
        }
        for (uint32_t kb = 0; kb < o4; kb += 4) {
            const int4 cf = *reinterpret_cast<const int4 *>(&sm->coefq[kb]);
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t i4 = 4u * (tid + (uint32_t)c * NT);
                if (i4 < n) {
                    const int4 nxt = *reinterpret_cast<const int4 *>(src + (int)i4 - (int)o4 + (int)kb + 4);
                    const uint32_t w0 = (uint32_t)cur[c].x, w1 = (uint32_t)cur[c].y, w2 = (uint32_t)cur[c].z, w3 = (uint32_t)cur[c].w;
                    const uint32_t w4 = (uint32_t)nxt.x, w5 = (uint32_t)nxt.y, w6 = (uint32_t)nxt.z;
                    const uint32_t f0 = (uint32_t)cf.x, f1 = (uint32_t)cf.y, f2 = (uint32_t)cf.z, f3 = (uint32_t)cf.w;
                    acc[c][0] += f0 * w0 + f1 * w1 + f2 * w2 + f3 * w3;
                    acc[c][1] += f0 * w1 + f1 * w2 + f2 * w3 + f3 * w4;
                    acc[c][2] += f0 * w2 + f1 * w3 + f2 * w4 + f3 * w5;
                    acc[c][3] += f0 * w3 + f1 * w4 + f2 * w5 + f3 * w6;
                    cur[c] = nxt;
                }
            }
        }
        uint32_t max_u = 0;
        int32_t *res_out = res_ws + it.res_off;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i4 = 4u * (tid + (uint32_t)c * NT);
            if (i4 < n) {
                int32_t rr[4];
                /* after the tap loop cur[c] holds src[i4 .. i4+3] */
                const int32_t y4[4] = { cur[c].x, cur[c].y, cur[c].z, cur[c].w };
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t s = i4 + i;
                    int32_t rv;
                    if (order == 0 || s == 0) rv = y4[i];
                    else if (s < order) rv = (int32_t)((uint32_t)y4[i] - (uint32_t)src[s - 1]);
                    else rv = (int32_t)((uint32_t)y4[i] + (uint32_t)((int32_t)acc[c][i] >> rshift));
                    if (s >= n) rv = 0;
                    rr[i] = rv;
                    const uint32_t z = zigzag32(rv);
                    uz[c][i] = z;
                    max_u = (z > max_u) ? z : max_u;
                }
                if (!jp.keep_residuals) { }
                else if (i4 + 4 <= n) *reinterpret_cast<int4 *>(res_out + i4) = make_int4(rr[0], rr[1], rr[2], rr[3]);
                else { for (int i = 0; i < 4; i++) if (i4 + i < n) res_out[i4 + i] = rr[i]; }
            }
        }
        max_u = wave_max_u32(max_u);
        if (lane == 0) atomicMax(&sm->max_u, max_u);
    }
    __syncthreads();   /* every FIR read of the signal is done: the zig-zag residual may overwrite it */
    uint32_t *u = (uint32_t *)(sigA + FIR_PAD);
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i4 = 4u * (tid + (uint32_t)c * NT);
        if (i4 < n) *reinterpret_cast<uint4 *>(u + i4) = make_uint4(uz[c][0], uz[c][1], uz[c][2], uz[c][3]);
    }

    rice_search_finish(u, g, means, sm, rice_thresholds, bps, period, jp.ltp_order, out);
}

/* ================================================================================================
 * Blocks above 8192 samples (-B 16384, -B 32768): the slow paths.  Nothing of such a block fits the LDS-resident
 * schemes above (a 32768-point transform is 256 KB), so the transform works in a global scratch buffer -- ping-pong, one
 * workgroup per item, the same butterflies in the same order -- and the residual / code-length pass keeps ONE int32
 * buffer in LDS that is rewritten in place from the top down.  Correct and complete (chain mode included), not fast.
 * ============================================================================================== */
#define NTB 1024

/* complex FFT of m points, src -> dst ping-pong in global memory (fft.c:71-136); returns where the result stands */
__device__ cplx *fft_complex_global(cplx *src, cplx *dst, uint32_t m, int flag, const cplx *__restrict__ tw)
{
    const uint32_t tid = threadIdx.x;
    uint32_t n = m, s = 1, log2s = 0;
    const uint32_t nb = m >> 2, m4 = m >> 2;
    while (n > 2) {
        const uint32_t n1 = n >> 2;
        for (uint32_t bf = tid; bf < nb; bf += NTB) {
            const uint32_t q = bf & (s - 1), p = bf >> log2s;
            const cplx w1 = tw[p], w2 = tw[n1 + p], w3 = tw[2 * n1 + p];
            const cplx a = src[bf], b = src[bf + m4], c = src[bf + 2 * m4], d = src[bf + 3 * m4];
            const cplx apc = c_add(a, c), amc = c_sub(a, c), bpd = c_add(b, d), bmd = c_sub(b, d);
            const cplx jbmd = (flag < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
            const uint32_t wb = 4u * bf - 3u * q;
            dst[wb] = c_add(apc, bpd);
            dst[wb + s] = c_mul(w1, c_sub(amc, jbmd));
            dst[wb + 2 * s] = c_mul(w2, c_sub(apc, bpd));
            dst[wb + 3 * s] = c_mul(w3, c_add(amc, jbmd));
        }
        __syncthreads();
        cplx *t = src; src = dst; dst = t;
        tw += 3 * n1;
        n >>= 2; s <<= 2; log2s += 2;
    }
    if (n == 2) {
        for (uint32_t q = tid; q < s; q += NTB) {
            const cplx a = src[q], b = src[q + s];
            dst[q] = c_add(a, b);
            dst[q + s] = c_sub(a, b);
        }
        __syncthreads();
        cplx *t = src; src = dst; dst = t;
    }
    return src;
}

/* srla_autocorr for items of more than 8192 points: persistent workgroups (each owns 2 x nfft / 2 complex of scratch) */
/* YGLOBAL: the 65536-point class -- the pre-emphasised signal in global memory (ywork) instead of LDS.  Two instantiations, so that
 * either form addresses ONE address space (a pointer that may be LDS or global makes every access a flat one). */
template <bool YGLOBAL>
__global__ __launch_bounds__(NTB) void srla_autocorr_big(
    SrlaJobParams jp, const int32_t *__restrict__ input, const cplx *__restrict__ twiddles, uint32_t pass,
    SrlaItemResult *__restrict__ results, double *__restrict__ lags_ws, double *__restrict__ dbg,
    const SrlaAutocorrItem *__restrict__ class_items, uint32_t count, double *__restrict__ chain_pool,
    const uint32_t *__restrict__ chain_tab, cplx *__restrict__ scratch, uint32_t scratch_stride /* cplx per workgroup */,
    int32_t *__restrict__ ywork /* 65536-point items: nfft words per workgroup in global memory instead of LDS (256 KB), else null */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int32_t *ylds;                                                /* nfft words: the pre-emphasised signal (LTP filter) */
    if constexpr (YGLOBAL) ylds = ywork + (size_t)blockIdx.x * scratch_stride; else ylds = (int32_t *)lds;
    __shared__ long long s_l[2 * (NTB / WAVE)];
    __shared__ uint32_t s_u[NTB / WAVE];
    __shared__ int32_t s_coef;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cplx *bufA = scratch + (size_t)blockIdx.x * scratch_stride, *bufB = bufA + (scratch_stride >> 1);
    for (uint32_t pos = blockIdx.x; pos < count; pos += gridDim.x) {
        const SrlaAutocorrItem it = class_items[pos];
        const InputView iv = input_view(jp, it.lshift);
        const uint32_t n = it.n, nfft = it.nfft, bps = jp.bits_per_sample, item_idx = it.item;
        const int32_t *in = input + it.sample_off;
        const bool first_pass = (pass == 1) || (jp.ltp_order == 0);
        SrlaItemResult *out = &results[item_idx];
        const bool chain = chain_pool != nullptr;
        int32_t coef;
        __syncthreads();                                           /* the previous item's last reads of LDS / scratch are done */
        if (first_pass) {
            long long r0 = 0, r1 = 0;
            uint32_t absmax = 0;
            for (uint32_t i = tid; i < n; i += NTB) {
                const long long x = load_variant(in, iv, it.variant, i);
                const long long y = (i + 1 < n) ? (long long)load_variant(in, iv, it.variant, i + 1) : 0;
                r0 += x * x; r1 += x * y;
                const uint32_t a = (x < 0) ? (uint32_t)(-x) : (uint32_t)x;
                absmax = (a > absmax) ? a : absmax;
            }
            r0 = wave_sum_i64(r0); r1 = wave_sum_i64(r1); absmax = wave_max_u32(absmax);
            if (lane == 0) { s_l[wave] = r0; s_l[NTB / WAVE + wave] = r1; s_u[wave] = absmax; }
            __syncthreads();
            long long t0 = 0, t1 = 0; uint32_t am = 0;
            for (int w = 0; w < NTB / WAVE; w++) { t0 += s_l[w]; t1 += s_l[NTB / WAVE + w]; am = (s_u[w] > am) ? s_u[w] : am; }
            uint32_t flags = (n & 1u) ? SRLA_ITEM_ODD_LENGTH : 0u;
            if (am == 0) flags |= SRLA_ITEM_INPUT_ZERO;
            if (am < (1u << 23) && t0 < (1LL << 53)) {
                const double d0 = (double)t0, d1 = (double)t1;
                int32_t c = 0;
                if (!(d0 < 1e-6)) { c = (int32_t)round_half_away((d1 / d0) * 16.0); c = (c < -16) ? -16 : ((c > 15) ? 15 : c); }
                coef = c;
            } else {
                /* srla_utility.c:226-240 literally (rounding depends on the order): one lane */
                if (tid == 0) {
                    double curr = load_variant(in, iv, it.variant, 0), succ = load_variant(in, iv, it.variant, 1);
                    double d0 = 0.0, d1 = 0.0;
                    for (uint32_t i = 0; i + 2 < n; i++) {
                        const double nn = load_variant(in, iv, it.variant, i + 2);
                        d0 += curr * curr; d1 += curr * succ; curr = succ; succ = nn;
                    }
                    d0 += curr * curr; d1 += curr * succ; curr = succ; d0 += curr * curr;
                    int32_t c = 0;
                    if (!(d0 < 1e-6)) { c = (int32_t)round_half_away((d1 / d0) * 16.0); c = (c < -16) ? -16 : ((c > 15) ? 15 : c); }
                    s_coef = c;
                }
                __syncthreads();
                coef = s_coef;
            }
            if (tid == 0) {
                out->preemph_prev = load_variant(in, iv, it.variant, 0);
                out->preemph_coef = coef;
                out->lpc_order = 0; out->lpc_rshift = 0; out->use_sum = 0; out->ltp_period = 0;
                out->ltp_coef[0] = 0; out->ltp_coef[1] = 0; out->ltp_coef[2] = 0;
                out->code_length = 0; out->res_code_type = 0; out->res_porder = 0; out->res_bits = 0;
                out->flags = flags; out->pad[0] = 0; out->pad[1] = 0;
            }
        } else {
            coef = out->preemph_coef;
            if (out->ltp_period == 0 && !chain && dbg == nullptr) continue;   /* the LPC lags are the first of the LTP lags (see srla_autocorr) */
        }
        if (pass == 0 && jp.max_order == 0 && !chain) continue;
        /* pre-emphasis (srla_utility.c:342) into LDS; optional long-term predictor (srla_lpc_predict.c:267-294) read from there */
        for (uint32_t i = tid; i < n; i += NTB) {
            const int32_t cur = load_variant(in, iv, it.variant, i);
            const int32_t prev = (i == 0) ? cur : load_variant(in, iv, it.variant, i - 1);
            ylds[i] = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)coef) >> 4));
        }
        __syncthreads();
        const uint32_t period = (pass == 0 && jp.ltp_order > 0) ? out->ltp_period : 0u;
        const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
        const int32_t c0 = out->ltp_coef[0], c1 = out->ltp_coef[1], c2 = out->ltp_coef[2];
        /* Welch window (lpc.c:256-266) on the [-1,1) normalised signal, zero padded to nfft, into the scratch buffer */
        {
            const double norm_bps = __builtin_ldexp(1.0, -(int)(bps - 1));
            const uint32_t half = n >> 1;
            double *dA = (double *)bufA;
            for (uint32_t e = tid; e < nfft; e += NTB) {
                double val = 0.0;
                if (e < n) {
                    int32_t y = ylds[e];
                    if (period > 0 && e >= period + half_order + 1) {
                        const uint32_t base = e - period - half_order;
                        uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)ylds[base];
                        if (taps == 3) acc += (uint32_t)c1 * (uint32_t)ylds[base + 1] + (uint32_t)c2 * (uint32_t)ylds[base + 2];
                        y = (int32_t)((uint32_t)y - (uint32_t)((int32_t)acc >> 5));
                    }
                    uint32_t smpl; bool touched = true;
                    if (e < half) smpl = e;
                    else if (e >= n - half) smpl = n - 1 - e;
                    else { smpl = 0; touched = false; }           /* middle sample of an odd block */
                    if (touched) {
                        const double in_d = (double)y * norm_bps;
                        const double wt = it.welch_divisor * (double)smpl * (double)(n - 1 - smpl);
                        val = in_d * wt;
                    } else if (chain && it.chain_src) {
                        val = chain_pool[it.chain_src - 1u];
                    }
                }
                dA[e] = val;
            }
        }
        __syncthreads();
        /* circular autocorrelation (lpc.c:330-376): real FFT = complex FFT of nfft / 2 points + symmetry pass, |X|^2, inverse */
        const uint32_t m = nfft >> 1, ct = complex_table_len(m), quarter = nfft >> 2;
        const cplx *twbase = twiddles + it.tw_off;
        cplx *res = fft_complex_global(bufA, bufB, m, -1, twbase);
        spectrum_power_pass<NTB, false>(res, nfft, twbase + 2 * ct, twbase + 2 * ct + quarter);
        cplx *other = (res == bufA) ? bufB : bufA;
        res = fft_complex_global(res, other, m, 1, twbase + ct);
        const uint32_t num_lags = (pass == 1) ? SRLA_LTP_LAGS : (jp.max_order + 1);
        if (chain && it.chain_dump) {
            double *dst = chain_pool + (it.chain_dump - 1u);
            for (uint32_t i = tid; i < nfft; i += NTB) { const cplx z = res[i >> 1]; dst[i] = (i & 1u) ? z.y : z.x; }
        }
        const size_t stride = jp.num_items;
        for (uint32_t i = tid; i < num_lags; i += NTB) {
            double lag = 0.0;
            if (i < nfft) { const cplx z = res[i >> 1]; lag = ((i & 1u) ? z.y : z.x) * it.acorr_norm; }
            else if (chain && it.chain_lags) {
                const uint32_t o = chain_tab[it.chain_lags - 1u + (i - nfft)];
                if (o) lag = chain_pool[o - 1u] * it.acorr_norm;
            }
            lags_ws[(size_t)i * stride + item_idx] = lag;
            if (dbg) dbg[(size_t)item_idx * SRLA_DBG_STRIDE + ((pass == 1) ? SRLA_DBG_LTPLAGS : SRLA_DBG_LAGS) + i] = lag;
        }
    }
}

/* srla_residual_cost for blocks of more than 8192 samples: ONE int32 buffer in LDS (pre-emphasised signal -> long-term
 * predictor -> FIR residual -> zig-zag residual, each rewritten in place from the top of the block down: every output reads
 * only lower indices), then the shared Rice search. */
template <bool SIG_GLOBAL /* blocks above 32768 samples: the signal in sig_ws (global) instead of LDS */>
__global__ __launch_bounds__(NT) void srla_residual_cost_big(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    const SrlaGeom *__restrict__ geoms, const double *__restrict__ rice_thresholds, int32_t *__restrict__ res_ws,
    SrlaItemResult *__restrict__ results, const uint32_t *__restrict__ big_items, uint32_t count, uint32_t sig_words,
    int32_t *__restrict__ sig_ws /* blocks above 32768 samples: sig_words words per workgroup in global memory (the signal no longer fits LDS), else null */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const size_t sig_lds = SIG_GLOBAL ? 0 : (size_t)sig_words * 4;
    int32_t *sig;                                                   /* FIR_PAD zeros, then the block */
    if constexpr (SIG_GLOBAL) sig = sig_ws + (size_t)blockIdx.x * sig_words; else sig = (int32_t *)lds;
    double *means = (double *)(lds + sig_lds);
    SmallC *sm = (SmallC *)(lds + sig_lds + 8u * 2048u);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if (blockIdx.x >= count) return;
    const uint32_t item_idx = big_items[blockIdx.x];
    const SrlaItemDesc it = items[item_idx];
    const InputView iv = input_view(jp, it.lshift);
    const SrlaGeom g = geoms[it.geom];
    const uint32_t n = it.n, bps = jp.bits_per_sample;
    const int32_t *in = input + it.sample_off;
    SrlaItemResult *out = &results[item_idx];
    const int32_t coef = out->preemph_coef;
    const uint32_t order = out->lpc_order, rshift = out->lpc_rshift, period = out->ltp_period;
    const uint32_t o4 = (order + 3u) & ~3u;
    int32_t *y = sig + FIR_PAD;
    for (uint32_t i = tid; i < FIR_PAD; i += NT) sig[i] = 0;
    for (uint32_t i = tid; i < n; i += NT) {
        const int32_t cur = load_variant(in, iv, it.variant, i);
        const int32_t prev = (i == 0) ? cur : load_variant(in, iv, it.variant, i - 1);
        y[i] = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)coef) >> 4));
    }
    for (uint32_t k = tid; k < o4; k += NT) sm->coefq[k] = (k < o4 - order) ? 0 : (int32_t)out->lpc_coef[k - (o4 - order)];
    if (tid < 16) sm->level_bits[tid] = 0;
    if (tid == 0) sm->max_u = 0;
    __syncthreads();
    const uint32_t nblk = (n + NT - 1) / NT;
    if (period > 0) {
        /* long-term predictor, srla_lpc_predict.c:267-294 */
        const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
        const int32_t c0 = out->ltp_coef[0], c1 = out->ltp_coef[1], c2 = out->ltp_coef[2];
        for (uint32_t b = nblk; b-- > 0;) {
            const uint32_t s = b * NT + tid;
            int32_t v = 0;
            const bool act = s < n && s >= period + half_order + 1;
            if (act) {
                const uint32_t base = s - period - half_order;
                uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)y[base];
                if (taps == 3) acc += (uint32_t)c1 * (uint32_t)y[base + 1] + (uint32_t)c2 * (uint32_t)y[base + 2];
                v = (int32_t)((uint32_t)y[s] - (uint32_t)((int32_t)acc >> 5));
            }
            __syncthreads();
            if (act) y[s] = v;
            __syncthreads();
        }
    }
    /* int32 wrap-around FIR (srla_lpc_predict.c:118-265), then the zig-zag residual in place */
    {
        const int32_t half = (int32_t)(1u << ((rshift - 1u) & 31u));
        int32_t *res_out = res_ws + it.res_off;
        uint32_t max_u = 0;
        for (uint32_t b = nblk; b-- > 0;) {
            const uint32_t s = b * NT + tid;
            uint32_t z = 0;
            if (s < n) {
                int32_t rv;
                if (order == 0 || s == 0) rv = y[s];
                else if (s < order) rv = (int32_t)((uint32_t)y[s] - (uint32_t)y[s - 1]);
                else {
                    uint32_t acc = (uint32_t)half;
                    const int32_t *w = y + (int)s - (int)o4;           /* FIR_PAD zeros in front cover s < o4 */
                    for (uint32_t k = 0; k < o4; k++) acc += (uint32_t)sm->coefq[k] * (uint32_t)w[k];
                    rv = (int32_t)((uint32_t)y[s] + (uint32_t)((int32_t)acc >> rshift));
                }
                res_out[s] = rv;
                z = zigzag32(rv);
                max_u = (z > max_u) ? z : max_u;
            }
            __syncthreads();
            if (s < n) y[s] = (int32_t)z;
            __syncthreads();
        }
        max_u = wave_max_u32(max_u);
        if (lane == 0) atomicMax(&sm->max_u, max_u);
    }
    __syncthreads();
    rice_search_finish((const uint32_t *)y, g, means, sm, rice_thresholds, bps, period, jp.ltp_order, out);
}

/* ================================================================================================
 * SVR refinement of the predictor (--svr-filter-learning-iteration > 0; lpc.c:1036-1136, reached from
 * srla_encoder.c:1084-1097).  Off by default and an order of magnitude more work than the rest of the analysis -- for the
 * reference as well.  One workgroup per item; everything whose value depends on the ORDER of a floating-point summation is
 * summed in the reference's order: the residual of a sample tap by tap (samples in parallel), mabse and the p entries of
 * r_vec sample by sample (one lane per running sum), the Cholesky factor and the two triangular solves row by row.
 * The covariance matrix (lpc.c:987-1020) is exact integer arithmetic whenever the products of the block cannot leave 53
 * bits (16-bit audio): then every partial sum of the reference is exact and the order is free; otherwise one thread per
 * matrix entry replays the reference's sum.
 * libm (H2): pow(x, -0.5) of the factorisation is the correctly rounded x^-1/2; log / pow of the objective
 * (lpc.c:1023-1033) are the device's -- the objective only steers comparisons, which are flagged (SRLA_ITEM_SVR_TIE) when
 * they are close enough for a last bit to matter.
 * ============================================================================================== */
#define SVR_NT 256
#define SVR_P  64           /* orders up to 64 (presets 0..4) */
#define SVR_PS 65           /* row stride of the matrix in LDS */

/* logscale: 1.0 (exact) in production; the tie tests falsify the device's log with it (SrlaJobParams) */
__device__ __forceinline__ double svr_rgr_mean_code_length(double mean_abs_error, bool *near_tie, double logscale)
{
    /* lpc.c:1023-1033 with BITS_PER_SAMPLE = 16 (:1042) */
    const double intmean = mean_abs_error * 65536.0;
    const double rho = 1.0 / (1.0 + intmean);
    const double l2 = (log(log(0.5127629514) / log(1.0 - rho)) * logscale) * 1.4426950408889634;
    const double m = (0.0 > l2) ? 0.0 : l2;
    const uint32_t k2 = (uint32_t)m;
    if (m > 0.5 && fabs(m - floor(m + 0.5)) < 1e-9) *near_tie = true;       /* the integer part hangs on log()'s last bits */
    const uint32_t k1 = k2 + 1;
    const double k1factor = pow(1.0 - rho, (double)(1u << k1));
    const double k2factor = pow(1.0 - rho, (double)(1u << k2));
    return (1.0 + k1) * (1.0 - k1factor) + (1.0 + k2 + (1.0 / (1.0 - k2factor))) * k1factor;
}

/* BIG = false: orders up to 64 and blocks up to n_cap <= 8192 samples, everything in LDS, one workgroup per item.
 * BIG = true: the other items (orders 128 / 255, blocks up to 32768 samples): block, residuals and the matrix live in a
 * region of global scratch per (persistent) workgroup, the vectors in LDS; the same code.  ws_stride: doubles per row of coef_ws. */
#define SVR_PMAX 256
template <bool BIG>
__device__ void svr_refine_item(const SrlaJobParams &jp, const int32_t *__restrict__ input, const SrlaItemDesc &it, SrlaItemResult *out,
                                double *row, const uint32_t iterations, int32_t *xi, double *rr, double *cov, const uint32_t PS,
                                double *low, double *r_vec, double *delta, double *coef, double *init_coef, double *best_coef,
                                double *s_scalar, long long *s_lag, uint32_t *s_flag, const SrlaSvrExtra ex, const uint32_t item_idx)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p = out->lpc_order;
    double *dump = (it.svr_dump != 0 && ex.chain_pool != nullptr) ? ex.chain_pool + (it.svr_dump - 1u) : nullptr;
    const double *under = (it.svr_under != 0 && ex.chain_pool != nullptr) ? ex.chain_pool + (it.svr_under - 1u) : nullptr;
    /* where the refinement does not touch the reference's buffer, its first n words stay what the LPC-pass call left */
    auto leave_untouched = [&]() { if (dump != nullptr && under != nullptr) for (uint32_t i = tid; i < it.n; i += SVR_NT) dump[i] = under[i]; };
    if (it.forced_svr != 0 && ex.forced_rows != nullptr) {
        /* the host has redone the refinement with its libm (host_ties.cpp) */
        __syncthreads();
        for (uint32_t i = tid; i < p; i += SVR_NT) row[i] = ex.forced_rows[(size_t)(it.forced_svr - 1u) * 256u + i];
        leave_untouched();
        return;
    }
    const InputView iv = input_view(jp, it.lshift);
    const uint32_t n = it.n;
    const int32_t *in = input + it.sample_off;
    const double norm = __builtin_ldexp(1.0, -(int)(jp.bits_per_sample - 1));
    __syncthreads();                                                     /* a persistent workgroup: the previous item is done with the buffers */
    /* ---- the signal the LPC analysis saw: pre-emphasis (srla_utility.c:342), long-term predictor (srla_lpc_predict.c:267) ---- */
    {
        const int32_t pc = out->preemph_coef;
        uint32_t am = 0;
        for (uint32_t i = tid; i < n; i += SVR_NT) {
            const int32_t cur = load_variant(in, iv, it.variant, i);
            const int32_t prev = (i == 0) ? cur : load_variant(in, iv, it.variant, i - 1);
            xi[i] = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)pc) >> 4));
        }
        if (tid < 4) s_flag[tid] = 0;
        __syncthreads();
        const uint32_t period = out->ltp_period;
        if (period > 0) {
            const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
            const int32_t c0 = out->ltp_coef[0], c1 = out->ltp_coef[1], c2 = out->ltp_coef[2];
            const uint32_t nblk = (n + SVR_NT - 1) / SVR_NT;
            for (uint32_t b = nblk; b-- > 0;) {                          /* in place, from the top down: every output reads lower indices only */
                const uint32_t s = b * SVR_NT + tid;
                int32_t v = 0;
                const bool act = s < n && s >= period + half_order + 1;
                if (act) {
                    const uint32_t base = s - period - half_order;
                    uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)xi[base];
                    if (taps == 3) acc += (uint32_t)c1 * (uint32_t)xi[base + 1] + (uint32_t)c2 * (uint32_t)xi[base + 2];
                    v = (int32_t)((uint32_t)xi[s] - (uint32_t)((int32_t)acc >> 5));
                }
                __syncthreads();
                if (act) xi[s] = v;
                __syncthreads();
            }
        }
        for (uint32_t i = tid; i < n; i += SVR_NT) { const int32_t v = xi[i]; const uint32_t a = (v < 0) ? (uint32_t)(-(int64_t)v) : (uint32_t)v; am = (a > am) ? a : am; }
        am = wave_max_u32(am);
        if (lane == 0) atomicMax(&s_flag[3], am);
        for (uint32_t i = tid; i < p; i += SVR_NT) { coef[i] = row[i]; init_coef[i] = row[i]; best_coef[i] = row[i]; }
        __syncthreads();
    }
    const uint32_t absmax = s_flag[3];
    const uint32_t m_len = n - p;                                        /* terms of every covariance sum */
    /* ---- covariance matrix, lpc.c:987-1020 ---- */
    if ((double)absmax * (double)absmax * (double)m_len < 4503599627370496.0 /* 2^52 */) {
        /* exact: row 0 by parallel integer dot products, the other entries by the exact recurrence along the diagonals
         * cov[i+1][j+1] = cov[i][j] - x[i] x[j] + x[m+i] x[m+j] */
        for (uint32_t d = wave; d < p; d += SVR_NT / WAVE) {
            long long acc = 0;
            for (uint32_t s0 = lane; s0 < m_len; s0 += WAVE) acc += (long long)xi[s0] * (long long)xi[s0 + d];
            acc = wave_sum_i64(acc);
            if (lane == 0) s_lag[d] = acc;
        }
        __syncthreads();
        const double scale = norm * norm;
        if (tid < p) {
            const uint32_t d = tid;
            long long v = s_lag[d];
            for (uint32_t i = 0; i + d < p; i++) {
                cov[i * PS + i + d] = (double)v * scale;
                v += (long long)xi[m_len + i] * (long long)xi[m_len + i + d] - (long long)xi[i] * (long long)xi[i + d];
            }
        }
    } else {
        /* the reference's own running sums, one thread per entry */
        const uint32_t npairs = p * (p + 1) / 2;
        for (uint32_t t = tid; t < npairs; t += SVR_NT) {
            uint32_t i = 0, r = t;
            while (r >= p - i) { r -= p - i; i++; }
            const uint32_t j = i + r;
            double acc = 0.0;
            for (uint32_t s0 = 0; s0 < m_len; s0++) acc += ((double)xi[s0 + i] * norm) * ((double)xi[s0 + j] * norm);
            cov[i * PS + j] = acc;
        }
    }
    __syncthreads();
    for (uint32_t t = tid; t < p * p; t += SVR_NT) { const uint32_t i = t / p, j = t % p; if (j > i) cov[j * PS + i] = cov[i * PS + j]; }
    __syncthreads();
    if (tid < p) cov[tid * PS + tid] *= (1.0 + 1e-5);                /* ridge, lpc.c:1067-1069 */
    __syncthreads();
    /* ---- Cholesky factorisation, lpc.c:573-600 ---- */
    for (uint32_t i = 0; i < p; i++) {
        if (tid == 0) {
            double sum = cov[i * PS + i];
            for (int k = (int)i - 1; k >= 0; k--) sum -= cov[i * PS + k] * cov[i * PS + k];
            if (sum <= 0.0) s_flag[0] = 1;
            else low[i] = inv_sqrt_cr(sum);
        }
        __syncthreads();
        if (s_flag[0]) break;
        const uint32_t j = i + 1 + tid;
        if (j < p) {
            double sum = cov[i * PS + j];
            for (int k = (int)i - 1; k >= 0; k--) sum -= cov[i * PS + k] * cov[j * PS + k];
            cov[j * PS + i] = sum * low[i];
        }
        __syncthreads();
    }
    if (s_flag[0]) {                                                     /* singular: all-zero input (lpc.c:1071-1077) */
        for (uint32_t i = tid; i < p; i += SVR_NT) row[i] = 0.0;
        leave_untouched();
        return;
    }
    /* ---- the learning loop, lpc.c:1083-1127 ---- */
    const double margins[6] = { 0.0, 1.0 / 4096, 1.0 / 1024, 1.0 / 256, 1.0 / 64, 1.0 / 16 };   /* srla_internal.c:27 */
    double min_obj = (double)FLT_MAX;                                    /* uniform: every thread keeps its own copy */
    for (int mi = 0; mi < 6; mi++) {
        const double margin = margins[mi];
        double prev_obj = (double)FLT_MAX;
        __syncthreads();
        for (uint32_t i = tid; i < p; i += SVR_NT) coef[i] = init_coef[i];
        __syncthreads();
        for (uint32_t itr = 0; itr < iterations; itr++) {
            /* residual of every sample: taps in index order (lpc.c:1098-1100), samples in parallel */
            for (uint32_t s0 = p + tid; s0 < n; s0 += SVR_NT) {
                double res = (double)xi[s0] * norm;
                for (uint32_t i = 0; i < p; i++) res += coef[i] * ((double)xi[s0 - i - 1] * norm);
                rr[s0] = res;
            }
            __syncthreads();
            /* the running sums over the samples, in sample order: r_vec[i] on lane i of wave 0, mabse on wave 1 */
            if (tid < p) {
                double acc = 0.0;
                for (uint32_t s0 = p; s0 < n; s0++) {
                    const double r = rr[s0];
                    const double a = (r > 0) ? r : -r;
                    const double t = (double)((r > 0) - (r < 0)) * (((a - margin) > 0.0) ? (a - margin) : 0.0);   /* LPC_SOFT_THRESHOLD, lpc.c:34 */
                    acc += t * ((double)xi[s0 - tid - 1] * norm);
                }
                r_vec[tid] = acc;
            } else if (tid == SVR_NT - 1) {                               /* p <= 255: this thread has no r_vec entry */
                double acc = 0.0;
                for (uint32_t s0 = p; s0 < n; s0++) { const double r = rr[s0]; acc += (r > 0) ? r : -r; }
                s_scalar[0] = acc;
            }
            __syncthreads();
            if (tid == 0) {
                bool tie = false;
                const double obj = svr_rgr_mean_code_length(s_scalar[0] / (double)n, &tie, jp.tie_logscale);
                /* cov delta = r_vec by the factor, lpc.c:605-631 */
                for (uint32_t i = 0; i < p; i++) {
                    double sum = r_vec[i];
                    for (int k = (int)i - 1; k >= 0; k--) sum -= cov[i * PS + k] * delta[k];
                    delta[i] = sum * low[i];
                }
                for (int k = (int)p - 1; k >= 0; k--) {
                    double sum = delta[k];
                    for (uint32_t j = (uint32_t)k + 1; j < p; j++) sum -= cov[j * PS + k] * delta[j];
                    delta[k] = sum * low[k];
                }
                s_scalar[1] = obj;
                if (tie) s_flag[2] = 1;
            }
            __syncthreads();
            const double obj = s_scalar[1];
            /* comparisons of objective values that differ by less than the device's log / pow can be trusted for */
            if (tid == 0) {
                const double tol = jp.tie_rel;
                if ((obj != min_obj && fabs(obj - min_obj) <= tol * fabs(obj)) || (obj != prev_obj && fabs(obj - prev_obj) <= tol * fabs(obj))
                    || fabs(fabs(prev_obj - obj) - 1e-8) <= 1e-8 * tol) s_flag[2] = 1;
            }
            if (obj < min_obj) {
                for (uint32_t i = tid; i < p; i += SVR_NT) best_coef[i] = coef[i];
                min_obj = obj;
            }
            if ((prev_obj < obj) || (fabs(prev_obj - obj) < 1e-8)) break;
            for (uint32_t i = tid; i < p; i += SVR_NT) coef[i] += delta[i];
            prev_obj = obj;
            __syncthreads();
        }
    }
    __syncthreads();
    if (dump != nullptr) {
        /* What the reference's `residual` = the calculator's persistent buffer holds now (lpc.c:1047,1095-1106): the block itself
         * below the order, from there on the soft-thresholded residual of the LAST pass that ran (rr[] still holds that pass's
         * residuals; its margin is the last of the list) -- what an odd-length or a short LTP block analysed later inherits. */
        const double margin = margins[5];
        for (uint32_t s0 = tid; s0 < n; s0 += SVR_NT) {
            double v;
            if (s0 < p) v = (double)xi[s0] * norm;
            else {
                const double r = rr[s0];
                const double a = (r > 0) ? r : -r;
                v = (double)((r > 0) - (r < 0)) * (((a - margin) > 0.0) ? (a - margin) : 0.0);
            }
            dump[s0] = v;
        }
    }
    for (uint32_t i = tid; i < p; i += SVR_NT) row[i] = best_coef[i];
    if (tid == 0 && s_flag[2]) {
        out->flags |= SRLA_ITEM_SVR_TIE;
        if (ex.ties != nullptr) (void)tie_append(ex.ties, item_idx, 2u);
    }
}


__global__ __launch_bounds__(SVR_NT) void srla_svr_refine(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    SrlaItemResult *__restrict__ results, double *__restrict__ coef_ws, uint32_t ws_stride, uint32_t iterations, uint32_t n_cap, SrlaSvrExtra ex)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int32_t *xi = (int32_t *)lds;                                        /* the pre-emphasised (+ LTP) block */
    double *rr = (double *)(lds + (((size_t)n_cap * 4 + 15) & ~(size_t)15));   /* residual of every sample under the current taps */
    double *cov = rr + n_cap;                                            /* [SVR_P][SVR_PS]: upper = matrix, lower = Cholesky factor */
    double *low = cov + SVR_P * SVR_PS, *r_vec = low + SVR_P, *delta = r_vec + SVR_P, *coef = delta + SVR_P;
    double *init_coef = coef + SVR_P, *best_coef = init_coef + SVR_P;
    __shared__ double s_scalar[2];
    __shared__ long long s_lag[SVR_P];
    __shared__ uint32_t s_flag[4];                                       /* 0: singular, 1: break, 2: near-tie, 3: |x| max */
    const uint32_t item_idx = xcd_position(blockIdx.x, jp.num_items);
    if (item_idx >= jp.num_items) return;
    if (ex.select != nullptr && ex.select[item_idx] != ex.round) return;
    SrlaItemResult *out = &results[item_idx];
    const uint32_t p = out->lpc_order;
    const SrlaItemDesc it = items[item_idx];
    if (p == 0 && it.svr_dump != 0 && it.svr_under != 0 && ex.chain_pool != nullptr)   /* order 0: no refinement (srla_encoder.c:1084) */
        for (uint32_t i = threadIdx.x; i < it.n; i += SVR_NT) ex.chain_pool[(size_t)(it.svr_dump - 1u) + i] = ex.chain_pool[(size_t)(it.svr_under - 1u) + i];
    if (p == 0 || p > SVR_P || it.n > n_cap) return;                     /* the others: srla_svr_refine_big */
    svr_refine_item<false>(jp, input, it, out, coef_ws + (size_t)item_idx * ws_stride, iterations, xi, rr, cov, SVR_PS,
                           low, r_vec, delta, coef, init_coef, best_coef, s_scalar, s_lag, s_flag, ex, item_idx);
}

/* per workgroup in `scratch`: n_max int32, n_max doubles, SVR_PMAX x (SVR_PMAX + 1) doubles */
__device__ __forceinline__ size_t srla_svr_big_scratch_bytes_dev(uint32_t n_max)
{
    return (((size_t)n_max * 4 + 15) & ~(size_t)15) + (size_t)n_max * 8 + (size_t)SVR_PMAX * (SVR_PMAX + 1) * 8;
}
extern "C" size_t srla_svr_big_scratch_bytes(uint32_t n_max)
{
    return (((size_t)n_max * 4 + 15) & ~(size_t)15) + (size_t)n_max * 8 + (size_t)SVR_PMAX * (SVR_PMAX + 1) * 8;
}

__global__ __launch_bounds__(SVR_NT) void srla_svr_refine_big(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    SrlaItemResult *__restrict__ results, double *__restrict__ coef_ws, uint32_t ws_stride, uint32_t iterations, uint32_t n_cap,
    unsigned char *__restrict__ scratch, uint32_t n_max, SrlaSvrExtra ex)
{
    __shared__ double vec[6][SVR_PMAX];
    __shared__ double s_scalar[2];
    __shared__ long long s_lag[SVR_PMAX];
    __shared__ uint32_t s_flag[4];
    unsigned char *mine = scratch + (size_t)blockIdx.x * srla_svr_big_scratch_bytes_dev(n_max);
    int32_t *xi = (int32_t *)mine;
    double *rr = (double *)(mine + (((size_t)n_max * 4 + 15) & ~(size_t)15));
    double *cov = rr + n_max;
    for (uint32_t item_idx = blockIdx.x; item_idx < jp.num_items; item_idx += gridDim.x) {
        if (ex.select != nullptr && ex.select[item_idx] != ex.round) continue;
        SrlaItemResult *out = &results[item_idx];
        const uint32_t p = out->lpc_order;
        const SrlaItemDesc it = items[item_idx];
        if (p == 0 || (p <= SVR_P && it.n <= n_cap)) continue;           /* done by srla_svr_refine */
        svr_refine_item<true>(jp, input, it, out, coef_ws + (size_t)item_idx * ws_stride, iterations, xi, rr, cov, SVR_PMAX + 1,
                              vec[0], vec[1], vec[2], vec[3], vec[4], vec[5], s_scalar, s_lag, s_flag, ex, item_idx);
    }
}

/* the quantiser and tap cost (lpc.c:1341-1405, srla_encoder.c:1141-1174) from the refined taps: one lane per item */
__global__ __launch_bounds__(WAVE) void srla_lpc_quantize_ws(SrlaJobParams jp, const double *__restrict__ coef_ws, uint32_t ws_stride,
                                                             const uint8_t *__restrict__ huff_len, SrlaItemResult *__restrict__ results,
                                                             const uint32_t *__restrict__ sel, uint32_t sel_round, uint32_t *__restrict__ ties)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int L = WAVE;
    const uint32_t lane = threadIdx.x;
    uint8_t *s_huff = lds;
    int32_t *q = (int32_t *)(lds + 512);
    for (uint32_t i = lane; i < 128; i += WAVE) ((uint32_t *)s_huff)[i] = ((const uint32_t *)huff_len)[i];
    __syncthreads();
    const uint32_t idx = blockIdx.x * WAVE + lane;
    if (idx >= jp.num_items) return;
    if (sel != nullptr && sel[idx] != sel_round) return;
    SrlaItemResult *out = &results[idx];
    const uint32_t order = out->lpc_order;
    const double *row = coef_ws + (size_t)idx * ws_stride;
    /* The refined taps carry the last bits of the refinement's pow(x, -0.5) (lpc.c:591 via :1071; the device's is the correctly
     * rounded value, glibc's within an ulp of it): where the quantiser's outcome hangs on such bits the item is flagged like an
     * objective near-tie and the host redoes the refinement with its own libm (host_ties.cpp: arbitrate_svr).  The band is the
     * near-tie threshold of the comparisons (1e-9 in production): eight orders of magnitude above what an ulp of a tap moves. */
    bool near = false;
    quantize_and_price(order, false,
                       [&](uint32_t i) -> double { return row[i]; },
                       [&](uint32_t i, int32_t v) { q[(size_t)i * L + lane] = v; },
                       [&](uint32_t i) -> int32_t { return q[(size_t)i * L + lane]; }, s_huff, out, jp.tie_rel, &near);
    if (near && order > 0 && !(out->flags & SRLA_ITEM_SVR_TIE)) {
        out->flags |= SRLA_ITEM_SVR_TIE;
        if (ties != nullptr) (void)tie_append(ties, idx, 2u);
    }
}

/* ------------------------------------------------------------------------- pricing -------- */
/* One wave per window.  Block cost: ComputeBlockSize (srla_encoder.c:1477-1546) on top of the
 * stereo decision of ComputeCoefficients (:1275-1327), one candidate per lane; path:
 * ApplyDijkstraMethod (:249-307) with its exact tie behaviour (lowest-index minimum, strict
 * improvement), the node scan and the edge relaxation spread over the lanes; partition read-back
 * (:397-421), one block record per lane. */
/* candidates of a window whose prices and end nodes are kept in LDS (8 bytes each); a window with more keeps them in a global
 * workspace (look-ahead / minimum block above 128 with a large maximum / minimum: slow, and so is everything else about such
 * parameters -- a window of `srla -e -V 7` has 57 000 candidates) */
#define SRLA_PRICE_LDS_CANDS 14400u

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, off, WAVE), hi = __shfl_xor((uint32_t)(v >> 32), off, WAVE);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        v = (o < v) ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(WAVE) void srla_price_windows(SrlaJobParams jp, const SrlaWindowDesc *__restrict__ windows,
                                                           const SrlaCandDesc *__restrict__ cands,
                                                           const SrlaItemResult *__restrict__ results,
                                                           SrlaBlockRecord *__restrict__ blocks, uint32_t lds_nodes, uint32_t lds_cands,
                                                           uint32_t *__restrict__ price_ws)
{
    NARROW_KERNEL_PRIORITY();
    /* dynamic LDS: five words per node (+ one: s_first has nodes + 1 entries), then the candidates' prices and end nodes -- of a
     * window of at most lds_cands candidates; a larger one keeps them in price_ws (two words per candidate of the job) */
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint32_t *s_cost = (uint32_t *)lds, *s_path = s_cost + lds_nodes, *s_via = s_path + lds_nodes, *s_used = s_via + lds_nodes;
    uint32_t *s_order = s_used + lds_nodes, *s_first = s_order + lds_nodes;
    const uint32_t w = blockIdx.x, lane = threadIdx.x;
    const SrlaWindowDesc wd = windows[w];
    const uint32_t nch = jp.num_channels, bps = jp.bits_per_sample, nodes = wd.num_nodes;
    const bool in_lds = wd.num_cands <= lds_cands;
    uint32_t *s_packed = in_lds ? (s_first + lds_nodes + 1u) : (price_ws + 2u * (size_t)wd.cand_base);
    uint32_t *s_nij = s_packed + (in_lds ? lds_cands : wd.num_cands);          /* node_i | node_j << 16 */

    for (uint32_t c = lane; c < wd.num_cands; c += WAVE) {
        const SrlaCandDesc cd = cands[wd.cand_base + c];
        const uint32_t raw_bytes = 11u + (bps * cd.n * nch) / 8u;
        uint32_t bytes = raw_bytes, type = SRLA_BLOCK_RAW, method = 0;
        if (cd.item_base != 0xFFFFFFFFu) {
            bool silent = true;
            if (cd.raw_silence != 0) silent = cd.raw_silence == 1u;
            else
            for (uint32_t ch = 0; ch < nch; ch++)
                if (!(results[cd.item_base + ch].flags & SRLA_ITEM_INPUT_ZERO)) { silent = false; break; }
            if (silent) { type = SRLA_BLOCK_SILENT; bytes = 11u; }
            else {
                uint32_t bits;
                if (nch == 1) { bits = results[cd.item_base].code_length; method = 0; }
                else {
                    const uint32_t l = results[cd.item_base + 0].code_length, r = results[cd.item_base + 1].code_length;
                    const uint32_t m = results[cd.item_base + nch].code_length, s = results[cd.item_base + nch + 1].code_length;
                    uint32_t len[4] = { l + r, m + s, l + s, r + s };
                    bits = len[0]; method = 0;
                    for (uint32_t k = 1; k < 4; k++) if (bits > len[k]) { bits = len[k]; method = k; }
                }
                bits += 2u;
                bits = ((bits + 7u) / 8u) * 8u;
                if (bits >= bps * cd.n * nch) { type = SRLA_BLOCK_RAW; bytes = raw_bytes; }
                else { type = SRLA_BLOCK_COMPRESS; bytes = 11u + bits / 8u; }
            }
        }
        s_packed[c] = bytes | (type << 28) | (method << 30);
        s_nij[c] = cd.node_i | (cd.node_j << 16);
    }
    for (uint32_t i = lane; i < nodes; i += WAVE) {
        s_cost[i] = (i == 0) ? 0u : SRLA_BIG_WEIGHT; s_path[i] = 0xFFFFFFFFu; s_via[i] = 0xFFFFFFFFu; s_used[i] = 0;
        s_first[i] = wd.num_cands;
    }
    if (lane == 0) s_first[nodes] = wd.num_cands;
    __threadfence();                 /* (the candidates' words may stand in global memory: price_ws) */
    __syncthreads();
    /* the candidates stand in the order of their start node (host_plan.cpp: i outer, j inner): s_first[i] = the first one leaving
     * node i, so that a step of the search looks at the edges of ITS node only (all of a window's candidates per step cost
     * nodes x candidates: 15 million visits for a window of 257 nodes) */
    for (uint32_t c = lane; c < wd.num_cands; c += WAVE) {
        const uint32_t ni = s_nij[c] & 0xFFFFu;
        if (c == 0 || (s_nij[c - 1] & 0xFFFFu) != ni) s_first[ni] = c;
    }
    __syncthreads();
    /* shortest path 0 -> nodes-1 over candidate edges */
    uint32_t target = 0;
    for (uint32_t guard = 0; guard <= nodes; guard++) {
        /* first unused node whose cost is below BIG and minimal (`mn > cost[i]`, ascending i) */
        uint64_t key = ~0ull;
        for (uint32_t i = lane; i < nodes; i += WAVE)
            if (!s_used[i] && s_cost[i] < SRLA_BIG_WEIGHT) { const uint64_t k = ((uint64_t)s_cost[i] << 11) | i; key = (k < key) ? k : key; }
        key = wave_min_u64(key);
        if (key != ~0ull) target = (uint32_t)(key & 0x7FFu);
        if (target == nodes - 1) break;
        const uint32_t base_cost = s_cost[target];
        /* relax every edge leaving `target`; its edges end on distinct nodes, so the lanes never collide */
        /* (every node but the last has the candidate (i, i + 1), so node i's run ends where node i + 1's starts) */
        for (uint32_t c = s_first[target] + lane; c < s_first[target + 1]; c += WAVE) {
            if ((s_nij[c] & 0xFFFFu) != target) continue;
            const uint32_t j = s_nij[c] >> 16;
            const uint32_t via = (s_packed[c] & 0x0FFFFFFFu) + base_cost;
            if (s_cost[j] > via) { s_cost[j] = via; s_path[j] = target; s_via[j] = c; }
        }
        if (lane == 0) s_used[target] = 1;
        __syncthreads();
    }
    /* read the partition back and emit block records in stream order */
    uint32_t count = 0;
    if (lane == 0) {
        for (uint32_t node = nodes - 1; node != 0 && s_path[node] != 0xFFFFFFFFu; node = s_path[node]) s_order[count++] = node;
    }
    count = __shfl(count, 0, WAVE);
    __syncthreads();
    for (uint32_t k = lane; k < nodes - 1; k += WAVE) {
        if (k >= count) { blocks[wd.block_base + k].valid = 0; continue; }
        const uint32_t node = s_order[k];
        const uint32_t c = s_via[node];
        const SrlaCandDesc cd = cands[wd.cand_base + c];
        const uint32_t packed = s_packed[c];
        uint32_t block_type = (packed >> 28) & 3u, bytes = packed & 0x0FFFFFFFu;
        const uint32_t ch_method = (packed >> 30) & 3u;
        if (block_type == SRLA_BLOCK_COMPRESS && nch > 2) {
            /* ComputeBlockSize prices only the first two channels (srla_encoder.c:1287-1301) and that
             * price drives the search; EncodeBlock then writes every channel and applies its RAW
             * fall-back to the size actually written (srla_encoder.c:1605-1611) */
            uint32_t bits = 2u;
            for (uint32_t ch = 2; ch < nch; ch++) bits += results[cd.item_base + ch].code_length;
            const uint32_t l = results[cd.item_base + 0].code_length, r = results[cd.item_base + 1].code_length;
            const uint32_t m = results[cd.item_base + nch].code_length, s2 = results[cd.item_base + nch + 1].code_length;
            bits += (ch_method == 0) ? l + r : (ch_method == 1) ? m + s2 : (ch_method == 2) ? l + s2 : r + s2;
            const uint32_t payload = (bits + 7u) / 8u;
            if (8u * payload >= bps * cd.n * nch) { block_type = SRLA_BLOCK_RAW; bytes = 11u + (bps * cd.n * nch) / 8u; }
            else bytes = 11u + payload;
        }
        SrlaBlockRecord *rec = &blocks[wd.block_base + (count - 1 - k)];
        rec->valid = 1;
        rec->sample_off = cd.sample_off;
        rec->n = cd.n;
        rec->block_type = block_type;
        rec->ch_method = ch_method;
        rec->bytes = bytes;
        for (uint32_t ch = 0; ch < SRLA_MAX_CH; ch++) {
            uint32_t it = 0xFFFFFFFFu;
            if (block_type == SRLA_BLOCK_COMPRESS && ch < nch) {
                it = cd.item_base + ch;
                if (nch >= 2) {
                    const uint32_t mi = cd.item_base + nch, si = cd.item_base + nch + 1;
                    if (ch == 0 && ch_method == 1) it = mi;
                    if (ch == 0 && ch_method == 3) it = si;
                    if (ch == 1 && (ch_method == 1 || ch_method == 2)) it = si;
                }
            }
            rec->item[ch] = it;
        }
        rec->seg = wd.seg; rec->price = packed & 0x0FFFFFFFu;
    }
}

/* ------------------------------------------------------------------------- pack ----------- */
/* srla_block_offsets (one workgroup per job): exclusive prefix sum of the chosen blocks' byte sizes in stream order;
 * per-window byte counts for the encode callback; then, per SEGMENT of the job (the run of windows that belong to one
 * stream): where its blocks go in the stream's output buffer -- at init_pos, or behind what the earlier jobs of the stream
 * wrote (a device-resident running offset per stream, so jobs are enqueued back to back without a host round trip) --, the
 * overflow check of SRLAEncoder_EncodeWhole (srla_encoder.c:1756-1783), and where the segment is assembled in the job's
 * staging buffer: with the 16-byte phase of its final address, so that srla_stream_out moves it with aligned 16-byte
 * accesses.  block_off[] ends up as offsets into the staging buffer. */
#define SRLA_SEGCTL_WORDS 8   /* device-side record per segment: bytes, pos, stage_off, skip, rel_start, - - - */
__global__ __launch_bounds__(NT) void srla_block_offsets(
    SrlaJobParams jp, const SrlaWindowDesc *__restrict__ windows, const SrlaBlockRecord *__restrict__ blocks,
    const SrlaItemResult *__restrict__ results, uint32_t num_slots, uint32_t *__restrict__ block_off,
    uint32_t *__restrict__ stream_pos /* per stream: [0] running offset, [1] sticky skip flag */,
    const SrlaSegDesc *__restrict__ segs, uint32_t *__restrict__ seg_ctl, uint64_t stage_addr,
    SrlaJobInfo *__restrict__ info, uint32_t *__restrict__ window_bytes, SrlaSegInfo *__restrict__ seg_info,
    const uint32_t *__restrict__ ties, SrlaTieGather tg)
{
    __shared__ uint32_t s_wave[NWAVES];
    __shared__ uint32_t s_cnt[6];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nch = jp.num_channels;
    if (tid < 6) s_cnt[tid] = 0;
    __syncthreads();
    uint32_t carry = 0;
    uint32_t nblk = 0, nraw = 0, nsil = 0, nodd = 0;
    for (uint32_t base = 0; base < num_slots; base += NT) {
        const uint32_t i = base + tid;
        uint32_t b = 0;
        if (i < num_slots && blocks[i].valid) {
            const SrlaBlockRecord *rec = &blocks[i];
            b = rec->bytes;
            nblk++;
            if (rec->block_type == SRLA_BLOCK_RAW) nraw++;
            else if (rec->block_type == SRLA_BLOCK_SILENT) nsil++;
            else for (uint32_t ch = 0; ch < nch; ch++) {
                const uint32_t f = results[rec->item[ch]].flags;
                if (f & SRLA_ITEM_ODD_LENGTH) nodd++;
            }
        }
        uint32_t incl = b;
        for (int off = 1; off < WAVE; off <<= 1) { const uint32_t t = __shfl_up(incl, off, WAVE); if (lane >= (uint32_t)off) incl += t; }
        if (lane == WAVE - 1) s_wave[wave] = incl;
        __syncthreads();
        uint32_t pre = carry, tot = 0;
        for (uint32_t k = 0; k < NWAVES; k++) { if (k < wave) pre += s_wave[k]; tot += s_wave[k]; }
        if (i < num_slots) block_off[i] = pre + incl - b;
        carry += tot;
        __syncthreads();
    }
    const uint32_t total = carry;
    {
        const uint32_t v[4] = { nblk, nraw, nsil, nodd };
        for (int k = 0; k < 4; k++) { const uint32_t s = wave_sum_u32(v[k]); if (lane == 0 && s) atomicAdd(&s_cnt[k], s); }
    }
    __syncthreads();
    /* per-window sizes + coverage: the chosen blocks of a window must tile it exactly */
    uint32_t bad = 0;
    for (uint32_t w = tid; w < jp.num_windows; w += NT) {
        const SrlaWindowDesc wd = windows[w];
        const uint32_t b0 = wd.block_base, b1 = wd.block_base + wd.num_nodes - 1;
        const uint32_t start = block_off[b0], end = (b1 < num_slots) ? block_off[b1] : total;
        window_bytes[w] = end - start;
        uint32_t covered = 0;
        for (uint32_t k = b0; k < b1; k++) { if (!blocks[k].valid) break; covered += blocks[k].n; }
        if (covered != wd.n) bad = 1;
    }
    bad = wave_max_u32(bad);
    if (lane == 0 && bad) atomicOr(&s_cnt[4], 1u);
    __syncthreads();
    const uint32_t cover_bad = s_cnt[4];
    /* segments */
    for (uint32_t sg = tid; sg < jp.num_segs; sg += NT) {
        const SrlaSegDesc sd = segs[sg];
        const SrlaWindowDesc w0 = windows[sd.first_window], w1 = windows[sd.first_window + sd.num_windows - 1];
        const uint32_t b0 = w0.block_base, b1 = w1.block_base + w1.num_nodes - 1;
        const uint32_t rel_start = block_off[b0], rel_end = (b1 < num_slots) ? block_off[b1] : total;
        const uint32_t bytes = rel_end - rel_start;
        const uint32_t pos = sd.use_init ? sd.init_pos : stream_pos[2u * sd.stream];
        uint32_t skip = sd.use_init ? 0u : stream_pos[2u * sd.stream + 1u];
        if ((uint64_t)pos + bytes > (uint64_t)sd.limit) { skip = 1; atomicOr(&s_cnt[5], SRLA_JOBERR_OVERFLOW); }
        if (cover_bad) skip = 1;
        stream_pos[2u * sd.stream] = skip ? pos : pos + bytes;
        stream_pos[2u * sd.stream + 1u] = skip;
        /* segment k starts at rel_start + 16 k + adj (adj < 16 gives it the phase of its destination): segments never overlap */
        const uint32_t phase = sd.dst ? (uint32_t)((sd.dst + pos) & 15u) : 0u;
        const uint32_t base = rel_start + 16u * sg;
        const uint32_t stage_off = base + ((phase - (uint32_t)((stage_addr + base) & 15u)) & 15u);
        uint32_t *c = seg_ctl + (size_t)SRLA_SEGCTL_WORDS * sg;
        c[0] = bytes; c[1] = pos; c[2] = stage_off; c[3] = skip; c[4] = rel_start;
        seg_info[sg].bytes = bytes; seg_info[sg].pos = pos; seg_info[sg].stage_off = stage_off; seg_info[sg].skip = skip;
    }
    __syncthreads();
    __threadfence_block();
    /* block offsets: from job-relative to staging-buffer positions */
    for (uint32_t w = tid; w < jp.num_windows; w += NT) {
        const SrlaWindowDesc wd = windows[w];
        const uint32_t *c = seg_ctl + (size_t)SRLA_SEGCTL_WORDS * wd.seg;
        const uint32_t delta = c[2] - c[4];
        for (uint32_t k = wd.block_base; k < wd.block_base + wd.num_nodes - 1; k++) block_off[k] += delta;
    }
    if (tid == 0) {
        info->total_bytes = total;
        info->base = seg_ctl[1];
        info->num_blocks = s_cnt[0];
        info->num_raw = s_cnt[1];
        info->num_silent = s_cnt[2];
        info->num_tie_items = ties ? ties[0] : 0u;
        info->num_odd_items = s_cnt[3];
        info->error = s_cnt[5] | (cover_bad ? SRLA_JOBERR_COVER : 0u);
    }
    /* the near-ties' numbers for the host (SrlaTieGather) */
    if (tg.out != nullptr && ties != nullptr) {
        const uint32_t count = ties[0];
        if (count != 0 && count <= tg.cap) {
            const uint32_t P = jp.max_order, stride = (P + 2u > 8u) ? P + 2u : 8u;
            for (uint32_t k = 0; k < count; k++) {
                const uint32_t e = ties[1 + k], item = e & 0x3FFFFFFFu, kind = e >> 30;
                double *dst = tg.out + tg.cap + (size_t)k * stride;
                if (tid == 0) tg.out[k] = (double)e;
                if (kind == 0 && item < tg.num_items && tg.err != nullptr) {
                    for (uint32_t o = tid; o <= P; o += NT) dst[o] = tg.err[(size_t)o * tg.num_items + item];
                    if (tid == 0) dst[P + 1] = (double)results[item].lpc_order;
                } else if (kind == 1 && tg.tie_data != nullptr) {
                    if (tid < 8) dst[tid] = tg.tie_data[8 * (size_t)k + tid];
                }
            }
        }
    }
}

/* srla_pack_blocks, one workgroup per chosen block: assembles the COMPLETE block of the stream -- the
 * 11-byte block header (srla_encoder.c:1583-1595, 1629-1636), the compress payload (:1368-1452: channel
 * method, pre-emphasis state, LPC order / shift / static-Huffman coded taps, LTP fields, then per channel the
 * residual code of SRLACoder_Encode, srla_coder.c:532-595: 2-bit code type, 10-bit partition order, per
 * partition the parameter -- 5 bits, then unary zig-zag deltas -- followed by the (recursive) Rice codes) or
 * the raw payload (:823-852), and the Fletcher-16 checksum (srla_utility.c:36-60) -- MSB first in LDS (bit
 * offsets from a workgroup prefix sum over the code lengths, bits merged with LDS atomic ORs), and stores it
 * at its final byte offset of the output stream with 16-byte stores.  Blocks too large for LDS are assembled in
 * a global scratch region instead (template parameter G). */
template <bool G>
__device__ __forceinline__ void put_bits(uint32_t *w, uint32_t bitpos, uint32_t value, uint32_t nbits)
{
    /* nbits in [1,32]; word k holds stream bits 32k..32k+31, most significant first */
    if (nbits < 32) value &= (1u << nbits) - 1u;
    const uint32_t wi = bitpos >> 5, o = bitpos & 31u;
    if (o + nbits <= 32u) atomicOr(&w[wi], value << (32u - o - nbits));
    else {
        const uint32_t spill = o + nbits - 32u;
        atomicOr(&w[wi], value >> spill);
        atomicOr(&w[wi + 1], value << (32u - spill));
    }
}

template <bool G>
__device__ __forceinline__ uint32_t get_word(const uint32_t *w, uint32_t i)
{
    /* the global scratch is filled by L2 atomics: read it past the (non-coherent) vector L1 */
    if (G) return __hip_atomic_load(&w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return w[i];
}

/* SRLA_MI355X_RECOMPUTE_RESIDUALS only (jp.keep_residuals == 0).  The residual of one channel of a chosen block (<= 8192
 * samples), recomputed into LDS by the block's pack workgroup -- pre-emphasis (srla_utility.c:342), long-term predictor
 * (srla_lpc_predict.c:267-294), int32 wrap-around FIR (:118-265), each stage rewriting the buffer in place from the top of
 * the block down (every output reads lower indices only).  In this mode srla_residual_cost prices every candidate x variant
 * and writes nothing but the item record; only the chosen blocks' residuals ever exist, here, on their way into the
 * bitstream.  Measured: HBM traffic of the analysis / 3, whole-job throughput -8 % (M) .. -1 % (C5), because
 * the pack workgroups hold twice the LDS for longer on a chip whose kernels are bound by VALU and LDS, not by HBM: hence an
 * option, not the default.  rl: FIR_PAD zeros, then the block. */
__device__ void pack_residual_lds(const SrlaJobParams &jp, const int32_t *__restrict__ input, const SrlaItemDesc &it,
                                  const SrlaItemResult *__restrict__ ir, int32_t *rl, int32_t *coefq)
{
    const uint32_t tid = threadIdx.x, n = it.n;
    const InputView iv = input_view(jp, it.lshift);
    const int32_t *in = input + it.sample_off;
    const bool aligned = input_aligned(in, iv);
    const int32_t pc = ir->preemph_coef;
    const uint32_t order = ir->lpc_order, rshift = ir->lpc_rshift, period = ir->ltp_period;
    const uint32_t o4 = (order + 3u) & ~3u;
    int32_t *y = rl + FIR_PAD;
    __syncthreads();                                             /* the previous channel's readers are done with the buffer */
    for (uint32_t i = tid; i < FIR_PAD; i += NT) rl[i] = 0;
    for (uint32_t i4 = 4u * tid; i4 < n; i4 += 4u * NT) {
        int32_t t4[4];
        load_chunk(in, iv, it.variant, i4, n, aligned, t4);
        int32_t prev = (i4 == 0) ? t4[0] : load_variant(in, iv, it.variant, i4 - 1);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int32_t cur = t4[i];
            t4[i] = (i4 + (uint32_t)i < n) ? (int32_t)((uint32_t)cur - (uint32_t)((int32_t)((uint32_t)prev * (uint32_t)pc) >> 4)) : 0;
            prev = cur;
        }
        *reinterpret_cast<int4 *>(y + i4) = make_int4(t4[0], t4[1], t4[2], t4[3]);
    }
    for (uint32_t k = tid; k < o4; k += NT) coefq[k] = (k < o4 - order) ? 0 : (int32_t)ir->lpc_coef[k - (o4 - order)];
    __syncthreads();
    const uint32_t nblk = (n + 4u * NT - 1u) / (4u * NT);
    if (period > 0) {
        const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
        const int32_t c0 = ir->ltp_coef[0], c1 = ir->ltp_coef[1], c2 = ir->ltp_coef[2];
        for (uint32_t b = nblk; b-- > 0;) {
            const uint32_t i4 = 4u * (b * NT + tid);
            int32_t v[4] = { 0, 0, 0, 0 };
            if (i4 < n) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t sidx = i4 + (uint32_t)i;
                    int32_t cur = y[sidx];
                    if (sidx < n && sidx >= period + half_order + 1) {
                        const uint32_t base = sidx - period - half_order;
                        uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)y[base];
                        if (taps == 3) acc += (uint32_t)c1 * (uint32_t)y[base + 1] + (uint32_t)c2 * (uint32_t)y[base + 2];
                        cur = (int32_t)((uint32_t)cur - (uint32_t)((int32_t)acc >> 5));
                    }
                    v[i] = cur;
                }
            }
            __syncthreads();
            if (i4 < n) *reinterpret_cast<int4 *>(y + i4) = make_int4(v[0], v[1], v[2], v[3]);
            __syncthreads();
        }
    }
    const int32_t half = (int32_t)(1u << ((rshift - 1u) & 31u));
    for (uint32_t b = nblk; b-- > 0;) {
        const uint32_t i4 = 4u * (b * NT + tid);
        int32_t rr[4] = { 0, 0, 0, 0 };
        if (i4 < n) {
            uint32_t acc[4] = { (uint32_t)half, (uint32_t)half, (uint32_t)half, (uint32_t)half };
            int4 cur = *reinterpret_cast<const int4 *>(y + (int)i4 - (int)o4);
            for (uint32_t kb = 0; kb < o4; kb += 4) {
                const int4 cf = *reinterpret_cast<const int4 *>(&coefq[kb]);
                const int4 nxt = *reinterpret_cast<const int4 *>(y + (int)i4 - (int)o4 + (int)kb + 4);
                const uint32_t w0 = (uint32_t)cur.x, w1 = (uint32_t)cur.y, w2 = (uint32_t)cur.z, w3 = (uint32_t)cur.w;
                const uint32_t w4 = (uint32_t)nxt.x, w5 = (uint32_t)nxt.y, w6 = (uint32_t)nxt.z;
                const uint32_t f0 = (uint32_t)cf.x, f1 = (uint32_t)cf.y, f2 = (uint32_t)cf.z, f3 = (uint32_t)cf.w;
                acc[0] += f0 * w0 + f1 * w1 + f2 * w2 + f3 * w3;
                acc[1] += f0 * w1 + f1 * w2 + f2 * w3 + f3 * w4;
                acc[2] += f0 * w2 + f1 * w3 + f2 * w4 + f3 * w5;
                acc[3] += f0 * w3 + f1 * w4 + f2 * w5 + f3 * w6;
                cur = nxt;
            }
            const int32_t y4[4] = { cur.x, cur.y, cur.z, cur.w };        /* after the tap loop: y[i4 .. i4 + 3] */
            const int32_t ym1 = (i4 > 0) ? y[i4 - 1] : 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t sidx = i4 + (uint32_t)i;
                int32_t rv;
                if (order == 0 || sidx == 0) rv = y4[i];
                else if (sidx < order) rv = (int32_t)((uint32_t)y4[i] - (uint32_t)((i == 0) ? ym1 : y4[i - 1]));
                else rv = (int32_t)((uint32_t)y4[i] + (uint32_t)((int32_t)acc[i] >> rshift));
                rr[i] = (sidx < n) ? rv : 0;
            }
        }
        __syncthreads();
        if (i4 < n) *reinterpret_cast<int4 *>(y + i4) = make_int4(rr[0], rr[1], rr[2], rr[3]);
        __syncthreads();
    }
}

template <bool G>
__device__ __forceinline__ void pack_block_body(
    const SrlaJobParams &jp, const SrlaBlockRecord *__restrict__ recp, const int32_t *__restrict__ input,
    const SrlaItemDesc *__restrict__ items, const SrlaItemResult *__restrict__ results, const int32_t *__restrict__ res_ws,
    const uint32_t *__restrict__ huff_code, const uint8_t *__restrict__ huff_len, uint32_t *w, uint32_t *aux, uint32_t *kpar,
    uint8_t *__restrict__ dst, SrlaJobInfo *__restrict__ info, int32_t *rl /* LDS for the recomputed residual, + 264 words of taps behind it */,
    uint32_t rl_samples)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    /* scalars of the record; item[] stays in memory (dynamic indexing of a by-value copy would spill) */
    const uint32_t nch = jp.num_channels, bps = jp.bits_per_sample, n = recp->n, T = recp->bytes;
    const uint32_t block_type = recp->block_type, sample_off = recp->sample_off;
    const uint32_t nwords = ((T + 3u) >> 2) + 1u;
    for (uint32_t i = tid; i < nwords; i += NT) w[i] = 0;
    __syncthreads();
    if (tid == 0) {
        put_bits<G>(w, 0, 0xFFFFu, 16);                 /* sync code */
        put_bits<G>(w, 16, T - 11u + 5u, 32);           /* size of what follows the size field */
        put_bits<G>(w, 64, block_type, 8);
        put_bits<G>(w, 72, n, 16);
    }
    uint32_t end_bits = 88;                             /* uniform: first bit behind the payload */
    if (block_type == SRLA_BLOCK_RAW) {
        /* interleaved, zig-zag mapped, big endian */
        const uint32_t count = n * nch;
        for (uint32_t i = tid; i < count; i += NT) {
            const uint32_t s = i / nch, ch = i - s * nch;
            const int32_t v = input[(size_t)ch * jp.channel_stride + sample_off + s];
            put_bits<G>(w, 88u + i * bps, zigzag32(v), bps);
        }
        end_bits = 88u + count * bps;
    } else if (block_type == SRLA_BLOCK_COMPRESS) {
        /* header bits of the payload: every thread needs their count, wave 0 writes them */
        uint32_t hdr_bits = 2u + nch * (bps + 1u + 5u);
        for (uint32_t ch = 0; ch < nch; ch++) {
            const SrlaItemResult *ir = &results[recp->item[ch]];
            hdr_bits += 8u + 4u + 1u + ir->pad[0] + 1u;
            if (ir->ltp_period > 0) hdr_bits += 1u + 8u + jp.ltp_order * 6u;
        }
        if (wave == 0) {
            uint32_t p = 88;
            if (lane == 0) put_bits<G>(w, p, recp->ch_method, 2);
            p += 2;
            if (lane < nch) {
                const SrlaItemResult *ir = &results[recp->item[lane]];
                put_bits<G>(w, p + lane * (bps + 6u), zigzag32(ir->preemph_prev), bps + 1u);
                put_bits<G>(w, p + lane * (bps + 6u) + bps + 1u, zigzag32(ir->preemph_coef), 5);
            }
            p += nch * (bps + 6u);
            for (uint32_t ch = 0; ch < nch; ch++) {
                const SrlaItemResult *ir = &results[recp->item[ch]];
                const uint32_t order = ir->lpc_order, use_sum = ir->use_sum;
                if (lane == 0) {
                    put_bits<G>(w, p, order, 8);
                    put_bits<G>(w, p + 8, ir->lpc_rshift, 4);
                    put_bits<G>(w, p + 12, use_sum, 1);
                }
                p += 13;
                for (uint32_t b0 = 0; b0 < order; b0 += WAVE) {
                    const uint32_t i = b0 + lane;
                    uint32_t code = 0, len = 0;
                    if (i < order) {
                        const int32_t c = ir->lpc_coef[i];
                        if (!use_sum || i == 0) { const uint32_t u = zigzag32(c); code = huff_code[u]; len = huff_len[u]; }
                        else { const uint32_t u = zigzag32(c + (int32_t)ir->lpc_coef[i - 1]); code = huff_code[256 + u]; len = huff_len[256 + u]; }
                    }
                    uint32_t incl = len;
                    for (int off = 1; off < WAVE; off <<= 1) { const uint32_t t = __shfl_up(incl, off, WAVE); if (lane >= (uint32_t)off) incl += t; }
                    if (len) put_bits<G>(w, p + incl - len, code, len);
                    p += __shfl(incl, WAVE - 1, WAVE);
                }
            }
            for (uint32_t ch = 0; ch < nch; ch++) {
                const SrlaItemResult *ir = &results[recp->item[ch]];
                const uint32_t period = ir->ltp_period;
                if (lane == 0) {
                    put_bits<G>(w, p, period != 0, 1);
                    if (period > 0) {
                        put_bits<G>(w, p + 1, (jp.ltp_order - 1u) / 2u, 1);
                        put_bits<G>(w, p + 2, period - SRLA_LTP_MIN_PERIOD, 8);
                        for (uint32_t i = 0; i < jp.ltp_order; i++) put_bits<G>(w, p + 10 + 6 * i, zigzag32(ir->ltp_coef[i]), 6);
                    }
                }
                p += 1u + (period > 0 ? 9u + 6u * jp.ltp_order : 0u);
            }
            if (lane == 0 && p != 88u + hdr_bits) info->error = SRLA_JOBERR_SIZE;
        }
        /* residual codes, channel after channel.  The partition parameters go to LDS first (a parameter fetched from
         * global memory at every partition boundary stalled the loops), the thread's residuals are fetched once with
         * 16-byte loads and kept in registers for both passes. */
        constexpr uint32_t CACHE = 16;                                   /* residuals a thread can keep */
        uint32_t chan_base = 88u + hdr_bits;
        for (uint32_t ch = 0; ch < nch; ch++) {
            const uint32_t item = recp->item[ch];
            const SrlaItemResult *ir = &results[item];
            const uint32_t total_bits = ir->res_bits, code_type = ir->res_code_type, porder = ir->res_porder;
            if (code_type == SRLA_CODE_ALLZERO) {
                if (tid == 0) put_bits<G>(w, chan_base, SRLA_CODE_ALLZERO, 2);
            } else {
                uint8_t *kp = reinterpret_cast<uint8_t *>(kpar) + (ch & 1u) * 1024u;
                {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(ir->kparam);
                    uint32_t *d32 = reinterpret_cast<uint32_t *>(kp);
                    for (uint32_t i = tid; i < (((1u << porder) + 3u) >> 2); i += NT) d32[i] = src[i];
                }
                /* blocks of at most 8192 samples: the residual is recomputed here (pack_residual_lds); larger ones were
                 * kept in HBM by srla_residual_cost_big */
                const int32_t *res = res_ws + items[item].res_off;
                if (n <= 8192u && !jp.keep_residuals) {
                    pack_residual_lds(jp, input, items[item], ir, rl, rl + FIR_PAD + rl_samples + 8);
                    res = rl + FIR_PAD;
                }
                const uint32_t plen = n >> porder;
                const uint32_t per = (n + NT - 1) / NT;                  /* contiguous samples per thread */
                const uint32_t s0 = (tid * per < n) ? tid * per : n, s1 = (s0 + per < n) ? (s0 + per) : n;
                const uint32_t part0 = (s0 < n) ? s0 / plen : 0;
                const bool cached = per <= CACHE && (per & 3u) == 0 && s0 + per <= n;   /* s0 is a multiple of 4: aligned */
                /* srla_residual_cost left the zig-zag mapped residual as uint16 where the block's values fit (see there) */
                const bool res_u16 = (ir->flags & SRLA_ITEM_RES_U16) != 0u;
                const uint16_t *res16 = reinterpret_cast<const uint16_t *>(res);
                uint32_t uc[CACHE];
                if (cached) {
#pragma unroll
                    for (uint32_t c = 0; c < CACHE / 4; c++) {
                        if (4 * c < per) {
                            if (res_u16) {
                                const uint2 q = *reinterpret_cast<const uint2 *>(res16 + s0 + 4 * c);
                                uc[4 * c] = q.x & 0xFFFFu; uc[4 * c + 1] = q.x >> 16; uc[4 * c + 2] = q.y & 0xFFFFu; uc[4 * c + 3] = q.y >> 16;
                            } else {
                                const int4 q = *reinterpret_cast<const int4 *>(res + s0 + 4 * c);
                                uc[4 * c] = zigzag32(q.x); uc[4 * c + 1] = zigzag32(q.y); uc[4 * c + 2] = zigzag32(q.z); uc[4 * c + 3] = zigzag32(q.w);
                            }
                        }
                    }
                }
                __syncthreads();                                         /* kp is complete */
                /* pass 1: bits this thread will emit */
                uint32_t mybits = 0;
                {
                    uint32_t part = part0, next = (part0 + 1) * plen;
                    uint32_t k = kp[part];
                    auto count = [&](uint32_t s, uint32_t u) {
                        if (s == next) { part++; next += plen; k = kp[part]; }
                        if (s == part * plen) {
                            if (part == 0) mybits += 2u + 10u + 5u;
                            else mybits += zigzag32((int32_t)k - (int32_t)kp[part - 1]) + 1u;
                        }
                        if (code_type == SRLA_CODE_RICE) mybits += 1u + k + (u >> k);
                        else mybits += (k + 2u) + (__builtin_elementwise_sub_sat(u, 2u << k) >> k);
                    };
                    if (cached) {
#pragma unroll
                        for (uint32_t i = 0; i < CACHE; i++) if (i < per) count(s0 + i, uc[i]);
                    } else {
                        for (uint32_t s = s0; s < s1; s++) count(s, res_u16 ? (uint32_t)res16[s] : zigzag32(res[s]));
                    }
                }
                /* exclusive prefix sum over the workgroup */
                uint32_t incl = mybits;
                for (int off = 1; off < WAVE; off <<= 1) { const uint32_t t = __shfl_up(incl, off, WAVE); if (lane >= (uint32_t)off) incl += t; }
                __syncthreads();                                         /* aux is reused channel after channel */
                if (lane == WAVE - 1) aux[wave] = incl;
                __syncthreads();
                uint32_t pos = chan_base + incl - mybits;
                for (uint32_t k = 0; k < wave; k++) pos += aux[k];
                /* pass 2: emit */
                {
                    uint32_t part = part0, next = (part0 + 1) * plen;
                    uint32_t k = kp[part];
                    auto emit = [&](uint32_t s, uint32_t u) {
                        if (s == next) { part++; next += plen; k = kp[part]; }
                        if (s == part * plen) {
                            if (part == 0) {
                                put_bits<G>(w, pos, (code_type << 15) | (porder << 5) | k, 17); pos += 17;   /* 2 + 10 + 5 bits */
                            } else {
                                pos += zigzag32((int32_t)k - (int32_t)kp[part - 1]);   /* zeros */
                                put_bits<G>(w, pos, 1u, 1); pos += 1;
                            }
                        }
                        if (code_type == SRLA_CODE_RICE) {
                            pos += u >> k;                                /* quotient in unary: zeros */
                            put_bits<G>(w, pos, (1u << k) | (u & ((1u << k) - 1u)), k + 1u); pos += k + 1u;
                        } else {
                            const uint32_t k1 = k + 1u, k1pow = 1u << k1;
                            if (u < k1pow) {
                                put_bits<G>(w, pos, 1u, 1); pos += 1;                       /* (2^k1 | u) in k1 + 1 bits */
                                put_bits<G>(w, pos, u, k1); pos += k1;
                            } else {
                                const uint32_t v = u - k1pow;
                                pos += 1u + (v >> k);
                                put_bits<G>(w, pos, (1u << k) | (v & ((1u << k) - 1u)), k + 1u); pos += k + 1u;
                            }
                        }
                    };
                    if (cached) {
#pragma unroll
                        for (uint32_t i = 0; i < CACHE; i++) if (i < per) emit(s0 + i, uc[i]);
                    } else {
                        for (uint32_t s = s0; s < s1; s++) emit(s, res_u16 ? (uint32_t)res16[s] : zigzag32(res[s]));
                    }
                }
            }
            chan_base += total_bits;
        }
        end_bits = chan_base;
    }
    if (tid == 0 && 11u + ((end_bits - 88u + 7u) >> 3) != T) info->error = SRLA_JOBERR_SIZE;
    __syncthreads();

    /* Fletcher-16 over bytes [8, T): the reference folds c0 += b, c1 += c0 modulo 255, i.e.
     * c0 = sum b_i, c1 = sum (L - i) b_i  (mod 255) with i counted from byte 8 and L = T - 8 */
    {
        const uint32_t L = T - 8u;
        uint64_t a = 0, ws = 0;
        for (uint32_t wi = 2u + tid; wi < nwords; wi += NT) {
            const uint32_t v = get_word<G>(w, wi);
            const uint32_t i0 = 4u * wi - 8u;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t b = (v >> (24u - 8u * k)) & 0xFFu;       /* zero beyond T */
                a += b;
                ws += (uint64_t)b * (uint64_t)(uint32_t)((L > i0 + k) ? (L - i0 - k) : 0u);
            }
        }
        uint32_t c0 = (uint32_t)(a % 255u), c1 = (uint32_t)(ws % 255u);
        c0 = wave_sum_u32(c0); c1 = wave_sum_u32(c1);
        if (lane == 0) { aux[8 + wave] = c0; aux[16 + wave] = c1; }
        __syncthreads();
        if (tid == 0) {
            uint32_t s0 = 0, s1 = 0;
            for (uint32_t k = 0; k < NWAVES; k++) { s0 += aux[8 + k]; s1 += aux[16 + k]; }
            put_bits<G>(w, 48, ((s1 % 255u) << 8) | (s0 % 255u), 16);
        }
        __syncthreads();
    }

    /* store at the block's byte offset of the stream: bytes up to the first 16-byte boundary, 16-byte body, tail */
    {
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
        uint32_t head = (16u - mis) & 15u;
        if (head > T) head = T;
        const uint32_t nvec = (T - head) >> 4;
        const uint32_t tail0 = head + (nvec << 4);
        auto byte_at = [&](uint32_t b) -> uint8_t { return (uint8_t)(get_word<G>(w, b >> 2) >> (24u - 8u * (b & 3u))); };
        if (tid < head) dst[tid] = byte_at(tid);
        if (tid >= 32 && tid - 32 < T - tail0) dst[tail0 + tid - 32] = byte_at(tail0 + tid - 32);
        const uint32_t r8 = 8u * (head & 3u);
        for (uint32_t v = tid; v < nvec; v += NT) {
            const uint32_t q = (head + (v << 4)) >> 2;
            uint32_t x[5];
#pragma unroll
            for (int k = 0; k < 5; k++) x[k] = get_word<G>(w, q + k);   /* q + 4 <= nwords - 1 */
            uint4 o;
            if (r8 == 0) { o.x = x[0]; o.y = x[1]; o.z = x[2]; o.w = x[3]; }
            else {
                o.x = (x[0] << r8) | (x[1] >> (32u - r8)); o.y = (x[1] << r8) | (x[2] >> (32u - r8));
                o.z = (x[2] << r8) | (x[3] >> (32u - r8)); o.w = (x[3] << r8) | (x[4] >> (32u - r8));
            }
            o.x = __builtin_bswap32(o.x); o.y = __builtin_bswap32(o.y); o.z = __builtin_bswap32(o.z); o.w = __builtin_bswap32(o.w);
            *reinterpret_cast<uint4 *>(dst + head + (v << 4)) = o;
        }
    }
}

__global__ __launch_bounds__(NT) void srla_pack_blocks(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    const SrlaBlockRecord *__restrict__ blocks, const SrlaItemResult *__restrict__ results,
    const int32_t *__restrict__ res_ws, const uint32_t *__restrict__ huff_code, const uint8_t *__restrict__ huff_len,
    const uint32_t *__restrict__ block_off, const uint32_t *__restrict__ seg_ctl, uint8_t *__restrict__ out,
    uint8_t *__restrict__ scratch, SrlaJobInfo *__restrict__ info, uint32_t lds_words, uint32_t rl_samples)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint32_t *aux = (uint32_t *)lds;                         /* 32 words: wave sums */
    uint32_t *kpar = aux + 32;                               /* 2 x 1024 bytes: partition parameters of the channel in work */
    int32_t *rl = (int32_t *)(kpar + 512);                   /* FIR_PAD + rl_samples + 8 words: recomputed residual; 264 words: taps */
    uint32_t *words = (uint32_t *)(rl + FIR_PAD + rl_samples + 8 + FIR_PAD + 8);   /* lds_words entries */
    const uint32_t slot = blockIdx.x;
    const SrlaBlockRecord *recp = &blocks[slot];
    if (!recp->valid || seg_ctl[(size_t)SRLA_SEGCTL_WORDS * recp->seg + 3u]) return;
    uint8_t *dst = out + block_off[slot];
    const uint32_t nwords = ((recp->bytes + 3u) >> 2) + 1u;
    if (nwords <= lds_words) {
        pack_block_body<false>(jp, recp, input, items, results, res_ws, huff_code, huff_len, words, aux, kpar, dst, info, rl, rl_samples);
    } else {
        const size_t off = ((size_t)recp->sample_off * jp.num_channels * (jp.bits_per_sample >> 3) + (size_t)slot * SRLA_PACK_SLACK + 3u) & ~(size_t)3u;
        pack_block_body<true>(jp, recp, input, items, results, res_ws, huff_code, huff_len, (uint32_t *)(scratch + off), aux, kpar, dst, info, rl, rl_samples);
    }
}

/* srla_stream_out: the job's finished bytes, device buffer -> their place in host memory (the stream's pinned buffer, or
 * the job's pinned staging buffer), segment by segment.  A handful of workgroups keep the PCIe link busy; letting the
 * 1000+ pack workgroups store to host memory themselves kept them (and their LDS) resident for the duration of the link
 * transfer and slowed the concurrently running srla_autocorr by 30 %. */
__device__ __forceinline__ void copy_same_phase(const uint8_t *__restrict__ src, uint8_t *__restrict__ d, uint32_t total,
                                                uint32_t gtid, uint32_t gsize, uint32_t pause)
{
    uint32_t head = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(d) & 15u)) & 15u;    /* src has the same phase */
    if (head > total) head = total;
    const uint32_t nvec = (total - head) >> 4, tail0 = head + (nvec << 4);
    if (gtid < head) d[gtid] = src[gtid];
    if (gtid >= 32 && gtid - 32 < total - tail0) d[tail0 + gtid - 32] = src[tail0 + gtid - 32];
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src + head);
    uint4 *d4 = reinterpret_cast<uint4 *>(d + head);
    uint32_t v = gtid;
    for (; v + 3 * gsize < nvec; v += 4 * gsize) {
        const uint4 a = s4[v], b = s4[v + gsize], c = s4[v + 2 * gsize], e = s4[v + 3 * gsize];
        d4[v] = a; d4[v + gsize] = b; d4[v + 2 * gsize] = c; d4[v + 3 * gsize] = e;
        for (uint32_t z = 0; z < pause; z++) __builtin_amdgcn_s_sleep(8);
    }
    for (; v < nvec; v += gsize) d4[v] = s4[v];
}

__global__ __launch_bounds__(NT) void srla_stream_out(const uint8_t *__restrict__ stage, const uint32_t *__restrict__ seg_ctl,
                                                      const SrlaSegDesc *__restrict__ segs, uint32_t num_segs,
                                                      uint8_t *__restrict__ host_stage, uint32_t pause)
{
    if (num_segs == 1) {
        /* the usual case: every workgroup of the launch works on the one segment */
        if (seg_ctl[3]) return;
        uint8_t *d = segs[0].dst ? reinterpret_cast<uint8_t *>(segs[0].dst) + seg_ctl[1] : host_stage + seg_ctl[2];
        copy_same_phase(stage + seg_ctl[2], d, seg_ctl[0], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, pause);
        return;
    }
    for (uint32_t sg = blockIdx.x; sg < num_segs; sg += gridDim.x) {
        const uint32_t *c = seg_ctl + (size_t)SRLA_SEGCTL_WORDS * sg;
        if (c[3]) continue;
        uint8_t *d = segs[sg].dst ? reinterpret_cast<uint8_t *>(segs[sg].dst) + c[1] : host_stage + c[2];
        copy_same_phase(stage + c[2], d, c[0], threadIdx.x, blockDim.x, pause);
    }
}

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

/* --------------------------------------------------------------------------- launchers ---- */
static SrlaLaunchTuning g_tune = {};
extern "C" void srla_set_launch_tuning(const SrlaLaunchTuning *t) { if (t) g_tune = *t; }

#define SET_LDS_ATTR(fn)                                                                                     \
    do {                                                                                                     \
        static bool done_ = false;                                                                           \
        if (!done_) { (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done_ = true; } \
    } while (0)

extern "C" int srla_launch_autocorr(hipStream_t stream, int rclass, const SrlaJobParams *jp, const int32_t *input,
                                    const SrlaItemDesc *items, const SrlaGeom *geoms, const void *twiddles,
                                    uint32_t pass, SrlaItemResult *results, double *lags_ws, double *dbg,
                                    const SrlaAutocorrItem *class_items, uint32_t count, hipEvent_t ev_start, hipEvent_t ev_stop,
                                    double *chain_pool, const uint32_t *chain_tab, int exact_nfft)
{
    if (count == 0) return 0;
    /* rclass = largest FFT size of the launch / 2048 (0: <= 1024 points).  LDS: nfft / 2 complex slots */
    const uint32_t nfft = rclass ? 2048u * (uint32_t)rclass : 1024u, m = nfft >> 1;
    const uint32_t fft_bytes = (m * 16u + 15u) & ~15u;
    const uint32_t lds = fft_bytes + srla_kernel_small_a_bytes();
    dim3 grid(8u * ((count + 7u) >> 3));
#define LAUNCH(RR, TT, FF)                                                                                   \
    do {                                                                                                     \
        SET_LDS_ATTR((srla_autocorr<RR, TT, FF>));                                                           \
        hipExtLaunchKernelGGL((srla_autocorr<RR, TT, FF>), grid, dim3(TT), lds, stream, ev_start, ev_stop, 0, *jp, input, items, geoms, \
                           (const cplx *)twiddles, fft_bytes, pass, results, lags_ws, dbg, class_items, count, chain_pool, chain_tab); \
    } while (0)
    /* SRLA_MI355X_FUSED_FFT=1: fft_complex_lds16 (two stages per LDS round trip) for 2048- and 4096-point items.  Bit-identical
     * and half the LDS cycles, but 21 % more VALU instructions and, for 2048 points, half the wavefronts per item: measured
     * 163 vs 159 us (4096) and 80 vs 61 us (2048) per launch at the metric configuration -- an option, not the default. */
    const int fused = g_tune.fused_fft ? 1 : 0;
    /* outside chain mode the items of a launch (classes of more than 1024 points; `exact_nfft`: also the 1024-point class) all
     * have the class's FFT size: the kernel with the transform's length compiled in */
    if (chain_pool == nullptr && !fused && !g_tune.generic_fft && (rclass != 0 || exact_nfft)) {
#define LAUNCH_CT(RR, TT, NF, WPV)                                                                           \
    do {                                                                                                     \
        SET_LDS_ATTR((srla_autocorr<RR, TT, false, NF, WPV>));                                               \
        hipExtLaunchKernelGGL((srla_autocorr<RR, TT, false, NF, WPV>), grid, dim3(TT), lds, stream, ev_start, ev_stop, 0, *jp, input, items, geoms, \
                           (const cplx *)twiddles, fft_bytes, pass, results, lags_ws, dbg, class_items, count, chain_pool, chain_tab); \
    } while (0)
        /* classes of at most four wavefronts: the region layout with wave-private stages (fft_regions); SRLA_MI355X_FFT_WP=0: round 4's form */
        const bool wp = g_tune.fft_wp != 0u;
        switch (rclass) {
        case 0: if (wp) LAUNCH_CT(1, 128, 1024, true); else LAUNCH_CT(1, 128, 1024, false); break;
        case 1: if (wp && g_tune.fft_thin) LAUNCH_CT(2, 128, 2048, true); else if (wp) LAUNCH_CT(1, 256, 2048, true); else LAUNCH_CT(1, 256, 2048, false); break;
        case 2: if (wp) LAUNCH_CT(2, 256, 4096, true); else LAUNCH_CT(2, 256, 4096, false); break;
        case 4: LAUNCH_CT(2, 512, 8192, false); break;
        default: return -1;
        }
#undef LAUNCH_CT
        return (hipGetLastError() == hipSuccess) ? 0 : -2;
    }
    switch (rclass * 10 + fused) {
    case 0: case 1: LAUNCH(1, 128, false); break;     /* <= 1024 points: 128 threads (one butterfly each and stage), 8 KB of LDS */
    case 10: LAUNCH(1, 256, false); break;
    case 20: LAUNCH(2, 256, false); break;
    /* 8192-point items run on 512 threads (two butterflies per thread and stage): their 70 KB of LDS allow two
     * workgroups per CU, which with 256 threads would be two wavefronts per SIMD */
    case 40: case 41: LAUNCH(2, 512, false); break;
    case 11: LAUNCH(2, 128, true); break;             /* two lanes per radix-16 unit: NTK = nfft / 16 */
    case 21: LAUNCH(2, 256, true); break;
    default: return -1;
    }
#undef LAUNCH
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" int srla_autocorr_pair_excluded(void) { return (g_tune.fused_fft || g_tune.generic_fft) ? 1 : 0; }

extern "C" int srla_launch_autocorr_pair(hipStream_t stream, const SrlaJobParams *jp, const int32_t *input, const SrlaItemDesc *items,
                                         const SrlaGeom *geoms, const void *twiddles, uint32_t pass, SrlaItemResult *results, double *lags_ws,
                                         double *dbg, const SrlaAutocorrItem *items_4096, uint32_t count_4096,
                                         const SrlaAutocorrItem *items_2048, uint32_t count_2048, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (count_4096 == 0 || count_2048 == 0) return -1;
    const uint32_t fft_bytes = 2048u * 16u, lds = fft_bytes + srla_kernel_small_a_bytes();
    dim3 grid(8u * ((count_4096 + 7u) >> 3) + 8u * ((count_2048 + 7u) >> 3));
    if (g_tune.fft_wp) {
        SET_LDS_ATTR(srla_autocorr_pair<true>);
        hipExtLaunchKernelGGL(srla_autocorr_pair<true>, grid, dim3(256), lds, stream, ev_start, ev_stop, 0, *jp, input, items, geoms, (const cplx *)twiddles,
                              fft_bytes, pass, results, lags_ws, dbg, items_4096, count_4096, items_2048, count_2048);
    } else {
        SET_LDS_ATTR(srla_autocorr_pair<false>);
        hipExtLaunchKernelGGL(srla_autocorr_pair<false>, grid, dim3(256), lds, stream, ev_start, ev_stop, 0, *jp, input, items, geoms, (const cplx *)twiddles,
                              fft_bytes, pass, results, lags_ws, dbg, items_4096, count_4096, items_2048, count_2048);
    }
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

/* The job's variant planes (SrlaJobParams::var16 / var32): every analysed variant of every sample with the offset shift applied,
 * once per job.  Four samples per thread; blockIdx.y = segment (segments of different streams may have different shifts). */
__global__ __launch_bounds__(NT) void srla_make_variants(const int32_t *__restrict__ src, uint32_t src_stride, uint32_t nch,
                                                         const uint32_t *__restrict__ lshift_dev, SrlaVarSegs segs,
                                                         int16_t *__restrict__ v16, int32_t *__restrict__ v32, uint32_t vstride,
                                                         uint32_t *__restrict__ flag)
{
    const uint32_t g = blockIdx.y;
    const uint32_t base = segs.base[g], ns = segs.ns[g], sh = lshift_dev ? *lshift_dev : segs.sh[g];
    const uint32_t i4 = 4u * (blockIdx.x * NT + threadIdx.x);
    if (i4 >= ns) return;
    const uint32_t o = base + i4, cnt = (ns - i4 < 4u) ? ns - i4 : 4u;
    const bool vec = cnt == 4u && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) && ((src_stride & 3u) == 0) && ((o & 3u) == 0);
    int32_t l[4] = { 0, 0, 0, 0 }, r[4] = { 0, 0, 0, 0 };
    auto load4 = [&](uint32_t ch, int32_t out[4]) {
        const int32_t *p = src + (size_t)ch * src_stride + o;
        if (vec) { const int4 a = *reinterpret_cast<const int4 *>(p); out[0] = a.x >> sh; out[1] = a.y >> sh; out[2] = a.z >> sh; out[3] = a.w >> sh; }
        else for (uint32_t i = 0; i < cnt; i++) out[i] = p[i] >> sh;
    };
    auto store4 = [&](uint32_t plane, const int32_t x[4], bool as32) {
        if (as32) {
            int32_t *q = v32 + (size_t)plane * vstride + o;
            if (cnt == 4u) *reinterpret_cast<int4 *>(q) = make_int4(x[0], x[1], x[2], x[3]);
            else for (uint32_t i = 0; i < cnt; i++) q[i] = x[i];
        } else {
            int16_t *q = v16 + (size_t)plane * vstride + o;
            if (cnt == 4u) *reinterpret_cast<short4 *>(q) = make_short4((short)x[0], (short)x[1], (short)x[2], (short)x[3]);
            else for (uint32_t i = 0; i < cnt; i++) q[i] = (int16_t)x[i];
        }
    };
    const bool narrow = v16 != nullptr;
    uint32_t wide = 0;
    for (uint32_t ch = 0; ch < nch; ch++) {
        int32_t x[4] = { 0, 0, 0, 0 };
        load4(ch, x);
        if (narrow) for (int i = 0; i < 4; i++) wide |= ((uint32_t)x[i] + 32768u) & 0xFFFF0000u;   /* (M of two int16 values is one) */
        store4(ch, x, !narrow);
        if (ch == 0) { l[0] = x[0]; l[1] = x[1]; l[2] = x[2]; l[3] = x[3]; }
        if (ch == 1) { r[0] = x[0]; r[1] = x[1]; r[2] = x[2]; r[3] = x[3]; }
    }
    if (nch >= 2) {
        int32_t m[4], d[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            d[i] = (int32_t)((uint32_t)r[i] - (uint32_t)l[i]);                  /* S = R - L,  M = L + (S >> 1)  (srla_utility.c:91-103) */
            m[i] = (int32_t)((uint32_t)l[i] + (uint32_t)(d[i] >> 1));
        }
        store4(nch, m, !narrow);
        store4(narrow ? 0u : nch + 1u, d, true);
    }
    if (__any((int)(wide != 0u)) && (threadIdx.x & 63u) == 0u) atomicOr(flag, 1u);
}

extern "C" int srla_launch_make_variants(hipStream_t stream, const int32_t *src, uint32_t src_stride, uint32_t nch, const uint32_t *lshift_dev,
                                         const SrlaVarSegs *segs, int16_t *v16, int32_t *v32, uint32_t vstride, uint32_t *flag)
{
    if (segs->count == 0) return 0;
    uint32_t longest = 0;
    for (uint32_t g = 0; g < segs->count; g++) longest = std::max(longest, segs->ns[g]);
    if (longest == 0) return 0;
    const uint32_t per = NT * 4u;
    hipLaunchKernelGGL(srla_make_variants, dim3((longest + per - 1) / per, segs->count), dim3(NT), 0, stream, src, src_stride, nch, lshift_dev, *segs, v16, v32, vstride, flag);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
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

extern "C" int srla_launch_pitch_solve(hipStream_t stream, const SrlaJobParams *jp, const SrlaItemDesc *items, const double *lags_ws,
                                       SrlaItemResult *results, hipEvent_t ev_start, hipEvent_t ev_stop,
                                       const uint32_t *select, uint32_t round, uint32_t *ties, double *tie_data)
{
    if (jp->num_items == 0) return 0;
    hipExtLaunchKernelGGL(srla_pitch_solve, dim3((jp->num_items + WAVE - 1) / WAVE), dim3(WAVE), 0, stream, ev_start, ev_stop, 0,
                          *jp, items, lags_ws, results, select, round, ties, tie_data);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

/* the recursion of the three-launch solve chain: order 64 in the lean form (srla_lpc_errvars_lean), smaller orders in registers */
#define ERRVARS_LEAN_L 32
#define ERRVARS_LEAN_AREG 36
template <int PP>
static void launch_errvars(hipStream_t stream, hipEvent_t ev_start, const SrlaJobParams *jp, const double *lags_ws, double *err_ws,
                           double *gamma_ws, const SrlaSvrExtra &ex)
{
    if constexpr (PP == 64) {
        /* a small job (a short stream, a piece of one) does not fill the chip: nothing to be starved by, and the register form is
         * twice as fast on its own (31 against 65 us) */
        if (!g_tune.errvars_regs && (jp->num_items >= 6144u || jp->crowded)) {
            const uint32_t lds = ((PP + 1) + (PP + 2 - ERRVARS_LEAN_AREG)) * 8 * ERRVARS_LEAN_L;
            hipExtLaunchKernelGGL((srla_lpc_errvars_lean<PP, ERRVARS_LEAN_L, ERRVARS_LEAN_AREG>), dim3((jp->num_items + ERRVARS_LEAN_L - 1) / ERRVARS_LEAN_L),
                                  dim3(WAVE), lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, gamma_ws, ex.select, ex.round);
            return;
        }
    }
    hipExtLaunchKernelGGL(srla_lpc_errvars<PP>, dim3((jp->num_items + 63) / 64), dim3(WAVE), 0, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws,
                          gamma_ws, ex.select, ex.round);
}

extern "C" int srla_launch_lpc_solve(hipStream_t stream, const SrlaJobParams *jp, const SrlaItemDesc *items,
                                     const SrlaGeom *geoms, const double *lags_ws, double *err_ws, const uint8_t *huff_len,
                                     SrlaItemResult *results, double *dbg, uint32_t *ties, hipEvent_t ev_start, hipEvent_t ev_stop,
                                     const int32_t *input, double *coef_ws, uint32_t svr_iterations, uint32_t svr_n_cap,
                                     void *svr_scratch, uint32_t svr_groups, double *gamma_ws, const SrlaSvrExtra *svr_extra)
{
    if (jp->num_items == 0) return 0;
    const uint32_t p = jp->max_order;
    SrlaSvrExtra ex = { nullptr, nullptr, nullptr, nullptr, 0u };
    if (svr_extra) ex = *svr_extra;
    const bool three = (!g_tune.solve_onepass || ex.select != nullptr) && gamma_ws != nullptr;   /* (the one-pass kernel has no round filter) */   /* orders 8 .. 64: errvars + order_select + taps */
    if (svr_iterations > 0) {
        /* solve (taps left unquantised) -> SVR refinement -> quantiser */
        const dim3 g64s((jp->num_items + 63) / 64), blks(WAVE);
        const uint32_t ws_stride = (p <= 64) ? 64u : 256u;
#define SVR_PATH(PP)                                                                                                     \
    do {                                                                                                                 \
        const uint32_t lds = 512 + PP * 8 * 64 + PP * 4 * 64;                                                            \
        if (three) {                                                                                                     \
            SET_LDS_ATTR(srla_lpc_taps<PP>);                                                                             \
            launch_errvars<PP>(stream, ev_start, jp, lags_ws, err_ws, gamma_ws, ex);                                     \
            hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blks, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round); \
            hipLaunchKernelGGL(srla_lpc_taps<PP>, g64s, blks, lds, stream, *jp, err_ws, gamma_ws, huff_len, results, coef_ws, ex.select, ex.round); \
        } else {                                                                                                         \
            SET_LDS_ATTR(srla_lpc_solve_regs<PP>);                                                                       \
            hipExtLaunchKernelGGL(srla_lpc_solve_regs<PP>, g64s, blks, lds, stream, ev_start, nullptr, 0, *jp, items, geoms, lags_ws, err_ws, \
                                  huff_len, results, dbg, ties, coef_ws);                                                \
        }                                                                                                                \
    } while (0)
        if (p == 8) SVR_PATH(8); else if (p == 16) SVR_PATH(16); else if (p == 32) SVR_PATH(32); else if (p == 64) SVR_PATH(64);
        else if (p <= 128) {
            const uint32_t lds = (2 * p + 3) * 8 * 64;
            SET_LDS_ATTR(srla_lpc_recursion<64>);
            SET_LDS_ATTR(srla_lpc_quantize<64>);
            hipExtLaunchKernelGGL(srla_lpc_recursion<64>, g64s, blks, lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, ex.select, ex.round);
            hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blks, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round);
            hipLaunchKernelGGL(srla_lpc_quantize<64>, g64s, blks, lds, stream, *jp, lags_ws, huff_len, results, coef_ws, ex.select, ex.round);
        } else {
            const uint32_t lds = (2 * p + 3) * 8 * 32;
            SET_LDS_ATTR(srla_lpc_recursion<32>);
            SET_LDS_ATTR(srla_lpc_quantize<32>);
            hipExtLaunchKernelGGL(srla_lpc_recursion<32>, dim3((jp->num_items + 31) / 32), blks, lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, ex.select, ex.round);
            hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blks, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round);
            hipLaunchKernelGGL(srla_lpc_quantize<32>, dim3((jp->num_items + 31) / 32), blks, lds, stream, *jp, lags_ws, huff_len, results, coef_ws, ex.select, ex.round);
        }
#undef SVR_PATH
        {   /* items of order <= 64 in blocks that fit LDS, whatever the preset's maximum */
            const uint32_t lds_svr = ((svr_n_cap * 4u + 15u) & ~15u) + svr_n_cap * 8u + (SVR_P * SVR_PS + 6 * SVR_P) * 8u;
            {
                static bool done_ = false;
                if (!done_) { (void)hipFuncSetAttribute((const void *)srla_svr_refine, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048); done_ = true; }
                (void)hipGetLastError();
            }
            hipLaunchKernelGGL(srla_svr_refine, dim3(8u * ((jp->num_items + 7u) >> 3)), dim3(SVR_NT), lds_svr, stream, *jp, input, items, results, coef_ws,
                               ws_stride, svr_iterations, svr_n_cap, ex);
        }
        if (p > 64 || jp->max_block > svr_n_cap) {
            /* orders 128 / 255, blocks above 8192 samples: persistent workgroups on global scratch */
            if (svr_scratch == nullptr || svr_groups == 0) return -1;
            const uint32_t groups = jp->num_items < svr_groups ? jp->num_items : svr_groups;
            hipLaunchKernelGGL(srla_svr_refine_big, dim3(groups), dim3(SVR_NT), 0, stream, *jp, input, items, results, coef_ws,
                               ws_stride, svr_iterations, svr_n_cap, (unsigned char *)svr_scratch, jp->max_block, ex);
        }
        SET_LDS_ATTR(srla_lpc_quantize_ws);
        hipExtLaunchKernelGGL(srla_lpc_quantize_ws, g64s, blks, 512 + (p < 64 ? 64 : p) * 4 * 64, stream, nullptr, ev_stop, 0, *jp, coef_ws, ws_stride, huff_len, results, ex.select, ex.round, ex.ties);
        return (hipGetLastError() == hipSuccess) ? 0 : -2;
    }
    const dim3 g64((jp->num_items + 63) / 64), blk(WAVE);
    /* orders 8 .. 64: the whole chain in one launch (srla_lpc_solve_regs) */
#define REGS_PATH(PP)                                                                                                    \
    do {                                                                                                                 \
        const uint32_t lds = 512 + PP * 8 * 64 + PP * 4 * 64;                                                            \
        if (three) {                                                                                                     \
            SET_LDS_ATTR(srla_lpc_taps<PP>);                                                                             \
            launch_errvars<PP>(stream, ev_start, jp, lags_ws, err_ws, gamma_ws, ex);                                     \
            hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blk, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round); \
            hipExtLaunchKernelGGL(srla_lpc_taps<PP>, g64, blk, lds, stream, nullptr, ev_stop, 0, *jp, err_ws, gamma_ws, huff_len, results, (double *)nullptr, ex.select, ex.round); \
        } else {                                                                                                         \
            SET_LDS_ATTR(srla_lpc_solve_regs<PP>);                                                                       \
            hipExtLaunchKernelGGL(srla_lpc_solve_regs<PP>, g64, blk, lds, stream, ev_start, ev_stop, 0, *jp, items, geoms, lags_ws, err_ws, \
                                  huff_len, results, dbg, ties, (double *)nullptr);                                      \
        }                                                                                                                \
    } while (0)
#define LDS_PATH(LL)                                                                                                     \
    do {                                                                                                                 \
        const uint32_t lds = (2 * p + 3) * 8 * LL;                                                                       \
        const dim3 gl((jp->num_items + LL - 1) / LL);                                                                    \
        SET_LDS_ATTR(srla_lpc_recursion<LL>);                                                                            \
        SET_LDS_ATTR(srla_lpc_quantize<LL>);                                                                             \
        hipExtLaunchKernelGGL(srla_lpc_recursion<LL>, gl, blk, lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, ex.select, ex.round); \
        hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blk, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round); \
        hipExtLaunchKernelGGL(srla_lpc_quantize<LL>, gl, blk, lds, stream, nullptr, ev_stop, 0, *jp, lags_ws, huff_len, results, (double *)nullptr, ex.select, ex.round); \
    } while (0)
    if (g_tune.solve_lds == 8 && p <= 64) LDS_PATH(8);
    else if (g_tune.solve_lds == 16 && p <= 64) LDS_PATH(16);
    else if (g_tune.solve_lds == 32 && p <= 64) LDS_PATH(32);
    else if (p == 8) REGS_PATH(8);
    else if (p == 16) REGS_PATH(16);
    else if (p == 32) REGS_PATH(32);
    else if (p == 64) REGS_PATH(64);
    else if (p <= 128) {
        const uint32_t lds = (2 * p + 3) * 8 * 64;
        SET_LDS_ATTR(srla_lpc_recursion<64>);
        SET_LDS_ATTR(srla_lpc_quantize<64>);
        hipExtLaunchKernelGGL(srla_lpc_recursion<64>, g64, blk, lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, ex.select, ex.round);
        hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blk, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round);
        hipExtLaunchKernelGGL(srla_lpc_quantize<64>, g64, blk, lds, stream, nullptr, ev_stop, 0, *jp, lags_ws, huff_len, results, (double *)nullptr, ex.select, ex.round);
    } else {
        const uint32_t lds = (2 * p + 3) * 8 * 32;
        SET_LDS_ATTR(srla_lpc_recursion<32>);
        SET_LDS_ATTR(srla_lpc_quantize<32>);
        hipExtLaunchKernelGGL(srla_lpc_recursion<32>, dim3((jp->num_items + 31) / 32), blk, lds, stream, ev_start, nullptr, 0, *jp, lags_ws, err_ws, ex.select, ex.round);
        hipLaunchKernelGGL(srla_order_select, dim3(8u * ((jp->num_items + 7u) >> 3)), blk, 0, stream, *jp, items, geoms, err_ws, results, dbg, ties, ex.select, ex.round);
        hipExtLaunchKernelGGL(srla_lpc_quantize<32>, dim3((jp->num_items + 31) / 32), blk, lds, stream, nullptr, ev_stop, 0, *jp, lags_ws, huff_len, results, (double *)nullptr, ex.select, ex.round);
    }
#undef REGS_PATH
#undef LDS_PATH
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" int srla_launch_residual_cost(hipStream_t stream, int rclass, const SrlaJobParams *jp, const int32_t *input,
                                         const SrlaItemDesc *items, const SrlaGeom *geoms, const SrlaLdsPlan *plan,
                                         const double *rice_thresholds, int32_t *res_ws, SrlaItemResult *results,
                                         hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (jp->num_items == 0) return 0;
    dim3 grid(8u * ((jp->num_items + 7u) >> 3)), block(NT);
#define LAUNCH(RR, MM)                                                                                       \
    do {                                                                                                     \
        SET_LDS_ATTR((srla_residual_cost<RR, MM>));                                                          \
        hipExtLaunchKernelGGL((srla_residual_cost<RR, MM>), grid, block, plan->total, stream, ev_start, ev_stop, 0, *jp, input, items, geoms, \
                           *plan, rice_thresholds, res_ws, results);                                         \
    } while (0)
    /* SRLA_MI355X_FIR_MFMA: the FIR of blocks of at most 4096 samples on the matrix pipe */
    const bool mf = g_tune.fir_mfma != 0u;
    switch (rclass) {
    case 1: if (mf) LAUNCH(1, true); else LAUNCH(1, false); break;
    case 2: if (mf) LAUNCH(2, true); else LAUNCH(2, false); break;
    case 4: LAUNCH(4, false); break;
    default: return -1;
    }
#undef LAUNCH
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" int srla_launch_autocorr_big(hipStream_t stream, const SrlaJobParams *jp, const int32_t *input, const void *twiddles, uint32_t pass,
                                        SrlaItemResult *results, double *lags_ws, double *dbg, const SrlaAutocorrItem *class_items,
                                        uint32_t count, uint32_t nfft, hipEvent_t ev_start, hipEvent_t ev_stop, double *chain_pool,
                                        const uint32_t *chain_tab, void *scratch, uint32_t scratch_groups)
{
    if (count == 0) return 0;
    const uint32_t groups = std::min(count, scratch_groups);
    {
        /* the kernel also has a few hundred bytes of static LDS: ask for what is left of the 160 KB */
        static bool done_ = false;
        if (!done_) { (void)hipFuncSetAttribute((const void *)srla_autocorr_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); done_ = true; }
        (void)hipGetLastError();
    }
    /* (scratch: scratch_groups regions of nfft complex words -- the two transform buffers --, then, for 65536 points, scratch_groups
     * regions of nfft int32 words for the signal that no longer fits LDS) */
    int32_t *ywork = (nfft > 32768u) ? (int32_t *)((cplx *)scratch + (size_t)scratch_groups * nfft) : nullptr;
    if (ywork)
        hipExtLaunchKernelGGL(srla_autocorr_big<true>, dim3(groups), dim3(NTB), 16u, stream, ev_start, ev_stop, 0, *jp, input, (const cplx *)twiddles, pass,
                              results, lags_ws, dbg, class_items, count, chain_pool, chain_tab, (cplx *)scratch, nfft, ywork);
    else
        hipExtLaunchKernelGGL(srla_autocorr_big<false>, dim3(groups), dim3(NTB), nfft * 4u, stream, ev_start, ev_stop, 0, *jp, input, (const cplx *)twiddles, pass,
                              results, lags_ws, dbg, class_items, count, chain_pool, chain_tab, (cplx *)scratch, nfft, ywork);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" uint32_t srla_residual_big_sig_words(uint32_t max_n) { return FIR_PAD + ((max_n + 3u) & ~3u) + 8u; }

extern "C" int srla_launch_residual_cost_big(hipStream_t stream, const SrlaJobParams *jp, const int32_t *input, const SrlaItemDesc *items,
                                              const SrlaGeom *geoms, const double *rice_thresholds, int32_t *res_ws, SrlaItemResult *results,
                                              const uint32_t *big_items, uint32_t count, uint32_t max_n, hipEvent_t ev_start, hipEvent_t ev_stop,
                                              int32_t *sig_ws)
{
    if (count == 0) return 0;
    const uint32_t sig_words = srla_residual_big_sig_words(max_n);
    if (max_n > 32768u && sig_ws == nullptr) return -1;
    if (max_n <= 32768u) sig_ws = nullptr;
    const uint32_t lds = (sig_ws ? 0u : sig_words * 4u) + 8u * 2048u + srla_kernel_small_c_bytes();
    SET_LDS_ATTR(srla_residual_cost_big<false>);
    if (sig_ws)
        hipExtLaunchKernelGGL(srla_residual_cost_big<true>, dim3(count), dim3(NT), lds, stream, ev_start, ev_stop, 0, *jp, input, items, geoms, rice_thresholds,
                              res_ws, results, big_items, count, sig_words, sig_ws);
    else
        hipExtLaunchKernelGGL(srla_residual_cost_big<false>, dim3(count), dim3(NT), lds, stream, ev_start, ev_stop, 0, *jp, input, items, geoms, rice_thresholds,
                              res_ws, results, big_items, count, sig_words, sig_ws);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" uint32_t srla_price_lds_cands(void) { return SRLA_PRICE_LDS_CANDS; }

extern "C" int srla_launch_price(hipStream_t stream, const SrlaJobParams *jp, const SrlaWindowDesc *windows,
                                 const SrlaCandDesc *cands, const SrlaItemResult *results,
                                 SrlaBlockRecord *blocks, hipEvent_t ev_start, hipEvent_t ev_stop,
                                 uint32_t max_nodes, uint32_t max_window_cands, uint32_t *price_ws)
{
    if (jp->num_windows == 0) return 0;
    /* LDS for the largest window of the job: its nodes, and its candidates where they fit (else price_ws, which the caller sized
     * for two words per candidate of the job) */
    const uint32_t lds_nodes = max_nodes, lds_cands = (max_window_cands <= SRLA_PRICE_LDS_CANDS) ? max_window_cands : 0u;
    if (max_window_cands > SRLA_PRICE_LDS_CANDS && price_ws == nullptr) return -1;
    const uint32_t lds = (6u * lds_nodes + 1u + 2u * lds_cands) * 4u + 16u;
    SET_LDS_ATTR(srla_price_windows);
    hipExtLaunchKernelGGL(srla_price_windows, dim3(jp->num_windows), dim3(WAVE), lds, stream, ev_start, ev_stop, 0,
                       *jp, windows, cands, results, blocks, lds_nodes, lds_cands, price_ws);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

/* LDS words the pack kernel gets for one block; larger blocks (see srla_pack_needs_scratch) are assembled in a
 * global scratch region.  SRLA_MI355X_PACK_LDS_WORDS lowers the cap (tests use it to reach the global path). */
static uint32_t pack_lds_cap()
{
    const uint32_t cap = 24 * 1024;                         /* <= 96 KB */
    return (g_tune.pack_lds_cap_words >= 8u && g_tune.pack_lds_cap_words < cap) ? g_tune.pack_lds_cap_words : cap;
}

extern "C" uint32_t srla_pack_lds_words(const SrlaJobParams *jp)
{
    const uint64_t bytes = 11ull + ((uint64_t)jp->bits_per_sample * jp->max_block * jp->num_channels) / 8;
    return (uint32_t)std::min<uint64_t>((bytes + 3) / 4 + 1, pack_lds_cap());
}

extern "C" int srla_pack_needs_scratch(const SrlaJobParams *jp)
{
    const uint64_t bytes = 11ull + ((uint64_t)jp->bits_per_sample * jp->max_block * jp->num_channels) / 8;
    return ((bytes + 3) / 4 + 1) > pack_lds_cap();
}

extern "C" int srla_launch_pack(hipStream_t stream, const SrlaJobParams *jp, uint32_t num_slots,
                                const int32_t *input, const SrlaItemDesc *items, const SrlaWindowDesc *windows,
                                const SrlaBlockRecord *blocks, const SrlaItemResult *results, const int32_t *res_ws,
                                const uint32_t *huff_code, const uint8_t *huff_len, uint32_t *block_off,
                                uint32_t *stream_pos, const SrlaSegDesc *segs, uint32_t *seg_ctl,
                                uint8_t *stage, uint8_t *host_stage, uint8_t *scratch, SrlaJobInfo *info,
                                uint32_t *window_bytes, SrlaSegInfo *seg_info, const uint32_t *ties,
                                hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t out_boost, hipStream_t out_stream, hipEvent_t ev_packed,
                                uint32_t no_stream_out, const SrlaTieGather *gather)
{
    if (num_slots == 0) return 0;
    SrlaTieGather tg{};
    if (gather != nullptr) tg = *gather;
    hipExtLaunchKernelGGL(srla_block_offsets, dim3(1), dim3(NT), 0, stream, ev_start, nullptr, 0, *jp, windows, blocks, results, num_slots, block_off,
                       stream_pos, segs, seg_ctl, (uint64_t)reinterpret_cast<uintptr_t>(stage), info, window_bytes, seg_info, ties, tg);
    const uint32_t lds_words = srla_pack_lds_words(jp);
    const uint32_t rl_samples = jp->keep_residuals ? 0u : ((jp->max_block < 8192u ? jp->max_block : 8192u) + 3u) & ~3u;
    const uint32_t lds = (lds_words + 32 + 512 + (FIR_PAD + rl_samples + 8) + (FIR_PAD + 8)) * 4;
    SET_LDS_ATTR(srla_pack_blocks);
    hipExtLaunchKernelGGL(srla_pack_blocks, dim3(num_slots), dim3(NT), lds, stream, nullptr, no_stream_out ? ev_stop : nullptr, 0,
                          *jp, input, items, blocks, results, res_ws, huff_code, huff_len, block_off, seg_ctl, stage, scratch, info,
                          lds_words, rl_samples);
    if (no_stream_out) return (hipGetLastError() == hipSuccess) ? 0 : -2;
    /* Two workgroups (three for 24-bit streams, which carry more bytes).  One moves a 4 M-sample job's 6.5 MB in 0.45-0.57 ms
     * beside the other kernels -- longer than the job's wide kernels take (0.46 ms), so the block assembly stream, not stream W,
     * set the pace of a long stream (kernel trace of round 3); two take 0.25 ms.  More slow srla_autocorr down through the
     * PCIe write path's back-pressure (4: 0.25 -> 0.30 ms per job) and lose more than they gain.  A stream of its own for this
     * kernel (so that it runs beside the next job's assembly) was measured again and is worse by 13 %: a fifth compute
     * queue serialises with the others. */
    const uint32_t wgs = g_tune.out_wgs ? g_tune.out_wgs : (jp->bits_per_sample > 16 ? 3u : 2u);
    hipStream_t os = stream;
    if (out_stream != nullptr && ev_packed != nullptr) {
        if (hipEventRecord(ev_packed, stream) != hipSuccess || hipStreamWaitEvent(out_stream, ev_packed, 0) != hipSuccess) return -2;
        os = out_stream;
    }
    hipExtLaunchKernelGGL(srla_stream_out, dim3(wgs * (out_boost ? out_boost : 1u)), dim3(NT), 0, os, nullptr, ev_stop, 0,
                          stage, seg_ctl, segs, jp->num_segs, host_stage, 0u);
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
