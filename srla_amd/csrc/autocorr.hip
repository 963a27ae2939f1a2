/*
 * autocorr.hip -- srla_autocorr / srla_autocorr_pair / srla_autocorr_big: the analysis front end of one item (exact integer
 * correlations -> pre-emphasis tap -> pre-emphasis (-> long-term predictor) -> Welch window) and its circular autocorrelation by
 * the reference's real FFT (libs/fft: radix-4 Stockham butterflies in the reference's operation order, fp64, no FMA) -> lags.
 * One workgroup per item, one launch per FFT-size class; the transform in ONE LDS buffer (classes up to 8192 points) or ping-ponging
 * between two global buffers (16384 .. 65536 points).  DESIGN.md 3.1.
 */
#include "kernels_common.h"
SRLA_DIAG_PHASE_READER(autocorr)

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
    constexpr bool SWZ12 = M >= 128;      /* the first stage's outputs permuted in LDS, see below */
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
__device__ __forceinline__ uint32_t fft_swz16(uint32_t e) { return e ^ ((e >> 4) & 7u); }    /* the eight-wavefront class, see fft_subregions */

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
template <int R, int NTK, int M, int FLAG, int SWZ /* 0: natural order, 1: at fft_swz, 2: at fft_swz16 (the eight-wavefront class) */>
__device__ __forceinline__ void fft_first_stage_regions(cplx *x, const cplx *__restrict__ tw)
{
    constexpr uint32_t QM = (uint32_t)M >> 2;
    static_assert(R * NTK == (int)QM, "one first-stage butterfly per thread and r");
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t bf = threadIdx.x + (uint32_t)r * NTK;
        const cplx *t = tw + bf;
        const cplx w1 = t[0], w2 = t[QM], w3 = t[2 * QM];
        cplx *xi = x + ((SWZ == 2) ? fft_swz16(bf) : ((SWZ == 1) ? fft_swz(bf) : bf));
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

/* ---- The same idea one level deeper, for the class of EIGHT wavefronts (round 6: 8192 points, M = 4096 complex on 512 threads) ----
 * Four regions cannot be dealt to eight wavefronts.  But the split repeats: after the SECOND radix-4 stage the data falls into
 * sixteen independent sub-transforms by index mod 16 (a stage with stride s >= 16 keeps q = index mod s, fft.c:71-128), and
 * inside a region the second stage is "the region's first stage", which is in place again on the region layout: butterfly j of
 * region rho reads the region's positions j + k QM/4 and leaves output k = position 4 j + k where input k stood -- slot
 * k QM/4 + j, i.e. position j of SUB-REGION k.  So: element e of the transform stands, after two stages, at
 *     (e & 3) QM + ((e >> 2) & 3) SQ + (e >> 4)        QM = M / 4, SQ = M / 16 (256 complex points),
 * every wavefront owns two whole sub-regions, and stages 3 .. last of either direction run without a workgroup barrier
 * (fft_subregions: the butterflies of an ordinary 256-point transform with the table entries of the M-point one -- the twiddle
 * index p of butterfly jj is the same number).  Barriers per item: behind the first and the second stage of either direction,
 * in front of / inside / behind the spectrum pass, in front of the lag stores -- 8 instead of 26.
 * The inverse's input is written by the spectrum pass, whose lanes run along a sub-region (bins 16 apart): natural order
 * permuted by fft_swz16(e) = e ^ ((e >> 4) & 7) puts the eight lanes of a 16-byte store on eight columns; the inverse's first two
 * stages work in place on that order (k QM and k SQ do not reach bits 0 .. 6), and its first wave-private stage reads position
 * jj + 64 k of its sub-region at 64 k + (fft_swz(jj with bits 4, 5) ^ 4 (k & 1)) -- see fft_subregions.
 * Same butterflies, same operands, same order of operations: the same bits (tests/test_kernel_models.py replays the addresses). */

/* the second stage, in place inside every region, all wavefronts; one barrier behind.  SWZ: the regions' positions stand at fft_swz16 */
template <int R, int NTK, int M, int FLAG, bool PRUNE, bool SWZ>
__device__ __forceinline__ void fft_second_stage_regions(cplx *x, const cplx *__restrict__ tw, const uint32_t need)
{
    constexpr uint32_t QM = (uint32_t)M >> 2, SQ = (uint32_t)M >> 4;
    static_assert(R * NTK == (int)QM, "one second-stage butterfly per thread and r");
    constexpr uint32_t twoff = 3u * ((uint32_t)M >> 2), n1 = (uint32_t)M >> 4;     /* behind the first stage's tables; entries per table */
    const bool k1 = !PRUNE || 4u < need, k2 = !PRUNE || 8u < need, k3 = !PRUNE || 12u < need;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t u = threadIdx.x + (uint32_t)r * NTK;
        const uint32_t rho = u / SQ, j = u % SQ;                                      /* q = rho, p = j */
        if (!PRUNE || rho < need) {
            const cplx *t = tw + twoff + j;
            cplx w1, w2, w3;
            if (k1) w1 = t[0];
            if (k2) w2 = t[n1];
            if (k3) w3 = t[2 * n1];
            cplx *xi = x + rho * QM + (SWZ ? fft_swz16(j) : j);
            const cplx a = xi[0], b = xi[SQ], c = xi[2 * SQ], d = xi[3 * SQ];
            const cplx apc = c_add(a, c), amc = c_sub(a, c), bpd = c_add(b, d), bmd = c_sub(b, d);
            const cplx jbmd = (FLAG < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
            xi[0] = c_add(apc, bpd);
            if (k1) xi[SQ] = c_mul(w1, c_sub(amc, jbmd));
            if (k2) xi[2 * SQ] = c_mul(w2, c_sub(apc, bpd));
            if (k3) xi[3 * SQ] = c_mul(w3, c_add(amc, jbmd));
        }
    }
    __syncthreads();
}

/* stages 3 .. last on the sub-region layout, wave-private.  Lane -> butterfly: u = lane + 64 r; the wavefront's GPW = 16 / (NTK / 64)
 * sub-regions have BPG = M / 64 butterflies each per stage: sub-region g = wave GPW + u / BPG (region g >> 2, residue 4 (g & 3) + (g >> 2)
 * of the element index mod 16), butterfly jj = u % BPG.  INSWZ (the inverse): the sub-regions' positions stand at fft_swz16. */
template <int R, int NTK, int M, int FLAG, bool PRUNE, bool INSWZ>
__device__ __forceinline__ void fft_subregions(cplx *x, const cplx *__restrict__ tw, const uint32_t need)
{
    constexpr int NW = NTK / 64, GPW = 16 / NW;
    constexpr uint32_t SQ = (uint32_t)M >> 4, BPG = (uint32_t)M >> 6, SUB4 = (uint32_t)M >> 6;   /* sub-region size; butterflies per sub-region and stage; quarter of a sub-region */
    static_assert(NW >= 1 && NW <= 16 && GPW * (int)BPG == 64 * R, "every wavefront owns whole sub-regions");
    static_assert(SUB4 >= 64, "the swizzles below leave k SUB4 alone");
    constexpr int NST = (M >= 4096) ? 6 : ((M >= 1024) ? 5 : ((M >= 256) ? 4 : 3));
    static_assert(((uint32_t)M >> (2 * NST)) == 1, "log2 M even (no closing radix-2 stage): 4096 points");
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t twoff = 3u * ((uint32_t)M >> 2) + 3u * ((uint32_t)M >> 4);           /* behind the first two stages' tables */
#pragma unroll
    for (int st = 2; st < NST; st++) {
        const uint32_t n = (uint32_t)M >> (2 * st), s = 1u << (2 * st), sl = s >> 4, log2sl = 2u * (uint32_t)(st - 2);
        const uint32_t n1 = n >> 2;
        const bool k1 = !PRUNE || s < need, k2 = !PRUNE || 2 * s < need, k3 = !PRUNE || 3 * s < need;
        cplx a[R], b[R], c[R], d[R], w1[R], w2[R], w3[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t u = lane + 64u * (uint32_t)r;
            const uint32_t g = wave * (uint32_t)GPW + u / BPG, jj = u % BPG;
            const uint32_t ql = jj & (sl - 1), qres = 4u * (g & 3u) + (g >> 2);
            if (!PRUNE || 16u * ql + qres < need) {
                const uint32_t p = jj >> log2sl;
                const cplx *t = tw + twoff + p;
                if (k1) w1[r] = t[0];
                if (k2) w2[r] = t[n1];
                if (k3) w3[r] = t[2 * n1];
                if (INSWZ && st == 2) {
                    /* position jj + SUB4 k stands at fft_swz16 of it: SUB4 k + ((jj ^ ((jj >> 4) & 3)) ^ 4 (k & 1))   (SUB4 = 64) */
                    const uint32_t pe = jj ^ ((jj >> 4) & 3u);
                    const cplx *xe = x + g * SQ + pe, *xo = x + g * SQ + (pe ^ 4u);
                    a[r] = xe[0]; b[r] = xo[SUB4]; c[r] = xe[2 * SUB4]; d[r] = xo[3 * SUB4];
                } else {
                    const cplx *xi = x + g * SQ + ((st == 3) ? fft_swz(jj) : jj);
                    a[r] = xi[0]; b[r] = xi[SUB4]; c[r] = xi[2 * SUB4]; d[r] = xi[3 * SUB4];
                }
            }
        }
        /* (no barrier: the wavefront's own loads above are executed before its stores below) */
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t u = lane + 64u * (uint32_t)r;
            const uint32_t g = wave * (uint32_t)GPW + u / BPG, jj = u % BPG;
            const uint32_t ql = jj & (sl - 1), qres = 4u * (g & 3u) + (g >> 2);
            if (!PRUNE || 16u * ql + qres < need) {
                const cplx apc = c_add(a[r], c[r]), amc = c_sub(a[r], c[r]), bpd = c_add(b[r], d[r]);
                const cplx bmd = c_sub(b[r], d[r]);
                const cplx jbmd = (FLAG < 0) ? make_double2(-bmd.y, bmd.x) : make_double2(bmd.y, -bmd.x);
                if (st == 2) {
                    /* the sub-region's first stage: a thread's four outputs are the consecutive positions 4 jj .. 4 jj + 3, permuted among
                     * themselves as in fft_regions (position e goes to fft_swz(e)); the next stage reads at fft_swz */
                    const uint32_t kx = (jj >> 1) & 3u;
                    cplx *xo = x + g * SQ + 4u * jj;
                    xo[kx] = c_add(apc, bpd);
                    if (k1) xo[1u ^ kx] = c_mul(w1[r], c_sub(amc, jbmd));
                    if (k2) xo[2u ^ kx] = c_mul(w2[r], c_sub(apc, bpd));
                    if (k3) xo[3u ^ kx] = c_mul(w3[r], c_add(amc, jbmd));
                } else {
                    cplx *xo = x + g * SQ + (4u * jj - 3u * ql);
                    xo[0] = c_add(apc, bpd);
                    if (k1) xo[sl] = c_mul(w1[r], c_sub(amc, jbmd));
                    if (k2) xo[2 * sl] = c_mul(w2[r], c_sub(apc, bpd));
                    if (k3) xo[3 * sl] = c_mul(w3[r], c_add(amc, jbmd));
                }
            }
        }
        asm volatile("" ::: "memory");
        twoff += 3 * n1;
    }
}

/* Between the two transforms, one pass over the spectrum: the symmetry pass of the forward real FFT
 * (fft.c:164-183) for the pair (i, N/2 - i), the power spectrum of both bins (lpc.c:357-365), and the
 * symmetry pass of the inverse real FFT on the result -- the same thread owns the same pair in all three,
 * so nothing goes back to LDS in between.  rtw_fwd / rtw_inv [i-1] = (wr, wi) for pair i. */
template <int NTK>
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
        const uint32_t ia = i, ib = m - i;
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

/* The same pass on the sub-region layout (fft_subregions): bin e stands at (e & 3) QM + ((e >> 2) & 3) SQ + (e >> 4).  Pair P = tid + it NTK:
 * class P / (M / 32) = 4 rho + sigma, tt = P % (M / 32) stand for bin i = rho + 4 sigma + 16 (tt + first) (first = 1 for the class of
 * the multiples of 16, which starts at 16 and ends with the self-paired bin M / 2) and its partner M - i in sub-region
 * ((4 - rho) & 3, 3 - sigma) -- (0, (4 - sigma) & 3) for rho = 0 -- at position SQ - 1 - tt - first: both runs contiguous over the
 * lanes.  The stores, 16 bins apart from lane to lane, go to fft_swz16 of their natural index: eight lanes, eight columns. */
template <int NTK, int M>
__device__ __forceinline__ void spectrum_power_pass_subregions(cplx *x, const cplx *__restrict__ rtw_fwd, const cplx *__restrict__ rtw_inv)
{
    constexpr uint32_t QM = (uint32_t)M >> 2, SQ = (uint32_t)M >> 4, PPC = (uint32_t)M >> 5;     /* pairs per class */
    constexpr int ITS = (M / 2) / NTK;
    static_assert(ITS * NTK == M / 2 && PPC % 64 == 0, "whole rounds; a wavefront stays inside one class");
    const uint32_t tid = threadIdx.x;
    cplx za[ITS], zb[ITS], wf[ITS], wi_[ITS];
    cplx z0 = make_double2(0.0, 0.0);
    if (tid == 0) z0 = x[0];
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        const uint32_t P = tid + (uint32_t)it * NTK;
        const uint32_t cls = P / PPC, tt = P % PPC, rho = cls >> 2, sg = cls & 3u;
        const uint32_t first = (cls == 0u) ? 1u : 0u;
        const uint32_t i = rho + 4u * sg + 16u * (tt + first);
        const uint32_t rho2 = (4u - rho) & 3u, sg2 = (rho != 0u) ? 3u - sg : ((4u - sg) & 3u);
        wf[it] = rtw_fwd[i - 1]; wi_[it] = rtw_inv[i - 1];
        za[it] = x[rho * QM + sg * SQ + tt + first];
        zb[it] = x[rho2 * QM + sg2 * SQ + (SQ - 1u - tt)];
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
        const uint32_t cls = P / PPC, tt = P % PPC, rho = cls >> 2, sg = cls & 3u;
        const uint32_t first = (cls == 0u) ? 1u : 0u;
        const uint32_t i = rho + 4u * sg + 16u * (tt + first);
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
            if (!self) x[fft_swz16(i)] = make_double2(y1, y2);
            x[fft_swz16((uint32_t)M - i)] = make_double2(y3, y4);
        }
    }
    __syncthreads();
}

/* circular autocorrelation of the (already windowed, zero padded) signal in buf (lpc.c:330-376): on return
 * complex slot i/2 (fft_regions: position i/8 of region (i/2) & 3) component i&1 holds the unscaled lag i, for i < num_lags */
template <int R, int NTK, int NFFT = 0, bool FIRSTREG = false, bool WP = false /* the region layout with wave-private stages: fft_regions */>
__device__ void autocorr_in_place(cplx *buf, uint32_t nfft, const cplx *__restrict__ twbase, uint32_t num_lags PHASE_PARAM)
{
    const uint32_t m = nfft >> 1;
    const uint32_t ct = complex_table_len(m), quarter = nfft >> 2;
    const cplx *tw_fwd = twbase;
    const cplx *tw_inv = twbase + ct;
    const cplx *rtw_fwd = twbase + 2 * ct;
    const cplx *rtw_inv = rtw_fwd + quarter;
    if constexpr (WP && NTK / 64 > 4) {
        /* eight wavefronts: sixteen sub-regions (fft_subregions).  The first forward stage has been done from registers. */
        fft_second_stage_regions<R, NTK, NFFT / 2, -1, false, false>(buf, tw_fwd, m);
        PHASE(4);                                                  /* forward stage 2 + barrier */
        fft_subregions<R, NTK, NFFT / 2, -1, false, false>(buf, tw_fwd, m);
        __syncthreads();
        spectrum_power_pass_subregions<NTK, NFFT / 2>(buf, rtw_fwd, rtw_inv);
        PHASE(5);                                                  /* forward stages 3.., spectrum pass (three barriers) */
        const uint32_t need = (num_lags + 1) >> 1;
        fft_first_stage_regions<R, NTK, NFFT / 2, 1, 2>(buf, tw_inv);
        fft_second_stage_regions<R, NTK, NFFT / 2, 1, true, true>(buf, tw_inv, need);
        PHASE(6);                                                  /* first two inverse stages + barriers */
        fft_subregions<R, NTK, NFFT / 2, 1, true, true>(buf, tw_inv, need);
        __syncthreads();
    } else if constexpr (WP) {
        /* the first forward stage has been done (from LDS in place, or from registers): the regions are complete behind its barrier */
        fft_regions<R, NTK, NFFT / 2, -1, false, false>(buf, tw_fwd, m);
        __syncthreads();
        PHASE(4);                                                  /* forward stages 2.. + barrier */
        spectrum_power_pass_regions<NTK, NFFT / 2>(buf, rtw_fwd, rtw_inv);
        PHASE(5);                                                  /* spectrum pass (two barriers) */
        fft_first_stage_regions<R, NTK, NFFT / 2, 1, 1>(buf, tw_inv);
        PHASE(6);                                                  /* first inverse stage + barrier */
        fft_regions<R, NTK, NFFT / 2, 1, true, true>(buf, tw_inv, (num_lags + 1) >> 1);
        __syncthreads();
    } else if constexpr (NFFT != 0) {
        /* the transform's length known at compile time (a launch of one FFT-size class outside chain mode) */
        fft_complex_lds_ct<R, NTK, false, NFFT / 2, -1, FIRSTREG>(buf, tw_fwd, m);
        spectrum_power_pass<NTK>(buf, nfft, rtw_fwd, rtw_inv);
        fft_complex_lds_ct<R, NTK, true, NFFT / 2, 1>(buf, tw_inv, (num_lags + 1) >> 1);
    } else {
        fft_complex_lds<R, NTK, false>(buf, m, -1, tw_fwd, m);
        spectrum_power_pass<NTK>(buf, nfft, rtw_fwd, rtw_inv);
        fft_complex_lds<R, NTK, true>(buf, m, 1, tw_inv, (num_lags + 1) >> 1);
    }
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

/* R: chunks of 8 samples per thread (8 R NTK >= nfft): R butterflies per thread and stage */
/* the analysis of one item: the body of srla_autocorr (one FFT-size class per launch) and of srla_autocorr_pair (two classes in
 * one launch); `bid`: the workgroup's index within its class */
template <int R, int NTK, int NFFT = 0 /* every item of the launch has this FFT size (0: they say themselves) */, bool WP = false /* fft_regions */>
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
    PHASE(9);                                                      /* item record + shift fetched */
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
    /* a block that fills its transform (the bulk: every candidate but a stream's tail) with 16-byte aligned planes: every chunk is whole,
     * so the loads need no per-lane bounds (a wave-uniform test instead of exec-mask bookkeeping on the path every wavefront starts with) */
    const bool full = NFFT != 0 && n == (uint32_t)NFFT && aligned && 4u * CH * NTK == (uint32_t)NFFT;
    if (full) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
            load_chunk(in, iv, it.variant, i4, i4 + 4u, true, v[c]);
            const bool need_prev = lane == 0 && i4 != 0, need_next = lane == 63 && first_pass && i4 + 4 < n;
            edge[c] = (need_prev || need_next) ? load_variant(in, iv, it.variant, need_prev ? i4 - 1 : i4 + 4) : 0;
        }
    } else {
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
        load_chunk(in, iv, it.variant, i4, n, aligned, v[c]);
        const bool need_prev = lane == 0 && i4 != 0 && i4 < n, need_next = lane == 63 && first_pass && i4 + 4 < n;
        edge[c] = (need_prev || need_next) ? load_variant(in, iv, it.variant, need_prev ? i4 - 1 : i4 + 4) : 0;
    }
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
        const int32_t from_below = __builtin_amdgcn_update_dpp(edge[c], v[c][3], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        pv[c] = (i4 == 0 || i4 >= n) ? v[c][0] : from_below;
        nxv[c] = __builtin_amdgcn_update_dpp(edge[c], v[c][0], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    }

    asm volatile("" :: "v"(v[0][0]), "v"(v[CH - 1][3]), "v"(pv[CH - 1]), "v"(nxv[CH - 1]));
    PHASE(0);                                                      /* item record fetched, sample loads landed */
    /* The Welch window's weights of a block that fills its transform come from a table (round 6; SrlaJobParams::welch_tab: the
     * products the window loop below would form, made once on the host in the same order -- the weight does not depend on the
     * data): five of a sample's six fp64 operations gone.  Chunk rounds c < CH / 2 lie in the window's first half (entries
     * i4 .. i4 + 3), the others in the second, which mirrors the first (entries n - 1 - i4 - i: an aligned group read backwards).
     * Requested here, behind the samples, so that the round trip to L2 runs beside the integer correlations and their barrier. */
    const bool tabled = full && jp.welch_tab != nullptr && NFFT >= 1024;
    double wtab[CH][4];
    if (tabled) {
        const double *wt = jp.welch_tab + (NFFT / 2 - 512);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
            const uint32_t e0 = (c < CH / 2) ? i4 : (uint32_t)NFFT - 4u - i4;
            const double2 lo = *reinterpret_cast<const double2 *>(wt + e0), hi = *reinterpret_cast<const double2 *>(wt + e0 + 2);
            if (c < CH / 2) { wtab[c][0] = lo.x; wtab[c][1] = lo.y; wtab[c][2] = hi.x; wtab[c][3] = hi.y; }
            else { wtab[c][0] = hi.y; wtab[c][1] = hi.x; wtab[c][2] = lo.y; wtab[c][3] = lo.x; }
        }
    }
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
    /* Chain mode (the odd-length tail window of a stream, host_encoder.cpp): the launch reproduces one call of
     * the reference on its persistent FFT buffer (lpc.c:58,211).  chain_src - 1 is where the buffer's middle word
     * stands in chain_pool (the Welch window leaves it untouched for odd n, lpc.c:260-264); at chain_dump - 1 the
     * call leaves the complete buffer (all nfft words of the inverse transform) for the calls after it. */
    const bool chain = chain_pool != nullptr;
    if (pass == 0 && jp.max_order == 0 && !chain) return;   /* preset 0: fixed order 0, no LPC analysis needed */
    PHASE(1);                                                      /* tap sums, reduction, pre-emphasis tap */

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
            /* A chunk's sources are the six consecutive words from i4 - back (back = period + half_order): fetched as the two (three
             * where they straddle: rr = 3 with three taps) ALIGNED groups of four words that hold them -- lanes 16 bytes apart:
             * conflict free -- and picked out of the registers by rr = (-back) & 3, which the whole item shares (round 6; a load
             * per word met a four-way bank conflict every time, lanes being four words apart).  A group in front of the block is
             * only ever looked at by samples below back + 1, which are not filtered. */
            const uint32_t back = period + half_order;
            const uint32_t rr = (uint32_t)__builtin_amdgcn_readfirstlane((int)((0u - back) & 3u));
            const bool third = rr == 3u && taps == 3u;
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
                if (i4 < n && i4 + 3u >= back + 1u) {
                    const int32_t g0 = ((int32_t)i4 - (int32_t)back) >> 2;
                    int32_t W[12];
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        int4 w = make_int4(0, 0, 0, 0);
                        if (j < 2 || third) w = *reinterpret_cast<const int4 *>(ylds + 4 * ((g0 + j > 0) ? g0 + j : 0));
                        W[4 * j] = w.x; W[4 * j + 1] = w.y; W[4 * j + 2] = w.z; W[4 * j + 3] = w.w;
                    }
                    auto filter = [&](auto rc) {
                        constexpr int RR = decltype(rc)::value;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const uint32_t s = i4 + i;
                            if (s < n && s >= back + 1u) {
                                uint32_t acc = 16u + (uint32_t)c0 * (uint32_t)W[RR + i];
                                if (taps == 3) acc += (uint32_t)c1 * (uint32_t)W[RR + i + 1] + (uint32_t)c2 * (uint32_t)W[RR + i + 2];
                                v[c][i] = (int32_t)((uint32_t)v[c][i] - (uint32_t)((int32_t)acc >> 5));
                            }
                        }
                    };
                    switch (rr) {
                    case 0: filter(std::integral_constant<int, 0>()); break;
                    case 1: filter(std::integral_constant<int, 1>()); break;
                    case 2: filter(std::integral_constant<int, 2>()); break;
                    default: filter(std::integral_constant<int, 3>()); break;
                    }
                }
            }
            __syncthreads();
        }
    }

    /* Welch window (lpc.c:256-266) on the [-1,1) normalised signal, zero padded to nfft */
    constexpr bool FIRSTREG = NFFT != 0 && R == 2 && 8 * R * NTK == NFFT;   /* fft_first_stage_regs: the windowed chunks stay in registers */
    double wreg[FIRSTREG ? 4 : 1][4];
    {
        const double norm_bps = __builtin_ldexp(1.0, -(int)(bps - 1));
        const uint32_t half = n >> 1;
        /* weight(e) = (divisor * smpl) * (n - 1 - smpl), smpl = e in the first half and n - 1 - e in the second:
         * both factors are small integers, so they are formed as doubles by exact additions from one conversion per
         * thread instead of two int -> double conversions per sample (quarter-rate instructions) */
        const double d_tid4 = (double)(4u * tid), d_nm1 = (double)(n - 1u);
        const double div_scaled = g.welch_divisor * norm_bps;      /* exact: norm_bps is a power of two */
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i4 = 4u * (tid + (uint32_t)c * NTK);
            if (i4 < nfft) {
                double w[4];
                const double de0 = d_tid4 + (double)(4 * c * NTK);          /* (double)i4, exact */
                const bool first = i4 + 4u <= half, second = i4 >= n - half && i4 + 4u <= n;
                if (tabled) {
#pragma unroll
                    for (int i = 0; i < 4; i++) w[i] = (double)v[c][i] * wtab[c][i];
                } else if (full) {
                    /* chunk rounds c < CH / 2 lie in the first half, the others in the second: no selects.  The sample's scaling is
                     * folded into the divisor (a power of two: every product keeps its rounding) */
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const double de = de0 + (double)i, dr = d_nm1 - de;
                        const double a = (c < CH / 2) ? de : dr, b = (c < CH / 2) ? dr : de;
                        w[i] = (double)v[c][i] * (div_scaled * a * b);
                    }
                } else if (first || second) {
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
                    buf[i4 >> 1] = make_double2(w[0], w[1]);
                    buf[(i4 >> 1) + 1u] = make_double2(w[2], w[3]);
                }
            }
        }
        if constexpr (!FIRSTREG) __syncthreads();
    }

    PHASE(2);                                                      /* pre-emphasis, (LTP), window */
    const uint32_t num_lags = (pass == 1) ? SRLA_LTP_LAGS : (jp.max_order + 1);
    const bool dump = chain && it.chain_dump;
    static_assert(!WP || NFFT != 0, "the region layout needs the transform's length at compile time");
    if constexpr (WP) {
        if constexpr (FIRSTREG) fft_first_stage_regs_regions<NFFT / 2>(buf, wreg, twiddles + g.tw_off);
        else fft_first_stage_regions<R, NTK, NFFT / 2, -1, 0>(buf, twiddles + g.tw_off);
    } else {
        if constexpr (FIRSTREG) fft_first_stage_regs<NFFT / 2>(buf, wreg, twiddles + g.tw_off);
    }
    PHASE(3);                                                      /* first forward stage + its barrier */
    autocorr_in_place<R, NTK, NFFT, FIRSTREG, WP>(buf, nfft, twiddles + g.tw_off, (num_lags < nfft && !dump) ? num_lags : nfft PHASE_ARG);
    PHASE(7);                                                      /* inverse stages 2.. + barrier (4-6: inside autocorr_in_place) */
    /* where complex element e of the result stands */
    auto slot = [&](uint32_t e) -> uint32_t {
        if (WP && NTK / 64 > 4) return (e & 3u) * (uint32_t)(NFFT / 8) + ((e >> 2) & 3u) * (uint32_t)(NFFT / 32) + (e >> 4);     /* fft_subregions */
        return WP ? ((e & 3u) * (uint32_t)(NFFT / 8) + (e >> 2)) : e;
    };
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
    PHASE(8);                                                      /* lag stores */
}

template <int R, int NTK, int NFFT = 0, bool WP = false>
__global__ __launch_bounds__(NTK) __attribute__((amdgpu_waves_per_eu(4, 8))) void srla_autocorr(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    const SrlaGeom *__restrict__ geoms, const cplx *__restrict__ twiddles, uint32_t fft_bytes, uint32_t pass,
    SrlaItemResult *__restrict__ results, double *__restrict__ lags_ws, double *__restrict__ dbg,
    const SrlaAutocorrItem *__restrict__ class_items, uint32_t count, double *__restrict__ chain_pool,
    const uint32_t *__restrict__ chain_tab)
{
    autocorr_item<R, NTK, NFFT, WP>(jp, input, items, geoms, twiddles, fft_bytes, pass, results, lags_ws, dbg, class_items, count, chain_pool, chain_tab, blockIdx.x);
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
        autocorr_item<2, 256, 4096, WP>(jp, input, items, geoms, twiddles, fft_bytes, pass, results, lags_ws, dbg, items_4096, count_4096, nullptr, nullptr, blockIdx.x);
    else
        autocorr_item<1, 256, 2048, WP>(jp, input, items, geoms, twiddles, fft_bytes, pass, results, lags_ws, dbg, items_2048, count_2048, nullptr, nullptr, blockIdx.x - g4);
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
        spectrum_power_pass<NTB>(res, nfft, twbase + 2 * ct, twbase + 2 * ct + quarter);
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
/* --------------------------------------------------------------------------- launchers ---- */
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
#define LAUNCH(RR, TT, NF, WPV)                                                                              \
    do {                                                                                                     \
        SET_LDS_ATTR((srla_autocorr<RR, TT, NF, WPV>));                                                      \
        hipExtLaunchKernelGGL((srla_autocorr<RR, TT, NF, WPV>), grid, dim3(TT), lds, stream, ev_start, ev_stop, 0, *jp, input, items, geoms, \
                           (const cplx *)twiddles, fft_bytes, pass, results, lags_ws, dbg, class_items, count, chain_pool, chain_tab); \
    } while (0)
    /* outside chain mode the items of a launch (classes of more than 1024 points; `exact_nfft`: also the 1024-point class) all
     * have the class's FFT size: the kernel with the transform's length compiled in */
    if (chain_pool == nullptr && (rclass != 0 || exact_nfft)) {
        /* the region layout with wave-private stages (fft_regions; the 8192-point class on eight wavefronts: fft_subregions); SRLA_MI355X_FFT_WP=0: round 4's form */
        const bool wp = g_srla_tune.fft_wp != 0u;
        switch (rclass) {
        case 0: if (wp) LAUNCH(1, 128, 1024, true); else LAUNCH(1, 128, 1024, false); break;
        case 1: if (wp) LAUNCH(1, 256, 2048, true); else LAUNCH(1, 256, 2048, false); break;
        case 2: if (wp) LAUNCH(2, 256, 4096, true); else LAUNCH(2, 256, 4096, false); break;
        /* the 8192-point class on sixteen sub-regions (fft_subregions, round 6): 9 barriers per item instead of 26 and the same bits, but
         * measured 5 % SLOWER stand-alone (164 -> 172 us per job of -B 8192 -V 2 -P 3, profiles/r06/ab_kernels.txt) -- like the
         * 4096-point class's barriers in round 5, this class's were not what its wavefronts wait for.  SRLA_MI355X_FFT_WP=2 selects it. */
        case 4: if (g_srla_tune.fft_wp >= 2u) LAUNCH(2, 512, 8192, true); else LAUNCH(2, 512, 8192, false); break;
        default: return -1;
        }
        return (hipGetLastError() == hipSuccess) ? 0 : -2;
    }
    switch (rclass) {
    case 0: LAUNCH(1, 128, 0, false); break;     /* <= 1024 points: 128 threads (one butterfly each and stage), 8 KB of LDS */
    case 1: LAUNCH(1, 256, 0, false); break;
    case 2: LAUNCH(2, 256, 0, false); break;
    /* 8192-point items run on 512 threads (two butterflies per thread and stage): their 70 KB of LDS allow two
     * workgroups per CU, which with 256 threads would be two wavefronts per SIMD */
    case 4: LAUNCH(2, 512, 0, false); break;
    default: return -1;
    }
#undef LAUNCH
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

extern "C" int srla_launch_autocorr_pair(hipStream_t stream, const SrlaJobParams *jp, const int32_t *input, const SrlaItemDesc *items,
                                         const SrlaGeom *geoms, const void *twiddles, uint32_t pass, SrlaItemResult *results, double *lags_ws,
                                         double *dbg, const SrlaAutocorrItem *items_4096, uint32_t count_4096,
                                         const SrlaAutocorrItem *items_2048, uint32_t count_2048, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (count_4096 == 0 || count_2048 == 0) return -1;
    const uint32_t fft_bytes = 2048u * 16u, lds = fft_bytes + srla_kernel_small_a_bytes();
    dim3 grid(8u * ((count_4096 + 7u) >> 3) + 8u * ((count_2048 + 7u) >> 3));
    if (g_srla_tune.fft_wp) {
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

