/*
 * residual_cost.hip -- srla_residual_cost / srla_residual_cost_big: one workgroup per item: pre-emphasis (+ long-term predictor), the
 * wrap-around int32 FIR (on the matrix pipe for blocks of at most 4096 samples: FIR_MFMA; v_dot2 / v_dot4 on int16 / int8 planes:
 * FIR_DOT; two 24-bit multiplies per tap for 24-bit input: FIR_WIDE), the residual to HBM (uint16 where it fits), the partitioned
 * (recursive) Rice code-length search and the channel's code length.  DESIGN.md 3.3.
 */
#include "kernels_common.h"
SRLA_DIAG_PHASE_READER(residual_cost)

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
#define MF_OFFZ 160                 /* ... index of tap 0 in the zero-padded tap string: the lowest byte a lane reads is MF_OFFZ - 16 FL - 14 (row 15 of the
                                     * last tile, the order rounded up by 15), >= 0 for FL <= 8 (144 until round 6, when the form stopped at FL = 4) */
#define MF_TZB  576                 /* ... bytes of one copy of it: MF_OFFZ + 64 k-blocks' worth for order 255 (6 at FL = 8) + a lane's 16 bytes, rounded up */
struct SmallF {
    union {
        int32_t  coefq[FIR_PAD + 8];          /* FIR_WIDE: taps, front padded with zeros to a multiple of four */
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
 * FIR_WIDE  samples beyond 24 bits (24-bit input: M/S, pre-emphasis and the LTP widen it to 28): every tap multiplies
 *           the two 16-bit halves of the sample separately on the 24-bit multiplier (x c = (x >> 16) c 2^16 + (x & 0xffff) c
 *           modulo 2^32), which is still twice as fast as 32-bit multiplies */
#define FIR_WIDE  1
#define FIR_DOT   2
#define FIR_MFMA  3
#define SRLA_FIR_NARROW FIR_DOT
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
/* The level sums of the Rice search leave the registers through LDS (round 6): every wavefront lays its lanes' eleven sums out as
 * eleven rows of 64 words (68 with the pad that keeps the column reads conflict free) over the signal's planes -- which nobody
 * reads any more behind the barrier in front of the search -- and four lanes per level add a row up.  The region in front of the
 * small structure is therefore at least four wavefronts' worth of rows, whatever the block length. */
#define RC_RED_ROW 68u
#define RC_RED_WAVE_WORDS (11u * RC_RED_ROW)
__host__ __device__ constexpr uint32_t fast_region_bytes(int fl, int lg, bool planes_only)
{
    const uint32_t sig = fast_sig_bytes(fl, lg, planes_only), red = (uint32_t)NWAVES * RC_RED_WAVE_WORDS * 4u;
    return (lg == 0 && red > sig) ? red : sig;
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
/* what the analysis kernels before this one left in the item record and every thread needs at once: fetched by the kernel FIRST,
 * beside the item descriptor (the record's address needs only the workgroup's index), not behind it */
struct ItemHead { int32_t preemph_coef; uint32_t order, rshift, period; int32_t ltp_coef[3]; };

template <int FL, int MODE, int LG = 0>
__device__ __forceinline__ void residual_cost_fast(const SrlaJobParams &jp, const InputView &iv, const int32_t *__restrict__ in, const SrlaItemDesc &it,
                                   unsigned char *lds, const double *__restrict__ rice_thresholds,
                                   int32_t *__restrict__ res_ws, SrlaItemResult *__restrict__ out, const ItemHead &head)
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
    static_assert(!MF || (LG == 0 && FL <= 8), "FIR_MFMA: one item per workgroup, blocks of at most 8192 samples");
    static_assert(!MF || (MF_OFFZ >= 16 * FL + 14 && MF_OFFZ % 16 == 0 && MF_OFFZ + 64 * 6 + 16 <= MF_TZB && MF_TZB % 4 == 0), "the tap string covers every lane's reach");
    static_assert(!MF || 3u * MF_PLS <= fast_sig_bytes(FL, LG, true), "the byte planes fit where the int16 / int8 planes would lie");
    /* FIR_DOT: the two planes lie over the int32 signal (which then only the LTP uses, before them): PADF zeros + the block, in
     * the same padded element order as the int32 layout */
    constexpr int PADF = ((FIR_PAD + S - 1) / S) * S;
    constexpr uint32_t PLANE_ELEMS = (uint32_t)((PADF + 1024 * FL) / S) * (S + PADW);
    constexpr uint32_t HIGH_OFF = (PLANE_ELEMS * 2 + 15) & ~15u;
    static_assert(HIGH_OFF + PLANE_ELEMS <= SIG_WORDS * 4, "the planes fit the int32 signal's LDS");
    int32_t *sig = (int32_t *)lds;
    static_assert(fast_sig_bytes(FL, LG, false) == ((SIG_WORDS * 4 + 15) & ~15u) && fast_sig_bytes(FL, LG, true) == ((HIGH_OFF + PLANE_ELEMS + 15) & ~15u), "one layout");
    SmallF *sm = (SmallF *)(lds + fast_region_bytes(FL, LG, DOT && jp.ltp_order == 0));
    const uint32_t tid = threadIdx.x & (uint32_t)(T - 1), lane = tid & 63, wave = tid >> 6;   /* thread, wavefront within the item */
    const uint32_t n = 1024u * FL, bps = jp.bits_per_sample;
    const bool aligned = input_aligned(in, iv);
    const int32_t coef = head.preemph_coef;
    const uint32_t order = head.order, rshift = head.rshift, period = head.period;
    const uint32_t o4 = (order + 3u) & ~3u;
    const uint32_t s_base = (uint32_t)S * tid;
    PHASE_INIT();
    /* FIR_MFMA: the words of the item record's tap array a thread's words of the shifted tap strings are made of (wavefront s builds
     * copy s; words lane and lane + 64 of a copy can hold taps, the array's 64 words are always there to be read: no need to know the
     * order yet, so these go out with the very first requests) */
    uint32_t tap_lo[2] = { 0u, 0u }, tap_hi[2] = { 0u, 0u };
    if constexpr (MODE == FIR_MFMA) {
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(out->lpc_coef);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int q0 = ((int)(4u * (lane + 64u * (uint32_t)r)) - (int)wave - MF_OFFZ) >> 2;
            if (q0 >= 0 && q0 < 64) tap_lo[r] = cw[q0];
            if (q0 + 1 >= 0 && q0 + 1 < 64) tap_hi[r] = cw[q0 + 1];
        }
    }
    /* (two more values that would otherwise be fetched late, in front of a barrier / of the record's last store: requested here,
     * where their round trips run beside the sample loads) */
    const double thr_mine = (tid >= 32 && tid < 64) ? rice_thresholds[tid - 32] : 0.0;
    const uint32_t tap_bits = (tid == 0) ? out->pad[0] : 0u;       /* tap codes (srla_lpc_taps) */

    /* load + pre-emphasis (srla_utility.c:342) */
    int32_t y[S];
    /* (every chunk of these blocks is whole: with aligned planes -- a wave-uniform test -- the loads carry no per-lane bounds) */
    if (aligned) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            int32_t t4[4];
            load_chunk(in, iv, it.variant, s_base + 4 * c, s_base + 4 * c + 4, true, t4);
            y[4 * c] = t4[0]; y[4 * c + 1] = t4[1]; y[4 * c + 2] = t4[2]; y[4 * c + 3] = t4[3];
        }
    } else {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            int32_t t4[4];
            load_chunk(in, iv, it.variant, s_base + 4 * c, n, aligned, t4);
            y[4 * c] = t4[0]; y[4 * c + 1] = t4[1]; y[4 * c + 2] = t4[2]; y[4 * c + 3] = t4[3];
        }
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
                /* bytes k0 .. k0 + 3 of the tap array (zero outside [0, order)): the two aligned words of it fetched at the top (tap_lo /
                 * tap_hi), funnel-shifted by the copy's shift, then masked */
                const int k0 = (int)(4u * dw) - (int)sft - MF_OFFZ;
                const uint32_t w = (r < 2) ? __builtin_amdgcn_alignbyte(tap_hi[r < 2 ? r : 0], tap_lo[r < 2 ? r : 0], (uint32_t)(k0 & 3)) : 0u;
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
    PHASE(0);                                                      /* sample loads landed */
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
            if constexpr (CH % 4 == 0) {
#pragma unroll
                for (int c = 0; c < CH; c += 4) {
                    *reinterpret_cast<uint4 *>(p0 + 4 * c) = make_uint4(w0[c], w0[c + 1], w0[c + 2], w0[c + 3]);
                    *reinterpret_cast<uint4 *>(p0 + MF_PLS + 4 * c) = make_uint4(w1[c], w1[c + 1], w1[c + 2], w1[c + 3]);
                    *reinterpret_cast<uint4 *>(p0 + 2 * MF_PLS + 4 * c) = make_uint4(w2[c], w2[c + 1], w2[c + 2], w2[c + 3]);
                }
            } else if constexpr (CH % 2 == 0) {
#pragma unroll
                for (int c = 0; c < CH; c += 2) {
                    *reinterpret_cast<uint2 *>(p0 + 4 * c) = make_uint2(w0[c], w0[c + 1]);
                    *reinterpret_cast<uint2 *>(p0 + MF_PLS + 4 * c) = make_uint2(w1[c], w1[c + 1]);
                    *reinterpret_cast<uint2 *>(p0 + 2 * MF_PLS + 4 * c) = make_uint2(w2[c], w2[c + 1]);
                }
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
    PHASE(1);                                                      /* pre-emphasis, planes + taps published, barrier */

    if (ltp_block) {
        if (period > 0) {
        /* long-term predictor, srla_lpc_predict.c:267-294 (in place: read everything, barrier, rewrite) */
        const uint32_t taps = jp.ltp_order, half_order = taps >> 1;
        const int32_t c0 = head.ltp_coef[0], c1 = head.ltp_coef[1], c2 = head.ltp_coef[2];
        /* The thread's S + 2 source samples are consecutive words of the padded layout (a pad of four words behind every S samples,
         * sig_index) from an index that is a multiple of four plus a remainder `rr` the whole item shares (s_base and PADS are
         * multiples of four).  They are fetched as the CH + 1 (rr = 3 with three taps: CH + 2) ALIGNED groups of four words that
         * hold them -- 16-byte loads, a lane stride of S (+ 4) words: conflict free -- and picked out of the registers by rr, one
         * of four copies of the loop chosen by a scalar branch.  (Until round 6 every word was a load of its own: with lanes 16 or
         * 20 words apart each of them met a four-way bank conflict, S + 2 times per thread -- what doubled this kernel's conflict
         * share under -P 3: profiles/r06/README.md.)  PADS >= the largest period + 2, so the first group is never negative, and
         * the period is at least 8, so the last one lies inside the thread's own run. */
        const uint32_t base0 = (uint32_t)PADS + s_base - period - half_order;
        const uint32_t rr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(base0 & 3u));
        const uint32_t g0 = base0 >> 2, gq = g0 / (uint32_t)CH, gr = g0 - gq * (uint32_t)CH;   /* group of four words; the run of CH groups it lies in */
        const bool last_group = rr == 3u && taps == 3u;
        int32_t W[4 * (CH + 2)];
#pragma unroll
        for (int j = 0; j < CH + 2; j++) {
            const uint32_t pads = (PADW != 0) ? gq + ((gr + (uint32_t)j >= 2u * CH) ? 2u : ((gr + (uint32_t)j >= (uint32_t)CH) ? 1u : 0u)) : 0u;
            int4 w = make_int4(0, 0, 0, 0);
            if (j <= CH || last_group) w = *reinterpret_cast<const int4 *>(sig + 4u * (g0 + (uint32_t)j + pads));
            W[4 * j] = w.x; W[4 * j + 1] = w.y; W[4 * j + 2] = w.z; W[4 * j + 3] = w.w;
        }
        auto filter = [&](auto rc) {
            constexpr int RR = decltype(rc)::value;
#pragma unroll
            for (int i = 0; i < S; i++) {
                const uint32_t s = s_base + i;
                if (s >= period + half_order + 1) {
                    uint32_t acc;
                    if constexpr (!WIDE) {
                        /* 6-bit taps, samples within 24 bits (see the FIR below): the full-rate 24-bit multiplier */
                        acc = mad24(c0, W[RR + i], 16u);
                        if (taps == 3) acc = mad24(c2, W[RR + i + 2], mad24(c1, W[RR + i + 1], acc));
                    } else {
                        acc = 16u + (uint32_t)c0 * (uint32_t)W[RR + i];
                        if (taps == 3) acc += (uint32_t)c1 * (uint32_t)W[RR + i + 1] + (uint32_t)c2 * (uint32_t)W[RR + i + 2];
                    }
                    y[i] = (int32_t)((uint32_t)y[i] - (uint32_t)((int32_t)acc >> 5));
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
        __syncthreads();
        if constexpr (DOT) publish_planes(); else PUBLISH_Y();
        __syncthreads();
    }
#undef PUBLISH_Y

    PHASE(2);                                                      /* LTP */
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
            /* lane 16 g + cc holds chunk t of thread 4 cc + g: every thread fetches its own from lane 16 (lane & 3) + (lane >> 2) */
            const int from = (int)(4u * (16u * (lane & 3u) + (lane >> 2)));
            /* The tiles in groups of at most four (round 6: blocks above 4096 samples, FL = 5 .. 8, take this form too): two accumulator
             * sets of a group are 8 TG registers, and a group's outputs go back to their owners before the next group starts -- all eight
             * tiles of an 8192-sample block at once would be 64 accumulator registers beside the 32 samples and 32 outputs a thread
             * holds.  The B operands (one aligned 16-byte load per plane and k-block) are fetched once per group. */
            constexpr int TG = (FL > 4) ? (FL + 1) / 2 : FL, NTG = (FL + TG - 1) / TG;
#pragma unroll
            for (int tg = 0; tg < NTG; tg++) {
                const int t0 = tg * TG;
                mf_v4i a0[TG], a1[TG];
#pragma unroll
                for (int t = 0; t < TG; t++) { a0[t] = (mf_v4i){ half, half, half, half }; a1[t] = (mf_v4i){ 0, 0, 0, 0 }; }
                for (uint32_t kb = 0; kb < mf_nkb; kb++) {
                    const mf_v4i b0 = *reinterpret_cast<const mf_v4i *>(bp + 64u * kb);
                    const mf_v4i b1 = *reinterpret_cast<const mf_v4i *>(bp + MF_PLS + 64u * kb);
#pragma unroll
                    for (int t = 0; t < TG; t++) {
                        if (t0 + t < FL) {
                            const uint32_t *ap = az + 16u * kb - (uint32_t)(t0 + t);
                            const mf_v4i a = (mf_v4i){ (int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3] };
                            a0[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b0, a0[t], 0, 0, 0);
                            a1[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b1, a1[t], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < TG; t++)
#pragma unroll
                    for (int i = 0; i < 4; i++) a0[t][i] = (int)((uint32_t)a0[t][i] + ((uint32_t)a1[t][i] << 8));
                if (high) {
                    /* the third digit, where a sample left 16 bits (full-scale input): one more pass */
#pragma unroll
                    for (int t = 0; t < TG; t++) a1[t] = (mf_v4i){ 0, 0, 0, 0 };
                    for (uint32_t kb = 0; kb < mf_nkb; kb++) {
                        const mf_v4i b2 = *reinterpret_cast<const mf_v4i *>(bp + 2 * MF_PLS + 64u * kb);
#pragma unroll
                        for (int t = 0; t < TG; t++) {
                            if (t0 + t < FL) {
                                const uint32_t *ap = az + 16u * kb - (uint32_t)(t0 + t);
                                const mf_v4i a = (mf_v4i){ (int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3] };
                                a1[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b2, a1[t], 0, 0, 0);
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < TG; t++)
#pragma unroll
                        for (int i = 0; i < 4; i++) a0[t][i] = (int)((uint32_t)a0[t][i] + ((uint32_t)a1[t][i] << 16));
                }
#pragma unroll
                for (int t = 0; t < TG; t++)
                    if (t0 + t < FL) {
#pragma unroll
                        for (int i = 0; i < 4; i++) acc[4 * (t0 + t) + i] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, a0[t][i]);
                    }
            }
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
        {
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

    asm volatile("" :: "v"(u[0]), "v"(u[S - 1]), "v"(max_u));
    PHASE(3);                                                      /* FIR, residual, zig-zag */
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
    PHASE(4);                                                      /* partition means, barrier */
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
    PHASE(5);                                                      /* residual store */

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
    PHASE(6);                                                      /* Rice parameters, table, barrier */
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
        if constexpr (LG == 0) {
            /* Eleven sums over the wavefront's 64 lanes: through LDS (see RC_RED_ROW) instead of eleven DPP reductions in the
             * VALU, which this kernel keeps busy four cycles out of five while LDS idles three out of four: 11 word stores a lane,
             * then lane 4 l + g adds up quarter g of row l -- four 16-byte loads whose columns (l + 4 g + i) mod 16 differ in every
             * lane group a load is served in (tests/test_kernel_models.py) -- and ONE atomic adds the 44 partial sums to the item's
             * eleven totals.  A wavefront's LDS operations execute in order: no barrier between its stores and its loads. */
            uint32_t *red = reinterpret_cast<uint32_t *>(lds) + wave * RC_RED_WAVE_WORDS;
#pragma unroll
            for (int l = 0; l <= 10; l++) red[(uint32_t)l * RC_RED_ROW + lane] = acc[l];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const uint32_t lv = lane >> 2, seg = lane & 3u;
            if (lv <= 10u) {
                const uint4 *row = reinterpret_cast<const uint4 *>(red + lv * RC_RED_ROW + 16u * seg);
                const uint4 q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
                const uint32_t sum = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w)) + ((q2.x + q2.y) + (q2.z + q2.w)) + ((q3.x + q3.y) + (q3.z + q3.w));
                atomicAdd(&sm->level_bits[lv], sum);
            }
        } else {
#pragma unroll
        for (int l = 0; l <= 10; l++) {
            const uint32_t sum = wave_sum_u32(acc[l]);
            if (lane == 0) atomicAdd(&sm->level_bits[l], sum);
        }
        }
    }
    __syncthreads();
    PHASE(7);                                                      /* side information, code bits of 11 levels, reductions, barrier */
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
    PHASE(8);                                                      /* arg-min, record */
}

extern "C" uint32_t srla_kernel_fast_lds_bytes(uint32_t fl, uint32_t ltp_order, uint32_t bits_per_sample)
{
    const bool dot = bits_per_sample <= 18;
    const int lg = 0;
    const uint32_t item = fast_region_bytes((int)fl, lg, dot && ltp_order == 0) + (uint32_t)((sizeof(SmallF) + 15) & ~15u);
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
#define SRLA_RC_WAVES 6       /* wavefronts per SIMD the forms for blocks of at most 4096 samples are compiled for: 80 registers and 10 spilled dwords per lane
                               * (round 6; 5 = 92 registers, no spills, until then: the stage 5 % slower at -V 2, 1 % at -V 1, profiles/r06/ab_launch_shapes.txt) */
#endif
#ifndef SRLA_RC4_WAVES
#define SRLA_RC4_WAVES 3      /* wavefronts per SIMD the 8192-sample form is compiled for: 168 registers and 17 spilled dwords per lane; 2 (228 registers, no spills) was 9 % slower at -B 8192 -V 2 -P 3, profiles/r04/ab_residual_cost_split.txt */
#endif
template <int R, bool MFMA = false /* narrow input: the FIR on the matrix pipe (FIR_MFMA) */>
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
    /* the item record's head and the item descriptor: two independent fetches, issued together */
    ItemHead head;
    {
        const SrlaItemResult *r = &results[block];
        head.preemph_coef = r->preemph_coef; head.order = r->lpc_order; head.rshift = r->lpc_rshift; head.period = r->ltp_period;
        head.ltp_coef[0] = r->ltp_coef[0]; head.ltp_coef[1] = r->ltp_coef[1]; head.ltp_coef[2] = r->ltp_coef[2];
    }
    const SrlaItemDesc itf = items[block];
    if (itf.n > 8192u) return;                       /* srla_residual_cost_big takes these */
    if (jp.rc_hi != 0u && (itf.n <= jp.rc_lo || itf.n > jp.rc_hi)) return;   /* the other launch of the job takes these */
    const InputView iv = input_view(jp, itf.lshift, input);
    {
        /* blocks of 1024 * FL samples take the register / shuffle fast path */
        const uint32_t fl = itf.n >> 10;
        if ((itf.n & 1023u) == 0 && fl >= 1 && fl <= 8 && fl <= (uint32_t)(2 * R)) {
            const int32_t *inf = input + itf.sample_off;
            SrlaItemResult *outf = &results[block];
#define FAST(FLV)                                                                                                   \
            do {                                                                                                    \
                if (jp.bits_per_sample <= 18) residual_cost_fast<FLV, MFMA ? FIR_MFMA : SRLA_FIR_NARROW>(jp, iv, inf, itf, lds, rice_thresholds, res_ws, outf, head); \
                else residual_cost_fast<FLV, FIR_WIDE>(jp, iv, inf, itf, lds, rice_thresholds, res_ws, outf, head); \
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

/* --------------------------------------------------------------------------- launchers ---- */
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
    const bool mf = g_srla_tune.fir_mfma != 0u;
    switch (rclass) {
    case 1: if (mf) LAUNCH(1, true); else LAUNCH(1, false); break;
    case 2: if (mf) LAUNCH(2, true); else LAUNCH(2, false); break;
    case 4: if (mf) LAUNCH(4, true); else LAUNCH(4, false); break;
    default: return -1;
    }
#undef LAUNCH
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

