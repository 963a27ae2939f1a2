/*
 * device_layout.h -- records shared by the host runtime and the HIP kernels.
 *
 * Vocabulary (follows the reference's domain):
 *   window     the look-ahead span the block-division search works on
 *              (SRLAEncoder_EncodeWhole's loop, srla_encoder.c:1756-1783); one block when the
 *              search is off (min == max block size)
 *   candidate  one block [i*min, j*min) of a window that the search prices
 *              (SearchOptimalBlockPartitions, srla_encoder.c:353-389)
 *   item       one analysed channel variant of one candidate: a plain channel, or the M / S
 *              signal of the first two channels (ComputeCoefficientsPerChannel, :966)
 */
#ifndef SRLA_DEVICE_LAYOUT_H
#define SRLA_DEVICE_LAYOUT_H

#include <stdint.h>

#define SRLA_MAX_CH          8
#define SRLA_MAX_ORDER       255
#define SRLA_MAX_NODES       1025    /* lookahead / min block + 1 (`srla -e -V 8` at the default look-ahead factor 4; 129 until round 4) */
#define SRLA_MAX_PORDER      10      /* srla_coder.c:18 */
#define SRLA_LTP_MIN_PERIOD  8
#define SRLA_WELCH_TAB_WORDS 7680   /* 512 + 1024 + 2048 + 4096 doubles: SrlaJobParams::welch_tab */
#define SRLA_LTP_MAX_PERIOD  262
#define SRLA_LTP_LAGS        (SRLA_LTP_MAX_PERIOD + 1)
#define SRLA_BIG_WEIGHT      (1u << 24)

enum { SRLA_BLOCK_COMPRESS = 0, SRLA_BLOCK_SILENT = 1, SRLA_BLOCK_RAW = 2 };
enum { SRLA_CODE_RICE = 0, SRLA_CODE_RECURSIVE_RICE = 1, SRLA_CODE_ALLZERO = 2 };

/* item flags */
#define SRLA_ITEM_INPUT_ZERO   1u   /* every sample of the variant's input is zero            */
#define SRLA_ITEM_ORDER_TIE    2u   /* order selection within the libm tolerance: host decides */
#define SRLA_ITEM_LTP_TIE      4u   /* LTP tap within tolerance of a rounding boundary         */
#define SRLA_ITEM_LTP_FAIL     8u   /* 3x3 Cholesky failed: the reference returns NG           */
#define SRLA_ITEM_ODD_LENGTH   16u  /* odd block length: reference result is history dependent */
#define SRLA_ITEM_SVR_TIE      32u  /* SVR refinement: a comparison of objective values within the libm tolerance */
#define SRLA_ITEM_RES_U16      64u  /* the item's residual stands in the scratch zig-zag mapped, as uint16 (srla_residual_cost) */

/* Per block-length constants the host prepares with the host libm. */
typedef struct SrlaGeom {
    uint32_t n;            /* block length                                        */
    uint32_t nfft;         /* next power of two >= n (lpc.c:346)                  */
    uint32_t log2_nfft;
    uint32_t max_porder;   /* min(trailing zeros of n, 10) (srla_coder.c:358-364) */
    uint32_t fine_len;     /* n >> max_porder                                     */
    uint32_t tw_off;       /* offset (in double2) of this nfft's twiddle tables   */
    uint32_t pad0, pad1;
    double welch_divisor;  /* 4 * pow(n - 1, -2)            (lpc.c:259)           */
    double welch_comp;     /* 15(n-2)^3 / (8(n-1)(n-3)(..)) (lpc.c:283)           */
    double acorr_norm;     /* 2.0 / n                       (lpc.c:335)           */
} SrlaGeom;

typedef struct SrlaItemDesc {
    uint32_t sample_off;   /* first sample, relative to the job's input          */
    uint32_t n;
    uint32_t variant;      /* 0..nch-1 plain channel, nch = M, nch + 1 = S       */
    uint32_t geom;         /* index into the SrlaGeom table                      */
    uint64_t res_off;      /* element offset of this item's residual in scratch  */
    int32_t  forced_order; /* -1, or the order the host arbitrated (tie path)    */
    uint32_t lshift;       /* offset left shift of the item's stream (header.offset_lshift, srla_utility.c:177) */
    uint32_t forced_ltp;   /* 0, or the LTP taps the host arbitrated: bit 31 | three 6-bit fields (stream order)  */
    uint32_t seg;          /* segment (stream part) of the job the item belongs to */
    uint32_t forced_svr;   /* 0, or 1 + row of the job's table of SVR-refined predictors the host arbitrated (host_ties.cpp) */
    /* chain / history mode with the SVR refinement on (host_chain.cpp): what the reference's persistent buffer holds in its
     * first n words once the item's analysis is over -- the refinement's `residual` (lpc.c:1047,1095-1106), or, where the
     * refinement does not run (order 0, singular matrix), what the LPC-pass autocorrelation left there */
    uint32_t svr_dump;     /* 1 + pool offset of the n words this item leaves, 0: not kept */
    uint32_t svr_under;    /* 1 + pool offset of the buffer its LPC-pass call left */
    uint32_t pad0;
} SrlaItemDesc;

/* what the SVR kernels need beyond the solve chain's arguments (all optional) */
typedef struct SrlaSvrExtra {
    uint32_t *ties;              /* the job's near-tie list (kind 2 entries) */
    const double *forced_rows;   /* rows of 256 doubles: predictors the host refined with its own libm (SrlaItemDesc::forced_svr) */
    double *chain_pool;          /* chain / history mode: where the calls leave their buffers (SrlaItemDesc::svr_dump / svr_under) */
    const uint32_t *select;      /* chain / history mode with SVR on: the solve chain runs round by round; only the items with */
    uint32_t round;              /* select[item] == round take part (null: all) */
} SrlaSvrExtra;

/* What srla_autocorr needs to know about an item, in one record (item descriptor + the constants of its block
 * length): one load at the head of the workgroup instead of a chain of three dependent ones.  The host keeps one
 * array of these per FFT-size class. */
typedef struct SrlaAutocorrItem {
    uint32_t item;          /* index into the item / result tables */
    uint32_t sample_off;
    uint32_t n;
    uint32_t variant;
    uint32_t nfft;
    uint32_t tw_off;
    uint32_t chain_src;     /* chain mode: 1 + pool offset of the word the odd block's middle sample inherits, 0: zero */
    uint32_t chain_dump;    /* chain mode: 1 + pool offset where the call leaves its whole FFT buffer, 0: not kept  */
    uint32_t chain_lags;    /* chain mode, LTP lags of an FFT shorter than the 263 lags: 1 + index into the gather table
                             * (one entry per lag from nfft on: 1 + pool offset of the word the reference reads, lpc.c:371-373) */
    uint32_t lshift;        /* offset left shift of the item's stream */
    double   welch_divisor;
    double   acorr_norm;
} SrlaAutocorrItem;        /* 56 bytes */

/* What kernel A leaves per item (everything SRLAEncoderCoefficient carries + costs). */
typedef struct SrlaItemResult {
    int32_t  preemph_prev;
    int32_t  preemph_coef;
    uint32_t lpc_order;
    uint32_t lpc_rshift;
    uint32_t use_sum;
    uint32_t ltp_period;
    int32_t  ltp_coef[3];
    uint32_t code_length;   /* bits of this channel inside a compress payload */
    uint32_t res_code_type;
    uint32_t res_porder;
    uint32_t res_bits;
    uint32_t flags;
    uint32_t pad[2];
    int8_t   lpc_coef[256];      /* reversed tap order, as written to the stream      */
    uint8_t  kparam[1024];       /* Rice k / recursive-Rice k2 per partition of porder */
} SrlaItemResult;               /* 64 + 256 + 1024 = 1344 bytes */

#define SRLA_PACK_SLACK 128u    /* bytes of slack per block slot in the global pack scratch */

/* Job summary the device writes into pinned host memory (srla_block_offsets / srla_pack_blocks). */
typedef struct SrlaJobInfo {
    uint32_t total_bytes;    /* bytes of all blocks of the job                          */
    uint32_t base;           /* stream offset of the first block of the job's first segment */
    uint32_t num_blocks, num_raw, num_silent;
    uint32_t num_tie_items, num_odd_items;   /* flagged items of the job (any candidate) / odd-length items among the chosen blocks */
    uint32_t error;          /* SRLA_JOBERR_* bits (OVERFLOW: in at least one segment) */
} SrlaJobInfo;
/* What the host needs to arbitrate a job's near-ties (host_ties.cpp), gathered by srla_block_offsets into pinned host memory so that
 * the host reads it where it stands instead of fetching it with a dozen blocking copies after the call's last job (0.16 ms of a
 * 600 s call).  out: `cap` doubles -- the list entries (item | kind << 30) -- then per entry max(P + 2, 8) doubles: an order tie's
 * error variances of orders 0..P and the order the device chose; an LTP tie's eight numbers (tie_data).  Jobs with more than `cap`
 * entries (the tests' widened thresholds) are fetched as before. */
#define SRLA_TIE_GATHER_CAP 16u
typedef struct SrlaTieGather {
    const double *err;       /* [order][item] */
    const double *tie_data;  /* [entry][8], may be null */
    double *out;
    uint32_t cap, num_items;
} SrlaTieGather;
/* per segment, behind SrlaJobInfo and the per-window byte counts */
typedef struct SrlaSegInfo {
    uint32_t bytes;          /* bytes of the segment's blocks                                           */
    uint32_t pos;            /* offset of its first block in the stream's output buffer                  */
    uint32_t stage_off;      /* where the segment starts in the job's staging buffer                     */
    uint32_t skip;           /* 1: nothing written (the stream would not fit its buffer, or an earlier job of it did not) */
} SrlaSegInfo;
#define SRLA_JOBERR_OVERFLOW 1u  /* the stream would not fit the caller's buffer: nothing written from here on */
#define SRLA_JOBERR_SIZE     2u  /* a packed block differs from its computed size (internal error)           */
#define SRLA_JOBERR_COVER    4u  /* a window's chosen blocks do not tile it (internal error)                 */

typedef struct SrlaCandDesc {
    uint32_t window;
    uint32_t node_i, node_j;
    uint32_t sample_off;
    uint32_t n;
    uint32_t item_base;     /* first item (variant 0) or 0xFFFFFFFF when not analysed (RAW by length) */
    uint32_t raw_silence;   /* 0: the items' own flags say whether the block is silent (every stream whose samples obey its offset
                             * shift); 1 / 2: the host looked at the RAW samples -- silent / not.  The reference decides silence before
                             * the shift (srla_encoder.c:783-791); a block call under a shift left by an earlier EncodeWhole can hold
                             * samples that are zero only after it */
    uint32_t pad1;
} SrlaCandDesc;

typedef struct SrlaWindowDesc {
    uint32_t sample_off;
    uint32_t n;
    uint32_t cand_base;
    uint32_t num_cands;
    uint32_t num_nodes;
    uint32_t block_base;    /* first slot of this window in the block table (num_nodes - 1 slots) */
    uint32_t seg;           /* segment (stream part) of the job the window belongs to */
    uint32_t pad1;
} SrlaWindowDesc;

/* A job holds windows of one or more streams: a segment is the run of consecutive windows that belong to one stream.
 * The blocks of a segment go, back to back, to the stream's output buffer. */
typedef struct SrlaSegDesc {
    uint32_t first_window, num_windows;
    uint32_t stream;        /* index into the device-resident stream positions (running offset + sticky skip flag) */
    uint32_t use_init;      /* 1: the segment starts at init_pos (first job of the stream in this pass), 0: at the running offset */
    uint32_t init_pos;
    uint32_t limit;         /* size of the stream's output buffer */
    uint64_t dst;           /* device-visible address of the stream's output buffer, 0: the job's pinned staging buffer */
} SrlaSegDesc;              /* 32 bytes */

/* One chosen block (srla_price_windows output, stream order inside a window). */
typedef struct SrlaBlockRecord {
    uint32_t valid;         /* 0 = unused slot */
    uint32_t sample_off;
    uint32_t n;
    uint32_t block_type;
    uint32_t ch_method;
    uint32_t bytes;         /* total block size incl. the 11-byte header */
    uint32_t item[SRLA_MAX_CH];  /* item index per output channel (compress blocks) */
    uint32_t seg;           /* segment of the job the block belongs to */
    uint32_t price;         /* what the block division search paid for the block = what SRLAEncoder_ComputeBlockSize returns: with more
                             * than two channels the reference prices only the first two (srla_encoder.c:1287-1301, :1519-1532) */
} SrlaBlockRecord;          /* 64 bytes */

typedef struct SrlaJobParams {
    uint32_t num_channels;
    uint32_t bits_per_sample;
    uint32_t num_segs;        /* segments (stream parts) of the job */
    uint32_t max_order;       /* preset's max_num_parameters */
    uint32_t order_fixed;     /* preset 0: MAX_FIXED tactic   */
    uint32_t ltp_order;       /* 0, 1, 3 */
    uint32_t num_samples;     /* per channel, of the job's input */
    uint32_t channel_stride;  /* elements between channel planes of the input */
    uint32_t max_block;
    uint32_t min_block;
    uint32_t num_items;
    uint32_t num_cands;
    uint32_t num_windows;
    uint32_t keep_residuals;  /* srla_residual_cost stores every item's residual and srla_pack_blocks reads the chosen ones (1: the
                               * default, zig-zag mapped uint16 where a block's values fit, SRLA_ITEM_RES_U16; 2: always the int32
                               * residual -- what SRLAMI355X_ProbeBlock returns, SRLA_MI355X_RES32); 0 (SRLA_MI355X_RECOMPUTE_RESIDUALS): none are
                               * stored and srla_pack_blocks recomputes the chosen blocks' (blocks > 8192 samples always keep) */
    uint32_t crowded;         /* the job runs beside other jobs' wide kernels (a call of several jobs): kernels that only fit an
                               * empty SIMD take their lean form whatever the job's size (srla_lpc_errvars_lean) */
    uint32_t rc_lo, rc_hi;    /* srla_residual_cost: when rc_hi != 0 the launch takes the items with rc_lo < n <= rc_hi only (a job with blocks
                               * above 4096 samples is analysed by two launches: the register budget of the 8192-sample form, two
                               * wavefronts per SIMD, would otherwise be every item's) */
    const double *welch_tab;  /* when non-null: the Welch window's weights of the blocks that fill their transform, n = 1024, 2048, 4096,
                               * 8192 (lpc.c:256-266 with the sample's scaling folded in: (4 (n-1)^-2 2^-(bps-1) e) (n-1-e), e < n / 2 -- the
                               * second half mirrors the first), n / 2 doubles each from word n / 2 - 512: the same products the kernel
                               * would form, made once on the host (SRLA_WELCH_TAB_WORDS) */
    const uint32_t *lshift_dev; /* when non-null the offset left shift is read from here (device memory) instead of the
                               * item's: lets a whole device-resident stream be enqueued before its OR-reduction has finished */
    /* near-tie detection (H2: decisions that hang on libm): an item is flagged, and arbitrated with the host libm, when the
     * two best code-length estimates differ by less than tie_rel (relative), or an LTP tap lies within tie_ltp of a
     * rounding boundary.  tie_logscale (1.0 in production) and tie_ltpbias (0.0) deliberately falsify the device's log and
     * its scaled LTP taps: the tests use them to make the device decide differently from the host, so that the arbitration
     * has work to do. */
    double tie_rel, tie_ltp, tie_logscale, tie_ltpbias;
} SrlaJobParams;

/* LDS carve-up of kernel A for one FFT-size group (bytes, 16-byte aligned); host decides overlays */
typedef struct SrlaLdsPlan {
    uint32_t y_off;        /* int32 [n]      pre-emphasised signal                         */
    uint32_t fft_off;      /* double [nfft]  FFT buffer; later zig-zag residual u32 [n]    */
    uint32_t lev_off;      /* double [4*(order+3)] Levinson scratch                        */
    uint32_t means_off;    /* double [2^(max_porder+1)] partition means                    */
    uint32_t small_off;    /* fixed-size scalars, lags, reductions                         */
    uint32_t total;
} SrlaLdsPlan;

#endif /* SRLA_DEVICE_LAYOUT_H */
