/*
 * srla_mi355x.h -- C ABI of libsrla_mi355x.so, the MI355X (gfx950) drop-in for the encode path
 * of aikiriao/SRLA (codec version 18, format version 10).
 *
 * Part 1 re-declares, with identical names, signatures, struct layouts and error behaviour,
 * the encoder API of the reference: every entry point cites the reference declaration it
 * replaces (paths relative to the reference tree).  A program that links the reference's
 * libsrlacodec.a for encoding can link this library instead; `srla -d` decodes the output
 * bit-identically.
 *
 * Part 2 adds what a GPU implementation needs beyond the reference's interface: encoding from
 * samples that already live in HBM, device selection, statistics, and a stage-level probe used
 * by the parity tests.  Plain pointers and sizes only.
 */
#ifndef SRLA_MI355X_H_INCLUDED
#define SRLA_MI355X_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Part 1 -- the reference's encoder interface
 * ---------------------------------------------------------------------------------------- */

/* include/srla.h:7-22 */
#define SRLA_FORMAT_VERSION         10
#define SRLA_CODEC_VERSION          18
#define SRLA_HEADER_SIZE            30
#define SRLA_MAX_NUM_CHANNELS       8
#define SRLA_MAX_COEFFICIENT_ORDER  255
#define SRLA_MAX_LTP_ORDER          3
#define SRLA_NUM_PARAMETER_PRESETS  7

/* include/srla.h:29-38 */
typedef enum SRLAApiResultTag {
    SRLA_APIRESULT_OK = 0,
    SRLA_APIRESULT_INVALID_ARGUMENT,
    SRLA_APIRESULT_INVALID_FORMAT,
    SRLA_APIRESULT_INSUFFICIENT_BUFFER,
    SRLA_APIRESULT_INSUFFICIENT_DATA,
    SRLA_APIRESULT_PARAMETER_NOT_SET,
    SRLA_APIRESULT_DETECT_DATA_CORRUPTION,
    SRLA_APIRESULT_NG
} SRLAApiResult;

/* include/srla.h:41-51 */
struct SRLAHeader {
    uint32_t format_version;
    uint32_t codec_version;
    uint16_t num_channels;
    uint32_t num_samples;
    uint32_t sampling_rate;
    uint16_t bits_per_sample;
    uint8_t  offset_lshift;
    uint32_t max_num_samples_per_block;
    uint8_t  preset;
};

/* include/srla_encoder.h:8-18 */
struct SRLAEncodeParameter {
    uint16_t num_channels;
    uint16_t bits_per_sample;
    uint32_t sampling_rate;
    uint32_t min_num_samples_per_block;
    uint32_t max_num_samples_per_block;
    uint32_t num_lookahead_samples;
    uint32_t ltp_order;
    uint32_t num_svr_filter_learning_iteration;
    uint8_t  preset;
};

/* include/srla_encoder.h:21-27 */
struct SRLAEncoderConfig {
    uint32_t max_num_channels;
    uint32_t min_num_samples_per_block;
    uint32_t max_num_samples_per_block;
    uint32_t max_num_lookahead_samples;
    uint32_t max_num_parameters;
};

/* include/srla_encoder.h:30 */
struct SRLAEncoder;

/* include/srla_encoder.h:33-34 -- invoked synchronously on the calling thread, once per
 * look-ahead window, in stream order, with a pointer into the caller's output buffer */
typedef void (*SRLAEncoder_EncodeBlockCallback)(
    uint32_t num_samples, uint32_t progress_samples, const uint8_t *encoded_block_data, uint32_t block_data_size);

/* include/srla_encoder.h:41-42 (libs/srla_encoder/src/srla_encoder.c:85) */
SRLAApiResult SRLAEncoder_EncodeHeader(const struct SRLAHeader *header, uint8_t *data, uint32_t data_size);

/* include/srla_encoder.h:45 (srla_encoder.c:468): size of the HOST handle only; device and
 * pinned memory are owned by the library and sized from the same config at Create */
int32_t SRLAEncoder_CalculateWorkSize(const struct SRLAEncoderConfig *config);

/* include/srla_encoder.h:48 (srla_encoder.c:549) */
struct SRLAEncoder *SRLAEncoder_Create(const struct SRLAEncoderConfig *config, void *work, int32_t work_size);

/* include/srla_encoder.h:51 (srla_encoder.c:697) */
void SRLAEncoder_Destroy(struct SRLAEncoder *encoder);

/* include/srla_encoder.h:54-55 (srla_encoder.c:710) */
SRLAApiResult SRLAEncoder_SetEncodeParameter(struct SRLAEncoder *encoder, const struct SRLAEncodeParameter *parameter);

/* include/srla_encoder.h:58-60 (srla_encoder.c:1477).  As in the reference: with one or two channels the size SRLAEncoder_EncodeBlock
 * writes; with more, the price of the block's first two channels (srla_encoder.c:1287-1301 adds up only those, :1519-1532 returns the
 * sum) -- the number the block division search works with.  Like every entry point below it reads and leaves the handle's persistent
 * analysis buffer as the reference's call does (DESIGN.md 4 "the buffer from call to call"). */
SRLAApiResult SRLAEncoder_ComputeBlockSize(
    struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples, uint32_t *output_size);

/* include/srla_encoder.h:63-66 (srla_encoder.c:1549) */
SRLAApiResult SRLAEncoder_EncodeBlock(
    struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
    uint8_t *data, uint32_t data_size, uint32_t *output_size);

/* include/srla_encoder.h:69-72 (srla_encoder.c:1646) */
SRLAApiResult SRLAEncoder_EncodeOptimalPartitionedBlock(
    struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
    uint8_t *data, uint32_t data_size, uint32_t *output_size);

/* include/srla_encoder.h:75-79 (srla_encoder.c:1701) */
SRLAApiResult SRLAEncoder_EncodeWhole(
    struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
    uint8_t *data, uint32_t data_size, uint32_t *output_size, SRLAEncoder_EncodeBlockCallback encode_callback);

/* ------------------------------------------------------------------------------------------
 * Part 2 -- MI355X extensions
 * ---------------------------------------------------------------------------------------- */

/* Select the HIP device used by encoders created afterwards on this thread's process
 * (one process per GPU is the intended deployment; default device 0). Returns 0 on success. */
int SRLAMI355X_SetDevice(int device_index);

/* Number of host pool threads (default: min(8, usable CPUs / 2), or the SRLA_MI355X_PACK_THREADS environment
 * variable).  The bitstream is assembled on the device; the pool only stages pageable input planes into pinned
 * memory (packing them to int16 where they fit) and copies finished blocks out of the pinned staging buffer when
 * the caller's output buffer is pageable.  With several processes per node (one per GPU) give each process
 * hardware_threads / processes; with device-resident input and a pinned output buffer the pool is idle. */
void SRLAMI355X_SetPackThreads(struct SRLAEncoder *encoder, uint32_t num_threads);

/* EncodeWhole for samples already resident in HBM: d_input is a device pointer to planar
 * int32 [num_channels][channel_stride] (channel_stride >= num_samples).  Output contract is
 * SRLAEncoder_EncodeWhole's. */
SRLAApiResult SRLAMI355X_EncodeWholeDevice(
    struct SRLAEncoder *encoder, const int32_t *d_input, uint32_t channel_stride, uint32_t num_samples,
    uint8_t *data, uint32_t data_size, uint32_t *output_size, SRLAEncoder_EncodeBlockCallback encode_callback);

/* A RANGE of a stream's look-ahead windows, without the stream header: exactly the bytes SRLAEncoder_EncodeWhole
 * (libs/srla_encoder/src/srla_encoder.c:1756-1783) writes for these windows when the whole stream has the offset left
 * shift `offset_lshift` (the trailing zeros of the OR of ALL its samples, srla_utility.c:177).  With it one long stream
 * is sharded over several GPUs: cut it at multiples of the look-ahead (num_lookahead_samples; the maximum block size when
 * min == max), give every process / GPU a contiguous range, concatenate the results in order behind the 30-byte header
 * (SRLAEncoder_EncodeHeader) -- srla_amd/multigpu.py does exactly that with one rank per GPU.  input points at the
 * first sample of the range.  is_stream_end: the range ends the stream (only then may it hold a partial window; give it
 * the stream's last two windows at least, the reference's analysis of an odd-length last window looks back one block). */
/* The OR of every sample of `input` (planar, the handle's channel count, num_samples per channel), computed by the handle's
 * host threads: what a rank contributes to the stream's offset left shift before SRLAMI355X_EncodeWindows (the shift is the
 * number of trailing zeros of the OR over ALL ranks' ranges, srla_utility.c:177-203). */
SRLAApiResult SRLAMI355X_OrMask(struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples, uint32_t *mask);

SRLAApiResult SRLAMI355X_EncodeWindows(
    struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples, uint32_t offset_lshift, int is_stream_end,
    uint8_t *data, uint32_t data_size, uint32_t *output_size);

/* Many streams in one call -- the way to encode a corpus of short files: windows of different streams share the device
 * jobs, so a 10 s file costs what 10 s of a long stream cost instead of a whole pipeline fill and drain of its own.
 * Every stream is what SRLAEncoder_EncodeWhole would write for it (tools/srla_codec/srla_codec.c:134 called once per
 * file), byte for byte, into its own buffer data[i] of data_size[i] bytes; output_size[i] receives its size.
 * inputs[i]: the stream's planar int32 channel pointers (host memory), num_samples[i] its length; all streams share the
 * handle's parameters (channels, bits per sample, sampling rate, preset, block sizes).
 * A stream whose buffer is too small gets results[i] = SRLA_APIRESULT_INSUFFICIENT_BUFFER (output_size[i] = 0) without
 * disturbing the others, and the call returns that code; results may be NULL. */
SRLAApiResult SRLAMI355X_EncodeBatch(
    struct SRLAEncoder *encoder, uint32_t num_streams, const int32_t *const *const *inputs, const uint32_t *num_samples,
    uint8_t *const *data, const uint32_t *data_size, uint32_t *output_size, SRLAApiResult *results);

/* The same with the OR of every stream's samples supplied by the caller (sample_or[i]; NULL: gathered by the library).  A
 * front end that touches every sample anyway -- a WAV reader de-interleaving into planes -- gets the OR for free; with it and
 * with input planes in pinned memory (below) the library's host threads do no per-sample work at all: the planes are read by
 * DMA where they lie. */
SRLAApiResult SRLAMI355X_EncodeBatchEx(
    struct SRLAEncoder *encoder, uint32_t num_streams, const int32_t *const *const *inputs, const uint32_t *num_samples,
    const uint32_t *sample_or, uint8_t *const *data, const uint32_t *data_size, uint32_t *output_size, SRLAApiResult *results);

/* The same for streams given as what a WAV `data` chunk holds: interleaved little-endian PCM frames (bytes_per_sample = 1, 2 or 3
 * and equal to the handle's bits_per_sample / 8; 8-bit samples are unsigned with offset 128, the others signed -- the conversion
 * of libs/wav/src/wav.c:707 that tools/srla_codec/srla_codec.c:75-134 runs before SRLAEncoder_EncodeWhole).  The frames cross the
 * link as they are and are de-interleaved, widened and OR-reduced on the device: a front end only has to read() its files into
 * memory.  Frames in page-locked memory (SRLAMI355X_AllocHost) are read where they lie; pageable ones are locked in place for the
 * call (or, where that is refused, de-interleaved on the host).  frames[i] points at the first frame of stream i. */
SRLAApiResult SRLAMI355X_EncodeBatchPcm(
    struct SRLAEncoder *encoder, uint32_t num_streams, const void *const *frames, const uint32_t *num_samples,
    uint32_t bytes_per_sample, uint8_t *const *data, const uint32_t *data_size, uint32_t *output_size, SRLAApiResult *results);

/* Pinned (page-locked, device-visible) host memory for callers that do not link HIP themselves: input planes in it are read
 * by DMA without a staging copy, output buffers in it are written by the device directly.  NULL when no device is usable. */
void *SRLAMI355X_AllocHost(size_t bytes);
void SRLAMI355X_FreeHost(void *p);

struct SRLAMI355XStats {
    uint64_t num_windows;
    uint64_t num_candidates;
    uint64_t num_items;          /* analysed (candidate, channel variant) pairs          */
    uint64_t num_blocks;         /* blocks written                                       */
    uint64_t num_raw_blocks;
    uint64_t num_silent_blocks;
    uint64_t num_tie_items;      /* items whose order / LTP tap decision was within the libm tolerance (arbitrated on the host) */
    uint64_t num_odd_items;      /* chosen items with an odd block length                */
    uint64_t analyze_launches;   /* jobs enqueued (one launch of srla_residual_cost each)  */
    double   analyze_ms;         /* autocorr_ms + solve_ms + residual_ms                   */
    double   price_ms;           /* srla_price_windows (timed jobs only)                   */
    double   gather_ms;          /* srla_block_offsets + srla_pack_blocks (timed jobs only) */
    double   h2d_ms;             /* host wall time spent enqueueing                        */
    double   history_ms;         /* host wall time inside history mode (parameters whose analysis depends on earlier blocks: window by window) */
    double   pack_ms;            /* host wall time collecting finished jobs                */
    double   total_ms;           /* host wall time inside Encode*                        */
    uint64_t analyzed_samples;   /* sum of item lengths                                  */
    double   autocorr_ms;        /* srla_autocorr launches (+ srla_pitch_solve), timed jobs only */
    double   solve_ms;           /* recursion + order selection + quantiser, timed jobs only  */
    double   residual_ms;        /* srla_residual_cost, timed jobs only (every job until round 5: a start event per launch is not free) */
    uint64_t timed_jobs;         /* jobs on which every stage was timed (one in four; a call of ONE job: every fourth such call) */
    uint64_t num_tie_resolved;   /* flagged items (any candidate, not only chosen ones) whose decision the host libm confirmed */
    uint64_t num_tie_overrides;  /* flagged items where the host libm decided otherwise: re-analysed with the host's decision */
    uint64_t num_restarts;       /* times the stream loop went back to a job because of such an override            */
    uint64_t num_inplace_pins;   /* pageable input planes page-locked in place for a call (instead of staged) */
    uint64_t num_inplace_out_pins;  /* pageable output buffers page-locked in place for a call (written by the device directly) */
    /* Bit-identity with the reference is the contract.  Where a call runs under parameters for which the output is valid and
     * lossless but NOT guaranteed to be the reference's bytes, SRLAEncoder_SetEncodeParameter says so on stderr, every
     * Encode* call made under them is counted here and the reasons are OR-ed into nonidentical_reasons (SRLAMI355X_NONIDENTICAL_*). */
    uint64_t num_nonidentical_calls;
    uint64_t nonidentical_reasons;
    uint64_t num_svr_tie_items;     /* SVR refinement: items with an objective comparison inside the libm tolerance (arbitrated on the host) */
    uint64_t num_history_windows;   /* look-ahead windows encoded in history mode (in the reference's own call order, window by window) */
    double   pitch_ms;              /* srla_pitch_solve, timed jobs only (jobs whose stage A runs in two parts; otherwise inside autocorr_ms) */
    uint64_t num_hybrid_jobs;       /* jobs of input planes locked in place of which half the channels went through the int16 staging buffer */
};
/* reasons (SRLAMI355XStats::nonidentical_reasons, SRLAMI355X_NonIdenticalReasons) */
#define SRLAMI355X_NONIDENTICAL_SVR_HISTORY 1u  /* SVR refinement on, in a window whose blocks depend on the calls before them (odd lengths,
                                                 * short LTP blocks): an item whose objective comparisons the HOST libm decided differently
                                                 * from the device gets the host's predictor, but leaves the transform's words -- not the
                                                 * refinement's residual (lpc.c:1047) -- for the blocks after it.  Counted where it happens
                                                 * (never seen on real input); the parameters alone no longer carry this reason. */
#define SRLAMI355X_NONIDENTICAL_LTP_TINY_BUFFER 2u  /* long-term predictor on an encoder created for blocks of at most 256 samples: the
                                                     * reference's FFT buffer is then shorter than the 263 lags it copies out of it
                                                     * (lpc.c:371-373 reads beyond the buffer, into the transform's scratch area) */
#define SRLAMI355X_NONIDENTICAL_HANDLE_HISTORY 4u  /* a block of this call inherits a word of the reference's persistent FFT buffer
                                                    * (encoder->lpcc: one calculator per handle, lpc.c:58,211) that an EARLIER call on
                                                    * the same handle left, and the library does not know that word.  The calls of the
                                                    * reference's entry points read and leave the handle's buffer as the reference's do
                                                    * (history-mode calls exactly; calls of the regular pipeline through a capture of their
                                                    * samples, encoded once more when a later call is about to read the buffer); what stays
                                                    * unknown are words a stream's kept windows did not rewrite and the buffer after a call
                                                    * that failed.  Counted where it happens. */
#define SRLAMI355X_NONIDENTICAL_SEARCH_BEYOND_PARAMETERS 8u  /* a block division search on an encoder CREATED for a larger maximum block than its
                                                             * parameters name: the reference's search runs up to the configuration's maximum
                                                             * (srla_encoder.c:598, :1669; SetEncodeParameter updates the minimum and the look-ahead
                                                             * only, :745-746), SRLAEncoder_ComputeBlockSize refuses the candidates beyond the
                                                             * parameters' (:1499) and SRLAEncoder_EncodeWhole returns SRLA_APIRESULT_NG -- in every
                                                             * case tried.  This library searches within the parameters and succeeds: valid and
                                                             * lossless bytes where the reference has none.  Counted per call. */
/* The reasons for which a stream of `num_samples` samples per channel encoded under the handle's current parameters would not be
 * guaranteed bit-identical to the reference (0: it is); num_samples = 0 asks about the parameters alone. */
uint32_t SRLAMI355X_NonIdenticalReasons(struct SRLAEncoder *encoder, uint32_t num_samples);
/* cumulative since Create or the last reset */
void SRLAMI355X_GetStats(struct SRLAEncoder *encoder, struct SRLAMI355XStats *stats, int reset);
/* The same for a caller that may have been built against another revision of this header (the structure only ever grows at its
 * end): at most `stats_bytes` bytes are written; returns the structure's size in THIS library (0: no such handle), so a caller
 * can tell which of its fields were filled.  SRLAMI355X_GetStats writes sizeof(struct SRLAMI355XStats) of the header it was
 * compiled with -- rebuild against the current header when the library is updated. */
uint32_t SRLAMI355X_GetStatsSized(struct SRLAEncoder *encoder, void *stats, uint32_t stats_bytes, int reset);

/* Stage-level probe for the parity tests: analyses one block exactly as the block-division
 * search would and returns every channel variant.  variants = num_channels (+2 when >= 2:
 * [plain channels..., M, S]).  records: variants * SRLAMI355X_ITEM_RECORD_BYTES;
 * residuals: variants * num_samples int32; debug: variants * SRLAMI355X_DEBUG_DOUBLES doubles
 * (LPC lags, error variances, per-order length estimates, LTP lags); any may be NULL. */
#define SRLAMI355X_ITEM_RECORD_BYTES 1344
#define SRLAMI355X_DEBUG_DOUBLES     1040
SRLAApiResult SRLAMI355X_ProbeBlock(
    struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
    void *records, int32_t *residuals, double *debug);

/* Library identification string ("srla-mi355x <version> gfx950 ..."). */
const char *SRLAMI355X_Version(void);

/* ---- test hooks -------------------------------------------------------------------------------------------------------------
 * The host-libm arbitration arithmetic of the library (srla_amd/csrc/host_ties.cpp) on its own: no handle, no device.  Not part
 * of the drop-in surface; exported so that a CPU-only test can compare it bit for bit with the oracle on the oracle's inputs.
 *   TestSelectOrder  srla_encoder.c:934-957 on error_vars[0..max_order] * compensation -> the chosen order
 *   TestLtpTaps      lpc.c:1620-1645 + srla_encoder.c:1031-1047 from R(0), R(1), R(2), R(p-1), R(p), R(p+1) -> the three 6-bit
 *                    taps packed 6 bits each in stream order, or 0xFFFFFFFF (singular)
 *   TestSvrRefine    lpc.c:1036-1136 on the normalised block, predictor refined in place
 *   TestLevinson     lpc.c:379-441 -> the predictor of `order` from ridge-regularised lags */
uint32_t SRLAMI355X_TestSelectOrder(const double *error_vars, uint32_t max_order, double compensation, uint32_t num_samples, uint32_t bits_per_sample);
uint32_t SRLAMI355X_TestLtpTaps(const double *lags6, uint32_t ltp_order);
void SRLAMI355X_TestSvrRefine(const double *data, uint32_t num_samples, double *coef, uint32_t order, uint32_t max_iter);
void SRLAMI355X_TestLevinson(const double *lags_ridged, uint32_t order, double *coef);
/*   TestPlanJobs     the job plan a call of these streams would get (host_plan.cpp: plan_jobs; handle with parameters set, no device):
 *                    per job the words { buffer set, segments, samples per plane } and per segment { stream, first sample, samples, offset
 *                    in the job's planes }; returns the number of words written, or -1 (out too small, no parameters) */
/*   TestPack16       the staging copy of streams of at most 16 bits (host_support.cpp: pack16_or, AVX2 where the CPU has it): dst[i] =
 *                    (int16) src[i]; returns the OR of the samples, *wide != 0 iff a sample does not fit 16 bits */
uint32_t SRLAMI355X_TestPack16(int16_t *dst, const int32_t *src, uint32_t n, uint32_t *wide);
int SRLAMI355X_TestPlanJobs(struct SRLAEncoder *encoder, uint32_t num_streams, const uint32_t *num_samples, int device_input,
                            uint32_t *out, uint32_t cap_words);

#ifdef __cplusplus
}
#endif
#endif /* SRLA_MI355X_H_INCLUDED */
