/* kernels.h -- host-callable launchers of kernels.hip (plain C linkage, no HIP types leak
 * beyond hipStream_t). */
#ifndef SRLA_KERNELS_H
#define SRLA_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "device_layout.h"

/* optional per-item debug dump (doubles) used by the stage-level parity tests */
#define SRLA_DBG_LAGS     0      /* 256: LPC autocorrelation lags (before the ridge)   */
#define SRLA_DBG_ERRVARS  256    /* 256: compensated error variances per order         */
#define SRLA_DBG_LENS     512    /* 256: estimated code length per order               */
#define SRLA_DBG_LTPLAGS  768    /* 264: LTP autocorrelation lags                      */
#define SRLA_DBG_STRIDE   1040

/* ev_start / ev_stop (either may be null): events attached to the first / last kernel dispatch of the call itself
 * (hipExtLaunchKernel), i.e. start and completion of the kernels with no extra barrier packet in the stream. */
#ifdef __cplusplus
extern "C" {
#endif

uint32_t srla_kernel_small_a_bytes(void);
uint32_t srla_kernel_small_c_bytes(void);
uint32_t srla_kernel_fast_lds_bytes(uint32_t fl, uint32_t ltp_order, uint32_t bits_per_sample);   /* LDS of a workgroup of the 1024*fl-sample fast path of srla_residual_cost (one item, or the items that share it) */
#define SRLA_FIR_PAD 256

/* pass 0: LPC lags (initialises the item record unless an LTP pass ran first), pass 1: LTP lags.
 * One launch per FFT-size class (rclass = 0, 1, 2, 4 for N' <= 1024, 2048, 4096, 8192): class_items holds the `count`
 * items of the class. */
/* a small job's 4096- and 2048-point classes in one launch (not in chain mode) */
int srla_launch_autocorr_pair(hipStream_t stream, const SrlaJobParams *jp, const int32_t *input, const SrlaItemDesc *items,
                              const SrlaGeom *geoms, const void *twiddles, uint32_t pass, SrlaItemResult *results, double *lags_ws,
                              double *dbg, const SrlaAutocorrItem *items_4096, uint32_t count_4096,
                              const SrlaAutocorrItem *items_2048, uint32_t count_2048, hipEvent_t ev_start, hipEvent_t ev_stop);
int srla_launch_autocorr(hipStream_t stream, int rclass, const SrlaJobParams *jp, const int32_t *input,
                         const SrlaItemDesc *items, const SrlaGeom *geoms, const void *twiddles,
                         uint32_t pass, SrlaItemResult *results, double *lags_ws, double *dbg,
                         const SrlaAutocorrItem *class_items, uint32_t count, hipEvent_t ev_start, hipEvent_t ev_stop,
                         double *chain_pool /* null outside chain mode (device_layout.h: chain_src / chain_dump) */,
                         const uint32_t *chain_tab /* gather table of chain_lags */,
                         int exact_nfft /* rclass 0 only: every item has exactly 1024 points */);
/* items of 16384 / 32768 points (blocks above 8192 samples): the global-memory slow path; scratch: scratch_groups x nfft
 * complex doubles, one region per (persistent) workgroup */
int srla_launch_autocorr_big(hipStream_t stream, const SrlaJobParams *jp, const int32_t *input, const void *twiddles, uint32_t pass,
                             SrlaItemResult *results, double *lags_ws, double *dbg, const SrlaAutocorrItem *class_items,
                             uint32_t count, uint32_t nfft, hipEvent_t ev_start, hipEvent_t ev_stop, double *chain_pool,
                             const uint32_t *chain_tab, void *scratch, uint32_t scratch_groups);
/* the items of more than 8192 samples (big_items: their indices), which srla_residual_cost leaves alone */
uint32_t srla_residual_big_sig_words(uint32_t max_n);
int srla_launch_residual_cost_big(hipStream_t stream, const SrlaJobParams *jp, const int32_t *input, const SrlaItemDesc *items,
                                  const SrlaGeom *geoms, const double *rice_thresholds, int32_t *res_ws, SrlaItemResult *results,
                                  const uint32_t *big_items, uint32_t count, uint32_t max_n, hipEvent_t ev_start, hipEvent_t ev_stop,
                                  int32_t *sig_ws /* blocks above 32768 samples: count x srla_residual_big_sig_words(max_n) words */);
/* History mode (host_chain.cpp): folds the buffers the calls of a phase left in the chain pool into the first words of the pool
 * -- the reference's persistent FFT buffer as the next phase finds it.  Word i of [lo[k], hi[k]) comes from pool[src[k] + i]: the
 * segments say which call was the last to write which words (a transform writes its whole length, an SVR refinement the
 * block's n words). */
#define SRLA_COMMIT_SEGS 40
int srla_launch_chain_commit(hipStream_t stream, double *pool, const uint32_t *lo, const uint32_t *hi, const uint32_t *src, uint32_t nseg);
/* int16 planes (stride16 elements apart) -> int32 planes (n apart): host input of at most 16 bits per sample */
int srla_launch_widen16(hipStream_t stream, const int16_t *src, size_t stride16, int32_t *dst, uint32_t n, uint32_t num_channels);
/* ties / tie_data: the job's near-tie list (device_layout.h: SrlaJobParams::tie_rel), may be null */
int srla_launch_pitch_solve(hipStream_t stream, const SrlaJobParams *jp, const SrlaItemDesc *items, const double *lags_ws,
                            SrlaItemResult *results, hipEvent_t ev_start, hipEvent_t ev_stop,
                            const uint32_t *select /* null: every item; else only items with select[item] == round */, uint32_t round,
                            uint32_t *ties, double *tie_data);
int srla_launch_lpc_solve(hipStream_t stream, const SrlaJobParams *jp, const SrlaItemDesc *items,
                          const SrlaGeom *geoms, const double *lags_ws, double *err_ws, const uint8_t *huff_len,
                          SrlaItemResult *results, double *dbg, uint32_t *ties, hipEvent_t ev_start, hipEvent_t ev_stop,
                          const int32_t *input, double *coef_ws /* 64 doubles per item (256 for orders above 64) */,
                          uint32_t svr_iterations /* 0: off */, uint32_t svr_n_cap /* longest LDS-resident block of the job */,
                          void *svr_scratch, uint32_t svr_groups /* srla_svr_refine_big: groups x srla_svr_big_scratch_bytes(max block) */,
                          double *gamma_ws /* [order][item] like err_ws: reflection coefficients (orders 8 .. 64: the three-launch chain) */,
                          const SrlaSvrExtra *svr_extra /* may be null */);
size_t srla_svr_big_scratch_bytes(uint32_t n_max);
int srla_launch_residual_cost(hipStream_t stream, int rclass, const SrlaJobParams *jp, const int32_t *input,
                              const SrlaItemDesc *items, const SrlaGeom *geoms, const SrlaLdsPlan *plan,
                              const double *rice_thresholds, int32_t *res_ws, SrlaItemResult *results,
                              hipEvent_t ev_start, hipEvent_t ev_stop);

int srla_launch_price(hipStream_t stream, const SrlaJobParams *jp, const SrlaWindowDesc *windows,
                      const SrlaCandDesc *cands, const SrlaItemResult *results,
                      SrlaBlockRecord *blocks, hipEvent_t ev_start, hipEvent_t ev_stop,
                      uint32_t max_nodes /* of a window of the job */, uint32_t max_window_cands,
                      uint32_t *price_ws /* two words per candidate of the job; needed when max_window_cands > srla_price_lds_cands() */);
uint32_t srla_price_lds_cands(void);

/* srla_block_offsets + srla_pack_blocks + srla_stream_out: the job's blocks, complete, assembled in the device
 * buffer `stage` and then moved, segment by segment (device_layout.h: SrlaSegDesc), to their byte offsets of their
 * streams' output buffers.
 *   stream_pos  device u32[2] per stream: running output offset + sticky skip flag, carried from job to job
 *   segs        the job's segments (device copy); seg_ctl: device scratch, 8 words per segment
 *   host_stage  pinned staging buffer of the job, for segments whose stream has no device-visible output buffer (dst = 0):
 *               they land there at their stage_off and the host copies them out
 *   info / window_bytes / seg_info  job summary, per-window and per-segment results (device-visible pinned host memory)
 *   ties        the job's near-tie list (its count goes into the summary), may be null
 *   out_boost   > 1: that many times the usual number of stream-out workgroups (the last jobs of a stream, when the
 *               wide kernels are about to run dry and PCIe back-pressure no longer slows anything down)      */
int srla_launch_pack(hipStream_t stream, const SrlaJobParams *jp, uint32_t num_slots,
                     const int32_t *input, const SrlaItemDesc *items, const SrlaWindowDesc *windows,
                     const SrlaBlockRecord *blocks, const SrlaItemResult *results, const int32_t *res_ws,
                     const uint32_t *huff_code, const uint8_t *huff_len, uint32_t *block_off,
                     uint32_t *stream_pos, const SrlaSegDesc *segs, uint32_t *seg_ctl,
                     uint8_t *stage, uint8_t *host_stage, uint8_t *scratch, SrlaJobInfo *info,
                     uint32_t *window_bytes, SrlaSegInfo *seg_info, const uint32_t *ties,
                     hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t out_boost,
                     hipStream_t out_stream /* null: srla_stream_out on `stream` too */, hipEvent_t ev_packed /* the hand-over to out_stream */,
                     uint32_t no_stream_out /* 1: the launch ends with the assembly (ev_stop behind it); the host has the segments' bytes copied (hipMemcpyAsync) when it collects the job; 2: it ends with the assembly too, which stores every block where its stream wants it (a call's last job) */,
                     const SrlaTieGather *gather /* may be null */);
/* What the launchers take from the environment (read in ONE place, host_tuning.cpp, and handed over here): */
typedef struct {
    uint32_t pack_lds_cap_words;   /* SRLA_MI355X_PACK_LDS_WORDS: 0 = the default cap (24 Ki words) */
    uint32_t fft_wp;               /* SRLA_MI355X_FFT_WP (default 1): the region layout with wave-private FFT stages for classes of at most 4096 points */
    uint32_t fir_mfma;             /* SRLA_MI355X_FIR_MFMA: srla_residual_cost's FIR as a Toeplitz product on the matrix pipe */
} SrlaLaunchTuning;
void srla_set_launch_tuning(const SrlaLaunchTuning *t);
#define SRLA_SEGCTL_WORDS_HOST 8
uint32_t srla_pack_lds_words(const SrlaJobParams *jp);
int srla_pack_needs_scratch(const SrlaJobParams *jp);   /* blocks may exceed the LDS staging: allocate the scratch */

int srla_launch_deinterleave(hipStream_t stream, const void *src /* device */, uint32_t bytes_per_sample, uint32_t num_channels,
                             uint32_t count, int32_t *dst, size_t stride);   /* interleaved LE PCM frames -> planar int32 */
int srla_launch_or_accumulate(hipStream_t stream, const int32_t *in, size_t channel_stride, size_t count,
                              uint32_t num_channels, uint32_t *out);   /* *out |= OR of the samples */
int srla_launch_or_reduce(hipStream_t stream, const int32_t *in, size_t channel_stride, size_t count,
                          uint32_t num_channels, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif
