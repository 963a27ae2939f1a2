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
#define SRLA_DBG_TIMES    1032   /* 8: phase time stamps (100 MHz wall clock ticks)  */
#define SRLA_DBG_STRIDE   1040

#ifdef __cplusplus
extern "C" {
#endif

uint32_t srla_kernel_small_bytes(void);

int srla_launch_analyze(hipStream_t stream, int fft_regs_class, uint32_t num_items_in_group,
                        const SrlaJobParams *jp, const int32_t *input, const SrlaItemDesc *items,
                        uint32_t item_first, const SrlaGeom *geoms, const void *twiddles,
                        const SrlaLdsPlan *plan, const double *rice_thresholds, const uint8_t *huff_len,
                        int32_t *res_ws, SrlaItemResult *results, double *dbg);

int srla_launch_price(hipStream_t stream, const SrlaJobParams *jp, const SrlaWindowDesc *windows,
                      const SrlaCandDesc *cands, const SrlaItemResult *results,
                      SrlaBlockRecord *blocks, uint32_t *cand_bytes);

int srla_launch_gather(hipStream_t stream, const SrlaJobParams *jp, uint32_t num_slots,
                       const int32_t *input, const SrlaItemDesc *items, const SrlaBlockRecord *blocks,
                       const SrlaItemResult *results, const int32_t *res_ws, int32_t *out,
                       SrlaItemResult *chan_out);

int srla_launch_or_reduce(hipStream_t stream, const int32_t *in, size_t count, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif
