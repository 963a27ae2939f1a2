/*
 * kernels_common.h -- what every kernel file of the library includes: the device data layout, the launchers' prototypes, the shared
 * device helpers, wave priority of the narrow kernels, the launchers' tuning record and the phase timer of the diagnostic build.
 * Every kernel file is compiled with -ffp-contract=off and pins it again here: the reference is C90 (no fused multiply-add), and
 * every integer in the stream is decided by double arithmetic.
 */
#ifndef SRLA_KERNELS_COMMON_H
#define SRLA_KERNELS_COMMON_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <float.h>
#include <algorithm>
#include <type_traits>

#include "device_layout.h"
#include "kernels.h"

#pragma clang fp contract(off)

/* The narrow kernels of stream N (a few hundred latency-bound wavefronts: serial fp64 recursions, the window pricing) share their
 * SIMDs with the wide kernels' wavefronts, which issue VALU instructions back to back: at the default priority a recursion's
 * next instruction waits its turn behind them and the solve stage of a job took 0.35-0.53 ms beside them against 0.1 ms alone --
 * longer than the wide stream had work for, so srla_residual_cost of the job waited for it (timeline).  Raised
 * wave priority lets the few instructions they have go first; they are too few to slow the wide kernels down. */
#define NARROW_KERNEL_PRIORITY() __builtin_amdgcn_s_setprio(3)

#include "device_common.h"

/* what the launchers take from the environment (host_tuning.cpp -> srla_set_launch_tuning, stage_in.hip) */
extern SrlaLaunchTuning g_srla_tune;

#define SET_LDS_ATTR(fn)                                                                                     \
    do {                                                                                                     \
        static bool done_ = false;                                                                           \
        if (!done_) { (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done_ = true; } \
    } while (0)

/* -DSRLA_DIAG_PHASES (tools/phase_probe.py; never in the shipped library): where the wavefronts of a wide kernel spend their time IN
 * FLIGHT -- every wavefront stamps the shader clock at phase boundaries and adds the differences to a per-workgroup row of the
 * file's table; SRLA_DIAG_PHASE_READER(name) defines SRLAMI355X_DiagPhases_<name>, which copies the table out and clears it. */
#ifdef SRLA_DIAG_PHASES
static __device__ unsigned long long srla_diag_phase[1024][16];
#define PHASE_INIT() unsigned long long ph_t_ = __builtin_amdgcn_s_memtime()
#define PHASE_PARAM , unsigned long long &ph_t_
#define PHASE_ARG , ph_t_
#define PHASE(K) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63u) == 0) atomicAdd(&srla_diag_phase[blockIdx.x & 1023u][K], t_ - ph_t_); ph_t_ = t_; } while (0)
#define SRLA_DIAG_PHASE_READER(NAME)                                                                         \
    extern "C" int SRLAMI355X_DiagPhases_##NAME(unsigned long long *out /* [16] */)                         \
    {                                                                                                        \
        static unsigned long long host[1024][16];                                                            \
        if (hipMemcpyFromSymbol(host, HIP_SYMBOL(srla_diag_phase), sizeof host) != hipSuccess) return -1;    \
        for (int p = 0; p < 16; p++) { unsigned long long t = 0; for (int r = 0; r < 1024; r++) t += host[r][p]; out[p] = t; } \
        memset(host, 0, sizeof host);                                                                        \
        return hipMemcpyToSymbol(HIP_SYMBOL(srla_diag_phase), host, sizeof host) == hipSuccess ? 0 : -1;     \
    }
#else
#define PHASE_INIT() do { } while (0)
#define PHASE(K) do { } while (0)
#define PHASE_PARAM
#define PHASE_ARG
#define SRLA_DIAG_PHASE_READER(NAME)
#endif

#endif /* SRLA_KERNELS_COMMON_H */
