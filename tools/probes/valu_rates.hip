// Issue rates of the VALU / LDS instruction classes the two wide kernels are made of, measured on the device
// (round 6, for roofline.int_valu: profiles/r06/valu_rates.txt).
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/probes/valu_rates.hip && /tmp/valu_rates
//
// Every test is a kernel of WPS wavefronts per SIMD on every SIMD of the chip (256 CUs x 4), each running ITER iterations of
// 64 copies of ONE instruction on 8 independent register chains (no dependency stall shorter than 8 instructions); time by HIP
// events, rate = wave-instructions x 64 lanes / time.  Printed: cycles per wave-instruction and SIMD at the 2.4 GHz the chip
// advertises (the real clock under load is lower; the ratios between classes are what the roof uses), and tera lane-ops/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 2048

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// 32-bit chains: a[8]; 64-bit chains: d[8] (double) / l[8] (int64)
#define KERNEL32(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                                        \
    {                                                                                                                \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u; \
        uint32_t b = seed | 1u, c = seed * 7u + 3u;                                                                    \
        for (int it = 0; it < ITER; it++) {                                                                          \
            REP64(ASM)                                                                                               \
        }                                                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                          \
    }

#define A_ADD(i)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_SHR(i)   asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a##i));
#define A_XOR(i)   asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_ADD3(i)  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a##i) : "v"(b));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_MUL24(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_MAD24(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_PKSUB(i) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a##i) : "v"(b));
#define A_PKSHR(i) asm volatile("v_pk_lshrrev_b16 %0, %1, %0" : "+v"(a##i) : "v"(b));
#define A_DOT2U(i) asm volatile("v_dot2_u32_u16 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_DOT2I(i) asm volatile("v_dot2_i32_i16 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_DOT4I(i) asm volatile("v_dot4_i32_i8 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_PERM(i)  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_ALIGNB(i) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a##i) : "v"(b));
#define A_DPPMOV(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
#define A_DPPADD(i) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(b));
#define A_WSHR(i)  asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
#define A_MAXU(i)  asm volatile("v_max_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_SUBSAT(i) asm volatile("v_sub_u32 %0, %0, %1 clamp" : "+v"(a##i) : "v"(b));
#define A_CLZ(i)   asm volatile("v_ffbh_u32 %0, %0" : "+v"(a##i));
#define A_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a##i) : "v"(b));
#define A_SWZ(i)   asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n\ts_waitcnt lgkmcnt(0)" : "+v"(a##i));

#define A_MOV(i)   asm volatile("v_mov_b32 %0, %1" : "=v"(a##i) : "v"(b));
#define A_AND(i)   asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_OR(i)    asm volatile("v_or_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_SHL(i)   asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a##i));
#define A_ASHR(i)  asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a##i));
#define A_SUB(i)   asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_MINU(i)  asm volatile("v_min_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a##i) : "v"(b));
#define A_OR3(i)   asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define A_BFE(i)   asm volatile("v_bfe_u32 %0, %0, 1, 30" : "+v"(a##i));
#define A_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(a##i) : "v"(b));
#define A_MULU24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a##i) : "v"(b) : "vcc");
#define A_CMP(i)   asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(a##i), "v"(b) : "vcc");
#define A_CNDS(i)  asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a##i) : "v"(b));
#define A_CMPCND(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
#define A_CVTU(i)  asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##i));
KERNEL32(k_mov, A_MOV) KERNEL32(k_and, A_AND) KERNEL32(k_or, A_OR) KERNEL32(k_shl, A_SHL) KERNEL32(k_ashr, A_ASHR) KERNEL32(k_sub, A_SUB)
KERNEL32(k_minu, A_MINU) KERNEL32(k_andor, A_ANDOR) KERNEL32(k_lshlor, A_LSHLOR) KERNEL32(k_or3, A_OR3) KERNEL32(k_bfe, A_BFE)
KERNEL32(k_alignbit, A_ALIGNBIT) KERNEL32(k_mulu24, A_MULU24) KERNEL32(k_mulhi, A_MULHI) KERNEL32(k_addco, A_ADDCO) KERNEL32(k_cmp, A_CMP)
KERNEL32(k_cnds, A_CNDS) KERNEL32(k_cmpcnd, A_CMPCND) KERNEL32(k_cvtu, A_CVTU)
KERNEL32(k_add, A_ADD) KERNEL32(k_shr, A_SHR) KERNEL32(k_xor, A_XOR) KERNEL32(k_add3, A_ADD3) KERNEL32(k_lshladd, A_LSHLADD)
KERNEL32(k_cndmask, A_CNDMASK) KERNEL32(k_mullo, A_MULLO) KERNEL32(k_mul24, A_MUL24) KERNEL32(k_mad24, A_MAD24)
KERNEL32(k_pksub, A_PKSUB) KERNEL32(k_pkshr, A_PKSHR) KERNEL32(k_dot2u, A_DOT2U) KERNEL32(k_dot2i, A_DOT2I) KERNEL32(k_dot4i, A_DOT4I)
KERNEL32(k_perm, A_PERM) KERNEL32(k_alignb, A_ALIGNB) KERNEL32(k_dppmov, A_DPPMOV) KERNEL32(k_dppadd, A_DPPADD) KERNEL32(k_wshr, A_WSHR)
KERNEL32(k_maxu, A_MAXU) KERNEL32(k_subsat, A_SUBSAT) KERNEL32(k_clz, A_CLZ) KERNEL32(k_bperm, A_BPERM) KERNEL32(k_swz, A_SWZ)

#define KERNEL64(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                                        \
    {                                                                                                                \
        double a0 = 1.0 + 1e-9 * (seed + threadIdx.x), a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3, a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7; \
        double b = 1.0 + 1e-12 * seed, c = 1e-13 * seed;                                                              \
        for (int it = 0; it < ITER; it++) {                                                                          \
            REP64(ASM)                                                                                               \
        }                                                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);             \
    }
#define D_ADD(i)  asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define D_MUL(i)  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define D_FMA(i)  asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL64(k_addf64, D_ADD) KERNEL64(k_mulf64, D_MUL) KERNEL64(k_fmaf64, D_FMA)

// conversions and the 64-bit integer multiply-add: own kernels (mixed operand widths)
__global__ __launch_bounds__(256) void k_cvt_f64_i32(uint32_t *out, uint32_t seed)
{
    int32_t x = (int32_t)(seed + threadIdx.x);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
#define C_CVT(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a##i) : "v"(x));
    for (int it = 0; it < ITER; it++) { REP64(C_CVT) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ __launch_bounds__(256) void k_mad_i64_i32(uint32_t *out, uint32_t seed)
{
    int32_t x = (int32_t)(seed + threadIdx.x), y = (int32_t)(seed * 3u + 1u);
    long long a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
#define C_MAD64(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a##i) : "v"(x), "v"(y) : "vcc");
    for (int it = 0; it < ITER; it++) { REP64(C_MAD64) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ __launch_bounds__(256) void k_mfma_i8(uint32_t *out, uint32_t seed)
{
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i a = { (int)seed, 1, 2, 3 }, b = { 4, 5, 6, (int)threadIdx.x };
    v4i c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
#define C_MFMA(i) c##i = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c##i, 0, 0, 0);
    for (int it = 0; it < ITER; it++) { REP64(C_MFMA) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(c0[0] ^ c1[1] ^ c2[2] ^ c3[3] ^ c4[0] ^ c5[1] ^ c6[2] ^ c7[3]);
}
// LDS: 16-byte loads and stores, conflict free (lane stride 16 bytes)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_ds_read_b128(uint32_t *out, uint32_t seed)
{
    __shared__ v4u buf[1024];
    buf[threadIdx.x] = (v4u){ seed, 1, 2, 3 };
    __syncthreads();
    v4u a0, a1, a2, a3, a4, a5, a6, a7;
    const v4u *p = buf + (threadIdx.x & 63);
#define C_LDR(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a##i) : "v"((uint32_t)(uintptr_t)p), "i"(1024 * i));
    for (int it = 0; it < ITER; it++) { REP64(C_LDR) asm volatile("s_waitcnt lgkmcnt(0)"); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] ^ a1[1] ^ a2[2] ^ a3[3] ^ a4[0] ^ a5[1] ^ a6[2] ^ a7[3];
}
__global__ __launch_bounds__(256) void k_ds_write_b128(uint32_t *out, uint32_t seed)
{
    __shared__ v4u buf[1024];
    v4u v = { seed, threadIdx.x, 2, 3 };
    v4u *p = buf + (threadIdx.x & 63);
#define C_LDW(i) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"((uint32_t)(uintptr_t)p), "v"(v), "i"(1024 * i) : "memory");
    for (int it = 0; it < ITER; it++) { REP64(C_LDW) asm volatile("s_waitcnt lgkmcnt(0)"); }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = buf[threadIdx.x][0];
}

// Do a wavefront's LDS stores run beside ANOTHER wavefront's fp64 arithmetic on the same SIMD?  Workgroups alternate (by blockIdx / 8, so
// that neighbours on an XCD differ): mode 0 = every workgroup fp64 only, 1 = every workgroup 16-byte LDS stores only, 2 = they alternate
// (with two and four workgroups per CU every SIMD then hosts both kinds), 3 = every wavefront does both, interleaved 8 : 1 as a butterfly
// stage does.  If mode 2 takes the longer of modes 0 and 1 at half the workgroups each, the two overlap; if it takes their sum, they do not.
template <int MODE>
__global__ __launch_bounds__(256) void k_mix(uint32_t *out, uint32_t seed)
{
    __shared__ v4u buf[1024];
    double a0 = 1.0 + 1e-9 * (seed + threadIdx.x), a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3, a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
    double b = 1.0 + 1e-12 * seed;
    v4u v = { seed, threadIdx.x, 2, 3 };
    v4u *p = buf + (threadIdx.x & 63);
    const bool valu = MODE == 0 || MODE == 3 || (MODE == 2 && ((blockIdx.x >> 3) & 1) == 0);
    const bool lds = MODE == 1 || MODE == 3 || (MODE == 2 && ((blockIdx.x >> 3) & 1) == 1);
    for (int it = 0; it < ITER; it++) {
        if (valu) { REP64(D_MUL) }
        if (lds) { REP8(C_LDW) asm volatile("s_waitcnt lgkmcnt(0)"); }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + buf[threadIdx.x][0];
}

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Test { const char *name; kern_t k; double lanes_per_inst; };

int main()
{
    uint32_t *out;
    hipMalloc(&out, 256u * 8u * 256u * 4u);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<Test> tests = {
        { "v_add_u32", k_add, 64 }, { "v_lshrrev_b32", k_shr, 64 }, { "v_xor_b32", k_xor, 64 }, { "v_add3_u32", k_add3, 64 }, { "v_lshl_add_u32", k_lshladd, 64 },
        { "v_cndmask_b32 (vcc)", k_cndmask, 64 }, { "v_cndmask_b32 (sgpr pair)", k_cnds, 64 }, { "v_cmp_gt_u32", k_cmp, 64 }, { "v_cmp_gt_u32 + v_cndmask_b32 (pair)", k_cmpcnd, 128 },
        { "v_mov_b32", k_mov, 64 }, { "v_and_b32", k_and, 64 }, { "v_or_b32", k_or, 64 }, { "v_lshlrev_b32", k_shl, 64 }, { "v_ashrrev_i32", k_ashr, 64 }, { "v_sub_u32", k_sub, 64 },
        { "v_min_u32", k_minu, 64 }, { "v_and_or_b32", k_andor, 64 }, { "v_lshl_or_b32", k_lshlor, 64 }, { "v_or3_b32", k_or3, 64 }, { "v_bfe_u32", k_bfe, 64 },
        { "v_alignbit_b32", k_alignbit, 64 }, { "v_mul_u32_u24", k_mulu24, 64 }, { "v_mul_hi_u32", k_mulhi, 64 }, { "v_add_co_u32", k_addco, 64 }, { "v_cvt_f32_u32", k_cvtu, 64 }, { "v_max_u32", k_maxu, 64 }, { "v_sub_u32 clamp", k_subsat, 64 }, { "v_ffbh_u32", k_clz, 64 },
        { "v_mul_lo_u32", k_mullo, 64 }, { "v_mul_i32_i24", k_mul24, 64 }, { "v_mad_i32_i24", k_mad24, 64 }, { "v_mad_i64_i32", k_mad_i64_i32, 64 },
        { "v_pk_sub_u16 clamp", k_pksub, 64 }, { "v_pk_lshrrev_b16", k_pkshr, 64 }, { "v_dot2_u32_u16", k_dot2u, 64 }, { "v_dot2_i32_i16", k_dot2i, 64 },
        { "v_dot4_i32_i8", k_dot4i, 64 }, { "v_perm_b32", k_perm, 64 }, { "v_alignbyte_b32", k_alignb, 64 },
        { "v_mov_b32 dpp row_shr", k_dppmov, 64 }, { "v_add_u32 dpp row_shr", k_dppadd, 64 }, { "v_mov_b32 dpp wave_shr", k_wshr, 64 },
        { "ds_bpermute_b32 (+wait)", k_bperm, 64 }, { "ds_swizzle_b32 (+wait)", k_swz, 64 },
        { "v_add_f64", k_addf64, 64 }, { "v_mul_f64", k_mulf64, 64 }, { "v_fma_f64", k_fmaf64, 64 }, { "v_cvt_f64_i32", k_cvt_f64_i32, 64 },
        { "v_mfma_i32_16x16x64_i8", k_mfma_i8, 64 }, { "ds_read_b128 (8 per wait)", k_ds_read_b128, 64 }, { "ds_write_b128 (8 per wait)", k_ds_write_b128, 64 },
    };
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("# %s, %d CUs, clockRate %d kHz; ITER %d x 64 instructions per wavefront\n", pr.name, pr.multiProcessorCount, pr.clockRate, ITER);
    printf("# %-28s %6s %12s %14s %16s\n", "instruction", "w/SIMD", "time ms", "cyc/inst/SIMD", "T lane-ops/s");
    for (const Test &t : tests) {
        for (int wps : { 1, 2, 4 }) {
            // 256-thread workgroups = 4 wavefronts = one per SIMD; wps workgroups per CU
            dim3 grid(pr.multiProcessorCount * wps), block(256);
            hipLaunchKernelGGL(t.k, grid, block, 0, 0, out, 12345u);     // warm
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(t.k, grid, block, 0, 0, out, 12345u + rep);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double insts_per_simd = (double)ITER * 64.0 * wps * (t.lanes_per_inst / 64.0);
            const double cyc = best * 1e-3 * 2.4e9 / insts_per_simd;
            const double tlops = insts_per_simd * pr.multiProcessorCount * 4.0 * 64.0 / (best * 1e-3) / 1e12;
            printf("%-30s %6d %12.4f %14.2f %16.2f\n", t.name, wps, best, cyc, tlops);
        }
    }
    printf("# fp64 arithmetic beside 16-byte LDS stores (per iteration: 64 v_mul_f64 and / or 8 ds_write_b128 per wavefront; time in ms)\n");
    printf("# %-44s %6s %12s\n", "mode", "WG/CU", "time ms");
    struct { const char *name; kern_t k; } mix[4] = { { "every workgroup: 64 v_mul_f64", k_mix<0> }, { "every workgroup: 8 ds_write_b128", k_mix<1> },
                                                      { "workgroups alternate between the two", k_mix<2> }, { "every wavefront: both, 64 : 8", k_mix<3> } };
    for (auto &m : mix) {
        for (int wps : { 2, 4 }) {
            dim3 grid(pr.multiProcessorCount * wps), block(256);
            hipLaunchKernelGGL(m.k, grid, block, 0, 0, out, 1u);
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(m.k, grid, block, 0, 0, out, 2u + rep);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%-46s %6d %12.4f\n", m.name, wps, best);
        }
    }
    return 0;
}
