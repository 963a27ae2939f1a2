// Which k does byte b of VGPR v of lane l carry in v_mfma_i32_16x16x64_i8?  (gpurun -- 'hipcc --offload-arch=gfx950 -o /tmp/p tools/probes/mfma_i8_layout.hip && /tmp/p')
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t *A /*16x64 row-major*/, const int8_t *B /*64x16 row-major*/, int *C /*16x16*/, int hyp)
{
    const int l = threadIdx.x, rc = l & 15, g = l >> 4;
    v4i a, b, c = {0, 0, 0, 0};
    for (int v = 0; v < 4; v++) {
        unsigned wa = 0, wb = 0;
        for (int by = 0; by < 4; by++) {
            const int kk = (hyp == 0) ? (16 * g + 4 * v + by) : (32 * (v >> 1) + 8 * g + 4 * (v & 1) + by);
            wa |= (unsigned)(uint8_t)A[rc * 64 + kk] << (8 * by);
            wb |= (unsigned)(uint8_t)B[kk * 16 + rc] << (8 * by);
        }
        a[v] = (int)wa; b[v] = (int)wb;
    }
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) C[(4 * g + i) * 16 + rc] = c[i];
}
int main()
{
    int8_t hA[16 * 64], hB[64 * 16]; int hC[256], ref[256];
    srand(1);
    for (auto &x : hA) x = (int8_t)(rand() % 256 - 128);
    for (auto &x : hB) x = (int8_t)(rand() % 256 - 128);
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int kk = 0; kk < 64; kk++) s += hA[i * 64 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    int8_t *dA, *dB; int *dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; hyp++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp);
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 256; i++) bad += hC[i] != ref[i];
        printf("hypothesis %d: %d of 256 wrong\n", hyp, bad);
    }
    return 0;
}
