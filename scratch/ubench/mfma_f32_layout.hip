// Operand / result layout of v_mfma_f32_32x32x2_f32 on gfx950: prints whether lane l supplies A[l & 31][l >> 5], B[l >> 5][l & 31] and
// receives D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31] in accumulator register r.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, float *D)   // A: 32 x 2, B: 2 x 32, D: 32 x 32 (row major)
{
    const int l = threadIdx.x;
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
int main()
{
    float hA[64], hB[64], hD[1024], *dA, *dB, *dD;
    for (int i = 0; i < 64; i++) { hA[i] = (float)(1 + i % 7) + 0.25f * (i / 7); hB[i] = (float)(3 - i % 5) + 0.5f * (i / 11); }
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float e = hA[i * 2] * hB[j] + hA[i * 2 + 1] * hB[32 + j]; if (e != hD[i * 32 + j]) bad++; }
    printf("v_mfma_f32_32x32x2_f32 layout as assumed: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return 0;
}
