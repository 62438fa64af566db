// How far is v_mfma_f32_32x32x16_f16 (chain of 3 or 5, f32 accumulator) from the exact dot product of its f16 inputs?
// Prints the largest |D - exact| / sum_k |a_k b_k| over many random tiles of several operand distributions.
// hipcc --offload-arch=gfx950 -O2 scratch/ubench/mfma_err.hip -o scratch/ubench/mfma_err && scratch/ubench/mfma_err
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define KSTEPS 5
#define K (16 * KSTEPS)
__global__ void k(const _Float16 *A, const _Float16 *B, float *D, int ntiles)
{
    const int lane = threadIdx.x & 63;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const _Float16 *a = A + (size_t)t * 32 * K, *b = B + (size_t)t * 32 * K;   // A[row][k], B[col][k]
        f32x16 acc = {0};
        for (int s = 0; s < KSTEPS; s++) {
            half8 af = *(const half8 *)(a + (lane & 31) * K + 16 * s + 8 * (lane >> 5));
            half8 bf = *(const half8 *)(b + (lane & 31) * K + 16 * s + 8 * (lane >> 5));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            D[(size_t)t * 1024 + row * 32 + col] = acc[r];
        }
    }
}
static double rnd() { return rand() / (double)RAND_MAX; }
int main()
{
    const int nt = 4096;
    std::vector<_Float16> A((size_t)nt * 32 * K), B((size_t)nt * 32 * K);
    std::vector<float> D((size_t)nt * 1024);
    _Float16 *dA, *dB; float *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
    const char *names[] = {"uniform +-1", "positive only", "log-uniform magnitudes 2^-10..2^10, random signs", "one huge term + small ones",
                           "cancelling pairs + small", "descriptor-like: decaying components + h pieces"};
    for (int dist = 0; dist < 6; dist++) {
        for (int rep = 0; rep < 4; rep++) {
            for (size_t i = 0; i < A.size(); i++) {
                const int kk = i % K;
                double a, b;
                switch (dist) {
                case 0: a = 2 * rnd() - 1; b = 2 * rnd() - 1; break;
                case 1: a = rnd(); b = rnd(); break;
                case 2: a = (rnd() < 0.5 ? -1 : 1) * exp2(20 * rnd() - 10); b = (rnd() < 0.5 ? -1 : 1) * exp2(20 * rnd() - 10); break;
                case 3: a = kk == 7 ? 200.0 * (1 + rnd()) : rnd() - 0.5; b = kk == 7 ? 150.0 * (1 + rnd()) : rnd() - 0.5; break;
                case 4: a = (kk & 1) ? 100 * (1 + 0.0 * rnd()) : 100.0; b = (kk & 1) ? -(1 + 1e-3 * rnd()) : 1.0; if (kk >= K - 8) { a = rnd(); b = rnd(); } break;
                default: a = (2 * rnd() - 1) * 300 * exp(-0.12 * kk); b = (2 * rnd() - 1) * 300 * exp(-0.12 * kk);
                         if (kk == K - 5) { a = -1; b = 20000 * rnd(); } if (kk == K - 4) { a = -1; b = 8 * rnd(); } break;
                }
                A[i] = (_Float16)a; B[i] = (_Float16)b;
            }
            hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(512), dim3(64), 0, 0, dA, dB, dD, nt);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, worst_rel_res = 0;
            for (int t = 0; t < nt; t++)
                for (int r = 0; r < 32; r++)
                    for (int c = 0; c < 32; c++) {
                        double ex = 0, ab = 0;
                        for (int kk = 0; kk < K; kk++) {
                            const double p = (double)(float)A[(size_t)t * 32 * K + r * K + kk] * (double)(float)B[(size_t)t * 32 * K + c * K + kk];
                            ex += p; ab += fabs(p);
                        }
                        const double err = fabs((double)D[(size_t)t * 1024 + r * 32 + c] - ex);
                        if (ab > 0 && err / ab > worst) worst = err / ab;
                        if (fabs(ex) > 0 && err / fabs(ex) > worst_rel_res) worst_rel_res = err / fabs(ex);
                    }
            printf("K=%d %-50s rep %d: max |err| / sum|products| = %.3e = 2^%.2f\n", K, names[dist], rep, worst, log2(worst));
        }
    }
    return 0;
}
