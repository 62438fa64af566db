// micro-benchmark: dependent vs independent issue of the f64 / select instructions of the BCD list walk (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ void k(double *out, unsigned long long *cyc, double a, double b)
{
    double x0 = a + threadIdx.x, x1 = b, x2 = a * 2, x3 = b * 3, x4 = a * 5, x5 = b * 7, x6 = a * 9, x7 = b * 11, y = b + 1e-3;
    unsigned long long t0, t1;
    // 1: dependent v_min_f64 chain
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP64("v_min_f64 %0, %0, %1\n\t") : "+v"(x0) : "v"(y));
    asm volatile("s_nop 0" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[0] = t1 - t0;
    // 2: 8 independent v_min_f64 chains
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP8("v_min_f64 %0, %0, %8\n\tv_min_f64 %1, %1, %8\n\tv_min_f64 %2, %2, %8\n\tv_min_f64 %3, %3, %8\n\t"
                      "v_min_f64 %4, %4, %8\n\tv_min_f64 %5, %5, %8\n\tv_min_f64 %6, %6, %8\n\tv_min_f64 %7, %7, %8\n\t")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[1] = t1 - t0;
    // 3: dependent v_add_f64 chain
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP64("v_add_f64 %0, %0, %1\n\t") : "+v"(x1) : "v"(y));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[2] = t1 - t0;
    // 4: the walk's per-candidate pattern, serial: add -> cmp -> min -> cndmask (x2 = bestv, x3 = c)
    int bk = threadIdx.x, kk = 5;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP64("v_add_f64 %1, %3, %4\n\tv_cmp_lt_f64 vcc, %1, %0\n\tv_min_f64 %0, %0, %1\n\tv_cndmask_b32 %2, %2, %5, vcc\n\t")
                 : "+v"(x2), "+v"(x3), "+v"(bk) : "v"(x4), "v"(y), "v"(kk) : "vcc");
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[3] = t1 - t0;
    // 5: dependent v_add_f32 chain (reference)
    float f = (float)a;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP64("v_add_f32 %0, %0, %1\n\t") : "+v"(f) : "v"((float)b));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[4] = t1 - t0;
    // 6: v_cvt_f64_u32 + v_add_f64 dependent pair
    unsigned u = threadIdx.x & 7;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP64("v_cvt_f64_u32 %0, %1\n\tv_add_f64 %0, %0, %2\n\t") : "+v"(x5) : "v"(u), "v"(y));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // 7: s_nop 1 + v_min_u32_dpp pairs
    unsigned w = threadIdx.x * 2654435761u;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP64("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(w));
    t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) cyc[6] = t1 - t0;
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + bk + f + w;
}
int main()
{
    double *out; unsigned long long *cyc, h[8];
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, 2.5); hipDeviceSynchronize(); }
    hipMemcpy(h, cyc, 56, hipMemcpyDeviceToHost);
    const char *n[] = {"64 dependent v_min_f64", "64 v_min_f64 on 8 chains", "64 dependent v_add_f64", "64 x (add,cmp,min,cndmask) serial",
                       "64 dependent v_add_f32", "64 x (cvt_f64_u32, add_f64) dependent", "64 x (s_nop 1, v_min_u32_dpp)"};
    for (int i = 0; i < 7; i++) printf("%-40s %6llu cycles  = %.1f per instruction group\n", n[i], h[i], h[i] / 64.0);
    return 0;
}
