// Principal axes of the image-2 descriptors for the kNN screen (knn_mfma.hip).
//
// Squared L2 distances do not change under an orthogonal change of basis, and DAISY descriptors are strongly
// correlated (overlapping Gaussian histograms): in the principal basis of a frame's descriptors the trailing 26 of
// the 68 directions carry < 0.1 % of the energy (scratch/pca_stats.py (round 3, git history)).  The screen runs its matrix products over the
// leading KM_KD directions only and bounds the rest by Cauchy-Schwarz (one extra K slot, |q_D| |c_D|), which cuts
// the MFMAs per tile from 5 to 3.  ANY orthonormal basis gives exact results; a good one keeps the bound tight.
//
//   knn_cov_kernel     partial second-moment matrices (about the origin, see knn_mfma.hip) of PCA_SAMPLES evenly spaced pixels, one block per
//                      PCA_BLOCK_SAMPLES samples, float64, fixed summation order (deterministic basis)
//   knn_jacobi_kernel  one workgroup: sum of the partials, cyclic Jacobi eigenvalue iteration in float32, Newton-Schulz
//                      polish of V in float64, columns sorted by decreasing eigenvalue, rounded to float32 and written as
//                      [component][dimension].  |V^T V - I|_F of the written matrix is measured (float64) and a value above
//                      PCA_DELTA_MAX (NaN input) sets the flag that sends the whole pass to the exact search.
#include "dflow_common.h"
#include "knn_pca.h"

#define PCA_N DFLOW_DESC
#define PCA_LD (PCA_N + 1)                 // LDS leading dimension (odd: conflict-free columns)
#define PCA_LDD (PCA_N + 2)                // of the float64 matrices: rows 16-byte aligned
#define PCA_BLOCK_SAMPLES 256
#define PCA_BLOCKS (PCA_SAMPLES / PCA_BLOCK_SAMPLES)
#ifndef PCA_SWEEPS
#define PCA_SWEEPS 5
#endif

template <typename T>
__global__ void __launch_bounds__(256) knn_cov_kernel(const T *__restrict__ d, double *__restrict__ partial, int npix)
{
    __shared__ float xs[PCA_BLOCK_SAMPLES][PCA_LD];
    const int nsamp = npix < PCA_SAMPLES ? npix : PCA_SAMPLES, stride = npix / nsamp;
    for (int e = threadIdx.x; e < PCA_BLOCK_SAMPLES * PCA_N; e += 256) {
        const int sl = e / PCA_N, k = e % PCA_N, sidx = blockIdx.x * PCA_BLOCK_SAMPLES + sl;
        const float v = sidx < nsamp ? (float)d[(size_t)sidx * stride * DescPitch<T>::value + k] : 0.0f;
        xs[sl][k] = fabsf(v) < 1e4f ? v : 0.0f;               // NaN / inf / absurd values do not steer the basis (any basis is valid)
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PCA_N * PCA_N; e += 256) {
        const int i = e / PCA_N, j = e % PCA_N;
        // float32 inside a block (256 terms: 1e-5 relative, the matrix only steers the choice of the basis), float64 across blocks
        float acc = 0.0f;
#pragma unroll 16
        for (int sl = 0; sl < PCA_BLOCK_SAMPLES; sl++) acc = fmaf(xs[sl][i], xs[sl][j], acc);
        partial[(size_t)blockIdx.x * PCA_N * PCA_N + e] = (double)acc;
    }
}

// One workgroup.  Phase 1, float32: cyclic Jacobi with the round-robin parallel ordering (34 disjoint rotations per
// round, 67 rounds per sweep): 34 lanes compute the rotations of a round (fast reciprocal / square root: the angle only
// steers the iteration), everybody applies them -- A <- J^T A J one 2x2 block (pair k1 <= pair k2) per work item and its
// mirror image, V <- V J one row of a pair per work item, both in place -- two barriers per round.
// Phase 2, float64: the rotations were only approximately orthogonal (|V^T V - I| ~ 1e-5 after 2 000 of them), so V is
// polished by one Newton-Schulz step V <- V (3 I - V^T V) / 2, which converges quadratically to the nearest orthonormal
// matrix (1e-4 -> 1e-8).
__global__ void __launch_bounds__(1024) knn_jacobi_kernel(const double *__restrict__ partial, float *__restrict__ vt,
                                                          int *__restrict__ flags)
{
    __shared__ float A[PCA_N][PCA_LD], Vf[PCA_N][PCA_LD];
    __shared__ __attribute__((aligned(16))) double Vd[PCA_N][PCA_LDD], G[PCA_N][PCA_LDD];
    __shared__ float4 rot[PCA_N / 2];            // (p, q as bits, c, s) of every pair of the round
    __shared__ int rank[PCA_N];
    __shared__ double red[16];
    const int tid = threadIdx.x;
    for (int e = tid; e < PCA_N * PCA_N; e += 1024) {
        const int i = e / PCA_N, j = e % PCA_N;
        double acc = 0.0;
        for (int b = 0; b < PCA_BLOCKS; b++) acc += partial[(size_t)b * PCA_N * PCA_N + e];
        A[i][j] = (float)acc;
        Vf[i][j] = i == j ? 1.0f : 0.0f;
    }
    __syncthreads();
    // the work items of a round: upper triangle of the 34 x 34 grid of 2x2 blocks, enumerated once
    constexpr int NP = PCA_N / 2, NBLK = NP * (NP + 1) / 2;
    int bk1[(NBLK + 1023) / 1024], bk2[(NBLK + 1023) / 1024];
#pragma unroll
    for (int u = 0; u < (NBLK + 1023) / 1024; u++) {
        int e = tid + u * 1024, k1 = 0;
        if (e >= NBLK) e = NBLK - 1;
        while (e >= NP - k1) { e -= NP - k1; k1++; }          // row k1 of the triangle holds NP - k1 blocks
        bk1[u] = k1; bk2[u] = k1 + e;
    }
    constexpr int NVI = (NP * PCA_N + 1023) / 1024;
    int vk[NVI], vi[NVI];
#pragma unroll
    for (int u = 0; u < NVI; u++) { const int e = tid + u * 1024; vk[u] = (e / PCA_N) % NP; vi[u] = e % PCA_N; }
    for (int sweep = 0; sweep < PCA_SWEEPS; sweep++) {
        for (int r = 0; r < PCA_N - 1; r++) {
            if (tid < NP) {
                // pair tid of round r: (N-1, r) and ((r + k) mod (N-1), (r - k) mod (N-1)), k = 1 .. N/2-1
                const int a = tid == 0 ? PCA_N - 1 : (r + tid) % (PCA_N - 1), b = tid == 0 ? r : (r - tid + (PCA_N - 1)) % (PCA_N - 1);
                const int p = a < b ? a : b, q = a < b ? b : a;
                const float app = A[p][p], aqq = A[q][q], apq = A[p][q];
                float c = 1.0f, s = 0.0f;
                if (fabsf(apq) > 1e-30f && fabsf(apq) > 1e-9f * (fabsf(app) + fabsf(aqq))) {
                    // Numerical Recipes 11.1: theta = (a_qq - a_pp) / 2 a_pq, t = sgn(theta) / (|theta| + sqrt(theta^2 + 1))
                    const float theta = (aqq - app) * 0.5f * __builtin_amdgcn_rcpf(apq);
                    float t = __builtin_amdgcn_rcpf(fabsf(theta) + __builtin_amdgcn_sqrtf(theta * theta + 1.0f));   // theta = inf: t = 0
                    t = theta >= 0.0f ? t : -t;
                    c = __builtin_amdgcn_rsqf(t * t + 1.0f);
                    s = t * c;
                    if (!(fabsf(s) <= 1.0f)) { c = 1.0f; s = 0.0f; }          // NaN guard
                }
                rot[tid] = make_float4(__int_as_float(p), __int_as_float(q), c, s);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < (NBLK + 1023) / 1024; u++) {
                if (tid + u * 1024 < NBLK) {
                    const int k1 = bk1[u], k2 = bk2[u];
                    const float4 ra = rot[k1], rb = rot[k2];
                    const int p1 = __float_as_int(ra.x), q1 = __float_as_int(ra.y), p2 = __float_as_int(rb.x), q2 = __float_as_int(rb.y);
                    const float c1 = ra.z, s1 = ra.w, c2 = rb.z, s2 = rb.w;
                    const float b00 = A[p1][p2], b01 = A[p1][q2], b10 = A[q1][p2], b11 = A[q1][q2];
                    // B' = R1^T B R2: rows (c1 b0. - s1 b1.), (s1 b0. + c1 b1.), then columns (c2 r.0 - s2 r.1), (s2 r.0 + c2 r.1)
                    const float r00 = c1 * b00 - s1 * b10, r01 = c1 * b01 - s1 * b11, r10 = s1 * b00 + c1 * b10, r11 = s1 * b01 + c1 * b11;
                    const float n00 = c2 * r00 - s2 * r01, n01 = s2 * r00 + c2 * r01, n10 = c2 * r10 - s2 * r11, n11 = s2 * r10 + c2 * r11;
                    A[p1][p2] = n00; A[p1][q2] = n01; A[q1][p2] = n10; A[q1][q2] = n11;
                    if (k1 != k2) { A[p2][p1] = n00; A[q2][p1] = n01; A[p2][q1] = n10; A[q2][q1] = n11; }
                }
            }
#pragma unroll
            for (int u = 0; u < NVI; u++) {
                if (tid + u * 1024 < NP * PCA_N) {
                    const float4 ra = rot[vk[u]];
                    const int i = vi[u], p = __float_as_int(ra.x), q = __float_as_int(ra.y);
                    const float c = ra.z, s = ra.w, vp = Vf[i][p], vq = Vf[i][q];
                    Vf[i][p] = c * vp - s * vq;
                    Vf[i][q] = s * vp + c * vq;
                }
            }
            __syncthreads();
        }
    }
    // rank of every column by decreasing eigenvalue (ties and NaN by index: any order is valid)
    if (tid < PCA_N) {
        const float lj = A[tid][tid];
        int rk = 0;
        for (int i = 0; i < PCA_N; i++) {
            const float li = A[i][i];
            rk += (li > lj || (!(li < lj) && i < tid)) ? 1 : 0;
        }
        rank[tid] = rk;
    }
    for (int e = tid; e < PCA_N * PCA_N; e += 1024) Vd[e / PCA_N][e % PCA_N] = (double)Vf[e / PCA_N][e % PCA_N];
    __syncthreads();
    // Newton-Schulz: V <- V (1.5 I - 0.5 V^T V), once (1e-4 -> 1e-8, far below the float32 rounding that follows).  Both
    // products are computed in 1 x 4 tiles: per k one value of the left factor and four of the right (two 16-byte reads).
    constexpr int NT = PCA_N * (PCA_N / 4);
    for (int t = tid; t < NT; t += 1024) {
        const int a = t / (PCA_N / 4), b4 = (t % (PCA_N / 4)) * 4;
        double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
#pragma unroll 4
        for (int i = 0; i < PCA_N; i++) {          // (fully unrolled the compiler hoists all 204 loads and spills)
            const double va = Vd[i][a];
            const double2 x01 = *reinterpret_cast<const double2 *>(&Vd[i][b4]), x23 = *reinterpret_cast<const double2 *>(&Vd[i][b4 + 2]);
            g0 = fma(va, x01.x, g0); g1 = fma(va, x01.y, g1); g2 = fma(va, x23.x, g2); g3 = fma(va, x23.y, g3);
        }
        G[a][b4] = (a == b4 ? 1.5 : 0.0) - 0.5 * g0; G[a][b4 + 1] = (a == b4 + 1 ? 1.5 : 0.0) - 0.5 * g1;
        G[a][b4 + 2] = (a == b4 + 2 ? 1.5 : 0.0) - 0.5 * g2; G[a][b4 + 3] = (a == b4 + 3 ? 1.5 : 0.0) - 0.5 * g3;
    }
    __syncthreads();
    double nv[(NT + 1023) / 1024][4];
#pragma unroll
    for (int u = 0; u < (NT + 1023) / 1024; u++) {
        const int t = tid + u * 1024;
        double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
        if (t < NT) {
            const int i = t / (PCA_N / 4), j4 = (t % (PCA_N / 4)) * 4;
#pragma unroll 4
            for (int k = 0; k < PCA_N; k++) {
                const double v = Vd[i][k];
                const double2 x01 = *reinterpret_cast<const double2 *>(&G[k][j4]), x23 = *reinterpret_cast<const double2 *>(&G[k][j4 + 2]);
                g0 = fma(v, x01.x, g0); g1 = fma(v, x01.y, g1); g2 = fma(v, x23.x, g2); g3 = fma(v, x23.y, g3);
            }
        }
        nv[u][0] = g0; nv[u][1] = g1; nv[u][2] = g2; nv[u][3] = g3;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (NT + 1023) / 1024; u++) {
        const int t = tid + u * 1024;
        if (t < NT) {
            const int i = t / (PCA_N / 4), j4 = (t % (PCA_N / 4)) * 4;
            Vd[i][j4] = nv[u][0]; Vd[i][j4 + 1] = nv[u][1]; Vd[i][j4 + 2] = nv[u][2]; Vd[i][j4 + 3] = nv[u][3];
        }
    }
    __syncthreads();
    // the basis the screen uses is V rounded to float32 (2^-24 relative per entry: |V^T V - I|_F ~ 5e-7); that matrix is
    // what is written and what is measured
    for (int e = tid; e < PCA_N * PCA_N; e += 1024) {
        const int j = e / PCA_N, i = e % PCA_N;
        const float vf = (float)Vd[i][j];
        vt[(size_t)rank[j] * PCA_N + i] = vf;
        Vd[i][j] = (double)vf;
    }
    __syncthreads();
    // |V^T V - I|_F
    double part = 0.0;
    for (int t = tid; t < NT; t += 1024) {
        const int a = t / (PCA_N / 4), b4 = (t % (PCA_N / 4)) * 4;
        double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
#pragma unroll 4
        for (int i = 0; i < PCA_N; i++) {          // (fully unrolled the compiler hoists all 204 loads and spills)
            const double va = Vd[i][a];
            const double2 x01 = *reinterpret_cast<const double2 *>(&Vd[i][b4]), x23 = *reinterpret_cast<const double2 *>(&Vd[i][b4 + 2]);
            g0 = fma(va, x01.x, g0); g1 = fma(va, x01.y, g1); g2 = fma(va, x23.x, g2); g3 = fma(va, x23.y, g3);
        }
        g0 -= a == b4 ? 1.0 : 0.0; g1 -= a == b4 + 1 ? 1.0 : 0.0; g2 -= a == b4 + 2 ? 1.0 : 0.0; g3 -= a == b4 + 3 ? 1.0 : 0.0;
        part += g0 * g0 + g1 * g1 + g2 * g2 + g3 * g3;
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; w++) tot += red[w];
        if (!(tot <= PCA_DELTA_MAX * PCA_DELTA_MAX)) atomicOr(flags, 1);          // also catches NaN
    }
}

size_t knn_pca_ws_bytes(void) { return (size_t)PCA_BLOCKS * PCA_N * PCA_N * sizeof(double); }

int launch_knn_pca(const void *d2, bool f16, float *vt, int *flags, void *ws, int npix, hipStream_t s)
{
    double *partial = (double *)ws;
    if (f16) hipLaunchKernelGGL(knn_cov_kernel<_Float16>, dim3(PCA_BLOCKS), dim3(256), 0, s, (const _Float16 *)d2, partial, npix);
    else hipLaunchKernelGGL(knn_cov_kernel<float>, dim3(PCA_BLOCKS), dim3(256), 0, s, (const float *)d2, partial, npix);
    hipLaunchKernelGGL(knn_jacobi_kernel, dim3(1), dim3(1024), 0, s, (const double *)partial, vt, flags);
    return dflow_check_launch("knn_jacobi_kernel");
}
