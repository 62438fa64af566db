// Principal axes of the image-2 descriptors for the kNN screen (knn_mfma.hip).
//
// Squared L2 distances do not change under an orthogonal change of basis, and DAISY descriptors are strongly
// correlated (overlapping Gaussian histograms): in the principal basis of a frame's descriptors the trailing 26 of
// the 68 directions carry < 0.1 % of the energy (scratch/pca_stats.py).  The screen runs its matrix products over the
// leading KM_KD directions only and bounds the rest by Cauchy-Schwarz (one extra K slot, |q_D| |c_D|), which cuts
// the MFMAs per tile from 5 to 3.  ANY orthonormal basis gives exact results; a good one keeps the bound tight.
//
//   knn_cov_kernel     partial scatter matrices of PCA_SAMPLES evenly spaced pixels about the centre mu, one block per
//                      PCA_BLOCK_SAMPLES samples, float64, fixed summation order (deterministic basis)
//   knn_jacobi_kernel  one workgroup: sum of the partials, cyclic Jacobi eigenvalue iteration in float64 with the
//                      round-robin parallel ordering (34 disjoint rotations per round, 67 rounds per sweep), columns
//                      sorted by decreasing eigenvalue; V is written as [component][dimension] float64.  The rotations keep V
//                      orthonormal to float64 rounding whatever the state of convergence; |V^T V - I|_F is measured and a
//                      value above 1e-9 (NaN input) sets the flag that sends the whole pass to the exact search.
#include "dflow_common.h"
#include "knn_pca.h"

#define PCA_N DFLOW_DESC
#define PCA_LD (PCA_N + 1)                 // LDS leading dimension (odd: conflict-free columns)
#define PCA_BLOCK_SAMPLES 64
#define PCA_BLOCKS (PCA_SAMPLES / PCA_BLOCK_SAMPLES)
#define PCA_SWEEPS 6

__global__ void __launch_bounds__(256) knn_cov_kernel(const float *__restrict__ d, const float *__restrict__ mu,
                                                      double *__restrict__ partial, int npix)
{
    __shared__ float xs[PCA_BLOCK_SAMPLES][PCA_LD];
    const int nsamp = npix < PCA_SAMPLES ? npix : PCA_SAMPLES, stride = npix / nsamp;
    for (int e = threadIdx.x; e < PCA_BLOCK_SAMPLES * PCA_N; e += 256) {
        const int sl = e / PCA_N, k = e % PCA_N, sidx = blockIdx.x * PCA_BLOCK_SAMPLES + sl;
        xs[sl][k] = sidx < nsamp ? d[(size_t)sidx * stride * PCA_N + k] - mu[k] : 0.0f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PCA_N * PCA_N; e += 256) {
        const int i = e / PCA_N, j = e % PCA_N;
        double acc = 0.0;
        for (int sl = 0; sl < PCA_BLOCK_SAMPLES; sl++) acc = fma((double)xs[sl][i], (double)xs[sl][j], acc);
        partial[(size_t)blockIdx.x * PCA_N * PCA_N + e] = acc;
    }
}

__global__ void __launch_bounds__(1024) knn_jacobi_kernel(const double *__restrict__ partial, double *__restrict__ vt,
                                                          int *__restrict__ flags)
{
    __shared__ double A[PCA_N][PCA_LD], V[PCA_N][PCA_LD];
    __shared__ double rc[PCA_N / 2], rs[PCA_N / 2];
    __shared__ int rp[PCA_N / 2], rq[PCA_N / 2], rank[PCA_N];
    __shared__ double red[16];
    const int tid = threadIdx.x;
    for (int e = tid; e < PCA_N * PCA_N; e += 1024) {
        const int i = e / PCA_N, j = e % PCA_N;
        double acc = 0.0;
        for (int b = 0; b < PCA_BLOCKS; b++) acc += partial[(size_t)b * PCA_N * PCA_N + e];
        A[i][j] = acc;
        V[i][j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int sweep = 0; sweep < PCA_SWEEPS; sweep++) {
        for (int r = 0; r < PCA_N - 1; r++) {
            // round-robin pairs of round r: (N-1, r) and ((r + k) mod (N-1), (r - k) mod (N-1)), k = 1 .. N/2-1
            if (tid < PCA_N / 2) {
                int p, q;
                if (tid == 0) { p = PCA_N - 1; q = r; }
                else { p = (r + tid) % (PCA_N - 1); q = (r - tid + (PCA_N - 1)) % (PCA_N - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                const double app = A[p][p], aqq = A[q][q], apq = A[p][q];
                double c = 1.0, s = 0.0;
                if (fabs(apq) > 1e-300 && fabs(apq) > 1e-17 * (fabs(app) + fabs(aqq))) {
                    const double theta = (aqq - app) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0);
                    s = t * c;
                }
                rp[tid] = p; rq[tid] = q; rc[tid] = c; rs[tid] = s;
            }
            __syncthreads();
            // A <- J^T A (rows p, q of every pair)
            for (int e = tid; e < (PCA_N / 2) * PCA_N; e += 1024) {
                const int k = e / PCA_N, j = e % PCA_N, p = rp[k], q = rq[k];
                const double c = rc[k], s = rs[k], ap = A[p][j], aq = A[q][j];
                A[p][j] = c * ap - s * aq;
                A[q][j] = s * ap + c * aq;
            }
            __syncthreads();
            // A <- A J, V <- V J (columns p, q of every pair)
            for (int e = tid; e < (PCA_N / 2) * PCA_N; e += 1024) {
                const int k = e / PCA_N, i = e % PCA_N, p = rp[k], q = rq[k];
                const double c = rc[k], s = rs[k];
                const double ap = A[i][p], aq = A[i][q];
                A[i][p] = c * ap - s * aq;
                A[i][q] = s * ap + c * aq;
                const double vp = V[i][p], vq = V[i][q];
                V[i][p] = c * vp - s * vq;
                V[i][q] = s * vp + c * vq;
            }
            __syncthreads();
        }
    }
    // rank of every column by decreasing eigenvalue (ties and NaN by index: any order is valid)
    if (tid < PCA_N) {
        const double lj = A[tid][tid];
        int rk = 0;
        for (int i = 0; i < PCA_N; i++) {
            const double li = A[i][i];
            rk += (li > lj || (!(li < lj) && i < tid)) ? 1 : 0;
        }
        rank[tid] = rk;
    }
    __syncthreads();
    for (int e = tid; e < PCA_N * PCA_N; e += 1024) {
        const int j = e / PCA_N, i = e % PCA_N;
        vt[(size_t)rank[j] * PCA_N + i] = V[i][j];
    }
    // |V^T V - I|_F
    double part = 0.0;
    for (int e = tid; e < PCA_N * PCA_N; e += 1024) {
        const int a = e / PCA_N, b = e % PCA_N;
        double g = a == b ? -1.0 : 0.0;
        for (int i = 0; i < PCA_N; i++) g = fma(V[i][a], V[i][b], g);
        part += g * g;
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; w++) tot += red[w];
        if (!(tot <= 1e-18)) atomicOr(flags, 1);          // also catches NaN
    }
}

size_t knn_pca_ws_bytes(void) { return (size_t)PCA_BLOCKS * PCA_N * PCA_N * sizeof(double); }

int launch_knn_pca(const float *d2, const float *mu, double *vt, int *flags, void *ws, int npix, hipStream_t s)
{
    double *partial = (double *)ws;
    hipLaunchKernelGGL(knn_cov_kernel, dim3(PCA_BLOCKS), dim3(256), 0, s, d2, mu, partial, npix);
    hipLaunchKernelGGL(knn_jacobi_kernel, dim3(1), dim3(1024), 0, s, (const double *)partial, vt, flags);
    return dflow_check_launch("knn_jacobi_kernel");
}
