// K5 neighbour_proposals: nasumicni, daisy i flann.py:205-233.  One thread per pixel.
//
// The reference draws int(np.random.normal(y, 8)), int(np.random.normal(x, 8)) from an unseeded global
// Mersenne Twister.  The build uses a counter-based stream instead (Philox4x32-10, key = seed, counter =
// (pixel, attempt)), so every pixel is independent and CPU oracle and GPU agree draw for draw: word 0 -> y
// offset, word 1 -> x offset, a 32-bit uniform mapped to floor(sigma*z) through a 127-entry threshold table
// (integer compares only).  Everything else is the reference's arithmetic, quirks included: truncation
// toward zero (Q7), component-wise 'in' with Python slice semantics (Q5), |signed sum| cost against the
// sampled neighbour's position in image 2 (Q6).
#include <math.h>
#include "dflow_common.h"

struct NbrArgs {
    Geom g;
    int LP, L, K, ngauss, max_attempts;
    float tphi;
    uint32_t k0, k1;
    const void *d1, *d2;       // float32 (H,W,68) or binary16 (H,W,72) rows
    uint32_t *proposals;
    float *lcosts;
    int32_t *nprop;
    const int32_t *bestlabels;
    const uint32_t *bestflow;  // [pix] the WTA proposal of every pixel (proposals[pix][bestlabels[pix]]), gathered once
    uint32_t thr[128];   // thr[i] = floor(Phi((i-63)/sigma) * 2^32), i < 127
};

__device__ static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t &o0, uint32_t &o1)
{
    uint32_t c2 = 0, c3 = 0;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1;
}

__device__ static inline int gauss_offset(const uint32_t *thr, uint32_t u)
{
    int lo = 0, hi = 127;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (u >= thr[mid]) lo = mid + 1; else hi = mid; }
    return -64 + lo;
}

__device__ static inline int trunc_draw(int c, int off) { int t = c + off; return t < 0 ? t + 1 : t; }

__device__ static inline void py_slice(int start, int stop, int n, int &lo, int &hi)
{
    if (start < 0) { start += n; if (start < 0) start = 0; } else if (start > n) start = n;
    if (stop < 0) { stop += n; if (stop < 0) stop = 0; } else if (stop > n) stop = n;
    lo = start; hi = stop > start ? stop : start;
}

// tv "in" proposals[lo:hi] (either component equal, Q5).  The slice holds at most K = 5 entries (one cell's 5-pack): all of them
// are fetched at once -- a loop with an early exit makes every load wait for the comparison before it, up to five dependent
// round trips per draw, which was most of this kernel's time -- and compared afterwards; slots beyond the slice read the
// row's first entry (always inside the row) and are ignored.
#define NBR_MAXK 5
__device__ static inline bool tv_in(const uint32_t *prow, int lo, int hi, uint32_t tv)
{
    if (hi - lo > NBR_MAXK) {               // other K: the plain loop
        for (int i = lo; i < hi; i++) {
            uint32_t v = prow[i];
            if ((v & 0xFFFFu) == (tv & 0xFFFFu) || (v >> 16) == (tv >> 16)) return true;
        }
        return false;
    }
    uint32_t v[NBR_MAXK];
#pragma unroll
    for (int j = 0; j < NBR_MAXK; j++) v[j] = prow[lo + j < hi ? lo + j : 0];
    bool in = false;
#pragma unroll
    for (int j = 0; j < NBR_MAXK; j++)
        in |= lo + j < hi && ((v[j] & 0xFFFFu) == (tv & 0xFFFFu) || (v[j] >> 16) == (tv >> 16));
    return in;
}

#define NBR_THREADS 128

// the flow every draw copies (daisy i flann.py:225): one 4-byte array that stays in L2 instead of two dependent loads per
// draw (0.90 -> 0.73 ms; fetching the next draw's inputs ahead of time on top of that was measured: no further gain, the
// 272-byte row gather of an accepted draw sets the pace)
__global__ void nbr_bestflow_kernel(const uint32_t *__restrict__ proposals, const int32_t *__restrict__ bestlabels,
                                    uint32_t *__restrict__ bestflow, int n, int LP)
{
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix < n) bestflow[pix] = proposals[(size_t)pix * LP + bestlabels[pix]];
}

template <typename T> __global__ void __launch_bounds__(NBR_THREADS) neighbour_kernel(NbrArgs a)
{
    __shared__ uint32_t thr[128];
    extern __shared__ uint32_t s_app[];                      // [ngauss][NBR_THREADS]: the proposals this thread has appended so far
    if (threadIdx.x < 128) thr[threadIdx.x] = a.thr[threadIdx.x];
    __syncthreads();
    const Geom g = a.g;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= g.H * g.W) return;
    const int y = pix / g.W, x = pix % g.W;
    const int mincellyl = max(0, g.celly(y) - g.win);                       // :212
    const int ncellyl = min(g.ncy, g.celly(y) + g.win) - mincellyl;         // :214
    const int mincellxl = max(0, g.cellx(x) - g.win);                       // :215
    uint32_t *prow = a.proposals + (size_t)pix * a.LP;
    // this pixel's descriptor stays in registers for all of its draws
    float q[DFLOW_DESC];
    desc_load_row(q, reinterpret_cast<const T *>(a.d1), (size_t)pix);
    int np_ = a.nprop[pix], ngp = 0, i = 0;
    for (uint32_t att = 0; i < a.ngauss && att < (uint32_t)a.max_attempts; att++) {
        uint32_t r0, r1;
        philox4x32_10((uint32_t)pix, att, a.k0, a.k1, r0, r1);
        const int tgy = trunc_draw(y, gauss_offset(thr, r0));               // :219
        if (tgy < 0 || tgy >= g.H) continue;
        const int tgx = trunc_draw(x, gauss_offset(thr, r1));               // :221
        if (tgx < 0 || tgx >= g.W) continue;
        const int broj = a.K * ((g.celly(tgy) - mincellyl) + (g.cellx(tgx) - mincellxl) * ncellyl);   // :223-224
        const int tpix = tgy * g.W + tgx;
        const uint32_t tv = a.bestflow[tpix];                                                           // :225

        int lo, hi, lo2, hi2;
        py_slice(broj, broj + a.K, a.L, lo, hi);
        py_slice(np_ - ngp, np_, a.L, lo2, hi2);
        // the second slice is exactly what this thread appended (np_ <= L and ngp <= np_ always): kept in LDS
        bool dup = tv_in(prow, lo, hi, tv);                                 // :226
        // eight entries of the LDS slice per round, fetched together (same reason as in tv_in)
        const int j1 = hi2 - (np_ - ngp);
        for (int j0 = lo2 - (np_ - ngp); j0 < j1; j0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = s_app[(j0 + u < j1 ? j0 + u : j0) * NBR_THREADS + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 8; u++) dup |= j0 + u < j1 && ((v[u] & 0xFFFFu) == (tv & 0xFFFFu) || (v[u] >> 16) == (tv >> 16));
        }
        if (!dup) {
            prow[np_] = tv;                                                 // :227
            s_app[ngp * NBR_THREADS + threadIdx.x] = tv;
            float diff[DFLOW_DESC];
            desc_load_row(diff, reinterpret_cast<const T *>(a.d2), (size_t)tpix);
#pragma unroll
            for (int k = 0; k < DFLOW_DESC; k++) diff[k] = q[k] - diff[k];
            const float s = fabsf(np_pairwise_sum68(diff));                 // :228-229 (Q6)
            a.lcosts[(size_t)pix * a.LP + np_] = s < a.tphi ? s : a.tphi;
            np_++; ngp++;                                                   // :230-231
        }
        i++;                                                                // :233
    }
    a.nprop[pix] = np_;
}

static void gauss_thresholds(double sigma, uint32_t *thr)
{
    for (int i = 0; i < 127; i++) {
        double phi = 0.5 * erfc(-((i - 63) / sigma) / sqrt(2.0));
        double v = floor(phi * 4294967296.0);
        thr[i] = v >= 4294967295.0 ? 4294967295u : (uint32_t)v;
    }
    thr[127] = 4294967295u;
}

size_t neighbour_ws_bytes(const dflow_params *p) { return (size_t)p->pich * p->picw * sizeof(uint32_t) + 256; }

int launch_neighbour(const dflow_params *p, const void *d1, const void *d2, uint32_t *proposals, float *lcosts,
                     int32_t *nprop, const int32_t *bestlabels, void *ws, hipStream_t s)
{
    NbrArgs a;
    a.g = make_geom(p);
    a.LP = p->label_pitch; a.L = p->maxnprop; a.K = p->knn; a.ngauss = p->ngauss; a.max_attempts = p->max_attempts;
    a.tphi = p->tphi; a.k0 = (uint32_t)p->seed; a.k1 = (uint32_t)(p->seed >> 32);
    a.d1 = d1; a.d2 = d2; a.proposals = proposals; a.lcosts = lcosts; a.nprop = nprop; a.bestlabels = bestlabels;
    gauss_thresholds((double)p->sigma, a.thr);
    int n = p->pich * p->picw;
    uint32_t *bestflow = (uint32_t *)ws;
    a.bestflow = bestflow;
    hipLaunchKernelGGL(nbr_bestflow_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const uint32_t *)proposals, bestlabels, bestflow, n, p->label_pitch);
    const size_t shmem = (size_t)(p->ngauss > 0 ? p->ngauss : 1) * NBR_THREADS * sizeof(uint32_t);
    if (descr_f16(p)) hipLaunchKernelGGL(neighbour_kernel<_Float16>, dim3((n + NBR_THREADS - 1) / NBR_THREADS), dim3(NBR_THREADS), shmem, s, a);
    else hipLaunchKernelGGL(neighbour_kernel<float>, dim3((n + NBR_THREADS - 1) / NBR_THREADS), dim3(NBR_THREADS), shmem, s, a);
    return dflow_check_launch("neighbour_kernel");
}
