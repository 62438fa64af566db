// K6 bcd_chain: one workgroup per row/column chain of a BCD phase.
//
// Restates bcd() (python bcd.py:101-257) as scheduled by ceoBCD (python bcd.py:261-284).  All chains of a
// phase touch disjoint image lines and read nothing outside their own line, so they run concurrently.
// Arithmetic is float64 with the reference's operation order (no contraction) so that labels come out
// identical: dp = mincost + ((lamda*lcost + s1) + s2), first-index tie breaks everywhere.
// The compat test of pakovanje (daisy i flann.py:256-309, tpsi > |dy-dy'|+|dx-dx'|) is evaluated on the fly.
//
// Thread layout (v1): 640 threads = 160 labels x 4 k-parts.  Thread (tl, part) scans a quarter of the
// previous pixel's labels; the 4 partial minima are merged with quad shuffles; the part-0 lane finishes the
// label.  dp and the previous pixel's flows live in LDS (double-buffered); back-pointers go to the
// workspace as uint8 and are walked by one thread at the end.
#include "dflow_common.h"

#define BCD_THREADS 640
#define BCD_PARTS 4

struct BcdArgs {
    int H, W, LP, tpsi, phase;
    double lamda;
    const uint32_t *proposals;
    const float *lcosts;
    const int32_t *nprop;
    int32_t *bestlabels;
    uint8_t *back;
};

__device__ static inline void chain_geom(int phase, int chain, int H, int W, int &ty, int &tx, int &ys, int &xs, int &len)
{
    // python bcd.py:265-277
    if (phase == 0) { ty = 0; tx = 2 * chain; ys = 1; xs = 0; len = H; }
    else if (phase == 1) { ty = 2 * chain; tx = W - 1; ys = 0; xs = -1; len = W; }
    else if (phase == 2) { ty = H - 1; tx = (W / 2) * 2 - 1 - 2 * chain; ys = -1; xs = 0; len = H; }
    else { ty = (H / 2) * 2 - 1 - 2 * chain; tx = 0; ys = 0; xs = 1; len = W; }
}

// lexicographic (value, index) minimum across the wave; every lane ends with the result
__device__ static inline void wave_argmin(double &v, int &k)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double ov = __shfl_xor(v, off);
        int ok = __shfl_xor(k, off);
        if (ov < v || (ov == v && ok < k)) { v = ov; k = ok; }
    }
}

__global__ void __launch_bounds__(BCD_THREADS) bcd_chain_kernel(BcdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *dpbuf = (double *)smem;                               // [2][DFLOW_MAX_LABELS]
    uint32_t *fpbuf = (uint32_t *)(dpbuf + 2 * DFLOW_MAX_LABELS); // [2][DFLOW_MAX_LABELS] biased flows
    uint32_t *bestf = fpbuf + 2 * DFLOW_MAX_LABELS;               // [len] biased flow of each chain pixel's current label

    const int tid = threadIdx.x, lane = tid & 63;
    const int tl = tid >> 2, part = tid & 3;
    const int chain = blockIdx.x;
    const int W = a.W, LP = a.LP;
    int ty0, tx0, ys, xs, len;
    chain_geom(a.phase, chain, a.H, W, ty0, tx0, ys, xs, len);
    const int pstep = ys * W + xs;            // pixel-index step along the chain
    const int pix0 = ty0 * W + tx0;
    // sidepsi neighbours (python bcd.py:107-112,119,161): (+side) then (-side) in IMAGE coordinates, which lie on
    // the chain itself.  Chain index of the +side neighbour is i+dirp, of the -side neighbour i-dirp.
    const int dirp = (ys + xs);               // +1 if the chain runs towards larger coordinates, else -1
    const uint32_t tpsi = (uint32_t)a.tpsi;
    const double tpsi_d = (double)a.tpsi;

    for (int i = tid; i < len; i += BCD_THREADS) {
        int pix = pix0 + i * pstep;
        bestf[i] = flow_bias(a.proposals[(size_t)pix * LP + a.bestlabels[pix]]);
    }
    __syncthreads();

    uint8_t *back = a.back + (size_t)chain * len * LP;

    // ---- chain start: dp[0,tl] = (s1 + s2) + lamda*lcost   (python bcd.py:118-120)
    int tn = a.nprop[pix0];
    uint32_t Fc = 0; float lc = 0.0f;
    if (tl < tn) { Fc = flow_bias(a.proposals[(size_t)pix0 * LP + tl]); lc = a.lcosts[(size_t)pix0 * LP + tl]; }
    {
        int ip = dirp, im = -dirp;
        uint32_t s1 = (ip >= 0 && ip < len) ? min(tpsi, flow_l1_biased(Fc, bestf[ip])) : 0u;
        uint32_t s2 = (im >= 0 && im < len) ? min(tpsi, flow_l1_biased(Fc, bestf[im])) : 0u;
        if (part == 0 && tl < tn) {
            dpbuf[tl] = __dadd_rn((double)(s1 + s2), __dmul_rn(a.lamda, (double)lc));
            fpbuf[tl] = Fc;
        }
    }
    __syncthreads();

    int cur = 1;
    int pn = tn;
    // prefetch pixel 1
    int tn_n = 0; uint32_t Fc_n = 0; float lc_n = 0.0f;
    if (len > 1) {
        int pix = pix0 + pstep;
        tn_n = a.nprop[pix];
        if (tl < tn_n) { Fc_n = flow_bias(a.proposals[(size_t)pix * LP + tl]); lc_n = a.lcosts[(size_t)pix * LP + tl]; }
    }
    for (int i = 1; i < len; i++) {
        tn = tn_n; Fc = Fc_n; lc = lc_n;
        if (i + 1 < len) {   // prefetch the next pixel's labels while this one is processed
            int pix = pix0 + (i + 1) * pstep;
            tn_n = a.nprop[pix];
            Fc_n = 0; lc_n = 0.0f;
            if (tl < tn_n) { Fc_n = flow_bias(a.proposals[(size_t)pix * LP + tl]); lc_n = a.lcosts[(size_t)pix * LP + tl]; }
        }
        const double *dp = dpbuf + (cur ^ 1) * DFLOW_MAX_LABELS;
        const uint32_t *fp = fpbuf + (cur ^ 1) * DFLOW_MAX_LABELS;

        // permmincost / permminlabel (python bcd.py:152-157), computed redundantly by every wave
        double permcost = 800000.0; int permlabel = 0x7fffffff;
        for (int k = lane; k < pn; k += 64) {
            double c = __dadd_rn(tpsi_d, dp[k]);
            if (c < permcost) { permcost = c; permlabel = k; }
        }
        wave_argmin(permcost, permlabel);

        // min over compatible previous labels (python bcd.py:163-176 / :198-219)
        bool found = false; double best = 0.0; int bk = 0;
        if (tl < tn) {
            int kper = (pn + BCD_PARTS - 1) / BCD_PARTS;
            int kb = part * kper, ke = min(pn, kb + kper);
            for (int k = kb; k < ke; k++) {
                uint32_t psi = flow_l1_biased(Fc, fp[k]);
                if (psi < tpsi) {
                    double c = __dadd_rn(dp[k], (double)psi);
                    if (!found || c < best) { best = c; bk = k; found = true; }
                }
            }
        }
#pragma unroll
        for (int off = 1; off <= 2; off <<= 1) {
            double ob = __shfl_xor(best, off);
            int ok = __shfl_xor(bk, off);
            int of = __shfl_xor((int)found, off);
            if (of && (!found || ob < best || (ob == best && ok < bk))) { best = ob; bk = ok; found = true; }
        }
        if (part == 0 && tl < tn) {
            int ip = i + dirp, im = i - dirp;
            uint32_t s1 = (ip >= 0 && ip < len) ? min(tpsi, flow_l1_biased(Fc, bestf[ip])) : 0u;
            uint32_t s2 = (im >= 0 && im < len) ? min(tpsi, flow_l1_biased(Fc, bestf[im])) : 0u;
            double mincost = found ? best : permcost;
            int pl = found ? bk : permlabel;
            double small = __dadd_rn(__dadd_rn(__dmul_rn(a.lamda, (double)lc), (double)s1), (double)s2);
            dpbuf[cur * DFLOW_MAX_LABELS + tl] = __dadd_rn(mincost, small);
            fpbuf[cur * DFLOW_MAX_LABELS + tl] = Fc;
            back[(size_t)i * LP + tl] = (uint8_t)pl;
        }
        __syncthreads();
        pn = tn;
        cur ^= 1;
    }

    // ---- end label: first minimum of dp[len-1] (python bcd.py:231-237), then traceback (:239-253)
    if (tid < 64) {
        const double *dp = dpbuf + (cur ^ 1) * DFLOW_MAX_LABELS;
        double v = 800000.0; int vk = 0x7fffffff;
        for (int k = lane; k < pn; k += 64) { double c = dp[k]; if (c < v) { v = c; vk = k; } }
        wave_argmin(v, vk);
        if (tid == 0) {
            int pl = vk == 0x7fffffff ? 0 : vk;
            a.bestlabels[pix0 + (len - 1) * pstep] = pl;
            for (int i = len - 1; i >= 1; i--) {
                pl = back[(size_t)i * LP + pl];
                a.bestlabels[pix0 + (i - 1) * pstep] = pl;
            }
        }
    }
}

static void phase_dims(const dflow_params *p, int phase, int &nchains, int &len)
{
    int H = p->pich, W = p->picw;
    if (phase == 0) { nchains = (W + 1) / 2; len = H; }
    else if (phase == 1) { nchains = (H + 1) / 2; len = W; }
    else if (phase == 2) { nchains = W / 2; len = H; }
    else { nchains = H / 2; len = W; }
}

size_t bcd_ws_bytes(const dflow_params *p)
{
    size_t m = 0;
    for (int ph = 0; ph < 4; ph++) {
        int n, len;
        phase_dims(p, ph, n, len);
        size_t b = (size_t)n * len * p->label_pitch;
        if (b > m) m = b;
    }
    return m;
}

int launch_bcd_phase(const dflow_params *p, const uint32_t *proposals, const float *lcosts, const int32_t *nprop,
                     int32_t *bestlabels, int phase, void *ws, hipStream_t s)
{
    int nchains, len;
    phase_dims(p, phase, nchains, len);
    if (nchains == 0) return DFLOW_OK;
    BcdArgs a;
    a.H = p->pich; a.W = p->picw; a.LP = p->label_pitch; a.tpsi = p->tpsi; a.phase = phase; a.lamda = p->lamda;
    a.proposals = proposals; a.lcosts = lcosts; a.nprop = nprop; a.bestlabels = bestlabels; a.back = (uint8_t *)ws;
    size_t shmem = 2 * DFLOW_MAX_LABELS * (sizeof(double) + sizeof(uint32_t)) + (size_t)len * sizeof(uint32_t);
    hipLaunchKernelGGL(bcd_chain_kernel, dim3(nchains), dim3(BCD_THREADS), shmem, s, a);
    return dflow_check_launch("bcd_chain_kernel");
}
