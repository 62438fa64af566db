// K6 BCD: bcd() (python bcd.py:101-257) as scheduled by ceoBCD (python bcd.py:261-284), plus pakovanje's compat
// bit matrices (daisy i flann.py:256-309) in the orientation the chains need them.
//
// Two kernels:
//   bcd_masks_kernel   once per pass (the proposals do not change during the sweeps): for every pixel p and both of
//                      its chains (column chain / row chain) the 160-bit rows mask[tl] = { k : tpsi > |dy-dy'|+|dx-dx'|
//                      between label tl of p and label k of p's predecessor on that chain } -- the reference's
//                      packedksets (Q8), restricted to the two neighbours that are ever used and already transposed
//                      for the direction in which the chain runs.  lanes = predecessor labels, one v_sad_u16 +
//                      v_cmp per 64 pairs, the wave ballot is the mask word.
//   bcd_chain_kernel   one workgroup per chain of a phase (all chains of a phase are independent: a chain reads and
//                      writes only its own image line).  640 threads = 160 labels x 4 lanes; a lane walks the set
//                      bits of its quarter of the label's mask row (only compatible predecessors cost float64 work),
//                      quad DPP merge, float64 arithmetic in the reference's association order
//                      (dp = mincost + ((lamda*lcost + s1) + s2); first-index ties everywhere, Q9-Q11).
//                      dp / predecessor flows live in LDS (double buffered), label data and mask rows are prefetched
//                      two steps ahead, back-pointers go to the workspace as uint8 and are walked chunk-wise from LDS.
#include "dflow_common.h"

#define BCD_THREADS 640
#define BCD_MASK_WORDS 5                 // 160 bits per label row
#define BCD_TB_STEPS 128                 // traceback chunk (steps) staged in LDS

__device__ static inline void chain_geom(int phase, int chain, int H, int W, int &ty, int &tx, int &ys, int &xs, int &len)
{
    // python bcd.py:265-277
    if (phase == 0) { ty = 0; tx = 2 * chain; ys = 1; xs = 0; len = H; }
    else if (phase == 1) { ty = 2 * chain; tx = W - 1; ys = 0; xs = -1; len = W; }
    else if (phase == 2) { ty = H - 1; tx = (W / 2) * 2 - 1 - 2 * chain; ys = -1; xs = 0; len = H; }
    else { ty = (H / 2) * 2 - 1 - 2 * chain; tx = 0; ys = 0; xs = 1; len = W; }
}

// ------------------------------------------------------------------------------------------------ masks
// grid: one wave per (pixel, dir); dir 0 = the pixel's column chain, dir 1 = its row chain.
__global__ void __launch_bounds__(256) bcd_masks_kernel(int H, int W, int LP, int tpsi, const uint32_t *__restrict__ proposals,
                                                        const int32_t *__restrict__ nprop, uint32_t *__restrict__ masks)
{
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= 2LL * H * W) return;
    const int dir = (int)(item & 1);
    const int pix = (int)(item >> 1);
    const int y = pix / W, x = pix % W;
    // predecessor on the chain (python bcd.py:265-277): even columns run down, odd columns up, even rows leftwards, odd rows rightwards
    int py = y, px = x;
    if (dir == 0) py = (x & 1) ? y + 1 : y - 1; else px = (y & 1) ? x - 1 : x + 1;
    if (py < 0 || py >= H || px < 0 || px >= W) return;       // chain start: no transition into this pixel
    const int ppix = py * W + px;
    const int tn = nprop[pix], pn = nprop[ppix];
    uint32_t fp[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int k = lane + 64 * j;
        fp[j] = k < pn ? flow_bias(proposals[(size_t)ppix * LP + k]) : 0u;   // 0: far from every biased flow
    }
    const uint32_t *cur = proposals + (size_t)pix * LP;
    uint32_t *out = masks + ((size_t)pix * 2 + dir) * (size_t)LP * BCD_MASK_WORDS;
    for (int tl = 0; tl < tn; tl++) {
        const uint32_t fc = flow_bias(cur[tl]);                 // wave-uniform
        const unsigned long long m0 = __ballot(flow_l1_biased(fc, fp[0]) < (uint32_t)tpsi);
        const unsigned long long m1 = __ballot(flow_l1_biased(fc, fp[1]) < (uint32_t)tpsi);
        const unsigned long long m2 = __ballot(flow_l1_biased(fc, fp[2]) < (uint32_t)tpsi);
        uint32_t w = (uint32_t)m0;
        w = lane == 1 ? (uint32_t)(m0 >> 32) : w;
        w = lane == 2 ? (uint32_t)m1 : w;
        w = lane == 3 ? (uint32_t)(m1 >> 32) : w;
        w = lane == 4 ? (uint32_t)m2 : w;
        if (lane < BCD_MASK_WORDS) out[(size_t)tl * BCD_MASK_WORDS + lane] = w;
    }
}

// ------------------------------------------------------------------------------------------------ chains
struct BcdArgs {
    int H, W, LP, tpsi, phase;
    double lamda;
    const uint32_t *proposals;
    const float *lcosts;
    const int32_t *nprop;
    int32_t *bestlabels;
    const uint32_t *masks;
    uint8_t *back;
};

struct Cand {
    double v; int k;
};

// (value, index) lexicographic minimum
__device__ static inline void cand_min(Cand &a, double ov, int ok)
{
    if (ov < a.v || (ov == a.v && ok < a.k)) { a.v = ov; a.k = ok; }
}

template <int CTRL> __device__ static inline void cand_dpp(const Cand &a, double &ov, int &ok)
{
    // lanes without a source keep their own value (bound_ctrl = false, old = self)
    const int lo = __double2loint(a.v), hi = __double2hiint(a.v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    ok = __builtin_amdgcn_update_dpp(a.k, a.k, CTRL, 0xF, 0xF, false);
    ov = __hiloint2double(ohi, olo);
}

#define DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
#define DPP_ROW_SHL4 0x104
#define DPP_ROW_SHL8 0x108


// minimum over the 16 label owners of a wave (lanes 0,4,..,60), valid in lane 0
__device__ static inline void wave_owner_min(Cand &pm)
{
    double ov; int ok;
    cand_dpp<DPP_ROW_SHL4>(pm, ov, ok); cand_min(pm, ov, ok);
    cand_dpp<DPP_ROW_SHL8>(pm, ov, ok); cand_min(pm, ov, ok);
    const int lo = __double2loint(pm.v), hi = __double2hiint(pm.v);
#pragma unroll
    for (int r = 1; r < 4; r++) {    // rows are in label order: merging row leaders 16, 32, 48 into lane 0 keeps first-index ties
        const double rv = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r), __builtin_amdgcn_readlane(lo, 16 * r));
        const int rk = __builtin_amdgcn_readlane(pm.k, 16 * r);
        cand_min(pm, rv, rk);
    }
}

__global__ void __launch_bounds__(BCD_THREADS) bcd_chain_kernel(BcdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *dpbuf = (double *)smem;                                   // [2][DFLOW_MAX_LABELS]
    double *permv = dpbuf + 2 * DFLOW_MAX_LABELS;                     // [2][16] per-wave minima of tpsi + dp
    uint32_t *fpbuf = (uint32_t *)(permv + 2 * 16);                   // [2][DFLOW_MAX_LABELS] biased flows
    int *permi = (int *)(fpbuf + 2 * DFLOW_MAX_LABELS);               // [2][16]; permi[32] = traceback hand-over
    uint8_t *tb = (uint8_t *)(permi + 2 * 16 + 4);                    // [BCD_TB_STEPS][LP] traceback chunk (16-byte aligned)
    uint32_t *bestf = (uint32_t *)(tb + BCD_TB_STEPS * a.LP);         // [len] biased flow of each chain pixel's current label
    int *s_label = permi + 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tl = tid >> 2, part = tid & 3;
    const int nwaves = BCD_THREADS / 64;
    const int chain = blockIdx.x;
    const int W = a.W, LP = a.LP;
    int ty0, tx0, ys, xs, len;
    chain_geom(a.phase, chain, a.H, W, ty0, tx0, ys, xs, len);
    const int pstep = ys * W + xs;            // pixel-index step along the chain
    const int pix0 = ty0 * W + tx0;
    const int dir = ys != 0 ? 0 : 1;          // column chain / row chain
    // sidepsi neighbours (python bcd.py:107-112,119,161): (+side) then (-side) in IMAGE coordinates, which lie on the
    // chain itself.  Chain index of the +side neighbour is i+dirp, of the -side neighbour i-dirp.
    const int dirp = ys + xs;                 // +1 if the chain runs towards larger coordinates, else -1
    const uint32_t tpsi = (uint32_t)a.tpsi;
    const double tpsi_d = (double)a.tpsi;

    for (int i = tid; i < len; i += BCD_THREADS) {
        int pix = pix0 + i * pstep;
        bestf[i] = flow_bias(a.proposals[(size_t)pix * LP + a.bestlabels[pix]]);
    }
    __syncthreads();

    uint8_t *back = a.back + (size_t)chain * len * LP;

    // per-step inputs of this thread, prefetched three steps ahead (A = next step, B, C)
    struct StepIn { int tn; uint32_t F; float lc; uint32_t mw, m4; };
    auto fetch = [&](int i) {
        StepIn r; r.tn = 0; r.F = 0; r.lc = 0.0f; r.mw = 0; r.m4 = 0;
        if (i < len) {
            const int pix = pix0 + i * pstep;
            r.tn = a.nprop[pix];
            if (tl < r.tn) {
                r.F = flow_bias(a.proposals[(size_t)pix * LP + tl]);
                r.lc = a.lcosts[(size_t)pix * LP + tl];
                if (i > 0) {
                    const uint32_t *row = a.masks + (((size_t)pix * 2 + dir) * LP + tl) * BCD_MASK_WORDS;
                    r.mw = row[part];
                    r.m4 = (row[4] >> (8 * part)) & 0xFFu;
                }
            }
        }
        return r;
    };
    const StepIn S0 = fetch(0);
    StepIn A = fetch(1), B = fetch(2), C = fetch(3);

    // ---- chain start: dp[0,tl] = (s1 + s2) + lamda*lcost   (python bcd.py:118-120)
    {
        const int tn = S0.tn;
        const uint32_t Fc = S0.F;
        const int ip = dirp, im = -dirp;
        const uint32_t s1 = (ip >= 0 && ip < len) ? min(tpsi, flow_l1_biased(Fc, bestf[ip])) : 0u;
        const uint32_t s2 = (im >= 0 && im < len) ? min(tpsi, flow_l1_biased(Fc, bestf[im])) : 0u;
        Cand pm; pm.v = 1e300; pm.k = 0x7fffffff;
        if (part == 0 && tl < tn) {
            const double d0 = __dadd_rn((double)(s1 + s2), __dmul_rn(a.lamda, (double)S0.lc));
            dpbuf[tl] = d0;
            fpbuf[tl] = Fc;
            pm.v = __dadd_rn(tpsi_d, d0); pm.k = tl;
        }
        wave_owner_min(pm);
        if (lane == 0) { permv[wave] = pm.v; permi[wave] = pm.k; }
    }
    __syncthreads();

    int cur = 1;
    int pn = S0.tn;
    for (int i = 1; i < len; i++) {
        const int tn = A.tn;
        const uint32_t Fc = A.F;
        const float lc = A.lc;
        uint32_t mw = A.mw, m4 = A.m4;
        A = B; B = C; C = fetch(i + 3);
        const double *dp = dpbuf + (cur ^ 1) * DFLOW_MAX_LABELS;
        const uint32_t *fp = fpbuf + (cur ^ 1) * DFLOW_MAX_LABELS;

        // permmincost / permminlabel (python bcd.py:152-157): first minimum of tpsi + dp[k] over the previous labels,
        // merged from the per-row minima the previous step left in LDS (rows are in label order)
        Cand perm; perm.v = 800000.0; perm.k = 0x7fffffff;
        {
            const double *pv = permv + (cur ^ 1) * 16;
            const int *pi = permi + (cur ^ 1) * 16;
            for (int w = 0; w < nwaves; w++) {
                if (w * 16 >= pn) break;
                cand_min(perm, pv[w], pi[w]);
            }
        }

        // min over compatible previous labels (python bcd.py:163-176 / :198-219): walk the set bits of my quarter
        Cand best; best.v = 1e300; best.k = 0x7fffffff;
        {
            const int kb = 32 * part;
            while (mw) {
                const int b = __ffs(mw) - 1; mw &= mw - 1;
                const int k = kb + b;
                const uint32_t psi = flow_l1_biased(Fc, fp[k]);
                cand_min(best, __dadd_rn(dp[k], (double)psi), k);
            }
            const int kb4 = 128 + 8 * part;
            while (m4) {
                const int b = __ffs(m4) - 1; m4 &= m4 - 1;
                const int k = kb4 + b;
                const uint32_t psi = flow_l1_biased(Fc, fp[k]);
                cand_min(best, __dadd_rn(dp[k], (double)psi), k);
            }
        }
        {
            double ov; int ok;
            cand_dpp<DPP_QUAD_XOR1>(best, ov, ok); cand_min(best, ov, ok);
            cand_dpp<DPP_QUAD_XOR2>(best, ov, ok); cand_min(best, ov, ok);
        }
        Cand pm; pm.v = 1e300; pm.k = 0x7fffffff;
        if (part == 0 && tl < tn) {
            const int ip = i + dirp, im = i - dirp;
            const uint32_t s1 = (ip >= 0 && ip < len) ? min(tpsi, flow_l1_biased(Fc, bestf[ip])) : 0u;
            const uint32_t s2 = (im >= 0 && im < len) ? min(tpsi, flow_l1_biased(Fc, bestf[im])) : 0u;
            const bool found = best.k != 0x7fffffff;
            const double mincost = found ? best.v : perm.v;
            const int pl = found ? best.k : perm.k;
            const double small = __dadd_rn(__dadd_rn(__dmul_rn(a.lamda, (double)lc), (double)s1), (double)s2);
            const double dpc = __dadd_rn(mincost, small);
            dpbuf[cur * DFLOW_MAX_LABELS + tl] = dpc;
            fpbuf[cur * DFLOW_MAX_LABELS + tl] = Fc;
            back[(size_t)i * LP + tl] = (uint8_t)pl;
            pm.v = __dadd_rn(tpsi_d, dpc); pm.k = tl;
        }
        wave_owner_min(pm);
        if (lane == 0) { permv[cur * 16 + wave] = pm.v; permi[cur * 16 + wave] = pm.k; }
        __syncthreads();
        pn = tn;
        cur ^= 1;
    }

    // ---- end label: first minimum of dp[len-1] (python bcd.py:231-237): tpsi + dp is monotone in dp, but two different
    // dp may round to the same sum, so the minimum is taken over dp itself
    if (tid < 64) {
        const double *dp = dpbuf + (cur ^ 1) * DFLOW_MAX_LABELS;
        Cand m; m.v = 800000.0; m.k = 0x7fffffff;
        for (int k = lane; k < pn; k += 64) { double c = dp[k]; if (c < m.v) { m.v = c; m.k = k; } }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            double ov = __shfl_xor(m.v, off);
            int ok = __shfl_xor(m.k, off);
            cand_min(m, ov, ok);
        }
        if (tid == 0) *s_label = m.k == 0x7fffffff ? 0 : m.k;
    }
    __syncthreads();
    // ---- traceback (python bcd.py:239-253): chunks of back-pointer rows are staged in LDS, one thread walks them
    int pl = *s_label;
    if (tid == 0) a.bestlabels[pix0 + (len - 1) * pstep] = pl;
    for (int hi = len - 1; hi >= 1; hi -= BCD_TB_STEPS) {
        const int lo = max(1, hi - BCD_TB_STEPS + 1);      // steps lo..hi
        const int nbytes = (hi - lo + 1) * LP;
        const uint4 *src = reinterpret_cast<const uint4 *>(back + (size_t)lo * LP);
        for (int j = tid; j < nbytes / 16; j += BCD_THREADS) reinterpret_cast<uint4 *>(tb)[j] = src[j];
        __syncthreads();
        if (tid == 0) {
            for (int i = hi; i >= lo; i--) {
                pl = tb[(i - lo) * LP + pl];
                a.bestlabels[pix0 + (i - 1) * pstep] = pl;
            }
            *s_label = pl;
        }
        __syncthreads();
        pl = *s_label;
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

static size_t back_bytes(const dflow_params *p)
{
    size_t m = 0;
    for (int ph = 0; ph < 4; ph++) {
        int n, len;
        phase_dims(p, ph, n, len);
        size_t b = (size_t)n * len * p->label_pitch;
        if (b > m) m = b;
    }
    return (m + 255) & ~(size_t)255;
}

static size_t mask_bytes(const dflow_params *p)
{
    return (size_t)p->pich * p->picw * 2 * p->label_pitch * BCD_MASK_WORDS * sizeof(uint32_t);
}

size_t bcd_ws_bytes(const dflow_params *p) { return back_bytes(p) + mask_bytes(p); }

int launch_bcd_prepare(const dflow_params *p, const uint32_t *proposals, const int32_t *nprop, void *ws, hipStream_t s)
{
    uint32_t *masks = (uint32_t *)((char *)ws + back_bytes(p));
    long long items = 2LL * p->pich * p->picw;
    hipLaunchKernelGGL(bcd_masks_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p->pich, p->picw, p->label_pitch,
                       p->tpsi, proposals, nprop, masks);
    return dflow_check_launch("bcd_masks_kernel");
}

int launch_bcd_phase(const dflow_params *p, const uint32_t *proposals, const float *lcosts, const int32_t *nprop,
                     int32_t *bestlabels, int phase, void *ws, hipStream_t s)
{
    int nchains, len;
    phase_dims(p, phase, nchains, len);
    if (nchains == 0) return DFLOW_OK;
    BcdArgs a;
    a.H = p->pich; a.W = p->picw; a.LP = p->label_pitch; a.tpsi = p->tpsi; a.phase = phase; a.lamda = p->lamda;
    a.proposals = proposals; a.lcosts = lcosts; a.nprop = nprop; a.bestlabels = bestlabels;
    a.back = (uint8_t *)ws; a.masks = (const uint32_t *)((char *)ws + back_bytes(p));
    size_t shmem = 2 * DFLOW_MAX_LABELS * (sizeof(double) + sizeof(uint32_t)) + 2 * 16 * (sizeof(double) + sizeof(int)) + 16 +
                   (size_t)BCD_TB_STEPS * p->label_pitch + (size_t)len * sizeof(uint32_t);
    hipLaunchKernelGGL(bcd_chain_kernel, dim3(nchains), dim3(BCD_THREADS), shmem, s, a);
    return dflow_check_launch("bcd_chain_kernel");
}
