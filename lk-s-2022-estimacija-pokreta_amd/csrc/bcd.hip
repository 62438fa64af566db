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
//                      writes only its own image line).  192 threads, one per label; a lane walks the set bits of
//                      its label's mask row (only compatible predecessors cost float64 work), float64 arithmetic in
//                      the reference's association order
//                      (dp = mincost + ((lamda*lcost + s1) + s2); first-index ties everywhere, Q9-Q11).
//                      dp / predecessor flows live in LDS (double buffered), label data and mask rows are prefetched
//                      three steps ahead, back-pointers go to the workspace as uint8 and are walked chunk-wise from LDS.
#include "dflow_common.h"

#define BCD_THREADS 192
#define BCD_MASK_WORDS 5                 // 160 bits per label row
#define BCD_REC_WORDS 6                  // row record the chain kernel reads every step: 4 words = the first 16 set bits as
                                         // bytes (0xFF = none), 1 word = popcount, 1 pad; the 5 mask words live in a second array
#define BCD_LIST 16
#define BCD_LDS_LABELS 256
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
                                                        const int32_t *__restrict__ nprop, uint32_t *__restrict__ masks,
                                                        uint32_t *__restrict__ recs)
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
    uint32_t fp[3], fcv[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int k = lane + 64 * j;
        fp[j] = k < pn ? flow_bias(proposals[(size_t)ppix * LP + k]) : 0u;   // 0: far from every biased flow
        fcv[j] = k < tn ? flow_bias(proposals[(size_t)pix * LP + k]) : 0u;
    }
    const size_t rowbase = ((size_t)pix * 2 + dir) * (size_t)LP;
#pragma unroll
    for (int grp = 0; grp < 3; grp++) {
        const int nrows = min(64, tn - 64 * grp);          // rows 64 grp .. 64 grp + nrows - 1 (wave-uniform)
        if (nrows <= 0) break;
        // lanes = predecessor labels: one v_sad_u16 + compare per 64 pairs, the ballot is a mask word; the 5
        // words of row r are deposited in lane r, so that afterwards lane = row
        uint32_t m[BCD_MASK_WORDS] = {0u, 0u, 0u, 0u, 0u};
        for (int r = 0; r < nrows; r++) {
            const uint32_t fc = __builtin_amdgcn_readlane(fcv[grp], r);      // label 64 grp + r of this pixel
            const unsigned long long m0 = __ballot(flow_l1_biased(fc, fp[0]) < (uint32_t)tpsi);
            const unsigned long long m1 = __ballot(flow_l1_biased(fc, fp[1]) < (uint32_t)tpsi);
            const unsigned long long m2 = __ballot(flow_l1_biased(fc, fp[2]) < (uint32_t)tpsi);
            const bool mine = lane == r;                       // deposit the 5 words of row r in lane r
            m[0] = mine ? (uint32_t)m0 : m[0];
            m[1] = mine ? (uint32_t)(m0 >> 32) : m[1];
            m[2] = mine ? (uint32_t)m1 : m[2];
            m[3] = mine ? (uint32_t)(m1 >> 32) : m[3];
            m[4] = mine ? (uint32_t)m2 : m[4];
        }
        if (lane < nrows) {
            // lane = row: the first 16 set bits as a byte list (what the chain kernel reads every step) and the popcount
            const int tl = 64 * grp + lane;
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < BCD_MASK_WORDS; j++) cnt += __popc(m[j]);
            uint32_t l[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            int n = 0;
#pragma unroll
            for (int j = 0; j < BCD_MASK_WORDS; j++) {
                uint32_t w = m[j];
                while (w && n < BCD_LIST) {
                    const uint32_t k = 32 * j + __ffs(w) - 1; w &= w - 1;
                    const uint32_t sh = 8 * (n & 3), clr = ~(0xFFu << sh), val = k << sh;
                    if (n < 4) l[0] = (l[0] & clr) | val; else if (n < 8) l[1] = (l[1] & clr) | val;
                    else if (n < 12) l[2] = (l[2] & clr) | val; else l[3] = (l[3] & clr) | val;
                    n++;
                }
            }
            uint32_t *rec = recs + (rowbase + tl) * BCD_REC_WORDS;
            *reinterpret_cast<uint2 *>(rec) = make_uint2(l[0], l[1]);
            *reinterpret_cast<uint2 *>(rec + 2) = make_uint2(l[2], l[3]);
            *reinterpret_cast<uint2 *>(rec + 4) = make_uint2((uint32_t)cnt, 0u);
            // the 160-bit row itself (read by the chain kernel only for rows with more than 16 set bits)
            uint32_t *out = masks + (rowbase + tl) * BCD_MASK_WORDS;
#pragma unroll
            for (int j = 0; j < BCD_MASK_WORDS; j++) out[j] = m[j];
        }
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
    const uint32_t *masks, *recs;
    uint8_t *back;
};

struct Cand {
    double v; int k;
};

// (value, index) lexicographic minimum, branch-free
__device__ static inline void cand_min(Cand &a, double ov, int ok)
{
    const bool take = ov < a.v || (ov == a.v && ok < a.k);
    a.v = take ? ov : a.v;
    a.k = take ? ok : a.k;
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

#define DPP_ROW_SHL1 0x101
#define DPP_ROW_SHL2 0x102
#define DPP_ROW_SHL4 0x104
#define DPP_ROW_SHL8 0x108

// lexicographic minimum over the 64 lanes of a wave, valid in lane 0
__device__ static inline void wave_min_lane0(Cand &pm)
{
    double ov; int ok;
    cand_dpp<DPP_ROW_SHL1>(pm, ov, ok); cand_min(pm, ov, ok);
    cand_dpp<DPP_ROW_SHL2>(pm, ov, ok); cand_min(pm, ov, ok);
    cand_dpp<DPP_ROW_SHL4>(pm, ov, ok); cand_min(pm, ov, ok);
    cand_dpp<DPP_ROW_SHL8>(pm, ov, ok); cand_min(pm, ov, ok);
    const int lo = __double2loint(pm.v), hi = __double2hiint(pm.v);
#pragma unroll
    for (int r = 1; r < 4; r++) {
        const double rv = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r), __builtin_amdgcn_readlane(lo, 16 * r));
        const int rk = __builtin_amdgcn_readlane(pm.k, 16 * r);
        cand_min(pm, rv, rk);
    }
}


// Minimum of a 64-bit key over the wave (keys = bit patterns of positive doubles, which order like the doubles) and
// the first lane that attains it: 4 DPP steps inside the rows of 16 lanes, the row leaders through v_readlane, then
// one ballot.  Returns the minimum in every lane, *first_lane likewise.
template <int CTRL> __device__ static inline uint32_t u32_dpp_min(uint32_t x)
{
    // lanes without a source keep their own value (bound_ctrl = false, old = self); fuses into v_min_u32_dpp
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, false);
    return o < x ? o : x;
}
// minimum of a 32-bit value over the wave, wave-uniform result: 4 DPP steps inside the rows of 16 lanes, then the row
// leaders through v_readlane and scalar minima
__device__ static inline uint32_t wave_u32_min(uint32_t x)
{
    x = u32_dpp_min<DPP_ROW_SHL1>(x);
    x = u32_dpp_min<DPP_ROW_SHL2>(x);
    x = u32_dpp_min<DPP_ROW_SHL4>(x);
    x = u32_dpp_min<DPP_ROW_SHL8>(x);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
// Minimum of a 64-bit key over the wave (keys = bit patterns of positive doubles, which order like the doubles) and the
// first lane that attains it: high words first, then the low words of the lanes that tie on the high word.
__device__ static inline unsigned long long wave_key_min(unsigned long long key, int *first_lane)
{
    const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    const uint32_t mh = wave_u32_min(hi);
    const uint32_t ml = wave_u32_min(hi == mh ? lo : 0xFFFFFFFFu);
    const unsigned long long hit = __ballot(hi == mh && lo == ml);
    *first_lane = __ffsll((long long)hit) - 1;
    return ((unsigned long long)mh << 32) | ml;
}

// One thread per label: 192 threads (3 waves) cover up to DFLOW_MAX_LABELS labels.  Few, busy threads keep the
// per-step instruction count (and with it the latency of the serial chain) low.
__global__ void __launch_bounds__(BCD_THREADS) bcd_chain_kernel(BcdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *dpbuf = (double *)smem;                                   // [2][BCD_LDS_LABELS]; entries >= 160 stay +inf (list sentinel 0xFF)
    unsigned long long *permv = (unsigned long long *)(dpbuf + 2 * BCD_LDS_LABELS);   // [2][4] per-wave minima of bits(tpsi + dp)
    uint32_t *fpbuf = (uint32_t *)(permv + 2 * 4);                    // [2][BCD_LDS_LABELS] biased flows
    int *permi = (int *)(fpbuf + 2 * BCD_LDS_LABELS);               // [2][4]; permi[8] = traceback hand-over
    uint8_t *tb = (uint8_t *)(permi + 2 * 4 + 4);                     // [BCD_TB_STEPS][LP] traceback chunk (16-byte aligned)
    uint32_t *bestf = (uint32_t *)(tb + BCD_TB_STEPS * a.LP);         // [len] biased flow of each chain pixel's current label
    int *tnl;                                                         // [len] nprop of each chain pixel (set below)
    int *s_label = permi + 8;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tl = tid < DFLOW_MAX_LABELS ? tid : DFLOW_MAX_LABELS - 1;   // threads 160..191 shadow the last label row and never write
    const bool owner = tid < DFLOW_MAX_LABELS;
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

    tnl = (int *)(bestf + len);
    for (int i = tid; i < 2 * BCD_LDS_LABELS; i += BCD_THREADS) { dpbuf[i] = 1e300; fpbuf[i] = 0u; }
    for (int i = tid; i < len; i += BCD_THREADS) {
        int pix = pix0 + i * pstep;
        bestf[i] = flow_bias(a.proposals[(size_t)pix * LP + a.bestlabels[pix]]);
        tnl[i] = a.nprop[pix];      // a wave-uniform global load inside the step loop would stall every step (it is
                                    // moved to an SGPR at once); the label counts are read from LDS instead
    }
    __syncthreads();

    uint8_t *back = a.back + (size_t)chain * len * LP;

    // Per-step inputs of this thread, prefetched three steps ahead.  All loads are unconditional (rows are LP wide;
    // mask rows of unused labels hold garbage that is masked at use), so nothing in a step waits for a load issued
    // in the same step.
    struct StepIn { uint32_t F; float lc; uint4 rl; uint32_t cnt; };   // rl = byte list of the first 16 compatible labels
    auto fetch = [&](int i) {
        StepIn r;
        const int pix = pix0 + min(i, len - 1) * pstep;
        r.F = flow_bias(a.proposals[(size_t)pix * LP + tl]);
        r.lc = a.lcosts[(size_t)pix * LP + tl];
        const uint32_t *rec = a.recs + (((size_t)pix * 2 + dir) * LP + tl) * BCD_REC_WORDS;
        const uint2 r01 = *reinterpret_cast<const uint2 *>(rec), r23 = *reinterpret_cast<const uint2 *>(rec + 2);
        r.rl = make_uint4(r01.x, r01.y, r23.x, r23.y); r.cnt = rec[4];
        return r;
    };
    const StepIn S0 = fetch(0);
    StepIn A = fetch(1), B = fetch(2), C = fetch(3);

    // ---- chain start: dp[0,tl] = (s1 + s2) + lamda*lcost   (python bcd.py:118-120)
    {
        const uint32_t Fc = S0.F;
        const int ip = dirp, im = -dirp;
        const uint32_t s1 = (ip >= 0 && ip < len) ? min(tpsi, flow_l1_biased(Fc, bestf[ip])) : 0u;
        const uint32_t s2 = (im >= 0 && im < len) ? min(tpsi, flow_l1_biased(Fc, bestf[im])) : 0u;
        unsigned long long key = ~0ull;
        if (owner && tl < tnl[0]) {
            const double d0 = __dadd_rn((double)(s1 + s2), __dmul_rn(a.lamda, (double)S0.lc));
            dpbuf[tl] = d0;
            fpbuf[tl] = Fc;
            key = (unsigned long long)__double_as_longlong(__dadd_rn(tpsi_d, d0));
        }
        int fl;
        const unsigned long long m = wave_key_min(key, &fl);
        if (lane == 0) { permv[wave] = m; permi[wave] = wave * 64 + fl; }
    }
    __syncthreads();

    int cur = 1;
    int pn = tnl[0];
    // One step of the chain.  `in` is consumed first and then refilled with the inputs of step i+3: the three slots are
    // used round-robin by the 3x unrolled loop below, so prefetched registers are never copied while their loads are
    // still in flight (a register rotation A=B, B=C would make every step wait for the loads it has just issued).
    auto step = [&](const int i, StepIn &in) __attribute__((always_inline)) {
        const int tn = tnl[i];
        const uint32_t Fc = in.F;
        const float lc = in.lc;
        const bool act = owner && tl < tn;
        const uint4 rl = in.rl; const uint32_t rcnt = in.cnt;
        const size_t rowidx = ((size_t)(pix0 + i * pstep) * 2 + dir) * LP + tl;
        in = fetch(i + 3);
        const double *dp = dpbuf + (cur ^ 1) * BCD_LDS_LABELS;
        const uint32_t *fp = fpbuf + (cur ^ 1) * BCD_LDS_LABELS;

        // min over compatible previous labels (python bcd.py:163-176 / :198-219): walk the set bits of my mask row in
        // increasing k (strict '<' keeps the first minimum); the LDS reads of the next candidate are issued before the
        // current one is evaluated
        // Everything that does not depend on the walk below is issued first, so that its LDS latency hides behind it:
        // permmincost / permminlabel (python bcd.py:152-157) merged from the per-wave partials of the previous step
        // (waves are in label order and every partial index is the first one inside its wave), and the unary term
        // small = (lamda*lcost + s1) + s2 (python bcd.py:161-162).
        Cand perm;
        double small;
        {
            const unsigned long long p0 = permv[(cur ^ 1) * 4], p1 = permv[(cur ^ 1) * 4 + 1], p2 = permv[(cur ^ 1) * 4 + 2];
            const int i0 = permi[(cur ^ 1) * 4], i1 = permi[(cur ^ 1) * 4 + 1], i2 = permi[(cur ^ 1) * 4 + 2];
            const int ip = i + dirp, im = i - dirp;
            const uint32_t s1 = (ip >= 0 && ip < len) ? min(tpsi, flow_l1_biased(Fc, bestf[ip])) : 0u;
            const uint32_t s2 = (im >= 0 && im < len) ? min(tpsi, flow_l1_biased(Fc, bestf[im])) : 0u;
            unsigned long long pmn = p0; int pix_ = i0;
            if (p1 < pmn) { pmn = p1; pix_ = i1; }
            if (p2 < pmn) { pmn = p2; pix_ = i2; }
            perm.v = __longlong_as_double((long long)pmn); perm.k = pix_;
            small = __dadd_rn(__dadd_rn(__dmul_rn(a.lamda, (double)lc), (double)s1), (double)s2);
        }

        double bestv = 1e300; int bestk = 0x7fffffff;
        const int cnt = act ? (int)rcnt : 0;
        {
            // The first 16 compatible predecessors of every row come as a byte list in increasing k (0xFF = none, which
            // reads the +inf tail of dp); the LDS reads of 8 candidates are issued together.
            auto list8 = [&](uint32_t la, uint32_t lb) {
                int kk[8]; double dd[8]; uint32_t ff[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    kk[j] = (int)(((j < 4 ? la : lb) >> (8 * (j & 3))) & 0xFFu);
                    dd[j] = dp[kk[j]]; ff[j] = fp[kk[j]];
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const double c = __dadd_rn(dd[j], (double)flow_l1_biased(Fc, ff[j]));
                    const bool t = c < bestv;               // +inf + psi = +inf never wins
                    bestv = t ? c : bestv; bestk = t ? kk[j] : bestk;
                }
            };
            list8(act ? rl.x : 0xFFFFFFFFu, act ? rl.y : 0xFFFFFFFFu);
            if (__ballot(cnt > 8)) list8(act ? rl.z : 0xFFFFFFFFu, act ? rl.w : 0xFFFFFFFFu);
            if (__ballot(cnt > BCD_LIST)) {
                // some rows are denser still: those lanes fetch their 160-bit row and walk what is left behind the 16th
                // list entry, four set bits per round; still increasing k, so strict '<' stands
                const bool more = cnt > BCD_LIST;
                const int k15 = (int)(rl.w >> 24);
                const uint32_t *mrow = a.masks + rowidx * BCD_MASK_WORDS;
                unsigned long long w0 = 0, w1 = 0, w2 = 0;
                if (more) {
                    w0 = (unsigned long long)mrow[0] | ((unsigned long long)mrow[1] << 32);
                    w1 = (unsigned long long)mrow[2] | ((unsigned long long)mrow[3] << 32);
                    w2 = (unsigned long long)mrow[4];
                    const int b = k15 & 63;
                    const unsigned long long keep = b == 63 ? 0ull : (~0ull << (b + 1));
                    if (k15 < 64) w0 &= keep; else if (k15 < 128) { w0 = 0; w1 &= keep; } else { w0 = 0; w1 = 0; w2 &= keep; }
                }
                int base = 0;
                auto next_bit = [&](bool &valid) {
                    if (w0 == 0) { w0 = w1; w1 = w2; w2 = 0; base += 64; if (w0 == 0) { w0 = w1; w1 = 0; base += 64; } }
                    valid = w0 != 0;
                    int k = 0;
                    if (valid) { k = base + __ffsll((long long)w0) - 1; w0 &= w0 - 1; }
                    return k;
                };
                while (w0 | w1 | w2) {
                    bool v0, v1, v2, v3;
                    const int k0 = next_bit(v0), k1 = next_bit(v1), k2 = next_bit(v2), k3 = next_bit(v3);
                    const double d0 = dp[k0], d1 = dp[k1], d2 = dp[k2], d3 = dp[k3];     // invalid slots read label 0: harmless
                    const uint32_t f0 = fp[k0], f1 = fp[k1], f2 = fp[k2], f3 = fp[k3];
                    const double c0 = __dadd_rn(d0, (double)flow_l1_biased(Fc, f0));
                    const double c1 = __dadd_rn(d1, (double)flow_l1_biased(Fc, f1));
                    const double c2 = __dadd_rn(d2, (double)flow_l1_biased(Fc, f2));
                    const double c3 = __dadd_rn(d3, (double)flow_l1_biased(Fc, f3));
                    bool t;
                    t = v0 && c0 < bestv; bestv = t ? c0 : bestv; bestk = t ? k0 : bestk;
                    t = v1 && c1 < bestv; bestv = t ? c1 : bestv; bestk = t ? k1 : bestk;
                    t = v2 && c2 < bestv; bestv = t ? c2 : bestv; bestk = t ? k2 : bestk;
                    t = v3 && c3 < bestv; bestv = t ? c3 : bestv; bestk = t ? k3 : bestk;
                }
            }
        }
        unsigned long long key = ~0ull;
        if (act) {
            const bool found = bestk != 0x7fffffff;
            const double mincost = found ? bestv : perm.v;
            const int pl = found ? bestk : perm.k;
            const double dpc = __dadd_rn(mincost, small);
            dpbuf[cur * BCD_LDS_LABELS + tl] = dpc;
            fpbuf[cur * BCD_LDS_LABELS + tl] = Fc;
            back[(size_t)i * LP + tl] = (uint8_t)pl;
            key = (unsigned long long)__double_as_longlong(__dadd_rn(tpsi_d, dpc));
        }
        {
            int fl;
            const unsigned long long m = wave_key_min(key, &fl);
            if (lane == 0) { permv[cur * 4 + wave] = m; permi[cur * 4 + wave] = wave * 64 + fl; }
        }
        // LDS-only barrier: __syncthreads() would also wait for the global prefetches issued in this step (vmcnt(0)) and
        // put their full latency on every step of the chain
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        pn = tn;
        cur ^= 1;
    };
    for (int i = 1; i < len; i += 3) {
        step(i, A);
        if (i + 1 < len) step(i + 1, B);
        if (i + 2 < len) step(i + 2, C);
    }

    // ---- end label: first minimum of dp[len-1] (python bcd.py:231-237): tpsi + dp is monotone in dp, but two different
    // dp may round to the same sum, so the minimum is taken over dp itself
    if (tid < 64) {
        const double *dp = dpbuf + (cur ^ 1) * BCD_LDS_LABELS;
        Cand m; m.v = 800000.0; m.k = 0x7fffffff;
        for (int k = lane; k < pn; k += 64) { double c = dp[k]; if (c < m.v) { m.v = c; m.k = k; } }
        wave_min_lane0(m);
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

static size_t rec_bytes(const dflow_params *p)
{
    return (size_t)p->pich * p->picw * 2 * p->label_pitch * BCD_REC_WORDS * sizeof(uint32_t);
}

size_t bcd_ws_bytes(const dflow_params *p) { return back_bytes(p) + mask_bytes(p) + rec_bytes(p); }

int launch_bcd_prepare(const dflow_params *p, const uint32_t *proposals, const int32_t *nprop, void *ws, hipStream_t s)
{
    uint32_t *masks = (uint32_t *)((char *)ws + back_bytes(p));
    uint32_t *recs = (uint32_t *)((char *)ws + back_bytes(p) + mask_bytes(p));
    long long items = 2LL * p->pich * p->picw;
    hipLaunchKernelGGL(bcd_masks_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p->pich, p->picw, p->label_pitch,
                       p->tpsi, proposals, nprop, masks, recs);
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
    a.recs = (const uint32_t *)((char *)ws + back_bytes(p) + mask_bytes(p));
    size_t shmem = 2 * 256 * (sizeof(double) + sizeof(uint32_t)) + 2 * 4 * (sizeof(double) + sizeof(int)) + 16 +
                   (size_t)BCD_TB_STEPS * p->label_pitch + (size_t)len * (sizeof(uint32_t) + sizeof(int));
    hipLaunchKernelGGL(bcd_chain_kernel, dim3(nchains), dim3(BCD_THREADS), shmem, s, a);
    return dflow_check_launch("bcd_chain_kernel");
}
