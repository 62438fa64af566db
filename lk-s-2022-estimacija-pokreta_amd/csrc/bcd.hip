// K6 BCD: bcd() (python bcd.py:101-257) as scheduled by ceoBCD (python bcd.py:261-284), plus pakovanje's compat
// bit matrices (daisy i flann.py:256-309) in the orientation the chains need them.
//
// Two kernels:
//   bcd_masks_kernel   once per pass (the proposals do not change during the sweeps): for every pixel p and both of
//                      its chains (column chain / row chain) and every label tl of p, the set { k : tpsi > |dy-dy'|+|dx-dx'|
//                      between label tl of p and label k of p's predecessor on that chain } -- the reference's
//                      packedksets (Q8), restricted to the two neighbours that are ever used and already transposed
//                      for the direction in which the chain runs.  lanes = labels of p, the predecessor's labels are
//                      wave-uniform: one v_sad_u16 + one v_alignbit per pair builds the 160-bit row in registers.  What
//                      the chain kernel reads every step is a 32-byte record per row: the first 16 members as bytes,
//                      their pairwise costs as nibbles, and the label's own flow and data cost.
//   bcd_chain_kernel   one workgroup per chain of a phase (all chains of a phase are independent: a chain reads and
//                      writes only its own image line).  192 threads, one per label; a lane walks its label's list
//                      (only compatible predecessors cost float64 work), float64 arithmetic in the reference's
//                      association order (dp = mincost + ((lamda*lcost + s1) + s2); first-index ties everywhere, Q9-Q11).
//                      dp lives in LDS (double buffered), the records are prefetched three steps ahead, back-pointers
//                      go to the workspace as uint8 and are walked chunk-wise from LDS.
#include "dflow_common.h"

#define BCD_THREADS 192
#define BCD_MASK_WORDS 5                 // 160 bits per label row (kept in HBM only for rows with more than 16 members)
#define BCD_REC_WORDS 8                  // row record: 4 words = the first 16 members as bytes, increasing (0xFF = none);
                                         // 2 words = their pairwise costs |dy-dy'|+|dx-dx'| (< tpsi <= 8) as nibbles, bit 63 =
                                         // "more than 16 members"; 1 word = the label's biased flow; 1 word = its data cost
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
                                                        const float *__restrict__ lcosts, const int32_t *__restrict__ nprop,
                                                        uint32_t *__restrict__ masks, uint32_t *__restrict__ recs)
{
    __shared__ uint32_t s_cols[4][192];                      // the predecessor's biased labels, per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long item = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wv);
    if (item >= 2LL * H * W) return;
    const int dir = (int)(item & 1);
    const int pix = (int)(item >> 1);
    const int y = pix / W, x = pix % W;
    // predecessor on the chain (python bcd.py:265-277): even columns run down, odd columns up, even rows leftwards, odd rows rightwards
    int py = y, px = x;
    if (dir == 0) py = (x & 1) ? y + 1 : y - 1; else px = (y & 1) ? x - 1 : x + 1;
    const bool start = py < 0 || py >= H || px < 0 || px >= W;     // chain start: no transition into this pixel
    const int ppix = start ? pix : py * W + px;
    const int tn = nprop[pix], pn = start ? 0 : nprop[ppix];
    uint32_t fp[3], fcv[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int k = lane + 64 * j;
        fp[j] = k < pn ? flow_bias(proposals[(size_t)ppix * LP + k]) : 0u;   // 0: far from every biased flow
        fcv[j] = k < tn ? flow_bias(proposals[(size_t)pix * LP + k]) : 0u;
        s_cols[wv][k] = fp[j];
    }
    // lanes = labels of this pixel (three groups of 64), predecessor labels come one by one as wave-uniform scalars
    // (scalar loads of the predecessor's row): D = sad + (2^31 - tpsi) has bit 31 set iff the pair is NOT compatible,
    // v_alignbit shifts that bit into the row word.  Columns run downwards inside each 32-bit word so that column c ends
    // up in bit c; columns >= pn (fill values) are forced to "not compatible" afterwards.
    const uint32_t kbias = 0x80000000u - (uint32_t)tpsi;
    const uint32_t *__restrict__ colp = proposals + (size_t)ppix * LP;
    uint32_t m[3][BCD_MASK_WORDS];
#pragma unroll
    for (int j = 0; j < BCD_MASK_WORDS; j++) {
        m[0][j] = m[1][j] = m[2][j] = 0xFFFFFFFFu;
        if (32 * j < pn) {
#pragma unroll
            for (int cc = 31; cc >= 0; cc--) {
                const uint32_t col = flow_bias(colp[32 * j + cc]);
                m[0][j] = __builtin_amdgcn_alignbit(m[0][j], __builtin_amdgcn_sad_u16(col, fcv[0], kbias), 31);
                m[1][j] = __builtin_amdgcn_alignbit(m[1][j], __builtin_amdgcn_sad_u16(col, fcv[1], kbias), 31);
                m[2][j] = __builtin_amdgcn_alignbit(m[2][j], __builtin_amdgcn_sad_u16(col, fcv[2], kbias), 31);
            }
            const int nv = pn - 32 * j;                      // valid columns in this word (wave-uniform)
            const uint32_t inval = nv >= 32 ? 0u : ~0u << nv;
            m[0][j] |= inval; m[1][j] |= inval; m[2][j] |= inval;
        }
    }
    const size_t rowbase = ((size_t)pix * 2 + dir) * (size_t)LP;
#pragma unroll
    for (int grp = 0; grp < 3; grp++) {
        const int tl = 64 * grp + lane;
        if (64 * grp >= tn) break;                           // wave-uniform
        if (tl < tn) {
            // the first 16 members as bytes and their pairwise costs as nibbles.  Both lists are shift registers filled
            // from the top (position-independent inserts) and moved down to their final place afterwards.
            uint32_t w[BCD_MASK_WORDS];
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < BCD_MASK_WORDS; j++) { w[j] = ~m[grp][j]; cnt += __popc(w[j]); }
            uint32_t l0 = 0xFFFFFFFFu, l1 = 0xFFFFFFFFu, l2 = 0xFFFFFFFFu, l3 = 0xFFFFFFFFu, p0 = 0u, p1 = 0u;
            int n = 0;
            const uint32_t me = fcv[grp];
#pragma unroll
            for (int j = 0; j < BCD_MASK_WORDS; j++) {
                uint32_t ww = w[j];
                while (ww && n < BCD_LIST) {
                    const uint32_t k = 32 * j + __ffs(ww) - 1; ww &= ww - 1;
                    const uint32_t psi = flow_l1_biased(me, s_cols[wv][k]);
                    l0 = __builtin_amdgcn_alignbit(l1, l0, 8); l1 = __builtin_amdgcn_alignbit(l2, l1, 8);
                    l2 = __builtin_amdgcn_alignbit(l3, l2, 8); l3 = __builtin_amdgcn_alignbit(k, l3, 8);
                    p0 = __builtin_amdgcn_alignbit(p1, p0, 4); p1 = __builtin_amdgcn_alignbit(psi, p1, 4);
                    n++;
                }
            }
            // n entries sit in the top n bytes / nibbles: shift right by 16-n places, 0xFF / 0 come in from the top
            {
                const int sh = BCD_LIST - n;                  // 0..16
                if (sh & 1) { l0 = __builtin_amdgcn_alignbit(l1, l0, 8); l1 = __builtin_amdgcn_alignbit(l2, l1, 8);
                              l2 = __builtin_amdgcn_alignbit(l3, l2, 8); l3 = __builtin_amdgcn_alignbit(0xFFFFFFFFu, l3, 8); }
                if (sh & 2) { l0 = __builtin_amdgcn_alignbit(l1, l0, 16); l1 = __builtin_amdgcn_alignbit(l2, l1, 16);
                              l2 = __builtin_amdgcn_alignbit(l3, l2, 16); l3 = __builtin_amdgcn_alignbit(0xFFFFFFFFu, l3, 16); }
                if (sh & 4) { l0 = l1; l1 = l2; l2 = l3; l3 = 0xFFFFFFFFu; }
                if (sh & 8) { l0 = l2; l1 = l3; l2 = 0xFFFFFFFFu; l3 = 0xFFFFFFFFu; }
                if (sh & 16) { l0 = l1 = l2 = l3 = 0xFFFFFFFFu; }
                unsigned long long pp = ((unsigned long long)p1 << 32) | p0;
                pp = n ? pp >> (4 * sh) : 0ull;
                p0 = (uint32_t)pp; p1 = (uint32_t)(pp >> 32);
            }
            if (cnt > BCD_LIST) p1 |= 0x80000000u;
            uint32_t *rec = recs + (rowbase + tl) * BCD_REC_WORDS;
            *reinterpret_cast<uint4 *>(rec) = make_uint4(l0, l1, l2, l3);
            *reinterpret_cast<uint4 *>(rec + 4) = make_uint4(p0, p1, me, __float_as_uint(lcosts[(size_t)pix * LP + tl]));
            if (cnt > BCD_LIST) {
                // the 160-bit row itself: read by the chain kernel only for rows with more than 16 members
                uint32_t *out = masks + (rowbase + tl) * BCD_MASK_WORDS;
#pragma unroll
                for (int j = 0; j < BCD_MASK_WORDS; j++) out[j] = w[j];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ chains
struct BcdArgs {
    int H, W, LP, tpsi, phase;
    double lamda;
    const uint32_t *proposals;
    const int32_t *nprop;
    int32_t *bestlabels;
    const uint32_t *masks, *recs;
    uint8_t *back, *back_trash;
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


// Minimum of a 64-bit key over the wave (keys = bit patterns of positive doubles, which order like the doubles) and the
// first lane that attains it: high words first, then the low words of the lanes that tie on the high word.
// minimum of a 32-bit value over the wave, wave-uniform result: 4 fused DPP minima inside the rows of 16 lanes (lanes
// without a source keep their own value; s_nop 1 = the two wait states a DPP read of a fresh VALU result needs), then
// the row leaders through v_readlane and scalar minima
__device__ static inline uint32_t wave_u32_min_asm(uint32_t x)
{
    asm volatile("s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0"
                 : "+v"(x));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ static inline unsigned long long wave_key_min_asm(unsigned long long key, int *first_lane)
{
    const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    const uint32_t mh = wave_u32_min_asm(hi);
    const unsigned long long tie = __ballot(hi == mh);
    uint32_t ml; int fl;
    // one lane alone has the smallest high word (the usual case): it is the minimum
    fl = __ffsll((long long)tie) - 1;
    ml = (uint32_t)__builtin_amdgcn_readlane((int)lo, fl);
    if (__builtin_expect(__popcll(tie) != 1, 0)) {
        ml = wave_u32_min_asm(hi == mh ? lo : 0xFFFFFFFFu);
        fl = __ffsll((long long)__ballot(hi == mh && lo == ml)) - 1;
    }
    *first_lane = fl;
    return ((unsigned long long)mh << 32) | ml;
}

template <int V> struct IntC { static constexpr int value = V; };

#ifdef BCD_PROF
// diagnostic build only (scratch/): per-wave cycle stamps of the step sections of block 0, summed over the chain
__device__ unsigned long long g_bcd_prof[3][8];
extern "C" void dflow_debug_bcd_prof(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bcd_prof), sizeof(g_bcd_prof)); }
extern "C" void dflow_debug_bcd_prof_reset() { unsigned long long z[24] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bcd_prof), z, sizeof(z)); }
#define PROF_DECL unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt = 0;
#define PROF_START pt = __builtin_amdgcn_s_memtime();
#define PROF(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc[k] += n_ - pt; pt = n_; }
#else
#define PROF_DECL
#define PROF_START
#define PROF(k)
#endif

// One thread per label: 192 threads (3 waves) cover up to DFLOW_MAX_LABELS labels.  Few, busy threads keep the
// per-step instruction count low: every wave is alone on its SIMD, so a step costs (instructions x issue cycles).
__global__ void __launch_bounds__(BCD_THREADS) bcd_chain_kernel(BcdArgs a)
{
    // static LDS has compile-time addresses, so the offsets fold into the ds_read immediates
    __shared__ double s_dp[2 * BCD_LDS_LABELS];                               // [2][labels]; entries >= 160 stay +inf (list sentinel 0xFF)
    __shared__ uint32_t s_fp[2 * BCD_LDS_LABELS];                             // [2][labels] biased flows (rows with more than 16 members only)
    __shared__ unsigned long long permv[2 * 4];                               // per-wave minima of bits(tpsi + dp)
    __shared__ int permi[2 * 4 + 4];                                          // their labels; [8] = traceback hand-over
    __shared__ __attribute__((aligned(16))) uint8_t tb[BCD_TB_STEPS * DFLOW_MAX_LABELS];   // traceback chunk
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    uint32_t *bestf = s_dyn + 1;                                      // [-1..len] biased flow of each chain pixel's current label
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

    tnl = (int *)(bestf + len + 1);
    for (int i = tid; i < 2 * BCD_LDS_LABELS; i += BCD_THREADS) { s_dp[i] = 1e300; s_fp[i] = 0u; }
    for (int i = tid; i < len; i += BCD_THREADS) {
        int pix = pix0 + i * pstep;
        bestf[i] = flow_bias(a.proposals[(size_t)pix * LP + a.bestlabels[pix]]);
        tnl[i] = a.nprop[pix];      // a wave-uniform global load inside the step loop would stall every step (it is
                                    // moved to an SGPR at once); the label counts are read from LDS instead
    }
    if (tid == 0) { bestf[-1] = 0u; bestf[len] = 0u; }       // read, never used (the limit of that side term is 0)
    __syncthreads();

    // Per-step inputs of this thread = its label's 32-byte record (see BCD_REC_WORDS), prefetched three steps ahead.
    // All loads are unconditional (rows are LP wide; records of unused labels hold garbage that indexes inside the LDS
    // arrays and is never written back), so nothing in a step waits for a load issued in the same step.
    struct StepIn { uint4 rl; uint4 px; };   // rl = byte list; px = {cost nibbles lo, hi | more flag, biased flow, data cost}
    const uint32_t offr = (uint32_t)tl * (BCD_REC_WORDS * 4u);
    const size_t rowbytesr = (size_t)LP * (BCD_REC_WORDS * 4u);
    auto fetch = [&](int i) {
        StepIn r;
        const size_t pix = (size_t)(pix0 + min(i, len - 1) * pstep);
        const char *pr = (const char *)a.recs + (pix * 2 + dir) * rowbytesr;
        r.rl = *reinterpret_cast<const uint4 *>(pr + offr);
        r.px = *reinterpret_cast<const uint4 *>(pr + offr + 16);
        return r;
    };
    const StepIn S0 = fetch(0);
    StepIn A = fetch(1), B = fetch(2), C = fetch(3);

    // ---- chain start: dp[0,tl] = (s1 + s2) + lamda*lcost   (python bcd.py:118-120)
    {
        const uint32_t Fc = S0.px.z;
        const int ip = dirp, im = -dirp;
        const uint32_t s1 = (ip >= 0 && ip < len) ? min(tpsi, flow_l1_biased(Fc, bestf[ip])) : 0u;
        const uint32_t s2 = (im >= 0 && im < len) ? min(tpsi, flow_l1_biased(Fc, bestf[im])) : 0u;
        unsigned long long key = ~0ull;
        if (owner && tl < tnl[0]) {
            const double d0 = __dadd_rn((double)(s1 + s2), __dmul_rn(a.lamda, (double)__uint_as_float(S0.px.w)));
            s_dp[tl] = d0;
            s_fp[tl] = Fc;
            key = (unsigned long long)__double_as_longlong(__dadd_rn(tpsi_d, d0));
        }
        int fl;
        const unsigned long long m = wave_key_min_asm(key, &fl);
        if (lane == 0) { permv[wave] = m; permi[wave] = wave * 64 + fl; }
    }
    __syncthreads();

    PROF_DECL
    int pn = tnl[0];
    uint8_t *backp = owner ? a.back + (size_t)chain * len * LP + tl : a.back_trash + (tid - DFLOW_MAX_LABELS);   // + i*LP per step
    const uint32_t bstride = owner ? (uint32_t)LP : 0u;
    // One step of the chain; CUR (compile-time) is the LDS buffer this step writes, CUR^1 holds the previous pixel.  `in`
    // is consumed first and then refilled with the record of step i+3: the three slots are used round-robin by the 6x
    // unrolled loop below, so prefetched registers are never copied while their loads are still in flight (a register
    // rotation A=B, B=C would make every step wait for the loads it has just issued).  The common path of a step is one
    // basic block (the wave is alone on its SIMD: every wave-uniform branch costs more than a few wasted instructions).
    auto step = [&](auto curc, const int i, StepIn &in) __attribute__((always_inline)) {
        constexpr int CUR = decltype(curc)::value;
        const char *prev = reinterpret_cast<const char *>(s_dp + (CUR ^ 1) * BCD_LDS_LABELS);
        const uint32_t *fprev = s_fp + (CUR ^ 1) * BCD_LDS_LABELS;
        PROF_START
        const int tn = tnl[i];
        const uint4 rl = in.rl;
        const uint32_t pw0 = in.px.x, pw1 = in.px.y, Fc = in.px.z;
        const float lc = __uint_as_float(in.px.w);
        const bool act = owner && tl < tn;
        in = fetch(i + 3);
        PROF(0)
        // min over compatible previous labels (python bcd.py:163-176 / :198-219) in increasing k (strict '<' keeps the
        // first minimum).  The first 16 compatible predecessors of every row come as a byte list in increasing k (0xFF =
        // none, which reads the +inf tail of dp) together with their pair costs; the LDS reads of the first 12 are issued
        // at once (most waves need 9 to 12; a wave-uniform exit after 8 costs as much as it saves).  Candidates are
        // tracked by their dp offset 8 k.
        uint32_t ad[12]; double dd[12];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const uint32_t w = j < 4 ? rl.x : (j < 8 ? rl.y : rl.z);
            ad[j] = ((w >> (8 * (j & 3))) & 0xFFu) << 3;
            dd[j] = *reinterpret_cast<const double *>(prev + ad[j]);
        }
        // permmincost / permminlabel (python bcd.py:152-157) merged from the per-wave partials of the previous step (waves
        // are in label order and every partial index is the first one inside its wave), and the unary term
        // small = (lamda*lcost + s1) + s2 (python bcd.py:161-162; a side neighbour outside the chain contributes 0)
        Cand perm;
        double small;
        {
            const unsigned long long p0 = permv[(CUR ^ 1) * 4], p1 = permv[(CUR ^ 1) * 4 + 1], p2 = permv[(CUR ^ 1) * 4 + 2];
            const int i0 = permi[(CUR ^ 1) * 4], i1 = permi[(CUR ^ 1) * 4 + 1], i2 = permi[(CUR ^ 1) * 4 + 2];
            const int ip = i + dirp, im = i - dirp;
            const uint32_t lim1 = (ip >= 0 && ip < len) ? tpsi : 0u, lim2 = (im >= 0 && im < len) ? tpsi : 0u;
            const uint32_t s1 = min(lim1, flow_l1_biased(Fc, bestf[ip]));
            const uint32_t s2 = min(lim2, flow_l1_biased(Fc, bestf[im]));
            unsigned long long pmn = p0; int pix_ = i0;
            if (p1 < pmn) { pmn = p1; pix_ = i1; }
            if (p2 < pmn) { pmn = p2; pix_ = i2; }
            perm.v = __longlong_as_double((long long)pmn); perm.k = pix_;
            small = __dadd_rn(__dadd_rn(__dmul_rn(a.lamda, (double)lc), (double)s1), (double)s2);
        }
        PROF(1)
        double bestv = 1e300; uint32_t besta = 0x7fffffffu;
        {
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const double c = __dadd_rn(dd[j], (double)(((j < 8 ? pw0 : pw1) >> (4 * (j & 7))) & 7u));
                const bool t = c < bestv;               // +inf + psi = +inf never wins
                bestv = __builtin_fmin(bestv, c); besta = t ? ad[j] : besta;
            }
            PROF(2)
            // few rows have more than 12 members (wave-uniform branch on the longest row of the wave), fewer still more than 16
            if (__builtin_expect(__ballot(act && (rl.w & 0xFFu) != 0xFFu) != 0ull, 0)) {
                {
                    uint32_t a4[4]; double d4[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { a4[j] = ((rl.w >> (8 * j)) & 0xFFu) << 3; d4[j] = *reinterpret_cast<const double *>(prev + a4[j]); }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const double c = __dadd_rn(d4[j], (double)((pw1 >> (4 * (4 + j))) & 7u));
                        const bool t = c < bestv;
                        bestv = __builtin_fmin(bestv, c); besta = t ? a4[j] : besta;
                    }
                }
                PROF(3)
                const bool more16 = act && (int)pw1 < 0;
                if (__builtin_expect(__ballot(more16) != 0ull, 0)) {
                    // some rows are denser still: those lanes fetch their 160-bit row and walk what is left behind the 16th
                    // list entry, four set bits per round; still increasing k, so strict '<' stands
                    const int k15 = (int)(rl.w >> 24);
                    const size_t rowidx = ((size_t)(pix0 + i * pstep) * 2 + dir) * LP + tl;
                    const uint32_t *mrow = a.masks + rowidx * BCD_MASK_WORDS;
                    unsigned long long w0 = 0, w1 = 0, w2 = 0;
                    if (more16) {
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
                        bool v[4]; int kk[4]; double d4[4]; uint32_t ff[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) kk[j] = next_bit(v[j]);
#pragma unroll
                        for (int j = 0; j < 4; j++) { d4[j] = *reinterpret_cast<const double *>(prev + 8 * kk[j]); ff[j] = fprev[kk[j]]; }   // invalid slots read label 0: harmless
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const double c = __dadd_rn(d4[j], (double)flow_l1_biased(Fc, ff[j]));
                            const bool t = v[j] && c < bestv;
                            bestv = t ? c : bestv; besta = t ? (uint32_t)kk[j] << 3 : besta;
                        }
                    }
                }
            }
        }
        PROF(4)
        // No branch on `act`: the slots of labels beyond this pixel's count (dp, flow, back-pointer) may hold anything, no
        // list and no traceback refers to them; the shadow lanes (tid >= 160) write to LDS slots 160..191 (equally
        // unreferenced) and to a trash line behind the back-pointer array.
        unsigned long long key;
        {
            const bool found = besta != 0x7fffffffu;
            const double mincost = found ? bestv : perm.v;
            const int pl = found ? (int)(besta >> 3) : perm.k;
            const double dpc = __dadd_rn(mincost, small);
            s_dp[CUR * BCD_LDS_LABELS + tid] = dpc;
            s_fp[CUR * BCD_LDS_LABELS + tid] = Fc;
            backp[(size_t)i * bstride] = (uint8_t)pl;
            key = act ? (unsigned long long)__double_as_longlong(__dadd_rn(tpsi_d, dpc)) : ~0ull;
        }
        PROF(5)
        {
            int fl;
            const unsigned long long m = wave_key_min_asm(key, &fl);
            if (lane == 0) { permv[CUR * 4 + wave] = m; permi[CUR * 4 + wave] = wave * 64 + fl; }
        }
        PROF(6)
        // LDS-only barrier: __syncthreads() would also wait for the global prefetches issued in this step (vmcnt(0)) and
        // put their full latency on every step of the chain
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PROF(7)
        pn = tn;
    };
    int i = 1;
    for (; i + 5 < len; i += 6) {                       // whole groups of six without per-step bounds checks
        step(IntC<1>(), i, A);
        step(IntC<0>(), i + 1, B);
        step(IntC<1>(), i + 2, C);
        step(IntC<0>(), i + 3, A);
        step(IntC<1>(), i + 4, B);
        step(IntC<0>(), i + 5, C);
    }
    if (i < len) step(IntC<1>(), i, A);
    if (i + 1 < len) step(IntC<0>(), i + 1, B);
    if (i + 2 < len) step(IntC<1>(), i + 2, C);
    if (i + 3 < len) step(IntC<0>(), i + 3, A);
    if (i + 4 < len) step(IntC<1>(), i + 4, B);
#ifdef BCD_PROF
    if (blockIdx.x == 7 && lane == 0) for (int k = 0; k < 8; k++) g_bcd_prof[wave][k] += pacc[k];
#endif
    const int cur = (len & 1) ? 1 : 0;         // the buffer the step after the last one would write; the last written is cur^1

    // ---- end label: first minimum of dp[len-1] (python bcd.py:231-237): tpsi + dp is monotone in dp, but two different
    // dp may round to the same sum, so the minimum is taken over dp itself
    if (tid < 64) {
        const double *dp = s_dp + (cur ^ 1) * BCD_LDS_LABELS;
        Cand m; m.v = 800000.0; m.k = 0x7fffffff;
        for (int k = lane; k < pn; k += 64) { double c = dp[k]; if (c < m.v) { m.v = c; m.k = k; } }
        wave_min_lane0(m);
        if (tid == 0) *s_label = m.k == 0x7fffffff ? 0 : m.k;
    }
    __syncthreads();
    // ---- traceback (python bcd.py:239-253): chunks of back-pointer rows are staged in LDS, one thread walks them
    uint8_t *back = a.back + (size_t)chain * len * LP;
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
    return ((m + 255) & ~(size_t)255) + 256;      // + a trash line for the shadow lanes of the chain kernel
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

int launch_bcd_prepare(const dflow_params *p, const uint32_t *proposals, const float *lcosts, const int32_t *nprop, void *ws,
                       hipStream_t s)
{
    uint32_t *masks = (uint32_t *)((char *)ws + back_bytes(p));
    uint32_t *recs = (uint32_t *)((char *)ws + back_bytes(p) + mask_bytes(p));
    long long items = 2LL * p->pich * p->picw;
    hipLaunchKernelGGL(bcd_masks_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p->pich, p->picw, p->label_pitch,
                       p->tpsi, proposals, lcosts, nprop, masks, recs);
    return dflow_check_launch("bcd_masks_kernel");
}

int launch_bcd_phase(const dflow_params *p, const uint32_t *proposals, const int32_t *nprop, int32_t *bestlabels, int phase,
                     void *ws, hipStream_t s)
{
    int nchains, len;
    phase_dims(p, phase, nchains, len);
    if (nchains == 0) return DFLOW_OK;
    BcdArgs a;
    a.H = p->pich; a.W = p->picw; a.LP = p->label_pitch; a.tpsi = p->tpsi; a.phase = phase; a.lamda = p->lamda;
    a.proposals = proposals; a.nprop = nprop; a.bestlabels = bestlabels;
    a.back = (uint8_t *)ws; a.back_trash = (uint8_t *)ws + back_bytes(p) - 256; a.masks = (const uint32_t *)((char *)ws + back_bytes(p));
    a.recs = (const uint32_t *)((char *)ws + back_bytes(p) + mask_bytes(p));
    size_t shmem = (size_t)(len + 2) * sizeof(uint32_t) + (size_t)len * sizeof(int);
    hipLaunchKernelGGL(bcd_chain_kernel, dim3(nchains), dim3(BCD_THREADS), shmem, s, a);
    return dflow_check_launch("bcd_chain_kernel");
}
