// K6 BCD: bcd() (python bcd.py:101-257) as scheduled by ceoBCD (python bcd.py:261-284), plus pakovanje's compat
// bit matrices (daisy i flann.py:256-309) in the orientation the chains need them.
//
// Two kernels:
//   bcd_lists_kernel   once per pass (the proposals do not change during the sweeps): for every pixel p, both of its
//                      chains (column chain / row chain) and every label tl of p, the set { k : tpsi > |dy-dy'|+|dx-dx'|
//                      between label tl of p and label k of p's predecessor on that chain } -- the reference's
//                      packedksets (Q8), restricted to the two neighbours that are ever used and already transposed
//                      for the direction in which the chain runs.  One wave per PREDECESSOR pixel: its labels are
//                      wave-uniform, lanes = the labels of its two successors (column chain, row chain); one v_sad_u16 +
//                      one v_alignbit per 64 pairs builds the 160-bit rows in registers.  What
//                      the chain kernel streams is compact: per (pixel, direction, label) ONE 8-byte block with the first 5
//                      members (index bytes + their pair costs) and the member count; labels with more than 5 (21 %) /
//                      more than 10 (2.6 %) members own a second / third block, stored COMPACTED per 64-label wave (ballot
//                      rank), so only the blocks that exist are ever fetched; rows with more than 15 members (0.2 %) keep
//                      their 160-bit row.  Per pixel one 8-byte {biased flow, data cost} per label serves both directions.
//   bcd_chain_kernel   one workgroup per (chain of a phase, pass of the batch): all chains of a phase are independent (a
//                      chain reads and writes only its own image line), and so are the passes of a batch (README.md:40 of
//                      the reference).  192 threads, one per label; a lane walks its label's list (only compatible
//                      predecessors cost float64 work), float64 arithmetic in the reference's association order
//                      (dp = mincost + ((lamda*lcost + s1) + s2); first-index ties everywhere, Q9-Q11).  dp lives in LDS
//                      (double buffered), the blocks are prefetched four (first block, label data) and two (second and
//                      third block) steps ahead, back-pointers go to the workspace as uint8 and are walked chunk-wise from
//                      LDS.  11 KB of static + the chain's dynamic LDS and 88 VGPRs (__launch_bounds__(192, 5)): six
//                      workgroups per CU (the measured optimum), so with the chains of 8 passes in one
//                      launch the kernel runs in a throughput regime instead of waiting on one chain step at a time.
#include "dflow_common.h"

#define BCD_THREADS 192
#ifndef BCD_MINWAVES
#define BCD_MINWAVES 5                   // waves per SIMD the chain kernel is compiled for (96 VGPRs: six workgroups per CU)
#endif
#define BCD_MASK_WORDS 5                 // 160 bits per label row (kept in HBM only for rows with more than 15 members)
#define BCD_BLK 5                        // list members per 8-byte block
#define BCD_LIST 15                      // members carried by blocks (3 blocks); longer rows continue in their bit row
#define BCD_LDS_LABELS 256
#define BCD_TB_STEPS 32                  // traceback chunk (steps) staged in LDS
#define BCD_GROUPS 3                     // 64-label waves per pixel
// Reserved second / third blocks per 64-label wave.  21 % / 2.6 % of the labels own one (mean 13 / 1.6 per wave of kNN
// labels, scratch/list_stats.py (round 3, git history): more than 40 / 24 on 0.1 % of the waves; the third wave holds the 22 neighbour labels: never
// more than 16 / 8); a label whose block does not fit (and every label with more than 15 members) is marked "more" and
// continues in its 160-bit row.  Those rows are bump-allocated from a pool (BCD_POOL_DIV-th of the worst case); a row that
// does not fit the pool is not stored and the chain kernel tests that label's remaining predecessors one by one.
#define BCD_CAP_B01 40                   // waves 0 and 1
#define BCD_CAP_B2 16                    // wave 2 (22 labels at most)
#define BCD_CAP_C01 24
#define BCD_CAP_C2 8
#define BCD_ROW_B (2 * BCD_CAP_B01 + BCD_CAP_B2)        // slots per (pixel, direction): second blocks, then the third blocks
#define BCD_ROW_C (2 * BCD_CAP_C01 + BCD_CAP_C2)
#define BCD_ROW_BC (BCD_ROW_B + BCD_ROW_C + 2)     // + 16 bytes: first pool row of each wave's labels marked "more" (3 x uint32)
#define BCD_POOL_DIV 32

// 8-byte block: x = member indices k0..k3 (bytes, increasing, 0xFF = none: reads the +inf tail of dp);
//               y = k4 | pair costs (nibble j = |dy-dy'|+|dx-dx'| < tpsi <= 8 of member j) << 8 | first block only:
//                   bit 11 = "more": members beyond the blocks that are stored, bit 15 = second block stored, bit 19 = third
//                   block stored (the top bits of the cost nibbles are free), bits 28..31 = min(count, 15), 15 if "more"
#define BLK_MORE 0x800u
#define BLK_HAS_B 0x8000u
#define BLK_HAS_C 0x80000u
#define BLK_EMPTY_X 0xFFFFFFFFu
#define BLK_EMPTY_Y 0x000000FFu

__device__ static inline void chain_geom(int phase, int chain, int H, int W, int &ty, int &tx, int &ys, int &xs, int &len)
{
    // python bcd.py:265-277
    if (phase == 0) { ty = 0; tx = 2 * chain; ys = 1; xs = 0; len = H; }
    else if (phase == 1) { ty = 2 * chain; tx = W - 1; ys = 0; xs = -1; len = W; }
    else if (phase == 2) { ty = H - 1; tx = (W / 2) * 2 - 1 - 2 * chain; ys = -1; xs = 0; len = H; }
    else { ty = (H / 2) * 2 - 1 - 2 * chain; tx = 0; ys = 0; xs = 1; len = W; }
}

// workspace layout (all planes 256-byte aligned), shared by both kernels
struct BcdPlanes {
    uint8_t *back;                   // back-pointers of the running phase [chain][step][192 threads]
    uint2 *lab;                      // [pix][LP]            {biased flow, data cost}
    uint2 *blkA;                     // [pix][2][LP]         first block of every label
    uint2 *blkBC;                    // [pix][2][BCD_ROW_BC] second blocks (BCD_ROW_B slots), third blocks, compacted per 64-label wave; pool bases
    uint32_t *masks;                 // [pool_rows][5]       160-bit rows of the labels marked "more", bump-allocated
    uint32_t *cursor;                // next free pool row (zeroed before bcd_lists_kernel runs)
    uint32_t pool_rows;
};

template <int V> struct BcdC { static constexpr int value = V; };

// ------------------------------------------------------------------------------------------------ lists
// grid: one wave per pixel q, as the PREDECESSOR: q's labels are the wave-uniform side of the tests for both pixels that
// have q in front of them -- its successor on the column chain (dir 0) and on the row chain (dir 1) -- so the scalar loads
// of q's row are shared, and the labels 128..159 of the two successors (22 of 32 in use) share ONE group of 64 lanes:
// 5 lane groups for 300 labels where one wave per (pixel, direction) had 6.  (The bench responds to this kernel's VALU
// count one to one)  A pixel that starts a chain has no predecessor: its empty rows are written by its own wave.
__global__ void __launch_bounds__(256) bcd_lists_kernel(int H, int W, int LP, int tpsi, const uint32_t *__restrict__ proposals,
                                                        const float *__restrict__ lcosts, const int32_t *__restrict__ nprop,
                                                        BcdPlanes pl)
{
    __shared__ uint32_t s_cols[4][192];                      // q's biased labels, per wave
    __shared__ __attribute__((aligned(16))) uint8_t s_list[4][64][16];   // member lists being built, per wave and lane
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wv);
    if (q >= H * W) return;
    const int y = q / W, x = q % W;
    // predecessors on the chains (python bcd.py:265-277): even columns run down, odd columns up, even rows leftwards, odd
    // rows rightwards; the successors lie on the other side
    const int py = (x & 1) ? y + 1 : y - 1, sy = (x & 1) ? y - 1 : y + 1;
    const int px = (y & 1) ? x - 1 : x + 1, sx = (y & 1) ? x + 1 : x - 1;
    const bool start0 = py < 0 || py >= H, start1 = px < 0 || px >= W;      // q starts its column / row chain
    const bool v0 = sy >= 0 && sy < H, v1 = sx >= 0 && sx < W;              // q has a successor on its column / row chain
    const int p0 = v0 ? sy * W + x : q, p1 = v1 ? y * W + sx : q;
    const int pn = nprop[q], tn0 = v0 ? nprop[p0] : 0, tn1 = v1 ? nprop[p1] : 0;
    // f[0], f[1]: labels lane, 64 + lane of p0;  f[2], f[3]: of p1;  f[4]: labels 128 + (lane & 31) of p0 (lanes 0..31) / p1 (32..63)
    const int half = lane >> 5;
    uint32_t fq[3], f[5];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int k = lane + 64 * j;
        fq[j] = k < pn ? flow_bias(proposals[(size_t)q * LP + k]) : 0u;      // 0: far from every biased flow
        s_cols[wv][k] = fq[j];
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int k = lane + 64 * j;
        f[j] = k < tn0 ? flow_bias(proposals[(size_t)p0 * LP + k]) : 0u;
        f[2 + j] = k < tn1 ? flow_bias(proposals[(size_t)p1 * LP + k]) : 0u;
    }
    {
        const int k = 128 + (lane & 31);
        f[4] = k < (half ? tn1 : tn0) ? flow_bias(proposals[(size_t)(half ? p1 : p0) * LP + k]) : 0u;
    }
    // q's label data, once (both directions read it)
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int k = lane + 64 * j;
        if (k < LP) pl.lab[(size_t)q * LP + k] = make_uint2(fq[j], __float_as_uint(k < pn ? lcosts[(size_t)q * LP + k] : DFLOW_FILL_COST));
    }
    // a chain start has no transition into it: empty first blocks (the chain kernel reads all LP of them)
    if (start0 || start1) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int k = lane + 64 * j;
            if (k < LP) {
                if (start0) pl.blkA[((size_t)q * 2 + 0) * LP + k] = make_uint2(BLK_EMPTY_X, BLK_EMPTY_Y);
                if (start1) pl.blkA[((size_t)q * 2 + 1) * LP + k] = make_uint2(BLK_EMPTY_X, BLK_EMPTY_Y);
            }
        }
    }
    if (!v0 && !v1) return;
    // lanes = labels of the successors (five groups of 64), q's labels come one by one as wave-uniform scalars (scalar
    // loads of q's row): D = sad + (2^31 - tpsi) has bit 31 set iff the pair is NOT compatible, v_alignbit shifts that bit
    // into the row word.  Columns run downwards inside each 32-bit word so that column c ends up in bit c; columns >= pn
    // (fill values) are forced to "not compatible" afterwards.
    const uint32_t kbias = 0x80000000u - (uint32_t)tpsi;
    const uint32_t *__restrict__ colp = proposals + (size_t)q * LP;
    uint32_t m[5][BCD_MASK_WORDS];
    // pixels with at most 128 labels (44 % of the frame: fewer than 25 window cells) have no labels in the fifth group
    const bool fifth = LP > 128 && (tn0 > 128 || tn1 > 128);
    auto build = [&](auto with5) {
#pragma unroll
        for (int j = 0; j < BCD_MASK_WORDS; j++) {
#pragma unroll
            for (int g = 0; g < 5; g++) m[g][j] = 0xFFFFFFFFu;
            if (32 * j < pn) {
#pragma unroll
                for (int cc = 31; cc >= 0; cc--) {
                    const uint32_t col = flow_bias(colp[32 * j + cc]);
#pragma unroll
                    for (int g = 0; g < 4; g++) m[g][j] = __builtin_amdgcn_alignbit(m[g][j], __builtin_amdgcn_sad_u16(col, f[g], kbias), 31);
                    if (decltype(with5)::value) m[4][j] = __builtin_amdgcn_alignbit(m[4][j], __builtin_amdgcn_sad_u16(col, f[4], kbias), 31);
                }
                const int nv = pn - 32 * j;                  // valid columns in this word (wave-uniform)
                const uint32_t inval = nv >= 32 ? 0u : ~0u << nv;
#pragma unroll
                for (int g = 0; g < 5; g++) m[g][j] |= inval;
            }
        }
    };
    if (fifth) build(BcdC<1>()); else build(BcdC<0>());

    // One group of 64 labels: member lists, blocks, pool rows.  MERGED = the fifth group (two pixels, 32 labels each): label,
    // count and output rows are per lane, ranks are taken inside each half.
    auto emit = [&](auto mergedc, const uint32_t (&mg)[BCD_MASK_WORDS], const uint32_t me, const int ig, const int tl, const int tn,
                    const bool valid, const size_t pd) {
        constexpr bool MERGED = decltype(mergedc)::value != 0;
        // the first 15 members as bytes and their pairwise costs as nibbles.  Labels beyond the pixel's count get an empty
        // first block: the chain kernel reads all LP of them.
        uint32_t w[BCD_MASK_WORDS];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < BCD_MASK_WORDS; j++) { w[j] = valid && tl < tn ? ~mg[j] : 0u; cnt += __popc(w[j]); }
        // the first 15 members as bytes: every lane appends to its 16-byte row in LDS (one ds_write_b8 and a pointer
        // increment per member; a shift register in VGPRs costs four instructions per member of the wave's LONGEST list) and
        // reads the row back in place
        uint8_t *row = &s_list[wv][lane][0];
        *reinterpret_cast<uint4 *>(row) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        const int n = min(cnt, BCD_LIST);
        const bool any = __ballot(cnt > 0) != 0ull;           // wave-uniform
        if (any) {
            // rows with more than BCD_LIST members (0.2 % of the labels) are cut back to their first BCD_LIST members, so
            // that the loops below need no capacity test (9 -> 7 instructions per member of the wave's longest list)
            uint32_t wt[BCD_MASK_WORDS];
#pragma unroll
            for (int j = 0; j < BCD_MASK_WORDS; j++) wt[j] = w[j];
            if (__ballot(cnt > BCD_LIST)) {
                int extra = cnt - BCD_LIST;
#pragma unroll
                for (int j = BCD_MASK_WORDS - 1; j >= 0; j--)
                    while (extra > 0 && wt[j]) { wt[j] &= ~(0x80000000u >> __clz(wt[j])); extra--; }
            }
            uint8_t *ap = row;
#pragma unroll
            for (int j = 0; j < BCD_MASK_WORDS; j++) {
                uint32_t ww = wt[j];
                while (ww) { *ap++ = (uint8_t)(32 * j + __ffs(ww) - 1); ww &= ww - 1; }
            }
        }
        const uint4 lst = *reinterpret_cast<const uint4 *>(row);
        const uint32_t l0 = lst.x, l1 = lst.y, l2 = lst.z, l3 = lst.w;
        // the members' pair costs, all at once (4 instructions per slot; in the loop they cost 5 per iteration of the
        // longest list): |dy-dy'| + |dx-dx'| against the member's flow; empty slots (0xFF) read a valid word and are zeroed
        uint32_t c0 = 0u, c1 = 0u;
        if (any) {
            auto slots = [&](auto lo, auto hi) {
#pragma unroll
                for (int e = decltype(lo)::value; e < decltype(hi)::value; e++) {
                    const uint32_t word = e < 4 ? l0 : (e < 8 ? l1 : (e < 12 ? l2 : l3));
                    const uint32_t k = (word >> (8 * (e & 3))) & 0xFFu;
                    const uint32_t psi = e < n ? flow_l1_biased(me, s_cols[wv][min(k, 191u)]) : 0u;    // k < 160, or 0xFF (empty slot) -> 191: inside the row
                    if (e < 8) c0 |= psi << (4 * e); else c1 |= psi << (4 * (e - 8));
                }
            };
            // slots 5..9 / 10..14 only if some label of the wave has that many members (wave-uniform: 97 % / 25 % of the waves)
            slots(BcdC<0>(), BcdC<BCD_BLK>());
            if (__ballot(n > BCD_BLK)) {
                slots(BcdC<BCD_BLK>(), BcdC<2 * BCD_BLK>());
                if (__ballot(n > 2 * BCD_BLK)) slots(BcdC<2 * BCD_BLK>(), BcdC<BCD_LIST>());
            }
        }
        // bytes 0..3 | 4, 5..8 | 9, 10..13 | 14 and nibbles 0..4, 5..9, 10..14 -> the three blocks
        const unsigned long long pp = ((unsigned long long)c1 << 32) | c0;
        // second / third blocks: compacted per 64-label wave of a pixel in lane order (rank = number of lower lanes of the
        // same pixel that own one); the first BCD_CAP_B / BCD_CAP_C owners get a slot
        const unsigned long long hasB = __ballot(cnt > BCD_BLK), hasC = __ballot(cnt > 2 * BCD_BLK);
        uint32_t rankB = __builtin_amdgcn_mbcnt_hi((uint32_t)(hasB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hasB, 0u));
        uint32_t rankC = __builtin_amdgcn_mbcnt_hi((uint32_t)(hasC >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hasC, 0u));
        if (MERGED && half) { rankB -= (uint32_t)__popc((uint32_t)hasB); rankC -= (uint32_t)__popc((uint32_t)hasC); }
        const uint32_t capB = ig < 2 ? BCD_CAP_B01 : BCD_CAP_B2, capC = ig < 2 ? BCD_CAP_C01 : BCD_CAP_C2;
        const bool storeB = cnt > BCD_BLK && rankB < capB, storeC = cnt > 2 * BCD_BLK && storeB && rankC < capC;
        const bool more = cnt > BCD_LIST || (cnt > BCD_BLK && !storeB) || (cnt > 2 * BCD_BLK && !storeC);
        const uint32_t ax = l0;
        const uint32_t ay = (l1 & 0xFFu) | (((uint32_t)pp & 0xFFFFFu) << 8) | (more ? BLK_MORE : 0u) | (storeB ? BLK_HAS_B : 0u) |
                            (storeC ? BLK_HAS_C : 0u) | ((uint32_t)(more ? 15 : min(cnt, 15)) << 28);     // "more" reports the largest count: the
                                                                                                       // chain kernel's block tests then lead to its row
        const uint32_t bx = __builtin_amdgcn_alignbit(l2, l1, 8);
        const uint32_t by = ((l2 >> 8) & 0xFFu) | (((uint32_t)(pp >> 20) & 0xFFFFFu) << 8);
        const uint32_t cx = __builtin_amdgcn_alignbit(l3, l2, 16);
        const uint32_t cy = ((l3 >> 16) & 0xFFu) | (((uint32_t)(pp >> 40) & 0xFFFFFu) << 8);
        if (valid && tl < LP) pl.blkA[pd * (size_t)LP + tl] = make_uint2(ax, ay);
        if (storeB) pl.blkBC[pd * BCD_ROW_BC + ig * BCD_CAP_B01 + rankB] = make_uint2(bx, by);
        if (storeC) pl.blkBC[pd * BCD_ROW_BC + BCD_ROW_B + ig * BCD_CAP_C01 + rankC] = make_uint2(cx, cy);
        // the 160-bit rows of the labels marked "more": one allocation per wave from the pool, in lane order (of each pixel)
        const unsigned long long dense = __ballot(more);
        if (dense) {                                         // wave-uniform
            uint32_t base = 0u;
            if (lane == 0) base = atomicAdd(pl.cursor, (uint32_t)__popcll(dense));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dense >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dense, 0u));
            if (MERGED) {
                // the upper half's rows follow the lower half's; every pixel's base goes to its own row
                const uint32_t nlo = (uint32_t)__popc((uint32_t)dense);
                if ((lane & 31) == 0 && (half ? (dense >> 32) != 0ull : (uint32_t)dense != 0u))
                    reinterpret_cast<uint32_t *>(pl.blkBC + pd * BCD_ROW_BC + BCD_ROW_B + BCD_ROW_C)[ig] = base + (half ? nlo : 0u);
            } else if (lane == 0) {
                reinterpret_cast<uint32_t *>(pl.blkBC + pd * BCD_ROW_BC + BCD_ROW_B + BCD_ROW_C)[ig] = base;
            }
            const uint32_t slot = base + rank;                // the rank inside a pixel's labels = rank - (rows of the lower half)
            if (more && slot < pl.pool_rows) {
                uint32_t *out = pl.masks + (size_t)slot * BCD_MASK_WORDS;
#pragma unroll
                for (int j = 0; j < BCD_MASK_WORDS; j++) out[j] = w[j];
            }
        }
    };
    const size_t pd0 = (size_t)p0 * 2 + 0, pd1 = (size_t)p1 * 2 + 1;
    if (v0) {
        emit(BcdC<0>(), m[0], f[0], 0, lane, tn0, true, pd0);
        if (LP > 64) emit(BcdC<0>(), m[1], f[1], 1, 64 + lane, tn0, true, pd0);
    }
    if (v1) {
        emit(BcdC<0>(), m[2], f[2], 0, lane, tn1, true, pd1);
        if (LP > 64) emit(BcdC<0>(), m[3], f[3], 1, 64 + lane, tn1, true, pd1);
    }
    if (LP > 128) emit(BcdC<1>(), m[4], f[4], 2, 128 + (lane & 31), half ? tn1 : tn0, half ? v1 : v0, half ? pd1 : pd0);
}

// ------------------------------------------------------------------------------------------------ chains
#ifndef BCD_MAX_BATCH
#define BCD_MAX_BATCH 8
#endif
struct BcdPass {
    const int32_t *nprop;
    int32_t *bestlabels;
    BcdPlanes pl;
};
struct BcdArgs {
    int H, W, LP, tpsi, phase;
    double lamda;
    BcdPass pass[BCD_MAX_BATCH];
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

__device__ static inline uint32_t lane_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// One thread per label: 192 threads (3 waves) cover up to DFLOW_MAX_LABELS labels.
#define BCD_BACK_PITCH BCD_THREADS       // back-pointer rows: one byte per THREAD (no special case for the shadow lanes)

typedef const __attribute__((address_space(1))) char *gptr_t;      // a pointer known to be global memory (not flat)
__device__ static inline gptr_t uniform_ptr(const void *p)
{
    // tells the compiler that the pointer is wave-uniform (it is: kernel arguments and block indices only) and global, so
    // that the loads below use the scalar-base + 32-bit lane offset form and the row pointers advance with scalar adds
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr_t)(((unsigned long long)hi << 32) | lo);
}
__device__ static inline unsigned long long ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }

__global__ void __launch_bounds__(BCD_THREADS, BCD_MINWAVES) bcd_chain_kernel(BcdArgs a)
{
    // static LDS has compile-time addresses, so the offsets fold into the ds_read immediates
    __shared__ double s_dp[2 * BCD_LDS_LABELS];                               // [2][labels]; entries >= 160 stay +inf (list sentinel 0xFF)
    __shared__ unsigned long long permv[2 * 4];                               // per-wave minima of bits(tpsi + dp)
    __shared__ int permi[2 * 4 + 4];                                          // their labels; [8] = traceback hand-over
    __shared__ __attribute__((aligned(16))) uint8_t tb[BCD_TB_STEPS * BCD_BACK_PITCH];   // traceback chunk
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    uint32_t *bestf = s_dyn + 1;                                      // [-1..len] biased flow of each chain pixel's current label
    int *tnl;                                                         // [len] nprop of each chain pixel (set below)
    int *s_label = permi + 8;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chain = blockIdx.x;
    const int W = a.W, LP = a.LP;
    const bool owner = tid < LP;                       // threads LP..191 shadow the last label row; nothing refers to what they compute
    const unsigned long long ownmask = ballot64(owner);
    const int tl = owner ? tid : LP - 1;
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
    const double lamda = a.lamda;
    // this pass's planes (copied out of the argument block once: nothing below reloads kernel arguments)
    const BcdPass &ps = a.pass[blockIdx.y];
    const int32_t *g_nprop = ps.nprop;
    int32_t *g_best = ps.bestlabels;
    const uint2 *g_lab = ps.pl.lab;
    const uint32_t *g_masks = ps.pl.masks;
    const uint2 *g_bc = ps.pl.blkBC;
    const uint32_t pool_rows = ps.pl.pool_rows;
    uint8_t *back = ps.pl.back + ((size_t)chain * len) * BCD_BACK_PITCH;

    tnl = (int *)(bestf + len + 2);
    for (int i = tid; i < 2 * BCD_LDS_LABELS; i += BCD_THREADS) s_dp[i] = 1e300;
    for (int i = tid; i < len; i += BCD_THREADS) {
        int pix = pix0 + i * pstep;
        bestf[i] = g_lab[(size_t)pix * LP + g_best[pix]].x;
        tnl[i] = g_nprop[pix];      // a wave-uniform global load inside the step loop would stall every step (it is
                                    // moved to an SGPR at once); the label counts are read from LDS instead
    }
    if (tid == 0) { bestf[-1] = 0u; bestf[len] = 0u; bestf[len + 1] = 0u; }       // read, never used (the limit of that side term is 0)
    __syncthreads();

    // Per-step inputs of this thread: its label's first block and label data, prefetched four steps ahead into four
    // static register slots, and its second / third block (if it owns one: the first block's count says so), fetched two
    // steps ahead.  Rows are LP wide and every first block is initialised, so the first-block and label loads are
    // unconditional; nothing in a step waits for a load issued in the same step.  The row pointers are wave-uniform and
    // advance by a constant stride per step (scalar adds); they stop at the chain's last pixel.
    const uint32_t offl = (uint32_t)tl * 8u;
    const long long rowl = (long long)LP * 8, rowg = (long long)BCD_ROW_BC * 8;
    const long long strA = (long long)pstep * 2 * rowl, strL = (long long)pstep * rowl, strG = (long long)pstep * 2 * rowg;
    gptr_t pA = uniform_ptr((const char *)ps.pl.blkA + ((long long)pix0 * 2 + dir) * rowl);      // row of step `nextA`
    gptr_t pL = uniform_ptr((const char *)ps.pl.lab + (long long)pix0 * rowl);
    // this wave's second blocks of step `nextG`; its third blocks lie coff bytes further
    gptr_t pB = uniform_ptr((const char *)ps.pl.blkBC + ((long long)pix0 * 2 + dir) * rowg + (long long)wave * (BCD_CAP_B01 * 8));
    const uint32_t coff = (uint32_t)(BCD_ROW_B * 8 + wave * (BCD_CAP_C01 * 8) - wave * (BCD_CAP_B01 * 8));
    int nextA = 0, nextG = 0;
    auto advA = [&]() { const bool m = nextA + 1 < len; pA += m ? strA : 0; pL += m ? strL : 0; nextA++; };
    auto advG = [&]() { const bool m = nextG + 1 < len; pB += m ? strG : 0; nextG++; };
    auto ld8 = [](gptr_t base, uint32_t off) {
        const unsigned long long v = *reinterpret_cast<const __attribute__((address_space(1))) unsigned long long *>(base + off);
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    };
    // second / third block of the step the (pB, pC) row pointers stand at, given that step's first block: lanes that own
    // none keep the empty block
    auto fetchBC = [&](const uint2 &blkA, uint2 &B, uint2 &C) {
        // the blocks that were stored are a prefix (in lane order) of the labels that own one: the rank among the stored
        // ones is the slot
        const bool hb = owner && (blkA.y & BLK_HAS_B) != 0u, hc = owner && (blkA.y & BLK_HAS_C) != 0u;
        const unsigned long long mb = ballot64(hb);
        B = make_uint2(BLK_EMPTY_X, BLK_EMPTY_Y); C = make_uint2(BLK_EMPTY_X, BLK_EMPTY_Y);
        if (mb) {                                                     // wave-uniform
            if (hb) B = ld8(pB, lane_rank(mb) * 8u);
            const unsigned long long mc = ballot64(hc);
            if (mc) { if (hc) C = ld8(pB, coff + lane_rank(mc) * 8u); }
        }
        advG();
    };
    uint2 A0, L0 = ld8(pL, offl); advA();                      // step 0 has no transition: only its label data
    uint2 A1 = ld8(pA, offl), L1 = ld8(pL, offl); advA();
    uint2 A2 = ld8(pA, offl), L2 = ld8(pL, offl); advA();
    uint2 A3 = ld8(pA, offl), L3 = ld8(pL, offl); advA();
    uint2 B1, C1, B0, C0;
    advG();
    fetchBC(A1, B1, C1);
    fetchBC(A2, B0, C0);

    // ---- chain start: dp[0,tl] = (s1 + s2) + lamda*lcost   (python bcd.py:118-120)
    {
        const uint32_t Fc = L0.x;
        const int ip = dirp, im = -dirp;
        const uint32_t s1 = (ip >= 0 && ip < len) ? min(tpsi, flow_l1_biased(Fc, bestf[ip])) : 0u;
        const uint32_t s2 = (im >= 0 && im < len) ? min(tpsi, flow_l1_biased(Fc, bestf[im])) : 0u;
        unsigned long long key = ~0ull;
        if (owner && tl < tnl[0]) {
            const double d0 = __dadd_rn((double)(s1 + s2), __dmul_rn(lamda, (double)__uint_as_float(L0.y)));
            s_dp[tl] = d0;
            key = (unsigned long long)__double_as_longlong(__dadd_rn(tpsi_d, d0));
        }
        int fl;
        const unsigned long long m = wave_key_min_asm(key, &fl);
        if (lane == 0) { permv[wave] = m; permi[wave] = wave * 64 + fl; }
    }
    A0 = ld8(pA, offl); L0 = ld8(pL, offl); advA();
    __syncthreads();

    int pn = tnl[0];
    __attribute__((address_space(1))) uint8_t *backrow = (__attribute__((address_space(1))) uint8_t *)uniform_ptr(back + BCD_BACK_PITCH);   // row of step 1; + BCD_BACK_PITCH per step (scalar)
    // One step of the chain; CUR (compile-time) is the LDS buffer this step writes, CUR^1 holds the previous pixel.
    // (inA, inL) = this step's first block and label data, refilled with those of step i+4; (inB, inC) = this step's
    // second / third blocks, refilled with those of step i+2, whose first block `nxA` arrived two steps ago.  The slots
    // are used round-robin by the 4x unrolled loop below, so prefetched registers are never copied while their loads
    // are still in flight.
    // the per-step scalars (label count, flows of the two side neighbours) never change during the kernel: those of step
    // i+1 are read from LDS during step i
    int tn_nx = tnl[1];
    uint32_t bfp_nx = bestf[1 + dirp], bfm_nx = bestf[1 - dirp];
    auto step = [&](auto curc, auto edgec, const int i, uint2 &inA, uint2 &inL, uint2 &inB, uint2 &inC, const uint2 &nxA) __attribute__((always_inline)) {
        constexpr int CUR = decltype(curc)::value;
        constexpr bool EDGE = decltype(edgec)::value != 0;     // the chain's last steps: a side neighbour may lie outside
        const char *prev = reinterpret_cast<const char *>(s_dp + (CUR ^ 1) * BCD_LDS_LABELS);
        const int tn = tn_nx;
        const uint32_t bfp = bfp_nx, bfm = bfm_nx;
        tn_nx = tnl[min(i + 1, len - 1)];
        bfp_nx = bestf[i + 1 + dirp]; bfm_nx = bestf[i + 1 - dirp];     // bestf[-1..len+1] exists (see the allocation)
        const uint32_t ax = inA.x, ay = inA.y, bx = inB.x, by = inB.y, cx = inC.x, cy = inC.y;
        const uint32_t Fc = inL.x;
        const float lc = __uint_as_float(inL.y);
        const unsigned long long actmask = ballot64(tl < tn) & ownmask;
        const bool act = owner && tl < tn;
        // min over compatible previous labels (python bcd.py:163-176 / :198-219) in increasing k (strict '<' keeps the
        // first minimum).  Members come as index bytes in increasing k (0xFF = none, which reads the +inf tail of dp)
        // with their pair costs; candidates are tracked by their dp offset 8 k.  Their dp values are requested FIRST (the
        // block has been in registers for four steps): the prefetches and the block bookkeeping below run while LDS answers.
        uint32_t ad[BCD_BLK]; double dd[BCD_BLK];
#pragma unroll
        for (int j = 0; j < BCD_BLK; j++) {
            ad[j] = ((j < 4 ? ax >> (8 * j) : ay) & 0xFFu) << 3;
            dd[j] = *reinterpret_cast<const double *>(prev + ad[j]);
        }
        asm volatile("" ::: "memory");              // keeps the LDS reads in front of what follows
        inA = ld8(pA, offl); inL = ld8(pL, offl); advA();
        fetchBC(nxA, inB, inC);
        // permmincost / permminlabel (python bcd.py:152-157) merged from the per-wave partials of the previous step (waves
        // are in label order and every partial index is the first one inside its wave), and the unary term
        // small = (lamda*lcost + s1) + s2 (python bcd.py:161-162; a side neighbour outside the chain contributes 0: its
        // limit is the scalar 0)
        Cand perm;
        double small;
        {
            const unsigned long long p0 = permv[(CUR ^ 1) * 4], p1 = permv[(CUR ^ 1) * 4 + 1], p2 = permv[(CUR ^ 1) * 4 + 2];
            const int i0 = permi[(CUR ^ 1) * 4], i1 = permi[(CUR ^ 1) * 4 + 1], i2 = permi[(CUR ^ 1) * 4 + 2];
            uint32_t lim1 = tpsi, lim2 = tpsi;
            if (EDGE) {
                const int ip = i + dirp, im = i - dirp;
                lim1 = (ip >= 0 && ip < len) ? tpsi : 0u; lim2 = (im >= 0 && im < len) ? tpsi : 0u;
            }
            const uint32_t s1 = min(flow_l1_biased(Fc, bfp), lim1);
            const uint32_t s2 = min(flow_l1_biased(Fc, bfm), lim2);
            unsigned long long pmn = p0; int pix_ = i0;
            if (p1 < pmn) { pmn = p1; pix_ = i1; }
            if (p2 < pmn) { pmn = p2; pix_ = i2; }
            perm.v = __longlong_as_double((long long)pmn); perm.k = pix_;
            small = __dadd_rn(__dadd_rn(__dmul_rn(lamda, (double)lc), (double)s1), (double)s2);
        }
        // first minimum of a block as a tree (depth 3 instead of a chain of 5 dependent compare / select steps): the members
        // are in increasing k, so "the right operand wins only if strictly smaller" keeps the first minimum at every node
        auto node = [](double &va, uint32_t &ka, const double vb, const uint32_t kb) {
            const bool t = vb < va;
            va = __builtin_fmin(va, vb); ka = t ? kb : ka;
        };
        auto block_min = [&](double (&d)[BCD_BLK], uint32_t (&k)[BCD_BLK], const uint32_t y, double &v, uint32_t &kk) {
#pragma unroll
            for (int j = 0; j < BCD_BLK; j++) d[j] = __dadd_rn(d[j], (double)((y >> (8 + 4 * j)) & 7u));    // +inf + psi = +inf
            node(d[0], k[0], d[1], k[1]); node(d[2], k[2], d[3], k[3]);
            node(d[0], k[0], d[2], k[2]); node(d[0], k[0], d[4], k[4]);
            v = d[0]; kk = k[0];
        };
        double bestv; uint32_t besta;
        block_min(dd, ad, ay, bestv, besta);
        // second block: some label of the wave has more than 5 members (almost always true for a full wave)
        if ((ballot64((ay >> 28) > BCD_BLK) & actmask) != 0ull) {
            uint32_t a2[BCD_BLK]; double d2[BCD_BLK];
#pragma unroll
            for (int j = 0; j < BCD_BLK; j++) { a2[j] = ((j < 4 ? bx >> (8 * j) : by) & 0xFFu) << 3; d2[j] = *reinterpret_cast<const double *>(prev + a2[j]); }
            { double vb; uint32_t kb; block_min(d2, a2, by, vb, kb); node(bestv, besta, vb, kb); }
            // third block (a quarter of the waves), labels marked "more" (2 % of the workgroup steps; their count field says 15)
            if (__builtin_expect((ballot64((ay >> 28) > 2 * BCD_BLK) & actmask) != 0ull, 0)) {
#pragma unroll
                for (int j = 0; j < BCD_BLK; j++) { a2[j] = ((j < 4 ? cx >> (8 * j) : cy) & 0xFFu) << 3; d2[j] = *reinterpret_cast<const double *>(prev + a2[j]); }
                { double vc; uint32_t kc; block_min(d2, a2, cy, vc, kc); node(bestv, besta, vc, kc); }
                const bool more = act && (ay & BLK_MORE) != 0u;
                const unsigned long long moremask = ballot64(more);
                if (__builtin_expect(moremask != 0ull, 0)) {
                    // members beyond the blocks this label was given (more than 15, or a second / third block that found no
                    // slot): the label walks what is left of its 160-bit row behind the last member it has seen, four set
                    // bits per round; still increasing k, so strict '<' stands.  The predecessor's flows come from its label
                    // data (this path is rare; everything it reads is L2-resident).  A row that found no place in the pool is
                    // rebuilt here from the predecessor's flows (the predicate of bcd_lists_kernel, one label at a time).
                    const int klast = (int)(((ay & BLK_HAS_C) ? cy : (ay & BLK_HAS_B) ? by : ay) & 0xFFu);
                    const size_t cpix = (size_t)(pix0 + i * pstep), ppix = (size_t)(pix0 + (i - 1) * pstep);
                    const uint2 *plab = g_lab + ppix * LP;
                    unsigned long long w0 = 0, w1 = 0, w2 = 0;
                    if (more) {
                        const uint32_t mbase = reinterpret_cast<const uint32_t *>(g_bc + (cpix * 2 + dir) * BCD_ROW_BC + BCD_ROW_B + BCD_ROW_C)[wave];
                        const uint32_t slot = mbase + lane_rank(moremask);
                        if (slot < pool_rows) {
                            const uint32_t *mrow = g_masks + (size_t)slot * BCD_MASK_WORDS;
                            w0 = (unsigned long long)mrow[0] | ((unsigned long long)mrow[1] << 32);
                            w1 = (unsigned long long)mrow[2] | ((unsigned long long)mrow[3] << 32);
                            w2 = (unsigned long long)mrow[4];
                        } else {
                            for (int k = klast + 1; k < pn; k++) {
                                const unsigned long long bit = flow_l1_biased(Fc, plab[k].x) < tpsi ? 1ull << (k & 63) : 0ull;
                                if (k < 64) w0 |= bit; else if (k < 128) w1 |= bit; else w2 |= bit;
                            }
                        }
                        const int b = klast & 63;
                        const unsigned long long keep = b == 63 ? 0ull : (~0ull << (b + 1));
                        if (klast < 64) w0 &= keep; else if (klast < 128) { w0 = 0; w1 &= keep; } else { w0 = 0; w1 = 0; w2 &= keep; }
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
                        for (int j = 0; j < 4; j++) { d4[j] = *reinterpret_cast<const double *>(prev + 8 * kk[j]); ff[j] = plab[kk[j]].x; }   // invalid slots read label 0: harmless
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
        // No branch on `act`: the slots of labels beyond this pixel's count (dp, back-pointer) may hold anything, no
        // list and no traceback refers to them; the shadow lanes (tid >= LP) write to LDS slots LP..191 and to
        // back-pointer columns LP..191, equally unreferenced.
        unsigned long long key;
        {
            const bool found = bestv < 1e300;        // no compatible predecessor: every slot read the +inf tail
            const double mincost = found ? bestv : perm.v;
            const int pl_ = found ? (int)(besta >> 3) : perm.k;
            const double dpc = __dadd_rn(mincost, small);
            s_dp[CUR * BCD_LDS_LABELS + tid] = dpc;
            backrow[tid] = (uint8_t)pl_;
            backrow += BCD_BACK_PITCH;
            key = act ? (unsigned long long)__double_as_longlong(__dadd_rn(tpsi_d, dpc)) : ~0ull;
        }
        {
            int fl;
            const unsigned long long m = wave_key_min_asm(key, &fl);
            if (lane == 0) { permv[CUR * 4 + wave] = m; permi[CUR * 4 + wave] = wave * 64 + fl; }
        }
        // LDS-only barrier: __syncthreads() would also wait for the global prefetches issued in this step (vmcnt(0)) and
        // put their full latency on every step of the chain
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        pn = tn;
    };
    // step i: first block / label slot i mod 4, second / third block slot i mod 2, next-next first block = slot (i+2) mod 4
    int i = 1;
    for (; i + 4 < len; i += 4) {                       // whole groups of four interior steps (both side neighbours on the chain)
        step(IntC<1>(), IntC<0>(), i, A1, L1, B1, C1, A3);
        step(IntC<0>(), IntC<0>(), i + 1, A2, L2, B0, C0, A0);
        step(IntC<1>(), IntC<0>(), i + 2, A3, L3, B1, C1, A1);
        step(IntC<0>(), IntC<0>(), i + 3, A0, L0, B0, C0, A2);
    }
    if (i < len) step(IntC<1>(), IntC<1>(), i, A1, L1, B1, C1, A3);
    if (i + 1 < len) step(IntC<0>(), IntC<1>(), i + 1, A2, L2, B0, C0, A0);
    if (i + 2 < len) step(IntC<1>(), IntC<1>(), i + 2, A3, L3, B1, C1, A1);
    if (i + 3 < len) step(IntC<0>(), IntC<1>(), i + 3, A0, L0, B0, C0, A2);
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
    int plb = *s_label;
    if (tid == 0) g_best[pix0 + (len - 1) * pstep] = plb;
    for (int hi = len - 1; hi >= 1; hi -= BCD_TB_STEPS) {
        const int lo = max(1, hi - BCD_TB_STEPS + 1);      // steps lo..hi
        const int nbytes = (hi - lo + 1) * BCD_BACK_PITCH;
        const uint4 *src = reinterpret_cast<const uint4 *>(back + (size_t)lo * BCD_BACK_PITCH);
        for (int j = tid; j < nbytes / 16; j += BCD_THREADS) reinterpret_cast<uint4 *>(tb)[j] = src[j];
        __syncthreads();
        if (tid == 0) {
            for (int i2 = hi; i2 >= lo; i2--) {
                plb = tb[(i2 - lo) * BCD_BACK_PITCH + plb];
                g_best[pix0 + (i2 - 1) * pstep] = plb;
            }
            *s_label = plb;
        }
        __syncthreads();
        plb = *s_label;
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

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t back_bytes(const dflow_params *p)
{
    size_t m = 0;
    for (int ph = 0; ph < 4; ph++) {
        int n, len;
        phase_dims(p, ph, n, len);
        size_t b = (size_t)n * len * BCD_BACK_PITCH;
        if (b > m) m = b;
    }
    return align256(m) + 256;
}

static size_t lab_bytes(const dflow_params *p) { return align256((size_t)p->pich * p->picw * p->label_pitch * 8); }
static size_t blka_bytes(const dflow_params *p) { return align256((size_t)p->pich * p->picw * 2 * p->label_pitch * 8); }
static size_t blkbc_bytes(const dflow_params *p) { return align256((size_t)p->pich * p->picw * 2 * BCD_ROW_BC * 8); }
static uint32_t pool_rows_of(const dflow_params *p)
{
    const size_t rows = (size_t)p->pich * p->picw * 2 * p->label_pitch / BCD_POOL_DIV;
    return (uint32_t)(rows < 4096 ? 4096 : rows);
}
static size_t mask_bytes(const dflow_params *p) { return align256((size_t)pool_rows_of(p) * BCD_MASK_WORDS * sizeof(uint32_t)); }

size_t bcd_ws_bytes(const dflow_params *p)
{
    return back_bytes(p) + lab_bytes(p) + blka_bytes(p) + blkbc_bytes(p) + mask_bytes(p) + 256;
}

static BcdPlanes planes_of(const dflow_params *p, void *ws)
{
    BcdPlanes pl;
    char *w = (char *)ws;
    pl.back = (uint8_t *)w; w += back_bytes(p);
    pl.lab = (uint2 *)w; w += lab_bytes(p);
    pl.blkA = (uint2 *)w; w += blka_bytes(p);
    pl.blkBC = (uint2 *)w; w += blkbc_bytes(p);
    pl.masks = (uint32_t *)w; w += mask_bytes(p);
    pl.cursor = (uint32_t *)w;
    pl.pool_rows = pool_rows_of(p);
    return pl;
}

int launch_bcd_prepare(const dflow_params *p, const uint32_t *proposals, const float *lcosts, const int32_t *nprop, void *ws,
                       hipStream_t s)
{
    long long items = (long long)p->pich * p->picw;          // one wave per pixel (as the predecessor of two others)
    if (hipMemsetAsync(planes_of(p, ws).cursor, 0, 256, s) != hipSuccess)
        return dflow_set_error(DFLOW_EHIP, "hipMemsetAsync failed in launch_bcd_prepare");
    hipLaunchKernelGGL(bcd_lists_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p->pich, p->picw, p->label_pitch,
                       p->tpsi, proposals, lcosts, nprop, planes_of(p, ws));
    return dflow_check_launch("bcd_lists_kernel");
}

int launch_bcd_phase_batch(const dflow_params *p, int npass, const int32_t *const *nprop, int32_t *const *bestlabels, int phase,
                           void *const *ws, hipStream_t s)
{
    int nchains, len;
    phase_dims(p, phase, nchains, len);
    if (nchains == 0) return DFLOW_OK;
    size_t shmem = (size_t)(len + 4) * sizeof(uint32_t) + (size_t)len * sizeof(int);
    for (int b0 = 0; b0 < npass; b0 += BCD_MAX_BATCH) {
        const int nb = npass - b0 < BCD_MAX_BATCH ? npass - b0 : BCD_MAX_BATCH;
        BcdArgs a;
        a.H = p->pich; a.W = p->picw; a.LP = p->label_pitch; a.tpsi = p->tpsi; a.phase = phase; a.lamda = p->lamda;
        for (int b = 0; b < BCD_MAX_BATCH; b++) {
            const int src = b0 + (b < nb ? b : 0);
            a.pass[b].nprop = nprop[src]; a.pass[b].bestlabels = bestlabels[src]; a.pass[b].pl = planes_of(p, ws[src]);
        }
        hipLaunchKernelGGL(bcd_chain_kernel, dim3(nchains, nb), dim3(BCD_THREADS), shmem, s, a);
        int rc = dflow_check_launch("bcd_chain_kernel");
        if (rc) return rc;
    }
    return DFLOW_OK;
}

int launch_bcd_phase(const dflow_params *p, const uint32_t *proposals, const int32_t *nprop, int32_t *bestlabels, int phase,
                     void *ws, hipStream_t s)
{
    (void)proposals;
    void *wsv = ws;
    return launch_bcd_phase_batch(p, 1, &nprop, &bestlabels, phase, &wsv, s);
}
