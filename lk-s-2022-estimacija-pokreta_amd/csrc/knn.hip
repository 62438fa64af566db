// K3/K4 knn_cells, exact VALU path: generisi, daisy i flann.py:157-189, with the FLANN search (:171-172) replaced by
// the build's canonical exact 5-NN: squared L2 as a sequential fmaf chain over k = 0..67, ties to the lower
// in-cell index, results in ascending (distance, index) order.
//
// This file holds (a) the brute-force kernel (DFLOW_FLAG_KNN_EXACT, and the fallback for geometries the MFMA path does
// not cover) and (b) the fix-up kernel that re-does, exactly, the few (query wave, candidate cell) pairs the MFMA
// path hands back (event-list overflow, or descriptors outside the f16 range).
//
// One thread = one image-1 pixel (query); its 68-float descriptor stays in VGPRs.  All lanes of a wave belong to
// ONE image-1 cell, share the same window of image-2 cells and walk the candidate points in lock-step: the
// candidate's address is wave-uniform, so it is fetched through the scalar cache (s_load_dwordx16) and used as
// SGPR operands -- no LDS traffic, no barriers.  Per (query, candidate): 68 x (v_sub_f32, v_fma_f32).
#include "dflow_common.h"

#define KNN_THREADS 256

struct KnnArgs {
    Geom g;
    int LP, chunks;
    float tphi;
};

struct Top5 {
    float d0, d1, d2, d3, d4;
    int i0, i1, i2, i3, i4;
};

// exact search of one candidate cell for the query held in q[]; gd2 must be a __restrict__ kernel argument
// squared L2 distance of the query to the candidate row c (wave-uniform address: scalar loads), sequential fmaf chain
template <typename T> __device__ static inline float canon_dist(const float (&q)[DFLOW_DESC], const T *__restrict__ c)
{
    float acc = 0.0f;
    if constexpr (sizeof(T) == 4) {
        const float4 *__restrict__ c4 = reinterpret_cast<const float4 *>(c);
#pragma unroll
        for (int k = 0; k < DFLOW_DESC / 4; k++) {
            float4 v = c4[k];
            float e;
            e = q[4 * k] - v.x; acc = __fmaf_rn(e, e, acc);
            e = q[4 * k + 1] - v.y; acc = __fmaf_rn(e, e, acc);
            e = q[4 * k + 2] - v.z; acc = __fmaf_rn(e, e, acc);
            e = q[4 * k + 3] - v.w; acc = __fmaf_rn(e, e, acc);
        }
    } else {
#pragma unroll
        for (int k = 0; k < DFLOW_DESC; k++) { const float e = q[k] - (float)c[k]; acc = __fmaf_rn(e, e, acc); }
    }
    return acc;
}

template <typename T>
__device__ static inline void search_cell_exact(const float (&q)[DFLOW_DESC], const T *__restrict__ gd2, const Geom &g,
                                                int ci, int cj, Top5 &t)
{
    constexpr int P = DescPitch<T>::value;
    const int cx0 = g.x0(ci), cx1 = g.x1(ci), cy0 = g.y0(cj), cy1 = g.y1(cj), ccw = cx1 - cx0;
    t.d0 = t.d1 = t.d2 = t.d3 = t.d4 = INFINITY;
    t.i0 = t.i1 = t.i2 = t.i3 = t.i4 = 0;
    for (int yy = cy0; yy < cy1; yy++) {
        const T *__restrict__ row = gd2 + ((size_t)yy * g.W + cx0) * P;
        for (int xx = 0; xx < ccw; xx++) {
            const float acc = canon_dist(q, row + (size_t)xx * P);   // wave-uniform address
            if (acc < t.d4) {
                // insert keeping ascending order; strict '<' leaves equal distances in index order; once the new
                // entry has found its place every later entry shifts down unconditionally (a displaced entry must
                // stay in front of an equal one that followed it)
                float cd = acc; int cidx = (yy - cy0) * ccw + xx; bool sh = false;
#define CSWAP(D, I) if (sh || cd < D) { float td = D; int ti = I; D = cd; I = cidx; cd = td; cidx = ti; sh = true; }
                CSWAP(t.d0, t.i0) CSWAP(t.d1, t.i1) CSWAP(t.d2, t.i2) CSWAP(t.d3, t.i3) CSWAP(t.d4, t.i4)
#undef CSWAP
            }
        }
    }
}

// daisy i flann.py:174-180: proposals [dy,dx] and truncated L1 costs of the 5 winners of one cell
template <typename T>
__device__ static inline void emit_cell(const Top5 &t, const Geom &g, int ci, int cj, size_t pix, int qy, int qx, int slot,
                                        int LP, float tphi, bool active, const T *__restrict__ gd1,
                                        const T *__restrict__ gd2, uint32_t *__restrict__ gproposals,
                                        float *__restrict__ glcosts, float *cost_out)
{
    const int cx0 = g.x0(ci), cy0 = g.y0(cj), ccw = g.x1(ci) - cx0;
    const int idx[5] = {t.i0, t.i1, t.i2, t.i3, t.i4};
#pragma unroll
    for (int qq = 0; qq < 5; qq++) {
        const int ty = cy0 + idx[qq] / ccw, tx = cx0 + idx[qq] % ccw;
        const float s = l1_cost_np(gd1 + pix * DescPitch<T>::value, gd2 + ((size_t)ty * g.W + tx) * DescPitch<T>::value);
        const float c = s < tphi ? s : tphi;   // python min(tphi, s)
        if (active) {
            gproposals[pix * LP + slot + qq] = pack_flow(ty - qy, tx - qx);
            glcosts[pix * LP + slot + qq] = c;
        }
        cost_out[qq] = c;
    }
}

// gd2 is only ever read and never aliases the outputs: with __restrict__ kernel arguments the compiler can prove it
// and turns the wave-uniform candidate loads into scalar loads.
template <typename T>
__global__ void __launch_bounds__(KNN_THREADS, 4) knn_exact_kernel(KnnArgs a, const T *__restrict__ gd1,
                                                                   const T *__restrict__ gd2,
                                                                   uint32_t *__restrict__ gproposals,
                                                                   float *__restrict__ glcosts,
                                                                   int32_t *__restrict__ gnprop,
                                                                   int32_t *__restrict__ gbestlabels)
{
    const Geom g = a.g;
    const int qcell = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
    const int qci = qcell % g.ncx, qcj = qcell / g.ncx;
    const int qx0 = g.x0(qci), qy0 = g.y0(qcj), qcw = g.x1(qci) - qx0, qnpts = qcw * (g.y1(qcj) - qy0);
    if (chunk * KNN_THREADS >= qnpts) return;
    int qi = chunk * KNN_THREADS + threadIdx.x;
    const bool active = qi < qnpts;
    if (!active) qi = qnpts - 1;
    const int qy = qy0 + qi / qcw, qx = qx0 + qi % qcw;
    const size_t pix = (size_t)qy * g.W + qx;
    float q[DFLOW_DESC];
    desc_load_row(q, gd1, pix);
    const int cimin = max(0, qci - g.win), cimax = min(g.ncx - 1, qci + g.win);
    const int cjmin = max(0, qcj - g.win), cjmax = min(g.ncy - 1, qcj + g.win);
    float mind = 1000.0f;   // mindists, daisy i flann.py:93
    int bestl = 0, slot = 0;
    for (int ci = cimin; ci <= cimax; ci++)          // ci outer, cj inner (Q2)
        for (int cj = cjmin; cj <= cjmax; cj++) {
            Top5 t;
            search_cell_exact(q, gd2, g, ci, cj, t);
            float c[5];
            emit_cell(t, g, ci, cj, pix, qy, qx, slot, a.LP, a.tphi, active, gd1, gd2, gproposals, glcosts, c);
#pragma unroll
            for (int qq = 0; qq < 5; qq++)
                if (c[qq] < mind) { mind = c[qq]; bestl = slot + qq; }   // WTA, strict '<' (:181-184)
            slot += 5;                                                     // nprop += 5 (:189)
        }
    if (active) {
        gnprop[pix] = slot;
        gbestlabels[pix] = bestl;
        for (int s = slot; s < a.LP; s++) { gproposals[pix * a.LP + s] = DFLOW_FILL_PROPOSAL; glcosts[pix * a.LP + s] = DFLOW_FILL_COST; }
    }
}

// Fix-up for the MFMA path.  Work item = (query cell, first query index, 64 queries, candidate cell).  Items come
// from the overflow list, or -- if the prep kernel flagged out-of-range descriptors or the list itself overflowed --
// every item of the pass is enumerated.  A workgroup of KNN_FIX_WAVES waves takes one item at a time (grid-stride): all
// waves hold the same 64 queries (lane = query); the candidate cell streams through LDS in chunks of KNN_FIX_CHUNK rows that
// all threads fetch together (one 16-byte piece per thread, the next chunk's loads in flight while this one is evaluated);
// wave w evaluates candidates w, w + 16 of every chunk (broadcast LDS reads), the partial top-5 lists meet in LDS and wave
// 0 merges them by (distance, index), so the result is that of one sequential scan.  A single overflowed list used to
// cost the latency of one wave walking a whole cell through the scalar cache (1.3 ms at 64x27 cells; 0.34 ms with the
// rows dealt to 16 waves: about 2 us per candidate, all of it load latency).
#define KNN_FIX_WAVES 16
#define KNN_FIX_CHUNK 64              // candidates per LDS chunk (four per wave)
#define KNN_FIX_BLOCKS 256            // one per CU: a launch that finds no item costs 4 096 wave starts (1 024 blocks: 16 384, which
                                  // delayed the stream by a millisecond when other kernels held the CUs)
__device__ static inline void top5_insert(Top5 &t, float cd, int cidx)
{
    bool sh = false;
#define CSWAP(D, I) if (sh || cd < D) { float td = D; int ti = I; D = cd; I = cidx; cd = td; cidx = ti; sh = true; }
    CSWAP(t.d0, t.i0) CSWAP(t.d1, t.i1) CSWAP(t.d2, t.i2) CSWAP(t.d3, t.i3) CSWAP(t.d4, t.i4)
#undef CSWAP
}

template <typename T>
__global__ void __launch_bounds__(64 * KNN_FIX_WAVES) knn_fix_kernel(KnnArgs a, const T *__restrict__ gd1, const T *__restrict__ gd2,
                                                        uint32_t *__restrict__ gproposals, float *__restrict__ glcosts,
                                                        const int *__restrict__ ovf_count, const int4 *__restrict__ ovf_list,
                                                        int ovf_cap, const int *__restrict__ flags, int qwaves)
{
    __shared__ float pd[KNN_FIX_WAVES][5][64];
    __shared__ int pi[KNN_FIX_WAVES][5][64];
    __shared__ __attribute__((aligned(16))) float cand[2][KNN_FIX_CHUNK][DFLOW_DESC_PITCH_H];      // rows widened to float32
    const Geom g = a.g;
    const int win = 2 * g.win + 1;
    const int nov = *ovf_count;
    const bool all = (*flags != 0) || nov > ovf_cap;
    const int total = all ? g.ncx * g.ncy * qwaves * win * win : nov;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int item = blockIdx.x; item < total; item += gridDim.x) {       // block-uniform: barriers inside are safe
        int qcell, qstart, ci, cj;
        if (all) {
            int b = item;
            const int wslot = b % (win * win); b /= win * win;
            qstart = (b % qwaves) * 64; qcell = b / qwaves;
            const int qci = qcell % g.ncx, qcj = qcell / g.ncx;
            const int cimin = max(0, qci - g.win), cjmin = max(0, qcj - g.win), cjmax = min(g.ncy - 1, qcj + g.win);
            const int ncyw = cjmax - cjmin + 1;
            ci = cimin + wslot / ncyw; cj = cjmin + wslot % ncyw;
            if (ci > min(g.ncx - 1, qci + g.win)) continue;
        } else {
            int4 e = ovf_list[item];
            qcell = e.x; qstart = e.y; ci = e.z; cj = e.w;
        }
        const int qci = qcell % g.ncx, qcj = qcell / g.ncx;
        const int qx0 = g.x0(qci), qy0 = g.y0(qcj), qcw = g.x1(qci) - qx0, qnpts = qcw * (g.y1(qcj) - qy0);
        if (qstart >= qnpts) continue;
        int qi = qstart + lane;
        const bool active = qi < qnpts;
        if (!active) qi = qnpts - 1;
        const int qy = qy0 + qi / qcw, qx = qx0 + qi % qcw;
        const size_t pix = (size_t)qy * g.W + qx;
        const int cimin = max(0, qci - g.win), cjmin = max(0, qcj - g.win), cjmax = min(g.ncy - 1, qcj + g.win);
        const int slot = 5 * ((ci - cimin) * (cjmax - cjmin + 1) + (cj - cjmin));
        float q[DFLOW_DESC];
        desc_load_row(q, gd1, pix);
        constexpr int P = DescPitch<T>::value;
        // ---- the candidate cell, chunk by chunk through LDS
        const int cx0 = g.x0(ci), cy0 = g.y0(cj), ccw = g.x1(ci) - cx0, cnpts = ccw * (g.y1(cj) - cy0);
        const int nchunks = (cnpts + KNN_FIX_CHUNK - 1) / KNN_FIX_CHUNK;
        constexpr int PIECES = P * (int)sizeof(T) / 16;                   // 16-byte pieces per row: 17 (float32) or 9 (binary16)
        constexpr int NPC = (KNN_FIX_CHUNK * PIECES + 64 * KNN_FIX_WAVES - 1) / (64 * KNN_FIX_WAVES);   // pieces of a chunk per thread
        uint4 piece[NPC];
#pragma unroll
        for (int u = 0; u < NPC; u++) piece[u] = make_uint4(0u, 0u, 0u, 0u);
        auto fetch = [&](int chunk) {
#pragma unroll
            for (int u = 0; u < NPC; u++) {
                const int tp = (int)threadIdx.x + u * 64 * KNN_FIX_WAVES, idx = chunk * KNN_FIX_CHUNK + tp / PIECES;
                if (tp < KNN_FIX_CHUNK * PIECES && idx < cnpts)
                    piece[u] = reinterpret_cast<const uint4 *>(gd2 + ((size_t)(cy0 + idx / ccw) * g.W + cx0 + idx % ccw) * P)[tp % PIECES];
            }
        };
        auto deposit = [&](int buf) {
#pragma unroll
            for (int u = 0; u < NPC; u++) {
                const int tp = (int)threadIdx.x + u * 64 * KNN_FIX_WAVES, lj = tp / PIECES, lpc = tp % PIECES;
                if (tp >= KNN_FIX_CHUNK * PIECES) continue;
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<uint4 *>(&cand[buf][lj][4 * lpc]) = piece[u];
                } else {
                    const dflow_h8 v = __builtin_bit_cast(dflow_h8, piece[u]);
                    float4 lo = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]), hi = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
                    *reinterpret_cast<float4 *>(&cand[buf][lj][8 * lpc]) = lo;
                    if (8 * lpc + 4 < DFLOW_DESC_PITCH_H) *reinterpret_cast<float4 *>(&cand[buf][lj][8 * lpc + 4]) = hi;
                }
            }
        };
        Top5 t;
        t.d0 = t.d1 = t.d2 = t.d3 = t.d4 = INFINITY;
        t.i0 = t.i1 = t.i2 = t.i3 = t.i4 = 0;
        fetch(0);
        deposit(0);
        __syncthreads();
        for (int chunk = 0; chunk < nchunks; chunk++) {
            if (chunk + 1 < nchunks) fetch(chunk + 1);
#pragma unroll
            for (int jj = 0; jj < KNN_FIX_CHUNK / KNN_FIX_WAVES; jj++) {
                const int j = wave + jj * KNN_FIX_WAVES, idx = chunk * KNN_FIX_CHUNK + j;      // increasing within a wave
                if (idx < cnpts) {                                                             // wave-uniform
                    const float4 *c4 = reinterpret_cast<const float4 *>(cand[chunk & 1][j]);
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < DFLOW_DESC / 4; k++) {
                        const float4 v = c4[k];
                        float e;
                        e = q[4 * k] - v.x; acc = __fmaf_rn(e, e, acc);
                        e = q[4 * k + 1] - v.y; acc = __fmaf_rn(e, e, acc);
                        e = q[4 * k + 2] - v.z; acc = __fmaf_rn(e, e, acc);
                        e = q[4 * k + 3] - v.w; acc = __fmaf_rn(e, e, acc);
                    }
                    if (acc < t.d4) top5_insert(t, acc, idx);
                }
            }
            if (chunk + 1 < nchunks) deposit((chunk + 1) & 1);
            __syncthreads();
        }
        pd[wave][0][lane] = t.d0; pd[wave][1][lane] = t.d1; pd[wave][2][lane] = t.d2; pd[wave][3][lane] = t.d3; pd[wave][4][lane] = t.d4;
        pi[wave][0][lane] = t.i0; pi[wave][1][lane] = t.i1; pi[wave][2][lane] = t.i2; pi[wave][3][lane] = t.i3; pi[wave][4][lane] = t.i4;
        __syncthreads();
        if (wave == 0) {
            // Merge.  The sequential scan orders equal distances by index; here entries arrive wave by wave, so the rule is
            // applied on (distance, index) explicitly: an entry goes in front of a slot iff its distance is smaller, or equal
            // with a smaller index.  Slots still at their initial (inf, 0) hold no candidate and are skipped.
            for (int w = 1; w < KNN_FIX_WAVES; w++)
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    float cd = pd[w][j][lane]; int cidx = pi[w][j][lane];
                    if (!(cd < INFINITY)) continue;
                    bool sh = false;
#define CSWAP(D, I) if (sh || cd < D || (cd == D && cidx < I)) { float td = D; int ti = I; D = cd; I = cidx; cd = td; cidx = ti; sh = true; }
                    CSWAP(t.d0, t.i0) CSWAP(t.d1, t.i1) CSWAP(t.d2, t.i2) CSWAP(t.d3, t.i3) CSWAP(t.d4, t.i4)
#undef CSWAP
                }
            float c[5];
            emit_cell(t, g, ci, cj, pix, qy, qx, slot, a.LP, a.tphi, active, gd1, gd2, gproposals, glcosts, c);
        }
        __syncthreads();
    }
}

int launch_knn(const dflow_params *p, const void *d1, const void *d2, uint32_t *proposals, float *lcosts,
               int32_t *nprop, int32_t *bestlabels, hipStream_t s)
{
    KnnArgs a;
    a.g = make_geom(p);
    a.LP = p->label_pitch; a.tphi = p->tphi;
    int maxpts = (a.g.x1(a.g.ncx - 1) - a.g.x0(a.g.ncx - 1)) * (a.g.y1(a.g.ncy - 1) - a.g.y0(a.g.ncy - 1));
    a.chunks = (maxpts + KNN_THREADS - 1) / KNN_THREADS;
    int nblocks = a.g.ncx * a.g.ncy * a.chunks;
    if (descr_f16(p))
        hipLaunchKernelGGL(knn_exact_kernel<_Float16>, dim3(nblocks), dim3(KNN_THREADS), 0, s, a, (const _Float16 *)d1, (const _Float16 *)d2, proposals, lcosts, nprop, bestlabels);
    else
        hipLaunchKernelGGL(knn_exact_kernel<float>, dim3(nblocks), dim3(KNN_THREADS), 0, s, a, (const float *)d1, (const float *)d2, proposals, lcosts, nprop, bestlabels);
    return dflow_check_launch("knn_exact_kernel");
}

int launch_knn_fix(const dflow_params *p, const void *d1, const void *d2, uint32_t *proposals, float *lcosts,
                   const int *ovf_count, const int4 *ovf_list, int ovf_cap, const int *flags, hipStream_t s)
{
    KnnArgs a;
    a.g = make_geom(p);
    a.LP = p->label_pitch; a.tphi = p->tphi; a.chunks = 0;
    int maxpts = (a.g.x1(a.g.ncx - 1) - a.g.x0(a.g.ncx - 1)) * (a.g.y1(a.g.ncy - 1) - a.g.y0(a.g.ncy - 1));
    int qwaves = (maxpts + 63) / 64;
    if (descr_f16(p))
        hipLaunchKernelGGL(knn_fix_kernel<_Float16>, dim3(KNN_FIX_BLOCKS), dim3(64 * KNN_FIX_WAVES), 0, s, a, (const _Float16 *)d1, (const _Float16 *)d2, proposals,
                           lcosts, ovf_count, ovf_list, ovf_cap, flags, qwaves);
    else
        hipLaunchKernelGGL(knn_fix_kernel<float>, dim3(KNN_FIX_BLOCKS), dim3(64 * KNN_FIX_WAVES), 0, s, a, (const float *)d1, (const float *)d2, proposals,
                           lcosts, ovf_count, ovf_list, ovf_cap, flags, qwaves);
    return dflow_check_launch("knn_fix_kernel");
}
