// K3/K4 knn_cells (exact path): generisi, daisy i flann.py:157-189, with the FLANN search (:171-172) replaced by
// the build's canonical exact 5-NN: squared L2 as a sequential fmaf chain over k = 0..67, ties to the lower
// in-cell index, results in ascending (distance, index) order.
//
// One thread = one image-1 pixel (query); its 68-float descriptor stays in VGPRs.  A workgroup = 256 queries
// of ONE cell of image 1, so all its lanes share the same window of image-2 cells and walk the candidate
// points in lock-step: the candidate's descriptor address is wave-uniform, so it is fetched through the
// scalar cache (s_load) and used as SGPR operands -- no LDS traffic, no barriers.  Per (query, candidate):
// 68 x (v_sub_f32, v_fma_f32).  Cells are visited in the reference's order (ci outer, cj inner, Q2) so slot
// numbers, truncated-L1 costs (numpy pairwise order, Q3) and the running WTA label (strict '<', Q4) come
// out as in the reference.
#include "dflow_common.h"

#define KNN_THREADS 256

struct KnnArgs {
    Geom g;
    int LP, chunks;
    float tphi;
};

// sum_k |a[k]-b[k]| in numpy's float32 pairwise order (np.sum(np.absolute(..)), daisy i flann.py:179-180).
// Cold path (5 winners per cell): both rows are re-read from memory in a rolled loop so that the hot search
// loop keeps its register budget.
__device__ __noinline__ static float l1_cost_np(const float *__restrict__ a, const float *__restrict__ b)
{
    const float4 *a4 = reinterpret_cast<const float4 *>(a), *b4 = reinterpret_cast<const float4 *>(b);
    float r[8];
    {
        float4 u0 = a4[0], u1 = a4[1], v0 = b4[0], v1 = b4[1];
        r[0] = fabsf(u0.x - v0.x); r[1] = fabsf(u0.y - v0.y); r[2] = fabsf(u0.z - v0.z); r[3] = fabsf(u0.w - v0.w);
        r[4] = fabsf(u1.x - v1.x); r[5] = fabsf(u1.y - v1.y); r[6] = fabsf(u1.z - v1.z); r[7] = fabsf(u1.w - v1.w);
    }
#pragma unroll 1
    for (int i = 2; i < 16; i += 2) {
        float4 u0 = a4[i], u1 = a4[i + 1], v0 = b4[i], v1 = b4[i + 1];
        r[0] = r[0] + fabsf(u0.x - v0.x); r[1] = r[1] + fabsf(u0.y - v0.y); r[2] = r[2] + fabsf(u0.z - v0.z); r[3] = r[3] + fabsf(u0.w - v0.w);
        r[4] = r[4] + fabsf(u1.x - v1.x); r[5] = r[5] + fabsf(u1.y - v1.y); r[6] = r[6] + fabsf(u1.z - v1.z); r[7] = r[7] + fabsf(u1.w - v1.w);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    float4 u = a4[16], v = b4[16];
    res = res + fabsf(u.x - v.x); res = res + fabsf(u.y - v.y); res = res + fabsf(u.z - v.z); res = res + fabsf(u.w - v.w);
    return res;
}

// d2 is only ever read and never aliases the outputs: with __restrict__ kernel arguments the compiler can prove
// it and turns the wave-uniform candidate loads into scalar loads.
__global__ void __launch_bounds__(KNN_THREADS, 4) knn_exact_kernel(KnnArgs a, const float *__restrict__ gd1,
                                                                   const float *__restrict__ gd2,
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
    {
        const float4 *s = reinterpret_cast<const float4 *>(gd1 + pix * DFLOW_DESC);
#pragma unroll
        for (int k = 0; k < DFLOW_DESC / 4; k++) { float4 v = s[k]; q[4 * k] = v.x; q[4 * k + 1] = v.y; q[4 * k + 2] = v.z; q[4 * k + 3] = v.w; }
    }
    const int cimin = max(0, qci - g.win), cimax = min(g.ncx - 1, qci + g.win);
    const int cjmin = max(0, qcj - g.win), cjmax = min(g.ncy - 1, qcj + g.win);
    float mind = 1000.0f;   // mindists, daisy i flann.py:93
    int bestl = 0, slot = 0;

    for (int ci = cimin; ci <= cimax; ci++)
        for (int cj = cjmin; cj <= cjmax; cj++) {
            const int cx0 = g.x0(ci), cx1 = g.x1(ci), cy0 = g.y0(cj), cy1 = g.y1(cj), ccw = cx1 - cx0;
            float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY, d3 = INFINITY, d4 = INFINITY;
            int i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0;
            for (int yy = cy0; yy < cy1; yy++) {
                const float4 *__restrict__ row = reinterpret_cast<const float4 *>(gd2 + ((size_t)yy * g.W + cx0) * DFLOW_DESC);
                for (int xx = 0; xx < ccw; xx++) {
                    const float4 *__restrict__ t = row + xx * (DFLOW_DESC / 4);   // wave-uniform address
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < DFLOW_DESC / 4; k++) {
                        float4 v = t[k];
                        float e;
                        e = q[4 * k] - v.x; acc = __fmaf_rn(e, e, acc);
                        e = q[4 * k + 1] - v.y; acc = __fmaf_rn(e, e, acc);
                        e = q[4 * k + 2] - v.z; acc = __fmaf_rn(e, e, acc);
                        e = q[4 * k + 3] - v.w; acc = __fmaf_rn(e, e, acc);
                    }
                    if (acc < d4) {   // insert keeping ascending order; strict '<' leaves equal distances in index order
                        // once the new entry has found its place every later entry shifts down unconditionally
                        // (a displaced entry must stay in front of an equal one that followed it)
                        float cd = acc; int cidx = (yy - cy0) * ccw + xx; bool sh = false;
#define CSWAP(D, I) if (sh || cd < D) { float td = D; int ti = I; D = cd; I = cidx; cd = td; cidx = ti; sh = true; }
                        CSWAP(d0, i0) CSWAP(d1, i1) CSWAP(d2, i2) CSWAP(d3, i3) CSWAP(d4, i4)
#undef CSWAP
                    }
                }
            }
            // daisy i flann.py:174-189: proposals [dy,dx], truncated L1 cost, WTA update, nprop += 5
            const int idx[5] = {i0, i1, i2, i3, i4};
#pragma unroll
            for (int qq = 0; qq < 5; qq++) {
                const int ty = cy0 + idx[qq] / ccw, tx = cx0 + idx[qq] % ccw;
                const float s = l1_cost_np(gd1 + pix * DFLOW_DESC, gd2 + ((size_t)ty * g.W + tx) * DFLOW_DESC);
                const float c = s < a.tphi ? s : a.tphi;   // python min(tphi, s)
                if (active) {
                    gproposals[pix * a.LP + slot + qq] = pack_flow(ty - qy, tx - qx);
                    glcosts[pix * a.LP + slot + qq] = c;
                }
                if (c < mind) { mind = c; bestl = slot + qq; }
            }
            slot += 5;
        }
    if (active) {
        gnprop[pix] = slot;
        gbestlabels[pix] = bestl;
        for (int s = slot; s < a.LP; s++) { gproposals[pix * a.LP + s] = DFLOW_FILL_PROPOSAL; glcosts[pix * a.LP + s] = DFLOW_FILL_COST; }
    }
}

int launch_knn(const dflow_params *p, const float *d1, const float *d2, uint32_t *proposals, float *lcosts,
               int32_t *nprop, int32_t *bestlabels, hipStream_t s)
{
    KnnArgs a;
    a.g = make_geom(p);
    a.LP = p->label_pitch; a.tphi = p->tphi;
    int maxpts = (a.g.x1(a.g.ncx - 1) - a.g.x0(a.g.ncx - 1)) * (a.g.y1(a.g.ncy - 1) - a.g.y0(a.g.ncy - 1));
    a.chunks = (maxpts + KNN_THREADS - 1) / KNN_THREADS;
    int nblocks = a.g.ncx * a.g.ncy * a.chunks;
    hipLaunchKernelGGL(knn_exact_kernel, dim3(nblocks), dim3(KNN_THREADS), 0, s, a, d1, d2, proposals, lcosts, nprop, bestlabels);
    return dflow_check_launch("knn_exact_kernel");
}
