// K7 outputs: labels -> flow (vratiKonacniFlow, python bcd.py:90-95), the forward/backward consistency
// check (postprocessing.py:7-17 load, :79-117 check) with the reference's transposed indexing (Q13), the reference's
// packedksets file layout (pakovanje, daisy i flann.py:256-309) for users of the reference's own BCD script, and the
// host-side segment filter (removeSmallSegments, postprocessing.py:29-76).
#include <vector>
#include "dflow_common.h"

// ------------------------------------------------------------------------------------------------ packedksets
// One wave per (pixel, slot): slot 0 = pixel vs the pixel below, slot 1 = vs the pixel to the right (Q8).  Bit
// tl*L + nl of the row-major L x L matrix (np.packbits: most significant bit first) is tpsi > |dy-dy'|+|dx-dx'|
// for tl < nprop[pixel], nl < nprop[neighbour], else 0.  kdim = L*L/8 + 1 bytes per matrix (daisy i flann.py:98).
__global__ void __launch_bounds__(256) pack_compat_kernel(int H, int W, int LP, int L, int tpsi,
                                                          const uint32_t *__restrict__ proposals,
                                                          const int32_t *__restrict__ nprop, uint8_t *__restrict__ packed)
{
    __shared__ uint32_t s_lab[4][2][DFLOW_MAX_LABELS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long item = (long long)blockIdx.x * 4 + wv;
    if (item >= 2LL * H * W) return;
    const int slot = (int)(item & 1), pix = (int)(item >> 1);
    const int y = pix / W, x = pix % W;
    const int ny = slot == 0 ? y + 1 : y, nx = slot == 0 ? x : x + 1;
    const int kdim = L * L / 8 + 1;
    uint8_t *out = packed + (size_t)item * kdim;
    if (ny >= H || nx >= W) {                                 // no such neighbour: the reference leaves zeros
        for (int i = lane; i < kdim; i += 64) out[i] = 0;
        return;
    }
    const int npix = ny * W + nx;
    const int tn = nprop[pix], pn = nprop[npix];
    for (int k = lane; k < L; k += 64) {
        s_lab[wv][0][k] = flow_bias(proposals[(size_t)pix * LP + k]);
        s_lab[wv][1][k] = flow_bias(proposals[(size_t)npix * LP + k]);
    }
    for (int i = lane; i < kdim; i += 64) {
        int b = 8 * i, tl = b / L, nl = b % L;
        uint32_t byte = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const bool on = tl < tn && nl < pn && tl < L && flow_l1_biased(s_lab[wv][0][min(tl, L - 1)], s_lab[wv][1][nl]) < (uint32_t)tpsi;
            byte |= (on ? 1u : 0u) << (7 - j);
            if (++nl == L) { nl = 0; tl++; }
        }
        out[i] = (uint8_t)byte;
    }
}

int launch_pack_compat(const dflow_params *p, const uint32_t *proposals, const int32_t *nprop, uint8_t *packed, hipStream_t s)
{
    long long items = 2LL * p->pich * p->picw;
    hipLaunchKernelGGL(pack_compat_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p->pich, p->picw, p->label_pitch,
                       p->maxnprop, p->tpsi, proposals, nprop, packed);
    return dflow_check_launch("pack_compat_kernel");
}

// ------------------------------------------------------------------------------------------------ segments (host)
// removeSmallSegments, postprocessing.py:29-76, on a HOST (A,B,3) float32 field (A = shape[0], which the reference calls
// "width").  The result depends on the scan order (an invalid seed joins the valid regions around it, :41-62), so this
// is the reference's sequential region growing, statement for statement: seeds in (v outer, u inner) order (with the name-reuse quirk noted below), 4-neighbours
// in the order (u-1, u+1, v-1, v+1), a neighbour joins if it is unchecked, valid and within `tresh` (L1, float32
// differences widened like numpy's float32 arithmetic) of the pixel it is reached from; segments with
// 1 < count < min_segment_size are invalidated.
int host_remove_small_segments(float *flow, int A, int B, float tresh, int min_segment_size)
{
    std::vector<uint8_t> check((size_t)A * B, 0);
    std::vector<int> seg_u, seg_v;
    auto at = [&](int u, int v, int c) -> float & { return flow[((size_t)u * B + v) * 3 + c]; };
    for (int vo = 0; vo < B; vo++) {
        int v = vo;     // the reference's invalidation loop (:74) reuses the names u, v: after it, v keeps the last segment
                        // pixel's value until the outer loop rebinds it, and the remaining seeds of this pass use that v
        for (int u = 0; u < A; u++) {
            if (check[(size_t)u * B + v]) continue;
            seg_u.clear(); seg_v.clear();
            seg_u.push_back(u); seg_v.push_back(v);
            size_t curr = 0;
            while (curr < seg_u.size()) {
                const int uc = seg_u[curr], vc = seg_v[curr];
                const int un[4] = {uc - 1, uc + 1, uc, uc}, vn[4] = {vc, vc, vc - 1, vc + 1};
                for (int k = 0; k < 4; k++) {
                    const int a = un[k], b = vn[k];
                    if (a < 0 || b < 0 || a >= A || b >= B) continue;
                    if (check[(size_t)a * B + b] || !(at(a, b, 2) > 0.5f)) continue;
                    const float d0 = fabsf(at(uc, vc, 0) - at(a, b, 0)), d1 = fabsf(at(uc, vc, 1) - at(a, b, 1));
                    if (d0 + d1 <= tresh) {
                        seg_u.push_back(a); seg_v.push_back(b);
                        check[(size_t)a * B + b] = 1;
                    }
                }
                curr++;
                check[(size_t)uc * B + vc] = 1;
            }
            const int count = (int)seg_u.size();
            if (1 < count && count < min_segment_size) {
                for (size_t i = 0; i < seg_u.size(); i++) at(seg_u[i], seg_v[i], 2) = 0.0f;
                v = seg_v.back();
            }
        }
    }
    return DFLOW_OK;
}

__global__ void labels_to_flow_kernel(const uint32_t *__restrict__ proposals, const int32_t *__restrict__ bestlabels,
                                      float2 *__restrict__ flow, int n, int LP)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t f = proposals[(size_t)i * LP + bestlabels[i]];
    flow[i] = make_float2((float)flow_dy(f), (float)flow_dx(f));
}

// fwd/bwd: (H,W,2) [dy,dx].  out: (H,W,3) [U=dx, V=dy, valid].  "width, height, _ = flow1.shape" (:80) makes u run
// over rows and v over columns; U is added to the ROW index and V to the COLUMN index (:87-88).
__global__ void fb_consistency_kernel(const float2 *__restrict__ fwd, const float2 *__restrict__ bwd, float tresh,
                                      float *__restrict__ out, int H, int W)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    int u1 = i / W, v1 = i % W;
    float2 f = fwd[i];
    float U = f.y, V = f.x, valid = 1.0f;
    int u2 = (int)(U + (float)u1), v2 = (int)(V + (float)v1);
    if (u2 < 0 || v2 < 0 || u2 >= H || v2 >= W) {
        U = 0.0f; V = 0.0f; valid = 0.0f;
    } else {
        float2 g = bwd[(size_t)u2 * W + v2];   // every loaded pixel is valid (:16)
        float du = U + g.y, dv = V + g.x;
        float err = __fsqrt_rn(dv * dv + du * du);
        if (err > tresh) { U = 0.0f; V = 0.0f; valid = 0.0f; }
    }
    out[3 * (size_t)i] = U; out[3 * (size_t)i + 1] = V; out[3 * (size_t)i + 2] = valid;
}

int launch_labels_to_flow(const dflow_params *p, const uint32_t *proposals, const int32_t *bestlabels, float *flow,
                          hipStream_t s)
{
    int n = p->pich * p->picw;
    hipLaunchKernelGGL(labels_to_flow_kernel, dim3((n + 255) / 256), dim3(256), 0, s, proposals, bestlabels,
                       (float2 *)flow, n, p->label_pitch);
    return dflow_check_launch("labels_to_flow_kernel");
}

int launch_fb_consistency(const dflow_params *p, const float *fwd, const float *bwd, float tresh, float *sparse,
                          hipStream_t s)
{
    int n = p->pich * p->picw;
    hipLaunchKernelGGL(fb_consistency_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float2 *)fwd,
                       (const float2 *)bwd, tresh, sparse, p->pich, p->picw);
    return dflow_check_launch("fb_consistency_kernel");
}
