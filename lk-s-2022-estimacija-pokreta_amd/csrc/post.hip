// K7 outputs: labels -> flow (vratiKonacniFlow, python bcd.py:90-95) and the forward/backward consistency
// check (postprocessing.py:7-17 load, :79-117 check) with the reference's transposed indexing (Q13).
#include "dflow_common.h"

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
