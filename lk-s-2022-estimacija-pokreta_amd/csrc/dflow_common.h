// Shared host/device helpers of libdflow.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dflow.h"

#define DFLOW_FILL_PROPOSAL 0xFFFFFFFFu /* [-1,-1], daisy i flann.py:89 */
#define DFLOW_FILL_COST 1000.0f         /* daisy i flann.py:90 */

// Geometry of the cell grid (daisy i flann.py:42-43,85-86; ragged last cells: DESIGN.md "Geometry").
struct Geom {
    int H, W, ch, cw, ncx, ncy, win;
    __host__ __device__ int cellx(int x) const { int c = x / cw; return c < ncx ? c : ncx - 1; }
    __host__ __device__ int celly(int y) const { int c = y / ch; return c < ncy ? c : ncy - 1; }
    __host__ __device__ int x0(int ci) const { return ci * cw; }
    __host__ __device__ int y0(int cj) const { return cj * ch; }
    __host__ __device__ int x1(int ci) const { return ci == ncx - 1 ? W : (ci + 1) * cw; }
    __host__ __device__ int y1(int cj) const { return cj == ncy - 1 ? H : (cj + 1) * ch; }
};

static inline Geom make_geom(const dflow_params *p)
{
    Geom g;
    g.H = p->pich; g.W = p->picw; g.ch = p->cellh; g.cw = p->cellw;
    g.ncx = p->picw / p->cellw; g.ncy = p->pich / p->cellh; g.win = p->window;
    return g;
}

// label packing: int16 dy | int16 dx << 16
__host__ __device__ static inline uint32_t pack_flow(int dy, int dx)
{
    return (uint32_t)(uint16_t)(int16_t)dy | ((uint32_t)(uint16_t)(int16_t)dx << 16);
}
__host__ __device__ static inline int flow_dy(uint32_t f) { return (int)(int16_t)(f & 0xFFFFu); }
__host__ __device__ static inline int flow_dx(uint32_t f) { return (int)(int16_t)(f >> 16); }

// |dy-dy'| + |dx-dx'| of two packed labels (purepsi, daisy i flann.py:114-115): flip the sign bits so that the
// int16 halves order like uint16, then one v_sad_u16.
__device__ static inline uint32_t flow_bias(uint32_t f) { return f ^ 0x80008000u; }
__device__ static inline uint32_t flow_l1_biased(uint32_t a, uint32_t b) { return __builtin_amdgcn_sad_u16(a, b, 0u); }

// ---- descriptor storage: float32 rows of 68 values (272 bytes), or -- DFLOW_FLAG_DESCR_F16 -- binary16 rows of 68 values
// + 4 zero pads (144 bytes, 16-byte aligned).  Arithmetic is always on the values widened to float32.
#define DFLOW_DESC_PITCH_H 72
typedef _Float16 dflow_h8 __attribute__((ext_vector_type(8)));
template <typename T> struct DescPitch { static constexpr int value = DFLOW_DESC; };
template <> struct DescPitch<_Float16> { static constexpr int value = DFLOW_DESC_PITCH_H; };
static inline bool descr_f16(const dflow_params *p) { return (p->flags & DFLOW_FLAG_DESCR_F16) != 0; }

// the descriptor of pixel pix into registers
template <typename T> __device__ static inline void desc_load_row(float (&q)[DFLOW_DESC], const T *__restrict__ base, size_t pix)
{
    if constexpr (sizeof(T) == 4) {
        const float4 *s = reinterpret_cast<const float4 *>(base + pix * DFLOW_DESC);
#pragma unroll
        for (int k = 0; k < DFLOW_DESC / 4; k++) { float4 v = s[k]; q[4 * k] = v.x; q[4 * k + 1] = v.y; q[4 * k + 2] = v.z; q[4 * k + 3] = v.w; }
    } else {
        const dflow_h8 *s = reinterpret_cast<const dflow_h8 *>(base + pix * DFLOW_DESC_PITCH_H);
#pragma unroll
        for (int k = 0; k < DFLOW_DESC_PITCH_H / 8; k++) {
            const dflow_h8 v = s[k];
#pragma unroll
            for (int j = 0; j < 8; j++) if (8 * k + j < DFLOW_DESC) q[8 * k + j] = (float)v[j];
        }
    }
}

// numpy float32 pairwise-sum order for 68 contiguous values (np.sum at daisy i flann.py:179,229)
__device__ static inline float np_pairwise_sum68(const float *a)
{
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = a[j];
#pragma unroll
    for (int i = 8; i < 64; i += 8)
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = __fadd_rn(r[j], a[i + j]);
    float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                          __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
#pragma unroll
    for (int i = 64; i < DFLOW_DESC; i++) res = __fadd_rn(res, a[i]);
    return res;
}

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

// the same for binary16 rows (both rows widened to float32 first; same summation order)
__device__ __noinline__ static float l1_cost_np(const _Float16 *__restrict__ a, const _Float16 *__restrict__ b)
{
    float u[DFLOW_DESC], v[DFLOW_DESC], r[8];
    desc_load_row(u, a, 0); desc_load_row(v, b, 0);
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = fabsf(u[j] - v[j]);
#pragma unroll
    for (int i = 8; i < 64; i += 8)
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = r[j] + fabsf(u[i + j] - v[i + j]);
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int i = 64; i < DFLOW_DESC; i++) res = res + fabsf(u[i] - v[i]);
    return res;
}

// error plumbing (abi.hip)
int dflow_set_error(int code, const char *fmt, ...);
int dflow_check_launch(const char *what);
int dflow_check_params(const dflow_params *p);

// stage launchers (one per .hip file)
int launch_daisy(const dflow_params *p, const uint8_t *bgr, void *descr, void *ws, hipStream_t s);
size_t daisy_ws_bytes(const dflow_params *p);
// d1, d2: float32 (H,W,68) or, with DFLOW_FLAG_DESCR_F16, binary16 (H,W,72)
int launch_knn(const dflow_params *p, const void *d1, const void *d2, uint32_t *proposals, float *lcosts,
               int32_t *nprop, int32_t *bestlabels, hipStream_t s);
#define KNN_MFMA_EVENTS 7
int launch_knn_mfma(const dflow_params *p, const void *d1, const void *d2, uint32_t *proposals, float *lcosts,
                    int32_t *nprop, int32_t *bestlabels, void *ws, hipStream_t s, hipEvent_t *ev = nullptr);
double knn_mfma_issued(const dflow_params *p);
int knn_mfma_stats(const dflow_params *p, void *ws, hipStream_t s, int64_t *h_out);
size_t knn_mfma_ws_bytes(const dflow_params *p);
bool knn_mfma_supported(const dflow_params *p);
int launch_neighbour(const dflow_params *p, const void *d1, const void *d2, uint32_t *proposals, float *lcosts,
                     int32_t *nprop, const int32_t *bestlabels, void *ws, hipStream_t s);
size_t neighbour_ws_bytes(const dflow_params *p);
int launch_bcd_phase(const dflow_params *p, const uint32_t *proposals, const int32_t *nprop, int32_t *bestlabels, int phase,
                     void *ws, hipStream_t s);
int launch_bcd_phase_batch(const dflow_params *p, int npass, const int32_t *const *nprop, int32_t *const *bestlabels, int phase,
                           void *const *ws, hipStream_t s);
size_t bcd_ws_bytes(const dflow_params *p);
int launch_bcd_prepare(const dflow_params *p, const uint32_t *proposals, const float *lcosts, const int32_t *nprop, void *ws,
                       hipStream_t s);
int launch_labels_to_flow(const dflow_params *p, const uint32_t *proposals, const int32_t *bestlabels, float *flow,
                          hipStream_t s);
int launch_pack_compat(const dflow_params *p, const uint32_t *proposals, const int32_t *nprop, uint8_t *packed, hipStream_t s);
int host_remove_small_segments(float *flow, int A, int B, float tresh, int min_segment_size);
int launch_fb_consistency(const dflow_params *p, const float *fwd, const float *bwd, float tresh, float *sparse,
                          hipStream_t s);
