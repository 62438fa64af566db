// K1/K2 dense DAISY: izracunajDaisy, daisy i flann.py:66,69-77 (cv2.xfeatures2d.DAISY radius=5, q_radius=4,
// q_theta=4, q_hist=4, NRM_NONE, interpolation on, no orientation) evaluated at every pixel.
// OpenCV-contrib is not available anywhere in this project, so the arithmetic below is the build's own
// definition (DESIGN.md "DAISY"; parity with cv2 unpinned); it is bit-identical to oracle/dflow_oracle.c:
// every float operation is a single IEEE op in a fixed order (the file is compiled with -ffp-contract=off).
//
// Pipeline (all HBM-streaming, 4 orientation layers interleaved as one float4 per pixel):
//   gray/255 -> 5-tap blur (sigma 0.5) -> central differences -> 4 half-rectified orientation layers
//   -> 7-tap blur (sigma sqrt(1.6^2-0.25)) -> 4 cascaded blurs (3/5/7/9 taps) = histogram cubes
//   -> gather: 17 grid points x bilinear x 4 bins = 68 floats per pixel.
#include <math.h>
#include "dflow_common.h"

#define MAX_TAPS 9
struct Taps { int n; float k[MAX_TAPS]; };

struct GridTab { double gy[17], gx[17]; };

__device__ static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void gray_kernel(const uint8_t *__restrict__ bgr, float *__restrict__ img, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int g = (1868 * bgr[3 * i] + 9617 * bgr[3 * i + 1] + 4899 * bgr[3 * i + 2] + 8192) >> 14;   // cv::cvtColor BGR2GRAY (u8)
    img[i] = (float)g / 255.0f;
}

template <typename T> __device__ static inline T tmul(float k, T v);
template <> __device__ inline float tmul<float>(float k, float v) { return k * v; }
template <> __device__ inline float4 tmul<float4>(float k, float4 v) { return make_float4(k * v.x, k * v.y, k * v.z, k * v.w); }
__device__ static inline float tadd(float a, float b) { return a + b; }
__device__ static inline float4 tadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// one pass of a separable blur, BORDER_REPLICATE; taps accumulated left to right, multiply then add
template <typename T, bool VERT>
__global__ void blur_kernel(const T *__restrict__ src, T *__restrict__ dst, int H, int W, Taps t)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    int r = t.n / 2;
    T acc;
    if (VERT) {
        acc = tmul<T>(t.k[0], src[(size_t)clampi(y - r, 0, H - 1) * W + x]);
        for (int j = 1; j < t.n; j++) acc = tadd(acc, tmul<T>(t.k[j], src[(size_t)clampi(y - r + j, 0, H - 1) * W + x]));
    } else {
        acc = tmul<T>(t.k[0], src[(size_t)y * W + clampi(x - r, 0, W - 1)]);
        for (int j = 1; j < t.n; j++) acc = tadd(acc, tmul<T>(t.k[j], src[(size_t)y * W + clampi(x - r + j, 0, W - 1)]));
    }
    dst[(size_t)y * W + x] = acc;
}

struct LayerW { float wc[4], ws[4]; };

__global__ void layers_kernel(const float *__restrict__ sm, float4 *__restrict__ lay, int H, int W, LayerW w)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    float dx = (sm[(size_t)y * W + clampi(x + 1, 0, W - 1)] - sm[(size_t)y * W + clampi(x - 1, 0, W - 1)]) * 0.5f;
    float dy = (sm[(size_t)clampi(y + 1, 0, H - 1) * W + x] - sm[(size_t)clampi(y - 1, 0, H - 1) * W + x]) * 0.5f;
    float v[4];
#pragma unroll
    for (int l = 0; l < 4; l++) { float t = dx * w.wc[l] + dy * w.ws[l]; v[l] = t > 0.0f ? t : 0.0f; }
    lay[(size_t)y * W + x] = make_float4(v[0], v[1], v[2], v[3]);
}

// one thread = one (pixel, grid point): a coalesced float4 store into the 68-float descriptor row
__global__ void gather_kernel(const float4 *__restrict__ cubes, float4 *__restrict__ descr, int H, int W, GridTab g)
{
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)H * W * 17;
    if (gid >= total) return;
    int region = (int)(gid % 17);
    size_t pix = gid / 17;
    int y = (int)(pix / W), x = (int)(pix % W);
    int ring = region == 0 ? 0 : (region - 1) / 4;
    const float4 *cube = cubes + (size_t)ring * H * W;
    double yy = (double)y + g.gy[region], xx = (double)x + g.gx[region];
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    bool ok = true;
    if (region != 0) {
        float xf = (float)xx, yf = (float)yy;
        ok = (0.0f <= xf && xf < (float)(W - 1) && 0.0f <= yf && yf < (float)(H - 1));
    }
    int mnx = (int)xx, mny = (int)yy;
    if (ok && !(mnx >= W - 2 || mny >= H - 2)) {
        float4 A = cube[(size_t)mny * W + mnx], B = cube[(size_t)(mny + 1) * W + mnx];
        float4 C = cube[(size_t)mny * W + mnx + 1], D = cube[(size_t)(mny + 1) * W + mnx + 1];
        double alpha = mnx + 1 - xx, beta = mny + 1 - yy;
        float w0 = (float)(alpha * beta);
        float w1 = (float)(beta - w0);
        float w2 = (float)(alpha - w0);
        float w3 = (float)(1 + w0 - alpha - beta);   // (1 + w0) is a float add, as in the oracle
        out = tmul<float4>(w0, A);
        out = tadd(out, tmul<float4>(w1, C));
        out = tadd(out, tmul<float4>(w2, B));
        out = tadd(out, tmul<float4>(w3, D));
    }
    descr[gid] = out;
}

// ---- host side: filter taps exactly as the oracle computes them (same libm, same expressions)
static int filter_size(double sigma)
{
    int fsz = (int)(5.0 * sigma);
    if (fsz % 2 == 0) fsz++;
    if (fsz < 3) fsz = 3;
    return fsz;
}

static Taps gaussian_taps(int n, double sigma)
{
    Taps t;
    t.n = n;
    double scale2x = -0.5 / (sigma * sigma), sum = 0.0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        t.k[i] = (float)exp(scale2x * x * x);
        sum += t.k[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) t.k[i] = (float)(t.k[i] * sum);
    for (int i = n; i < MAX_TAPS; i++) t.k[i] = 0.0f;
    return t;
}

size_t daisy_ws_bytes(const dflow_params *p)
{
    size_t N = (size_t)p->pich * p->picw;
    return N * sizeof(float) * (1 + 1 + 4 + 4 + 16);   // img, sm, tmp(float4), lay(float4), 4 cubes(float4)
}

int launch_daisy(const dflow_params *p, const uint8_t *bgr, float *descr, void *ws, hipStream_t s)
{
    const double pi = 3.14159265358979323846;
    int H = p->pich, W = p->picw;
    size_t N = (size_t)H * W;
    float *img = (float *)ws, *sm = img + N;
    float4 *tmp = (float4 *)(sm + N), *lay = tmp + N, *cubes = lay + N;
    dim3 blk(256), grd((W + 255) / 256, H);

    hipLaunchKernelGGL(gray_kernel, dim3((unsigned)((N + 255) / 256)), blk, 0, s, bgr, img, (int)N);
    Taps t = gaussian_taps(5, 0.5);
    hipLaunchKernelGGL((blur_kernel<float, false>), grd, blk, 0, s, (const float *)img, (float *)tmp, H, W, t);
    hipLaunchKernelGGL((blur_kernel<float, true>), grd, blk, 0, s, (const float *)tmp, sm, H, W, t);
    LayerW lw;
    for (int l = 0; l < 4; l++) {
        float angle = (float)(2 * l * pi / 4);
        lw.wc[l] = (float)cos((double)angle);
        lw.ws[l] = (float)sin((double)angle);
    }
    hipLaunchKernelGGL(layers_kernel, grd, blk, 0, s, (const float *)sm, lay, H, W, lw);
    {
        double sg = sqrt(1.6 * 1.6 - 0.25);
        t = gaussian_taps(filter_size((float)sg), (float)sg);
        hipLaunchKernelGGL((blur_kernel<float4, false>), grd, blk, 0, s, (const float4 *)lay, tmp, H, W, t);
        hipLaunchKernelGGL((blur_kernel<float4, true>), grd, blk, 0, s, (const float4 *)tmp, lay, H, W, t);
    }
    double sig[4];
    for (int r = 0; r < 4; r++) sig[r] = (r + 1) * (5.0 / 4 / 2);
    const float4 *prev = lay;
    for (int r = 0; r < 4; r++) {
        double sg = r == 0 ? sig[0] : sqrt(sig[r] * sig[r] - sig[r - 1] * sig[r - 1]);
        t = gaussian_taps(filter_size(sg), sg);
        hipLaunchKernelGGL((blur_kernel<float4, false>), grd, blk, 0, s, prev, tmp, H, W, t);
        hipLaunchKernelGGL((blur_kernel<float4, true>), grd, blk, 0, s, (const float4 *)tmp, cubes + (size_t)r * N, H, W, t);
        prev = cubes + (size_t)r * N;
    }
    GridTab g;
    double r_step = 5.0 / 4.0, t_step = 2 * pi / 4;
    g.gy[0] = 0.0; g.gx[0] = 0.0;
    for (int r = 0; r < 4; r++)
        for (int a = 0; a < 4; a++) {
            g.gy[1 + r * 4 + a] = (r + 1) * r_step * sin(a * t_step);
            g.gx[1 + r * 4 + a] = (r + 1) * r_step * cos(a * t_step);
        }
    size_t total = N * 17;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((total + 255) / 256)), blk, 0, s, (const float4 *)cubes,
                       (float4 *)descr, H, W, g);
    return dflow_check_launch("daisy kernels");
}
