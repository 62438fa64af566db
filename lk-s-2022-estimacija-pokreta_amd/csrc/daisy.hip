// K1/K2 dense DAISY: izracunajDaisy, daisy i flann.py:66,69-77 (cv2.xfeatures2d.DAISY radius=5, q_radius=4,
// q_theta=4, q_hist=4, NRM_NONE, interpolation on, no orientation) evaluated at every pixel.
// OpenCV-contrib is not available anywhere in this project, so the arithmetic below is the build's own
// definition (DESIGN.md "DAISY"; parity with cv2 unpinned); it is bit-identical to oracle/dflow_oracle.c:
// every float operation is a single IEEE op in a fixed order (the file is compiled with -ffp-contract=off).
//
// Pipeline (all HBM-streaming, 4 orientation layers interleaved as one float4 per pixel), 7 launches per image:
//   front_kernel   gray/255 -> 5-tap blur (sigma 0.5) -> central differences -> 4 half-rectified orientation layers,
//                  one LDS tile per workgroup (the three stencils need a halo of 3)
//   blur2d_kernel  x5: 7-tap blur (sigma sqrt(1.6^2-0.25)), then 4 cascaded blurs (3/5/7/9 taps) = histogram cubes; both
//                  passes of a separable blur in one launch through an LDS tile
//   gather_kernel  17 grid points x bilinear x 4 bins = 68 floats per pixel.
// BORDER_REPLICATE everywhere: a halo cell holds the value of the clamped coordinate (for the nested stencils of the
// front kernel: the value of the stencil centred at the clamped coordinate), so tiles reproduce the whole-image result.
#include <math.h>
#include "dflow_common.h"

#define MAX_TAPS 9
struct Taps { int n; float k[MAX_TAPS]; };

struct GridTab { double gy[17], gx[17]; };

__device__ static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ static inline float gray_of(const uint8_t *__restrict__ bgr, int W, int y, int x)
{
    const uint8_t *px = bgr + ((size_t)y * W + x) * 3;
    const int g = (1868 * px[0] + 9617 * px[1] + 4899 * px[2] + 8192) >> 14;   // cv::cvtColor BGR2GRAY (u8)
    return (float)g / 255.0f;
}

template <typename T> __device__ static inline T tmul(float k, T v);
template <> __device__ inline float tmul<float>(float k, float v) { return k * v; }
template <> __device__ inline float4 tmul<float4>(float k, float4 v) { return make_float4(k * v.x, k * v.y, k * v.z, k * v.w); }
__device__ static inline float tadd(float a, float b) { return a + b; }
__device__ static inline float4 tadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

struct LayerW { float wc[4], ws[4]; };

// gray -> sm (5-tap separable blur, taps accumulated left to right, multiply then add) -> gradient -> layers.
// Tile FX x FY outputs; g = gray on the tile + 3, th = horizontally blurred on (FX + 2) x (FY + 6), sm on the tile + 1.
#define FX 32
#define FY 8
__global__ void __launch_bounds__(FX * FY) front_kernel(const uint8_t *__restrict__ bgr, float4 *__restrict__ lay, int H, int W, Taps t, LayerW w)
{
    __shared__ float g[FY + 6][FX + 6], th[FY + 6][FX + 2], sm[FY + 2][FX + 2];
    const int gx0 = blockIdx.x * FX, gy0 = blockIdx.y * FY, tid = threadIdx.y * FX + threadIdx.x;
    for (int i = tid; i < (FY + 6) * (FX + 6); i += FX * FY) {
        const int r = i / (FX + 6), c = i % (FX + 6);
        g[r][c] = gray_of(bgr, W, clampi(gy0 - 3 + r, 0, H - 1), clampi(gx0 - 3 + c, 0, W - 1));
    }
    __syncthreads();
    for (int i = tid; i < (FY + 6) * (FX + 2); i += FX * FY) {
        const int r = i / (FX + 2), c = i % (FX + 2);
        const int X = clampi(gx0 - 1 + c, 0, W - 1);                    // the stencil is centred at the CLAMPED column
        float acc = t.k[0] * g[r][clampi(X - 2, 0, W - 1) - (gx0 - 3)];
#pragma unroll
        for (int j = 1; j < 5; j++) acc = acc + t.k[j] * g[r][clampi(X - 2 + j, 0, W - 1) - (gx0 - 3)];
        th[r][c] = acc;
    }
    __syncthreads();
    for (int i = tid; i < (FY + 2) * (FX + 2); i += FX * FY) {
        const int r = i / (FX + 2), c = i % (FX + 2);
        const int Y = clampi(gy0 - 1 + r, 0, H - 1);                    // ... and at the CLAMPED row
        float acc = t.k[0] * th[clampi(Y - 2, 0, H - 1) - (gy0 - 3)][c];
#pragma unroll
        for (int j = 1; j < 5; j++) acc = acc + t.k[j] * th[clampi(Y - 2 + j, 0, H - 1) - (gy0 - 3)][c];
        sm[r][c] = acc;
    }
    __syncthreads();
    const int x = gx0 + threadIdx.x, y = gy0 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int c = threadIdx.x + 1, r = threadIdx.y + 1;
    const float dx = (sm[r][c + 1] - sm[r][c - 1]) * 0.5f;
    const float dy = (sm[r + 1][c] - sm[r - 1][c]) * 0.5f;
    float v[4];
#pragma unroll
    for (int l = 0; l < 4; l++) { float q = dx * w.wc[l] + dy * w.ws[l]; v[l] = q > 0.0f ? q : 0.0f; }
    lay[(size_t)y * W + x] = make_float4(v[0], v[1], v[2], v[3]);
}

// both passes of a separable blur of the float4 layers, BORDER_REPLICATE; taps accumulated left to right, multiply then
// add.  Tile BX x BY outputs (two rows per thread), halo r = n/2 <= 4.
#define BX 32
#define BY 16
#define BR 4
__global__ void __launch_bounds__(BX * BY / 2) blur2d_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, int H, int W, Taps t)
{
    __shared__ float4 s[BY + 2 * BR][BX + 2 * BR], h[BY + 2 * BR][BX];
    const int r = t.n / 2;
    const int gx0 = blockIdx.x * BX, gy0 = blockIdx.y * BY, tid = threadIdx.y * BX + threadIdx.x, nthr = BX * BY / 2;
    const int sw = BX + 2 * r, sh = BY + 2 * r;
    for (int i = tid; i < sh * sw; i += nthr) {
        const int rr = i / sw, cc = i % sw;
        s[rr][cc] = src[(size_t)clampi(gy0 - r + rr, 0, H - 1) * W + clampi(gx0 - r + cc, 0, W - 1)];
    }
    __syncthreads();
    for (int i = tid; i < sh * BX; i += nthr) {
        const int rr = i / BX, cc = i % BX;
        float4 acc = tmul<float4>(t.k[0], s[rr][cc]);
        for (int j = 1; j < t.n; j++) acc = tadd(acc, tmul<float4>(t.k[j], s[rr][cc + j]));
        h[rr][cc] = acc;
    }
    __syncthreads();
    const int x = gx0 + threadIdx.x;
    if (x >= W) return;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int ly = threadIdx.y + half * (BY / 2), y = gy0 + ly;
        if (y >= H) break;
        float4 acc = tmul<float4>(t.k[0], h[ly][threadIdx.x]);
        for (int j = 1; j < t.n; j++) acc = tadd(acc, tmul<float4>(t.k[j], h[ly + j][threadIdx.x]));
        dst[(size_t)y * W + x] = acc;
    }
}

// one thread = one (pixel, grid point): a coalesced float4 store into the 68-float descriptor row
__global__ void gather_kernel(const float4 *__restrict__ cubes, float4 *__restrict__ descr, int H, int W, GridTab g, int f16)
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
    if (f16) {
        // DFLOW_FLAG_DESCR_F16: descriptor values rounded to binary16 (round to nearest even) and STORED as binary16: rows of
        // 68 values + 4 zero pads = 144 bytes (the last region's thread writes the pad)
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 *row = reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(descr) + pix * DFLOW_DESC_PITCH_H);
        row[region] = (h4){(_Float16)out.x, (_Float16)out.y, (_Float16)out.z, (_Float16)out.w};
        if (region == 16) row[17] = (h4){(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
    } else {
        descr[gid] = out;
    }
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

int launch_daisy(const dflow_params *p, const uint8_t *bgr, void *descr, void *ws, hipStream_t s)
{
    const double pi = 3.14159265358979323846;
    int H = p->pich, W = p->picw;
    size_t N = (size_t)H * W;
    float *img = (float *)ws, *sm = img + N;            // (first two planes of the workspace: unused since the front kernel is fused)
    float4 *tmp = (float4 *)(sm + N), *lay = tmp + N, *cubes = lay + N;
    dim3 blk(256);

    Taps t = gaussian_taps(5, 0.5);
    LayerW lw;
    for (int l = 0; l < 4; l++) {
        float angle = (float)(2 * l * pi / 4);
        lw.wc[l] = (float)cos((double)angle);
        lw.ws[l] = (float)sin((double)angle);
    }
    hipLaunchKernelGGL(front_kernel, dim3((W + FX - 1) / FX, (H + FY - 1) / FY), dim3(FX, FY), 0, s, bgr, lay, H, W, t, lw);
    dim3 bgrd((W + BX - 1) / BX, (H + BY - 1) / BY), bblk(BX, BY / 2);
    {
        double sg = sqrt(1.6 * 1.6 - 0.25);
        t = gaussian_taps(filter_size((float)sg), (float)sg);
        hipLaunchKernelGGL(blur2d_kernel, bgrd, bblk, 0, s, (const float4 *)lay, tmp, H, W, t);
    }
    double sig[4];
    for (int r = 0; r < 4; r++) sig[r] = (r + 1) * (5.0 / 4 / 2);
    const float4 *prev = tmp;
    for (int r = 0; r < 4; r++) {
        double sg = r == 0 ? sig[0] : sqrt(sig[r] * sig[r] - sig[r - 1] * sig[r - 1]);
        t = gaussian_taps(filter_size(sg), sg);
        hipLaunchKernelGGL(blur2d_kernel, bgrd, bblk, 0, s, prev, cubes + (size_t)r * N, H, W, t);
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
                       (float4 *)descr, H, W, g, (p->flags & DFLOW_FLAG_DESCR_F16) ? 1 : 0);
    return dflow_check_launch("daisy kernels");
}
