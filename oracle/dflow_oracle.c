/*
 * dflow_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X build.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path (lk-s-2022-estimacija-pokreta_amd/) never does.
 *
 * Every function cites the reference lines it restates (paths are into /root/reference).
 * Pinning (see oracle/gen_golden.py, tests/golden/):
 *   - orc_knn_proposals glue, orc_neighbour_proposals, orc_pack_compat, orc_bcd_chain, orc_bcd_sweep,
 *     orc_labels_to_flow, orc_fb_consistency: PINNED label-for-label / bit-for-bit against the reference's
 *     own functions executed in the build container on injected descriptors / kNN results / normal draws.
 *   - orc_daisy (OpenCV-contrib xfeatures2d::DAISY arithmetic) and the exact kNN search itself (FLANN):
 *     PARITY UNPINNED -- those libraries are not installed and the reference holds no vectors for them; the
 *     definitions here are the spec for this build (DESIGN.md "Oracle").
 *
 * Build: gcc -O3 -ffp-contract=off -mavx2 -mfma -fPIC -shared (see oracle/Makefile).  -ffp-contract=off matters:
 * every float op below is a single IEEE operation unless it is an explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads: the reference is single-threaded and so is this restatement by default.  orc_set_threads(n) lets the loops
 * over independent units -- image rows in generisi's (ci,cj) body, image columns in nasumicni, the chains of one BCD
 * phase (python bcd.py:265-277: they touch disjoint pixels) -- run on n threads; every unit is still computed by the same
 * code in the same order, so the results do not depend on n (tests/test_oracle_golden.py checks that). */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

#define ORC_DESC 68

typedef struct orc_params {
    int32_t pich, picw;     /* daisy i flann.py:34-35 */
    int32_t cellh, cellw;   /* daisy i flann.py:42-43 */
    int32_t maxnprop;       /* daisy i flann.py:88  (150) */
    int32_t knn;            /* daisy i flann.py:172 (5)   */
    int32_t window;         /* daisy i flann.py:167-168 (+-2 cells) */
    int32_t ngauss;         /* daisy i flann.py:207 (25) */
    int32_t tpsi;           /* daisy i flann.py:47  (8)  */
    int32_t max_attempts;   /* build addition: bound on rejected draws per pixel */
    float tphi;             /* daisy i flann.py:46  (2.5) */
    float sigma;            /* daisy i flann.py:208 (8)  */
    double lamda;           /* daisy i flann.py:48  (0.05) */
    uint64_t seed;          /* build addition: counter-based RNG key (reference is unseeded, :219) */
} orc_params;

/* ------------------------------------------------------------------------------------------------
 * Geometry (daisy i flann.py:85-86 assumes exact tiling; ragged last row/column of cells is this
 * build's generalisation, SURVEY Q12: the last cell absorbs the remainder).
 * ---------------------------------------------------------------------------------------------- */
static inline int ncellx_of(const orc_params *p) { return p->picw / p->cellw; }
static inline int ncelly_of(const orc_params *p) { return p->pich / p->cellh; }
static inline int cell_x(const orc_params *p, int x) { int c = x / p->cellw, n = ncellx_of(p); return c < n ? c : n - 1; }
static inline int cell_y(const orc_params *p, int y) { int c = y / p->cellh, n = ncelly_of(p); return c < n ? c : n - 1; }
static inline int cell_x0(const orc_params *p, int ci) { return ci * p->cellw; }
static inline int cell_y0(const orc_params *p, int cj) { return cj * p->cellh; }
static inline int cell_x1(const orc_params *p, int ci) { return ci == ncellx_of(p) - 1 ? p->picw : (ci + 1) * p->cellw; }
static inline int cell_y1(const orc_params *p, int cj) { return cj == ncelly_of(p) - 1 ? p->pich : (cj + 1) * p->cellh; }

/* numpy float32 pairwise summation order for n = 68 contiguous elements
 * (np.sum at daisy i flann.py:179-180 and :229): eight running sums over the first 64 elements,
 * a fixed combine tree, then the 4 leftovers added in order. */
static inline float np_pairwise_sum68(const float *a)
{
    float r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    for (int i = 8; i < 64; i += 8)
        for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (int i = 64; i < ORC_DESC; i++) res = res + a[i];
    return res;
}

/* ================================================================================================
 * DAISY  (daisy i flann.py:66 DAISY_create(radius=5,q_radius=4,q_theta=4,q_hist=4), :69-77 compute)
 * PARITY UNPINNED: restates OpenCV-contrib xfeatures2d/src/daisy.cpp from memory:
 * NRM_NONE, interpolation on, no orientation.  This definition is the spec of the build.
 * ============================================================================================== */
static int filter_size(double sigma)
{
    int fsz = (int)(5.0 * sigma);
    if (fsz % 2 == 0) fsz++;
    if (fsz < 3) fsz = 3;
    return fsz;
}

/* cv::getGaussianKernel(n, sigma, CV_32F): taps rounded to float, summed in double, renormalised. */
static void gaussian_taps(int n, double sigma, float *k)
{
    double scale2x = -0.5 / (sigma * sigma), sum = 0.0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        k[i] = (float)exp(scale2x * x * x);
        sum += k[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* separable blur, rows then columns, BORDER_REPLICATE, taps accumulated left-to-right, mul then add */
static void blur_sep(const float *src, float *dst, float *tmp, int H, int W, int C, const float *k, int n)
{
    int r = n / 2;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < C; c++) {
                float acc = k[0] * src[((size_t)y * W + clampi(x - r, 0, W - 1)) * C + c];
                for (int j = 1; j < n; j++)
                    acc = acc + k[j] * src[((size_t)y * W + clampi(x - r + j, 0, W - 1)) * C + c];
                tmp[((size_t)y * W + x) * C + c] = acc;
            }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < C; c++) {
                float acc = k[0] * tmp[((size_t)clampi(y - r, 0, H - 1) * W + x) * C + c];
                for (int j = 1; j < n; j++)
                    acc = acc + k[j] * tmp[((size_t)clampi(y - r + j, 0, H - 1) * W + x) * C + c];
                dst[((size_t)y * W + x) * C + c] = acc;
            }
}

#define DAISY_RINGS 4
#define DAISY_ANGLES 4
#define DAISY_HIST 4
#define DAISY_POINTS 17

void orc_daisy_grid(double *gy, double *gx)
{
    const double pi = 3.14159265358979323846;
    double r_step = 5.0 / (double)DAISY_RINGS, t_step = 2 * pi / DAISY_ANGLES;
    gy[0] = 0.0; gx[0] = 0.0;
    for (int r = 0; r < DAISY_RINGS; r++)
        for (int a = 0; a < DAISY_ANGLES; a++) {
            gy[1 + r * DAISY_ANGLES + a] = (r + 1) * r_step * sin(a * t_step);
            gx[1 + r * DAISY_ANGLES + a] = (r + 1) * r_step * cos(a * t_step);
        }
}

/* cubes: out[4][H][W][4] smoothed orientation layers (exposed for tests) */
void orc_daisy_cubes(const uint8_t *bgr, int H, int W, float *cubes)
{
    const double pi = 3.14159265358979323846;
    size_t N = (size_t)H * W;
    float *img = (float *)malloc(N * sizeof(float));
    float *sm = (float *)malloc(N * sizeof(float));
    float *tmp = (float *)malloc(N * 4 * sizeof(float));
    float *lay = (float *)malloc(N * 4 * sizeof(float));
    float k[16];
    /* cv::cvtColor BGR2GRAY on u8 (fixed point, 14 bits), then /255 in float */
    for (size_t i = 0; i < N; i++) {
        int g = (1868 * bgr[3 * i] + 9617 * bgr[3 * i + 1] + 4899 * bgr[3 * i + 2] + 8192) >> 14;
        img[i] = (float)g / 255.0f;
    }
    /* layered_gradient: GaussianBlur 5x5 sigma 0.5; central differences * 0.5; 4 half-rectified layers */
    gaussian_taps(5, 0.5, k);
    blur_sep(img, sm, tmp, H, W, 1, k, 5);
    float wc[4], ws[4];
    for (int l = 0; l < 4; l++) {
        float angle = (float)(2 * l * pi / 4);
        wc[l] = (float)cos((double)angle);
        ws[l] = (float)sin((double)angle);
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float dx = (sm[(size_t)y * W + clampi(x + 1, 0, W - 1)] - sm[(size_t)y * W + clampi(x - 1, 0, W - 1)]) * 0.5f;
            float dy = (sm[(size_t)clampi(y + 1, 0, H - 1) * W + x] - sm[(size_t)clampi(y - 1, 0, H - 1) * W + x]) * 0.5f;
            for (int l = 0; l < 4; l++) {
                float v = dx * wc[l] + dy * ws[l];
                lay[((size_t)y * W + x) * 4 + l] = v > 0.0f ? v : 0.0f;
            }
        }
    /* "assuming a 0.5 image smoothness, pull this to 1.6": sigma = sqrt(1.6^2 - 0.25) */
    {
        double s = sqrt(1.6 * 1.6 - 0.25);
        int n = filter_size((float)s);
        gaussian_taps(n, (float)s, k);
        blur_sep(lay, lay, tmp, H, W, 4, k, n);
    }
    /* cube sigmas (r+1)*rad/rad_q_no/2 = 0.625 (r+1); incremental smoothing between cubes */
    double sig[4];
    for (int r = 0; r < 4; r++) sig[r] = (r + 1) * (5.0 / 4 / 2);
    const float *prev = lay;
    for (int r = 0; r < 4; r++) {
        double s = r == 0 ? sig[0] : sqrt(sig[r] * sig[r] - sig[r - 1] * sig[r - 1]);
        int n = filter_size(s);
        gaussian_taps(n, s, k);
        blur_sep(prev, cubes + (size_t)r * N * 4, tmp, H, W, 4, k, n);
        prev = cubes + (size_t)r * N * 4;
    }
    free(img); free(sm); free(tmp); free(lay);
}

static void bi_get_histogram(float *hist, double y, double x, const float *cube, int H, int W)
{
    int mnx = (int)x, mny = (int)y;
    if (mnx >= W - 2 || mny >= H - 2) { memset(hist, 0, 4 * sizeof(float)); return; }
    const float *A = cube + ((size_t)mny * W + mnx) * 4;
    const float *B = cube + ((size_t)(mny + 1) * W + mnx) * 4;
    const float *C = cube + ((size_t)mny * W + mnx + 1) * 4;
    const float *D = cube + ((size_t)(mny + 1) * W + mnx + 1) * 4;
    double alpha = mnx + 1 - x, beta = mny + 1 - y;
    float w0 = (float)(alpha * beta);
    float w1 = (float)(beta - w0);
    float w2 = (float)(alpha - w0);
    float w3 = (float)(1 + w0 - alpha - beta);
    for (int h = 0; h < 4; h++) hist[h] = w0 * A[h];
    for (int h = 0; h < 4; h++) hist[h] = hist[h] + w1 * C[h];
    for (int h = 0; h < 4; h++) hist[h] = hist[h] + w2 * B[h];
    for (int h = 0; h < 4; h++) hist[h] = hist[h] + w3 * D[h];
}

/* descr: (H,W,68) float32, row y*W+x = keypoint order of daisy i flann.py:70 */
void orc_daisy(const uint8_t *bgr, int H, int W, float *descr)
{
    size_t N = (size_t)H * W;
    float *cubes = (float *)malloc(N * 16 * sizeof(float));
    double gy[DAISY_POINTS], gx[DAISY_POINTS];
    orc_daisy_cubes(bgr, H, W, cubes);
    orc_daisy_grid(gy, gx);
    memset(descr, 0, N * ORC_DESC * sizeof(float));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float *d = descr + ((size_t)y * W + x) * ORC_DESC;
            bi_get_histogram(d, (double)y, (double)x, cubes, H, W); /* centre reads cube 0 */
            for (int r = 0; r < DAISY_RINGS; r++)
                for (int a = 0; a < DAISY_ANGLES; a++) {
                    int region = 1 + r * DAISY_ANGLES + a;
                    double yy = y + gy[region], xx = x + gx[region];
                    float xf = (float)xx, yf = (float)yy;
                    if (!(0.0f <= xf && xf < (float)(W - 1) && 0.0f <= yf && yf < (float)(H - 1))) continue;
                    bi_get_histogram(d + region * DAISY_HIST, yy, xx, cubes + (size_t)r * N * 4, H, W);
                }
        }
    free(cubes);
}

/* ================================================================================================
 * kNN proposals: generisi, daisy i flann.py:157-189 (+ napraviCD2 :144-148 cell tiling).
 * The FLANN search (:171-172) is replaced by the build's canonical exact search (PARITY UNPINNED for the
 * search itself): squared L2 as a sequential fmaf chain over k = 0..67, ties to the lower in-cell index,
 * results in ascending (distance, index) order.  The glue (slot order Q2, [dy,dx] Q1, truncated L1 cost
 * in numpy order Q3, WTA with strict '<' Q4) is pinned against the reference.
 * ============================================================================================== */
static inline float knn_dist(const float *a, const float *b)
{
    float acc = 0.0f;
    for (int k = 0; k < ORC_DESC; k++) { float d = a[k] - b[k]; acc = fmaf(d, d, acc); }
    return acc;
}

/* insertion of (d, r) into an ascending top-K list of n entries; strict '<' keeps the lower index first */
static inline int topk_insert(int K, int n, float *dist, int32_t *idx, float d, int r)
{
    int pos = n < K ? n : K;
    while (pos > 0 && d < dist[pos - 1]) pos--;
    if (pos >= K) return n;
    int last = n < K ? n : K - 1;
    for (int m = last; m > pos; m--) { dist[m] = dist[m - 1]; idx[m] = idx[m - 1]; }
    dist[pos] = d; idx[pos] = r;
    return n < K ? n + 1 : n;
}

/* exact K-NN of q among n points stored as rows (what flann.nn_index is asked for at daisy i flann.py:171) */
void orc_knn_points(const float *q, const float *pts, int n, int K, int32_t *idx, float *dist)
{
    int m = 0;
    for (int r = 0; r < n; r++) m = topk_insert(K, m, dist, idx, knn_dist(q, pts + (size_t)r * ORC_DESC), r);
}

/* eight independent fmaf chains at once (same per-candidate arithmetic as knn_dist, just interleaved so the
 * CPU pipelines them); rows are consecutive 68-float descriptors */
static inline void knn_dist8(const float *a, const float *rows, float *out)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < ORC_DESC; k++)
        for (int c = 0; c < 8; c++) { float d = a[k] - rows[c * ORC_DESC + k]; acc[c] = fmaf(d, d, acc[c]); }
    for (int c = 0; c < 8; c++) out[c] = acc[c];
}

/* exact 5-NN of query q among the points of cell (ci,cj) of image 2; idx = in-cell row-major index */
void orc_knn_cell(const orc_params *p, const float *q, const float *d2, int ci, int cj, int32_t *idx, float *dist)
{
    int K = p->knn, x0 = cell_x0(p, ci), x1 = cell_x1(p, ci), y0 = cell_y0(p, cj), y1 = cell_y1(p, cj);
    int cw = x1 - x0, n = 0;
    float d8[8];
    for (int yy = y0; yy < y1; yy++) {
        const float *row = d2 + ((size_t)yy * p->picw + x0) * ORC_DESC;
        int xx = 0;
        for (; xx + 8 <= cw; xx += 8) {
            knn_dist8(q, row + (size_t)xx * ORC_DESC, d8);
            for (int c = 0; c < 8; c++) n = topk_insert(K, n, dist, idx, d8[c], (yy - y0) * cw + xx + c);
        }
        for (; xx < cw; xx++) n = topk_insert(K, n, dist, idx, knn_dist(q, row + (size_t)xx * ORC_DESC), (yy - y0) * cw + xx);
    }
}

void orc_knn_proposals(const orc_params *p, const float *d1, const float *d2,
                       int64_t *proposals, double *lcosts, int64_t *nprop, int64_t *bestlabels)
{
    int H = p->pich, W = p->picw, L = p->maxnprop, K = p->knn;
    int ncx = ncellx_of(p), ncy = ncelly_of(p);
    size_t N = (size_t)H * W;
    double *mindists = (double *)malloc(N * sizeof(double));
    for (size_t i = 0; i < N * L * 2; i++) proposals[i] = -1;   /* :89 */
    for (size_t i = 0; i < N * L; i++) lcosts[i] = 1000.0;      /* :90 */
    for (size_t i = 0; i < N; i++) { nprop[i] = 0; bestlabels[i] = 0; mindists[i] = 1000.0; } /* :91-95 */
    for (int ci = 0; ci < ncx; ci++)          /* :162 ci outer */
        for (int cj = 0; cj < ncy; cj++) {    /* :163 cj inner */
            int cw = cell_x1(p, ci) - cell_x0(p, ci);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
            for (int y = 0; y < H; y++) {
                int32_t idx[16]; float dist[16], diff[ORC_DESC];
                if (abs(cell_y(p, y) - cj) > p->window) continue;       /* :168 */
                for (int x = 0; x < W; x++) {
                    if (abs(cell_x(p, x) - ci) > p->window) continue;   /* :167 */
                    size_t pix = (size_t)y * W + x;
                    const float *q = d1 + pix * ORC_DESC;
                    orc_knn_cell(p, q, d2, ci, cj, idx, dist);
                    int64_t base = nprop[pix];
                    for (int qq = 0; qq < K; qq++) {
                        int ty = cell_y0(p, cj) + idx[qq] / cw, tx = cell_x0(p, ci) + idx[qq] % cw;
                        proposals[(pix * L + base + qq) * 2 + 0] = ty - y;    /* :176-177 */
                        proposals[(pix * L + base + qq) * 2 + 1] = tx - x;    /* :174-175 */
                        const float *t = d2 + ((size_t)ty * W + tx) * ORC_DESC;
                        for (int k = 0; k < ORC_DESC; k++) diff[k] = fabsf(q[k] - t[k]);
                        float s = np_pairwise_sum68(diff);
                        double c = (s < p->tphi) ? (double)s : (double)p->tphi;   /* python min(tphi, s) :179 */
                        lcosts[pix * L + base + qq] = c;
                        if (c < mindists[pix]) { mindists[pix] = c; bestlabels[pix] = base + qq; } /* :181-184 */
                    }
                    nprop[pix] = base + K;  /* :189 */
                }
            }
        }
    free(mindists);
}

/* ================================================================================================
 * Counter-based RNG for the neighbour sampler (build addition; reference uses unseeded MT19937, :219,221).
 * Philox4x32-10, key = seed, counter = (pixel index, attempt, 0, 0); word 0 drives the y draw, word 1 the
 * x draw.  A 32-bit uniform u maps to the integer offset o = floor(sigma * z) through the thresholds
 * thr[i] = floor(Phi((i - 63) / sigma) * 2^32) (i = 0..126, clamped to 2^32-1): o = -64 + #{i : u >= thr[i]}.
 * ============================================================================================== */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *out)
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox(uint32_t c0, uint32_t c1, uint64_t seed, uint32_t *out)
{
    philox4x32_10(c0, c1, 0, 0, (uint32_t)seed, (uint32_t)(seed >> 32), out);
}

void orc_gauss_thresholds(double sigma, uint32_t *thr /*127*/)
{
    for (int i = 0; i < 127; i++) {
        double phi = 0.5 * erfc(-((i - 63) / sigma) / sqrt(2.0));
        double v = floor(phi * 4294967296.0);
        thr[i] = v >= 4294967295.0 ? 4294967295u : (uint32_t)v;
    }
}

int orc_gauss_offset(const uint32_t *thr, uint32_t u)
{
    int lo = 0, hi = 127; /* count of thresholds <= u; thresholds are non-decreasing */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (u >= thr[mid]) lo = mid + 1; else hi = mid; }
    return -64 + lo;
}

/* int(np.random.normal(c, sigma)) for integer c >= 0 (daisy i flann.py:219,221): truncation toward zero */
static inline int trunc_draw(int c, int off) { int t = c + off; return t < 0 ? t + 1 : t; }

/* python slice a[start:stop] on a length-n axis (step 1), both possibly negative/out of range */
static inline void py_slice(int start, int stop, int n, int *lo, int *hi)
{
    if (start < 0) { start += n; if (start < 0) start = 0; } else if (start > n) start = n;
    if (stop < 0) { stop += n; if (stop < 0) stop = 0; } else if (stop > n) stop = n;
    *lo = start; *hi = stop > start ? stop : start;
}

/* 'tv in proposals[y,x,lo:hi]' == (rows == tv).any(): component-wise (Q5) */
static inline int tv_in(const int64_t *prow, int lo, int hi, int64_t tv0, int64_t tv1)
{
    for (int i = lo; i < hi; i++) if (prow[2 * i] == tv0 || prow[2 * i + 1] == tv1) return 1;
    return 0;
}

/* nasumicni, daisy i flann.py:205-233.  attempts_out (optional, may be NULL): draws used per pixel. */
void orc_neighbour_proposals(const orc_params *p, const float *d1, const float *d2,
                             int64_t *proposals, double *lcosts, int64_t *nprop, const int64_t *bestlabels,
                             int32_t *attempts_out)
{
    int H = p->pich, W = p->picw, L = p->maxnprop, K = p->knn, ncy = ncelly_of(p);
    uint32_t thr[127];
    orc_gauss_thresholds((double)p->sigma, thr);
    /* pixels are independent: a draw reads the sampled neighbour's WTA proposal (a kNN slot, never rewritten here) */
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            uint32_t rnd[4];
            float diff[ORC_DESC];
            size_t pix = (size_t)y * W + x;
            int mincellyl = cell_y(p, y) - p->window; if (mincellyl < 0) mincellyl = 0;            /* :212 */
            int ncellyl = (ncy < cell_y(p, y) + p->window ? ncy : cell_y(p, y) + p->window) - mincellyl; /* :214 */
            int mincellxl = cell_x(p, x) - p->window; if (mincellxl < 0) mincellxl = 0;            /* :215 */
            int64_t *prow = proposals + pix * L * 2;
            int i = 0, ngaussprop = 0;
            uint32_t a = 0;
            for (; i < p->ngauss && a < (uint32_t)p->max_attempts; a++) {
                orc_philox((uint32_t)pix, a, p->seed, rnd);
                int tgy = trunc_draw(y, orc_gauss_offset(thr, rnd[0]));                       /* :219 */
                if (tgy < 0 || tgy >= H) continue;                                              /* :220 */
                int tgx = trunc_draw(x, orc_gauss_offset(thr, rnd[1]));                       /* :221 */
                if (tgx < 0 || tgx >= W) continue;                                              /* :222 */
                int broj = K * ((cell_y(p, tgy) - mincellyl) + (cell_x(p, tgx) - mincellxl) * ncellyl); /* :223-224 */
                size_t tpix = (size_t)tgy * W + tgx;
                int64_t bl = bestlabels[tpix];
                int64_t tv0 = proposals[(tpix * L + bl) * 2], tv1 = proposals[(tpix * L + bl) * 2 + 1]; /* :225 */
                int lo, hi, lo2, hi2, np_ = (int)nprop[pix];
                py_slice(broj, broj + K, L, &lo, &hi);
                py_slice(np_ - ngaussprop, np_, L, &lo2, &hi2);
                if (!tv_in(prow, lo, hi, tv0, tv1) && !tv_in(prow, lo2, hi2, tv0, tv1)) {      /* :226 */
                    prow[2 * np_] = tv0; prow[2 * np_ + 1] = tv1;                              /* :227 */
                    const float *q = d1 + pix * ORC_DESC, *t = d2 + tpix * ORC_DESC;
                    for (int k = 0; k < ORC_DESC; k++) diff[k] = q[k] - t[k];
                    float s = fabsf(np_pairwise_sum68(diff));                                   /* :228-229 (Q6) */
                    lcosts[pix * L + np_] = (s < p->tphi) ? (double)s : (double)p->tphi;   /* python min(tphi, s) */
                    nprop[pix] = np_ + 1; ngaussprop++;                                         /* :230-231 */
                }
                i++;                                                                            /* :233 */
            }
            if (attempts_out) attempts_out[pix] = (int32_t)a;
        }
}

/* ================================================================================================
 * pakovanje, daisy i flann.py:256-309: compat bit-matrices ksets[tl,nl] = (tpsi > |dy-dy'|+|dx-dx'|),
 * slot 0 = pixel below, slot 1 = pixel to the right, np.packbits big-endian.  Faithful, including
 * the fact that the scratch matrix is not cleared inside the two border loops (:290-307).
 * packed: (H,W,2,kdim) uint8, kdim = L*L/8+1 (:98), must be zero-initialised by the caller.
 * ============================================================================================== */
static void packbits_row(const uint8_t *bits, size_t nbits, uint8_t *out)
{
    size_t nbytes = (nbits + 7) / 8;
    for (size_t b = 0; b < nbytes; b++) {
        uint8_t v = 0;
        for (int j = 0; j < 8; j++) { size_t i = b * 8 + j; v = (uint8_t)((v << 1) | (i < nbits ? bits[i] : 0)); }
        out[b] = v;
    }
}

static void fill_ksets(const orc_params *p, const int64_t *proposals, const int64_t *nprop,
                       int ty, int tx, int ny, int nx, uint8_t *ks)
{
    int W = p->picw, L = p->maxnprop;
    size_t a = (size_t)ty * W + tx, b = (size_t)ny * W + nx;
    for (int tl = 0; tl < nprop[a]; tl++) {
        int64_t v0 = proposals[(a * L + tl) * 2], v1 = proposals[(a * L + tl) * 2 + 1];
        for (int nl = 0; nl < nprop[b]; nl++) {
            int64_t raz = llabs(proposals[(b * L + nl) * 2] - v0) + llabs(proposals[(b * L + nl) * 2 + 1] - v1); /* purepsi :114 */
            ks[(size_t)tl * L + nl] = (uint8_t)(p->tpsi > raz);
        }
    }
}

void orc_pack_compat(const orc_params *p, const int64_t *proposals, const int64_t *nprop, uint8_t *packed)
{
    int H = p->pich, W = p->picw, L = p->maxnprop;
    size_t LL = (size_t)L * L, kdim = LL / 8 + 1;
    uint8_t *ks0 = (uint8_t *)calloc(LL, 1), *ks1 = (uint8_t *)calloc(LL, 1);
    for (int ty = 0; ty < H - 1; ty++)
        for (int tx = 0; tx < W - 1; tx++) {
            fill_ksets(p, proposals, nprop, ty, tx, ty + 1, tx, ks0);
            fill_ksets(p, proposals, nprop, ty, tx, ty, tx + 1, ks1);
            packbits_row(ks0, LL, packed + (((size_t)ty * W + tx) * 2 + 0) * kdim);
            packbits_row(ks1, LL, packed + (((size_t)ty * W + tx) * 2 + 1) * kdim);
            memset(ks0, 0, LL); memset(ks1, 0, LL);   /* :289 */
        }
    for (int tx = 0; tx < W - 1; tx++) {              /* :290-297, ks1 not cleared between pixels */
        fill_ksets(p, proposals, nprop, H - 1, tx, H - 1, tx + 1, ks1);
        packbits_row(ks1, LL, packed + (((size_t)(H - 1) * W + tx) * 2 + 1) * kdim);
    }
    for (int ty = 0; ty < H - 1; ty++) {              /* :298-307, ks0 not cleared between pixels */
        fill_ksets(p, proposals, nprop, ty, W - 1, ty + 1, W - 1, ks0);
        packbits_row(ks0, LL, packed + (((size_t)ty * W + W - 1) * 2 + 0) * kdim);
    }
    free(ks0); free(ks1);
}

/* ================================================================================================
 * BCD: sidepsi python bcd.py:84-88, bcd :101-257, ceoBCD :261-284.
 * The compat test is evaluated on the fly (identical to reading packedksets inside label ranges).
 * ============================================================================================== */
static inline int64_t sidepsi(const orc_params *p, const int64_t *proposals, const int64_t *bestlabels,
                              int y1, int x1, int l1, int y2, int x2)
{
    if (y2 >= 0 && y2 < p->pich && x2 >= 0 && x2 < p->picw) {
        int L = p->maxnprop;
        size_t a = (size_t)y1 * p->picw + x1, b = (size_t)y2 * p->picw + x2;
        const int64_t *pa = proposals + (a * L + l1) * 2, *pb = proposals + (b * L + bestlabels[b]) * 2;
        int64_t s = llabs(pa[0] - pb[0]) + llabs(pa[1] - pb[1]);
        return s < p->tpsi ? s : p->tpsi;
    }
    return 0;
}

/* one chain; bestlabels updated in place (python bcd.py:101-257) */
void orc_bcd_chain(const orc_params *p, const int64_t *proposals, const double *lcosts, const int64_t *nprop,
                   int64_t *bestlabels, int ystep, int xstep, int ty, int tx)
{
    int H = p->pich, W = p->picw, L = p->maxnprop;
    int xside = ystep == 0 ? 1 : 0, yside = ystep == 0 ? 0 : 1;      /* :107-112 */
    int maxlen = (H > W ? H : W) + 1;
    double *dp = (double *)malloc((size_t)maxlen * L * sizeof(double));
    int32_t *past = (int32_t *)malloc((size_t)maxlen * L * sizeof(int32_t));
    int i = 0;
    size_t pix = (size_t)ty * W + tx;
    for (int tl = 0; tl < nprop[pix]; tl++)                           /* :118-120 */
        dp[tl] = (double)(sidepsi(p, proposals, bestlabels, ty, tx, tl, ty + yside, tx + xside) +
                          sidepsi(p, proposals, bestlabels, ty, tx, tl, ty - yside, tx - xside)) +
                 p->lamda * lcosts[pix * L + tl];
    for (;;) {
        ty += ystep; tx += xstep; i++;                                /* :124-126 */
        if (tx < 0 || ty < 0 || tx >= W || ty >= H) break;            /* :127 */
        pix = (size_t)ty * W + tx;
        size_t ppix = (size_t)(ty - ystep) * W + (tx - xstep);
        int tnprop = (int)nprop[pix], pnprop = (int)nprop[ppix];
        const double *dprev = dp + (size_t)(i - 1) * L;
        double permmincost = 800000.0; int permminlabel = -8;        /* :152-157 */
        for (int tk = 0; tk < pnprop; tk++)
            if ((double)p->tpsi + dprev[tk] < permmincost) { permmincost = (double)p->tpsi + dprev[tk]; permminlabel = tk; }
        for (int tl = 0; tl < tnprop; tl++) {                         /* :159-219 (both direction branches agree) */
            double smallcosts = p->lamda * lcosts[pix * L + tl] +
                                (double)sidepsi(p, proposals, bestlabels, ty, tx, tl, ty + yside, tx + xside) +
                                (double)sidepsi(p, proposals, bestlabels, ty, tx, tl, ty - yside, tx - xside);
            double mincost = permmincost; int pl = permminlabel;
            int64_t af = proposals[(pix * L + tl) * 2], bf = proposals[(pix * L + tl) * 2 + 1];
            int found = 0;
            for (int tk = 0; tk < pnprop; tk++) {
                int64_t raz = llabs(proposals[(ppix * L + tk) * 2] - af) + llabs(proposals[(ppix * L + tk) * 2 + 1] - bf);
                if (!(p->tpsi > raz)) continue;                       /* compat bit (Q8) */
                double c = dprev[tk] + (double)raz;                   /* :171-172 */
                if (!found || c < mincost) { mincost = c; pl = tk; found = 1; }   /* np.min / np.argmin: first minimum (Q9) */
            }
            dp[(size_t)i * L + tl] = mincost + smallcosts;            /* :176 */
            past[(size_t)i * L + tl] = pl;
        }
    }
    ty -= ystep; tx -= xstep; i--;                                    /* :228-230 */
    pix = (size_t)ty * W + tx;
    double mincost = 800000.0; int minlabel = 0;
    for (int tl = 0; tl < nprop[pix]; tl++)
        if (dp[(size_t)i * L + tl] < mincost) { mincost = dp[(size_t)i * L + tl]; minlabel = tl; }
    bestlabels[pix] = minlabel;
    int pl = minlabel;
    for (;;) {                                                        /* :239-253 */
        ty -= ystep; tx -= xstep;
        if (tx < 0 || ty < 0 || tx >= W || ty >= H) break;
        pl = past[(size_t)i * L + pl];
        i--;
        bestlabels[(size_t)ty * W + tx] = pl;
    }
    free(dp); free(past);
}

/* one phase of ceoBCD (python bcd.py:265-277): phase 0 even columns down, 1 even rows leftwards,
 * 2 odd columns up, 3 odd rows rightwards */
void orc_bcd_phase(const orc_params *p, const int64_t *proposals, const double *lcosts, const int64_t *nprop,
                   int64_t *bestlabels, int phase)
{
    int H = p->pich, W = p->picw;
    /* the chains of a phase read and write disjoint image lines (their side terms read the other parity only) */
    if (phase == 0) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
        for (int x = 0; x < W; x += 2) orc_bcd_chain(p, proposals, lcosts, nprop, bestlabels, 1, 0, 0, x);
    }
    if (phase == 1) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
        for (int y = 0; y < H; y += 2) orc_bcd_chain(p, proposals, lcosts, nprop, bestlabels, 0, -1, y, W - 1);
    }
    if (phase == 2) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
        for (int x = (W / 2) * 2 - 1; x > -1; x -= 2) orc_bcd_chain(p, proposals, lcosts, nprop, bestlabels, -1, 0, H - 1, x);
    }
    if (phase == 3) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
        for (int y = (H / 2) * 2 - 1; y > -1; y -= 2) orc_bcd_chain(p, proposals, lcosts, nprop, bestlabels, 0, 1, y, 0);
    }
}

void orc_bcd_sweep(const orc_params *p, const int64_t *proposals, const double *lcosts, const int64_t *nprop,
                   int64_t *bestlabels)
{
    for (int ph = 0; ph < 4; ph++) orc_bcd_phase(p, proposals, lcosts, nprop, bestlabels, ph);
}

/* vratiKonacniFlow, python bcd.py:90-95 / daisy i flann.py:192-197: (H,W,2) float64 [dy,dx] */
void orc_labels_to_flow(const orc_params *p, const int64_t *proposals, const int64_t *bestlabels, double *flow)
{
    size_t N = (size_t)p->pich * p->picw; int L = p->maxnprop;
    for (size_t i = 0; i < N; i++) {
        flow[2 * i] = (double)proposals[(i * L + bestlabels[i]) * 2];
        flow[2 * i + 1] = (double)proposals[(i * L + bestlabels[i]) * 2 + 1];
    }
}

/* ================================================================================================
 * Forward/backward consistency: postprocessing.py:7-17 (load [dy,dx] -> [U,V,valid] float32),
 * :79-117 (check, with the transposed indexing Q13).  fwd/bwd: (H,W,2) float64 [dy,dx];
 * out: (H,W,3) float32.
 * ============================================================================================== */
void orc_fb_consistency(int H, int W, const double *fwd, const double *bwd, double tresh, float *out)
{
    size_t N = (size_t)H * W;
    float *f2 = (float *)malloc(N * 3 * sizeof(float));
    for (size_t i = 0; i < N; i++) {
        out[3 * i] = (float)fwd[2 * i + 1]; out[3 * i + 1] = (float)fwd[2 * i]; out[3 * i + 2] = 1.0f;
        f2[3 * i] = (float)bwd[2 * i + 1]; f2[3 * i + 1] = (float)bwd[2 * i]; f2[3 * i + 2] = 1.0f;
    }
    /* flow.shape = (H,W,3) is unpacked as "width, height" (:80): u1 runs over rows, v1 over columns */
    int width = H, height = W;
    for (int u1 = 0; u1 < width; u1++)
        for (int v1 = 0; v1 < height; v1++) {
            float *f = out + ((size_t)u1 * W + v1) * 3;
            if (!(f[2] > 0.5f)) continue;
            int u2 = (int)(f[0] + (float)u1);      /* np.float32 + python int -> float32, int() truncates (:87) */
            int v2 = (int)(f[1] + (float)v1);
            if (u2 < 0 || v2 < 0 || u2 >= width || v2 >= height) { f[0] = f[1] = f[2] = 0.0f; continue; }
            const float *g = f2 + ((size_t)u2 * W + v2) * 3;
            if (!(g[2] > 0.5f)) { f[0] = f[1] = f[2] = 0.0f; continue; }
            float du = f[0] + g[0], dv = f[1] + g[1];
            float err = sqrtf(dv * dv + du * du);
            if ((double)err > tresh) { f[0] = f[1] = f[2] = 0.0f; }
        }
    free(f2);
}

int orc_version(void) { return 1; }
