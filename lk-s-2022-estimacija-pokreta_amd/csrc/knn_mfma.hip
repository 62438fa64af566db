// K3 knn_cells on the matrix cores (default path of dflow_knn_proposals).
//
// Same result as the exact VALU search (knn.hip) -- the canonical 5-NN of generisi (daisy i flann.py:157-189),
// bit for bit -- but the 1.7e10 descriptor pairs of a Sintel pass are screened by f16 MFMA instead of being
// evaluated one by one:
//
//   prep      both images -> rows of 80 f16: 68 scaled descriptor values (alpha = 64), then for image 2 three f16
//             pieces of h = 0.5*|c~|^2 (for image 1: -1,-1,-1), zero padding; |q~|, |c~| and the per-cell max |c~|.
//             One MFMA chain then yields t(q,c) = q~.c~ - 0.5|c~|^2 = 0.5(|q~|^2 - d^2) up to a rigorous error
//             eps(q, cell) = 1.25 * 2^-10 * (|q~| C + C^2/2) (+ tiny absolute term), C = max |c~| in the cell
//             (f16 rounding of both operands, f32 accumulation, rounding of the canonical distance).
//   pass 1    t for every (query, candidate) of a (256-query block, candidate cell); every lane keeps the 5 largest
//             maxima of its 16-value tile columns -> a lower bound a5 of the 5th largest t of its query.
//   pass 2    t again; every candidate with t >= a5 - 2 eps is an "event" -- only those can be among the exact 5 NN.
//   resolve   events (about 18 of 1728 candidates per query) get the canonical float32 distance (sequential fmaf
//             chain) and are merged into the query's exact top-5 by cascaded 64-bit LDS atomic minima on
//             (distance bits, index) keys, which is exactly the canonical (distance, index) order.
//   emit      proposals [dy,dx] and truncated L1 costs (numpy order) into the cell's 5 slots (Q1-Q3);
//             knn_finalize_kernel then sets nprop, the WTA label (first minimum, Q4) and the fills.
//
// MFMA layout (v_mfma_f32_32x32x16_f16): A = candidates (rows), B = queries (columns): lane l holds
// A[row l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col l&31]; D: col = l&31, row = (r&3) + 8(r>>2) + 4(l>>5).
// A workgroup is 4 waves x 64 queries (2 column groups) of one image-1 cell; candidates stream through LDS in
// chunks of 96 rows (176-byte pitch: conflict-free ds_read_b128), double buffered.
#include "dflow_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KM_ALPHA 64.0f
#define KM_K 80                 // f16 per prepared row (160 bytes)
#define KM_THREADS 256
#define KM_WAVES 4
#define KM_QPW 64               // queries per wave: 2 column groups of 32
#define KM_QPB (KM_WAVES * KM_QPW)
#define KM_CHUNK 96             // candidates per LDS chunk (3 tiles of 32)
#define KM_PITCH 176            // LDS row pitch in bytes (44 dwords: 16 consecutive rows hit 16 distinct 4-bank groups)
#define KM_EVCAP 2048           // events per wave and candidate cell
#define KM_MAXPTS 4096          // candidate index must fit 12 bits

struct KmGeom {
    Geom g;
    int LP, qchunks;
    float tphi;
};

// ------------------------------------------------------------------------------------------------ prep
// one thread per pixel of one image; which = 0: image 1 (queries), 1: image 2 (candidates)
__global__ void knn_prep_kernel(const float *__restrict__ d, _Float16 *__restrict__ h, float *__restrict__ nrm,
                                unsigned int *__restrict__ cellmax, int *__restrict__ flags, Geom g, int which)
{
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= g.H * g.W) return;
    const float4 *s = reinterpret_cast<const float4 *>(d + (size_t)pix * DFLOW_DESC);
    _Float16 row[KM_K];
    float ss = 0.0f;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < DFLOW_DESC / 4; k++) {
        float4 v = s[k];
        float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float sc = KM_ALPHA * e[j];
            bad |= !(fabsf(sc) < 60000.0f);          // also catches NaN
            _Float16 hv = (_Float16)sc;
            row[4 * k + j] = hv;
            float f = (float)hv;
            ss = ss + f * f;
        }
    }
    if (which == 0) {
        row[68] = (_Float16)-1.0f; row[69] = (_Float16)-1.0f; row[70] = (_Float16)-1.0f;
    } else {
        float hc = 0.5f * ss;
        bad |= !(hc < 60000.0f);
        _Float16 p1 = (_Float16)hc;
        float r1 = hc - (float)p1;
        _Float16 p2 = (_Float16)r1;
        float r2 = r1 - (float)p2;
        _Float16 p3 = (_Float16)r2;
        row[68] = p1; row[69] = p2; row[70] = p3;
    }
#pragma unroll
    for (int k = 71; k < KM_K; k++) row[k] = (_Float16)0.0f;
    float4 *o = reinterpret_cast<float4 *>(h + (size_t)pix * KM_K);
    const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int k = 0; k < KM_K * 2 / 16; k++) o[k] = r4[k];
    float n = sqrtf(ss) * 1.0001f;                   // a slight over-estimate of |x~| keeps eps on the safe side
    nrm[pix] = n;
    if (which == 1) atomicMax(&cellmax[g.celly(pix / g.W) * g.ncx + g.cellx(pix % g.W)], __float_as_uint(n));
    if (bad) atomicOr(flags, 1);
}

// ------------------------------------------------------------------------------------------------ main kernel
struct KmPtrs {
    const float *d1, *d2;          // float32 descriptors (exact distances, costs)
    const _Float16 *h1, *h2;       // prepared f16 rows
    const float *qn;               // |q~| per image-1 pixel
    const unsigned int *cellmax;   // max |c~| per image-2 cell (float bits)
    uint32_t *proposals;
    float *lcosts;
    int *ovf_count;                // overflow list: entries (qcell, qchunk, ci, cj) for knn_fix_kernel
    int4 *ovf_list;
    int ovf_cap;
};

__device__ static inline float exact_dist(const float *__restrict__ q, const float *__restrict__ c)
{
    const float4 *q4 = reinterpret_cast<const float4 *>(q), *c4 = reinterpret_cast<const float4 *>(c);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < DFLOW_DESC / 4; k++) {
        float4 u = q4[k], v = c4[k];
        float e;
        e = u.x - v.x; acc = __fmaf_rn(e, e, acc);
        e = u.y - v.y; acc = __fmaf_rn(e, e, acc);
        e = u.z - v.z; acc = __fmaf_rn(e, e, acc);
        e = u.w - v.w; acc = __fmaf_rn(e, e, acc);
    }
    return acc;
}

__device__ static inline void top5_insert_desc(float (&a)[5], float m)
{
#pragma unroll
    for (int i = 0; i < 5; i++) { float hi = fmaxf(a[i], m); m = fminf(a[i], m); a[i] = hi; }
}

__device__ static inline float max16(const f32x16 &v)
{
    float m0 = fmaxf(fmaxf(v[0], v[1]), v[2]), m1 = fmaxf(fmaxf(v[3], v[4]), v[5]);
    float m2 = fmaxf(fmaxf(v[6], v[7]), v[8]), m3 = fmaxf(fmaxf(v[9], v[10]), v[11]);
    float m4 = fmaxf(fmaxf(v[12], v[13]), v[14]);
    return fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), fmaxf(m4, v[15]));
}

__global__ void __launch_bounds__(KM_THREADS, 2) knn_mfma_kernel(KmGeom a, KmPtrs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *abuf = smem;                                                               // [2][KM_CHUNK][KM_PITCH]
    uint32_t *evlist = reinterpret_cast<uint32_t *>(smem + 2 * KM_CHUNK * KM_PITCH);   // [KM_WAVES][KM_EVCAP]
    unsigned long long *top = reinterpret_cast<unsigned long long *>(evlist + KM_WAVES * KM_EVCAP);   // [KM_QPB][5]
    int *qpix_lds = reinterpret_cast<int *>(top + KM_QPB * 5);                          // [KM_QPB]

    const Geom g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    // ---- which (query cell, query chunk, candidate cell)
    const int win = 2 * g.win + 1;
    int b = blockIdx.x;
    const int wslot = b % (win * win); b /= win * win;
    const int qchunk = b % a.qchunks; const int qcell = b / a.qchunks;
    const int qci = qcell % g.ncx, qcj = qcell / g.ncx;
    const int qx0 = g.x0(qci), qy0 = g.y0(qcj), qcw = g.x1(qci) - qx0, qnpts = qcw * (g.y1(qcj) - qy0);
    if (qchunk * KM_QPB >= qnpts) return;
    const int cimin = max(0, qci - g.win), cimax = min(g.ncx - 1, qci + g.win);
    const int cjmin = max(0, qcj - g.win), cjmax = min(g.ncy - 1, qcj + g.win);
    const int ncyw = cjmax - cjmin + 1;
    const int ci = cimin + wslot / ncyw, cj = cjmin + wslot % ncyw;      // reference order: ci outer, cj inner (Q2)
    if (ci > cimax) return;
    const int slot_base = 5 * wslot;
    const int cx0 = g.x0(ci), cy0 = g.y0(cj), ccw = g.x1(ci) - cx0, cnpts = ccw * (g.y1(cj) - cy0);
    const float C = __uint_as_float(p.cellmax[cj * g.ncx + ci]);

    // ---- my queries: group gq (0/1), column col -> in-cell index, pixel; B fragments and eps
    half8 bfrag[2][5];
    float eps[2];
    int qpix[2];
#pragma unroll
    for (int gq = 0; gq < 2; gq++) {
        int qi = qchunk * KM_QPB + wave * KM_QPW + gq * 32 + col;
        if (qi >= qnpts) qi = qnpts - 1;                                 // inactive columns shadow the last query
        qpix[gq] = (qy0 + qi / qcw) * g.W + qx0 + qi % qcw;
        const half8 *src = reinterpret_cast<const half8 *>(p.h1 + (size_t)qpix[gq] * KM_K);
#pragma unroll
        for (int s = 0; s < 5; s++) bfrag[gq][s] = src[2 * s + half];
        if (half == 0) qpix_lds[wave * KM_QPW + gq * 32 + col] = qpix[gq];
        float qn = p.qn[qpix[gq]];
        eps[gq] = 1.25f * 0.0009765625f * (qn * C + 0.5f * C * C) + 1e-5f * (qn + C) + 1e-6f;
    }

    const int nchunks = (cnpts + KM_CHUNK - 1) / KM_CHUNK;
    // cooperative staging of one chunk: 96 rows x 10 pieces of 16 bytes
    auto stage = [&](int chunk, int buf) {
        for (int piece = tid; piece < KM_CHUNK * 10; piece += KM_THREADS) {
            int r = piece / 10, part = piece % 10;
            int idx = chunk * KM_CHUNK + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < cnpts) {
                int cpix = (cy0 + idx / ccw) * g.W + cx0 + idx % ccw;
                v = reinterpret_cast<const float4 *>(p.h2 + (size_t)cpix * KM_K)[part];
            }
            *reinterpret_cast<float4 *>(abuf + (size_t)buf * KM_CHUNK * KM_PITCH + r * KM_PITCH + part * 16) = v;
        }
    };

    float a5[2][5];
    float thr[2] = {0.f, 0.f};
    int evcount = 0;           // wave-uniform
    bool overflow = false;     // wave-uniform
    uint32_t *myev = evlist + wave * KM_EVCAP;

#pragma unroll
    for (int gq = 0; gq < 2; gq++)
#pragma unroll
        for (int i = 0; i < 5; i++) a5[gq][i] = -INFINITY;

    for (int pass = 0; pass < 2; pass++) {
        stage(0, 0);
        __syncthreads();
        for (int chunk = 0; chunk < nchunks; chunk++) {
            const int buf = chunk & 1;
            if (chunk + 1 < nchunks) stage(chunk + 1, buf ^ 1);          // the other buffer was released by the barrier below
            const char *ab = abuf + (size_t)buf * KM_CHUNK * KM_PITCH;
#pragma unroll 1
            for (int tile = 0; tile < KM_CHUNK / 32; tile++) {
                const int cbase = chunk * KM_CHUNK + tile * 32;
                if (cbase >= cnpts) break;
                half8 af[5];
                const char *arow = ab + (tile * 32 + col) * KM_PITCH + half * 16;
#pragma unroll
                for (int s = 0; s < 5; s++) af[s] = *reinterpret_cast<const half8 *>(arow + s * 32);
#pragma unroll
                for (int gq = 0; gq < 2; gq++) {
                    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 5; s++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bfrag[gq][s], acc, 0, 0, 0);
                    if (pass == 0) {
                        if (cbase + 32 > cnpts) {     // last, partial tile: rows beyond the cell must not enter the maxima
#pragma unroll
                            for (int r = 0; r < 16; r++)
                                if (cbase + (r & 3) + 8 * (r >> 2) + 4 * half >= cnpts) acc[r] = -INFINITY;
                        }
                        top5_insert_desc(a5[gq], max16(acc));
                    } else {
                        const float th = thr[gq];
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int idx = cbase + (r & 3) + 8 * (r >> 2) + 4 * half;
                            const bool ev = acc[r] >= th && idx < cnpts;
                            const unsigned long long m = __ballot(ev);
                            if (m) {
                                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                const int slot = evcount + rank;
                                if (ev) {
                                    if (slot < KM_EVCAP) myev[slot] = ((uint32_t)(gq * 32 + col) << 12) | (uint32_t)idx;
                                }
                                evcount += __popcll(m);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (pass == 0) {
            // merge the two half-lanes of every query, threshold = 5th largest tile-column maximum - 2 eps
#pragma unroll
            for (int gq = 0; gq < 2; gq++) {
                float o[5];
#pragma unroll
                for (int i = 0; i < 5; i++) o[i] = __shfl_xor(a5[gq][i], 32);
#pragma unroll
                for (int i = 0; i < 5; i++) top5_insert_desc(a5[gq], o[i]);
                thr[gq] = a5[gq][4] - 2.0f * eps[gq];
            }
        }
    }
    overflow = evcount > KM_EVCAP;

    // ---- resolve: canonical float32 distance of every event, cascaded atomic minima into the owner's top-5
    unsigned long long *mytop = top + (size_t)wave * KM_QPW * 5;
    for (int i = lane; i < KM_QPW * 5; i += 64) mytop[i] = ~0ull;
    __builtin_amdgcn_wave_barrier();
    const int nev = overflow ? 0 : evcount;
    for (int e = lane; e < nev; e += 64) {
        const uint32_t key = myev[e];
        const int owner = key >> 12, idx = key & 0xFFF;
        const int opix = qpix_lds[wave * KM_QPW + owner];
        const int cpix = (cy0 + idx / ccw) * g.W + cx0 + idx % ccw;
        const float dist = exact_dist(p.d1 + (size_t)opix * DFLOW_DESC, p.d2 + (size_t)cpix * DFLOW_DESC);
        unsigned long long k64 = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)idx;
        unsigned long long *t5 = mytop + owner * 5;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            unsigned long long old = atomicMin(&t5[j], k64);
            k64 = old > k64 ? old : k64;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();

    if (overflow) {
        if (lane == 0) {
            int pos = atomicAdd(p.ovf_count, 1);
            if (pos < p.ovf_cap) p.ovf_list[pos] = make_int4(qcell, qchunk * KM_QPB + wave * KM_QPW, ci, cj);
        }
        return;
    }
    // ---- emit (daisy i flann.py:174-180): lane = query (group lane>>5, column lane&31)
    {
        const int qi = qchunk * KM_QPB + wave * KM_QPW + lane;
        const int mypix = qpix_lds[wave * KM_QPW + lane];
        if (qi < qnpts) {
            const int qy = mypix / g.W, qx = mypix % g.W;
            const unsigned long long *t5 = mytop + lane * 5;
            for (int j = 0; j < 5; j++) {
                const int idx = (int)(t5[j] & 0xFFFFFFFFu);
                const int ty = cy0 + idx / ccw, tx = cx0 + idx % ccw;
                const float s = l1_cost_np(p.d1 + (size_t)mypix * DFLOW_DESC, p.d2 + ((size_t)ty * g.W + tx) * DFLOW_DESC);
                p.proposals[(size_t)mypix * a.LP + slot_base + j] = pack_flow(ty - qy, tx - qx);
                p.lcosts[(size_t)mypix * a.LP + slot_base + j] = s < a.tphi ? s : a.tphi;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ finalize
// nprop = 5 x window cells (daisy i flann.py:189), WTA label = first minimum of the costs with strict '<'
// from 1000.0 (:93,181-184), fills beyond nprop (:89-90)
__global__ void knn_finalize_kernel(Geom g, int LP, uint32_t *__restrict__ proposals, float *__restrict__ lcosts,
                                    int32_t *__restrict__ nprop, int32_t *__restrict__ bestlabels)
{
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= g.H * g.W) return;
    const int cy = g.celly(pix / g.W), cx = g.cellx(pix % g.W);
    const int wy = min(g.ncy - 1, cy + g.win) - max(0, cy - g.win) + 1;
    const int wx = min(g.ncx - 1, cx + g.win) - max(0, cx - g.win) + 1;
    const int n = 5 * wy * wx;
    const float *lc = lcosts + (size_t)pix * LP;
    float mind = 1000.0f; int best = 0;
    for (int l = 0; l < n; l++) { float c = lc[l]; if (c < mind) { mind = c; best = l; } }
    nprop[pix] = n;
    bestlabels[pix] = best;
    for (int l = n; l < LP; l++) { proposals[(size_t)pix * LP + l] = DFLOW_FILL_PROPOSAL; lcosts[(size_t)pix * LP + l] = DFLOW_FILL_COST; }
}

// ------------------------------------------------------------------------------------------------ host side
size_t knn_mfma_ws_bytes(const dflow_params *p)
{
    size_t N = (size_t)p->pich * p->picw;
    size_t ncells = (size_t)(p->picw / p->cellw) * (p->pich / p->cellh);
    return 2 * N * KM_K * sizeof(_Float16) + 2 * N * sizeof(float) + ncells * sizeof(unsigned) + 256 + 4096 * sizeof(int4) + 256;
}

bool knn_mfma_supported(const dflow_params *p)
{
    Geom g = make_geom(p);
    int maxpts = (g.x1(g.ncx - 1) - g.x0(g.ncx - 1)) * (g.y1(g.ncy - 1) - g.y0(g.ncy - 1));
    return maxpts <= KM_MAXPTS && p->window >= 0 && p->window <= 2;
}

int launch_knn_fix(const dflow_params *p, const float *d1, const float *d2, uint32_t *proposals, float *lcosts,
                   const int *ovf_count, const int4 *ovf_list, int ovf_cap, const int *flags, hipStream_t s);

int launch_knn_mfma(const dflow_params *p, const float *d1, const float *d2, uint32_t *proposals, float *lcosts,
                    int32_t *nprop, int32_t *bestlabels, void *ws, hipStream_t s)
{
    Geom g = make_geom(p);
    size_t N = (size_t)g.H * g.W, ncells = (size_t)g.ncx * g.ncy;
    char *w = (char *)ws;
    _Float16 *h1 = (_Float16 *)w; w += N * KM_K * sizeof(_Float16);
    _Float16 *h2 = (_Float16 *)w; w += N * KM_K * sizeof(_Float16);
    float *qn = (float *)w; w += N * sizeof(float);
    float *cn = (float *)w; w += N * sizeof(float);
    unsigned *cellmax = (unsigned *)w; w += ncells * sizeof(unsigned);
    w = (char *)(((uintptr_t)w + 255) & ~(uintptr_t)255);
    int *ctr = (int *)w; w += 256;              // ctr[0] = overflow count, ctr[1] = flags
    int4 *ovf = (int4 *)w;
    const int ovf_cap = 4096;
    if (hipMemsetAsync(cellmax, 0, ncells * sizeof(unsigned), s) != hipSuccess || hipMemsetAsync(ctr, 0, 256, s) != hipSuccess)
        return dflow_set_error(DFLOW_EHIP, "hipMemsetAsync failed in launch_knn_mfma");
    int nb = (int)((N + 255) / 256);
    hipLaunchKernelGGL(knn_prep_kernel, dim3(nb), dim3(256), 0, s, d1, h1, qn, cellmax, ctr + 1, g, 0);
    hipLaunchKernelGGL(knn_prep_kernel, dim3(nb), dim3(256), 0, s, d2, h2, cn, cellmax, ctr + 1, g, 1);

    KmGeom a;
    a.g = g; a.LP = p->label_pitch; a.tphi = p->tphi;
    int maxpts = (g.x1(g.ncx - 1) - g.x0(g.ncx - 1)) * (g.y1(g.ncy - 1) - g.y0(g.ncy - 1));
    a.qchunks = (maxpts + KM_QPB - 1) / KM_QPB;
    KmPtrs q;
    q.d1 = d1; q.d2 = d2; q.h1 = h1; q.h2 = h2; q.qn = qn; q.cellmax = cellmax; q.proposals = proposals; q.lcosts = lcosts;
    q.ovf_count = ctr; q.ovf_list = ovf; q.ovf_cap = ovf_cap;
    int win = 2 * g.win + 1;
    int nblocks = g.ncx * g.ncy * a.qchunks * win * win;
    size_t shmem = 2 * KM_CHUNK * KM_PITCH + KM_WAVES * KM_EVCAP * sizeof(uint32_t) + KM_QPB * 5 * sizeof(unsigned long long) +
                   KM_QPB * sizeof(int);
    static thread_local bool attr_set = false;   // > 64 KB of dynamic LDS needs an explicit opt-in (per device context)
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)knn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
            return dflow_set_error(DFLOW_EHIP, "hipFuncSetAttribute(knn_mfma_kernel, %zu bytes of LDS) failed", shmem);
        attr_set = true;
    }
    hipLaunchKernelGGL(knn_mfma_kernel, dim3(nblocks), dim3(KM_THREADS), shmem, s, a, q);
    int rc = dflow_check_launch("knn_mfma_kernel");
    if (rc) return rc;
    rc = launch_knn_fix(p, d1, d2, proposals, lcosts, ctr, ovf, ovf_cap, ctr + 1, s);
    if (rc) return rc;
    hipLaunchKernelGGL(knn_finalize_kernel, dim3(nb), dim3(256), 0, s, g, a.LP, proposals, lcosts, nprop, bestlabels);
    return dflow_check_launch("knn_finalize_kernel");
}
