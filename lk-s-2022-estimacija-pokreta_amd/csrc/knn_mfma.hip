// K3 knn_cells on the matrix cores (default path of dflow_knn_proposals).
//
// Same result as the exact VALU search (knn.hip) -- the canonical 5-NN of generisi (daisy i flann.py:157-189),
// bit for bit -- but the 1.7e10 descriptor pairs of a Sintel pass are screened by f16 MFMA instead of being
// evaluated one by one:
//
//   basis     the principal axes V of the second-moment matrix (about the ORIGIN) of a sample of image-2 descriptors
//             (knn_pca.hip).  x = 64 d, y = V^T x: distances do not change (V orthonormal to 2e-6, measured), and the energy
//             of y collects in its leading components: the screen multiplies the first KM_KD = 40 of them (y_P) and bounds
//             the product of the dropped 28 (y_D) by Cauchy-Schwarz, |y_D(q) . y_D(c)| <= n_q n_c with n >= |y_D|.
//             No centring (rounds 2-3 subtracted the mean descriptor): DAISY without normalisation (daisy i flann.py:66) is a
//             gradient histogram, so the descriptors of low-texture regions (road, sky, blur) cluster at the ORIGIN, and f16
//             keeps its 11 bits on them only if the origin stays where it is (tools/screen_model.py: a road pixel against a
//             road cell emits 95 events with the mean as centre, 6 with the origin).
//   prep      (knn_prep_kernel) both images become rows of KM_K = 48 f16: y~ = f16(y_P), then the row's factors of the
//             bound.  With G^ = the MFMA's value of y~_q . y~_c (exact products, f32 accumulation) and E = y~ - y_P the
//             rounding errors actually made (the f16 part is measured, nothing is assumed about f16 rounding, subnormals
//             included; the float32 rotation enters with its worst-case bound KM_RHO |x|),
//                 y_P(q) . y_P(c) = y~_q . y~_c - y~_q . E_c - E_q . y_P(c)
//                 |y_P(q) . y_P(c) - G^| <= |y~_q| (|E_c| + eta |y~_c|) + |E_q| |y_P(c)| + eta (other products)
//             and every term is a product of a per-query and a per-candidate number, i.e. ONE K slot of the same MFMA
//             (rounds 2-3 split the cross terms by AM-GM into S_q + S_c with a fixed t = 2^-12.5, which is loose by
//             |y_q| / |y_c| when a textured query meets a flat cell).  eta = 2^-17 bounds the f32 accumulation of the 48
//             exact products inside the matrix core: it holds for ANY order of the additions and any rounding mode with at
//             most one ulp (2^-23) per addition, up to 64 terms; measured on gfx950, tools/ubench/mfma_err.hip: <= 2^-20.7
//             at K = 80 over six operand distributions.  Candidate rows: [y~(40), n_c, 64 (|E_c| + eta |y~_c|), |y_P(c)| / 64,
//             two f16 pieces of h = 0.5 |x_c|^2, 4096 S_c, 1, 1] with S_c = eta h + 3e-6 h + the measured residual of the
//             two pieces (3e-6 h covers |V^T V - I| and the roundings); query rows: [y~(40), -+n_q, -+|y~_q| / 64,
//             -+64 |E_q|, -1, -1, -+2^-12, -th_1, -th_2]; all bound factors rounded UP into f16.  So tau = y_q . y_c -
//             0.5 |y_c|^2 = 0.5 (|y_q|^2 - |y_q - y_c|^2) lies between the two values one MFMA chain yields:
//   screen    (knn_screen_kernel) pass 1: w = G^ - bounds - h - S_c <= tau for every (query, candidate) of a (64-query wave,
//             candidate cell).  Every lane keeps the 5 largest maxima of its 16-value tile columns -> a5 <= the 5th largest
//             tau of the query.  pass 2: v = G^ + bounds - h + S_c >= tau.  Only candidates with v >= a5 - s (s: rounding
//             of the canonical float32 distance) can be among the exact 5 NN: these "events" (a 16-bit row mask per lane
//             and tile) go to per-lane lists in the workspace (capacity = the tiles of a cell: a list cannot overflow).
//             Events per (query, cell) on the bench frame: 5.7 (5 is the minimum).
//   hard rows candidate rows that are all zero (saturated / constant regions: exactly-zero DAISY) tie with each other for
//             every query, and the canonical order breaks ties by index: only the 5 of lowest index in a cell can enter
//             a top 5, the others become sentinel rows (knn_cell_post_kernel).  An all-zero QUERY has the same answer in a
//             cell wherever it is (distance = |c|^2): the 5 candidates of smallest (|c|^2, index), found once per cell
//             (knn_cell_post_kernel) and copied by the resolve kernel; such lanes emit no events.
//   resolve   (knn_resolve_kernel) one wave per (64 queries, window column): canonical float32 distance (sequential
//             fmaf chain) and truncated L1 cost (numpy order) of every event, exact (distance, index) top-5 in
//             registers, proposals [dy,dx] and costs into the cell's 5 slots (Q1-Q3).  The rows are fetched by the
//             whole wave through LDS (see the kernel).  A query with more than KM_HEAVY_ENTRIES list entries in a cell (fringes
//             of flat regions: hundreds of near-ties) would hold its 63 neighbours for hundreds of rounds: it is handed to
//             knn_resolve_heavy_kernel, one wave per (query, cell) with the lanes over the EVENTS and a merge of the 64 partial
//             top-5 lists by the same keys.
//   fix       lists that cannot be screened (a query or candidate outside the range the f16 rows cover, NaN: flagged per
//             LIST by the prep kernel) are redone by the exact brute-force search (knn.hip), list by list -- never the whole
//             pass, unless the basis itself fails its orthonormality check.  finalize sets nprop, the WTA label (first
//             minimum, Q4) and the fills.
//
// MFMA layout (v_mfma_f32_32x32x16_f16): A = candidates (rows), B = queries (columns): lane l holds
// A[row l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col l&31]; D: col = l&31, row = (r&3) + 8(r>>2) + 4(l>>5).
// A workgroup is 8 waves x 64 queries (2 column groups); candidates stream through LDS in chunks of 192 rows (112-byte
// pitch: conflict-free ds_read_b128) through a ring of 3 buffers filled by global_load_lds DMA.
#include "dflow_common.h"
#include "knn_pca.h"
#include "row_stage.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V> struct KmC { static constexpr int value = V; };

#define KM_ALPHA 64.0f
#define KM_K 48                 // f16 per prepared row (96 bytes)
#define KM_KSTEPS (KM_K / 16)   // MFMAs per 32x32 tile
#define KM_KD 40                // leading principal components in the matrix product
#define KM_SLOT_N 40            // n = bound of the norm of the dropped components (queries: -n in pass 1, +n in pass 2)
#define KM_SLOT_QE 41           // candidates: 64 (|E_c| + eta |y~_c|); queries: -+|y~_q| / 64
#define KM_SLOT_EN 42           // candidates: |y_P(c)| / 64; queries: -+64 |E_q|
#define KM_SLOT_H 43            // 43, 44: two f16 pieces of h (candidates) / -1, -1 (queries)
#define KM_SLOT_S 45            // 4096 S_c (candidates) / -2^-12 in pass 1, +2^-12 in pass 2 (queries)
#define KM_SLOT_ONE 46          // 46, 47: 1.0 (candidates) / minus the two pieces of the threshold in pass 2 (queries)
#define KM_XSCALE 64.0          // scaling of the two cross-term slot pairs (keeps both factors in f16's normal range)
#define KM_SSCALE 4096.0        // scaling of the S_c slot
#define KM_NORM2_MAX 30000.0    // rows with |x|^2 beyond this are not screened (their lists go to the exact search): with both
                                // norms below it every real MFMA value lies above -50000 and the sentinel rows (-60000) below
#define KM_WAVES 8
#define KM_THREADS (64 * KM_WAVES)
#define KM_QPW 64               // queries per wave: 2 column groups of 32
#define KM_QPB (KM_WAVES * KM_QPW)
#ifndef KM_CHUNK
#define KM_CHUNK 192            // candidates per LDS chunk (6 tiles of 32)
#endif
#define KM_PITCH 112            // LDS row pitch in bytes (28 dwords: 16 consecutive rows hit 16 distinct 4-bank groups)
#define KM_SLOTS (KM_PITCH / 16)                  // 16-byte slots per staged row (6 data + 1 pad)
#define KM_STAGE_INS ((KM_CHUNK * KM_SLOTS + 63) / 64)      // wave-instructions per staged chunk (21)
#define KM_ABUF (KM_STAGE_INS * 1024)            // bytes per staged chunk buffer
#ifndef KM_NBUF
#define KM_NBUF 3                // ring of staged chunks: the DMA runs KM_NBUF - 1 chunks ahead of the MFMAs
#endif
#define KM_MAXNW ((KM_STAGE_INS + KM_WAVES - 1) / KM_WAVES)     // DMA instructions of a wave per chunk: KM_MAXNW or one fewer
#define KM_SPC (2 * (KM_CHUNK / 32))                           // event stores of a wave per chunk in pass 2 (tiles x groups)
#define KM_EVROWS_MAX 96        // event entries (tile, 16-bit row mask) per lane, group and candidate cell: one per tile of the largest
                                // cell + 1 (km_evrows), so that no list can run out (mean 2.7, 99 % <= 7 on dense texture, but every
                                // tile of a cell in flat regions); cells of more than 95 tiles (3040 points): 96, and a list
                                // that does run out is redone by knn_fix_kernel
#define KM_MAXPTS 65535         // candidate index must fit 16 bits
#define KM_LIST_WORDS(evrows) (2 * (evrows) * 64)       // one event list: [group][entry][lane] uint32
#define KM_ETA 7.62939453125e-6                  // eta = 2^-17: allowance for the f32 accumulation inside the matrix core

struct KmGeom {
    Geom g;
    int LP, qwaves;             // qwaves = 64-query groups per cell (of the largest cell)
    int evrows;                 // entries per lane of an event list
    float tphi;
};

// list id of (query cell, 64-query wave, window slot)
__device__ static inline size_t list_id(const KmGeom &a, int qcell, int qwave, int wslot)
{
    const int win = 2 * a.g.win + 1;
    return ((size_t)qcell * a.qwaves + qwave) * (win * win) + wslot;
}

// ------------------------------------------------------------------------------------------------ prep
// Candidate (image 2) rows are stored cell by cell in TILE POSITION order, every cell padded to whole chunks: position
// (tile, row) of cell c holds candidate row * ntiles + tile of that cell (see the screen kernel), positions without a
// candidate hold the sentinel row.  The screen then stages a chunk as one contiguous 18 KB read.
__host__ __device__ static inline int km_pad(int npts) { return (npts + KM_CHUNK - 1) / KM_CHUNK * KM_CHUNK; }
__host__ __device__ static inline size_t km_cell_base(const Geom &g, int ci, int cj)
{
    const int wl = g.x1(g.ncx - 1) - g.x0(g.ncx - 1), hl = g.y1(g.ncy - 1) - g.y0(g.ncy - 1);
    const size_t rowsum = (size_t)(g.ncx - 1) * km_pad(g.cw * g.ch) + km_pad(wl * g.ch);      // a regular row of cells
    const int hj = cj == g.ncy - 1 ? hl : g.ch;
    return (size_t)cj * rowsum + (size_t)ci * km_pad(g.cw * hj);
}
__host__ __device__ static inline size_t km_total_rows(const Geom &g)
{
    const int wl = g.x1(g.ncx - 1) - g.x0(g.ncx - 1), hl = g.y1(g.ncy - 1) - g.y0(g.ncy - 1);
    return km_cell_base(g, 0, g.ncy - 1) + (size_t)(g.ncx - 1) * km_pad(g.cw * hl) + km_pad(wl * hl);
}

// the row of a position without a candidate: h = 60000, everything else 0 -> MFMA value -60000 in both passes, below every
// real one (rows the screen accepts lie above -50000, KM_NORM2_MAX) and below every threshold (>= -55000; no 1.0 here to subtract it)
__device__ static inline void km_store_sentinel(_Float16 *row)
{
    half8 *o = reinterpret_cast<half8 *>(row);
    const half8 z = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
    half8 t8 = z;
    t8[KM_SLOT_H - 40] = (_Float16)60000.0f;
#pragma unroll
    for (int k = 0; k < KM_K / 8; k++) o[k] = k == KM_K / 8 - 1 ? t8 : z;
}

// f16 value not below v (v >= 0, below the f16 range): the factor exceeds one half-ulp of the rounding to nearest, the
// constant the spacing of the subnormals
__device__ static inline _Float16 km_f16_up(double v) { return (_Float16)(float)(v * 1.001 + 1e-7); }

// per-pixel record of image 1 next to its f16 row
#define KM_Q_ZERO 1.0f          // the descriptor is all zero: the answer of every cell comes from the cell's own list (KmCellTop)
#define KM_Q_BAD 2.0f           // outside the range the screen covers (or NaN): every list of its 64-query wave goes to the exact search
// per-position record of image 2: canonical squared norm (= the canonical distance of an all-zero query) and |c|_1 in numpy's
// summation order (= its L1 cost), so that the all-zero queries' answer can be found once per cell
struct KmCellTop { uint32_t idx[5]; float cost[5]; };

// which = 0: image 1 (queries), one thread per pixel, rows in pixel order.  which = 1: image 2 (candidates), one thread
// per (cell = blockIdx.y, tile position), rows in position order (above).  The rotation y_j = sum_i V[j][i] x_i runs as two
// float32 fmaf chains (even / odd dimensions: one chain of packed fmas) that are added at the end; the components of V arrive
// through the scalar cache (vt is wave-uniform), 8 components per iteration = one 16-byte store.  Its rounding is part of E:
// for ANY order of the 68 additions |y^_j - y_j| <= gamma_68 sum_i |V_ji x_i| <= 68 * 2^-24 (1 + 5e-6) |x|, over the 40
// components |y^ - y_P| <= 2.57e-5 |x| (KM_RHO; x = 64 d is exact).
#define KM_RHO 2.65e-5
typedef float km_f2p __attribute__((ext_vector_type(2)));
template <typename T>
__global__ void __launch_bounds__(256) knn_prep_kernel(const T *__restrict__ d, const float *__restrict__ vt, _Float16 *__restrict__ h,
                                                       float2 *__restrict__ qs, float2 *__restrict__ z0, uint8_t *__restrict__ zflag,
                                                       int *__restrict__ cell_bad, Geom g, int which)
{
    int pix;
    size_t orow;
    if (which == 0) {
        pix = blockIdx.x * blockDim.x + threadIdx.x;
        if (pix >= g.H * g.W) return;
        orow = (size_t)pix;
    } else {
        const int ci = blockIdx.y % g.ncx, cj = blockIdx.y / g.ncx;
        const int cx0 = g.x0(ci), cy0 = g.y0(cj), ccw = g.x1(ci) - cx0, cnpts = ccw * (g.y1(cj) - cy0);
        const int pos = blockIdx.x * blockDim.x + threadIdx.x;
        if (pos >= km_pad(cnpts)) return;
        orow = km_cell_base(g, ci, cj) + pos;
        const int ntiles = km_pad(cnpts) / 32;
        const int idx = (pos & 31) * ntiles + (pos >> 5);
        if (idx >= cnpts) {
            km_store_sentinel(h + orow * KM_K);
            z0[orow] = make_float2(INFINITY, 0.0f);
            zflag[orow] = 0;
            return;
        }
        pix = (cy0 + idx / ccw) * g.W + cx0 + idx % ccw;
    }
    float x[DFLOW_DESC];
    desc_load_row(x, d, (size_t)pix);
    if (which == 1) {
        // what an all-zero query would compute for this candidate: e = 0 - c_k, acc = fmaf(e, e, acc) over k = 0..67, and
        // sum |e| in numpy's pairwise order (knn_resolve_kernel, l1_cost_np)
        float acc = 0.0f, rs[8];
#pragma unroll
        for (int k = 0; k < DFLOW_DESC; k++) acc = __fmaf_rn(x[k], x[k], acc);
#pragma unroll
        for (int j = 0; j < 8; j++) rs[j] = fabsf(x[j]);
#pragma unroll
        for (int i = 8; i < 64; i += 8)
#pragma unroll
            for (int j = 0; j < 8; j++) rs[j] = rs[j] + fabsf(x[i + j]);
        float l1 = ((rs[0] + rs[1]) + (rs[2] + rs[3])) + ((rs[4] + rs[5]) + (rs[6] + rs[7]));
#pragma unroll
        for (int i = 64; i < DFLOW_DESC; i++) l1 = l1 + fabsf(x[i]);
        z0[orow] = make_float2(acc, l1);
    }
    double sxall = 0.0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < DFLOW_DESC; k++) {
        const float sc = KM_ALPHA * x[k];             // exact
        bad |= (__float_as_uint(sc) & 0x7FFFFFFFu) >= 0x7F800000u;      // inf, NaN (tested on the bits: -fno-honor-nans)
        x[k] = sc;
        sxall = fma((double)sc, (double)sc, sxall);          // |x|^2; |V^T x|^2 = |x|^2 (1 +- PCA_DELTA_MAX)
    }
    bad |= !(sxall < KM_NORM2_MAX);
    const bool zero = sxall == 0.0;                   // every value +-0: ties with every other such row for every query
    double ss = 0.0, sx = 0.0, se = 0.0;        // |y~|^2, |y^|^2, |y~ - y^|^2 over the KM_KD leading components
    auto component = [&](int j) -> _Float16 {
        // two fmaf chains (even / odd dimensions) as ONE chain of v_pk_fma_f32: half the instructions of a single chain
        const km_f2p *__restrict__ vj = reinterpret_cast<const km_f2p *>(vt + (size_t)j * DFLOW_DESC);
        km_f2p acc = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < DFLOW_DESC / 2; i++) acc = __builtin_elementwise_fma(vj[i], (km_f2p){x[2 * i], x[2 * i + 1]}, acc);
        const float y = acc.x + acc.y;
        const _Float16 hv = (_Float16)y;
        const double f = (double)(float)hv, yd = (double)y, er = f - yd;
        ss = fma(f, f, ss); sx = fma(yd, yd, sx); se = fma(er, er, se);
        return hv;
    };
    half8 *o = reinterpret_cast<half8 *>(h + orow * KM_K);
    static_assert(KM_KD % 8 == 0 && KM_KD + 8 == KM_K, "the last 16-byte piece of a row holds the eight bound slots");
#pragma unroll 1
    for (int jo = 0; jo < KM_KD / 8; jo++) {
        half8 r;
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = component(8 * jo + j);
        o[jo] = r;
    }
    half8 last;
    // |E| <= |y~ - y^| + KM_RHO |x|;  |y_P| <= |y^| + KM_RHO |x|;
    // |y_D|^2 = |y|^2 - |y_P|^2 <= (1 + delta)|x|^2 - (|y^| - KM_RHO |x|)^2
    const double rx = sqrt(sxall), ry = sqrt(sx);
    const double en = sqrt(se) + KM_RHO * rx;
    const double ypl = fmax(0.0, ry - KM_RHO * rx), ypu = ry + KM_RHO * rx;
    const double nd = sqrt(fmax(0.0, sxall * (1.0 + 1.1 * PCA_DELTA_MAX) - ypl * ypl));
    // km_f16_up exceeds its argument by >= 5e-4 relative: that also pays the accumulation allowance eta on the slot's own product
    if (which == 0) {
        // stored in pass-1 form (all bound terms subtracted); the screen flips the four signs for pass 2
        last[KM_SLOT_N - 40] = -km_f16_up(nd);
        last[KM_SLOT_QE - 40] = -km_f16_up(sqrt(ss) * (1.0 / KM_XSCALE));
        last[KM_SLOT_EN - 40] = -km_f16_up(en * KM_XSCALE);
        last[KM_SLOT_H - 40] = (_Float16)-1.0f; last[KM_SLOT_H + 1 - 40] = (_Float16)-1.0f;
        last[KM_SLOT_S - 40] = (_Float16)(float)(-1.0 / KM_SSCALE);
        last[KM_SLOT_ONE - 40] = (_Float16)0.0f; last[KM_SLOT_ONE + 1 - 40] = (_Float16)0.0f;
        // (flag, 0.5 |y_q|^2 rounded up into float32)
        qs[pix] = make_float2(bad ? KM_Q_BAD : zero ? KM_Q_ZERO : 0.0f, (float)(0.5 * sxall * (1.0 + 1.5 * PCA_DELTA_MAX) * 1.0000002));
    } else {
        const double hh = 0.5 * sxall;
        const _Float16 p1 = (_Float16)(float)hh;
        const _Float16 p2 = (_Float16)(float)(hh - (double)(float)p1);
        const double hres = fabs(hh - (double)(float)p1 - (double)(float)p2);     // what the two pieces miss (exact in double)
        // S_c: eta on the h products (|p1| + |p2| <= 1.001 h) and on S_c's own; 3e-6 h: |0.5 |y_c|^2 - h| <= delta h;
        // the measured residual of the two-piece form
        const double scs = (KM_ETA * 1.001 * hh + 3e-6 * hh + hres) * 1.0004;
        last[KM_SLOT_N - 40] = km_f16_up(nd);
        last[KM_SLOT_QE - 40] = km_f16_up((en + KM_ETA * sqrt(ss)) * KM_XSCALE);
        last[KM_SLOT_EN - 40] = km_f16_up(ypu * (1.0 / KM_XSCALE));
        last[KM_SLOT_H - 40] = p1; last[KM_SLOT_H + 1 - 40] = p2;
        last[KM_SLOT_S - 40] = km_f16_up(scs * KM_SSCALE);
        last[KM_SLOT_ONE - 40] = (_Float16)1.0f; last[KM_SLOT_ONE + 1 - 40] = (_Float16)1.0f;
        zflag[orow] = zero ? 1 : 0;
        if (bad) cell_bad[blockIdx.y] = 1;
    }
    o[KM_K / 8 - 1] = last;
}

// One block per candidate cell, after knn_prep_kernel: (a) of the all-zero rows of the cell only the 5 of lowest index stay
// candidates (equal rows have equal distances to every query and the canonical order prefers the lower index), the others
// become sentinel rows; (b) the cell's answer for an all-zero query: the 5 candidates of smallest (|c|^2, index) with their
// L1 costs, in that order (what knn_resolve_kernel would find).
__global__ void __launch_bounds__(256) knn_cell_post_kernel(_Float16 *__restrict__ h2, const float2 *__restrict__ z0,
                                                            const uint8_t *__restrict__ zflag, KmCellTop *__restrict__ ztop, Geom g)
{
    __shared__ int wsum[4];
    __shared__ unsigned long long wmin[4];
    const int ci = blockIdx.x % g.ncx, cj = blockIdx.x / g.ncx;
    const int cnpts = (g.x1(ci) - g.x0(ci)) * (g.y1(cj) - g.y0(cj));
    const int npad = km_pad(cnpts), ntiles = npad / 32;
    const size_t base = km_cell_base(g, ci, cj);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int running = 0;
    for (int c0 = 0; c0 < cnpts; c0 += 256) {                            // block-uniform trip count
        const int idx = c0 + tid;
        const int pos = (idx % ntiles) * 32 + idx / ntiles;
        const bool z = idx < cnpts && zflag[base + pos] != 0;
        const unsigned long long bal = __ballot(z);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = running + __popcll(bal & ((1ull << lane) - 1ull));
        int total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { before += w < wave ? wsum[w] : 0; total += wsum[w]; }
        if (z && before >= 5) km_store_sentinel(h2 + (base + pos) * KM_K);
        running += total;
        __syncthreads();
    }
    // (b): every thread keeps the 5 smallest keys of its positions, then 5 rounds of a block-wide minimum above the last winner
    unsigned long long k5[5];
#pragma unroll
    for (int i = 0; i < 5; i++) k5[i] = ~0ull;
    for (int pos = tid; pos < npad; pos += 256) {
        const int idx = (pos & 31) * ntiles + (pos >> 5);
        const float dd = z0[base + pos].x;
        if (idx < cnpts && dd < INFINITY) {
            unsigned long long x = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)idx;
#pragma unroll
            for (int i = 0; i < 5; i++) { const bool lt = x < k5[i]; const unsigned long long lo = lt ? x : k5[i]; x = lt ? k5[i] : x; k5[i] = lo; }
        }
    }
    unsigned long long last = 0ull;
    bool first = true;
    for (int r = 0; r < 5; r++) {
        unsigned long long m = ~0ull;
#pragma unroll
        for (int i = 4; i >= 0; i--) if (first || k5[i] > last) m = k5[i];              // smallest of mine above the last winner
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(m, off); m = o < m ? o : m; }
        if (lane == 0) wmin[wave] = m;
        __syncthreads();
        m = wmin[0];
#pragma unroll
        for (int w = 1; w < 4; w++) m = wmin[w] < m ? wmin[w] : m;
        if (tid == 0) {
            const uint32_t idx = m == ~0ull ? 0u : (uint32_t)(m & 0xFFFFFFFFu);       // fewer than 5 finite rows: the cell is flagged bad
            ztop[blockIdx.x].idx[r] = idx;
            ztop[blockIdx.x].cost[r] = z0[base + (idx % ntiles) * 32 + idx / ntiles].y;
        }
        last = m; first = false;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ screen
struct KmScreen {
    const _Float16 *h1, *h2;       // prepared f16 rows
    const float2 *qs;              // (KM_Q_* flag, 0.5 |x_q|^2) per image-1 pixel
    const int *cell_bad;           // per candidate cell: holds a row the screen does not cover
    uint32_t *ev;                  // [list][2][evrows][64]: (tile << 16) | row mask
    uint8_t *ev_cnt;               // [list][2][64]; 255 = overflow
};

__device__ static inline void top5_insert_desc(float (&a)[5], float m)
{
#pragma unroll
    // a is sorted (descending): the new a[i] is the median of (old a[i-1], old a[i], m); one v_med3_f32 per slot instead
    // of a max / min pair
    for (int i = 4; i >= 1; i--) a[i] = __builtin_amdgcn_fmed3f(a[i - 1], a[i], m);
    a[0] = fmaxf(a[0], m);
}

__device__ static inline float max16(const f32x16 &v)
{
    float m0 = fmaxf(fmaxf(v[0], v[1]), v[2]), m1 = fmaxf(fmaxf(v[3], v[4]), v[5]);
    float m2 = fmaxf(fmaxf(v[6], v[7]), v[8]), m3 = fmaxf(fmaxf(v[9], v[10]), v[11]);
    float m4 = fmaxf(fmaxf(v[12], v[13]), v[14]);
    return fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), fmaxf(m4, v[15]));
}

__global__ void __launch_bounds__(KM_THREADS, 4) knn_screen_kernel(KmGeom a, KmScreen p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *abuf = smem;                                                   // [KM_NBUF][KM_ABUF]

    const Geom g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;
    // ---- which candidate cell, and which 64-query wave of which query cell in its window.  A workgroup streams ONE
    // candidate cell; its 8 waves take 8 consecutive entries of the flat list of (window slot, 64-query wave) pairs of
    // that cell, so waves of one workgroup may belong to different query cells and no wave idles except in the last
    // workgroup of a cell (a cell has 27 waves of queries at 64x27: chunks of 8 per query cell would leave 5 of 32 idle).
    // Candidate-cell-major order: the blocks in flight at any time stream the same few candidate cells, so their rows
    // stay in the XCDs' L2.
    const int win = 2 * g.win + 1;
    const int wgs_per_cell = (win * win * a.qwaves + KM_WAVES - 1) / KM_WAVES;
    const int ccell = blockIdx.x / wgs_per_cell, w8 = blockIdx.x % wgs_per_cell;
    const int ci = ccell % g.ncx, cj = ccell / g.ncx;
    int flat = w8 * KM_WAVES + __builtin_amdgcn_readfirstlane(wave);
    int total = 0, qci = -1, qcj = -1, qwave = 0, fqci = -1, fqcj = -1;
    for (int qslot = 0; qslot < win * win; qslot++) {          // wave-uniform scan of the window (<= 25 slots)
        const int sci = ci - g.win + qslot / win, scj = cj - g.win + qslot % win;
        if (sci < 0 || sci >= g.ncx || scj < 0 || scj >= g.ncy) continue;
        const int nw = ((g.x1(sci) - g.x0(sci)) * (g.y1(scj) - g.y0(scj)) + KM_QPW - 1) / KM_QPW;
        if (fqci < 0) { fqci = sci; fqcj = scj; }
        if (qci < 0 && flat >= total && flat < total + nw) { qci = sci; qcj = scj; qwave = flat - total; }
        total += nw;
    }
    if (w8 * KM_WAVES >= total) return;                               // block-uniform: nothing left for this workgroup
    const bool wave_active = qci >= 0;                                // idle waves only help staging (barriers stay block-uniform)
    if (!wave_active) { qci = fqci; qcj = fqcj; qwave = 0; }         // they shadow a valid query wave and store nothing
    const int qcell = qcj * g.ncx + qci;
    const int qx0 = g.x0(qci), qy0 = g.y0(qcj), qcw = g.x1(qci) - qx0, qnpts = qcw * (g.y1(qcj) - qy0);
    const int cimin = max(0, qci - g.win), cjmin = max(0, qcj - g.win), cjmax = min(g.ncy - 1, qcj + g.win);
    const int wslot = (ci - cimin) * (cjmax - cjmin + 1) + (cj - cjmin);   // reference order: ci outer, cj inner (Q2)
    const int cx0 = g.x0(ci), cy0 = g.y0(cj), ccw = g.x1(ci) - cx0, cnpts = ccw * (g.y1(cj) - cy0);
    (void)cx0; (void)cy0;

    // ---- my queries: group gq (0/1), column col; B fragments (last k-step differs between the passes) and slack
    half8 bfrag[2][KM_KSTEPS];
    float hq[2];
    bool qzero[2], qbad = false;
#pragma unroll
    for (int gq = 0; gq < 2; gq++) {
        int qi = qwave * KM_QPW + gq * 32 + col;
        if (qi >= qnpts) qi = qnpts - 1;                                 // inactive columns shadow the last query
        const int qpix = (qy0 + qi / qcw) * g.W + qx0 + qi % qcw;
        const half8 *src = reinterpret_cast<const half8 *>(p.h1 + (size_t)qpix * KM_K);
#pragma unroll
        for (int s = 0; s < KM_KSTEPS; s++) bfrag[gq][s] = src[2 * s + half];
        // k = 40..47 sit in half 1 of the last k-step: -n_q, -|y~_q| / 64, -64 |E_q|, -1, -1 (h), -2^-12 (S_c), 0, 0 as
        // stored: pass 1 computes G^ - bounds - h - S_c
        const float2 qq = p.qs[qpix];
        hq[gq] = qq.y;
        qzero[gq] = qq.x == KM_Q_ZERO;
        qbad |= qq.x == KM_Q_BAD;
    }
    // a list with a row the screen does not cover (range, NaN) is handed to the exact search as a whole
    const bool list_bad = __ballot(qbad) != 0ull || p.cell_bad[ccell] != 0;

    const int nchunks = (cnpts + KM_CHUNK - 1) / KM_CHUNK;
    const size_t cbase = km_cell_base(g, ci, cj);
    // Asynchronous staging of one chunk straight into LDS (global_load_lds_dwordx4: the LDS address is wave-uniform
    // base + 16*lane, the global address is per lane).  The LDS image is 192 rows of 7 16-byte slots (6 data + 1
    // pad = 112-byte pitch) = 1344 slots = 21 wave-instructions (the buffer is 21 KB; 3 of them per workgroup).
    // Pad slots re-read part 0; positions beyond the cell hold sentinel rows (MFMA value -60000 < every real one).
    // the byte offsets of this wave's (at most two) DMA instructions inside a chunk do not depend on the chunk: computed
    // once; per chunk only the scalar base moves (scalar base + 32-bit lane offset addressing)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    uint32_t soff[KM_MAXNW];
#pragma unroll
    for (int i = 0; i < KM_MAXNW; i++) {
        const int slot = (wave_u + i * KM_WAVES) * 64 + lane;
        // the candidate rows are stored in tile position order (knn_prep_kernel): position p = (tile, row) holds
        // candidate row * ntiles + tile, so that the 16 rows a lane sees of one tile are far apart in the cell.
        // Neighbouring pixels have similar descriptors: with raster order several of a query's 5 best would share a
        // lane's tile column, of which only the maximum enters a5.
        int part = slot % KM_SLOTS;
        if (part >= KM_K / 8) part = 0;
        soff[i] = (uint32_t)((slot / KM_SLOTS) * (KM_K * 2) + part * 16);
    }
    const rs_gptr h2cell = (rs_gptr)(p.h2 + cbase * KM_K);
    const int n_w = (KM_STAGE_INS - wave_u + KM_WAVES - 1) / KM_WAVES;           // DMA instructions of this wave per chunk (1 or 2)
    auto stage = [&](int chunk, int buf) {
        char *base = abuf + (size_t)buf * KM_ABUF + wave_u * 1024;
        const rs_gptr src = h2cell + (size_t)min(chunk, nchunks - 1) * (KM_CHUNK * KM_K * 2);   // chunks staged past the end (never used) re-read the last one
#pragma unroll
        for (int i = 0; i < KM_MAXNW; i++)
            if (i < KM_MAXNW - 1 || n_w == KM_MAXNW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + soff[i]),
                                                 (__attribute__((address_space(3))) void *)(base + i * KM_WAVES * 1024), 16, 0, 0);
    };

    float a5[2][5];
    int cnt[2] = {0, 0};
    const int evrows = a.evrows;
    const uint32_t evoff[2] = {(uint32_t)lane * 4u, (uint32_t)lane * 4u + (uint32_t)evrows * 256u};
    const size_t lid = list_id(a, qcell, qwave, wslot);
    // wave-uniform buffer resource over this list; entry (gq, row) of a lane at 4 lane + 256 (32 gq + row)
    const __amdgpu_buffer_rsrc_t evrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.ev + lid * KM_LIST_WORDS(a.evrows)), 0, KM_LIST_WORDS(a.evrows) * 4, 0x00020000);

#pragma unroll
    for (int gq = 0; gq < 2; gq++)
#pragma unroll
        for (int i = 0; i < 5; i++) a5[gq][i] = -INFINITY;

    // A wave issues n_w (KM_MAXNW or one fewer) DMA instructions per chunk, and in pass 2 exactly KM_SPC event stores (one per
    // tile and column group, unconditionally).  Before the barrier that publishes chunk c+1 it waits until only the DMAs of
    // chunks c+2 .. c+KM_NBUF-1 and the stores of chunks c-1 and c may still be in flight (vmcnt counts all of them in
    // issue order).  Chunks and tiles past the end of the cell are still staged/processed
    // (sentinel rows) so that these counts are exact.
    // (counted waits need immediates: one instantiation per count)
    // (counted waits need immediates: one instantiation per count)
    auto wait_ring = [&](int pass) {
        if (pass == 0) {
            if (n_w == KM_MAXNW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KM_NBUF - 2) * KM_MAXNW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KM_NBUF - 2) * (KM_MAXNW - 1)) : "memory");
        } else {
            if (n_w == KM_MAXNW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KM_NBUF - 2) * KM_MAXNW + 2 * KM_SPC) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KM_NBUF - 2) * (KM_MAXNW - 1) + 2 * KM_SPC) : "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    // The pass-2 count above assumes that the stores of chunks c-1 and c really were issued: a wave that has not issued them
    // (every wave in front of the first chunk; idle waves, which stage but skip the tiles, in every chunk) would pass the wait
    // before the DMA of chunk c+1 has landed.  Such a wave issues the same number of stores with an offset beyond the
    // buffer's range instead (dropped by the hardware, counted by vmcnt like the real ones).
    auto dummy_stores = [&]() {
#pragma unroll
        for (int i = 0; i < KM_SPC; i++) __builtin_amdgcn_raw_buffer_store_b32(0u, evrsrc, (int)0xFFFFFF00u, 0, 0);
    };
    // Epilogues.  Pass 1: the maximum of the 16 values of this lane's tile column enters the lane's top-5.
    // Pass 2: 16-bit mask of the rows that qualify (bit r <-> accumulator register r) -> one event word.
    auto epi1 = [&](const f32x16 &acc, int gq) { top5_insert_desc(a5[gq], max16(acc)); };
    auto epi2 = [&](const f32x16 &acc, int gq, int tileidx) {
        // pass 2 accumulates v - th' (the threshold rides in the k = 46, 47 products), so a row qualifies iff its
        // accumulator is not negative: one v_alignbit per row shifts the sign bit into the word (bit r <-> register r)
        uint32_t neg = 0u;
#pragma unroll
        for (int r = 15; r >= 0; r--) neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(acc[r]), 31);
        // entry = (tile << 16) | (~neg & 0xFFFF) = neg ^ ((tile << 16) | 0xFFFF).  Always one store instruction (the vmcnt
        // bookkeeping of the ring counts on it): lanes with a mask write entry cnt of their list, the others (95 %) an offset
        // beyond the buffer's range, which the hardware drops without any memory traffic (raw buffer store, range checked).
        // Entry evrows-1 is never valid: a list that reaches it is reported as overflowed.
        const bool has = neg != 0xFFFFu;
        const uint32_t off = has ? evoff[gq] + ((uint32_t)min(cnt[gq], evrows - 1) << 8) : 0xFFFFFF00u;
        __builtin_amdgcn_raw_buffer_store_b32(neg ^ (((uint32_t)tileidx << 16) | 0xFFFFu), evrsrc, (int)off, 0, 0);
        cnt[gq] += has ? 1 : 0;
    };
    // the pipeline starts with a harmless unit: -inf never enters a top-5 (pass 1) and is negative (pass 2)
    const f32x16 minus_inf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY,
                              -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};
    // the tiles of one staged chunk; PASS is a compile-time constant (the scheduling directives need constants)
    f32x16 pend;
    int pend_tile = 0;
    // one tile: af = its A fragments (already in registers), afn <- those of the next tile of the chunk (if any), fetched
    // while the MFMAs of this tile run, so that only the first tile of a chunk waits for LDS
    auto one_tile = [&](auto passc, const int tileidx, const half8 (&af)[KM_KSTEPS], half8 (&afn)[KM_KSTEPS], const char *arow_next) __attribute__((always_inline)) {
        constexpr int PASS = decltype(passc)::value;
        if (arow_next) {
#pragma unroll
            for (int s = 0; s < KM_KSTEPS; s++) afn[s] = *reinterpret_cast<const half8 *>(arow_next + s * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        // unit (tile, group 0): its three MFMAs (one dependent chain, 32 cycles each) with the pending epilogue of
        // (previous tile, group 1) issued in their shadow, a few VALU per MFMA: the scheduler is told to build that
        // pipeline (sched_group_barrier) and not to mix the two halves (sched_barrier), otherwise it clusters the
        // MFMAs of both groups and the epilogues behind them, and the waves of a SIMD then alternate in lockstep
        // between matrix-only and vector-only phases
        f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KM_KSTEPS; s++) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bfrag[0][s], acc0, 0, 0, 0);
        if (PASS == 0) epi1(pend, 1); else epi2(pend, 1, pend_tile);
#pragma unroll
        for (int s = 0; s < KM_KSTEPS; s++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, PASS == 0 ? 5 : 8, 0);   // VALU of the epilogue in its shadow
        }
        __builtin_amdgcn_sched_barrier(0);
        // unit (tile, group 1): MFMAs, in their shadow the epilogue of (tile, group 0)
        f32x16 acc1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KM_KSTEPS; s++) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bfrag[1][s], acc1, 0, 0, 0);
        if (PASS == 0) epi1(acc0, 0); else epi2(acc0, 0, tileidx);
#pragma unroll
        for (int s = 0; s < KM_KSTEPS; s++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x002, PASS == 0 ? 5 : 8, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        pend = acc1; pend_tile = tileidx;
    };
    auto tiles = [&](auto passc, const int chunk, const char *ab) __attribute__((always_inline)) {
        static_assert((KM_CHUNK / 32) % 2 == 1 || KM_CHUNK / 32 >= 2, "tiles per chunk");
        half8 afa[KM_KSTEPS], afb[KM_KSTEPS];
        const char *arow = ab + col * KM_PITCH + half * 16;
#pragma unroll
        for (int s = 0; s < KM_KSTEPS; s++) afa[s] = *reinterpret_cast<const half8 *>(arow + s * 32);
        const int tile0 = chunk * (KM_CHUNK / 32);
#pragma unroll 1
        for (int tile = 0; tile + 1 < KM_CHUNK / 32; tile += 2) {
            one_tile(passc, tile0 + tile, afa, afb, arow + (tile + 1) * 32 * KM_PITCH);
            one_tile(passc, tile0 + tile + 1, afb, afa, tile + 2 < KM_CHUNK / 32 ? arow + (tile + 2) * 32 * KM_PITCH : nullptr);
        }
        if ((KM_CHUNK / 32) % 2 == 1) one_tile(passc, tile0 + KM_CHUNK / 32 - 1, afa, afb, nullptr);
    };
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
        for (int c = 0; c < KM_NBUF - 1; c++) stage(c, c);
        if (n_w == KM_MAXNW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KM_NBUF - 2) * KM_MAXNW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KM_NBUF - 2) * (KM_MAXNW - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        // Software pipeline: the epilogue of the previous (tile, group) unit is issued behind the MFMAs of the current
        // one, so the VALU work runs while the matrix pipe is busy.  The pipeline starts with a harmless unit (-inf).
        pend = minus_inf;
        pend_tile = 0;
        if (pass == 1) dummy_stores();                      // stand in for the stores of a chunk in front of the first
        for (int chunk = 0; chunk < nchunks; chunk++) {
            const int buf = chunk % KM_NBUF;
            stage(chunk + KM_NBUF - 1, (chunk + KM_NBUF - 1) % KM_NBUF);                // that buffer was released by the previous barrier
            const char *ab = abuf + (size_t)buf * KM_ABUF;
            if (wave_active) { if (pass == 0) tiles(KmC<0>(), chunk, ab); else tiles(KmC<1>(), chunk, ab); }
            else if (pass == 1) dummy_stores();
            wait_ring(pass);
        }
        if (wave_active) { if (pass == 0) epi1(pend, 1); else epi2(pend, 1, pend_tile); }   // drain the pipeline
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // drain the over-staged chunks before the buffers are reused
        __builtin_amdgcn_s_barrier();
        if (pass == 0) {
            // merge the two half-lanes of every query; pass-2 values v qualify iff v >= a5 - s, where s covers the rounding of
            // the canonical float32 distance (relative < 1.1e-5) and |y_q - y_c|^2 = |x_q - x_c|^2 (1 +- 2e-6): twice
            // 1.35e-5 of 64^2 d^2 / 2 = 0.5 |x_q|^2 - tau <= hq - a5
#pragma unroll
            for (int gq = 0; gq < 2; gq++) {
                float o[5];
#pragma unroll
                for (int i = 0; i < 5; i++) o[i] = __shfl_xor(a5[gq][i], 32);
#pragma unroll
                for (int i = 0; i < 5; i++) top5_insert_desc(a5[gq], o[i]);
                const float a5v = a5[gq][4];
                const float x = a5v - 2.7e-5f * fmaxf(hq[gq] - a5v, 0.0f);
                // (the float32 roundings of x itself: a few ulp of its two terms)
                float th = fmaxf(x - 1e-6f * (fabsf(a5v) + 2.7e-5f * hq[gq]), -55000.0f);   // real values are > -50000, sentinel rows -60000
                // an all-zero query takes the cell's own list (knn_cell_post_kernel): no events
                if (qzero[gq]) th = 60000.0f;
                // Pass 2 subtracts the threshold inside the matrix core: two f16 pieces of th' (22 bits) times the 1.0 the
                // candidate rows carry at k = 46, 47.  th' lies below th by the accumulation allowance eta = 7.7e-6 on the two extra
                // products (|pieces| <= 1.001 |th'|) and by more than the pieces' truncation (2^-22 = 2.4e-7 relative, < 2^-24
                // absolute in the subnormal range), so v >= th implies a computed v - th' >= 0; sentinel rows (-60000, no
                // 1.0) stay negative.  The bound slots and the S_c selector change sign: v = G^ + bounds - h + S_c.
                const float thp = th - 1e-5f * fabsf(th) - 6e-8f;
                if (half == 1) {
                    half8 b2 = bfrag[gq][KM_KSTEPS - 1];
                    b2[KM_SLOT_N - 40] = -b2[KM_SLOT_N - 40];
                    b2[KM_SLOT_QE - 40] = -b2[KM_SLOT_QE - 40];
                    b2[KM_SLOT_EN - 40] = -b2[KM_SLOT_EN - 40];
                    b2[KM_SLOT_S - 40] = -b2[KM_SLOT_S - 40];
                    const _Float16 p1 = (_Float16)thp;
                    const _Float16 p2 = (_Float16)(thp - (float)p1);
                    b2[KM_SLOT_ONE - 40] = -p1; b2[KM_SLOT_ONE + 1 - 40] = -p2;
                    bfrag[gq][KM_KSTEPS - 1] = b2;
                }
            }
        }
    }
    if (!wave_active) return;
#pragma unroll
    for (int gq = 0; gq < 2; gq++) p.ev_cnt[lid * 128 + gq * 64 + lane] = (uint8_t)(list_bad || cnt[gq] >= evrows ? 255 : cnt[gq]);
}

// ------------------------------------------------------------------------------------------------ resolve
// binary16 rows (144 bytes = 9 pieces of 16 bytes): every lane fetches its own row into registers; 9 cache accesses per row
// instead of the 17 of a float32 row, which is what made the float32 version stage its rows through LDS (row_stage.h)
typedef unsigned int km_u4 __attribute__((ext_vector_type(4)));
struct RowDirectH {
    km_u4 r[9];
    __device__ __forceinline__ void init(char *, uint32_t *, int) {}
    __device__ __forceinline__ void issue(rs_gptr base, uint32_t off16, bool act)
    {
        if (act) {
            const __attribute__((address_space(1))) km_u4 *s = reinterpret_cast<const __attribute__((address_space(1))) km_u4 *>(base + ((unsigned long long)off16 << 4));
#pragma unroll
            for (int j = 0; j < 9; j++) r[j] = s[j];
        }
    }
    __device__ __forceinline__ void fetch(float4 (&cv)[17])
    {
        float f[72];
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const dflow_h8 v = __builtin_bit_cast(dflow_h8, r[j]);
#pragma unroll
            for (int i = 0; i < 8; i++) f[8 * j + i] = (float)v[i];
        }
#pragma unroll
        for (int k = 0; k < 17; k++) cv[k] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
    }
};
template <bool F16> struct RowsOf { typedef RowStage type; static constexpr uint32_t PIECES = 17u; };
template <> struct RowsOf<true> { typedef RowDirectH type; static constexpr uint32_t PIECES = 9u; };

struct KmResolve {
    const void *d1, *d2;           // float32 (H,W,68) or binary16 (H,W,72)
    const uint32_t *ev;
    const uint8_t *ev_cnt;
    const float2 *qs;              // (KM_Q_* flag, .) per image-1 pixel
    const KmCellTop *ztop;         // per candidate cell: the answer for an all-zero query
    uint32_t *proposals;
    float *lcosts;
    int *ovf_count;                // lists for knn_fix_kernel: entries (qcell, first query, ci, cj); one slot per list of the pass
    int4 *ovf_list;
    int ovf_cap;
    int *heavy_count;              // (query, cell) pairs with more than KM_HEAVY_ENTRIES list entries: (qcell, query index, ci, cj) for
    int4 *heavy_list;              // knn_resolve_heavy_kernel, which spreads ONE query's events over the 64 lanes
    int heavy_cap;
};

// sorted insertion of (key, cost) into an ascending top-5; key order = canonical (distance, index) order
__device__ static inline void key_insert(unsigned long long (&k)[5], float (&c)[5], unsigned long long x, float xc)
{
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const bool lt = x < k[i];
        const unsigned long long lo = lt ? x : k[i], hi = lt ? k[i] : x;
        const float clo = lt ? xc : c[i], chi = lt ? c[i] : xc;
        k[i] = lo; c[i] = clo; x = hi; xc = chi;
    }
}

// ------------------------------------------------------------------------------------------------ resolve, staged rows
// Round 1's kernel (one wave per (candidate cell, 64 queries) event list, every lane fetching its own rows with 17 dwordx4
// loads) was bound by the L1's access rate: 17 cache accesses per row, TCP_TOTAL_CACHE_ACCESSES = 1.1e9 per launch = 0.94
// per CU and cycle, 40 % of them for the query rows, which it fetched once per candidate cell.  Here
//  * a wave keeps its 64 queries (lane = query: both half-lane lists of the query are pooled) over the window cells of
//    one window column, so a query row is fetched once per 5 candidate cells;
//  * the 64 rows of a round are fetched by the whole wave: one global_load_lds_dwordx4 covers 4 rows x 16 pieces of
//    16 bytes (256 contiguous bytes per row = a few cache accesses instead of 16), straight into LDS; every lane then
//    reads its own row back with ds_read_b128 (column g of LDS row r holds piece g ^ (r & 15): conflict free) and keeps
//    it in registers, so the fetches of the next round overlap the arithmetic of this one.  The 17th piece of a row
//    (bytes 256..271) is fetched by its own lane.
// One wave per block.  The arithmetic per (query, candidate) pair is that of knn_resolve_kernel and the top 5 are ordered
// by the same (distance, index) keys, so the results are identical.
#ifndef KM_HEAVY_ENTRIES
#define KM_HEAVY_ENTRIES 32
#endif
                                 // a query with more list entries in a cell (fringes of flat regions: hundreds of events) leaves the
                                 // lane-per-query kernel: 63 lanes would wait for it round after round
#define KM_EVLIST2 30            // candidates per query and candidate cell listed in LDS at a time (more: further passes);
                                 // 30: stage + lists + offsets = 20 KB per wave = 8 waves per CU
typedef float km_f2 __attribute__((ext_vector_type(2)));
template <bool F16>
__global__ void __launch_bounds__(64, 2) knn_resolve_kernel(KmGeom a, KmResolve p)
{
    const Geom g = a.g;
    const int lane = threadIdx.x, gq = lane >> 5, col = lane & 31;
    const int win = 2 * g.win + 1;
    int b = blockIdx.x;
    const int qwave = b % a.qwaves; b /= a.qwaves;
    const int cir = b % win; const int qcell = b / win;
    const int qci = qcell % g.ncx, qcj = qcell / g.ncx;
    const int ci = qci - g.win + cir;
    if (ci < 0 || ci >= g.ncx) return;
    const int qx0 = g.x0(qci), qy0 = g.y0(qcj), qcw = g.x1(qci) - qx0, qnpts = qcw * (g.y1(qcj) - qy0);
    if (qwave * KM_QPW >= qnpts) return;
    const int cimin = max(0, qci - g.win), cjmin = max(0, qcj - g.win), cjmax = min(g.ncy - 1, qcj + g.win);
    const int cx0 = g.x0(ci), ccw = g.x1(ci) - cx0;

    __shared__ __attribute__((aligned(1024))) char stage[F16 ? 16 : ROW_STAGE_BYTES];   // row_stage.h (float32 rows only)
    __shared__ uint16_t evl[KM_EVLIST2][64];                          // this query's candidate indices, [slot][lane]
    __shared__ __attribute__((aligned(16))) uint32_t s_cand[64];      // row offsets (16-byte units) of the round, [lane & 3][lane >> 2]
    typename RowsOf<F16>::type rows;
    constexpr uint32_t PIECES = RowsOf<F16>::PIECES;          // 16-byte units per descriptor row
    rows.init(stage, s_cand, lane);
    // ---- the queries of this wave
    float q[DFLOW_DESC];
    int qi = qwave * KM_QPW + lane;
    const bool qvalid = qi < qnpts;
    if (!qvalid) qi = qnpts - 1;
    const int qy = qy0 + qi / qcw, qx = qx0 + qi % qcw;
    const size_t qpix = (size_t)qy * g.W + qx;
    const bool qzero = p.qs[qpix].x == KM_Q_ZERO;          // all-zero descriptor: no events, the cell's own list is the answer
    {
        rows.issue((rs_gptr)p.d1, (uint32_t)qpix * PIECES, true);
        float4 qv[17];
        rows.fetch(qv);
#pragma unroll
        for (int k = 0; k < 17; k++) { q[4 * k] = qv[k].x; q[4 * k + 1] = qv[k].y; q[4 * k + 2] = qv[k].z; q[4 * k + 3] = qv[k].w; }
    }
    const rs_gptr d2g = (rs_gptr)p.d2;

    // counts and first entries of the query's two event lists (half-lane 0: tile rows 0..3, 8..11, ...; half-lane 1: rows
    // 4..7, 12..15, ...) of a candidate cell; fetched one cell ahead
    const int nrows = cjmax - cjmin + 1;
    int nA_n, nB_n; uint32_t entA_n[4], entB_n[4];
    auto prefetch = [&](int cj) {
        const size_t lid = list_id(a, qcell, qwave, (ci - cimin) * nrows + (cj - cjmin));
        nA_n = p.ev_cnt[lid * 128 + gq * 64 + col]; nB_n = p.ev_cnt[lid * 128 + gq * 64 + col + 32];
        const uint32_t *ev = p.ev + lid * KM_LIST_WORDS(a.evrows) + (size_t)gq * a.evrows * 64 + col;
#pragma unroll
        for (int e = 0; e < 4; e++) { entA_n[e] = ev[e * 64]; entB_n[e] = ev[e * 64 + 32]; }     // entries past the count are ignored below
    };
    prefetch(cjmin);
    for (int cj = cjmin; cj <= cjmax; cj++) {
        const int wslot = (ci - cimin) * nrows + (cj - cjmin);   // reference order: ci outer, cj inner (Q2)
        const size_t lid = list_id(a, qcell, qwave, wslot);
        const int cy0 = g.y0(cj);
        const int ntiles = (ccw * (g.y1(cj) - cy0) + KM_CHUNK - 1) / KM_CHUNK * (KM_CHUNK / 32);   // as in the screen kernel
        int nA = nA_n, nB = nB_n;
        bool ovf = nA == 255 || nB == 255;
        const uint32_t *evA = p.ev + (size_t)lid * KM_LIST_WORDS(a.evrows) + (size_t)gq * a.evrows * 64 + col;
        uint32_t entA[4], entB[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { entA[e] = e < nA && !ovf ? entA_n[e] : 0u; entB[e] = e < nB && !ovf ? entB_n[e] : 0u; }
        if (cj < cjmax) prefetch(cj + 1);
        if (__ballot(ovf)) {                             // an event list ran out of entries in the screen: exact redo by knn_fix_kernel
            if (lane == 0) {
                int pos = atomicAdd(p.ovf_count, 1);
                if (pos < p.ovf_cap) p.ovf_list[pos] = make_int4(qcell, qwave * KM_QPW, ci, cj);
            }
            continue;
        }
        // a query with very many events in this cell goes to knn_resolve_heavy_kernel (its lane idles here); if that kernel's
        // list is full the query stays (slow, still exact)
        bool heavy = false;
        if (nA + nB > KM_HEAVY_ENTRIES) {
            heavy = true;
            if (qvalid) {
                const int pos = atomicAdd(p.heavy_count, 1);
                if (pos < p.heavy_cap) p.heavy_list[pos] = make_int4(qcell, qi, ci, cj); else heavy = false;
            }
            if (heavy) {
                nA = 0; nB = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) { entA[e] = 0u; entB[e] = 0u; }
            }
        }
        unsigned long long keys[5];
        float costs[5];
#pragma unroll
        for (int i = 0; i < 5; i++) { keys[i] = 0x7F800000FFFFFFFFull; costs[i] = 0.0f; }   // (+inf, no index)
        // The LDS list holds KM_EVLIST2 candidates per query; a query with more events (flat or repetitive image regions)
        // takes further passes over its entries: `done` events have been evaluated, the next KM_EVLIST2 are listed.
        int done = 0, total;
        do {
        int nev = 0, seen = 0;
        auto expand = [&](uint32_t entry, int h) {
            const int tile = (int)(entry >> 16);
            uint32_t mask = entry & 0xFFFFu;
            while (mask) {
                const int r = __ffs(mask) - 1;
                mask &= mask - 1;
                // accumulator register r of half-lane h = tile row 4 h + (r & 3) + 8 (r >> 2) = candidate row * ntiles + tile
                if (seen >= done && nev < KM_EVLIST2) { evl[nev][lane] = (uint16_t)((4 * h + (r & 3) + 8 * (r >> 2)) * ntiles + tile); nev++; }
                seen++;
            }
        };
#pragma unroll
        for (int e = 0; e < 4; e++) expand(entA[e], 0);
        for (int e = 4; e < nA; e++) expand(evA[e * 64], 0);
#pragma unroll
        for (int e = 0; e < 4; e++) expand(entB[e], 1);
        for (int e = 4; e < nB; e++) expand(evA[e * 64 + 32], 1);
        total = seen;
        auto issue_cand = [&](int e) -> int {
            const bool act = e < nev;
            const int idx = act ? evl[e][lane] : 0;
            rows.issue(d2g, (uint32_t)((cy0 + idx / ccw) * g.W + cx0 + idx % ccw) * PIECES, act);
            return idx;
        };
        if (__ballot(nev > 0)) {
            int e = 0;
            int idx_next = issue_cand(0);
            while (true) {
                float4 cv[17];
                rows.fetch(cv);
                const bool act = e < nev;
                const int idx = idx_next;
                e++;
                const bool more = __ballot(e < nev) != 0;
                if (more) idx_next = issue_cand(e);
                // ---- canonical distance (sequential fmaf chain) and L1 cost (numpy pairwise order), as in knn_resolve_kernel
                const float worst = __uint_as_float((unsigned)(keys[4] >> 32));
                float acc = 0.0f, rs[8], tl[4];
#pragma unroll
                for (int k = 0; k < 17; k++) {
                    const float4 v = cv[k];
                    // two differences per instruction (v_pk_add_f32 with negated operand: the same IEEE subtraction per half)
                    const km_f2 ea = (km_f2){q[4 * k], q[4 * k + 1]} - (km_f2){v.x, v.y};
                    const km_f2 eb = (km_f2){q[4 * k + 2], q[4 * k + 3]} - (km_f2){v.z, v.w};
                    const float e0 = ea.x, e1 = ea.y, e2 = eb.x, e3 = eb.y;
                    acc = __fmaf_rn(e0, e0, acc); acc = __fmaf_rn(e1, e1, acc);
                    acc = __fmaf_rn(e2, e2, acc); acc = __fmaf_rn(e3, e3, acc);
                    const int j = (4 * k) & 7;
                    if (k < 2) { rs[j] = fabsf(e0); rs[j + 1] = fabsf(e1); rs[j + 2] = fabsf(e2); rs[j + 3] = fabsf(e3); }
                    else if (k < 16) { rs[j] = rs[j] + fabsf(e0); rs[j + 1] = rs[j + 1] + fabsf(e1); rs[j + 2] = rs[j + 2] + fabsf(e2); rs[j + 3] = rs[j + 3] + fabsf(e3); }
                    else { tl[0] = fabsf(e0); tl[1] = fabsf(e1); tl[2] = fabsf(e2); tl[3] = fabsf(e3); }
                }
                if (act && !(acc > worst)) {
                    float l1 = ((rs[0] + rs[1]) + (rs[2] + rs[3])) + ((rs[4] + rs[5]) + (rs[6] + rs[7]));
                    l1 = l1 + tl[0]; l1 = l1 + tl[1]; l1 = l1 + tl[2]; l1 = l1 + tl[3];
                    key_insert(keys, costs, ((unsigned long long)__float_as_uint(acc) << 32) | (unsigned)idx, l1);
                }
                if (!more) break;
            }
        }
        done += nev;
        } while (__ballot(done < total));
        if (qzero) {
            const KmCellTop zt = p.ztop[cj * g.ncx + ci];
#pragma unroll
            for (int i = 0; i < 5; i++) { keys[i] = zt.idx[i]; costs[i] = zt.cost[i]; }
        }
        // ---- emit (daisy i flann.py:174-180)
        if (qvalid && !heavy) {
            const size_t pix = (size_t)qy * g.W + qx;
            const int slot_base = 5 * wslot;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int idx = (int)(keys[j] & 0xFFFFFFFFu);
                const float s = costs[j];
                const int ty = cy0 + idx / ccw, tx = cx0 + idx % ccw;
                p.proposals[pix * a.LP + slot_base + j] = pack_flow(ty - qy, tx - qx);
                p.lcosts[pix * a.LP + slot_base + j] = s < a.tphi ? s : a.tphi;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ resolve, heavy queries
// One wave per (query, cell) pair that knn_resolve_kernel handed over: the query's row sits in every lane's registers, the
// lanes take DIFFERENT candidates (64 events per round instead of one), every lane keeps the 5 best of those it saw, and the
// 64 partial lists are merged at the end by five wave-wide minima over (distance, index) keys -- the same keys, so the same
// five winners in the same order as the sequential scan.  Same arithmetic per pair as knn_resolve_kernel.
#define KM_HEAVY_CAND 1024       // candidates expanded per batch: 64 list entries of up to 16 events
template <bool F16>
__global__ void __launch_bounds__(64, 2) knn_resolve_heavy_kernel(KmGeom a, KmResolve p)
{
    const Geom g = a.g;
    const int lane = threadIdx.x;
    __shared__ __attribute__((aligned(1024))) char stage[F16 ? 16 : ROW_STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) uint32_t s_cand[64];
    __shared__ uint16_t cand[KM_HEAVY_CAND];
    typename RowsOf<F16>::type rows;
    constexpr uint32_t PIECES = RowsOf<F16>::PIECES;
    rows.init(stage, s_cand, lane);
    const int nitems = min(*p.heavy_count, p.heavy_cap);
    const rs_gptr d2g = (rs_gptr)p.d2;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {          // wave-uniform
        const int4 it = p.heavy_list[item];
        const int qcell = it.x, qi = it.y, ci = it.z, cj = it.w;
        const int qci = qcell % g.ncx, qcj = qcell / g.ncx;
        const int qx0 = g.x0(qci), qy0 = g.y0(qcj), qcw = g.x1(qci) - qx0;
        const int qy = qy0 + qi / qcw, qx = qx0 + qi % qcw;
        const size_t qpix = (size_t)qy * g.W + qx;
        const int cimin = max(0, qci - g.win), cjmin = max(0, qcj - g.win), cjmax = min(g.ncy - 1, qcj + g.win);
        const int wslot = (ci - cimin) * (cjmax - cjmin + 1) + (cj - cjmin);
        const int cx0 = g.x0(ci), cy0 = g.y0(cj), ccw = g.x1(ci) - cx0;
        const int ntiles = (ccw * (g.y1(cj) - cy0) + KM_CHUNK - 1) / KM_CHUNK * (KM_CHUNK / 32);
        const size_t lid = list_id(a, qcell, qi / KM_QPW, wslot);
        const int ql = qi % KM_QPW, gq = ql >> 5, col = ql & 31;            // the query's lane in its list
        const uint32_t *ev = p.ev + (size_t)lid * KM_LIST_WORDS(a.evrows) + (size_t)gq * a.evrows * 64 + col;
        // the query's row, in every lane; the list lengths and the first 64 entries of both lists are requested beside it
        // (one memory round trip instead of three)
        float q[DFLOW_DESC];
        int nA, nB;
        uint32_t pre_a, pre_b;
        {
            rows.issue((rs_gptr)p.d1, (uint32_t)qpix * PIECES, true);
            nA = p.ev_cnt[lid * 128 + gq * 64 + col]; nB = p.ev_cnt[lid * 128 + gq * 64 + col + 32];
            const int te = min(lane, a.evrows - 1);
            pre_a = ev[(size_t)te * 64]; pre_b = ev[(size_t)te * 64 + 32];
            float4 qv[17];
            rows.fetch(qv);
#pragma unroll
            for (int k = 0; k < 17; k++) { q[4 * k] = qv[k].x; q[4 * k + 1] = qv[k].y; q[4 * k + 2] = qv[k].z; q[4 * k + 3] = qv[k].w; }
        }
        unsigned long long keys[5];
        float costs[5];
#pragma unroll
        for (int i = 0; i < 5; i++) { keys[i] = 0x7F800000FFFFFFFFull; costs[i] = 0.0f; }
        // batches of 64 list entries (first list, then second): lane = entry, its events' candidate indices go to LDS at the
        // wave-wide prefix of the popcounts
        for (int e0 = 0; e0 < nA + nB; e0 += 64) {
            const int t = e0 + lane;
            uint32_t en = 0u; int h = 0;
            // entry t of the concatenated lists: the first 64 of each list are in the lanes' registers already
            const uint32_t shb = (uint32_t)__shfl((int)pre_b, (t - nA) & 63);
            if (t < nA) en = t < 64 ? pre_a : ev[(size_t)t * 64];
            else if (t < nA + nB) { en = t - nA < 64 ? shb : ev[(size_t)(t - nA) * 64 + 32]; h = 1; }
            const int cntl = t < nA + nB ? __popc(en & 0xFFFFu) : 0;
            int incl = cntl;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
            const int nc = __shfl(incl, 63);                    // candidates of this batch (<= 1024)
            {
                int o = incl - cntl;
                const int tile = (int)(en >> 16);
                uint32_t m = t < nA + nB ? (en & 0xFFFFu) : 0u;
                while (m) { const int r = __ffs(m) - 1; m &= m - 1; cand[o++] = (uint16_t)((4 * h + (r & 3) + 8 * (r >> 2)) * ntiles + tile); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // rounds of 64 candidates; the fetch of the next round overlaps this round's arithmetic
            auto issue_cand = [&](int c0) -> int {
                const bool act = c0 + lane < nc;
                const int idx = act ? cand[c0 + lane] : 0;
                rows.issue(d2g, (uint32_t)((cy0 + idx / ccw) * g.W + cx0 + idx % ccw) * PIECES, act);
                return idx;
            };
            int idx_next = issue_cand(0);
            for (int c0 = 0; c0 < nc; c0 += 64) {
                float4 cv[17];
                rows.fetch(cv);
                const bool act = c0 + lane < nc;
                const int idx = idx_next;
                if (c0 + 64 < nc) idx_next = issue_cand(c0 + 64);
                const float worst = __uint_as_float((unsigned)(keys[4] >> 32));
                float acc = 0.0f, rs[8], tl[4];
#pragma unroll
                for (int k = 0; k < 17; k++) {
                    const float4 v = cv[k];
                    const km_f2 ea = (km_f2){q[4 * k], q[4 * k + 1]} - (km_f2){v.x, v.y};
                    const km_f2 eb = (km_f2){q[4 * k + 2], q[4 * k + 3]} - (km_f2){v.z, v.w};
                    const float d0 = ea.x, d1 = ea.y, d2 = eb.x, d3 = eb.y;
                    acc = __fmaf_rn(d0, d0, acc); acc = __fmaf_rn(d1, d1, acc);
                    acc = __fmaf_rn(d2, d2, acc); acc = __fmaf_rn(d3, d3, acc);
                    const int j = (4 * k) & 7;
                    if (k < 2) { rs[j] = fabsf(d0); rs[j + 1] = fabsf(d1); rs[j + 2] = fabsf(d2); rs[j + 3] = fabsf(d3); }
                    else if (k < 16) { rs[j] = rs[j] + fabsf(d0); rs[j + 1] = rs[j + 1] + fabsf(d1); rs[j + 2] = rs[j + 2] + fabsf(d2); rs[j + 3] = rs[j + 3] + fabsf(d3); }
                    else { tl[0] = fabsf(d0); tl[1] = fabsf(d1); tl[2] = fabsf(d2); tl[3] = fabsf(d3); }
                }
                if (act && !(acc > worst)) {
                    float l1 = ((rs[0] + rs[1]) + (rs[2] + rs[3])) + ((rs[4] + rs[5]) + (rs[6] + rs[7]));
                    l1 = l1 + tl[0]; l1 = l1 + tl[1]; l1 = l1 + tl[2]; l1 = l1 + tl[3];
                    key_insert(keys, costs, ((unsigned long long)__float_as_uint(acc) << 32) | (unsigned)idx, l1);
                }
            }
        }
        // ---- merge: five times the smallest head of the 64 sorted partial lists (keys are unique: the index is part of them)
        unsigned long long rkey = 0x7F800000FFFFFFFFull;
        float rcost = 0.0f;
        for (int r = 0; r < 5; r++) {
            unsigned long long m = keys[0];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(m, off); m = o < m ? o : m; }
            const unsigned long long own = __ballot(keys[0] == m && m != 0x7F800000FFFFFFFFull);
            float c = 0.0f;
            if (own) {
                const int wl = __ffsll((long long)own) - 1;
                c = __shfl(costs[0], wl);
                if (lane == wl) {                                 // pop
#pragma unroll
                    for (int i = 0; i < 4; i++) { keys[i] = keys[i + 1]; costs[i] = costs[i + 1]; }
                    keys[4] = 0x7F800000FFFFFFFFull; costs[4] = 0.0f;
                }
            }
            if (lane == r) { rkey = m; rcost = c; }
        }
        // ---- emit (daisy i flann.py:174-180): lane j writes winner j
        if (lane < 5) {
            const int idx = (int)(rkey & 0xFFFFFFFFu);
            const int ty = cy0 + idx / ccw, tx = cx0 + idx % ccw;
            p.proposals[qpix * a.LP + 5 * wslot + lane] = pack_flow(ty - qy, tx - qx);
            p.lcosts[qpix * a.LP + 5 * wslot + lane] = rcost < a.tphi ? rcost : a.tphi;
        }
    }
}

// ------------------------------------------------------------------------------------------------ finalize
// nprop = 5 x window cells (daisy i flann.py:189), WTA label = first minimum of the costs with strict '<'
// from 1000.0 (:93,181-184), fills beyond nprop (:89-90).  16 lanes per pixel: coalesced reads of the cost row,
// (cost, slot) lexicographic minimum over the 16 lanes with DPP row shifts, coalesced fills.
__global__ void __launch_bounds__(256) knn_finalize_kernel(Geom g, int LP, uint32_t *__restrict__ proposals, float *__restrict__ lcosts,
                                                           int32_t *__restrict__ nprop, int32_t *__restrict__ bestlabels)
{
    const int sub = threadIdx.x & 15;
    const int pix = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool ok = pix < g.H * g.W;
    const int px = ok ? pix : 0;
    const int cy = g.celly(px / g.W), cx = g.cellx(px % g.W);
    const int wy = min(g.ncy - 1, cy + g.win) - max(0, cy - g.win) + 1;
    const int wx = min(g.ncx - 1, cx + g.win) - max(0, cx - g.win) + 1;
    const int n = 5 * wy * wx;
    const float *lc = lcosts + (size_t)px * LP;
    float mind = 1000.0f; int best = 0x7fffffff;
    // all of a lane's costs are fetched before the first comparison (a rolled loop waits for every load in turn)
    float cs[DFLOW_MAX_LABELS / 16];
#pragma unroll
    for (int j = 0; j < DFLOW_MAX_LABELS / 16; j++) cs[j] = sub + 16 * j < n ? lc[sub + 16 * j] : 1000.0f;
#pragma unroll
    for (int j = 0; j < DFLOW_MAX_LABELS / 16; j++) if (cs[j] < mind) { mind = cs[j]; best = sub + 16 * j; }
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1) {       // row_shl 1,2,4,8: lane 0 of every 16-lane row ends with the row minimum
        const int ctrl = 0x100 + sh;
        float om; int ob;
        if (sh == 1) { om = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, mind), __builtin_bit_cast(int, mind), 0x101, 0xF, 0xF, false)); ob = __builtin_amdgcn_update_dpp(best, best, 0x101, 0xF, 0xF, false); }
        else if (sh == 2) { om = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, mind), __builtin_bit_cast(int, mind), 0x102, 0xF, 0xF, false)); ob = __builtin_amdgcn_update_dpp(best, best, 0x102, 0xF, 0xF, false); }
        else if (sh == 4) { om = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, mind), __builtin_bit_cast(int, mind), 0x104, 0xF, 0xF, false)); ob = __builtin_amdgcn_update_dpp(best, best, 0x104, 0xF, 0xF, false); }
        else { om = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, mind), __builtin_bit_cast(int, mind), 0x108, 0xF, 0xF, false)); ob = __builtin_amdgcn_update_dpp(best, best, 0x108, 0xF, 0xF, false); }
        (void)ctrl;
        if (om < mind || (om == mind && ob < best)) { mind = om; best = ob; }
    }
    if (ok) {
        if (sub == 0) { nprop[pix] = n; bestlabels[pix] = best == 0x7fffffff ? 0 : best; }
        for (int l = n + sub; l < LP; l += 16) { proposals[(size_t)pix * LP + l] = DFLOW_FILL_PROPOSAL; lcosts[(size_t)pix * LP + l] = DFLOW_FILL_COST; }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int max_cell_points(const Geom &g)
{
    return (g.x1(g.ncx - 1) - g.x0(g.ncx - 1)) * (g.y1(g.ncy - 1) - g.y0(g.ncy - 1));
}

static size_t num_lists(const dflow_params *p)
{
    Geom g = make_geom(p);
    size_t qwaves = (max_cell_points(g) + KM_QPW - 1) / KM_QPW, win = 2 * g.win + 1;
    return (size_t)g.ncx * g.ncy * qwaves * win * win;
}

#define KM_HEAVY_BLOCKS 2048      // waves of knn_resolve_heavy_kernel (grid-stride over the pairs; 8 per CU)
#define KM_HEAVY_CAP (1 << 20)    // (query, cell) pairs the cooperative kernel can take per pass (16 MB); more stay with knn_resolve_kernel
// entries per lane of the event lists: one per tile of the largest cell, + 1 (the last entry of a list is never valid), so
// that no list can run out; cells beyond KM_EVROWS_MAX - 1 tiles: KM_EVROWS_MAX
static int km_evrows(const dflow_params *p)
{
    const int tiles = km_pad(max_cell_points(make_geom(p))) / 32;
    return tiles + 1 < KM_EVROWS_MAX ? tiles + 1 : KM_EVROWS_MAX;
}

// the kNN stage's part of the workspace
struct KmWs {
    _Float16 *h1, *h2;          // prepared rows: image 1 in pixel order, image 2 cell by cell in tile position order
    float2 *qs, *z0;            // per image-1 pixel (flag, 0.5 |x|^2); per image-2 position (|c|^2, |c|_1)
    uint8_t *zflag;             // per image-2 position: all-zero row
    int *ctr;                   // [0] lists handed to knn_fix_kernel, [1] flags (bit 0: the basis failed its check), then one
    int *cell_bad;              //     flag per candidate cell; zeroed together at the start of a call
    size_t zero_bytes;
    float *vt;                  // principal axes, [component][dimension]
    void *pca_ws;
    KmCellTop *ztop;
    int4 *ovf;                  // one slot per list
    int4 *heavy;                // KM_HEAVY_CAP (query, cell) pairs for knn_resolve_heavy_kernel
    uint32_t *ev;
    uint8_t *ev_cnt;
    size_t nl, bytes;
    int evrows;
};

static KmWs km_ws(const dflow_params *p, void *ws)
{
    const Geom g = make_geom(p);
    const size_t N = (size_t)g.H * g.W, rows2 = km_total_rows(g), ncells = (size_t)g.ncx * g.ncy;
    auto align256 = [](char *w) { return (char *)(((uintptr_t)w + 255) & ~(uintptr_t)255); };
    KmWs k;
    char *w = (char *)ws;
    k.nl = num_lists(p);
    k.evrows = km_evrows(p);
    k.h1 = (_Float16 *)w; w += N * KM_K * sizeof(_Float16);
    k.h2 = (_Float16 *)w; w += rows2 * KM_K * sizeof(_Float16);
    k.qs = (float2 *)w; w += N * sizeof(float2);
    k.z0 = (float2 *)w; w += rows2 * sizeof(float2);
    k.zflag = (uint8_t *)w; w += rows2;
    w = align256(w);
    k.ctr = (int *)w; w += 256;
    k.cell_bad = (int *)w; w += ncells * sizeof(int);
    k.zero_bytes = 256 + ncells * sizeof(int);
    w = align256(w);
    k.vt = (float *)w; w += DFLOW_DESC * DFLOW_DESC * sizeof(float);
    w = align256(w);
    k.pca_ws = w; w += knn_pca_ws_bytes();
    w = align256(w);
    k.ztop = (KmCellTop *)w; w += ncells * sizeof(KmCellTop);
    w = align256(w);
    k.ovf = (int4 *)w; w += k.nl * sizeof(int4);
    w = align256(w);
    k.heavy = (int4 *)w; w += (size_t)KM_HEAVY_CAP * sizeof(int4);
    w = align256(w);
    k.ev = (uint32_t *)w; w += k.nl * KM_LIST_WORDS(k.evrows) * sizeof(uint32_t);
    k.ev_cnt = (uint8_t *)w; w += k.nl * 128;
    k.bytes = (size_t)(w - (char *)ws) + 256;
    return k;
}

size_t knn_mfma_ws_bytes(const dflow_params *p) { return km_ws(p, nullptr).bytes; }

bool knn_mfma_supported(const dflow_params *p)
{
    Geom g = make_geom(p);
    return max_cell_points(g) <= KM_MAXPTS && p->window >= 0 && p->window <= 2;
}

int launch_knn_fix(const dflow_params *p, const void *d1, const void *d2, uint32_t *proposals, float *lcosts,
                   const int *ovf_count, const int4 *ovf_list, int ovf_cap, const int *flags, hipStream_t s);

// ev (optional, profiling): KNN_MFMA_EVENTS events recorded on s at the boundaries basis | prep | screen | resolve | fix | finalize
int launch_knn_mfma(const dflow_params *p, const void *d1, const void *d2, uint32_t *proposals, float *lcosts,
                    int32_t *nprop, int32_t *bestlabels, void *ws, hipStream_t s, hipEvent_t *tev)
{
    auto mark = [&](int k) { if (tev) (void)hipEventRecord(tev[k], s); };
    Geom g = make_geom(p);
    const size_t N = (size_t)g.H * g.W;
    const KmWs k = km_ws(p, ws);
    if (hipMemsetAsync(k.ctr, 0, k.zero_bytes, s) != hipSuccess)
        return dflow_set_error(DFLOW_EHIP, "hipMemsetAsync failed in launch_knn_mfma");
    int nb = (int)((N + 255) / 256);
    mark(0);
    const bool f16 = descr_f16(p);
    int rc = launch_knn_pca(d2, f16, k.vt, k.ctr + 1, k.pca_ws, (int)N, s);
    if (rc) return rc;
    mark(1);
    const dim3 cgrid((km_pad(max_cell_points(g)) + 255) / 256, g.ncx * g.ncy);
    if (f16) {
        hipLaunchKernelGGL(knn_prep_kernel<_Float16>, dim3(nb), dim3(256), 0, s, (const _Float16 *)d1, (const float *)k.vt, k.h1, k.qs, (float2 *)nullptr, (uint8_t *)nullptr, (int *)nullptr, g, 0);
        hipLaunchKernelGGL(knn_prep_kernel<_Float16>, cgrid, dim3(256), 0, s, (const _Float16 *)d2, (const float *)k.vt, k.h2, (float2 *)nullptr, k.z0, k.zflag, k.cell_bad, g, 1);
    } else {
        hipLaunchKernelGGL(knn_prep_kernel<float>, dim3(nb), dim3(256), 0, s, (const float *)d1, (const float *)k.vt, k.h1, k.qs, (float2 *)nullptr, (uint8_t *)nullptr, (int *)nullptr, g, 0);
        hipLaunchKernelGGL(knn_prep_kernel<float>, cgrid, dim3(256), 0, s, (const float *)d2, (const float *)k.vt, k.h2, (float2 *)nullptr, k.z0, k.zflag, k.cell_bad, g, 1);
    }
    hipLaunchKernelGGL(knn_cell_post_kernel, dim3(g.ncx * g.ncy), dim3(256), 0, s, k.h2, (const float2 *)k.z0, (const uint8_t *)k.zflag, k.ztop, g);
    rc = dflow_check_launch("knn_cell_post_kernel");
    if (rc) return rc;

    mark(2);
    KmGeom a;
    a.g = g; a.LP = p->label_pitch; a.tphi = p->tphi;
    a.qwaves = (max_cell_points(g) + KM_QPW - 1) / KM_QPW;
    a.evrows = k.evrows;
    int win = 2 * g.win + 1;
    int wgs_per_cell = (win * win * a.qwaves + KM_WAVES - 1) / KM_WAVES;
    KmScreen sc;
    sc.h1 = k.h1; sc.h2 = k.h2; sc.qs = k.qs; sc.cell_bad = k.cell_bad; sc.ev = k.ev; sc.ev_cnt = k.ev_cnt;
    size_t shmem = (size_t)KM_NBUF * KM_ABUF;
    hipLaunchKernelGGL(knn_screen_kernel, dim3(g.ncx * g.ncy * wgs_per_cell), dim3(KM_THREADS), shmem, s, a, sc);
    rc = dflow_check_launch("knn_screen_kernel");
    if (rc) return rc;
    mark(3);
    KmResolve rs;
    rs.d1 = d1; rs.d2 = d2; rs.ev = k.ev; rs.ev_cnt = k.ev_cnt; rs.qs = k.qs; rs.ztop = k.ztop; rs.proposals = proposals; rs.lcosts = lcosts;
    rs.ovf_count = k.ctr; rs.ovf_list = k.ovf; rs.ovf_cap = (int)k.nl;
    rs.heavy_count = k.ctr + 2; rs.heavy_list = k.heavy; rs.heavy_cap = KM_HEAVY_CAP;
    if (f16) hipLaunchKernelGGL(knn_resolve_kernel<true>, dim3((unsigned)(g.ncx * g.ncy * win * a.qwaves)), dim3(64), 0, s, a, rs);
    else hipLaunchKernelGGL(knn_resolve_kernel<false>, dim3((unsigned)(g.ncx * g.ncy * win * a.qwaves)), dim3(64), 0, s, a, rs);
    rc = dflow_check_launch("knn_resolve_kernel");
    if (rc) return rc;
    // queries with hundreds of events in a cell (knn_resolve_kernel listed them): one wave each, its lanes over the events
    if (f16) hipLaunchKernelGGL(knn_resolve_heavy_kernel<true>, dim3(KM_HEAVY_BLOCKS), dim3(64), 0, s, a, rs);
    else hipLaunchKernelGGL(knn_resolve_heavy_kernel<false>, dim3(KM_HEAVY_BLOCKS), dim3(64), 0, s, a, rs);
    rc = dflow_check_launch("knn_resolve_heavy_kernel");
    if (rc) return rc;
    mark(4);
    rc = launch_knn_fix(p, d1, d2, proposals, lcosts, k.ctr, k.ovf, (int)k.nl, k.ctr + 1, s);
    if (rc) return rc;
    mark(5);
    hipLaunchKernelGGL(knn_finalize_kernel, dim3((unsigned)((N * 16 + 255) / 256)), dim3(256), 0, s, g, a.LP, proposals, lcosts, nprop, bestlabels);
    mark(6);
    return dflow_check_launch("knn_finalize_kernel");
}

// ------------------------------------------------------------------------------------------------ statistics (measurement aid)
// What the screen left in the workspace: one block per event list (threads = 2 groups x 64 lanes), then one pass over the
// image-1 records and the image-2 positions.
__global__ void __launch_bounds__(128) knn_stats_lists_kernel(const uint32_t *__restrict__ ev, const uint8_t *__restrict__ ev_cnt, int evrows,
                                                              unsigned long long *__restrict__ out)
{
    const size_t lid = blockIdx.x;
    const int gq = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cnt = ev_cnt[lid * 128 + threadIdx.x];
    unsigned long long entries = 0, events = 0;
    if (cnt != 255) {
        entries = (unsigned long long)cnt;
        const uint32_t *e = ev + lid * KM_LIST_WORDS(evrows) + (size_t)gq * evrows * 64 + lane;
        for (int i = 0; i < cnt; i++) events += (unsigned long long)__popc(e[i * 64] & 0xFFFFu);
    }
    unsigned long long mx = entries;
    for (int off = 32; off > 0; off >>= 1) { entries += __shfl_xor(entries, off); events += __shfl_xor(events, off); const unsigned long long o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
    if (lane == 0) { atomicAdd(out + 0, entries); atomicAdd(out + 1, events); atomicMax(out + 2, mx); }
}

__global__ void __launch_bounds__(256) knn_stats_rows_kernel(const float2 *__restrict__ qs, int npix, const uint8_t *__restrict__ zflag,
                                                             const _Float16 *__restrict__ h2, size_t rows2, unsigned long long *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool zq = i < (size_t)npix && qs[i].x == KM_Q_ZERO, bq = i < (size_t)npix && qs[i].x == KM_Q_BAD;
    const bool zc = i < rows2 && zflag[i] != 0;
    const bool removed = zc && h2[i * KM_K + KM_SLOT_H] == (_Float16)60000.0f;
    const unsigned long long b0 = __ballot(zq), b1 = __ballot(bq), b2 = __ballot(zc), b3 = __ballot(removed);
    if ((threadIdx.x & 63) == 0) {
        if (b0) atomicAdd(out + 3, (unsigned long long)__popcll(b0));
        if (b1) atomicAdd(out + 4, (unsigned long long)__popcll(b1));
        if (b2) atomicAdd(out + 5, (unsigned long long)__popcll(b2));
        if (b3) atomicAdd(out + 6, (unsigned long long)__popcll(b3));
    }
}

// h_out[KNN_STATS_N] (host), after a dflow_knn_proposals call on the same workspace: see include/dflow.h
int knn_mfma_stats(const dflow_params *p, void *ws, hipStream_t s, int64_t *h_out)
{
    const KmWs k = km_ws(p, ws);
    const Geom g = make_geom(p);
    const size_t N = (size_t)g.H * g.W, rows2 = km_total_rows(g);
    unsigned long long *dev = (unsigned long long *)(k.ctr + 16);            // 8 counters inside the zeroed control block
    if (hipMemsetAsync(dev, 0, 8 * sizeof(unsigned long long), s) != hipSuccess) return dflow_set_error(DFLOW_EHIP, "hipMemsetAsync failed");
    hipLaunchKernelGGL(knn_stats_lists_kernel, dim3((unsigned)k.nl), dim3(128), 0, s, (const uint32_t *)k.ev, (const uint8_t *)k.ev_cnt, k.evrows, dev);
    const size_t m = N > rows2 ? N : rows2;
    hipLaunchKernelGGL(knn_stats_rows_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const float2 *)k.qs, (int)N, (const uint8_t *)k.zflag,
                       (const _Float16 *)k.h2, rows2, dev);
    int rc = dflow_check_launch("knn_stats kernels");
    if (rc) return rc;
    unsigned long long c[8];
    int ctr[4];
    if (hipMemcpyAsync(c, dev, sizeof(c), hipMemcpyDeviceToHost, s) != hipSuccess || hipMemcpyAsync(ctr, k.ctr, sizeof(ctr), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return dflow_set_error(DFLOW_EHIP, "knn_mfma_stats: copy back failed");
    // (query, cell) pairs of the pass
    long long pairs = 0;
    for (int qcj = 0; qcj < g.ncy; qcj++)
        for (int qci = 0; qci < g.ncx; qci++) {
            const long long qn = (long long)(g.x1(qci) - g.x0(qci)) * (g.y1(qcj) - g.y0(qcj));
            const int nx = min(g.ncx - 1, qci + g.win) - max(0, qci - g.win) + 1, ny = min(g.ncy - 1, qcj + g.win) - max(0, qcj - g.win) + 1;
            pairs += qn * nx * ny;
        }
    h_out[0] = ctr[0]; h_out[1] = ctr[1]; h_out[2] = (int64_t)k.nl; h_out[3] = (int64_t)c[0]; h_out[4] = (int64_t)c[1]; h_out[5] = (int64_t)c[2];
    h_out[6] = (int64_t)c[3]; h_out[7] = (int64_t)c[4]; h_out[8] = (int64_t)c[5]; h_out[9] = (int64_t)c[6]; h_out[10] = pairs; h_out[11] = k.evrows;
    h_out[12] = ctr[2] < KM_HEAVY_CAP ? ctr[2] : KM_HEAVY_CAP;
    return DFLOW_OK;
}

// MFMA instructions the screen issues for these parameters (2 passes x KM_KSTEPS per 32 x 32 tile of padded cells), for
// bench.py's "issued flops": each is 32 x 32 x 16 x 2 flop
double knn_mfma_issued(const dflow_params *p)
{
    Geom g = make_geom(p);
    double n = 0.0;
    for (int qcj = 0; qcj < g.ncy; qcj++)
        for (int qci = 0; qci < g.ncx; qci++) {
            const int qnpts = (g.x1(qci) - g.x0(qci)) * (g.y1(qcj) - g.y0(qcj));
            const int qgroups = 2 * ((qnpts + KM_QPW - 1) / KM_QPW);               // 32-query column groups incl. padding
            for (int ci = qci - g.win; ci <= qci + g.win; ci++)
                for (int cj = qcj - g.win; cj <= qcj + g.win; cj++) {
                    if (ci < 0 || ci >= g.ncx || cj < 0 || cj >= g.ncy) continue;
                    const int cnpts = (g.x1(ci) - g.x0(ci)) * (g.y1(cj) - g.y0(cj));
                    n += (double)qgroups * (km_pad(cnpts) / 32) * 2.0 * KM_KSTEPS;
                }
        }
    return n;
}
