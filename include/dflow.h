/*
 * dflow.h -- C-ABI of libdflow.so: the MI355X (gfx950) dense discrete optical-flow stage.
 *
 * The reference (pfe-rs/lk-s-2022-estimacija-pokreta) has no FFI: its hot path is two flat Python scripts
 * that talk through .npy files.  Each entry point below replaces the reference function(s) named next to
 * it; INTEGRATION.md shows the ctypes binding a maintainer would add to the reference scripts.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  All d_* pointers are DEVICE pointers owned by the caller.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous with
 *     respect to the host and ordered on that stream; the library never synchronises and never allocates
 *     device memory: scratch comes from the caller's workspace (dflow_workspace_bytes).
 *   - Return 0 on success, a negative DFLOW_E* code otherwise; dflow_last_error() gives the message
 *     (thread-local).  No global state; re-entrant across devices.
 *
 * Device layouts (H = pich, W = picw, LP = label_pitch >= maxnprop, multiple of 16)
 *   image      uint8   (H,W,3)   BGR, as cv2.imread returns it            daisy i flann.py:26-27,52-53
 *   descr      float32 (H,W,68)  row y*W+x = keypoint order               daisy i flann.py:69-77
 *              or, with DFLOW_FLAG_DESCR_F16, binary16 (H,W,72): 68 values + 4 zero pads per pixel (144-byte rows)
 *   proposals  uint32  (H,W,LP)  one label = int16 dy | int16 dx << 16,   daisy i flann.py:89 (int64 (H,W,150,2), -1 fill)
 *                                unused slots 0xFFFFFFFF (= [-1,-1])
 *   lcosts     float32 (H,W,LP)  unused slots 1000.0f                     daisy i flann.py:90 (float64; values are float32-exact)
 *   nprop      int32   (H,W)                                              daisy i flann.py:91
 *   bestlabels int32   (H,W)                                              daisy i flann.py:95
 *   flow       float32 (H,W,2)   [dy,dx]                                  python bcd.py:90-95 (float64; values are small integers)
 *   sparse     float32 (H,W,3)   [U=dx, V=dy, valid]                      postprocessing.py:7-17,123-135
 */
#ifndef DFLOW_H
#define DFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFLOW_VERSION 1
#define DFLOW_DESC 68            /* (4 rings * 4 angles + 1) * 4 bins, daisy i flann.py:66 */
#define DFLOW_MAX_LABELS 160     /* kernels are built for maxnprop <= 160 (reference: 150) */

/* dflow_params.flags */
#define DFLOW_FLAG_KNN_EXACT 1   /* dflow_knn_proposals: brute-force VALU search instead of the MFMA-screened one (same
                                    results bit for bit; the cross-check of the screen's error bound) */

#define DFLOW_FLAG_DESCR_F16 8    /* BASELINE configs[4] "fp16 DAISY descriptors": dflow_daisy rounds the descriptor values to IEEE
                                    binary16 (round to nearest even) and stores them as binary16, (H,W,72) with 4 zero pads per
                                    pixel; dflow_knn_proposals / dflow_neighbour_proposals called with the flag read that layout.
                                    All arithmetic downstream (canonical float32 distance, costs, BCD) is on those values widened to
                                    float32; the reference has no such mode (cv2 returns float32) */
#define DFLOW_DESC_PITCH_F16 72  /* elements per pixel of a binary16 descriptor plane */

#define DFLOW_OK 0
#define DFLOW_EINVAL (-1)        /* bad parameter / null pointer / unsupported geometry */
#define DFLOW_ENOSPC (-2)        /* workspace too small */
#define DFLOW_EHIP (-3)          /* HIP runtime error (launch failure, no device) */

/* Algorithm constants.  Field-for-field the module globals of the reference scripts. */
typedef struct dflow_params {
    int32_t pich, picw;          /* daisy i flann.py:34-35 */
    int32_t cellh, cellw;        /* daisy i flann.py:42-43; ragged last row/column of cells absorbs the remainder */
    int32_t maxnprop;            /* daisy i flann.py:88   (150) */
    int32_t knn;                 /* daisy i flann.py:172  (5; kernels require 5) */
    int32_t window;              /* daisy i flann.py:167-168 (2 cells each side) */
    int32_t ngauss;              /* daisy i flann.py:207  (25) */
    int32_t tpsi;                /* daisy i flann.py:47   (8; kernels support 1..8) */
    int32_t max_attempts;        /* bound on draws per pixel in the neighbour sampler (65536) */
    float tphi;                  /* daisy i flann.py:46   (2.5) */
    float sigma;                 /* daisy i flann.py:208  (8)  */
    double lamda;                /* daisy i flann.py:48   (0.05) */
    uint64_t seed;               /* key of the counter-based sampler (reference: unseeded np.random, :219) */
    int32_t label_pitch;         /* LP, elements per pixel in proposals/lcosts (160; multiple of 16) */
    int32_t flags;               /* DFLOW_FLAG_* bits (0 = defaults); per call, the library keeps no process-global switches */
} dflow_params;

int dflow_version(void);
const char *dflow_last_error(void);

/* Fills *p with the reference constants for the given geometry. */
void dflow_default_params(dflow_params *p, int32_t pich, int32_t picw, int32_t cellh, int32_t cellw);

/* Scratch bytes any entry point may need for these parameters (one buffer serves all stages). */
size_t dflow_workspace_bytes(const dflow_params *p);

/* izracunajDaisy, daisy i flann.py:69-77 (cv2.xfeatures2d.DAISY_create(radius=5,q_radius=4,q_theta=4,q_hist=4)
 * .compute on every pixel).  d_bgr (H,W,3) uint8 -> d_descr (H,W,68) float32 (binary16 (H,W,72) with DFLOW_FLAG_DESCR_F16). */
int dflow_daisy(const dflow_params *p, const uint8_t *d_bgr, void *d_descr,
                void *d_ws, size_t ws_bytes, void *stream);

/* napraviCD2 + generisi, daisy i flann.py:144-189: per-cell exact 5-NN proposals, truncated-L1 costs,
 * WTA labels.  Initialises and fills proposals/lcosts/nprop/bestlabels. */
int dflow_knn_proposals(const dflow_params *p, const void *d_descr1, const void *d_descr2,
                        uint32_t *d_proposals, float *d_lcosts, int32_t *d_nprop, int32_t *d_bestlabels,
                        void *d_ws, size_t ws_bytes, void *stream);

/* Measurement aid (bench.py's roofline object): the launches of dflow_knn_proposals with HIP events between the kernels on
 * `stream`.  Same kernels, same results; unlike every other entry point it WAITS for the stream.  h_ms[6] (host) receives the
 * milliseconds of { basis (centre, covariance, principal axes), prep (both images), knn_screen_kernel, knn_resolve_kernel,
 * knn_fix_kernel, knn_finalize_kernel }, *h_mfma_issued (host, optional) the number of v_mfma_f32_32x32x16_f16 (32768 flop
 * each) the screen issues for these parameters.  No reference counterpart. */
int dflow_knn_proposals_timed(const dflow_params *p, const void *d_descr1, const void *d_descr2,
                              uint32_t *d_proposals, float *d_lcosts, int32_t *d_nprop, int32_t *d_bestlabels,
                              void *d_ws, size_t ws_bytes, void *stream, float *h_ms, double *h_mfma_issued);

/* Measurement aid (bench.py's other_configs, the low-texture tests): what the MFMA screen of the LAST dflow_knn_proposals call on
 * this workspace did, read back from the workspace (WAITS for the stream; call it before another stage reuses the workspace).
 * h_stats[DFLOW_KNN_STATS_N] (host): [0] event lists handed to the exact brute-force search (rows outside the f16 range / NaN),
 * [1] flags (bit 0: the whole pass went to the exact search: the basis failed its orthonormality check), [2] event lists of the
 * pass, [3] list entries written, [4] events = (query, candidate) pairs evaluated exactly, [5] most entries in one lane's list,
 * [6] all-zero queries (answered from their cells' own lists), [7] queries outside the screen's range, [8] all-zero candidate
 * rows, [9] of those removed as duplicates, [10] (query, cell) pairs of the pass, [11] list capacity per lane, [12] (query, cell)
 * pairs with so many events that one wave took the query alone (its 64 lanes over the events).
 * No reference counterpart. */
#define DFLOW_KNN_STATS_N 13
int dflow_knn_screen_stats(const dflow_params *p, void *d_ws, size_t ws_bytes, void *stream, int64_t *h_stats);

/* nasumicni, daisy i flann.py:205-233: appends up to ngauss neighbour proposals per pixel (in place).
 * d_bestlabels must still hold the WTA labels written by dflow_knn_proposals.  Uses 4 bytes per pixel of the workspace
 * (the WTA flow of every pixel, gathered once). */
int dflow_neighbour_proposals(const dflow_params *p, const void *d_descr1, const void *d_descr2,
                              uint32_t *d_proposals, float *d_lcosts, int32_t *d_nprop,
                              const int32_t *d_bestlabels, void *d_ws, size_t ws_bytes, void *stream);

/* Builds the compat bit matrices of pakovanje (daisy i flann.py:256-309) into the workspace, in the layout the chain
 * kernel reads: per pixel, per chain direction and per label one record with the compatible labels of the predecessor
 * on that chain and their pairwise costs, plus per pixel every label's own flow and data cost (what ucitajSvePodatkeDoBCD,
 * python bcd.py:67-81, loads for bcd()).  Must be called after proposals and lcosts are final (after
 * dflow_neighbour_proposals / an upload) and before dflow_bcd_phase / dflow_bcd_sweep; the records stay valid until
 * another dflow_* stage call (daisy, knn) reuses the same workspace. */
int dflow_bcd_prepare(const dflow_params *p, const uint32_t *d_proposals, const float *d_lcosts, const int32_t *d_nprop,
                      void *d_ws, size_t ws_bytes, void *stream);

/* One of the four loops of ceoBCD's body, python bcd.py:265-277 (phase 0 even columns top->bottom,
 * 1 even rows right->left, 2 odd columns bottom->top, 3 odd rows left->right); every chain is one call of
 * bcd(), python bcd.py:101-257, reading the records dflow_bcd_prepare left in the workspace (label costs included).
 * Updates d_bestlabels in place. */
int dflow_bcd_phase(const dflow_params *p, const uint32_t *d_proposals, const int32_t *d_nprop,
                    int32_t *d_bestlabels, int32_t phase, void *d_ws, size_t ws_bytes, void *stream);

/* One iteration of ceoBCD's loop (all four phases), python bcd.py:264-277. */
int dflow_bcd_sweep(const dflow_params *p, const uint32_t *d_proposals, const int32_t *d_nprop,
                    int32_t *d_bestlabels, void *d_ws, size_t ws_bytes, void *stream);

/* The same loops for a BATCH of independent passes ((pair, direction) runs share nothing: README.md:40 of the reference)
 * with identical parameters: the chains of all npass passes go into one launch, which fills the GPU where one pass
 * alone (218-512 chains) cannot.  d_nprop / d_bestlabels / d_ws are HOST arrays of npass device pointers; every pass has its
 * own workspace of ws_bytes, prepared by dflow_bcd_prepare.  Results are identical to npass separate calls. */
int dflow_bcd_phase_batch(const dflow_params *p, int32_t npass, const int32_t *const *d_nprop, int32_t *const *d_bestlabels,
                          int32_t phase, void *const *d_ws, size_t ws_bytes, void *stream);
int dflow_bcd_sweep_batch(const dflow_params *p, int32_t npass, const int32_t *const *d_nprop, int32_t *const *d_bestlabels,
                          void *const *d_ws, size_t ws_bytes, void *stream);

/* vratiKonacniFlow, python bcd.py:90-95 / daisy i flann.py:192-197. */
int dflow_labels_to_flow(const dflow_params *p, const uint32_t *d_proposals, const int32_t *d_bestlabels,
                         float *d_flow, void *stream);

/* postProcessing = FlowImage.ucitajFlow x2 + fowardBackwardConsistency, postprocessing.py:7-17,79-135
 * (including its transposed indexing).  d_fwd/d_bwd (H,W,2) [dy,dx] -> d_sparse (H,W,3) [U,V,valid]. */
int dflow_fb_consistency(const dflow_params *p, const float *d_fwd, const float *d_bwd, float tresh,
                         float *d_sparse, void *stream);

/* The packedksets file of pakovanje, daisy i flann.py:256-309, for users who feed the reference's own `python bcd.py`:
 * d_packed (H,W,2,maxnprop*maxnprop/8+1) uint8, slot 0 = pixel vs the pixel below, slot 1 = vs the pixel to the right,
 * np.packbits bit order.  Every matrix is computed from clean scratch; the reference's bottom-row / right-column loops
 * (:283-307) reuse theirs, which the host wrapper (compat.py) replays on those H+W matrices. */
int dflow_pack_compat(const dflow_params *p, const uint32_t *d_proposals, const int32_t *d_nprop, uint8_t *d_packed,
                      void *stream);

/* removeSmallSegments, postprocessing.py:29-76 (unused upstream, :129), on a HOST (dim0,dim1,3) float32 [U,V,valid]
 * field, in place.  Host code: the reference's region growing depends on its scan order. */
int dflow_remove_small_segments_host(float *h_sparse, int32_t dim0, int32_t dim1, float tresh, int32_t min_segment_size);

#ifdef __cplusplus
}
#endif
#endif /* DFLOW_H */
