// C-ABI entry points of libdflow.so (declared in include/dflow.h): parameter validation, workspace
// accounting and dispatch to the per-stage launchers.  No torch types, no allocation, no synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dflow_common.h"

static thread_local char g_err[512] = "";

int dflow_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int dflow_check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dflow_set_error(DFLOW_EHIP, "%s: %s", what, hipGetErrorString(e));
    return DFLOW_OK;
}

int dflow_check_params(const dflow_params *p)
{
    if (!p) return dflow_set_error(DFLOW_EINVAL, "params is NULL");
    if (p->pich < 8 || p->picw < 8 || p->pich > 8192 || p->picw > 8192)
        return dflow_set_error(DFLOW_EINVAL, "image size %dx%d outside [8,8192]", p->picw, p->pich);
    if (p->cellh < 1 || p->cellw < 1 || p->cellh > p->pich || p->cellw > p->picw)
        return dflow_set_error(DFLOW_EINVAL, "cell size %dx%d does not fit the image", p->cellw, p->cellh);
    if (p->knn != 5) return dflow_set_error(DFLOW_EINVAL, "knn=%d unsupported (kernels are built for 5)", p->knn);
    if (p->cellh * p->cellw < p->knn) return dflow_set_error(DFLOW_EINVAL, "cells hold fewer than knn points");
    if (p->window < 0 || p->window > 2) return dflow_set_error(DFLOW_EINVAL, "window=%d outside [0,2]", p->window);
    if (p->ngauss < 0 || p->ngauss > 64) return dflow_set_error(DFLOW_EINVAL, "ngauss=%d outside [0,64]", p->ngauss);
    int maxknn = (2 * p->window + 1) * (2 * p->window + 1) * p->knn;
    if (p->maxnprop < maxknn + p->ngauss || p->maxnprop > DFLOW_MAX_LABELS)
        return dflow_set_error(DFLOW_EINVAL, "maxnprop=%d must be in [%d,%d]", p->maxnprop, maxknn + p->ngauss, DFLOW_MAX_LABELS);
    if (p->label_pitch < p->maxnprop || p->label_pitch % 16 != 0 || p->label_pitch > DFLOW_MAX_LABELS)
        return dflow_set_error(DFLOW_EINVAL, "label_pitch=%d must be a multiple of 16 in [maxnprop,%d]", p->label_pitch, DFLOW_MAX_LABELS);
    // pair costs below tpsi travel as 3-bit fields in the BCD label records (bcd.hip)
    if (p->tpsi < 1 || p->tpsi > 8) return dflow_set_error(DFLOW_EINVAL, "tpsi=%d outside [1,8]", p->tpsi);
    if (!(p->sigma > 0.0f) || p->sigma > 8.0f) return dflow_set_error(DFLOW_EINVAL, "sigma=%g outside (0,8]", (double)p->sigma);
    if (p->max_attempts < p->ngauss) return dflow_set_error(DFLOW_EINVAL, "max_attempts < ngauss");
    if (p->flags & ~(DFLOW_FLAG_KNN_EXACT | DFLOW_FLAG_DESCR_F16)) return dflow_set_error(DFLOW_EINVAL, "unknown flags 0x%x", (unsigned)p->flags);
    return DFLOW_OK;
}

extern "C" {

int dflow_version(void) { return DFLOW_VERSION; }

const char *dflow_last_error(void) { return g_err; }

void dflow_default_params(dflow_params *p, int32_t pich, int32_t picw, int32_t cellh, int32_t cellw)
{
    memset(p, 0, sizeof(*p));
    p->pich = pich; p->picw = picw; p->cellh = cellh; p->cellw = cellw;
    p->maxnprop = 150; p->knn = 5; p->window = 2; p->ngauss = 25; p->tpsi = 8; p->max_attempts = 1 << 16;
    p->tphi = 2.5f; p->sigma = 8.0f; p->lamda = 0.05; p->seed = 0; p->label_pitch = 160;
}

size_t dflow_workspace_bytes(const dflow_params *p)
{
    if (dflow_check_params(p) != DFLOW_OK) return 0;
    size_t a = daisy_ws_bytes(p), b = bcd_ws_bytes(p), c = knn_mfma_supported(p) ? knn_mfma_ws_bytes(p) : 0;
    size_t d = neighbour_ws_bytes(p);
    size_t m = a > b ? a : b;
    if (d > m) m = d;
    return (m > c ? m : c) + 256;
}

#define CHECK_PTR(x) do { if (!(x)) return dflow_set_error(DFLOW_EINVAL, "%s: %s is NULL", __func__, #x); } while (0)
#define CHECK_WS(need) do { if (!d_ws || ws_bytes < (need)) \
    return dflow_set_error(DFLOW_ENOSPC, "%s: workspace %zu < %zu bytes", __func__, ws_bytes, (size_t)(need)); } while (0)

int dflow_daisy(const dflow_params *p, const uint8_t *d_bgr, void *d_descr, void *d_ws, size_t ws_bytes, void *stream)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_bgr); CHECK_PTR(d_descr); CHECK_WS(daisy_ws_bytes(p));
    return launch_daisy(p, d_bgr, d_descr, d_ws, (hipStream_t)stream);
}

int dflow_knn_proposals(const dflow_params *p, const void *d_descr1, const void *d_descr2, uint32_t *d_proposals,
                        float *d_lcosts, int32_t *d_nprop, int32_t *d_bestlabels, void *d_ws, size_t ws_bytes, void *stream)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_descr1); CHECK_PTR(d_descr2); CHECK_PTR(d_proposals); CHECK_PTR(d_lcosts); CHECK_PTR(d_nprop); CHECK_PTR(d_bestlabels);
    // DFLOW_FLAG_KNN_EXACT selects the brute-force VALU kernel (same results; used to cross-check the MFMA path)
    if ((p->flags & DFLOW_FLAG_KNN_EXACT) || !knn_mfma_supported(p))
        return launch_knn(p, d_descr1, d_descr2, d_proposals, d_lcosts, d_nprop, d_bestlabels, (hipStream_t)stream);
    CHECK_WS(knn_mfma_ws_bytes(p));
    return launch_knn_mfma(p, d_descr1, d_descr2, d_proposals, d_lcosts, d_nprop, d_bestlabels, d_ws, (hipStream_t)stream);
}

int dflow_knn_proposals_timed(const dflow_params *p, const void *d_descr1, const void *d_descr2, uint32_t *d_proposals,
                              float *d_lcosts, int32_t *d_nprop, int32_t *d_bestlabels, void *d_ws, size_t ws_bytes, void *stream,
                              float *h_ms, double *h_mfma_issued)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_descr1); CHECK_PTR(d_descr2); CHECK_PTR(d_proposals); CHECK_PTR(d_lcosts); CHECK_PTR(d_nprop); CHECK_PTR(d_bestlabels);
    CHECK_PTR(h_ms);
    if ((p->flags & DFLOW_FLAG_KNN_EXACT) || !knn_mfma_supported(p))
        return dflow_set_error(DFLOW_EINVAL, "%s: the MFMA-screened search does not run for these parameters", __func__);
    CHECK_WS(knn_mfma_ws_bytes(p));
    hipEvent_t ev[KNN_MFMA_EVENTS];
    for (int k = 0; k < KNN_MFMA_EVENTS; k++)
        if (hipEventCreate(&ev[k]) != hipSuccess) return dflow_set_error(DFLOW_EHIP, "hipEventCreate failed");
    rc = launch_knn_mfma(p, d_descr1, d_descr2, d_proposals, d_lcosts, d_nprop, d_bestlabels, d_ws, (hipStream_t)stream, ev);
    if (rc == DFLOW_OK && hipEventSynchronize(ev[KNN_MFMA_EVENTS - 1]) != hipSuccess) rc = dflow_set_error(DFLOW_EHIP, "hipEventSynchronize failed");
    for (int k = 0; rc == DFLOW_OK && k + 1 < KNN_MFMA_EVENTS; k++)
        if (hipEventElapsedTime(&h_ms[k], ev[k], ev[k + 1]) != hipSuccess) rc = dflow_set_error(DFLOW_EHIP, "hipEventElapsedTime failed");
    for (int k = 0; k < KNN_MFMA_EVENTS; k++) (void)hipEventDestroy(ev[k]);
    if (h_mfma_issued) *h_mfma_issued = knn_mfma_issued(p);
    return rc;
}

int dflow_knn_screen_stats(const dflow_params *p, void *d_ws, size_t ws_bytes, void *stream, int64_t *h_stats)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(h_stats);
    if ((p->flags & DFLOW_FLAG_KNN_EXACT) || !knn_mfma_supported(p))
        return dflow_set_error(DFLOW_EINVAL, "%s: the MFMA-screened search does not run for these parameters", __func__);
    CHECK_WS(knn_mfma_ws_bytes(p));
    return knn_mfma_stats(p, d_ws, (hipStream_t)stream, h_stats);
}

int dflow_neighbour_proposals(const dflow_params *p, const void *d_descr1, const void *d_descr2, uint32_t *d_proposals,
                              float *d_lcosts, int32_t *d_nprop, const int32_t *d_bestlabels, void *d_ws, size_t ws_bytes,
                              void *stream)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_descr1); CHECK_PTR(d_descr2); CHECK_PTR(d_proposals); CHECK_PTR(d_lcosts); CHECK_PTR(d_nprop); CHECK_PTR(d_bestlabels);
    CHECK_WS(neighbour_ws_bytes(p));
    return launch_neighbour(p, d_descr1, d_descr2, d_proposals, d_lcosts, d_nprop, d_bestlabels, d_ws, (hipStream_t)stream);
}

int dflow_bcd_prepare(const dflow_params *p, const uint32_t *d_proposals, const float *d_lcosts, const int32_t *d_nprop,
                      void *d_ws, size_t ws_bytes, void *stream)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_proposals); CHECK_PTR(d_lcosts); CHECK_PTR(d_nprop); CHECK_WS(bcd_ws_bytes(p));
    return launch_bcd_prepare(p, d_proposals, d_lcosts, d_nprop, d_ws, (hipStream_t)stream);
}

int dflow_bcd_phase(const dflow_params *p, const uint32_t *d_proposals, const int32_t *d_nprop, int32_t *d_bestlabels,
                    int32_t phase, void *d_ws, size_t ws_bytes, void *stream)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_proposals); CHECK_PTR(d_nprop); CHECK_PTR(d_bestlabels); CHECK_WS(bcd_ws_bytes(p));
    if (phase < 0 || phase > 3) return dflow_set_error(DFLOW_EINVAL, "phase=%d outside [0,3]", phase);
    return launch_bcd_phase(p, d_proposals, d_nprop, d_bestlabels, phase, d_ws, (hipStream_t)stream);
}

int dflow_bcd_sweep(const dflow_params *p, const uint32_t *d_proposals, const int32_t *d_nprop, int32_t *d_bestlabels,
                    void *d_ws, size_t ws_bytes, void *stream)
{
    for (int ph = 0; ph < 4; ph++) {
        int rc = dflow_bcd_phase(p, d_proposals, d_nprop, d_bestlabels, ph, d_ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return DFLOW_OK;
}

int dflow_bcd_phase_batch(const dflow_params *p, int32_t npass, const int32_t *const *d_nprop, int32_t *const *d_bestlabels,
                          int32_t phase, void *const *d_ws, size_t ws_bytes, void *stream)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_nprop); CHECK_PTR(d_bestlabels); CHECK_PTR(d_ws);
    if (npass < 1 || npass > 1024) return dflow_set_error(DFLOW_EINVAL, "npass=%d outside [1,1024]", npass);
    if (phase < 0 || phase > 3) return dflow_set_error(DFLOW_EINVAL, "phase=%d outside [0,3]", phase);
    if (ws_bytes < bcd_ws_bytes(p)) return dflow_set_error(DFLOW_ENOSPC, "%s: workspace %zu < %zu bytes", __func__, ws_bytes, bcd_ws_bytes(p));
    for (int i = 0; i < npass; i++)
        if (!d_nprop[i] || !d_bestlabels[i] || !d_ws[i]) return dflow_set_error(DFLOW_EINVAL, "%s: pass %d has a NULL pointer", __func__, i);
    return launch_bcd_phase_batch(p, npass, d_nprop, d_bestlabels, phase, d_ws, (hipStream_t)stream);
}

int dflow_bcd_sweep_batch(const dflow_params *p, int32_t npass, const int32_t *const *d_nprop, int32_t *const *d_bestlabels,
                          void *const *d_ws, size_t ws_bytes, void *stream)
{
    for (int ph = 0; ph < 4; ph++) {
        int rc = dflow_bcd_phase_batch(p, npass, d_nprop, d_bestlabels, ph, d_ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return DFLOW_OK;
}

int dflow_labels_to_flow(const dflow_params *p, const uint32_t *d_proposals, const int32_t *d_bestlabels, float *d_flow,
                         void *stream)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_proposals); CHECK_PTR(d_bestlabels); CHECK_PTR(d_flow);
    return launch_labels_to_flow(p, d_proposals, d_bestlabels, d_flow, (hipStream_t)stream);
}

int dflow_fb_consistency(const dflow_params *p, const float *d_fwd, const float *d_bwd, float tresh, float *d_sparse,
                         void *stream)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_fwd); CHECK_PTR(d_bwd); CHECK_PTR(d_sparse);
    return launch_fb_consistency(p, d_fwd, d_bwd, tresh, d_sparse, (hipStream_t)stream);
}

int dflow_pack_compat(const dflow_params *p, const uint32_t *d_proposals, const int32_t *d_nprop, uint8_t *d_packed,
                      void *stream)
{
    int rc = dflow_check_params(p); if (rc) return rc;
    CHECK_PTR(d_proposals); CHECK_PTR(d_nprop); CHECK_PTR(d_packed);
    return launch_pack_compat(p, d_proposals, d_nprop, d_packed, (hipStream_t)stream);
}

int dflow_remove_small_segments_host(float *h_sparse, int32_t dim0, int32_t dim1, float tresh, int32_t min_segment_size)
{
    if (!h_sparse) return dflow_set_error(DFLOW_EINVAL, "h_sparse is NULL");
    if (dim0 <= 0 || dim1 <= 0) return dflow_set_error(DFLOW_EINVAL, "field size %dx%d", dim0, dim1);
    return host_remove_small_segments(h_sparse, dim0, dim1, tresh, min_segment_size);
}

}  // extern "C"
