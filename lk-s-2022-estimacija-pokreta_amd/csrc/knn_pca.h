// Principal axes for the kNN screen (knn_pca.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define PCA_SAMPLES 4096

// scratch of the two kernels (partial scatter matrices)
size_t knn_pca_ws_bytes(void);
// d2: (npix, 68) float32 descriptors, mu: their centre (68 float32) -> vt: [68 components][68 dimensions] float64, rows
// sorted by decreasing eigenvalue; sets bit 0 of *flags if the basis is not orthonormal (NaN input)
int launch_knn_pca(const float *d2, const float *mu, double *vt, int *flags, void *ws, int npix, hipStream_t s);
