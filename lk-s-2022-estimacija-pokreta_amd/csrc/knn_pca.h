// Principal axes for the kNN screen (knn_pca.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define PCA_SAMPLES 4096
#define PCA_DELTA_MAX 2e-6      // guaranteed |V^T V - I|_2 of the basis (else the flag is set)

// scratch of the two kernels (partial scatter matrices)
size_t knn_pca_ws_bytes(void);
// d2: (npix, 68) float32 or (f16) (npix, 72) binary16 descriptors -> vt: [68 components][68 dimensions] float32, the principal
// axes of their second-moment matrix, rows sorted by decreasing eigenvalue; sets bit 0 of *flags if |V^T V - I|_F > PCA_DELTA_MAX
int launch_knn_pca(const void *d2, bool f16, float *vt, int *flags, void *ws, int npix, hipStream_t s);
