// times knn_cov_kernel + knn_jacobi_kernel alone: for S in 0 1 5; do hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DPCA_SWEEPS=$S ...
#include "../../lk-s-2022-estimacija-pokreta_amd/csrc/knn_pca.hip"
#include <stdio.h>
#include <vector>
int dflow_set_error(int code, const char *fmt, ...) { return code; }
int dflow_check_launch(const char *what) { return hipGetLastError() == hipSuccess ? 0 : -3; }
int main()
{
    const int npix = 446464;
    std::vector<float> d((size_t)npix * 68);
    srand(1);
    // correlated data: smooth random walk per dimension
    for (int p = 0; p < npix; p++) for (int k = 0; k < 68; k++) d[(size_t)p * 68 + k] = 0.3f * sinf(0.001f * p * (1 + k % 7)) + 0.1f * (rand() / (float)RAND_MAX) * (1.0f / (1 + k));
    float *dd, *mu, *vt; int *flags; void *ws;
    hipMalloc(&dd, d.size() * 4); hipMalloc(&mu, 512); hipMalloc(&vt, 68 * 68 * 8); hipMalloc(&flags, 256); hipMalloc(&ws, knn_pca_ws_bytes());
    hipMemcpy(dd, d.data(), d.size() * 4, hipMemcpyHostToDevice); hipMemset(mu, 0, 512); hipMemset(flags, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        launch_knn_pca(dd, false, mu, vt, flags, ws, npix, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int f; hipMemcpy(&f, flags, 4, hipMemcpyDeviceToHost);
        printf("sweeps %d: cov + jacobi %.3f ms (flag %d)\n", PCA_SWEEPS, ms, f);
    }
    return 0;
}
