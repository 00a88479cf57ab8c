// Which rocSOLVER / rocBLAS calls of the eigh stage survive n > 46340 (n^2 > 2^31)?
// hipcc --offload-arch=gfx950 -O2 large_n_probe.cpp -o large_n_probe -lrocsolver -lrocblas ; ./large_n_probe 50000
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 50000;
    const int what = argc > 2 ? atoi(argv[2]) : 7;
    rocblas_handle h;
    rocblas_create_handle(&h);
    const size_t nn = (size_t)n * n;
    double *a, *c, *tau, *d, *e;
    rocblas_int *info;
    CK(hipMalloc(&a, nn * 8)); CK(hipMalloc(&c, nn * 8)); CK(hipMalloc(&tau, n * 8)); CK(hipMalloc(&d, n * 8)); CK(hipMalloc(&e, n * 8));
    CK(hipMalloc(&info, 4));
    CK(hipMemset(a, 0, nn * 8)); CK(hipMemset(c, 0, nn * 8)); CK(hipMemset(tau, 0, n * 8));
    if (what & 1) {  // dormtr with tau = 0 (identity reflectors): indexing only
        double t0 = now();
        rocblas_status rs = rocsolver_dormtr(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, n, n, a, n, tau, c, n);
        CK(hipDeviceSynchronize());
        printf("dormtr n=%d status %d  %.2f s\n", n, (int)rs, now() - t0); fflush(stdout);
    }
    if (what & 2) {  // dsyr2k on the trailing part with k = 64
        const double m1 = -1.0, one = 1.0;
        double t0 = now();
        rocblas_status rs = rocblas_dsyr2k(h, rocblas_fill_lower, rocblas_operation_none, n - 64, 64, &m1, a + 64, n, c + 64, n, &one, a + 64 + (size_t)64 * n, n);
        CK(hipDeviceSynchronize());
        printf("dsyr2k n=%d status %d  %.3f s\n", n, (int)rs, now() - t0); fflush(stdout);
    }
    if (what & 8) {  // dgemm (n/2 x n/2) * (n/2 x n) into the top half of c
        const double one = 1.0, zero = 0.0;
        const int k = n / 2;
        double t0 = now();
        rocblas_status rs = rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, k, n, k, &one, a, n, a + (size_t)k * n, n, &zero, c, n);
        CK(hipDeviceSynchronize());
        printf("dgemm (%d x %d x %d) status %d  %.3f s  %.1f TFLOP/s\n", k, n, k, (int)rs, now() - t0, 2.0 * k * (double)n * k / (now() - t0) / 1e12); fflush(stdout);
    }
    if (what & 4) {  // dstedc on a random tridiagonal of size n (expected to fail above 46340) and n/2
        for (int sz : {n / 2, n}) {
            std::vector<double> hd(sz), he(sz);
            srand(1);
            for (int i = 0; i < sz; ++i) { hd[i] = 2.0 + (rand() % 1000) * 1e-3; he[i] = 0.5 + (rand() % 1000) * 1e-3; }
            CK(hipMemcpy(d, hd.data(), sz * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(e, he.data(), sz * 8, hipMemcpyHostToDevice));
            double t0 = now();
            printf("dstedc n=%d ...\n", sz); fflush(stdout);
            rocblas_status rs = rocsolver_dstedc(h, rocblas_evect_tridiagonal, sz, d, e, c, sz, info);
            CK(hipDeviceSynchronize());
            int hi = -1; CK(hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost));
            printf("dstedc n=%d status %d info %d  %.2f s\n", sz, (int)rs, hi, now() - t0); fflush(stdout);
        }
    }
    return 0;
}
