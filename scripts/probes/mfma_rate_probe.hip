// issue rate of v_mfma_f64_4x4x4 against v_mfma_f64_16x16x4 (independent accumulators, one wave per SIMD and four)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k4(double *out, int iters) {
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[j], 0, 0, 0);
    }
    double s = 0;
    for (int j = 0; j < 8; ++j) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k16(double *out, int iters) {
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
    d4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = (d4){0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
    }
    double s = 0;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *d;
    hipMalloc(&d, 8 * 1024 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves = 1; waves <= 4; waves *= 2) {        // waves per SIMD (block = 256 threads = 1 wave per SIMD)
        for (int which = 0; which < 2; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which == 0) k4<<<256 * waves, 256>>>(d, iters); else k16<<<256 * waves, 256>>>(d, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double per = best * 1e-3 / ((double)iters * 8 * waves);   // seconds per instruction per SIMD
            printf("%s waves/SIMD %d: %.1f ns per instruction per SIMD (%.1f cycles at 2.4 GHz), %.1f TFLOP/s chip\n", which ? "16x16x4" : "4x4x4  ",
                   waves, per * 1e9, per * 2.4e9, (which ? 2048.0 : 512.0) / per * 1024 / 1e12);
        }
    }
    return 0;
}
