// issue rate of v_mfma_f64_16x16x4 with NACC accumulators in rotation (dependent distance NACC), one wave per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k16(double *out, int iters) {
    double a = threadIdx.x * 0.001, b = 1.0 + threadIdx.x * 0.002;
    d4 c[NACC];
    for (int j = 0; j < NACC; ++j) c[j] = (d4){0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
    }
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(double *d, const char *tag) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k16<NACC><<<256, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double per = best * 1e-3 / ((double)iters * 8);
    printf("%s: %.1f cycles per instruction at 2.4 GHz\n", tag, per * 2.4e9);
}
int main() {
    double *d;
    hipMalloc(&d, 8 * 1024 * 1024);
    run<1>(d, "1 accumulator (back to back dependent)");
    run<2>(d, "2 accumulators");
    run<4>(d, "4 accumulators");
    run<8>(d, "8 accumulators");
    return 0;
}
