// f64 MFMA issue-rate probe for k_dgemm.hip: which ingredient of the GEMM inner loop costs matrix-pipe time?
//   hipcc --offload-arch=gfx950 -O3 dgemm_probe.hip -o dgemm_probe && ./dgemm_probe
// Every variant runs the same MFMA count (8 waves x 32 v_mfma_f64_16x16x4_f64 per 16-k step, 8 accumulators per wave) in
// 512-thread workgroups, two per CU:
//   0  MFMAs only (operands stay in registers)
//   1  + the six ds_read_b64 fragment reads per 4-k sub-step (static LDS image)
//   2  + one workgroup barrier per 16-k step
//   3  + eight ds_write_b64 per thread and step (the staging stores)
//   4  + eight global 8-byte loads per thread and step (the operand prefetch), consumed by the stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 4) void probe(const double *__restrict__ g, double *__restrict__ out, int steps, int64_t gstride) {
    constexpr int PA = 128 + 17;
    __shared__ double as[2 * 16 * PA], bs[2 * 16 * PA];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lx = lane & 15, lk = lane >> 4;
    const int wm = (wave & 3) * 32, wn = (wave >> 2) * 64;
    for (int i = t; i < 2 * 16 * PA; i += 512) {
        as[i] = 1.0 + 1e-9 * i;
        bs[i] = 1.0 - 1e-9 * i;
    }
    __syncthreads();
    d4 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = (d4){0.0, 0.0, 0.0, 0.0};
    double fa[2] = {1.0 + lane, 2.0}, fb[4] = {1.0, 0.5, 0.25, 0.125 * lane};
    double ra[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
    const double *gp = g + (int64_t)blockIdx.x * gstride + t;
    int buf = 0;
    for (int s = 0; s < steps; ++s) {
        if (MODE >= 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = gp[(int64_t)(s & 63) * 4096 + i * 512];
                rb[i] = gp[(int64_t)(s & 63) * 4096 + 2048 + i * 512];
            }
        }
        const double *ap = as + buf * 16 * PA + wm + lx;
        const double *bp = bs + buf * 16 * PA + wn + lx;
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) {
            if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = ap[(ks + lk) * PA + i * 16];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = bp[(ks + lk) * PA + j * 16];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[j][i], 0, 0, 0);
        }
        if (MODE >= 3) {
            double *ad = as + (buf ^ 1) * 16 * PA, *bd = bs + (buf ^ 1) * 16 * PA;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = i * 512 + t;
                ad[(idx / 128) * PA + idx % 128] = MODE >= 4 ? ra[i] : 1.0 + 1e-9 * idx;
                bd[(idx % 16) * PA + idx / 16] = MODE >= 4 ? rb[i] : 1.0 - 1e-9 * idx;
            }
        }
        if (MODE >= 2) {
            __syncthreads();
            if (MODE >= 3) buf ^= 1;
        }
    }
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) sum += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
    out[(int64_t)blockIdx.x * 512 + t] = sum;
}

template <int MODE>
static void run(const double *g, double *out, int wgs, int steps, int64_t gstride) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), 0, 0, g, out, steps, gstride);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(wgs), dim3(512), 0, 0, g, out, steps, gstride);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double flops = (double)wgs * 8 * 32 * 2048.0 * steps;
    printf("mode %d: %8.3f ms  %6.1f TFLOP/s\n", MODE, ms, flops / ms / 1e9);
}

int main() {
    const int wgs = 512 * 4, steps = 1024;
    const int64_t gstride = 64 * 4096;
    double *g, *out;
    hipMalloc(&g, sizeof(double) * gstride * wgs);
    hipMemset(g, 0, sizeof(double) * gstride * wgs);
    hipMalloc(&out, sizeof(double) * 512 * wgs);
    run<0>(g, out, wgs, steps, gstride);
    run<1>(g, out, wgs, steps, gstride);
    run<2>(g, out, wgs, steps, gstride);
    run<3>(g, out, wgs, steps, gstride);
    run<4>(g, out, wgs, steps, gstride);
    return 0;
}
