// Streaming-pattern probe for the symv: same bytes (lower triangle of an n x n f64 matrix in 64x64 tiles, 8 x 16-byte
// loads per thread per tile, next tile prefetched), tiles of a workgroup's strip taken along a ROW block (mode 0, the
// current kernel: 512-byte runs, stride ld between them) or down a COLUMN block (mode 1: every column of the strip is
// one contiguous K x 512-byte run).   hipcc --offload-arch=gfx950 -O3 tile_stream_probe.hip -o tile_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const double *__restrict__ a, int64_t ld, int side, int ktiles, double *sink) {
    const int nsx = (side + ktiles - 1) / ktiles;
    const int blk = blockIdx.x / nsx;             // row block (mode 0) or column block (mode 1)
    const int t0 = (blockIdx.x - blk * nsx) * ktiles;
    int tb, te;
    if (MODE == 0) { if (t0 > blk) return; tb = t0; te = min(t0 + ktiles, blk + 1); }          // column tiles 0..blk
    else { tb = blk + t0; if (tb >= side) return; te = min(tb + ktiles, side); }                // row tiles blk..side-1
    const int tid = threadIdx.x, rp = tid & 31, cg = tid >> 5;
    double2 tv[8];
    auto load = [&](int t) {
        const int R = (MODE == 0) ? blk : t, C = (MODE == 0) ? t : blk;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            tv[q] = *reinterpret_cast<const double2 *>(a + (int64_t)(R * 64 + 2 * rp) + (int64_t)(C * 64 + cg + 8 * q) * ld);
    };
    double acc = 0.0;
    load(tb);
    for (int t = tb; t < te; ++t) {
        double2 cur[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cur[q] = tv[q];
        if (t + 1 < te) load(t + 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += cur[q].x + cur[q].y;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
    const int side = n / 64;
    const int64_t ld = n;
    double *a, *sink;
    hipMalloc(&a, (size_t)n * n * 8);
    hipMalloc(&sink, 8);
    hipMemset(a, 0, (size_t)n * n * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)side * (side + 1) / 2 * 32768.0;
    for (int kt : {8, 32, 64}) {
        for (int mode = 0; mode < 2; ++mode) {
            const int nsx = (side + kt - 1) / kt;
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(side * nsx), dim3(256), 0, 0, a, ld, side, kt, sink);
                else hipLaunchKernelGGL(k<1>, dim3(side * nsx), dim3(256), 0, 0, a, ld, side, kt, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("n=%d ktiles=%d mode=%d (%s): %.3f ms  %.2f TB/s\n", n, kt, mode, mode ? "column strips" : "row strips", best, bytes / best / 1e9);
        }
    }
    return 0;
}
