// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with both operands fp4 (e2m1): operand layout hypothesis, exactness of integer sums in
// the f32 accumulator, and the issue rate against v_mfma_i32_32x32x32_i8.
// Hypothesis: lane l holds, for A, row l % 32 and the 32 k values 32 (l / 32) .. + 31 as the 32 nibbles of registers 0 .. 3
// (nibble q = k 32 (l / 32) + q, little endian); B likewise with its column.  build: hipcc -O3 --offload-arch=gfx950 fp4_probe.hip -o fp4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const unsigned *a, const unsigned *b, float *d, int reps) {
    const int l = threadIdx.x;
    v8i A = {0, 0, 0, 0, 0, 0, 0, 0}, B = A;
    for (int r = 0; r < 4; ++r) {
        A[r] = (int)a[l * 4 + r];
        B[r] = (int)b[l * 4 + r];
    }
    v16f c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    for (int i = 0; i < reps; ++i) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 127, 0, 127);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}

__global__ void rate_fp4(float *out, int iters) {
    v8i A = {0x22222222, 0x24242424, 0x42424242, 0x20202020, 0, 0, 0, 0}, B = {0x44444444, 0x22222222, 0x02020202, 0x24242424, 0, 0, 0, 0};
    A[0] += threadIdx.x;
    v16f c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c0, 4, 4, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c1, 4, 4, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c2, 4, 4, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c3, 4, 4, 0, 127, 0, 127);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void rate_i8(int *out, int iters) {
    v4i A = {0x01020001, 0x02010002, 0x00010201, 0x01000102}, B = {0x02020101, 0x01000201, 0x00020100, 0x02010001};
    A[0] += threadIdx.x & 1;
    v16i c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    // counts {0, 1, 2} as e2m1: 0 -> 0x0, 1.0 -> 0x2, 2.0 -> 0x4
    const unsigned enc[3] = {0u, 2u, 4u};
    std::vector<int> ma(32 * 64), mb(32 * 64);
    srand(7);
    for (auto &v : ma) v = rand() % 3;
    for (auto &v : mb) v = rand() % 3;
    std::vector<unsigned> ha(64 * 4, 0u), hb(64 * 4, 0u);
    for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 32; ++q) {
            const int k = 32 * (l / 32) + q;
            ha[l * 4 + q / 8] |= enc[ma[(l % 32) * 64 + k]] << (4 * (q % 8));
            hb[l * 4 + q / 8] |= enc[mb[(l % 32) * 64 + k]] << (4 * (q % 8));
        }
    unsigned *da, *db;
    float *dd;
    hipMalloc(&da, ha.size() * 4);
    hipMalloc(&db, hb.size() * 4);
    hipMalloc(&dd, 64 * 16 * 4);
    hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    for (int reps : {1, 40000}) {
        probe<<<1, 64>>>(da, db, dd, reps);
        std::vector<float> hd(64 * 16);
        hipMemcpy(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        double maxv = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l % 32, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                long ref = 0;
                for (int k = 0; k < 64; ++k) ref += ma[row * 64 + k] * mb[col * 64 + k];
                ref *= reps;
                if ((double)hd[l * 16 + r] != (double)ref) ++bad;
                if (ref > maxv) maxv = (double)ref;
            }
        printf("reps %d: mismatches %d of 1024 (largest exact sum %.0f, 2^24 = 16777216)\n", reps, bad, maxv);
    }
    // issue rate: 4 independent accumulators per wave, 4 waves per SIMD
    const int iters = 20000, blocks = 256 * 4, threads = 256;
    float *of;
    int *oi;
    hipMalloc(&of, blocks * threads * 4);
    hipMalloc(&oi, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(e0);
        rate_fp4<<<blocks, threads>>>(of, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = 2.0 * 32 * 32 * 64 * 4.0 * iters * (double)blocks * (threads / 64);
        if (pass) printf("fp4 32x32x64: %.3f ms  %.1f TOP/s\n", ms, ops / ms / 1e9);
        hipEventRecord(e0);
        rate_i8<<<blocks, threads>>>(oi, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double ops8 = 2.0 * 32 * 32 * 32 * 4.0 * iters * (double)blocks * (threads / 64);
        if (pass) printf("i8  32x32x32: %.3f ms  %.1f TOP/s\n", ms, ops8 / ms / 1e9);
    }
    return 0;
}
