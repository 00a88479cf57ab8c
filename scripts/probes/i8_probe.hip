// Probe (GPU box): lane maps of ds_read_b64_tr_b8 and of the A/B operands of v_mfma_i32_32x32x32_i8 on gfx950.
//   hipcc -O2 --offload-arch=gfx950 scripts/probes/i8_probe.hip -o scripts/probes/i8_probe && scripts/probes/i8_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// mode 0: byte value = supplier lane; mode 1: byte value = byte offset inside the supplier's 8 bytes
__global__ void tr8_kernel(int mode, uint8_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t sm[64 * 8];
    for (int i = threadIdx.x; i < 512; i += 64) sm[i] = mode == 0 ? (uint8_t)(i >> 3) : (uint8_t)(i & 7);
    __syncthreads();
    typedef __attribute__((address_space(3))) v2i lds_v2i;
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(sm + threadIdx.x * 8));
    reinterpret_cast<v2i *>(out)[threadIdx.x] = r;
}

// C = A B with A (32 x 32 i8, row-major), B (32 x 32 i8, [k][col]); lane operand bytes taken under hypothesis `hyp`
__global__ void mfma_kernel(const int8_t *A, const int8_t *B, int hyp, int *C) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    int8_t a[16], b[16];
    for (int j = 0; j < 16; ++j) {
        int k = hyp == 0 ? 16 * h + j : (8 * h + (j & 7) + 16 * (j >> 3));
        a[j] = A[r * 32 + k];
        b[j] = B[k * 32 + r];
    }
    v4i av, bv;
    memcpy(&av, a, 16);
    memcpy(&bv, b, 16);
    v16i c = {};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * h;     // C/D map of the 32x32 shapes
        C[row * 32 + r] = c[q];
    }
}

int main() {
    uint8_t *d, h[2][512];
    hipMalloc(&d, 512);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(tr8_kernel, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h[mode], d, 512, hipMemcpyDeviceToHost);
    }
    printf("ds_read_b64_tr_b8: output lane i, byte e <- (supplier lane, byte offset)\n");
    for (int i = 0; i < 64; ++i) {
        printf("lane %2d:", i);
        for (int e = 0; e < 8; ++e) printf(" (%2d,%d)", h[0][i * 8 + e], h[1][i * 8 + e]);
        printf("\n");
    }
    int8_t hA[1024], hB[1024];
    srand(1);
    for (int i = 0; i < 1024; ++i) {
        hA[i] = (int8_t)(rand() % 7 - 3);
        hB[i] = (int8_t)(rand() % 5 - 2);
    }
    int ref[1024];
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int s = 0;
            for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 32 + j];
            ref[i * 32 + j] = s;
        }
    int8_t *dA, *dB;
    int *dC, hC[1024];
    hipMalloc(&dA, 1024);
    hipMalloc(&dB, 1024);
    hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; ++hyp) {
        hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, hyp, dC);
        hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += hC[i] != ref[i];
        printf("mfma_i32_32x32x32_i8 operand hypothesis %d (%s): %d mismatches\n", hyp,
               hyp == 0 ? "lane holds k = 16 h + j" : "lane holds k = 8 h + (j & 7) + 16 (j >> 3)", bad);
    }
    return 0;
}
