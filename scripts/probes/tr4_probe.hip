// Probe (GPU box): lane map of ds_read_b64_tr_b4 on gfx950: for every output lane and each of its 16 nibbles, which supplier lane and
// which nibble of that lane's 8 bytes it came from.   hipcc -O2 --offload-arch=gfx950 tr4_probe.hip -o tr4_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v2i __attribute__((ext_vector_type(2)));

// pass p: nibble value = bits [4p, 4p + 4) of the source nibble's global index (lane * 16 + nibble): 3 passes cover 1024 indices
__global__ void tr4_kernel(int pass, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t sm[64 * 8];
    for (int i = threadIdx.x; i < 512; i += 64) {
        const int n0 = 2 * i, n1 = 2 * i + 1;                 // nibble indices of this byte (low, high)
        sm[i] = (uint8_t)(((n0 >> (4 * pass)) & 15) | (((n1 >> (4 * pass)) & 15) << 4));
    }
    __syncthreads();
    typedef __attribute__((address_space(3))) v2i lds_v2i;
    v2i r = __builtin_amdgcn_ds_read_tr4_b64_v2i32((lds_v2i *)(sm + threadIdx.x * 8));
    out[threadIdx.x * 2] = (uint32_t)r.x;
    out[threadIdx.x * 2 + 1] = (uint32_t)r.y;
}

int main() {
    uint32_t *d, h[3][128];
    if (hipMalloc(&d, 512) != hipSuccess) return 1;
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(tr4_kernel, dim3(1), dim3(64), 0, 0, p, d);
        if (hipMemcpy(h[p], d, 512, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    }
    printf("ds_read_b64_tr_b4: output lane i, nibble q <- (supplier lane, nibble of its 8 bytes)\n");
    for (int i = 0; i < 64; ++i) {
        printf("lane %2d:", i);
        for (int q = 0; q < 16; ++q) {
            int idx = 0;
            for (int p = 0; p < 3; ++p) {
                const uint32_t w = h[p][i * 2 + (q >> 3)];
                idx |= (int)((w >> (4 * (q & 7))) & 15u) << (4 * p);
            }
            printf(" (%2d,%2d)", idx >> 4, idx & 15);
        }
        printf("\n");
    }
    return 0;
}
