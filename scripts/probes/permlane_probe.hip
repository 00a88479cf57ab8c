#include <hip/hip_runtime.h>
__global__ void k(unsigned *o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%u ", h[i]);
    printf("\n");
    for (int i = 0; i < 64; ++i) printf("%u ", h[64 + i]);
    printf("\n");
    return 0;
}
