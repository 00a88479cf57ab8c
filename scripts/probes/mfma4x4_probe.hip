// layout probe of v_mfma_f64_4x4x4f64 (4 blocks): unit impulses in A lane la and B lane lb -> which D lane lights up
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int *out) {
    const int l = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
    if (d != 0.0) out[la * 64 + lb] = l;
}
int main() {
    int *dout, h[4096];
    hipMalloc(&dout, sizeof(h));
    hipMemset(dout, 0xff, sizeof(h));
    probe<<<dim3(64, 64), 64>>>(dout);
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d pairs with B lanes:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf(" %d->D%d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
