// Grid-barrier latency probe (cooperative launch): hipcc --offload-arch=gfx950 -O3 bar_probe.hip -o bar_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__device__ __forceinline__ void bar(unsigned *ctr, unsigned *grp, unsigned &epoch, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += 1;
        if (MODE == 0) {  // single counter, acq/rel atomics
            const unsigned target = epoch * nwg;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        } else if (MODE == 1) {  // relaxed polling + one fence
            const unsigned target = epoch * nwg;
            __threadfence();
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
            __threadfence();
        } else {  // two-level: 8 group counters (b % 8 ~ XCD), last arriver bumps the top flag
            const unsigned g = blockIdx.x & 7;
            const unsigned gsize = (nwg + 7 - g) / 8;
            __threadfence();
            const unsigned old = __hip_atomic_fetch_add(&grp[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == epoch * gsize) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * 8;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
            __threadfence();
        }
    } else {
        epoch += 1;
    }
    __syncthreads();
}

// flag-array barrier: every workgroup publishes its epoch in its own word; everybody polls all words
template <int MODE>
__device__ __forceinline__ void bar_flags(unsigned *flags, unsigned &epoch, unsigned nwg) {
    __syncthreads();
    epoch += 1;
    if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 3) {
        for (;;) {
            int ok = 1;
            for (unsigned t = threadIdx.x; t < nwg; t += blockDim.x)
                ok &= (__hip_atomic_load(&flags[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
            if (__syncthreads_and(ok)) break;
        }
        __threadfence();
    } else {
        if (threadIdx.x < 64) {
            for (;;) {
                int ok = 1;
                for (unsigned t = threadIdx.x; t < nwg; t += 64)
                    ok &= (__hip_atomic_load(&flags[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
                if (__all(ok)) break;
            }
            __threadfence();
        }
        __syncthreads();
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(unsigned *ctr, unsigned *grp, int iters, double *sink) {
    unsigned epoch = 0;
    double acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE >= 3)
            bar_flags<MODE>(grp, epoch, gridDim.x);
        else
            bar<MODE>(ctr, grp, epoch, gridDim.x);
        acc += sink[(blockIdx.x + i) & 1023];
    }
    if (acc == 123.456) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int iters = 2000;
    unsigned *ctr;
    double *sink;
    hipMalloc(&ctr, 4096 * 4);
    hipMalloc(&sink, 1024 * 8);
    hipMemset(sink, 0, 1024 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int grids[] = {64, 128, 256, 512};
    for (int mode = 0; mode < 5; ++mode)
        for (int gi = 0; gi < 4; ++gi) {
            int g = grids[gi];
            hipMemset(ctr, 0, 4096 * 4);
            unsigned *c = ctr, *grp = ctr + 64;
            int it = iters;
            void *args[] = {&c, &grp, &it, &sink};
            const void *fn = mode == 0 ? (const void *)k<0> : mode == 1 ? (const void *)k<1> : mode == 2 ? (const void *)k<2> : mode == 3 ? (const void *)k<3> : (const void *)k<4>;
            hipEventRecord(e0, 0);
            hipError_t err = hipLaunchCooperativeKernel(fn, dim3(g), dim3(256), args, 0, 0);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d grid %d: %s  %.2f us per barrier\n", mode, g, hipGetErrorString(err), ms * 1e3 / iters);
        }
    return 0;
}
