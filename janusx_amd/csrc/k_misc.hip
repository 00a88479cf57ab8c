// Small dense helpers around the eigendecomposition: ridge, symmetrise, transpose, principal sub-matrix.
// Reference: src/math/eigh.rs:179-207 (symmetrise), python/janusx/assoc/workflow.py:5639-5641 (ridge, subset).
#include "jx_common.h"

namespace jx {

__global__ void add_diag_kernel(double *a, int n, int64_t ld, double ridge) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[(int64_t)i * ld + i] += ridge;
}

// a <- (a + a^T) / 2, in place, one thread per (i > j) pair via 32x32 tiles
__global__ __launch_bounds__(256) void symmetrize_kernel(double *a, int n) {
    const int bx = blockIdx.x, by = blockIdx.y;
    if (bx > by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = by * 32 + r, j = bx * 32 + tx;
        if (i < n && j < n && j < i) {
            const double lo = a[(int64_t)i * n + j], up = a[(int64_t)j * n + i];
            const double v = 0.5 * (lo + up);
            a[(int64_t)i * n + j] = v;
            a[(int64_t)j * n + i] = v;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T *__restrict__ src, T *__restrict__ dst, int n) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int i = blockIdx.y * 32 + r, j = blockIdx.x * 32 + tx;
        if (i < n && j < n) tile[r][tx] = src[(int64_t)i * n + j];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = blockIdx.x * 32 + r, j = blockIdx.y * 32 + tx;
        if (i < n && j < n) dst[(int64_t)i * n + j] = tile[tx][r];
    }
}

// dst (k,k) f64 = src[idx, idx] where src is (n,n) f32 or f64
template <typename T>
__global__ __launch_bounds__(256) void gather_sub_kernel(const T *__restrict__ src, int n, const int32_t *__restrict__ idx,
                                                         int k, double *__restrict__ dst) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)k * k) return;
    const int r = (int)(gid / k), c = (int)(gid - (int64_t)r * k);
    dst[gid] = (double)src[(int64_t)idx[r] * n + idx[c]];
}

// Lower-triangle tiles (ti >= tj, JXG_TILE x JXG_TILE each) of the (npad, npad) f64 GRM accumulator <-> a packed buffer of
// T (T + 1) / 2 tiles: the multi-GPU reduce of the partial GRMs moves only what the GRM kernel writes (SURVEY.md 8e:
// n (n + 1) / 2 values, half of the square).  One workgroup per tile, 16-byte accesses.
template <bool PACK>
__global__ __launch_bounds__(256) void tri_tiles_kernel(double *__restrict__ acc, int npad, double *__restrict__ buf) {
    const int bid = blockIdx.x;
    int ti = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    while ((int64_t)ti * (ti + 1) / 2 > bid) --ti;
    const int tj = bid - (int)((int64_t)ti * (ti + 1) / 2);
    double2 *b2 = reinterpret_cast<double2 *>(buf + (int64_t)bid * JXG_TILE * JXG_TILE);
    for (int e = threadIdx.x; e < JXG_TILE * JXG_TILE / 2; e += 256) {
        const int r = e / (JXG_TILE / 2), c2 = e % (JXG_TILE / 2);
        double2 *a2 = reinterpret_cast<double2 *>(acc + (int64_t)(ti * JXG_TILE + r) * npad + tj * JXG_TILE) + c2;
        if (PACK) b2[e] = *a2;
        else *a2 = b2[e];
    }
}

int launch_add_diag(double *d_a, int n, int64_t ld, double ridge, hipStream_t st) {
    hipLaunchKernelGGL(add_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_a, n, ld, ridge);
    JX_LAUNCH_CHECK();
    return 0;
}

int launch_symmetrize(double *d_a, int n, hipStream_t st) {
    const int nb = (n + 31) / 32;
    hipLaunchKernelGGL(symmetrize_kernel, dim3(nb, nb), dim3(256), 0, st, d_a, n);
    JX_LAUNCH_CHECK();
    return 0;
}

int launch_transpose_f64(const double *src, double *dst, int n, hipStream_t st) {
    const int nb = (n + 31) / 32;
    hipLaunchKernelGGL(transpose_kernel<double>, dim3(nb, nb), dim3(256), 0, st, src, dst, n);
    JX_LAUNCH_CHECK();
    return 0;
}

}  // namespace jx

using namespace jx;

extern "C" int64_t jxg_tri_tiles_doubles(int npad) {
    const int64_t t = npad / JXG_TILE;
    return t * (t + 1) / 2 * JXG_TILE * JXG_TILE;
}

extern "C" int jxg_tri_tiles_pack_f64(double *d_acc, int npad, double *d_buf, int unpack, void *stream) {
    if (npad <= 0 || npad % JXG_TILE != 0) return fail("jxg_tri_tiles_pack_f64: npad must be a positive multiple of the tile size");
    const int64_t t = npad / JXG_TILE;
    const unsigned blocks = (unsigned)(t * (t + 1) / 2);
    if (unpack) hipLaunchKernelGGL(tri_tiles_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_acc, npad, d_buf);
    else hipLaunchKernelGGL(tri_tiles_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_acc, npad, d_buf);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_symmetrize_f64(double *d_a, int n, void *stream) { return launch_symmetrize(d_a, n, (hipStream_t)stream); }

extern "C" int jxg_transpose_f64(const double *d_src, double *d_dst, int n, void *stream) {
    return launch_transpose_f64(d_src, d_dst, n, (hipStream_t)stream);
}

extern "C" int jxg_gather_sub_f64(const void *d_src, int src_is_f64, int n, const int32_t *d_idx, int k, double *d_dst,
                                  void *stream) {
    const int64_t total = (int64_t)k * k;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (src_is_f64)
        hipLaunchKernelGGL(gather_sub_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const double *)d_src, n, d_idx, k, d_dst);
    else
        hipLaunchKernelGGL(gather_sub_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float *)d_src, n, d_idx, k, d_dst);
    JX_LAUNCH_CHECK();
    return 0;
}
