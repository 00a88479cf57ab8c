// Symmetric tridiagonal eigensolver by one Cuppen divide-and-conquer merge on top of two rocSOLVER `dstedc` halves.
// rocSOLVER 7.2's dstedc faults once n^2 exceeds 2^31 (n > 46340; measured at n = 50 000, BASELINE config C4), so the
// eigendecomposition behind src/math/eigh.rs:1422-1528 needs its own top level there: T = diag(T1', T2') + rho u u'
// (LAPACK dlaed0/dlaed1 organisation: deflation as in dlaed2, secular equation in the origin-shifted form of dlaed4,
// eigenvectors from the Gu-Eisenstat / Loewner re-derived z as in dlaed3).  All O(n^2) and O(n^3) work runs on the
// device; the O(n) deflation scan is host code.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <numeric>
#include <thread>
#include <chrono>
#include <vector>

#include "k_ozgemm.h"

namespace jx {

int dgemm(hipStream_t st, bool ta, bool tb, int m, int n, int k, double alpha, const double *a, int64_t lda, const double *b,
          int64_t ldb, double beta, double *c, int64_t ldc, int ksplit, double *ws, size_t ws_doubles);
int ormtr_oz_min_n();

constexpr int SD_THREADS = 256;
constexpr int SD_WAVES = SD_THREADS / 64;

__device__ __forceinline__ double sd_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double sd_wave_prod(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v *= __shfl_xor(v, off, 64);
    return v;
}

// All rank-one tears of the recursion at once: split point k (between rows k-1 and k): d[k-1] -= |e[k-1]|, d[k] -= |e[k-1]|.
// Every diagonal entry belongs to at most one tear on each side, leaves have >= 2 rows, so the updates are disjoint.
__global__ void sd_tear_all_kernel(double *__restrict__ d, const double *__restrict__ e, const int *__restrict__ split,
                                   int nsplit) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsplit) return;
    const int k = split[t];
    const double a = fabs(e[k - 1]);
    d[k - 1] -= a;
    d[k] -= a;
}

// Leaf problems (<= SD_LEAF rows), one workgroup each (`sd_leaf_ql_kernel` below).  d receives the eigenvalues
// (unsorted), zpool + zoff[leaf] the eigenvectors (column-major nl x nl).
constexpr int SD_LEAF = 128;
// lane-distributed vector of up to 128 doubles held by one wave: element i lives in lane i & 63 of x0 (i < 64) or x1
struct LaneVec {
    double x0, x1;
    __device__ __forceinline__ double get(int i) const {   // i is wave-uniform
        const double v = (i < 64) ? x0 : x1;
        const int l = i & 63;
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ void set(int i, double v, int lane) {   // every lane holds the same v
        if (lane == (i & 63)) {
            if (i < 64) x0 = v;
            else x1 = v;
        }
    }
};

// 1 / sqrt(h), h > 0: hardware estimate + two Newton steps (full double precision)
__device__ __forceinline__ double sd_rsqrt(double h) {
    double y = __builtin_amdgcn_rsq(h);
    double t = h * y;
    double u = fma(-t, y, 1.0);
    y = fma(0.5 * y, u, y);
    t = h * y;
    u = fma(-t, y, 1.0);
    y = fma(0.5 * y, u, y);
    return y;
}

// One workgroup per leaf (<= SD_LEAF = 128 rows): implicit-shift QL (LAPACK dsteqr's QL branch) with the eigenvector
// block resident in LDS.  Wave 0 runs the scalar recurrence of a sweep: d and e live in its registers for the whole
// solve (lane-distributed, `LaneVec`), every lane executes the same recurrence and the lane that owns an element
// captures its new value, so the chain never waits on LDS; plane rotations use 1/sqrt(f^2 + g^2) (rsq + Newton)
// instead of a square root followed by two divisions.  The rotations of the sweep go to LDS, then every thread
// applies the sweep to its own row of Z.
__global__ __launch_bounds__(256) void sd_leaf_ql_kernel(double *__restrict__ dg, const double *__restrict__ eg,
                                                         const int *__restrict__ off, const int *__restrict__ len,
                                                         const int64_t *__restrict__ zoff, double *__restrict__ zpool,
                                                         int *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int leaf = blockIdx.x;
    const int nl = len[leaf];
    const int g0 = off[leaf];
    double *z = lds;                       // z[i * nl + k] = Z(k, i)
    double *cs = lds + (size_t)nl * nl;    // (c_i, s_i) pairs
    __shared__ int sh_m, sh_lo;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    for (int t = tid; t < nl * nl; t += 256) z[t] = 0.0;
    __syncthreads();
    if (tid < nl) z[tid * nl + tid] = 1.0;
    LaneVec d, e, cc, ss;
    d.x0 = d.x1 = e.x0 = e.x1 = 0.0;
    cc.x0 = cc.x1 = ss.x0 = ss.x1 = 0.0;
    if (tid < 64) {
        if (lane < nl) d.x0 = dg[g0 + lane];
        if (lane + 64 < nl) d.x1 = dg[g0 + lane + 64];
        if (lane < nl - 1) e.x0 = eg[g0 + lane];
        if (lane + 64 < nl - 1) e.x1 = eg[g0 + lane + 64];
    }
    __syncthreads();
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < nl; ++l) {
        for (int iter = 0;; ++iter) {
            if (tid < 64) {
                // first m >= l with a negligible e[m] (m = nl - 1 when there is none)
                int m;
                {
                    // d[k + 1] for this lane's two elements
                    double dn0 = __shfl_down(d.x0, 1, 64), dn1 = __shfl_down(d.x1, 1, 64);
                    const double d1first = d.get(64 < nl ? 64 : 0);
                    if (lane == 63) dn0 = d1first;
                    const int k0 = lane, k1 = lane + 64;
                    const bool t0 = k0 >= l && k0 < nl - 1 && fabs(e.x0) <= eps * (fabs(d.x0) + fabs(dn0));
                    const bool t1 = k1 >= l && k1 < nl - 1 && fabs(e.x1) <= eps * (fabs(d.x1) + fabs(dn1));
                    const unsigned long long b0 = __ballot(t0), b1 = __ballot(t1);
                    if (b0) m = __builtin_ctzll(b0);
                    else if (b1) m = 64 + __builtin_ctzll(b1);
                    else m = nl - 1;
                }
                int lo = m;
                if (m != l) {
                    if (iter >= 60) {
                        if (lane == 0) *err = 1;
                        m = l;   // give up on this eigenvalue
                        lo = l;
                    } else {
                        const double dl = d.get(l), el = e.get(l);
                        double g = (d.get(l + 1) - dl) / (2.0 * el);
                        double r = sqrt(g * g + 1.0);
                        g = d.get(m) - dl + el / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
                        double sn = 1.0, c = 1.0, p = 0.0;
                        int i = m - 1;
                        bool underflow = false;
                        for (; i >= l; --i) {
                            const double ei = e.get(i), di = d.get(i), di1 = d.get(i + 1);
                            const double f = sn * ei;
                            const double b = c * ei;
                            const double h = f * f + g * g;
                            if (h == 0.0) {
                                e.set(i + 1, 0.0, lane);
                                d.set(i + 1, di1 - p, lane);
                                e.set(m, 0.0, lane);
                                underflow = true;
                                break;
                            }
                            const double rinv = sd_rsqrt(h);
                            e.set(i + 1, h * rinv, lane);
                            sn = f * rinv;
                            c = g * rinv;
                            g = di1 - p;
                            r = (di - g) * sn + 2.0 * c * b;
                            p = sn * r;
                            d.set(i + 1, g + p, lane);
                            g = c * r - b;
                            cc.set(i, c, lane);
                            ss.set(i, sn, lane);
                        }
                        lo = i + 1;
                        if (!underflow) {
                            d.set(l, d.get(l) - p, lane);
                            e.set(l, g, lane);
                            e.set(m, 0.0, lane);
                        }
                        // rotations [lo, m-1] of this sweep -> LDS
                        if (lane >= lo && lane < m) {
                            cs[2 * lane] = cc.x0;
                            cs[2 * lane + 1] = ss.x0;
                        }
                        if (lane + 64 >= lo && lane + 64 < m) {
                            cs[2 * (lane + 64)] = cc.x1;
                            cs[2 * (lane + 64) + 1] = ss.x1;
                        }
                    }
                }
                if (lane == 0) {
                    sh_m = m;
                    sh_lo = lo;
                }
            }
            __syncthreads();
            const int m = sh_m, lo = sh_lo;
            if (m == l) break;
            if (tid < nl && lo < m) {
                // the value carried from rotation i + 1 to rotation i stays in a register (no LDS write -> read chain)
                double f = z[m * nl + tid];
                for (int i = m - 1; i >= lo; --i) {
                    const double c = cs[2 * i], sn = cs[2 * i + 1];
                    const double zi = z[i * nl + tid];
                    z[(i + 1) * nl + tid] = sn * zi + c * f;
                    f = c * zi - sn * f;
                }
                z[lo * nl + tid] = f;
            }
            __syncthreads();
        }
    }
    if (tid < 64) {
        if (lane < nl) dg[g0 + lane] = d.x0;
        if (lane + 64 < nl) dg[g0 + lane + 64] = d.x1;
    }
    double *zo = zpool + zoff[leaf];
    for (int t = tid; t < nl * nl; t += 256) zo[t] = z[t];
}

__global__ void sd_tear_kernel(double *d2, const double *amt) {
    if (threadIdx.x < 2) d2[threadIdx.x] -= amt[threadIdx.x];
}

// z = [last row of Q1 ; sgn * first row of Q2] / sqrt(2)   (dlaed1: the rank-one vector in the eigenbasis)
__global__ void sd_extract_z_kernel(const double *__restrict__ q1, int k1, const double *__restrict__ q2, int k2,
                                    double sgn, double *__restrict__ z) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const double r = 0.70710678118654752440;
    if (j < k1) z[j] = q1[(int64_t)j * k1 + (k1 - 1)] * r;
    else if (j < k1 + k2) z[j] = sgn * q2[(int64_t)(j - k1) * k2] * r;
}

// Secular equation 1 + rho sum_j z_j^2 / (d_j - lambda) = 0, root i in (d_i, d_{i+1}) (the last one in
// (d_{K-1}, d_{K-1} + rho |z|^2)).  One wave per root.  The root is kept as (origin pole, offset tau) so that every
// d_j - lambda = (d_j - d_org) - tau is formed without cancellation (dlaed4); iteration = the two-pole rational
// interpolation of dlaed4 (poles i and i+1) safeguarded by a bracket.
__global__ __launch_bounds__(SD_THREADS) void sd_secular_kernel(const double *__restrict__ d,
                                                                const double *__restrict__ z, int K, double rho,
                                                                double *__restrict__ tau_out, int *__restrict__ org_out,
                                                                double *__restrict__ lam_out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * SD_WAVES + (threadIdx.x >> 6);
    if (i >= K) return;
    const double eps = 2.220446049250313e-16;
    const bool last = (i == K - 1);
    const double di = d[i];
    double znorm2 = 0.0;
    if (last) {
        for (int j = lane; j < K; j += 64) znorm2 += z[j] * z[j];
        znorm2 = sd_wave_sum(znorm2);
    }
    const double dip1 = last ? di + rho * znorm2 : d[i + 1];
    const double delta = dip1 - di;
    // which half of the interval holds the root: sign of f at the midpoint, evaluated around pole i
    int org = i;
    {
        const double half = 0.5 * delta;
        double acc = 0.0;
        for (int j = lane; j < K; j += 64) acc += z[j] * z[j] / ((d[j] - di) - half);
        const double fmid = 1.0 + rho * sd_wave_sum(acc);
        if (!last && fmid < 0.0) org = i + 1;
    }
    const double dorg = d[org];
    const double A = di - dorg;                    // 0 or -delta
    const double B = dip1 - dorg;                  // +delta or 0 (for the last root: the virtual upper end)
    double lo = (org == i) ? 0.0 : -0.5 * delta;
    double hi = (org == i) ? (last ? delta : 0.5 * delta) : 0.0;
    // starting point: the two nearest poles exactly, the rest frozen at the midpoint (dlaed4's initial guess)
    double tau;
    {
        const double mid = 0.5 * (lo + hi);
        double rest = 0.0;
        for (int j = lane; j < K; j += 64)
            if (j != i && (last || j != i + 1)) rest += z[j] * z[j] / ((d[j] - dorg) - mid);
        const double c = 1.0 + rho * sd_wave_sum(rest);
        const double a2 = rho * z[i] * z[i];
        const double b2 = last ? 0.0 : rho * z[i + 1] * z[i + 1];
        // c (A - t)(B - t) + a2 (B - t) + b2 (A - t) = 0
        const double qa = c, qb = -(c * (A + B) + a2 + b2), qc = c * A * B + a2 * B + b2 * A;
        tau = mid;
        if (last) {
            // one pole: c + a2 / (A - t) = 0
            if (c > 0.0) {
                const double t = A + a2 / c;
                if (t > lo && t < hi) tau = t;
            }
        } else if (qa != 0.0) {
            const double disc = qb * qb - 4.0 * qa * qc;
            if (disc >= 0.0) {
                const double sq = sqrt(disc);
                const double qq = -0.5 * (qb + (qb >= 0.0 ? sq : -sq));
                const double t1 = qq / qa, t2 = (qq != 0.0) ? qc / qq : t1;
                if (t1 > lo && t1 < hi) tau = t1;
                else if (t2 > lo && t2 < hi) tau = t2;
            }
        } else if (qb != 0.0) {
            const double t = -qc / qb;
            if (t > lo && t < hi) tau = t;
        }
    }
    for (int it = 0; it < 80; ++it) {
        double psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0;
        for (int j = lane; j < K; j += 64) {
            const double dl = (d[j] - dorg) - tau;
            const double t = z[j] / dl;
            const double zt = z[j] * t;   // z^2 / (d_j - lambda)
            const double t2 = t * t;      // z^2 / (d_j - lambda)^2
            if (j <= i) {
                psi += zt;
                dpsi += t2;
            } else {
                phi += zt;
                dphi += t2;
            }
        }
        psi = rho * sd_wave_sum(psi);
        dpsi = rho * sd_wave_sum(dpsi);
        phi = rho * sd_wave_sum(phi);
        dphi = rho * sd_wave_sum(dphi);
        const double w = 1.0 + psi + phi;
        const double dw = dpsi + dphi;
        const double erretm = 8.0 * (phi - psi) + 2.0 + fabs(tau) * dw;
        if (fabs(w) <= eps * erretm) break;
        if (w > 0.0) hi = tau; else lo = tau;      // f is increasing between two poles
        if (hi - lo <= 2.0 * eps * fmax(fabs(lo), fabs(hi))) break;
        const double D1 = A - tau;                  // d_i - lambda (< 0)
        double eta;
        if (last) {
            // single-pole model of psi around d_i: c + p / (D1 - eta) = 0
            const double c = w - dpsi * D1, p = dpsi * D1 * D1;
            eta = (c > 0.0) ? (D1 + p / c) : -w / dw;
        } else {
            const double D2 = B - tau;              // d_{i+1} - lambda (> 0)
            const double c = w - dpsi * D1 - dphi * D2;
            const double a = (D1 + D2) * w - D1 * D2 * dw;
            const double b = D1 * D2 * w;
            if (c == 0.0) {
                eta = (a != 0.0) ? b / a : -w / dw;
            } else {
                const double disc = sqrt(fabs(a * a - 4.0 * b * c));
                eta = (a <= 0.0) ? (a - disc) / (2.0 * c) : 2.0 * b / (a + disc);
            }
        }
        if (!isfinite(eta) || w * eta > 0.0) eta = -w / dw;
        double tn = tau + eta;
        if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
        if (tn == tau) break;
        tau = tn;
    }
    if (lane == 0) {
        tau_out[i] = tau;
        org_out[i] = org;
        lam_out[i] = dorg + tau;
    }
}

// zhat_j = sign(z_j) sqrt( prod_i (lambda_i - d_j) / prod_{i != j} (d_i - d_j) ) / sqrt(rho)  (dlaed3: the z for which the
// computed lambdas are the exact roots, which is what makes the eigenvectors orthogonal).  One wave per j.
__global__ __launch_bounds__(SD_THREADS) void sd_lowner_kernel(const double *__restrict__ d, const double *__restrict__ z,
                                                               int K, double rho, const double *__restrict__ tau,
                                                               const int *__restrict__ org, double *__restrict__ zhat) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * SD_WAVES + (threadIdx.x >> 6);
    if (j >= K) return;
    const double dj = d[j];
    double prod = 1.0;
    for (int i = lane; i < K; i += 64) {
        const double num = (d[org[i]] - dj) + tau[i];      // lambda_i - d_j, cancellation-free
        if (i == j) prod *= num;
        else prod *= num / (d[i] - dj);
    }
    prod = sd_wave_prod(prod);
    if (lane == 0) {
        const double v = sqrt(fabs(prod) / rho);
        zhat[j] = (z[j] < 0.0) ? -v : v;
    }
}

// 1 / || ( zhat_j / (d_j - lambda_i) )_j ||  per root i.  One wave per i.
__global__ __launch_bounds__(SD_THREADS) void sd_colnorm_kernel(const double *__restrict__ d,
                                                                const double *__restrict__ zhat, int K,
                                                                const double *__restrict__ tau,
                                                                const int *__restrict__ org, double *__restrict__ invn) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * SD_WAVES + (threadIdx.x >> 6);
    if (i >= K) return;
    const double dorg = d[org[i]], t = tau[i];
    double acc = 0.0;
    for (int j = lane; j < K; j += 64) {
        const double v = zhat[j] / ((d[j] - dorg) - t);
        acc += v * v;
    }
    acc = sd_wave_sum(acc);
    if (lane == 0) invn[i] = 1.0 / sqrt(acc);
}

// U[r, i] = zhat_j / (d_j - lambda_i) * invn_i  for the kept entries j = rowmap[r]  (column-major, ld = ldu)
__global__ __launch_bounds__(SD_THREADS) void sd_form_u_kernel(const double *__restrict__ d,
                                                               const double *__restrict__ zhat,
                                                               const double *__restrict__ tau,
                                                               const int *__restrict__ org,
                                                               const double *__restrict__ invn,
                                                               const int *__restrict__ rowmap, int nrows, int K,
                                                               double *__restrict__ u, int64_t ldu,
                                                               const int *__restrict__ colsel) {
    // colsel (optional): output column y = root colsel[y] (K = number of selected roots)
    const int r = blockIdx.x * SD_THREADS + threadIdx.x;
    const int y = blockIdx.y;
    if (r >= nrows || y >= K) return;
    const int i = colsel ? colsel[y] : y;
    const int j = rowmap[r];
    u[(int64_t)y * ldu + r] = zhat[j] / ((d[j] - d[org[i]]) - tau[i]) * invn[i];
}

// columns x <- c x + s y ; y <- -s x + c y   (drot on two columns of length n)
__global__ void sd_rot_kernel(double *__restrict__ x, double *__restrict__ y, int n, double c, double s) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        const double a = x[r], b = y[r];
        x[r] = c * a + s * b;
        y[r] = c * b - s * a;
    }
}

// dst[:, c] (n rows, ld ldd) = [zeros(off) ; src column ; zeros]  for a list of column pairs
__global__ void sd_place_cols_kernel(const double *__restrict__ src, int64_t lds, int rows, const int *__restrict__ scol,
                                     double *__restrict__ dst, int64_t ldd, int n, int row_off,
                                     const int *__restrict__ dcol, int ncols) {
    const int c = blockIdx.y;
    if (c >= ncols) return;
    const double *s = src + (int64_t)scol[c] * lds;
    double *t = dst + (int64_t)dcol[c] * ldd;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const int rr = r - row_off;
        t[r] = (rr >= 0 && rr < rows) ? s[rr] : 0.0;
    }
}

namespace {
// Stream-ordered bump allocator over one device block: every temporary of the recursion is carved from it, so a
// merge performs no hipMalloc / hipFree (each hipFree is a device-wide synchronisation).  Reuse after a reset is
// safe because all work of one eigendecomposition is ordered on a single stream.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
    void *take(size_t bytes) {
        const size_t a = (off + 255) & ~(size_t)255;
        if (a + bytes > cap) return nullptr;
        off = a + bytes;
        return base + a;
    }
};
struct ABuf {
    void *p = nullptr;
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};
struct DcCtx {
    rocblas_handle h;
    hipStream_t st;
    int leaf;
    int depth = 0;      // recursion depth of the current call
    int par_depth = 0;  // the two halves of a problem run concurrently (own host thread, stream, handle) above it
    int device = 0;
    // Column window of the ROOT problem (several ranks: a rank back-transforms the eigenvectors [sel_lo, sel_hi) only): the
    // top-level merge -- three quarters of the divide and conquer's product flops -- then forms and multiplies only the
    // columns whose eigenvalues rank inside the window; sel_cols[t] = column of d_c that holds the (sel_lo + t)-th smallest
    // eigenvalue on return.  sel_hi <= sel_lo: all columns (sel_cols stays empty).
    int sel_lo = 0, sel_hi = 0;
    std::vector<int> sel_cols;
    // own leaf solver (sd_leaf_ql_kernel): every leaf of the tree is solved by one batched launch before the recursion
    bool own_leaf = false;
    const std::vector<int> *leaf_off = nullptr;     // ascending global offsets of the leaves
    const std::vector<int64_t> *leaf_zoff = nullptr;
    const double *zpool = nullptr;                  // device
    const double *hleaf_w = nullptr;                // host copy of the leaf eigenvalues (whole problem)
    Arena ar;
    const double *hd0, *he0;              // host copy of the tridiagonal (whole problem)
    double *d_base, *e_base;              // device d / e of the whole problem (offsets recover the host index)
    std::vector<std::vector<char>> keep;  // host staging blocks that must outlive their asynchronous copies
    template <class T> const T *stage(const std::vector<T> &v) {
        keep.emplace_back((const char *)v.data(), (const char *)v.data() + sizeof(T) * v.size());
        return reinterpret_cast<const T *>(keep.back().data());
    }
};
// (stream, rocBLAS handle) pairs for sub-problems that run concurrently with their sibling
struct Lane {
    hipStream_t st = nullptr;
    rocblas_handle h = nullptr;
};
static std::mutex g_lane_mu;
static std::vector<Lane> g_free_lanes;
static int lane_acquire(Lane &l) {
    {
        std::lock_guard<std::mutex> lk(g_lane_mu);
        if (!g_free_lanes.empty()) {
            l = g_free_lanes.back();
            g_free_lanes.pop_back();
            return 0;
        }
    }
    JX_HIP(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
    if (rocblas_create_handle(&l.h) != rocblas_status_success) return fail("rocblas_create_handle failed");
    rocblas_set_atomics_mode(l.h, rocblas_atomics_not_allowed);   // bit-identical merges on every rank (eigh.cpp)
    if (rocblas_set_stream(l.h, l.st) != rocblas_status_success) return fail("rocblas_set_stream failed");
    return 0;
}
static void lane_release(const Lane &l) {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    g_free_lanes.push_back(l);
}
// merge products on the int8 pipes: from JXGPU_OZ_MIN_N / 4 rows, columns and inner dimension on (JXGPU_STEDC_OZ=0: never)
static bool sd_use_oz(int qrows, int cols, int inner) {
    static const bool on = !(getenv("JXGPU_STEDC_OZ") && atoi(getenv("JXGPU_STEDC_OZ")) == 0);
    const int lim = std::max(256, ormtr_oz_min_n() / 4);
    return on && qrows >= lim && cols >= lim && inner >= lim;
}
static size_t dc_arena_bytes(int n) {
    const size_t nn = (size_t)n * (size_t)n;
    // q1/q2 (n^2/2), their compacted copies (<= n^2/2) and one U factor at a time (<= n^2/2), or q1/q2 plus the two
    // children's arenas (n^2 + ...): 2 n^2 covers both for n >= 128; vectors, index lists and the 256-byte alignment
    // of every block are O(n) per level
    // + the int8 images of one merge product (k_ozgemm.hip): planes (n / 2 + n) (n / 2) bytes and their row scales
    return sizeof(double) * (2 * nn + 256 * (size_t)n) + (64u << 10) + (size_t)6 * (3 * nn / 4 + 512 * (size_t)n) +
           (64u << 10);
}

static std::chrono::steady_clock::time_point g_dc_t0;
static bool g_dc_trace = false;
// JXGPU_EIGH_TRACE=1: time line of the merges of the top three levels (synchronises the stream at every mark)
#define DC_MARK(what)                                                                                              \
    do {                                                                                                           \
        if (g_dc_trace && C.depth <= 2) {                                                                          \
            (void)hipStreamSynchronize(st);                                                                        \
            fprintf(stderr, "[jxgpu stedc] depth %d n %5d %-14s at %7.2f ms\n", C.depth, n, what,                   \
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_dc_t0).count()); \
        }                                                                                                          \
    } while (0)

#define SD_TAKE(buf, bytes)                                                        \
    do {                                                                           \
        (buf).p = C.ar.take(bytes);                                                \
        if (!(buf).p) return fail("stedc: workspace arena exhausted");             \
    } while (0)

struct Loc {      // where the current image of a basis column lives
    int kind;     // 0: column of Q1 (rows [0,k1)), 1: column of Q2 (rows [k1,n)), 2: dense n-vector in the side buffer
    int idx;
};
}  // namespace

// Recursive worker.  d_d (n), d_e (n-1): tridiagonal T on the device (both overwritten).  d_c: (n,n) column-major,
// receives the eigenvectors of T as columns; h_w[c] = eigenvalue of column c (not sorted above the leaves).
// Problems of at most C.leaf rows go to rocSOLVER.
static int stedc_dc(DcCtx &C, int n, double *d_d, double *d_e, double *d_c, std::vector<double> &h_w) {
    rocblas_handle h = C.h;
    hipStream_t st = C.st;
    const size_t mark = C.ar.off;
    if (C.own_leaf && (n <= C.leaf || n < 4)) {
        const int g0 = (int)(d_d - C.d_base);
        const auto it = std::lower_bound(C.leaf_off->begin(), C.leaf_off->end(), g0);
        if (it == C.leaf_off->end() || *it != g0) return fail("stedc: leaf plan mismatch");
        const int64_t zo = (*C.leaf_zoff)[it - C.leaf_off->begin()];
        JX_HIP(hipMemcpyAsync(d_c, C.zpool + zo, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, st));
        h_w.assign(C.hleaf_w + g0, C.hleaf_w + g0 + n);
        return 0;
    }
    if (n <= C.leaf || n < 4) {
        ABuf linfo;
        SD_TAKE(linfo, sizeof(rocblas_int));
        JX_HIP(hipMemsetAsync(linfo.p, 0, sizeof(rocblas_int), st));
        rocblas_status ls = rocsolver_dstedc(h, rocblas_evect_tridiagonal, n, d_d, d_e, d_c, n, linfo.as<rocblas_int>());
        if (ls != rocblas_status_success) return fail("rocsolver_dstedc (leaf) failed: " + std::to_string((int)ls));
        rocblas_int li = 0;
        h_w.resize((size_t)n);
        JX_HIP(hipMemcpyAsync(&li, linfo.p, sizeof(li), hipMemcpyDeviceToHost, st));
        JX_HIP(hipMemcpyAsync(h_w.data(), d_d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        JX_HIP(hipStreamSynchronize(st));
        C.ar.off = mark;
        if (li != 0) return fail("rocsolver_dstedc did not converge on a leaf problem");
        return 0;
    }
    const int k1 = n / 2, k2 = n - k1;
    const size_t g0 = (size_t)(d_d - C.d_base);          // position of this block in the whole problem
    const double rho0 = C.he0[g0 + k1 - 1];
    const double absrho = fabs(rho0), sgn = (rho0 < 0.0) ? -1.0 : 1.0;
    if (!C.own_leaf) {
        // rank-one tear: d[k1-1] -= |rho|, d[k1] -= |rho| (dlaed0).  Each diagonal entry is torn at most once per
        // level, and by exactly one ancestor chain, so the host copy needs no update: apply on the device values.
        std::vector<double> two = {absrho, absrho};
        const double *hp = C.stage(two);
        ABuf t2;
        SD_TAKE(t2, 2 * sizeof(double));
        JX_HIP(hipMemcpyAsync(t2.p, hp, 2 * sizeof(double), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(sd_tear_kernel, dim3(1), dim3(64), 0, st, d_d + k1 - 1, t2.as<double>());
        JX_LAUNCH_CHECK();
    }

    ABuf q1, q2;
    SD_TAKE(q1, sizeof(double) * (size_t)k1 * k1);
    SD_TAKE(q2, sizeof(double) * (size_t)k2 * k2);
    std::vector<double> D((size_t)n), z((size_t)n);
    {
        std::vector<double> w1, w2;
        const int child_depth = C.depth + 1;
        static const int par_min_n = getenv("JXGPU_STEDC_PARMIN") ? atoi(getenv("JXGPU_STEDC_PARMIN")) : 1024;
        if (C.depth < C.par_depth && n >= par_min_n) {
            // the halves are independent: the second one gets its own host thread, stream, rocBLAS handle and arena
            // slice (rocSOLVER's leaf solver is a chain of latency-bound launches, two of them overlap almost fully)
            const size_t s1 = dc_arena_bytes(k1), s2 = dc_arena_bytes(k2);
            ABuf a1, a2;
            SD_TAKE(a1, s1);
            SD_TAKE(a2, s2);
            hipEvent_t ready;
            JX_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
            JX_HIP(hipEventRecord(ready, st));     // the tear above, and every earlier user of the recycled arena
            Lane lane;
            if (lane_acquire(lane)) return 1;
            int rc2 = 0;
            DcCtx C2;
            C2.h = lane.h;
            C2.st = lane.st;
            C2.leaf = C.leaf;
            C2.depth = child_depth;
            C2.par_depth = C.par_depth;
            C2.device = C.device;
            C2.ar.base = (char *)a2.p;
            C2.ar.cap = s2;
            C2.hd0 = C.hd0;
            C2.he0 = C.he0;
            C2.d_base = C.d_base;
            C2.e_base = C.e_base;
            C2.own_leaf = C.own_leaf;
            C2.leaf_off = C.leaf_off;
            C2.leaf_zoff = C.leaf_zoff;
            C2.zpool = C.zpool;
            C2.hleaf_w = C.hleaf_w;
            const int planes_ov = oz_planes_override_get();
            std::thread th([&]() {
                oz_planes_override_set(planes_ov);     // the override is per thread: the child merges with its parent's planes
                if (hipSetDevice(C2.device) != hipSuccess || hipStreamWaitEvent(C2.st, ready, 0) != hipSuccess) {
                    rc2 = 1;
                    return;
                }
                rc2 = stedc_dc(C2, k2, d_d + k1, d_e + k1, q2.as<double>(), w2);
                if (hipStreamSynchronize(C2.st) != hipSuccess) rc2 = 1;
            });
            DcCtx C1;
            C1.h = C.h;
            C1.st = C.st;
            C1.leaf = C.leaf;
            C1.depth = child_depth;
            C1.par_depth = C.par_depth;
            C1.device = C.device;
            C1.ar.base = (char *)a1.p;
            C1.ar.cap = s1;
            C1.hd0 = C.hd0;
            C1.he0 = C.he0;
            C1.d_base = C.d_base;
            C1.e_base = C.e_base;
            C1.own_leaf = C.own_leaf;
            C1.leaf_off = C.leaf_off;
            C1.leaf_zoff = C.leaf_zoff;
            C1.zpool = C.zpool;
            C1.hleaf_w = C.hleaf_w;
            const int rc1 = stedc_dc(C1, k1, d_d, d_e, q1.as<double>(), w1);
            th.join();
            (void)hipEventDestroy(ready);
            lane_release(lane);
            if (rc1) return 1;
            if (rc2) return fail("stedc: the concurrent half-problem failed");
            JX_HIP(hipStreamSynchronize(st));   // C1's staging blocks die with it
            C.ar.off = (size_t)((char *)a1.p - C.ar.base);   // both slices are free again
        } else {
            const int saved = C.depth;
            C.depth = child_depth;
            const int r1 = stedc_dc(C, k1, d_d, d_e, q1.as<double>(), w1);
            const int r2 = r1 ? 1 : stedc_dc(C, k2, d_d + k1, d_e + k1, q2.as<double>(), w2);
            C.depth = saved;
            if (r1 || r2) return 1;
        }
        std::copy(w1.begin(), w1.end(), D.begin());
        std::copy(w2.begin(), w2.end(), D.begin() + k1);
    }
    DC_MARK("children done");
    ABuf dz;
    SD_TAKE(dz, sizeof(double) * (size_t)n);
    hipLaunchKernelGGL(sd_extract_z_kernel, dim3((n + 255) / 256), dim3(256), 0, st, q1.as<double>(), k1,
                       q2.as<double>(), k2, sgn, dz.as<double>());
    JX_LAUNCH_CHECK();
    JX_HIP(hipMemcpyAsync(z.data(), dz.p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    const double rho = 2.0 * absrho;
    DC_MARK("z on host");

    // ---- deflation (dlaed2) ---------------------------------------------------------------------------
    std::vector<int> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return D[a] < D[b] || (D[a] == D[b] && a < b); });
    double dmax = 0.0, zmax = 0.0;
    for (int i = 0; i < n; ++i) {
        dmax = std::max(dmax, fabs(D[i]));
        zmax = std::max(zmax, fabs(z[i]));
    }
    const double eps = 2.220446049250313e-16;
    const double tol = 8.0 * eps * std::max(dmax, zmax);
    std::vector<Loc> loc((size_t)n);
    for (int i = 0; i < n; ++i) loc[i] = (i < k1) ? Loc{0, i} : Loc{1, i - k1};
    DevBuf side;          // dense n-vectors of columns that took part in a cross-block rotation
    int nside = 0, side_cap = 0;
    auto col_ptr = [&](const Loc &l) -> double * {
        if (l.kind == 0) return q1.as<double>() + (size_t)l.idx * k1;
        if (l.kind == 1) return q2.as<double>() + (size_t)l.idx * k2;
        return side.as<double>() + (size_t)l.idx * n;
    };
    auto promote = [&](int col) -> int {   // give `col` a dense image in the side buffer
        if (loc[col].kind == 2) return 0;
        if (nside == side_cap) {
            const int ncap = side_cap ? 2 * side_cap : 16;
            DevBuf nb;
            if (nb.alloc(sizeof(double) * (size_t)ncap * n)) return 1;
            if (nside) JX_HIP(hipMemcpyAsync(nb.p, side.p, sizeof(double) * (size_t)nside * n, hipMemcpyDeviceToDevice, st));
            JX_HIP(hipStreamSynchronize(st));
            std::swap(side.p, nb.p);
            std::swap(side.bytes, nb.bytes);
            side_cap = ncap;
        }
        double *dst = side.as<double>() + (size_t)nside * n;
        JX_HIP(hipMemsetAsync(dst, 0, sizeof(double) * (size_t)n, st));
        const Loc l = loc[col];
        if (l.kind == 0) JX_HIP(hipMemcpyAsync(dst, col_ptr(l), sizeof(double) * k1, hipMemcpyDeviceToDevice, st));
        else JX_HIP(hipMemcpyAsync(dst + k1, col_ptr(l), sizeof(double) * k2, hipMemcpyDeviceToDevice, st));
        loc[col] = Loc{2, nside++};
        return 0;
    };
    std::vector<int> kept, defl;
    kept.reserve(n);
    if (rho * zmax <= tol) {
        for (int t = 0; t < n; ++t) defl.push_back(order[t]);   // the halves do not interact
    } else {
        int pj = -1;
        for (int t = 0; t < n; ++t) {
            const int nj = order[t];
            if (rho * fabs(z[nj]) <= tol) {
                defl.push_back(nj);
                continue;
            }
            if (pj < 0) {
                pj = nj;
                continue;
            }
            double s = z[pj], c = z[nj];
            const double tt = hypot(c, s);
            const double gap = D[nj] - D[pj];
            c /= tt;
            s = -s / tt;
            if (fabs(gap * c * s) <= tol) {
                // rotate the pair so that all of z sits in nj; pj becomes an eigenvector of its own
                z[nj] = tt;
                z[pj] = 0.0;
                if (loc[pj].kind != loc[nj].kind || loc[pj].kind == 2) {
                    if (promote(pj) || promote(nj)) return 1;
                }
                const int len = (loc[pj].kind == 0) ? k1 : (loc[pj].kind == 1 ? k2 : n);
                hipLaunchKernelGGL(sd_rot_kernel, dim3((len + 255) / 256), dim3(256), 0, st, col_ptr(loc[pj]),
                                   col_ptr(loc[nj]), len, c, s);
                JX_LAUNCH_CHECK();
                const double t1 = D[pj] * c * c + D[nj] * s * s;
                D[nj] = D[pj] * s * s + D[nj] * c * c;
                D[pj] = t1;
                defl.push_back(pj);
                pj = nj;
            } else {
                kept.push_back(pj);
                pj = nj;
            }
        }
        if (pj >= 0) kept.push_back(pj);
    }
    const int K = (int)kept.size();

    // ---- output column assignment: roots 0..K-1, then the deflated columns ---------------------------------
    std::vector<double> w((size_t)n);
    std::vector<double> lam((size_t)std::max(K, 1));
    ABuf dk, zk, dtau, dorg, dlam, dzh, dinv;
    if (K > 0) {
        std::vector<double> hk((size_t)K), hz((size_t)K);
        for (int r = 0; r < K; ++r) {
            hk[r] = D[kept[r]];
            hz[r] = z[kept[r]];
        }
        SD_TAKE(dk, sizeof(double) * K);
        SD_TAKE(zk, sizeof(double) * K);
        SD_TAKE(dtau, sizeof(double) * K);
        SD_TAKE(dorg, sizeof(int) * K);
        SD_TAKE(dlam, sizeof(double) * K);
        SD_TAKE(dzh, sizeof(double) * K);
        SD_TAKE(dinv, sizeof(double) * K);
        JX_HIP(hipMemcpyAsync(dk.p, C.stage(hk), sizeof(double) * K, hipMemcpyHostToDevice, st));
        JX_HIP(hipMemcpyAsync(zk.p, C.stage(hz), sizeof(double) * K, hipMemcpyHostToDevice, st));
        DC_MARK("deflated");
        const int gw = (K + SD_WAVES - 1) / SD_WAVES;
        hipLaunchKernelGGL(sd_secular_kernel, dim3(gw), dim3(SD_THREADS), 0, st, dk.as<double>(), zk.as<double>(), K, rho,
                           dtau.as<double>(), dorg.as<int>(), dlam.as<double>());
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sd_lowner_kernel, dim3(gw), dim3(SD_THREADS), 0, st, dk.as<double>(), zk.as<double>(), K, rho,
                           dtau.as<double>(), dorg.as<int>(), dzh.as<double>());
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sd_colnorm_kernel, dim3(gw), dim3(SD_THREADS), 0, st, dk.as<double>(), dzh.as<double>(), K,
                           dtau.as<double>(), dorg.as<int>(), dinv.as<double>());
        JX_LAUNCH_CHECK();
        JX_HIP(hipMemcpyAsync(lam.data(), dlam.p, sizeof(double) * K, hipMemcpyDeviceToHost, st));
        JX_HIP(hipStreamSynchronize(st));   // lam is needed on the host
    }
    for (int r = 0; r < K; ++r) w[r] = lam[r];
    for (size_t r = 0; r < defl.size(); ++r) w[K + r] = D[defl[r]];

    // ---- column window of the root problem (DcCtx::sel_lo / sel_hi): only the columns whose eigenvalues rank inside it ------
    const bool windowed = C.depth == 0 && C.sel_hi > C.sel_lo;
    int Kout = K;                                   // root columns this call forms (all of them without a window)
    std::vector<int> root_sel;                      // selected roots, ascending (windowed)
    std::vector<int> defl_dst(defl.size());         // destination column of a deflated vector, -1 = not wanted
    for (size_t r = 0; r < defl.size(); ++r) defl_dst[r] = K + (int)r;
    ABuf dcolsel;
    if (windowed) {
        std::vector<int> order((size_t)n);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return w[a] < w[b] || (w[a] == w[b] && a < b); });
        std::vector<int> pos((size_t)n, -1);        // full-layout column -> compact column
        std::vector<char> want((size_t)n, 0);
        for (int t = C.sel_lo; t < C.sel_hi; ++t) want[order[t]] = 1;
        int at = 0;
        for (int c = 0; c < K; ++c)
            if (want[c]) {
                root_sel.push_back(c);
                pos[c] = at++;
            }
        Kout = at;
        for (size_t r = 0; r < defl.size(); ++r) {
            defl_dst[r] = want[K + r] ? at : -1;
            if (want[K + r]) pos[K + r] = at++;
        }
        C.sel_cols.resize((size_t)(C.sel_hi - C.sel_lo));
        for (int t = C.sel_lo; t < C.sel_hi; ++t) C.sel_cols[t - C.sel_lo] = pos[order[t]];
        if (Kout > 0) {
            SD_TAKE(dcolsel, sizeof(int) * (size_t)Kout);
            JX_HIP(hipMemcpyAsync(dcolsel.p, C.stage(root_sel), sizeof(int) * (size_t)Kout, hipMemcpyHostToDevice, st));
        }
    }

    // ---- eigenvectors: C[:, 0:K] = [Q1(:,S1) U1 ; Q2(:,S2) U2] + Side(:,Sm) Um ; deflated columns copied -----
    std::vector<int> rows1, rows2, rowsm;      // kept-list positions by where the column lives
    std::vector<int> c1, c2, cm;               // and the column index inside that storage
    for (int r = 0; r < K; ++r) {
        const Loc l = loc[kept[r]];
        if (l.kind == 0) { rows1.push_back(r); c1.push_back(l.idx); }
        else if (l.kind == 1) { rows2.push_back(r); c2.push_back(l.idx); }
        else { rowsm.push_back(r); cm.push_back(l.idx); }
    }
    // deflated columns go out first (their storage is about to be compacted)
    {
        std::vector<int> s0, t0, s1, t1, s2, t2;
        for (size_t r = 0; r < defl.size(); ++r) {
            const Loc l = loc[defl[r]];
            const int dstc = defl_dst[r];
            if (dstc < 0) continue;                     // outside the column window
            if (l.kind == 0) { s0.push_back(l.idx); t0.push_back(dstc); }
            else if (l.kind == 1) { s1.push_back(l.idx); t1.push_back(dstc); }
            else { s2.push_back(l.idx); t2.push_back(dstc); }
        }
        auto place = [&](const std::vector<int> &sc, const std::vector<int> &dc, const double *src, int64_t lds,
                         int rows, int row_off) -> int {
            if (sc.empty()) return 0;
            ABuf a, b;
            SD_TAKE(a, sizeof(int) * sc.size());
            SD_TAKE(b, sizeof(int) * dc.size());
            JX_HIP(hipMemcpyAsync(a.p, C.stage(sc), sizeof(int) * sc.size(), hipMemcpyHostToDevice, st));
            JX_HIP(hipMemcpyAsync(b.p, C.stage(dc), sizeof(int) * dc.size(), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(sd_place_cols_kernel, dim3(64, (unsigned)sc.size()), dim3(256), 0, st, src, lds, rows,
                               a.as<int>(), d_c, (int64_t)n, n, row_off, b.as<int>(), (int)sc.size());
            JX_LAUNCH_CHECK();
            return 0;
        };
        if (place(s0, t0, q1.as<double>(), k1, k1, 0)) return 1;
        if (place(s1, t1, q2.as<double>(), k2, k2, k1)) return 1;
        if (place(s2, t2, side.as<double>(), n, n, 0)) return 1;
    }
    if (K > 0 && Kout > 0) {
        // the kept columns of Q1 / Q2 as contiguous blocks: in place when nothing was deflated there, else one
        // gather launch into the arena
        auto compact = [&](std::vector<int> &rows, std::vector<int> &cols, double *&q, int len) -> int {
            std::vector<int> idx(rows.size());
            std::iota(idx.begin(), idx.end(), 0);
            std::sort(idx.begin(), idx.end(), [&](int a, int b) { return cols[a] < cols[b]; });
            std::vector<int> r2(rows.size()), c2v(rows.size());
            bool identity = true;
            for (size_t t = 0; t < idx.size(); ++t) {
                r2[t] = rows[idx[t]];
                c2v[t] = cols[idx[t]];
                identity = identity && (c2v[t] == (int)t);
            }
            rows.swap(r2);
            cols.swap(c2v);
            if (identity || cols.empty()) return 0;
            ABuf g, sc, dc;
            SD_TAKE(g, sizeof(double) * cols.size() * (size_t)len);
            SD_TAKE(sc, sizeof(int) * cols.size());
            SD_TAKE(dc, sizeof(int) * cols.size());
            std::vector<int> seq(cols.size());
            std::iota(seq.begin(), seq.end(), 0);
            JX_HIP(hipMemcpyAsync(sc.p, C.stage(cols), sizeof(int) * cols.size(), hipMemcpyHostToDevice, st));
            JX_HIP(hipMemcpyAsync(dc.p, C.stage(seq), sizeof(int) * cols.size(), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(sd_place_cols_kernel, dim3(16, (unsigned)cols.size()), dim3(256), 0, st, q, (int64_t)len,
                               len, sc.as<int>(), g.as<double>(), (int64_t)len, len, 0, dc.as<int>(), (int)cols.size());
            JX_LAUNCH_CHECK();
            q = g.as<double>();
            return 0;
        };
        double *q1c = q1.as<double>(), *q2c = q2.as<double>();
        if (compact(rows1, c1, q1c, k1)) return 1;
        if (compact(rows2, c2, q2c, k2)) return 1;
        const double one = 1.0, zero = 0.0;
        auto gemm_part = [&](const std::vector<int> &rows, const double *q, int64_t ldq, int qrows, double *cdst,
                             double beta) -> int {
            const int nr = (int)rows.size();
            if (nr == 0) {
                if (beta == 0.0)     // nothing of this block survives: its rows of the root columns are zero
                    JX_HIP(hipMemset2DAsync(cdst, sizeof(double) * (size_t)n, 0, sizeof(double) * (size_t)qrows,
                                            (size_t)Kout, st));
                return 0;
            }
            const size_t gmark = C.ar.off;
            ABuf u, rm;
            SD_TAKE(u, sizeof(double) * (size_t)nr * Kout);
            SD_TAKE(rm, sizeof(int) * (size_t)nr);
            JX_HIP(hipMemcpyAsync(rm.p, C.stage(rows), sizeof(int) * (size_t)nr, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(sd_form_u_kernel, dim3((nr + SD_THREADS - 1) / SD_THREADS, Kout), dim3(SD_THREADS), 0, st,
                               dk.as<double>(), dzh.as<double>(), dtau.as<double>(), dorg.as<int>(), dinv.as<double>(),
                               rm.as<int>(), nr, Kout, u.as<double>(), (int64_t)nr, windowed ? dcolsel.as<int>() : (const int *)nullptr);
            JX_LAUNCH_CHECK();
            // C (qrows x K) = Q (qrows x nr) U (nr x K) + beta C: on the int8 matrix pipes (k_ozgemm.hip) when the product is
            // large enough to pay for slicing its operands, else the own f64 MFMA GEMM.  No vendor GEMM (rounds 1 - 3: rocBLAS).
            if (sd_use_oz(qrows, Kout, nr)) {
                ABuf img;
                static const int pl = getenv("JXGPU_STEDC_OZ_PLANES") ? std::min(6, std::max(4, atoi(getenv("JXGPU_STEDC_OZ_PLANES")))) : 0;
                const size_t ba = oz_image_bytes(qrows, nr, pl), bb = oz_image_bytes(Kout, nr, pl);
                SD_TAKE(img, ba + bb);
                OzImage ia = oz_image_at(img.p, qrows, nr, pl), ib = oz_image_at(img.as<char>() + ba, Kout, nr, pl);
                if (oz_slice(st, q, 1, ldq, ia)) return 1;                       // element (r, k) = q[r + k ldq]
                if (oz_slice(st, u.as<double>(), nr, 1, ib)) return 1;           // element (j, k) = u[k + j nr]
                if (oz_mm(st, ia, ib, qrows, Kout, one, beta, cdst, n, 0)) return 1;
            } else {
                if (dgemm(st, false, false, qrows, Kout, nr, one, q, ldq, u.as<double>(), nr, beta, cdst, n, 1, nullptr, 0)) return 1;
            }
            C.ar.off = gmark;   // stream order protects the reuse
            return 0;
        };
        if (gemm_part(rows1, q1c, k1, k1, d_c, zero)) return 1;
        if (gemm_part(rows2, q2c, k2, k2, d_c + k1, zero)) return 1;
        if (!rowsm.empty()) {
            // side columns are dense: gather them contiguously, then C(:, 0:K) += Side U_m
            ABuf sg;
            SD_TAKE(sg, sizeof(double) * (size_t)rowsm.size() * n);
            for (size_t t = 0; t < cm.size(); ++t)
                JX_HIP(hipMemcpyAsync(sg.as<double>() + t * (size_t)n, side.as<double>() + (size_t)cm[t] * n,
                                      sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
            if (gemm_part(rowsm, sg.as<double>(), n, n, d_c, one)) return 1;
        }
    }
    if (side.p) JX_HIP(hipStreamSynchronize(st));   // the side buffer (hipMalloc) is released on return
    DC_MARK("merged");
    C.ar.off = mark;
    h_w.swap(w);
    return 0;
}

// d_d (n), d_e (n-1): tridiagonal T (both overwritten).  d_c: (n,n) column-major, receives the eigenvectors of T as
// columns in the order given by h_perm (h_perm[r] = column holding the r-th smallest eigenvalue); d_d receives the
// eigenvalues ascending.  leaf: largest problem handed to rocSOLVER's dstedc.
// sel_lo < sel_hi (several ranks): only the columns of the eigenvalues ranked [sel_lo, sel_hi) are formed by the top-level
// merge; h_perm then has sel_hi - sel_lo entries, h_perm[t] = column of d_c holding the (sel_lo + t)-th smallest eigenvalue.
int stedc_split(rocblas_handle h, hipStream_t st, int n, double *d_d, double *d_e, double *d_c, int leaf,
                std::vector<int> &h_perm, int sel_lo, int sel_hi) {
    std::vector<double> hd((size_t)n), he((size_t)n);
    JX_HIP(hipMemcpyAsync(hd.data(), d_d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    JX_HIP(hipMemcpyAsync(he.data(), d_e, sizeof(double) * (size_t)(n - 1), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    // workspace: per level the two half-size eigenvector blocks (n^2/2), their compacted copies when columns were
    // deflated (<= n^2/2) and one U factor at a time (<= n^2/2); the levels below reuse the space above q1/q2, so
    // 2 n^2 doubles cover the recursion; vectors and index lists are O(n) per level
    ScratchLease arena;
    const size_t bytes = dc_arena_bytes(n);
    if (arena.take(1, bytes)) return 1;
    g_dc_trace = getenv("JXGPU_EIGH_TRACE") != nullptr;
    g_dc_t0 = std::chrono::steady_clock::now();
    DcCtx C;
    C.h = h;
    C.st = st;
    C.leaf = leaf;
    C.par_depth = getenv("JXGPU_STEDC_PAR") ? atoi(getenv("JXGPU_STEDC_PAR")) : 3;   // up to 8 concurrent sub-trees
    // rocprofv3 counter collection serialises dispatches across queues; a WRITE_SIZE pass over this code with its
    // concurrent host threads and streams stopped making progress (25 minutes, killed) -- run the halves in turn there
    const char *pmc = getenv("ROCPROF_COUNTER_COLLECTION");
    if (pmc && pmc[0] == '1' && !getenv("JXGPU_STEDC_PAR")) C.par_depth = 0;
    JX_HIP(hipGetDevice(&C.device));
    C.ar.base = (char *)arena.p;
    C.ar.cap = bytes;
    C.hd0 = hd.data();
    C.he0 = he.data();
    C.d_base = d_d;
    C.e_base = d_e;
    std::vector<int> leaf_off, leaf_len, splits;
    std::vector<int64_t> leaf_zoff;
    std::vector<double> hleaf_w;
    DevBuf zpool, dplan, derr;
    // leaves: one batched launch of the QL kernel (default) or rocSOLVER dstedc per leaf (JXGPU_STEDC_OWNLEAF=0, and
    // whenever the requested leaf size exceeds what fits the LDS-resident eigenvector block)
    const char *ol = getenv("JXGPU_STEDC_OWNLEAF");
    const bool want_own = !(ol && ol[0] == '0');
    const int own_leaf_rows = (getenv("JXGPU_STEDC_LEAF") && leaf < SD_LEAF) ? (leaf < 2 ? 2 : leaf) : SD_LEAF;
    if (want_own && n > own_leaf_rows) {
        C.own_leaf = true;
        C.leaf = own_leaf_rows;
        // the same halving rule as the recursion, down to <= SD_LEAF rows
        std::vector<std::pair<int, int>> stack = {{0, n}};
        std::vector<std::pair<int, int>> leaves;
        while (!stack.empty()) {
            const auto [g0, len] = stack.back();
            stack.pop_back();
            if (len <= own_leaf_rows || len < 4) {
                leaves.push_back({g0, len});
                continue;
            }
            const int k1 = len / 2;
            splits.push_back(g0 + k1);
            stack.push_back({g0, k1});
            stack.push_back({g0 + k1, len - k1});
        }
        std::sort(leaves.begin(), leaves.end());
        int64_t zo = 0;
        for (const auto &lf : leaves) {
            leaf_off.push_back(lf.first);
            leaf_len.push_back(lf.second);
            leaf_zoff.push_back(zo);
            zo += (int64_t)lf.second * lf.second;
        }
        const int nleaf = (int)leaves.size(), nsplit = (int)splits.size();
        if (zpool.alloc(sizeof(double) * (size_t)zo) || derr.alloc(sizeof(int)) ||
            dplan.alloc(sizeof(int) * (size_t)(2 * nleaf + nsplit) + sizeof(int64_t) * (size_t)nleaf + 64))
            return 1;
        int64_t *p_zoff = reinterpret_cast<int64_t *>(dplan.p);
        int *p_off = reinterpret_cast<int *>(p_zoff + nleaf), *p_len = p_off + nleaf, *p_split = p_len + nleaf;
        JX_HIP(hipMemcpyAsync(p_zoff, leaf_zoff.data(), sizeof(int64_t) * nleaf, hipMemcpyHostToDevice, st));
        JX_HIP(hipMemcpyAsync(p_off, leaf_off.data(), sizeof(int) * nleaf, hipMemcpyHostToDevice, st));
        JX_HIP(hipMemcpyAsync(p_len, leaf_len.data(), sizeof(int) * nleaf, hipMemcpyHostToDevice, st));
        JX_HIP(hipMemcpyAsync(p_split, splits.data(), sizeof(int) * nsplit, hipMemcpyHostToDevice, st));
        JX_HIP(hipMemsetAsync(derr.p, 0, sizeof(int), st));
        hipLaunchKernelGGL(sd_tear_all_kernel, dim3((nsplit + 255) / 256), dim3(256), 0, st, d_d, d_e, p_split, nsplit);
        JX_LAUNCH_CHECK();
        const size_t lds = sizeof(double) * ((size_t)SD_LEAF * SD_LEAF + 4 * SD_LEAF);
        static bool attr_done = false;
        if (!attr_done) {
            JX_HIP(hipFuncSetAttribute((const void *)sd_leaf_ql_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
            attr_done = true;
        }
        hipLaunchKernelGGL(sd_leaf_ql_kernel, dim3(nleaf), dim3(256), lds, st, d_d, d_e, p_off, p_len, p_zoff,
                           zpool.as<double>(), derr.as<int>());
        JX_LAUNCH_CHECK();
        hleaf_w.resize((size_t)n);
        int herr = 0;
        JX_HIP(hipMemcpyAsync(hleaf_w.data(), d_d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        JX_HIP(hipMemcpyAsync(&herr, derr.p, sizeof(int), hipMemcpyDeviceToHost, st));
        JX_HIP(hipStreamSynchronize(st));
        if (herr) return fail("stedc: the QL leaf solver did not converge");
        if (g_dc_trace)
            fprintf(stderr, "[jxgpu stedc] leaves done (%d) at %7.2f ms\n", nleaf,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_dc_t0).count());
        C.leaf_off = &leaf_off;
        C.leaf_zoff = &leaf_zoff;
        C.zpool = zpool.as<double>();
        C.hleaf_w = hleaf_w.data();
    }
    std::vector<double> w;
    const bool windowed = sel_hi > sel_lo && sel_lo >= 0 && sel_hi <= n;
    if (windowed) {
        C.sel_lo = sel_lo;
        C.sel_hi = sel_hi;
    }
    if (stedc_dc(C, n, d_d, d_e, d_c, w)) return 1;
    h_perm.resize((size_t)n);
    std::iota(h_perm.begin(), h_perm.end(), 0);
    std::sort(h_perm.begin(), h_perm.end(), [&](int a, int b) { return w[a] < w[b] || (w[a] == w[b] && a < b); });
    std::vector<double> ws((size_t)n);
    for (int r = 0; r < n; ++r) ws[r] = w[h_perm[r]];
    if (windowed) {
        // the root problem was a leaf or had nothing to merge (sel_cols empty): the full layout stands, take its window
        if (C.sel_cols.empty()) h_perm.assign(h_perm.begin() + sel_lo, h_perm.begin() + sel_hi);
        else h_perm = C.sel_cols;
    }
    JX_HIP(hipMemcpyAsync(d_d, ws.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
    JX_HIP(hipStreamSynchronize(st));
    return 0;
}

// dst row r (row-major, n x n) = column perm[r] of the column-major src: the final "eigenvector j in row j" layout
__global__ void sd_gather_cols_kernel(const double *__restrict__ src, const int *__restrict__ perm, int n,
                                      double *__restrict__ dst) {
    // rows on gridDim.x (limit 2^31 - 1; gridDim.y stops at 65535), a grid-stride loop along the row
    const int r = blockIdx.x;
    const double *s = src + (int64_t)perm[r] * n;
    double *t = dst + (int64_t)r * n;
    for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < n; c += gridDim.y * blockDim.x) t[c] = s[c];
}

// launch geometry of the gather (exposed for the CPU-side test of large n)
void gather_cols_grid(int n, unsigned *gx, unsigned *gy) {
    *gx = (unsigned)n;
    *gy = 16;
}

// dst row r (r = 0 .. count-1, n entries each) = src column d_perm[r]
int launch_gather_cols_range(const double *src, const int *d_perm, int n, int count, double *dst, hipStream_t st) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(sd_gather_cols_kernel, dim3((unsigned)count, 16), dim3(256), 0, st, src, d_perm, n, dst);
    JX_LAUNCH_CHECK();
    return 0;
}

int launch_gather_cols(const double *src, const int *d_perm, int n, double *dst, hipStream_t st) {
    unsigned gx, gy;
    gather_cols_grid(n, &gx, &gy);
    hipLaunchKernelGGL(sd_gather_cols_kernel, dim3(gx, gy), dim3(256), 0, st, src, d_perm, n, dst);
    JX_LAUNCH_CHECK();
    return 0;
}

}  // namespace jx
