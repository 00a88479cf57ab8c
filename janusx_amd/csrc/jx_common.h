// Shared host/device helpers for libjxgpu (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/jxgpu.h"

namespace jx {

void set_error(const std::string &msg);
int fail(const std::string &msg);

#define JX_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (call);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            return ::jx::fail(std::string(#call) + " failed: " + hipGetErrorString(_e) + " (" +       \
                              __FILE__ + ":" + std::to_string(__LINE__) + ")");                       \
        }                                                                                             \
    } while (0)

#define JX_LAUNCH_CHECK()                                                                             \
    do {                                                                                              \
        hipError_t _e = hipGetLastError();                                                            \
        if (_e != hipSuccess) {                                                                       \
            return ::jx::fail(std::string("kernel launch failed: ") + hipGetErrorString(_e) + " (" +  \
                              __FILE__ + ":" + std::to_string(__LINE__) + ")");                       \
        }                                                                                             \
    } while (0)

// true when `p` points into device memory (a payload that is already resident in HBM: torch CUDA tensor, hipMalloc).  api.cpp
bool is_device_ptr(const void *p);

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int num_tiles(int n) { return ceil_div(n, JXG_TILE); }

// RAII device buffer
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t nbytes) {
        release();
        if (nbytes == 0) nbytes = 16;
        hipError_t e = hipMalloc(&p, nbytes);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(std::string("hipMalloc(") + std::to_string(nbytes) + ") failed: " + hipGetErrorString(e));
        }
        bytes = nbytes;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Scratch block that outlives the call (one per purpose, grown on demand): the eigendecomposition asks for ~4 n^2
// doubles per call, and each hipMalloc / hipFree pair of that size costs milliseconds (hipFree synchronises the device).
// Blocks larger than kScratchKeepBytes are released again by the caller's ScratchLease so big problems do not pin HBM.
constexpr size_t kScratchKeepBytes = (size_t)6 << 30;
// which: 0 eigh C, 1 stedc arena, 2 sytrd workspace, 3 / 5 / 6 ormtr, 4 stage-2 reflectors, 7 nibble image of the fp4 Gram.  Returns 0 and *p (kept block), or 1 when
// the block should be a private allocation (slot in use by a concurrent call, or too large to keep).  api.cpp
int scratch_acquire(int which, size_t bytes, void **p);
void async_pool_keep();   // release threshold of the stream-ordered pool (api.cpp): call in front of hipMallocAsync users
void scratch_release(int which);
// stream-ordered allocation released behind the work queued on the same stream, on every path out of the scope
struct AsyncBlock {
    void *p = nullptr;
    hipStream_t st = nullptr;
    AsyncBlock() = default;
    AsyncBlock(const AsyncBlock &) = delete;
    AsyncBlock &operator=(const AsyncBlock &) = delete;
    ~AsyncBlock() {
        if (p) (void)hipFreeAsync(p, st);
    }
    int alloc(size_t bytes, hipStream_t s) {
        st = s;
        async_pool_keep();
        hipError_t e = hipMallocAsync(&p, bytes ? bytes : 16, s);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(std::string("hipMallocAsync(") + std::to_string(bytes) + ") failed: " + hipGetErrorString(e));
        }
        return 0;
    }
};

struct ScratchLease {
    int which = -1;
    DevBuf own;
    void *p = nullptr;
    int take(int w, size_t bytes) {
        if (bytes <= kScratchKeepBytes) {
            const int rc = scratch_acquire(w, bytes, &p);
            if (rc == 0) {
                which = w;
                return 0;
            }
            if (rc == 2) return 1;   // allocation failure (message set)
        }
        if (own.alloc(bytes)) return 1;
        p = own.p;
        return 0;
    }
    ~ScratchLease() {
        if (which >= 0) scratch_release(which);
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace jx
