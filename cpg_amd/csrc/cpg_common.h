// Shared helpers for the libcpg_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/cpg_hip.h"

namespace cpg {

// thread-local last-error text (no global mutable state shared between host threads)
char *err_buf();
int fail(int code, const char *fmt, ...);
// cpg_set_shared_chip_hint() of the calling host thread (thread-local; never changes results, only split counts / grids)
int shared_chip_hint();

inline int hip_status(hipError_t e, const char *what) {
    if (e == hipSuccess) return CPG_OK;
    return fail(CPG_E_HIP_BASE + (int)e, "%s: %s", what, hipGetErrorString(e));
}

#define CPG_CHECK_LAUNCH(what)                                   \
    do {                                                         \
        hipError_t e__ = hipGetLastError();                      \
        if (e__ != hipSuccess) return cpg::hip_status(e__, what); \
    } while (0)

#define CPG_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return cpg::fail(CPG_E_INVALID, __VA_ARGS__); \
    } while (0)

// models/layers.py:14-19 -- fp32 compare, NaN falls through unchanged
__device__ __forceinline__ float binarize(float pm, float thr) {
    return pm > thr ? 1.0f : (pm <= thr ? 0.0f : pm);
}

constexpr int kCUs = 256;        // MI355X
constexpr int kXCDs = 8;

// grid for an HBM-streaming elementwise pass: <= 8 blocks of 256 threads per CU, grid-stride beyond
inline unsigned stream_grid(int64_t work_items, int per_block) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (int64_t)kCUs * 8) b = (int64_t)kCUs * 8;
    return (unsigned)b;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

}  // namespace cpg
