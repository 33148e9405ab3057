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
// cpg_set_shared_chip_hint(): PROCESS-wide (a std::atomic -- the weight-gradient planners that read it run on autograd's
// engine threads, not on the thread that set it); never changes results, only split counts / grids
int shared_chip_hint();

// ---- library options -----------------------------------------------------------------------------------------------
// Every CPG_* switch of the dispatch code lives in ONE process-wide table of atomics that is filled from the environment
// ONCE, when the library is loaded -- no getenv() on any launch path -- and changed afterwards only through the C ABI
// (cpg_set_option / cpg_get_option, include/cpg_hip.h).  A switch that was never given reads OPT_UNSET.
// Boolean switches: environment text "0" or "" = off, anything else = on.  Integer switches: atoi().
enum Opt {
    // policy: which arithmetic / kernel family a launch may use (results stay inside the documented tolerances)
    OPT_NO_WINO,          // CPG_NO_WINO: direct kernels instead of Winograd F(2x2,3x3) everywhere
    OPT_NO_WINO_WGRAD,    // CPG_NO_WINO_WGRAD: ... for the weight gradient only
    OPT_NO_WINO_ODD,      // CPG_NO_WINO_ODD: no Winograd on odd-sized maps
    OPT_NO_STEM,          // CPG_NO_STEM: the general kernels instead of the 3-channel stem kernels
    OPT_NO_STEM_FUSE,     // CPG_NO_STEM_FUSE: report the fused stem (conv -> BatchNorm -> ReLU) as unsupported
    OPT_NO_DEAD_SKIP,     // CPG_NO_DEAD_SKIP: inference kernels do not skip dead channels
    OPT_NO_S2,            // CPG_NO_S2: generic kernel for 3x3 stride-2
    OPT_NO_V14,           // CPG_NO_V14: no channel-split tiles on 14x14 maps (direct kernels)
    OPT_DISABLE_CONV3X3,  // CPG_DISABLE_CONV3X3 / _CONV1X1 / _CONV1X1_WGRAD / _PW_GEMM: fall to the generic implicit-GEMM kernel
    OPT_DISABLE_CONV1X1,
    OPT_DISABLE_CONV1X1_WGRAD,
    OPT_DISABLE_PW_GEMM,
    // development / A-B tuning (tools/): tile, variant and split-count overrides
    OPT_C3_FORCE,         // CPG_C3_FORCE=n: tile configuration of k_c3_fwd
    OPT_C3W_BPC,          // CPG_C3W_BPC=n: split blocks per CU of k_c3_wgrad
    OPT_W3_PICK,          // CPG_W3_PICK=n: unit shape of k_c3_wgrad
    OPT_PWW_BPC,          // CPG_PWW_BPC=n: split blocks per CU of k_pw_wgrad
    OPT_STEM_BLOCKS,      // CPG_STEM_BLOCKS=n: persistent blocks of the stem kernels
    OPT_WINO_KERNEL,      // CPG_WINO_KERNEL=block|wave|pair|64 -> 0|1|2|3
    OPT_WINO_NW,          // CPG_WINO_NW=4|8
    OPT_WINO_PERSIST,     // CPG_WINO_PERSIST=0: one block per logical block
    OPT_WINO_GRIDS,       // CPG_WINO_GRIDS=n
    OPT_WW_UNITS,         // CPG_WW_UNITS=n: units per wave slot of k_wgw
    OPT_WW_XCD,           // CPG_WW_XCD=0: dispatch-order units in k_wgw
    OPT_PW_TILE,          // CPG_PW_TILE=0: the pointwise kernels' 128-row tile as 4 x 1 waves of 7 pixel fragments (224 pixels, round 3); 1: 2 x 2 waves of 2 x 4 fragments (256 pixels)
    OPT_WG3_SHARE,        // CPG_WG3_SHARE=0: k_wg3 as blocks of one unit (two waves), every unit transforming all of its B operands (round 3)
    OPT_WW_SHARE,         // CPG_WW_SHARE=0: every wave of k_wgw stages its own x rows (round 3); 2 / 4: force that sharing group
    OPT_WINO_TAIL,        // CPG_WINO_TAIL=0: no channel-split tail launch for the leftover units of k_wg3's last round (round 4 behaviour)
    OPT_FC_SMALL,         // CPG_FC_SMALL=0: linear layers at <= 64 rows on the batch-256 tiles / the generic split-K kernel (round 4 behaviour)
    OPT_COUNT
};
constexpr int OPT_UNSET = INT32_MIN;
int opt(Opt o);                                   // the value, or OPT_UNSET
inline bool opt_on(Opt o) { const int v = opt(o); return v != OPT_UNSET && v != 0; }     // boolean switches
inline int opt_or(Opt o, int dflt) { const int v = opt(o); return v == OPT_UNSET ? dflt : v; }

// ---- packed weight operands handed in by the caller (include/cpg_hip.h: cpg_conv2d_pack / cpg_conv2d_use_packed, ABI 3) ------------
// The pointwise and the one-/two-wave Winograd kernels stream the effective weight W * bin(pm) from a packed copy that every call used
// to produce into its own workspace first.  A caller may now produce it ONCE (and the forward's and the input gradient's in one launch)
// and hand it to the calls that need it.  The plumbing is a per-THREAD one-shot context: cpg_conv2d_use_packed arms it, the next conv
// entry point on that thread consumes (or drops) it; the same context in QUERY mode is how cpg_conv2d_pack finds out what a call of
// this shape would pack -- the call runs to its pack site, records the job and returns before anything is launched.
struct PackJob {
    int family;               // 0 none, 1 pointwise (k_pw_pack's layout), 2 Winograd per-lane U (k_wg1_pack's layout)
    int K, C;                 // the layer's weight is [K][C][R][S]
    int a, b, c, d;           // family 1: rows, Mp, dgrad, 0;  family 2: M, Cin, nch, dgrad
    long long total;          // work items of the pack pass
    size_t bytes;             // size of the packed operand
};
enum { PACK_NONE = 0, PACK_QUERY = 1, PACK_USE = 2 };
struct PackCtx {
    int mode;
    bool hit;                 // QUERY: a job was recorded;  USE: the operand was consumed
    PackJob job;
    const float *use;
    size_t use_bytes;
};
PackCtx &pack_ctx();          // thread-local
inline bool pack_query() { return pack_ctx().mode == PACK_QUERY; }
// At a pack site: 0 = pack into the workspace as always; 1 = QUERY (job recorded: return CPG_OK now); 2 = *pre is the caller's operand
// (skip the pack); < 0 = the caller's operand does not fit this launch (status to return).
int pack_site(const PackJob &job, const float **pre, const char *what);
// one launch that carries out up to two jobs (conv3x3_wino.hip)
int pack_jobs_launch(const PackJob *ja, float *dst_a, const PackJob *jb, float *dst_b, const float *w, const float *pm, float thr,
                     hipStream_t stream);

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
