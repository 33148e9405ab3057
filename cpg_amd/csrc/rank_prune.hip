// K6: magnitude-rank pruning of one layer, entirely on device (utils/prune.py:30-53).
//
// The reference gathers the candidate weights, copies them to the host and calls CPU kthvalue.
// Here the k-th smallest |w| is found by a 3-pass MSB-first radix select over the fp32 bit
// pattern of |w| (non-negative floats order like their bit patterns; NaN patterns sort last,
// as torch.kthvalue treats them):   pass 0: bits 30..20 (2048 bins, also counts candidates),
// pass 1: bits 19..10, pass 2: bits 9..0 (1024 bins each).  A one-block scan after each pass
// narrows the prefix; a final pass rewrites the owner ids.  Nothing synchronises with the host:
// k = round-half-even(ratio * n_cand) is formed on device in fp64 exactly as python's round().
//
// HBM traffic per element: 3 x (4 B w + 1 B owner) + final (4 + 1 r, 1 w) = 21 B.
#include "cpg_common.h"

using namespace cpg;

namespace {

constexpr int kThreads = 256;
constexpr int kBins0 = 2048, kBins12 = 1024;

struct RpState {               // lives at the head of the workspace
    unsigned long long k_rem;  // rank still to descend (1-indexed within the current prefix bucket)
    unsigned long long n_cand;
    unsigned long long k;
    unsigned prefix;           // selected high bits so far, right-aligned
    int status;
    unsigned long long released;
};

struct RpWs {
    RpState st;
    unsigned hist[kBins0];
};

__device__ __forceinline__ unsigned key_of(float w) { return __float_as_uint(w) & 0x7FFFFFFFu; }

// PASS 0: bin = key >> 20; PASS 1: needs key >> 20 == prefix, bin = (key >> 10) & 1023;
// PASS 2: needs key >> 10 == prefix, bin = key & 1023.
template <int PASS>
__global__ __launch_bounds__(kThreads) void k_rp_hist(const float *__restrict__ w, const uint8_t *__restrict__ owner,
                                                      int cur, int64_t n, RpWs *__restrict__ ws, int vec_ok) {
    constexpr int NB = PASS == 0 ? kBins0 : kBins12;
    __shared__ unsigned h[NB];
    if (PASS > 0 && ws->st.status != CPG_OK) return;         // k out of range: nothing to select
    for (int i = threadIdx.x; i < NB; i += kThreads) h[i] = 0;
    __syncthreads();
    const unsigned prefix = PASS > 0 ? ws->st.prefix : 0u;
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    unsigned zero_cnt = 0;      // key == 0 (released slots zeroed by apply_mask) would hammer one bin
    auto visit = [&](float wv, int o) {
        if (o != cur && o != 0) return;
        const unsigned key = key_of(wv);
        if (PASS == 0) {
            if (key == 0) { zero_cnt++; return; }
            atomicAdd(&h[key >> 20], 1u);
        } else if (PASS == 1) {
            if ((key >> 20) != prefix) return;
            if (key == 0) { zero_cnt++; return; }
            atomicAdd(&h[(key >> 10) & 1023u], 1u);
        } else {
            if ((key >> 10) != prefix) return;
            if (key == 0) { zero_cnt++; return; }
            atomicAdd(&h[key & 1023u], 1u);
        }
    };
    if (vec_ok) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nthreads) {
            const uint32_t o4 = reinterpret_cast<const uint32_t *>(owner)[i];
            const int o0 = o4 & 255, o1 = (o4 >> 8) & 255, o2 = (o4 >> 16) & 255, o3 = o4 >> 24;
            const bool c0 = (o0 == cur || o0 == 0), c1 = (o1 == cur || o1 == 0), c2 = (o2 == cur || o2 == 0),
                       c3 = (o3 == cur || o3 == 0);
            if (c0 || c1 || c2 || c3) {
                const float4 v = reinterpret_cast<const float4 *>(w)[i];
                visit(v.x, o0);
                visit(v.y, o1);
                visit(v.z, o2);
                visit(v.w, o3);
            }
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) visit(w[i], owner[i]);
    } else {
        for (int64_t i = tid; i < n; i += nthreads) visit(w[i], owner[i]);
    }
    if (zero_cnt) atomicAdd(&h[0], zero_cnt);
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += kThreads)
        if (h[i]) atomicAdd(&ws->hist[i], h[i]);
}

// one block: locate the bin holding rank k_rem, extend the prefix, clear the histogram
template <int PASS>
__global__ __launch_bounds__(kThreads) void k_rp_scan(RpWs *__restrict__ ws, double ratio) {
    constexpr int NB = PASS == 0 ? kBins0 : kBins12;
    constexpr int PER = NB / kThreads;          // bins per thread (8 or 4), contiguous
    __shared__ unsigned long long part[kThreads];
    __shared__ unsigned long long total_s;
    const int t = threadIdx.x;
    if (PASS > 0 && ws->st.status != CPG_OK) return;
    // read the incoming rank BEFORE any barrier: the thread owning the crossing bin rewrites it below
    const unsigned long long k_in = PASS > 0 ? ws->st.k_rem : 0ull;
    const unsigned prefix_in = PASS > 0 ? ws->st.prefix : 0u;
    unsigned local[PER];
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        local[j] = ws->hist[t * PER + j];
        s += local[j];
    }
    part[t] = s;
    __syncthreads();
    if (t == 0) {                               // 256-entry serial exclusive scan: negligible
        unsigned long long run = 0;
        for (int i = 0; i < kThreads; ++i) {
            const unsigned long long v = part[i];
            part[i] = run;
            run += v;
        }
        total_s = run;
    }
    __syncthreads();
    unsigned long long k_rem;
    if (PASS == 0) {
        const unsigned long long ncand = total_s;
        // python: round(pruning_ratio * tensor.numel()) -- fp64 product, round-half-even
        const double kd = rint(ratio * (double)ncand);
        long long k = (kd < 0.0) ? 0 : (long long)kd;
        if (kd > 9.0e18) k = (long long)9.0e18;
        const bool ok = (k >= 1) && ((unsigned long long)k <= ncand);
        if (t == 0) {
            ws->st.n_cand = ncand;
            ws->st.k = (unsigned long long)k;
            ws->st.status = ok ? CPG_OK : CPG_E_KRANGE;
            ws->st.released = 0;
            if (!ok) {
                ws->st.prefix = 0;
                ws->st.k_rem = 0;
            }
        }
        if (!ok) {                              // leave a clean histogram behind for the next call
#pragma unroll
            for (int j = 0; j < PER; ++j) ws->hist[t * PER + j] = 0;
            return;
        }
        k_rem = (unsigned long long)k;
    } else {
        k_rem = k_in;
    }
    // thread owning the crossing bin: part[t] < k_rem <= part[t] + s
    const unsigned long long before = part[t];
    if (before < k_rem && k_rem <= before + s) {
        unsigned long long run = before;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (run < k_rem && k_rem <= run + local[j]) {
                const unsigned bin = t * PER + j;
                ws->st.prefix = PASS == 0 ? bin : ((prefix_in << 10) | bin);
                ws->st.k_rem = k_rem - run;
            }
            run += local[j];
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) ws->hist[t * PER + j] = 0;
}

// final pass: owner[(|w| <= cutoff) & (owner == cur)] = 0, in the integer domain (NaN never <=)
__global__ __launch_bounds__(kThreads) void k_rp_apply(const float *__restrict__ w, uint8_t *__restrict__ owner, int cur,
                                                       int64_t n, RpWs *__restrict__ ws, int vec_ok) {
    if (ws->st.status != CPG_OK) return;
    const unsigned cut = ws->st.prefix;          // full 31-bit key of the k-th smallest |w|
    if (cut > 0x7F800000u) return;               // cutoff is NaN: `abs(w) <= nan` is false everywhere
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    unsigned cnt = 0;
    if (vec_ok) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nthreads) {
            uint32_t o4 = reinterpret_cast<const uint32_t *>(owner)[i];
            const int o0 = o4 & 255, o1 = (o4 >> 8) & 255, o2 = (o4 >> 16) & 255, o3 = o4 >> 24;
            if (o0 == cur || o1 == cur || o2 == cur || o3 == cur) {
                const float4 v = reinterpret_cast<const float4 *>(w)[i];
                uint32_t keep = 0xFFFFFFFFu;
                if (o0 == cur && key_of(v.x) <= cut) { keep &= 0xFFFFFF00u; cnt++; }
                if (o1 == cur && key_of(v.y) <= cut) { keep &= 0xFFFF00FFu; cnt++; }
                if (o2 == cur && key_of(v.z) <= cut) { keep &= 0xFF00FFFFu; cnt++; }
                if (o3 == cur && key_of(v.w) <= cut) { keep &= 0x00FFFFFFu; cnt++; }
                if (keep != 0xFFFFFFFFu) reinterpret_cast<uint32_t *>(owner)[i] = o4 & keep;
            }
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads)
            if (owner[i] == cur && key_of(w[i]) <= cut) { owner[i] = 0; cnt++; }
    } else {
        for (int64_t i = tid; i < n; i += nthreads)
            if (owner[i] == cur && key_of(w[i]) <= cut) { owner[i] = 0; cnt++; }
    }
    // wave reduce, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&ws->st.released, (unsigned long long)cnt);
}

__global__ void k_rp_init(RpWs *ws) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < kBins0) ws->hist[t] = 0;
    if (t == 0) {
        ws->st.k_rem = 0;
        ws->st.n_cand = 0;
        ws->st.k = 0;
        ws->st.prefix = 0;
        ws->st.status = CPG_OK;
        ws->st.released = 0;
    }
}

__global__ void k_rp_finish(const RpWs *ws, cpg_prune_result *res) {
    res->n_candidates = (int64_t)ws->st.n_cand;
    res->k = (int64_t)ws->st.k;
    res->n_released = (int64_t)ws->st.released;
    res->status = ws->st.status;
    res->cutoff = ws->st.status == CPG_OK ? __uint_as_float(ws->st.prefix) : 0.0f;
}

}  // namespace

extern "C" size_t cpg_rank_prune_workspace_bytes(void) { return sizeof(RpWs); }

extern "C" int cpg_rank_prune(const float *w, uint8_t *owner, int32_t cur, double ratio, int64_t n,
                              cpg_prune_result *result, void *ws_v, size_t ws_bytes, void *stream_v) {
    CPG_REQUIRE(w && owner && result && ws_v, "cpg_rank_prune: null pointer");
    CPG_REQUIRE(n >= 0 && n < (int64_t)0xFFFFFFFFll, "cpg_rank_prune: n=%lld outside [0, 2^32)", (long long)n);
    CPG_REQUIRE(cur >= 0 && cur <= 255, "cpg_rank_prune: owner id %d out of uint8 range", cur);
    if (ws_bytes < sizeof(RpWs)) return fail(CPG_E_WORKSPACE, "cpg_rank_prune: workspace %zu < %zu", ws_bytes, sizeof(RpWs));
    CPG_REQUIRE((((uintptr_t)ws_v) & 7) == 0, "cpg_rank_prune: workspace must be 8-byte aligned");
    hipStream_t stream = (hipStream_t)stream_v;
    RpWs *ws = (RpWs *)ws_v;
    const int vec = (((uintptr_t)w) & 15) == 0 && (((uintptr_t)owner) & 3) == 0;
    const dim3 block(kThreads);
    const dim3 grid(n > 0 ? stream_grid(n, kThreads * 8) : 1);
    hipLaunchKernelGGL(k_rp_init, dim3(kBins0 / 256), dim3(256), 0, stream, ws);
    hipLaunchKernelGGL(k_rp_hist<0>, grid, block, 0, stream, w, owner, cur, n, ws, vec);
    hipLaunchKernelGGL(k_rp_scan<0>, dim3(1), block, 0, stream, ws, ratio);
    hipLaunchKernelGGL(k_rp_hist<1>, grid, block, 0, stream, w, owner, cur, n, ws, vec);
    hipLaunchKernelGGL(k_rp_scan<1>, dim3(1), block, 0, stream, ws, ratio);
    hipLaunchKernelGGL(k_rp_hist<2>, grid, block, 0, stream, w, owner, cur, n, ws, vec);
    hipLaunchKernelGGL(k_rp_scan<2>, dim3(1), block, 0, stream, ws, ratio);
    hipLaunchKernelGGL(k_rp_apply, grid, block, 0, stream, w, owner, cur, n, ws, vec);
    hipLaunchKernelGGL(k_rp_finish, dim3(1), dim3(1), 0, stream, ws, result);
    CPG_CHECK_LAUNCH("cpg_rank_prune");
    return CPG_OK;
}
