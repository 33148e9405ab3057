// Data-parallel gradient payload compaction ("only owner == cur slots need reducing").
//
// After backward, the reference's gradient routing (utils/prune.py:195-211) zeroes every weight-gradient slot that the
// current task does not own and every piggymask-gradient slot outside the older tasks' weights -- so an all-reduce of those
// slots moves bytes over xGMI that are discarded one kernel later.  From task 2 on that is most of the 537 MB: a task owns
// what earlier tasks released.  These kernels gather the surviving slots of one layer into a dense buffer (in natural
// element order, i.e. exactly g.flatten()[selected]) and scatter the reduced values back; owner masks are replicated state,
// so every rank packs the same slots in the same order and the all-reduce of the packed buffers is element-aligned.
//
//   select 0: owner == cur            (weight gradients)
//   select 1: 0 < owner < cur         (piggymask gradients in finetune mode)
//
// A block owns CPG_PACK_BLOCK = 1024 consecutive elements (4 per thread: one dword of owner ids, one float4 of gradient,
// the latter only when the quad has a selected slot).  Positions inside a block come from wave ballots + popcounts (four
// ballots per wave give every lane the number of selected slots in lower lanes) and a 4-entry LDS scan over the block's
// waves; positions of blocks from an exclusive prefix sum of the per-block counts (cpg_owned_block_counts -> cumsum by the
// caller, cached until a mask mutates).  All three kernels are single HBM streaming passes.
#include "cpg_common.h"

using namespace cpg;

namespace {

constexpr int kThreads = 256;
constexpr int kPerBlock = 1024;

__device__ __forceinline__ bool selected(int o, int cur, int select) { return select == 0 ? (o == cur) : (o > 0 && o < cur); }

// owner ids of the thread's 4 elements -> predicate bits (bit e = element base + e selected); elements >= n are never selected
__device__ __forceinline__ unsigned quad_bits(const uint8_t *__restrict__ owner, int64_t base, int64_t n, int cur, int select) {
    unsigned bits = 0;
    if (base + 3 < n) {
        const uint32_t o4 = *reinterpret_cast<const uint32_t *>(owner + base);
#pragma unroll
        for (int e = 0; e < 4; ++e) bits |= selected((o4 >> (8 * e)) & 255, cur, select) ? (1u << e) : 0u;
    } else {
        for (int e = 0; e < 4; ++e)
            if (base + e < n && selected(owner[base + e], cur, select)) bits |= 1u << e;
    }
    return bits;
}

// exclusive position of the thread's first selected element inside its block, and (via *block_total) the block's count
__device__ __forceinline__ int block_scan(unsigned bits, int *block_total) {
    __shared__ int wave_tot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int below = 0, wtot = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned long long m = __ballot((bits >> e) & 1u);
        below += __popcll(m & lt);
        wtot += __popcll(m);
    }
    if (lane == 0) wave_tot[wave] = wtot;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += wave_tot[w];
        tot += wave_tot[w];
    }
    *block_total = tot;
    return base + below;
}

__global__ __launch_bounds__(kThreads) void k_owned_counts(const uint8_t *__restrict__ owner, int cur, int select, int64_t n,
                                                           int *__restrict__ counts) {
    const int64_t base = (int64_t)blockIdx.x * kPerBlock + 4 * threadIdx.x;
    int tot;
    block_scan(quad_bits(owner, base, n, cur, select), &tot);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

template <bool PACK>
__global__ __launch_bounds__(kThreads) void k_pack(float *__restrict__ g, const uint8_t *__restrict__ owner, int cur, int select,
                                                   int64_t n, const int64_t *__restrict__ block_offsets, float *__restrict__ packed) {
    const int64_t base = (int64_t)blockIdx.x * kPerBlock + 4 * threadIdx.x;
    const unsigned bits = quad_bits(owner, base, n, cur, select);
    int tot;
    int pos = block_scan(bits, &tot);
    if (bits == 0) return;                               // quads without a selected slot touch neither g nor packed
    float *dst = packed + block_offsets[blockIdx.x];
    const bool vec = base + 3 < n && ((((uintptr_t)(g + base)) & 15) == 0);
    if (PACK) {
        float v[4];
        if (vec) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(g + base);
            v[0] = q[0], v[1] = q[1], v[2] = q[2], v[3] = q[3];
        } else {
            for (int e = 0; e < 4; ++e) v[e] = (bits >> e) & 1u ? g[base + e] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((bits >> e) & 1u) dst[pos++] = v[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((bits >> e) & 1u) g[base + e] = dst[pos++];
    }
}

}  // namespace

extern "C" int64_t cpg_owned_num_blocks(int64_t n) { return n <= 0 ? 0 : (n + kPerBlock - 1) / kPerBlock; }

extern "C" int cpg_owned_block_counts(const uint8_t *owner, int32_t cur, int32_t select, int64_t n, int32_t *counts, void *stream) {
    CPG_REQUIRE(n >= 0 && (select == 0 || select == 1), "cpg_owned_block_counts: bad argument");
    if (n == 0) return CPG_OK;
    CPG_REQUIRE(owner && counts, "cpg_owned_block_counts: null pointer");
    CPG_REQUIRE((((uintptr_t)owner) & 3) == 0, "cpg_owned_block_counts: owner must be 4-byte aligned");
    const int64_t blocks = cpg_owned_num_blocks(n);
    CPG_REQUIRE(blocks <= 0x7FFFFFFFll, "cpg_owned_block_counts: tensor too large");
    hipLaunchKernelGGL(k_owned_counts, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, owner, cur, select, n, counts);
    CPG_CHECK_LAUNCH("cpg_owned_block_counts");
    return CPG_OK;
}

extern "C" int cpg_pack_owned(const float *g, const uint8_t *owner, int32_t cur, int32_t select, int64_t n,
                              const int64_t *block_offsets, float *packed, void *stream) {
    CPG_REQUIRE(n >= 0 && (select == 0 || select == 1), "cpg_pack_owned: bad argument");
    if (n == 0) return CPG_OK;
    CPG_REQUIRE(g && owner && block_offsets && packed, "cpg_pack_owned: null pointer");
    CPG_REQUIRE((((uintptr_t)owner) & 3) == 0, "cpg_pack_owned: owner must be 4-byte aligned");
    hipLaunchKernelGGL(k_pack<true>, dim3((unsigned)cpg_owned_num_blocks(n)), dim3(kThreads), 0, (hipStream_t)stream, const_cast<float *>(g),
                       owner, cur, select, n, block_offsets, packed);
    CPG_CHECK_LAUNCH("cpg_pack_owned");
    return CPG_OK;
}

extern "C" int cpg_unpack_owned(const float *packed, const uint8_t *owner, int32_t cur, int32_t select, int64_t n,
                                const int64_t *block_offsets, float *g, void *stream) {
    CPG_REQUIRE(n >= 0 && (select == 0 || select == 1), "cpg_unpack_owned: bad argument");
    if (n == 0) return CPG_OK;
    CPG_REQUIRE(g && owner && block_offsets && packed, "cpg_unpack_owned: null pointer");
    CPG_REQUIRE((((uintptr_t)owner) & 3) == 0, "cpg_unpack_owned: owner must be 4-byte aligned");
    hipLaunchKernelGGL(k_pack<false>, dim3((unsigned)cpg_owned_num_blocks(n)), dim3(kThreads), 0, (hipStream_t)stream, g, owner, cur, select, n,
                       block_offsets, const_cast<float *>(packed));
    CPG_CHECK_LAUNCH("cpg_unpack_owned");
    return CPG_OK;
}
