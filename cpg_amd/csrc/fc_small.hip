// Input gradient of a masked linear layer at <= 64 rows (images per GPU): gx[b][i] = sum_o gy[b][o] * (W * bin(pm))[o][i]
// (models/layers.py:184-194, the autograd of F.linear(input, weight * mask, bias) with respect to input).
//
// The reference's own data-parallel split hands every GPU 256 / 8 = 32 images (CPG_cifar100_main_normal.py:112-114,199); at that
// batch features.45's input gradient is a STREAM of its 411 MB weight (25 088 x 4 096 fp32) against 6.6 GFLOP -- HBM-bound
// (0.065 ms at 6.3 TB/s).  The batch-256 path (a 128-row tile with too few tiles to fill the chip -> the generic split-K kernel) took
// 0.35 ms.  This kernel has no LDS in its main loop:
//   * B operand of v_mfma_f32_32x32x2_f32 (lane (li, lh) holds B[k = lh][n = li]): each lane loads a float4 of 4 consecutive input
//     features of row o + lh -- a half-wave reads 512 contiguous bytes of one weight row -- and the four components feed four MFMAs whose
//     output columns are i0 + 4 li + j (j = 0..3): a permutation of the 128 columns the wave owns, undone by storing the four
//     accumulators' elements as one float4.  With a piggymask the lane loads the piggymask float4 beside it and multiplies by
//     bin(pm) in registers (W and pm are read once, nothing is materialised).
//   * A operand (A[m = li][k = lh] = gy[b = li][o + lh]) comes from a K-major copy gyT[o][32 FM] made by a pack launch (0.5 MB,
//     L2-resident): one coalesced dword per lane and step.
//   * every wave keeps FC_D steps (row pairs) in flight in registers; a block's 4 waves take 4 consecutive row ranges of the same 128
//     columns and add their accumulators through LDS in a fixed order; row ranges are split over `nsplit` blocks when the column
//     tiles alone do not fill the chip, combined by k_split_reduce in a fixed order (deterministic for a given shape).
// MFMA issue would sustain 16 B / clk / CU (9.8 TB/s) at 32 rows, 4.9 TB/s at 64 rows: above 64 rows the GEMM kernels take over.
#include <algorithm>
#include "igemm_core.h"

using namespace cpg;

namespace {

constexpr int FC_COLS = 128;     // input features per block
constexpr int FC_UNIT = 16;      // weight rows per unit of the row split (a multiple of 2 D for every pipeline depth D below)

// gyT[o][MP] = gy[b][o] (b < batch, o < out_f), zero elsewhere
__global__ __launch_bounds__(256) void k_fc_pack_t(const float *__restrict__ gy, float *__restrict__ gyT, int batch, int out_f, int rows, int MP) {
    const int64_t total = (int64_t)rows * MP, nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += nthreads) {
        const int b = (int)(e % MP), o = (int)(e / MP);
        gyT[e] = (b < batch && o < out_f) ? gy[(int64_t)b * out_f + o] : 0.0f;
    }
}

// FC_D = D steps (pairs of weight rows) in flight per wave, NB resident blocks per CU
template <int FM, bool MASKED, int FC_D, int NB>
__global__ __launch_bounds__(256, NB) void k_fc_dgrad_small(const float *__restrict__ w, const float *__restrict__ pm, float thr,
                                                           const float *__restrict__ gyT, float *__restrict__ out, int batch, int in_f,
                                                           int tiles, int units_per_wave, unsigned w_bytes, unsigned a_bytes) {
    constexpr int MP = 32 * FM;
    static_assert(FC_UNIT % (2 * FC_D) == 0 || (2 * FC_D) % FC_UNIT == 0, "units and pipeline rounds nest");
    __shared__ float red[4 * 2 * 16 * 64];          // [wave][column phase j of a pair][accumulator element e][lane]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
    const int col = tile * FC_COLS + 4 * li;
    const bool col_ok = col < in_f;                  // in_f % 4 == 0: a float4 is inside the row or outside it
    const int r0 = (split * 4 + wave) * units_per_wave * FC_UNIT;
    const int rounds = units_per_wave * FC_UNIT / (2 * FC_D);          // (host: a whole number >= 1)

    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_p = __builtin_amdgcn_make_buffer_rsrc((void *)(MASKED ? pm : w), 0, MASKED ? w_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void *)gyT, 0, a_bytes, 0x00020000);
    // rows past out_f are past the buffer: the range check returns 0 (the whole offset is per-lane); columns past in_f are sent out of range
    unsigned voff_w = col_ok ? (unsigned)(((r0 + lh) * in_f + col) * 4) : 0x80000000u;
    unsigned voff_a = (unsigned)(((r0 + lh) * MP + li) * 4);
    const unsigned step_w = (unsigned)(2 * in_f * 4), step_a = 2 * MP * 4;

    f32x16 acc[FM][4];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[fm][j][e] = 0.0f;

    f32x4 bw[FC_D], bp[MASKED ? FC_D : 1];
    float a[FC_D][FM];
    auto load = [&](int d) {
        bw[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_w, (int)voff_w, 0, 0));
        if constexpr (MASKED) bp[d] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_p, (int)voff_w, 0, 0));
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
            a[d][fm] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_a, (int)voff_a + fm * 128, 0, 0));
        voff_w += step_w, voff_a += step_a;
    };
    auto consume = [&](int d) {
        f32x4 b = bw[d];
        if constexpr (MASKED) {
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] *= binarize(bp[d][q], thr);
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[fm][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d][fm], b[j], acc[fm][j], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < FC_D; ++d) load(d);
    for (int r = 1; r < rounds; ++r) {
#pragma unroll
        for (int d = 0; d < FC_D; ++d) {
            consume(d);
            load(d);
        }
    }
#pragma unroll
    for (int d = 0; d < FC_D; ++d) consume(d);

    // the block's four row ranges, added in wave order (two column phases per pass through LDS); wave v stores accumulator elements
    // 4 v .. 4 v + 3 = rows 8 v + (0..3) + 4 lh
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
        f32x4 v[4];
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            if (fm || jp) __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[((wave * 2 + j) * 16 + e) * 64 + lane] = acc[fm][2 * jp + j][e];
            __syncthreads();
#pragma unroll
            for (int ee = 0; ee < 4; ++ee)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float s = red[((0 * 2 + j) * 16 + 4 * wave + ee) * 64 + lane];
#pragma unroll
                    for (int v2 = 1; v2 < 4; ++v2) s += red[((v2 * 2 + j) * 16 + 4 * wave + ee) * 64 + lane];
                    v[ee][2 * jp + j] = s;
                }
        }
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int row = fm * 32 + 8 * wave + ee + 4 * lh;
            if (row < batch && col_ok)
                *reinterpret_cast<f32x4 *>(out + ((int64_t)split * batch + row) * in_f + col) = v[ee];
        }
    }
}

struct FcPlan {
    int tiles, nsplit, units_per_wave, rows_pad, fm;
    size_t a_bytes, ws_bytes;
};
FcPlan fc_plan(int batch, int in_f, int out_f) {
    FcPlan p;
    p.fm = batch <= 32 ? 1 : 2;
    p.tiles = (in_f + FC_COLS - 1) / FC_COLS;
    const int units = (out_f + FC_UNIT - 1) / FC_UNIT;
    int want = std::max(1, 2 * kCUs / p.tiles);                  // two resident blocks per CU, all column tiles of a split at once
    const int forced = opt_or(OPT_FC_SMALL, 1) & 255;
    if (forced > 1) want = forced;                               // (development: CPG_FC_SMALL=n forces n row splits; bit 8: pointwise.hip)
    want = std::max(1, std::min(want, units / 8));               // >= 2 units per wave
    p.units_per_wave = (units + want * 4 - 1) / (want * 4);
    p.units_per_wave = (p.units_per_wave + 1) / 2 * 2;            // whole rounds at every pipeline depth (D <= 16: 32 rows)
    p.nsplit = (units + p.units_per_wave * 4 - 1) / (p.units_per_wave * 4);
    p.rows_pad = p.nsplit * 4 * p.units_per_wave * FC_UNIT;
    p.a_bytes = ((size_t)p.rows_pad * 32 * p.fm * sizeof(float) + 255) / 256 * 256;
    p.ws_bytes = p.a_bytes + (p.nsplit > 1 ? (size_t)p.nsplit * batch * in_f * sizeof(float) : 0);
    return p;
}

}  // namespace

// (igemm_conv.hip routes cpg_linear_dgrad here)
bool cpg_fc_small_dgrad_ok(const float *w, const float *pm, const float *gx, int batch, int in_f, int out_f) {
    if (opt_or(OPT_FC_SMALL, 1) == 0 || batch > 64 || in_f % 4 != 0) return false;
    if ((((uintptr_t)w) & 15) != 0 || (((uintptr_t)pm) & 15) != 0 || (((uintptr_t)gx) & 15) != 0) return false;
    const FcPlan p = fc_plan(batch, in_f, out_f);
    return (int64_t)(p.rows_pad + 2) * in_f * 4 < (1ll << 31) && (int64_t)p.nsplit * p.tiles < (1ll << 30);
}
size_t cpg_fc_small_dgrad_workspace(int batch, int in_f, int out_f) { return batch <= 64 ? fc_plan(batch, in_f, out_f).ws_bytes : 0; }
int cpg_fc_small_dgrad(const float *gy, const float *w, const float *pm, float thr, float *gx, int batch, int in_f, int out_f, void *ws,
                       size_t ws_bytes, hipStream_t stream, const char *what) {
    const FcPlan p = fc_plan(batch, in_f, out_f);
    if (ws == nullptr || ws_bytes < p.ws_bytes || (((uintptr_t)ws) & 15) != 0)
        return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes (or not 16-byte aligned)", what, ws_bytes, p.ws_bytes);
    float *gyT = (float *)ws, *part = (float *)((char *)ws + p.a_bytes);
    const int MP = 32 * p.fm;
    hipLaunchKernelGGL(k_fc_pack_t, dim3(stream_grid((int64_t)p.rows_pad * MP, 256)), dim3(256), 0, stream, gy, gyT, batch, out_f, p.rows_pad, MP);
    float *dst = p.nsplit > 1 ? part : gx;
    const unsigned w_bytes = (unsigned)((int64_t)out_f * in_f * 4), a_bytes = (unsigned)((int64_t)p.rows_pad * MP * 4);
    const dim3 grid((unsigned)(p.tiles * p.nsplit));
#define FC_LAUNCH(FM_, MASKED_, D_, NB_)                                                                                                 \
    hipLaunchKernelGGL((k_fc_dgrad_small<FM_, MASKED_, D_, NB_>), grid, dim3(256), 0, stream, w, pm, thr, gyT, dst, batch, in_f, p.tiles, \
                       p.units_per_wave, w_bytes, a_bytes)
    if (p.fm == 2) {                                      // two row fragments: 128 accumulator registers, two blocks per CU
        if (pm) FC_LAUNCH(2, true, 8, 2); else FC_LAUNCH(2, false, 8, 2);
    } else {
#define FC_PICK(D_, NB_)                                                            \
    do {                                                                            \
        if (pm) FC_LAUNCH(1, true, D_, NB_); else FC_LAUNCH(1, false, D_, NB_);     \
    } while (0)
        switch ((opt_or(OPT_FC_SMALL, 1) >> 9) & 3) {     // (development: bits 9-10 pick the pipeline depth / resident blocks)
            case 1: FC_PICK(16, 2); break;
            case 2: FC_PICK(4, 4); break;
            default: FC_PICK(8, 2); break;                // (the register count allows 3 - 4 resident blocks)
        }
    }
#undef FC_PICK
#undef FC_LAUNCH
    if (p.nsplit > 1) {
        Epilogue ep{gx, nullptr, BIAS_NONE, 1, 1, nullptr, nullptr, nullptr, thr};
        launch_split_reduce(part, p.nsplit, (int64_t)batch * in_f, 0, ep, stream);
    }
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}
