// Masked 3x3 / stride 1 / pad 1 convolution of an image-like input (<= 3 channels) to 64 (.. 96, round 5) channels: the network stem
// (VGG16 features.0, models/vgg.py:131-141 of the reference; SharableConv2d.forward, models/layers.py:98-109).
//
// The layer is HBM-bound: 27 multiplies per output against 4 bytes written -- batch 256 @ 224 x 224 writes 3.29 GB and reads 0.15 GB,
// 0.55 ms at the 6.3 TB/s a float4 copy reaches on this part; the MFMA work is 0.3 ms.  The general direct kernel
// (conv3x3.hip, 8 x 32 tile) took 1.23 ms: one block per tile, so each tile pays a prologue (operand staging, a barrier), an
// epilogue of 64 dword stores per lane and the dispatch of the next block, with only 72 MFMAs per wave in between.
//
// Here ONE WAVE = ONE TILE (8 rows x 32 columns x all output channels) at a time, a dozen tiles per wave at batch 256, no barriers:
//   * k of the MFMA runs over (channel, tap): 27 values padded to 28 = 14 steps of v_mfma_f32_32x32x2_f32.  The A operands
//     (W .* bin(pm), 2 blocks of 32 output channels x 14 steps) stay in 28 registers for the whole launch -- no weight pack kernel;
//   * the B operand of step t and output row j is one ds_read_b32 from the wave's private patch (3 channels x 10 rows x 34 columns):
//     per-lane base address of (channel, tap) + an immediate row offset;
//   * the next tile's patch is requested before the current tile's MFMAs (16 loads per tile, range-checked to zero outside the image);
//   * per output row: 28 MFMAs, 32 stores of 2 x 128 contiguous bytes; the BatchNorm statistics (STATS) are summed per lane over the
//     tile's 8 rows and reduced once per tile with DPP adds -> stats[k][tile][2], the layout of the other forward kernels.
//
// Round 3 -- the stem FUSED with the training-mode BatchNorm2d -> ReLU behind it (models/vgg.py:137-141: conv, BatchNorm2d, ReLU): the
// conv output y (3.3 GB at batch 256) is never written.  Every pass that needs y recomputes it from the image -- 27 multiply-adds
// per output against 4 bytes of HBM traffic saved per pass -- with this kernel's tile loop and a different epilogue (MODE):
//   ST_STATS_ONLY   forward pass A: the statistics partial sums of y, no store                   (reads x: 0.15 GB)
//   ST_BN_RELU      forward pass B: z = relu(bn(y)) stored                                       (writes z: 3.3 GB)
//   ST_BWD_REDUCE   backward: partial sums {sum gm, sum gm xhat}, gm = gz [z > 0]                 (reads gz: 3.3 GB)
//   ST_BWD_APPLY    backward: gy = (gm - mean(gm) - xhat mean(gm xhat)) invstd gamma stored       (reads gz, writes gy)
//   ST_BWD_WGRAD    backward: the same gy, not stored but contracted with the image patch right away -- the stem's weight gradient
//                   gW[co][(c, r, s)] = sum over pixels of gy[co][pix] x[c][pix + (r, s) - 1] as a second MFMA (32 channels x 32 taps,
//                   k = pixel pairs of a tile row), gy transposed through LDS; per-wave partial sums, k_split_reduce (reads gz only)
// against conv (write y) + BN apply (read y, write z) + BN backward reduce (read y, gz) + apply (read y, gz, write gy): 13.2 GB of
// y traffic per step gone.  The arithmetic per element is the unfused kernels' (bn_affine of bn_kernels.hip), so results agree to
// the last bit wherever the summation order is the same.
#include <algorithm>
#include <type_traits>
#include "igemm_core.h"

using namespace cpg;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ST_R = 8, ST_W = 32;              // output rows / columns of a tile
constexpr int ST_PW = 36;                       // row stride of the patch in LDS (34 used)
constexpr int ST_ROWS = ST_R + 2;
constexpr int ST_CMAX = 3, ST_KS = 14;          // channels, MFMA steps (2 k each)
constexpr int ST_PATCH = ST_CMAX * ST_ROWS * ST_PW;

struct StemGeom {
    int N, C, K, H, W;
    int tiles_x, tiles_y;
    unsigned ntiles;
};

enum { ST_PLAIN = 0, ST_STATS = 1, ST_STATS_ONLY = 2, ST_BN_RELU = 3, ST_BWD_REDUCE = 4, ST_BWD_APPLY = 5, ST_BWD_WGRAD = 6 };
constexpr int ST_TS = 36;                       // row stride of the gy transpose tile (ST_BWD_WGRAD): [channel 32][pixel parity 2][16], 16-byte rows
struct StemBn {                                 // the BatchNorm behind the stem (fused modes): per-channel arrays of K floats
    const float *gamma, *beta, *mean, *invstd;
    const float *coef;                          // ST_BWD_APPLY: {mean(gm), mean(gm xhat)} per channel
    const float *gz;                            // ST_BWD_*: gradient w.r.t. z = relu(bn(y)), laid out like y
};

// NB / RAG (round 5): blocks of 32 output channels and whether the last one is partial.  <.., 2, false> is the 64-channel stem of the
// reference's width-1.0 networks, unchanged; <.., 3, true> takes 65 .. 96 channels -- the GROWN VGG16's features.0 has int(64 sqrt(1.5)) = 78
// (models/vgg.py:131-136 with CPG_cifar100_main_normal.py:115) and ran the general direct kernel + the unfused BatchNorm passes until
// round 5.  A channel past K has zero weights (its accumulator is an exact 0) and a per-lane offset that is out of range: its stores are
// dropped, its loads read 0 (only the per-lane part of a buffer offset is range-checked, the scalar channel offset is not).
template <int MODE, int NB = 2, bool RAG = false>
__global__ __launch_bounds__(256, 2)
void k_stem_fwd(StemGeom g, const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ pm, float thr,
                const float *__restrict__ bias, float *__restrict__ y, float *__restrict__ stats, StemBn bn) {
    constexpr bool STATS = MODE == ST_STATS || MODE == ST_STATS_ONLY || MODE == ST_BWD_REDUCE;     // two sums per channel and tile
    constexpr bool BN = MODE >= ST_BN_RELU;                                                        // needs the BatchNorm's constants
    constexpr bool STORE = MODE == ST_PLAIN || MODE == ST_STATS || MODE == ST_BN_RELU || MODE == ST_BWD_APPLY;
    constexpr bool WG = MODE == ST_BWD_WGRAD;   // y = the per-wave partial weight gradients [wave][64][C 9]
    constexpr bool TWO_PASS = STATS || BN;      // the two blocks of 32 output channels in two passes over the patch (register budget)
    __shared__ float smem_all[4 * ST_PATCH];
    // backward modes: the BatchNorm's per-channel constants {mean, invstd, gamma, beta, mean(gm), mean(gm xhat)} live in LDS (64 x 8
    // floats) and are read per accumulator element -- as registers (16 channels x 6 per lane) they pushed the kernel into scratch
    __shared__ __attribute__((aligned(16))) float cst[(MODE >= ST_BWD_REDUCE) ? NB * 32 * 8 : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (MODE >= ST_BWD_REDUCE) {
        if (tid < NB * 32) {
            const bool kv = tid < g.K;
            cst[tid * 8 + 0] = kv ? bn.mean[tid] : 0.0f, cst[tid * 8 + 1] = kv ? bn.invstd[tid] : 0.0f;
            cst[tid * 8 + 2] = kv ? bn.gamma[tid] : 0.0f, cst[tid * 8 + 3] = kv ? bn.beta[tid] : 0.0f;
            cst[tid * 8 + 4] = (MODE >= ST_BWD_APPLY && kv) ? bn.coef[2 * tid] : 0.0f;
            cst[tid * 8 + 5] = (MODE >= ST_BWD_APPLY && kv) ? bn.coef[2 * tid + 1] : 0.0f;
        }
        __syncthreads();                         // (the only barrier: before any wave leaves)
    }
    const int li = lane & 31, lh = lane >> 5;
    float *smem = smem_all + wave * ST_PATCH;
    const int HW = g.H * g.W, CK = g.C * 9;
    __shared__ __attribute__((aligned(16))) float tsm_all[WG ? 4 * 32 * ST_TS : 4];
    float *tsm = tsm_all + (WG ? wave * 32 * ST_TS : 0);
    // ST_BWD_WGRAD: tap k = li of the weight gradient's B operand: patch offset of (c, r, s) + the half-wave's pixel parity
    const int wk = li < CK ? li : 0, wboff = ((wk / 9) * ST_ROWS + (wk % 9) / 3) * ST_PW + wk % 3 + lh;
    f32x16 accw[WG ? NB : 1];
    if constexpr (WG) {
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) accw[mb][e] = 0.0f;
    }

    // A operands: W_eff[co = 32 mb + li][k = 2 t + lh], zero beyond the layer's channels / taps
    // ALDS (three blocks in the two-pass modes): 42 weight registers beside the BatchNorm constants of a pass spilled (up to 301 dwords of
    // scratch in the fused forward); the block keeps the operands in LDS instead ([block][step][lane], 10.5 KB, written once) and a pass
    // reads its 14 back.
    constexpr bool ALDS = NB == 3 && TWO_PASS;
    __shared__ float a_lds[ALDS ? NB * ST_KS * 64 : 1];
    float A[ALDS ? 1 : NB][ST_KS];
    int boff[ST_KS];                             // B operand: float index of (channel, tap) of k = 2 t + lh in the patch, + li
#pragma unroll
    for (int t = 0; t < ST_KS; ++t) {
        const int k = 2 * t + lh;
        const bool kv = k < CK;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
            const int co = mb * 32 + li;
            float v = 0.0f;
            if (kv && co < g.K) {
                v = w[co * CK + k];
                if (pm != nullptr) v *= binarize(pm[co * CK + k], thr);
            }
            if constexpr (ALDS) {
                if (wave == 0) a_lds[(mb * ST_KS + t) * 64 + lane] = v;
            } else {
                A[mb][t] = v;
            }
        }
        const int kk = kv ? k : 0;               // (padding taps: a zero weight against any finite patch element)
        const int c = kk / 9, r = (kk % 9) / 3, s = kk % 3;
        boff[t] = (c * ST_ROWS + r) * ST_PW + s + li;
    }
    constexpr bool HASB = !BN;                  // (the fused BatchNorm modes take no conv bias -- the host refuses one: 32 registers)
    constexpr bool BPASS = HASB && TWO_PASS && NB == 3;     // three blocks, two-pass modes: the bias of ONE block at a time (48 registers spilled)
    float bv[HASB ? (BPASS ? 1 : NB) : 1][HASB ? 16 : 1];
    if constexpr (HASB && !BPASS) {
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                bv[mb][e] = (bias != nullptr && co < g.K) ? bias[co] : 0.0f;
            }
    }

    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, g.N * g.C * HW * 4, 0x00020000);
    constexpr int kOOR = (int)0x80000000;
    // staging items: item = channel * 10 + patch row; centre columns: two items per load (half-waves), 15 loads; halo columns: one load
    const int hw_half = lane >> 5, hcol = lane & 31;
    float pc[15], ph;
    auto tile_coords = [&](unsigned tile, int &n, int &y0, int &x0) {
        const unsigned per_img = (unsigned)(g.tiles_x * g.tiles_y);
        n = (int)(tile / per_img);
        const unsigned r = tile % per_img;
        y0 = (int)(r / (unsigned)g.tiles_x) * ST_R, x0 = (int)(r % (unsigned)g.tiles_x) * ST_W;
    };
    auto issue_loads = [&](unsigned tile) {
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
#pragma unroll
        for (int q = 0; q < 15; ++q) {
            const int item = 2 * q + hw_half, c = item / ST_ROWS, r = item % ST_ROWS;
            const int row = y0 - 1 + r, col = x0 + hcol;
            const bool ok = c < g.C && (unsigned)row < (unsigned)g.H && col < g.W;
            const int off = ok ? (((n * g.C + c) * g.H + row) * g.W + col) * 4 : kOOR;
            pc[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, off, 0, 0));
        }
        {
            const int item = lane >> 1, side = lane & 1, c = item / ST_ROWS, r = item % ST_ROWS;
            const int row = y0 - 1 + r, col = side ? x0 + ST_W : x0 - 1;
            const bool ok = item < ST_CMAX * ST_ROWS && c < g.C && (unsigned)row < (unsigned)g.H && (unsigned)col < (unsigned)g.W;
            const int off = ok ? (((n * g.C + c) * g.H + row) * g.W + col) * 4 : kOOR;
            ph = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, off, 0, 0));
        }
    };
    auto write_patch = [&]() {
#pragma unroll
        for (int q = 0; q < 15; ++q) smem[(2 * q + hw_half) * ST_PW + 1 + hcol] = pc[q];
        if (lane < 2 * ST_CMAX * ST_ROWS) smem[(lane >> 1) * ST_PW + ((lane & 1) ? ST_W + 1 : 0)] = ph;
    };

    if constexpr (ALDS) __syncthreads();          // (before any wave leaves)
    const unsigned nwaves = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
    if (wid >= g.ntiles) return;                 // (no barriers anywhere: a wave may leave)
    const int HW4 = HW * 4;
    issue_loads(wid);
    for (unsigned tile = wid; tile < g.ntiles; tile += nwaves) {
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
        write_patch();
        if (tile + nwaves < g.ntiles) issue_loads(tile + nwaves);
        float s1[16], s2[16];                      // (one block of 32 channels at a time: STATS modes run two passes)
        auto zero_sums = [&]() {
            if (STATS) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s1[e] = s2[e] = 0.0f;
            }
        };
        // y of this image through a buffer descriptor: the lane part of a store's address (pixel, + 4 channels for the upper
        // half-wave) is a 32-bit offset that is out of range for pixels outside the image (the store is dropped), the uniform
        // part of the channel is the instruction's scalar offset (K = 64: every channel of the two blocks exists)
        const __amdgpu_buffer_rsrc_t srd_y = __builtin_amdgcn_make_buffer_rsrc((void *)((STORE ? y : (float *)x) + (STORE ? (int64_t)n * g.K * HW : 0)), 0, STORE ? g.K * HW4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t srd_gz = __builtin_amdgcn_make_buffer_rsrc((void *)(MODE >= ST_BWD_REDUCE ? bn.gz + (int64_t)n * g.K * HW : x), 0, MODE >= ST_BWD_REDUCE ? g.K * HW4 : 0, 0x00020000);
        const bool cok = x0 + li < g.W;
        const int pix0 = (y0 * g.W + x0 + li) * 4 + lh * 4 * HW4;
        // (a tile inside the image needs no per-pixel masks: `full` is wave-uniform)
        const bool full = x0 + ST_W <= g.W && y0 + ST_R <= g.H;
        // STATS: the two blocks of 32 output channels in two passes over the patch (the 64 sums of one pass would cost the second
        // resident block its registers: 0.81 instead of 0.68 ms); plain: both blocks share every B operand read
        // fused modes: the BatchNorm's constants of this pass's 16 channels per lane (channel mb 32 + (e & 3) + 8 (e >> 2) + 4 lh)
        constexpr bool REGC = MODE == ST_BN_RELU;
        float cm[REGC ? 16 : 1], cis[REGC ? 16 : 1], cga[REGC ? 16 : 1], cbe[REGC ? 16 : 1];
        // (RAG + REGC: the BatchNorm's four per-channel arrays behind buffer descriptors of K floats: beta, mean, invstd, gamma)
        __amdgpu_buffer_rsrc_t srd_bn[4];
        if constexpr (RAG && REGC) {
            srd_bn[0] = __builtin_amdgcn_make_buffer_rsrc((void *)bn.beta, 0, g.K * 4, 0x00020000);
            srd_bn[1] = __builtin_amdgcn_make_buffer_rsrc((void *)bn.mean, 0, g.K * 4, 0x00020000);
            srd_bn[2] = __builtin_amdgcn_make_buffer_rsrc((void *)bn.invstd, 0, g.K * 4, 0x00020000);
            srd_bn[3] = __builtin_amdgcn_make_buffer_rsrc((void *)bn.gamma, 0, g.K * 4, 0x00020000);
        }
        auto load_bn = [&](int mb) {
            if constexpr (BPASS) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    bv[0][e] = (bias != nullptr && co < g.K) ? bias[co] : 0.0f;
                }
            }
            if constexpr (ALDS) {
#pragma unroll
                for (int t = 0; t < ST_KS; ++t) A[0][t] = a_lds[(mb * ST_KS + t) * 64 + lane];
            }
            if constexpr (REGC) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if constexpr (RAG) {
                        // range-checked loads (a channel past K reads 0): one per-lane offset + a scalar offset -- a clamped index per element
                        // made 64 address pairs and 273 dwords of scratch
                        const int vo = (4 * lh + (e & 3) + 8 * (e >> 2)) * 4 + mb * 128;
                        cm[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_bn[1], vo, 0, 0));
                        cis[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_bn[2], vo, 0, 0));
                        cga[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_bn[3], vo, 0, 0));
                        cbe[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_bn[0], vo, 0, 0));
                    } else {
                        const int co = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        cm[e] = bn.mean[co], cis[e] = bn.invstd[co], cga[e] = bn.gamma[co], cbe[e] = bn.beta[co];
                    }
                }
            }
        };
        auto row = [&](int j, auto fullc, auto mbsel) {
            constexpr bool FULL = decltype(fullc)::value;
            constexpr int MB0 = decltype(mbsel)::value < 0 ? 0 : decltype(mbsel)::value, MB1 = decltype(mbsel)::value < 0 ? NB : MB0 + 1;
            const bool pok = FULL || (cok && y0 + j < g.H);
            const int voff = pok ? pix0 + j * g.W * 4 : kOOR;
            // RAG: the per-lane offset of output e of block mb -- out of range for a channel past K (see the kernel's header)
            // (only the LAST block can be partial: K >= 64.  The channel count goes through an opaque scalar per row: as a loop invariant the
            //  sixteen masked offsets were hoisted out of the row loop and spilled -- 273 dwords of scratch in the fused forward)
            int klim = g.K;
            if constexpr (RAG) asm volatile("" : "+s"(klim));
            auto voff_of = [&](int mb, int e) -> int {
                if constexpr (RAG) {
                    if (mb == NB - 1) return ((NB - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh < klim) ? voff : kOOR;
                }
                return voff;
            };
            float gzv[MODE >= ST_BWD_REDUCE ? 16 : 1];
            int copq = 0;
            if constexpr (MODE >= ST_BWD_REDUCE) asm volatile("" : "+v"(copq));      // (hoisted out of the row loop the constants are 96 registers again)
            if constexpr (MODE >= ST_BWD_REDUCE) {           // the row's 16 gradient values fly while its 14 MFMAs run (out of range: 0)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    gzv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_gz, voff_of(MB0, e), (MB0 * 32 + (e & 3) + 8 * (e >> 2)) * HW4, 0));
            }
            f32x16 acc[NB];
#pragma unroll
            for (int mb = MB0; mb < MB1; ++mb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mb][e] = 0.0f;
            const float *prow = smem + j * ST_PW;
#pragma unroll
            for (int t = 0; t < ST_KS; ++t) {
                const float b = prow[boff[t]];
#pragma unroll
                for (int mb = MB0; mb < MB1; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[ALDS ? 0 : mb][t], b, acc[mb], 0, 0, 0);
            }
#pragma unroll
            for (int mb = MB0; mb < MB1; ++mb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int cu = mb * 32 + (e & 3) + 8 * (e >> 2);       // + 4 lh: in voff
                    const float v = HASB ? acc[mb][e] + bv[(HASB && !BPASS) ? mb : 0][HASB ? e : 0] : acc[mb][e];
                    if constexpr (MODE == ST_PLAIN || MODE == ST_STATS) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), srd_y, voff_of(mb, e), cu * HW4, 0);
                    }
                    if constexpr (MODE == ST_STATS || MODE == ST_STATS_ONLY) {
                        const float vm = (FULL || pok) ? v : 0.0f;
                        s1[e] += vm;
                        s2[e] = fmaf(vm, vm, s2[e]);
                    }
                    if constexpr (MODE == ST_BN_RELU) {                    // bn_affine of bn_kernels.hip, then the ReLU
                        const float z = fmaxf((v - cm[e]) * cis[e] * cga[e] + cbe[e], 0.0f);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, z), srd_y, voff_of(mb, e), cu * HW4, 0);
                    }
                    if constexpr (MODE >= ST_BWD_REDUCE) {
                        const float *cc = cst + (cu + 4 * lh) * 8 + copq;     // (copq = 0, opaque per row: keeps the reads in the row)
                        const f32x4 c4 = *reinterpret_cast<const f32x4 *>(cc);                  // mean, invstd, gamma, beta
                        const float xh = (v - c4[0]) * c4[1];
                        const float gm = ((v - c4[0]) * c4[1] * c4[2] + c4[3] > 0.0f) ? gzv[e] : 0.0f;        // (gz of a pixel outside the image reads 0)
                        if constexpr (MODE == ST_BWD_REDUCE) {
                            s1[e] += gm;
                            s2[e] = fmaf(gm, xh, s2[e]);
                        } else {
                            const f32x2 c2 = *reinterpret_cast<const f32x2 *>(cc + 4);           // mean(gm), mean(gm xhat)
                            float gyv = (gm - c2[0] - xh * c2[1]) * (c4[1] * c4[2]);
                            if constexpr (MODE == ST_BWD_APPLY) {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gyv), srd_y, voff_of(mb, e), cu * HW4, 0);
                            } else {                                   // a pixel outside the image has no gradient; [channel][parity][pixel / 2]
                                if (!FULL && !pok) gyv = 0.0f;
                                tsm[((e & 3) + 8 * (e >> 2) + 4 * lh) * ST_TS + (li & 1) * 16 + (li >> 1)] = gyv;
                            }
                        }
                    }
                }
            if constexpr (WG) {
                // gW[co = li of block MB0][tap li] += sum over the row's 32 pixels: A = gy[co][pixel 2 t + lh] (four 16-byte reads of the
                // lane's channel row, its half-wave's parity), B = x[c][j + r][2 t + lh + s] (the patch at the tap's offset)
                f32x4 ga[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ga[q] = *reinterpret_cast<const f32x4 *>(tsm + li * ST_TS + lh * 16 + 4 * q);
                const float *brow = smem + j * ST_PW + wboff;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    accw[MB0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[t >> 2][t & 3], brow[2 * t], accw[MB0], 0, 0, 0);
            }
        };
        auto rows = [&](auto mbsel) {
            if (full) {
#pragma unroll 1
                for (int j = 0; j < ST_R; ++j) row(j, std::true_type{}, mbsel);
            } else {
#pragma unroll 1
                for (int j = 0; j < ST_R; ++j) row(j, std::false_type{}, mbsel);
            }
        };
        auto stats_out = [&](int mb) {
#pragma unroll
            for (int e = 0; e < 16; e += 8) half_wave_sum8(s1 + e), half_wave_sum8(s2 + e);
            if (li == kHalfSumLane) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (RAG && co >= g.K) continue;
                    f32x2 o;
                    o[0] = s1[e], o[1] = s2[e];
                    *reinterpret_cast<f32x2 *>(stats + ((int64_t)co * g.ntiles + tile) * 2) = o;
                }
            }
        };
        if (TWO_PASS) {
            load_bn(0);
            zero_sums();
            rows(std::integral_constant<int, 0>{});
            if (STATS) stats_out(0);
            load_bn(1);
            zero_sums();
            rows(std::integral_constant<int, 1>{});
            if (STATS) stats_out(1);
            if constexpr (NB == 3) {
                load_bn(2);
                zero_sums();
                rows(std::integral_constant<int, 2>{});
                if (STATS) stats_out(2);
            }
        } else {
            rows(std::integral_constant<int, -1>{});
        }
    }
    if constexpr (WG) {                          // this wave's partial sums: y[wid][co][k], D row = channel, D column (lane) = tap
        float *dstw = y + (int64_t)wid * g.K * CK;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (li < CK && co < g.K) dstw[co * CK + li] = accw[mb][e];
            }
    }
}

// (the 64-channel instance, or the three-block instance with the partial-block masks)
#define STEM_LAUNCH(MODE, grid, block, shm, stream, ...)                                                          \
    do {                                                                                                          \
        if (g.K == 64) hipLaunchKernelGGL((k_stem_fwd<MODE, 2, false>), grid, block, shm, stream, __VA_ARGS__);   \
        else hipLaunchKernelGGL((k_stem_fwd<MODE, 3, true>), grid, block, shm, stream, __VA_ARGS__);              \
    } while (0)

bool stem_geom(int N, int C, int K, int H, int W, StemGeom &g) {
    if (opt_on(OPT_NO_STEM)) return false;                       // (A/B experiments, tests)
    if (N < 1 || C < 1 || C > ST_CMAX || K < 64 || K > 96 || H < 1 || W < 1) return false;      // (64: <.., 2, false>; 65 .. 96: <.., 3, true>)
    if ((int64_t)N * C * H * W * 4 >= (1ll << 31) || (int64_t)K * H * W * 4 >= (1ll << 31)) return false;   // (32-bit byte offsets into x and into one image of y)
    g.N = N, g.C = C, g.K = K, g.H = H, g.W = W;
    g.tiles_x = (W + ST_W - 1) / ST_W, g.tiles_y = (H + ST_R - 1) / ST_R;
    const int64_t nt = (int64_t)N * g.tiles_x * g.tiles_y;
    if (nt >= (1ll << 31)) return false;
    g.ntiles = (unsigned)nt;
    return true;
}

}  // namespace

// 1: cpg_conv2d_fwd / cpg_conv2d_fwd_bnstats run this layer on the stem kernel (CPG_NO_STEM in the environment: never)
extern "C" int cpg_conv3x3_stem_ok(int N, int C, int K, int H, int W) {
    StemGeom g;
    return stem_geom(N, C, K, H, W, g) ? 1 : 0;
}

// BatchNorm-statistics tiles per channel (stats[K][tiles][2])
extern "C" int cpg_conv3x3_stem_tiles(int N, int C, int K, int H, int W) {
    StemGeom g;
    return stem_geom(N, C, K, H, W, g) ? (int)g.ntiles : 0;
}

extern "C" int cpg_conv3x3_stem_run(int N, int C, int K, int H, int W, const float *x, const float *w, const float *pm, float thr,
                                    const float *bias, float *y, float *stats, hipStream_t stream) {
    StemGeom g;
    if (!stem_geom(N, C, K, H, W, g)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_fwd(stem): shape not supported");
    // Two resident grids' worth of blocks (4 per CU; a CU holds 2): measured the same as exactly one grid (0.776 vs 0.778 ms,
    // 12 vs 24 tiles per wave), 3-10 % faster than 3, 6 or 12 grids (the weights are re-fetched by every wave) -- and when another
    // stream (RCCL) holds some CUs, an exactly-resident grid would have to wait for its last blocks.  CPG_STEM_BLOCKS overrides (A/B).
    unsigned blocks = (unsigned)std::min<int64_t>(((int64_t)g.ntiles + 3) / 4, 4 * kCUs);
    blocks = (unsigned)std::max(1, opt_or(OPT_STEM_BLOCKS, (int)blocks));
    if (stats != nullptr)
        STEM_LAUNCH(ST_STATS, dim3(blocks), dim3(256), 0, stream, g, x, w, pm, thr, bias, y, stats, StemBn{});
    else
        STEM_LAUNCH(ST_PLAIN, dim3(blocks), dim3(256), 0, stream, g, x, w, pm, thr, bias, y, nullptr, StemBn{});
    CPG_CHECK_LAUNCH("cpg_conv2d_fwd(stem)");
    return CPG_OK;
}

// ------------------------------------------------------------------------------ the stem fused with BatchNorm2d -> ReLU (see the header)
namespace {
bool stem_bn_geom(const cpg_conv_desc *d, StemGeom &g) {
    if (opt_on(OPT_NO_STEM_FUSE)) return false;
    if (!(d->R == 3 && d->S == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1 && d->dil_h == 1 &&
          d->dil_w == 1 && d->groups == 1))
        return false;
    return stem_geom(d->N, d->C, d->K, d->H, d->W, g);
}
unsigned stem_blocks(const StemGeom &g) {
    unsigned blocks = (unsigned)std::min<int64_t>(((int64_t)g.ntiles + 3) / 4, 4 * kCUs);
    blocks = (unsigned)std::max(1, opt_or(OPT_STEM_BLOCKS, (int)blocks));
    return blocks;
}
}  // namespace

extern "C" int32_t cpg_stem_bn_supported(const cpg_conv_desc *d) {
    StemGeom g;
    return (d != nullptr && stem_bn_geom(d, g)) ? 1 : 0;
}
extern "C" int32_t cpg_stem_bn_tiles(const cpg_conv_desc *d) {
    StemGeom g;
    return (d != nullptr && stem_bn_geom(d, g)) ? (int32_t)g.ntiles : 0;
}
// pass A: stats[K][tiles][2] = {sum y, sum y^2} per tile of y = conv(x) (+ bias); y is not written
extern "C" int cpg_stem_bn_stats(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                                 float *stats, size_t stats_bytes, void *stream_v) {
    StemGeom g;
    CPG_REQUIRE(d && x && w && stats, "cpg_stem_bn_stats: null pointer");
    if (!stem_bn_geom(d, g)) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_stats: shape not supported");
    if (bias != nullptr) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_stats: the fused stem takes no conv bias");
    if (stats_bytes < (size_t)g.K * g.ntiles * 2 * sizeof(float)) return fail(CPG_E_WORKSPACE, "cpg_stem_bn_stats: statistics buffer too small");
    STEM_LAUNCH(ST_STATS_ONLY, dim3(stem_blocks(g)), dim3(256), 0, (hipStream_t)stream_v, g, x, w, pm, thr, bias,
                       (float *)nullptr, stats, StemBn{});
    CPG_CHECK_LAUNCH("cpg_stem_bn_stats");
    return CPG_OK;
}
// pass B: z = relu((y - mean) invstd gamma + beta), y recomputed
extern "C" int cpg_stem_bn_relu_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                                    const float *gamma, const float *beta, const float *mean, const float *invstd, float *z, void *stream_v) {
    StemGeom g;
    CPG_REQUIRE(d && x && w && gamma && beta && mean && invstd && z, "cpg_stem_bn_relu_fwd: null pointer");
    if (!stem_bn_geom(d, g)) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_relu_fwd: shape not supported");
    if (bias != nullptr) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_relu_fwd: the fused stem takes no conv bias");
    STEM_LAUNCH(ST_BN_RELU, dim3(stem_blocks(g)), dim3(256), 0, (hipStream_t)stream_v, g, x, w, pm, thr, bias, z,
                       (float *)nullptr, StemBn{gamma, beta, mean, invstd, nullptr, nullptr});
    CPG_CHECK_LAUNCH("cpg_stem_bn_relu_fwd");
    return CPG_OK;
}
// backward 1: partials[K][tiles][2] = {sum gm, sum gm xhat}, gm = gz [z > 0] (cpg_bn_bwd_finalize_partials merges them)
extern "C" int cpg_stem_bn_relu_bwd_reduce(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr,
                                           const float *bias, const float *gamma, const float *beta, const float *mean, const float *invstd,
                                           const float *gz, float *partials, size_t partials_bytes, void *stream_v) {
    StemGeom g;
    CPG_REQUIRE(d && x && w && gamma && beta && mean && invstd && gz && partials, "cpg_stem_bn_relu_bwd_reduce: null pointer");
    if (!stem_bn_geom(d, g)) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_relu_bwd_reduce: shape not supported");
    if (bias != nullptr) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_relu_bwd_reduce: the fused stem takes no conv bias");
    if (partials_bytes < (size_t)g.K * g.ntiles * 2 * sizeof(float)) return fail(CPG_E_WORKSPACE, "cpg_stem_bn_relu_bwd_reduce: partial-sum buffer too small");
    STEM_LAUNCH(ST_BWD_REDUCE, dim3(stem_blocks(g)), dim3(256), 0, (hipStream_t)stream_v, g, x, w, pm, thr, bias,
                       (float *)nullptr, partials, StemBn{gamma, beta, mean, invstd, nullptr, gz});
    CPG_CHECK_LAUNCH("cpg_stem_bn_relu_bwd_reduce");
    return CPG_OK;
}
// backward 2: gy = (gm - coef[2c] - xhat coef[2c + 1]) invstd gamma -- the gradient w.r.t. the conv output, for cpg_conv2d_wgrad
extern "C" int cpg_stem_bn_relu_bwd_apply(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr,
                                          const float *bias, const float *gamma, const float *beta, const float *mean, const float *invstd,
                                          const float *coef, const float *gz, float *gy, void *stream_v) {
    StemGeom g;
    CPG_REQUIRE(d && x && w && gamma && beta && mean && invstd && coef && gz && gy, "cpg_stem_bn_relu_bwd_apply: null pointer");
    if (!stem_bn_geom(d, g)) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_relu_bwd_apply: shape not supported");
    if (bias != nullptr) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_relu_bwd_apply: the fused stem takes no conv bias");
    STEM_LAUNCH(ST_BWD_APPLY, dim3(stem_blocks(g)), dim3(256), 0, (hipStream_t)stream_v, g, x, w, pm, thr, bias, gy,
                       (float *)nullptr, StemBn{gamma, beta, mean, invstd, coef, gz});
    CPG_CHECK_LAUNCH("cpg_stem_bn_relu_bwd_apply");
    return CPG_OK;
}

// backward 2 + the stem's weight gradient in one pass: gy is formed per tile row and contracted with the image patch at once (never
// written); per-wave partial sums in the workspace, merged by k_split_reduce with the autograd epilogue gW = g bin(pm), gPM = g W
extern "C" size_t cpg_stem_bn_wgrad_workspace(const cpg_conv_desc *d) {
    StemGeom g;
    if (d == nullptr || !stem_bn_geom(d, g)) return 0;
    return (size_t)stem_blocks(g) * 4 * g.K * g.C * 9 * sizeof(float);
}
extern "C" int cpg_stem_bn_relu_bwd_wgrad(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr,
                                          const float *bias, const float *gamma, const float *beta, const float *mean, const float *invstd,
                                          const float *coef, const float *gz, float *gw, float *gpm, void *ws, size_t ws_bytes,
                                          void *stream_v) {
    StemGeom g;
    CPG_REQUIRE(d && x && w && gamma && beta && mean && invstd && coef && gz && gw && ws, "cpg_stem_bn_relu_bwd_wgrad: null pointer");
    CPG_REQUIRE((pm == nullptr) == (gpm == nullptr), "cpg_stem_bn_relu_bwd_wgrad: piggymask and its gradient come as a pair");
    if (!stem_bn_geom(d, g)) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_relu_bwd_wgrad: shape not supported");
    if (bias != nullptr) return fail(CPG_E_UNSUPPORTED, "cpg_stem_bn_relu_bwd_wgrad: the fused stem takes no conv bias");
    const unsigned blocks = stem_blocks(g);
    if (ws_bytes < (size_t)blocks * 4 * g.K * g.C * 9 * sizeof(float)) return fail(CPG_E_WORKSPACE, "cpg_stem_bn_relu_bwd_wgrad: workspace too small");
    hipStream_t stream = (hipStream_t)stream_v;
    STEM_LAUNCH(ST_BWD_WGRAD, dim3(blocks), dim3(256), 0, stream, g, x, w, pm, thr, bias, (float *)ws, (float *)nullptr,
                       StemBn{gamma, beta, mean, invstd, coef, gz});
    const int nsplit = (int)std::min<int64_t>((int64_t)g.ntiles, (int64_t)blocks * 4);        // waves that had at least one tile
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    launch_split_reduce((const float *)ws, nsplit, (int64_t)g.K * g.C * 9, 0, ep, stream);
    CPG_CHECK_LAUNCH("cpg_stem_bn_relu_bwd_wgrad");
    return CPG_OK;
}
