// Masked 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2, 3x3) on fp32 MFMA: forward and input gradient.
//
//   Y(2x2 tile) = A^T [ (G g G^T) .* (B^T d B) ] A        (d: 4x4 input patch, g: 3x3 filter; Lavin & Gray 2015)
//
// 16 multiplies per 2x2 output tile and (c, k) pair instead of 36: the contraction over input channels becomes 16
// independent GEMMs  M_p[k][t] = sum_c U_p[k][c] * V_p[c][t]  (p = transform-domain position, t = tile), 2.25x fewer MFMAs
// than the direct kernels of conv3x3.hip for the same result up to fp32 rounding (the transforms only add, subtract and
// halve; measured 1-4e-6 of the output scale against fp64, the direct kernels 0.5-1e-6).  docs/LAB_NOTEBOOK.md section 4.9 has the
// measurements behind the choices below.
//
//   k_wg_pack   U = G (W .* bin(piggymask)) G^T per (k, c), written in the order the conv kernel streams it:
//               Up[k block of BK][channel chunk of 4][p][k][c]  (one contiguous record per block and chunk)
//   k_wg_fwd    block = NW waves = BK output channels x 64 tiles (256 output pixels); a tile run is 64 consecutive tiles in
//               (image, tile row, tile column) order, whatever the map size -- no padding tiles, blocks may straddle rows and
//               images.  NW = 4 (the default): BK = 32, two blocks per CU; NW = 8: BK = 64, one block per CU (every
//               transformed input element feeds twice the MFMAs, less staging work per MFMA -- but measured slower, see
//               wino_nw()).  Wave (tq, ph, kq) owns tiles tq*32..+31, positions ph*8..+7 and
//               channels kq*32..+31: 8 accumulators of 32 x 32.  Per chunk of 4 input channels:
//     G  global -> registers.  A vector-memory instruction occupies the texture addresser for 16 cycles whatever its width
//        (gathering every tile's 4 x 4 patch with 16 dword loads made the kernel TA-bound), so the patch rows are fetched
//        ONCE, as aligned column pairs: per channel and patch row one buffer_load_dwordx2 (lane t = columns 2 tx, 2 tx + 1 of
//        tile t; range-checked: rows outside the image read as zeros), plus one dword load for the two halo columns left /
//        right of the tile run.  U: two float4 per thread.
//     W  registers -> LDS raw[c][row][slot 1 + t][2]  (slot 0 / 65: halos; their unused halves hold zeros that the tiles on
//        the image's left / right border read instead of a neighbour)
//     T  every thread reads a tile's patch back (own pair + the neighbours' halves), transforms it in registers and writes the
//        positions to LDS V[p][c][t].  NW = 8: two threads share a tile, positions 0-7 (patch rows 0-2) / 8-15 (rows 1-3).
//     M  16 MFMAs per wave; operands: lanes 0-31 take channels (0, 1), lanes 32-63 channels (2, 3) of the chunk.
//   Main-loop iteration `it` runs M(it), T(it + 1), W(it + 2), G(it + 5) -- three register sets -- software-pipelined ACROSS its
//   one barrier: every LDS write the other waves wait for and every LDS read of this chunk's operands is issued by position 3;
//   positions 4-7 run from registers after the barrier while the wave issues G and already reads the next chunk's first operands
//   and raw patch.  sched_barrier fences pin that order (left alone, the compiler clumps the MFMAs at the top).
//   Epilogue: the output transform is linear, so each wave reduces its 8 positions to partial 2x2 outputs, the two waves of a
//   tile swap halves through LDS (row 0 of every tile is finished by ph = 0, row 1 by ph = 1) and store float2 per lane.
//   STATS: per-channel sum / sum of squares of the block's outputs for the BatchNorm that follows (as k_c3_fwd<.., STATS>).
//   dgrad       the same kernel on gy with the filter transposed and spatially flipped (k_wg_pack's dgrad flavour).
#include <algorithm>
#include <type_traits>
#include "igemm_core.h"

using namespace cpg;

namespace {

constexpr int WG_T = 64;                      // tiles per block
constexpr int WG_CK = 4;                      // input channels per chunk
constexpr int WG_V = 16 * WG_T * WG_CK;       // floats of V per chunk
constexpr int WG_RAWC = 4 * 66 * 2;           // raw floats per channel: [row][slot][2]
constexpr int WG_RAW = WG_CK * WG_RAWC;

struct WgGeom {
    int N, C, H, W, M;        // C: channels read, M: channels produced
    int th, tw;               // tiles per image column / row (H / 2, W / 2)
    int tiles_img;            // th * tw
    int64_t tiles_total;      // N * th * tw
    int nkb, nch;             // blocks of BK output channels, chunks of 4 input channels
    int span;                 // images a block's 64 consecutive tiles can touch
    unsigned nblocks;         // k_wg1 / k_wg3: logical blocks of the launch (the grid may be smaller: blocks loop over them)
    float inv_timg, inv_tw;   // 1 / tiles_img, 1 / tw (wg_divmod)
    // k_wg3<.., SPLIT> (the tail launch of wino_run): logical blocks split_first .. of the layer, each cut into split_s pieces of split_nch
    // channel chunks; piece s writes its partial outputs to y + s * split_stride
    unsigned split_first;
    int split_s, split_nch;
    int64_t split_stride;
};

// rel / d and rel % d for rel < 2^24 (exact in fp32) and a small quotient: a multiply by the reciprocal and one correction step, ~10
// vector instructions where the compiler's generic 32-bit division is ~30 -- a unit's set-up did eight of those, its epilogue four,
// and on this chip every vector instruction of the single resident wave is MFMA time (the twin of pointwise.hip's divmod_small).
__device__ __forceinline__ void wg_divmod(unsigned rel, unsigned d, float inv, unsigned &qt, unsigned &rm) {
    const unsigned n = (unsigned)((float)rel * inv);
    const int r = (int)(rel - n * d);
    const int lt = r < 0 ? 1 : 0, ge = r >= (int)d ? 1 : 0;            // (selects, no branches)
    qt = n + ge - lt;
    rm = (unsigned)(r + (lt - ge) * (int)d);
}

// First input channel of chunk `ch`.  A channel count that is not a multiple of the chunk (the reference's grown networks: 78 / 313 /
// 627 channels, models/vgg.py:124-154 with sqrt(1.5)) makes the LAST chunk start at C - 4 instead of past the end: it overlaps the chunk
// before it, and the pack kernels give the overlapped channels (already contracted there) zero filters.  The conv kernels then never read
// a channel that does not exist -- no range check, no predicate in the main loop, one scalar min per chunk.
__host__ __device__ __forceinline__ int wg_chunk_base(int ch, int C) {
    const int b = ch * WG_CK;
    return b < C - WG_CK ? b : C - WG_CK;
}

// ------------------------------------------------------------------------------ weight transform
__global__ __launch_bounds__(256) void k_wg_pack(const float *__restrict__ w, const float *__restrict__ pm, float thr,
                                                 float *__restrict__ up, int K, int C, int M, int Cin, int nch, int dgrad, int BK) {
    // one thread per (kb, ch, kl, cl); writes 16 values at stride BK * 4 floats
    const int64_t total = (int64_t)((M + BK - 1) / BK) * nch * BK * WG_CK;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int cl = (int)(o % WG_CK);
        const int kl = (int)((o / WG_CK) % BK);
        const int64_t rec = o / (WG_CK * BK);                // kb * nch + ch
        const int ch = (int)(rec % nch), kb = (int)(rec / nch);
        const int m = kb * BK + kl, c = wg_chunk_base(ch, Cin) + cl;     // produced / read channel (the last chunk may overlap the one before)
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) g[r][s] = 0.0f;
        if (m < M && c >= ch * WG_CK) {
            const int co = dgrad ? c : m, ci = dgrad ? m : c;
            const int64_t off = ((int64_t)co * C + ci) * 9;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int tap = dgrad ? 8 - (r * 3 + s) : r * 3 + s;
                    float v = w[off + tap];
                    if (pm != nullptr) v *= binarize(pm[off + tap], thr);
                    g[r][s] = v;
                }
        }
        float t[4][3];                                        // G g
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = g[0][s];
            t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
            t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
            t[3][s] = g[2][s];
        }
        float *dst = up + rec * (16 * BK * WG_CK) + kl * WG_CK + cl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[(i * 4 + 0) * (BK * WG_CK)] = t[i][0];
            dst[(i * 4 + 1) * (BK * WG_CK)] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
            dst[(i * 4 + 2) * (BK * WG_CK)] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
            dst[(i * 4 + 3) * (BK * WG_CK)] = t[i][2];
        }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int NW_>
struct WgCfg {
    static constexpr int NW = NW_, NT = 64 * NW, BK = 8 * NW;
    static constexpr int U = 16 * BK * WG_CK;              // floats of U per chunk
    // LDS map (floats): V[2 stages] | U[3 stages] | raw[2 stages]
    static constexpr int OFF_U = 2 * WG_V, OFF_RAW = OFF_U + 3 * U, SMEM = OFF_RAW + 2 * WG_RAW;   // 72.5 KB (NW = 4) / 96.5 KB (NW = 8)
    static constexpr int NROW = NW == 4 ? 4 : 2;           // patch rows a wave fetches per chunk
    static constexpr int ND = NW == 4 ? 16 : 12;           // patch elements a thread transforms (4 or 3 rows)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
};

// ------------------------------------------------------------------------------ forward / input gradient
template <int NW, bool DGRAD, bool STATS>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void k_wg_fwd(WgGeom g, const float *__restrict__ x,
                                                                      const float *__restrict__ up,
                                                                      const float *__restrict__ bias, float *__restrict__ y,
                                                                      float *__restrict__ stats) {
    using Cfg = WgCfg<NW>;
    constexpr int BK = Cfg::BK;
    __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tq = wave & 1, ph = (wave >> 1) & 1, kq = wave >> 2;
    const int li = lane & 31, lh = lane >> 5;
    const int HW = g.H * g.W;
    // staging roles: channel of the chunk, first patch row fetched (G / W), position half transformed (T)
    const int sc = NW == 4 ? wave : wave >> 1;
    const int srow = NW == 4 ? 0 : 2 * (wave & 1);
    const int half = NW == 4 ? 0 : wave & 1;               // NW = 8: positions 8 * half .. + 7 = transform rows 2 * half, + 1

    unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int kb = lb % g.nkb;
    const unsigned tb = lb / g.nkb;
    const int64_t t0 = (int64_t)tb * WG_T;                  // first tile of the block
    const int n0 = (int)(t0 / g.tiles_img);                 // first image the block touches

    // ---- G descriptors: tile `lane`, channel `sc` of the chunk ----
    constexpr int kOutOfRange = (int)0x80000000;
    int roff[Cfg::NROW];                                    // byte offset of (row, column 2 tx), or out of range (-> zeros)
    int hoff;
    int lo, ro;                                             // raw-row float index of the patch's column 0 / 3 (a zero slot on the image border)
    {
        const int64_t tg = t0 + lane;
        const bool tv = tg < g.tiles_total;
        const int n = (int)(tg / g.tiles_img), r = (int)(tg % g.tiles_img);
        const int ty = r / g.tw, tx = r % g.tw;
        const int cbase = ((n - n0) * g.C + sc) * HW;
#pragma unroll
        for (int i = 0; i < Cfg::NROW; ++i) {
            const int gh = 2 * ty - 1 + srow + i;
            roff[i] = (tv && (unsigned)gh < (unsigned)g.H) ? (cbase + gh * g.W + 2 * tx) * 4 : kOutOfRange;
        }
        lo = tx == 0 ? 0 : (lane + 1) * 2 - 1;
        ro = tx == g.tw - 1 ? 65 * 2 + 1 : (lane + 1) * 2 + 2;
        // halo loads, one instruction: lanes 0-3 fetch row (lane) of the column LEFT of tile t0 (slot 0, element 1), lanes 4-7 row
        // (lane - 4) of the column RIGHT of tile t0 + 63 (slot 65, element 0); out of range when that tile sits on the image border
        const int side = (lane >> 2) & 1, hi = lane & 3;
        const int64_t th = side ? t0 + WG_T - 1 : t0;
        const int nh = (int)(th / g.tiles_img), rh = (int)(th % g.tiles_img);
        const int tyh = rh / g.tw, txh = rh % g.tw;
        const int ghh = 2 * tyh - 1 + hi, gwh = side ? 2 * txh + 2 : 2 * txh - 1;
        const bool okh = lane < 8 && th < g.tiles_total && (unsigned)ghh < (unsigned)g.H && (unsigned)gwh < (unsigned)g.W;
        hoff = okh ? (((nh - n0) * g.C + sc) * HW + ghh * g.W + gwh) * 4 : kOutOfRange;
    }
    const bool halo_wave = NW == 4 || (wave & 1) == 0;
    const int nimg_here = min(g.span, g.N - n0);
    const __amdgpu_buffer_rsrc_t srd_x =
        __builtin_amdgcn_make_buffer_rsrc((void *)(x + (int64_t)n0 * g.C * HW), 0, nimg_here * g.C * HW * 4, 0x00020000);
    const float *ubase = up + (int64_t)kb * g.nch * Cfg::U + tid * 4;

    struct Regs {
        f32x4 u[2];
        i32x2 row[Cfg::NROW];
        float halo;
    };
    Regs rs[3];
    auto G = [&](int ch, Regs &r) {
        r.u[0] = *reinterpret_cast<const f32x4 *>(ubase + (int64_t)ch * Cfg::U);
        r.u[1] = *reinterpret_cast<const f32x4 *>(ubase + (int64_t)ch * Cfg::U + Cfg::NT * 4);
        const int soff = wg_chunk_base(ch, g.C) * HW * 4;
#pragma unroll
        for (int i = 0; i < Cfg::NROW; ++i) r.row[i] = __builtin_amdgcn_raw_buffer_load_b64(srd_x, roff[i], soff, 0);
        if (halo_wave) r.halo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, hoff, soff, 0));
    };
    // raw[c][row][slot][2]: slot 0 = left halo, 1 + t = tile t, 65 = right halo
    const int raw_c = Cfg::OFF_RAW + sc * WG_RAWC;
    const int raw_w = raw_c + srow * 132 + (lane + 1) * 2;
    const int halo_w = raw_c + (lane & 3) * 132 + ((lane >> 2) & 1 ? 65 * 2 : 1);
    // (stage indices are compile-time constants at every call site of the 6-fold unrolled main loop: LDS addresses are a
    //  per-thread base register + an immediate)
    auto W = [&](int rstage, int ustage, const Regs &r) {
        float *us = smem + Cfg::OFF_U + ustage * Cfg::U;
        *reinterpret_cast<f32x4 *>(us + tid * 4) = r.u[0];
        *reinterpret_cast<f32x4 *>(us + tid * 4 + Cfg::NT * 4) = r.u[1];
        float *raw = smem + rstage * WG_RAW;
#pragma unroll
        for (int i = 0; i < Cfg::NROW; ++i) *reinterpret_cast<i32x2 *>(raw + raw_w + i * 132) = r.row[i];
        if (halo_wave && lane < 8) raw[halo_w] = r.halo;
    };
    // T in pieces, so that the main loop can spread it between its MFMAs.  d holds patch rows half .. half + ND / 4 - 1.
    const int t_row0 = raw_c + half * 132;
    auto T_read = [&](int rstage, float (&d)[Cfg::ND]) {
        const float *raw = smem + rstage * WG_RAW + t_row0;
#pragma unroll
        for (int i = 0; i < Cfg::ND / 4; ++i) {
            const f32x2 own = *reinterpret_cast<const f32x2 *>(raw + i * 132 + (lane + 1) * 2);
            d[i * 4 + 0] = raw[i * 132 + lo];
            d[i * 4 + 1] = own[0];
            d[i * 4 + 2] = own[1];
            d[i * 4 + 3] = raw[i * 132 + ro];
        }
    };
    // V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: column pass of columns j0, j0 + 1, in place.  NW = 4: all four
    // transform rows; NW = 8: rows 2 * half, 2 * half + 1 land in d[0..3], d[4..7].  (Written on register pairs to get packed
    // v_pk_add_f32 the compiler spent more v_mov on forming the pairs than it saved: 110 VALU per chunk instead of 45.)
    auto T_col = [&](float (&d)[Cfg::ND], int j0) {
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) {
            if (NW == 4) {
                const float d0 = d[0 * 4 + j], d1 = d[1 * 4 + j], d2 = d[2 * 4 + j], d3 = d[3 * 4 + j];
                d[0 * 4 + j] = d0 - d2;
                d[1 * 4 + j] = d1 + d2;
                d[2 * 4 + j] = d2 - d1;
                d[3 * 4 + j] = d1 - d3;
            } else {
                const float e0 = d[0 * 4 + j], e1 = d[1 * 4 + j], e2 = d[2 * 4 + j];
                // half 0 holds patch rows 0, 1, 2: rows 0, 1 of B^T d = d0 - d2, d1 + d2;  half 1 holds 1, 2, 3: rows 2, 3 = d2 - d1, d1 - d3
                d[0 * 4 + j] = half ? e1 - e0 : e0 - e2;
                d[1 * 4 + j] = half ? e0 - e2 : e1 + e2;
            }
        }
    };
    // ... row pass of transform rows i0, i0 + 1 (as stored in d) and their 8 positions to V[p][c][t]
    const int v_w = (half * 8) * (WG_T * WG_CK) + sc * WG_T + lane;
    auto T_row = [&](int vstage, const float (&t)[Cfg::ND], int i0) {
        float *v = smem + vstage * WG_V + v_w;
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            v[(i * 4 + 0) * (WG_T * WG_CK)] = t[i * 4 + 0] - t[i * 4 + 2];
            v[(i * 4 + 1) * (WG_T * WG_CK)] = t[i * 4 + 1] + t[i * 4 + 2];
            v[(i * 4 + 2) * (WG_T * WG_CK)] = t[i * 4 + 2] - t[i * 4 + 1];
            v[(i * 4 + 3) * (WG_T * WG_CK)] = t[i * 4 + 1] - t[i * 4 + 3];
        }
    };

    // operand lane bases (floats): position p = ph * 8 + pp
    const int a_base = Cfg::OFF_U + (ph * 8) * (BK * WG_CK) + (kq * 32 + li) * WG_CK + lh * 2;
    const int b_base = (ph * 8) * (WG_T * WG_CK) + (2 * lh) * WG_T + tq * 32 + li;
    struct Ops {
        f32x2 a;
        float b0, b1;
    };
    auto read_ops = [&](int vstage, int k, int pp, Ops &o) {
        o.a = *reinterpret_cast<const f32x2 *>(smem + k * Cfg::U + a_base + pp * (BK * WG_CK));
        const float *vs = smem + vstage * WG_V + b_base + pp * (WG_T * WG_CK);
        o.b0 = vs[0], o.b1 = vs[WG_T];
    };

    f32x16 acc[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pp][e] = 0.0f;

    // zero slots of the raw rows (never written by W): slot 0 element 0, slot 65 element 1 -- 2 stages x 4 channels x 4 rows x 2
    if (tid < 64) smem[Cfg::OFF_RAW + (tid >> 1) * 132 + (tid & 1 ? 65 * 2 + 1 : 0)] = 0.0f;

    const int last = g.nch - 1;
    auto clampc = [&](int c) { return min(c, last); };
    float d[Cfg::ND];                  // the raw patch of T(it + 1), read before iteration `it` starts
    Ops o0, o1;                        // operands of M(it)'s first two positions, likewise
    G(0, rs[0]);
    G(clampc(1), rs[1]);
    G(clampc(2), rs[2]);
    W(0, 0, rs[0]);
    G(clampc(3), rs[0]);
    __syncthreads();
    T_read(0, d);
    T_col(d, 0);
    T_col(d, 2);
    T_row(0, d, 0);
    if (NW == 4) T_row(0, d, 2);
    W(1, 1, rs[1]);
    G(clampc(4), rs[1]);
    __syncthreads();
    T_read(1, d);
    read_ops(0, 0, 0, o0);
    read_ops(0, 0, 1, o1);
#define WG_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WG_MMA(pp, o)                                                                        \
    acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32((o).a[0], (o).b0, acc[pp], 0, 0, 0);      \
    acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32((o).a[1], (o).b1, acc[pp], 0, 0, 0)
    // iteration `it`: k = it % 3 (U stage, register set), p = it & 1 (V / raw stage) -- constants at the six call sites.
    // (operands are read two positions = 256 MFMA cycles before their MFMAs)
    auto iter = [&](int it, int k, int p, Regs &r) {
        Ops o2, o3, o4, o5, o6, o7;
        read_ops(p, k, 2, o2);
        WG_MMA(0, o0);
        T_col(d, 0);
        WG_FENCE();
        read_ops(p, k, 3, o3);
        WG_MMA(1, o1);
        T_col(d, 2);
        WG_FENCE();
        read_ops(p, k, 4, o4);
        WG_MMA(2, o2);
        T_row(p ^ 1, d, 0);                                // T(it + 1) -> V stage of it + 1
        WG_FENCE();
        read_ops(p, k, 5, o5);
        read_ops(p, k, 6, o6);
        read_ops(p, k, 7, o7);
        WG_MMA(3, o3);
        if (NW == 4) T_row(p ^ 1, d, 2);
        W(p, (k + 2) % 3, r);                              // W(it + 2) -> raw stage of it + 2, U stage (it + 2) % 3
        WG_FENCE();
        __syncthreads();
        WG_FENCE();
        WG_MMA(4, o4);
        G(clampc(it + 5), r);
        WG_FENCE();
        WG_MMA(5, o5);
        T_read(p, d);                                      // raw patch of T(it + 2)
        WG_FENCE();
        WG_MMA(6, o6);
        read_ops(p ^ 1, (k + 1) % 3, 0, o0);
        WG_FENCE();
        WG_MMA(7, o7);
        read_ops(p ^ 1, (k + 1) % 3, 1, o1);
        WG_FENCE();
    };
    for (int it = 0; it < g.nch; it += 6) {
        iter(it, 0, 0, rs[2]);
        if (it + 1 < g.nch) iter(it + 1, 1, 1, rs[0]);
        if (it + 2 < g.nch) iter(it + 2, 2, 0, rs[1]);
        if (it + 3 < g.nch) iter(it + 3, 0, 1, rs[2]);
        if (it + 4 < g.nch) iter(it + 4, 1, 0, rs[0]);
        if (it + 5 < g.nch) iter(it + 5, 2, 1, rs[1]);
    }
    __syncthreads();                   // (the trailing prefetch reads are done before the epilogue reuses the LDS)

    // ---- epilogue: output transform  Y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1];  position p = 4 i + j, this wave holds i = 2 ph, 2 ph + 1
    f32x16 own[2], give[2];                    // [b]: the output row this wave finishes (a = ph) / the other row's partial
    {
        f32x16 r0[2], r1[2];                   // R[il][b] = sum_j A^T[b][j] M[i][j]
        r0[0] = acc[0] + acc[1] + acc[2];
        r0[1] = acc[1] - acc[2] - acc[3];
        r1[0] = acc[4] + acc[5] + acc[6];
        r1[1] = acc[5] - acc[6] - acc[7];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (ph == 0) {                     // i = 0, 1:  Y0 += R0 + R1,  Y1 += R1
                own[b] = r0[b] + r1[b];
                give[b] = r1[b];
            } else {                           // i = 2, 3:  Y0 += R2,  Y1 += -R2 - R3
                own[b] = -r0[b] - r1[b];
                give[b] = r0[b];
            }
        }
    }
    // exchange buffer [wave][b * 16 + e][lane]; the partner of wave w is w ^ 2
    float *xch = smem;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[(wave * 32 + b * 16 + e) * 64 + lane] = give[b][e];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) own[b][e] += xch[((wave ^ 2) * 32 + b * 16 + e) * 64 + lane];

    const int64_t tg = t0 + tq * 32 + li;
    const bool tv = tg < g.tiles_total;
    const int n = (int)(tg / g.tiles_img), r = (int)(tg % g.tiles_img);
    const int ty = r / g.tw, tx = r % g.tw;
    float *yout = y + ((int64_t)n * g.M) * HW + (2 * ty + ph) * g.W + 2 * tx;
    float s1[16], s2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int co = kb * BK + kq * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float v0 = own[0][e], v1 = own[1][e];
        if (bias != nullptr) {
            const float bv = bias[co < g.M ? co : 0];
            v0 += bv, v1 += bv;
        }
        if (tv && co < g.M) {
            f32x2 o;
            o[0] = v0, o[1] = v1;
            *reinterpret_cast<f32x2 *>(yout + (int64_t)co * HW) = o;
        }
        if (STATS) {
            s1[e] = tv ? v0 + v1 : 0.0f;
            s2[e] = tv ? v0 * v0 + v1 * v1 : 0.0f;
        }
    }
    if (STATS) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                s1[e] += __shfl_xor(s1[e], off);
                s2[e] += __shfl_xor(s2[e], off);
            }
        __syncthreads();                       // the exchange buffer has been read
        float *red = smem;                     // [wave][32 channels][2]
        if (li == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cl = (e & 3) + 8 * (e >> 2) + 4 * lh;
                red[(wave * 32 + cl) * 2 + 0] = s1[e];
                red[(wave * 32 + cl) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        if (tid < BK && kb * BK + tid < g.M) {  // channel tid: waves kq = tid / 32, (tq, ph) = 0..3
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                a += red[(((tid >> 5) * 4 + w2) * 32 + (tid & 31)) * 2 + 0];
                b += red[(((tid >> 5) * 4 + w2) * 32 + (tid & 31)) * 2 + 1];
            }
            const unsigned ntb = gridDim.x / g.nkb;
            float *dst = stats + ((int64_t)(kb * BK + tid) * ntb + tb) * 2;
            dst[0] = a;
            dst[1] = b;
        }
    }
}

// ------------------------------------------------------------------------------ one wave = one unit ("k_wg1")
// A wave computes 32 output channels x 32 tiles for ALL 16 positions by itself: 256 accumulator registers (one wave per SIMD, the
// whole 512-entry register file), no cooperation between waves, hence NO barrier and no V / U staging through LDS:
//   * lane (li, lh) transforms the patches of tile li for channels 2 lh, 2 lh + 1 of the chunk -- and those 2 x 16 values ARE
//     its B operands (lanes 0-31 feed k = channel 0 / 1, lanes 32-63 channel 2 / 3 of the two MFMAs of a position);
//   * the A operands come straight from global memory in per-lane order: Up1[k block][chunk][q][lane][4] (positions 2q, 2q + 1 x
//     channels 2 lh, 2 lh + 1 of output channel li), 8 float4 per lane and chunk, L1 / L2 resident (the four waves of a block
//     share the k block);
//   * only the raw patch rows go through LDS, inside the wave (no synchronisation: a wave's LDS operations complete in order):
//     8 buffer_load_dwordx2 (4 rows x 2 channels, lane = aligned column pair of its tile) + 1 dword load for the 32 halo values,
//     written to raw[c][row][slot][2], read back with the neighbours' halves.
// Per chunk and wave: 32 MFMAs against 17 vector-memory, 25 LDS and ~70 vector-ALU instructions (the 4-wave block: 16 MFMAs
// against 7 + 33 + 50) -- and nothing to wait for at a barrier.
constexpr int W1_T = 32;                                  // tiles per wave
constexpr int W1_ROW = (W1_T + 2) * 2;                    // floats of a raw row: slot 0 = left halo, 1 + t, 33 = right halo
constexpr int W1_RAW = 4 * 4 * W1_ROW;                    // one stage: [channel][row][slot][2] = 1088 floats
constexpr int W1_U = 16 * 32 * WG_CK;                     // floats of U per (k block, chunk)

// U in per-lane order (see above); rec = kb * nch + ch
__device__ __forceinline__ void wg1_pack_one(int64_t o, const float *__restrict__ w, const float *__restrict__ pm, float thr,
                                             float *__restrict__ up, int C, int M, int Cin, int nch, int dgrad,
                                             int *__restrict__ live, int Mp) {
    {
        const int cl = (int)(o % WG_CK);
        const int kl = (int)((o / WG_CK) % 32);
        const int64_t rec = o / (WG_CK * 32);
        const int ch = (int)(rec % nch), kb = (int)(rec / nch);
        const int m = kb * 32 + kl, c = wg_chunk_base(ch, Cin) + cl;     // (the last chunk may overlap the one before: wg_chunk_base)
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) g[r][s] = 0.0f;
        if (m < M && c >= ch * WG_CK) {
            const int co = dgrad ? c : m, ci = dgrad ? m : c;
            const int64_t off = ((int64_t)co * C + ci) * 9;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int tap = dgrad ? 8 - (r * 3 + s) : r * 3 + s;
                    float v = w[off + tap];
                    if (pm != nullptr) v *= binarize(pm[off + tap], thr);
                    g[r][s] = v;
                }
        }
        if (live != nullptr) {                  // liveness flags (zeroed by the caller): plain stores of 1, same-value races are benign
            bool nz = false;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) nz |= g[r][s] != 0.0f;
            if (nz) {
                live[m] = 1;
                live[Mp + 4 + ch] = 1;
            }
        }
        float t[4][3], u[16];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = g[0][s];
            t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
            t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
            t[3][s] = g[2][s];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[i * 4 + 0] = t[i][0];
            u[i * 4 + 1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
            u[i * 4 + 2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
            u[i * 4 + 3] = t[i][2];
        }
        // lane = (lh = cl / 2) * 32 + kl; float (p & 1) * 2 + (cl & 1) of float4 q = p / 2
        float *dst = up + rec * W1_U + ((cl >> 1) * 32 + kl) * 4 + (cl & 1);
#pragma unroll
        for (int p = 0; p < 16; ++p) dst[(p >> 1) * 256 + (p & 1) * 2] = u[p];
    }
}

__global__ __launch_bounds__(256) void k_wg1_pack(const float *__restrict__ w, const float *__restrict__ pm, float thr,
                                                  float *__restrict__ up, int K, int C, int M, int Cin, int nch, int dgrad,
                                                  int *__restrict__ live, int Mp) {
    const int64_t total = (int64_t)((M + 31) / 32) * nch * 32 * WG_CK;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x)
        wg1_pack_one(o, w, pm, thr, up, C, M, Cin, nch, dgrad, live, Mp);
}

// Up to two pack jobs in ONE launch (cpg_conv2d_pack: a training step's forward and input-gradient operands of a layer -- the same
// weights in two layouts): work items [0, a.total) are job a's, [a.total, a.total + b.total) job b's.  Family 1 is pointwise.hip's
// k_pw_pack (Wp[c][m] = W[co][ci] * bin(pm), fwd: c = ci, m = co; dgrad: c = co, m = ci), family 2 wg1_pack_one above.
struct PackArgs {
    int family, K, C, a, b, c, d;
    long long total;
    float *out;
};
__device__ __forceinline__ void pack_job_one(const PackArgs &j, int64_t o, const float *__restrict__ w, const float *__restrict__ pm, float thr) {
    if (j.family == 1) {
        const int Mp = j.b, dgrad = j.c;
        const int m = (int)(o % Mp), c = (int)(o / Mp);
        const int co = dgrad ? c : m, ci = dgrad ? m : c;
        float v = 0.0f;
        if (co < j.K && ci < j.C) {
            const int64_t off = (int64_t)co * j.C + ci;
            v = w[off];
            if (pm != nullptr) v *= binarize(pm[off], thr);
        }
        j.out[o] = v;
    } else {
        wg1_pack_one(o, w, pm, thr, j.out, j.C, j.a, j.b, j.c, j.d, nullptr, 0);
    }
}
__global__ __launch_bounds__(256) void k_pack_jobs(const PackArgs a, const PackArgs b, const float *__restrict__ w,
                                                   const float *__restrict__ pm, float thr) {
    const int64_t total = a.total + b.total;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        if (o < a.total)
            pack_job_one(a, o, w, pm, thr);
        else
            pack_job_one(b, o - a.total, w, pm, thr);
    }
}

// Inference: eval-mode BatchNorm (+ ReLU) in the epilogue -- the expression of conv3x3.hip's C3BnEval -- and the dead-channel skip:
// `live` (may be null) are the flags k_wg1_pack wrote, same layout as k_c3_pack's (live[m]: output channel m has a non-zero
// weight; live[Mp + 4 + q]: input chunk q has one; live[Mp] receives 4 x the chunks up to the last live one, live[Mp + 1] counts
// the waves that skipped their MFMA loop).
struct WgBnEval {
    const float *gamma, *beta, *mean, *var;
    float eps;
    int relu;
    int *live;
    int Mp;
};

// -DWG_TIMING (development builds only, tools/attic/diag_wg_timing.py): every wave of k_wg1 leaves the constant-clock time of its
// entry / prologue start / main loop start / epilogue start / exit and its hardware slot in wg_dbg.
#ifdef WG_TIMING
__device__ unsigned long long wg_dbg[65536 * 8];
#define WG_STAMP(i) do { if (lane == 0 && dbg_u < 65536) wg_dbg[dbg_u * 8 + (i)] = wall_clock64(); } while (0)
#else
#define WG_STAMP(i)
#endif
template <bool DGRAD, bool STATS, bool BNE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_wg1(WgGeom g, const float *__restrict__ x, const float *__restrict__ up, const float *__restrict__ bias,
           float *__restrict__ y, float *__restrict__ stats, WgBnEval bn) {
    __shared__ __attribute__((aligned(16))) float smem_all[4 * 2 * W1_RAW];
    // (readfirstlane: tells the compiler the wave index is wave-uniform, so that everything derived from it -- the tile run, the
    //  buffer descriptor of its first image -- lives in scalar registers; without it every buffer load became a waterfall loop)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int HW = g.H * g.W;
    float *smem = smem_all + wave * 2 * W1_RAW;               // this wave's private raw stages
    // PERSISTENT blocks: the grid is at most a few times what the chip holds (one block per CU: 512 registers per wave; see
    // wino_grids) and every wave walks over the logical blocks base + blockIdx, base += gridDim -- no dispatch gap between
    // units (0.6 us of 27 on the 64-channel layers), the kernel arguments and the code stay where they are.
    for (unsigned base = 0; base < g.nblocks; base += gridDim.x) {
    // (virtual block v runs on XCD v % 8 = blockIdx % 8 -- the grid is a multiple of 8 blocks whenever there is more than one
    //  round -- and XCD x owns the x-th eighth of ALL logical blocks, walking through it round after round: consecutive tile runs,
    //  which share half of their input rows, stay in one XCD's L2 as with one block per logical block)
    const unsigned v = base + blockIdx.x;
    if (v >= g.nblocks) break;
    const unsigned lb = xcd_remap(v, g.nblocks);
    const int kb = lb % g.nkb;
    // (32-bit tile arithmetic: the host refuses N * tiles >= 2^31; 64-bit divisions cost the prologue ~1.5 us per wave)
    const unsigned ttot = (unsigned)g.tiles_total, timg = (unsigned)g.tiles_img, twu = (unsigned)g.tw;
    const unsigned run = (lb / g.nkb) * 4 + wave;             // tile run of 32
    const unsigned t0 = run * W1_T;
    if (t0 >= ttot) continue;                                 // (no barriers anywhere: a wave may skip)
#ifdef WG_TIMING
    const unsigned dbg_u = lb * 4 + wave;
    WG_STAMP(0);
    if (lane == 0 && dbg_u < 65536) wg_dbg[dbg_u * 8 + 6] = __builtin_amdgcn_s_getreg(63492), wg_dbg[dbg_u * 8 + 7] = __builtin_amdgcn_s_getreg(63508);
#endif
    const int n0 = (int)(t0 / timg);

    constexpr int kOutOfRange = (int)0x80000000;
    int roff[4], hoff, lo, ro;
    {
        const unsigned tg = t0 + li;
        const bool tv = tg < ttot;
        const int n = (int)(tg / timg), r = (int)(tg % timg);
        const int ty = (int)((unsigned)r / twu), tx = (int)((unsigned)r % twu);
        const int cbase = ((n - n0) * g.C + 2 * lh) * HW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gh = 2 * ty - 1 + i;
            roff[i] = (tv && (unsigned)gh < (unsigned)g.H) ? (cbase + gh * g.W + 2 * tx) * 4 : kOutOfRange;
        }
        lo = tx == 0 ? 0 : (li + 1) * 2 - 1;
        ro = tx == g.tw - 1 ? (W1_T + 1) * 2 + 1 : (li + 1) * 2 + 2;
        // halo: lanes 0-31 = (side, row, channel): the column left of tile t0 / right of tile t0 + 31
        const int side = (lane >> 4) & 1, hi = (lane >> 2) & 3, hc = lane & 3;
        const unsigned th = side ? t0 + W1_T - 1 : t0;
        const int nh = (int)(th / timg), rh = (int)(th % timg);
        const int tyh = (int)((unsigned)rh / twu), txh = (int)((unsigned)rh % twu);
        const int ghh = 2 * tyh - 1 + hi, gwh = side ? 2 * txh + 2 : 2 * txh - 1;
        const bool okh = lane < 32 && th < ttot && (unsigned)ghh < (unsigned)g.H && (unsigned)gwh < (unsigned)g.W;
        hoff = okh ? (((nh - n0) * g.C + hc) * HW + ghh * g.W + gwh) * 4 : kOutOfRange;
    }
    const int span = (W1_T + g.tiles_img - 1) / g.tiles_img + 1;
    const int nimg_here = min(span, g.N - n0);
    // byte offset of the LAST chunk's first channel, set once the chunk count is known: min(last chunk * 4, C - 4) channels.  Every row load
    // takes min(chunk * stride, x_last_off): chunks past the end re-read the last one (what clampc did for them), and with a channel count
    // that is no multiple of 4 the last chunk starts at C - 4, overlapping its neighbour (wg_chunk_base; the pack kernel zeroes its filters
    // for the channels the neighbour already contracted) -- one scalar multiply and one min per chunk, as before.
    int x_last_off = 0, x_last = 0;
    const __amdgpu_buffer_rsrc_t srd_x =
        __builtin_amdgcn_make_buffer_rsrc((void *)(x + (int64_t)n0 * g.C * HW), 0, nimg_here * g.C * HW * 4, 0x00020000);
    // (U through a buffer descriptor: two per-lane byte offsets per unit, the chunk in the scalar offset, q in the instruction offset)
    const __amdgpu_buffer_rsrc_t srd_u = __builtin_amdgcn_make_buffer_rsrc((void *)up, 0, (int)((int64_t)g.nkb * g.nch * W1_U * 4), 0x00020000);
    const int ubase = (kb * g.nch * W1_U + lane * 4) * 4;

    // raw[c][row][slot][2] offsets of this lane (channels 2 lh + j)
    const int raw_own = (2 * lh) * 4 * W1_ROW + (li + 1) * 2;
    const int halo_w = ((lane & 3) * 4 + ((lane >> 2) & 3)) * W1_ROW + (((lane >> 4) & 1) ? (W1_T + 1) * 2 : 1);
    // zero slots (never written): slot 0 element 0, slot 33 element 1 of the 2 x 16 rows
    if (lane < 32) {
        smem[lane * W1_ROW] = 0.0f;
        smem[lane * W1_ROW + (W1_T + 1) * 2 + 1] = 0.0f;
    }

    struct Rows {
        i32x2 r[2][4];
        float halo;
    };
    auto G_row1 = [&](int ch, Rows &q, int k) {             // k = 0..7: row k & 3 of channel 2 lh + k / 4;  k = 8: the halo values
        // (the inference instance has no scalar register left for x_last_off -- it spilled a vector register pair to scratch --: it clamps the
        //  chunk index with `last`, which its filter loads keep live anyway)
        const int soff = BNE ? wg_chunk_base(min(ch, x_last), g.C) * HW * 4 : min(ch * (WG_CK * HW * 4), x_last_off);      // (see x_last_off)
        if (k < 8)
            q.r[k >> 2][k & 3] = __builtin_amdgcn_raw_buffer_load_b64(srd_x, roff[k & 3], soff + (k >> 2) * HW * 4, 0);
        else
            q.halo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, hoff, soff, 0));
    };
    auto G_rows = [&](int ch, Rows &q) {
#pragma unroll
        for (int k = 0; k < 9; ++k) G_row1(ch, q, k);
    };
    auto W_row1 = [&](int stage, const Rows &q, int k) {
        float *raw = smem + stage * W1_RAW;
        if (k < 8)
            *reinterpret_cast<i32x2 *>(raw + raw_own + k * W1_ROW) = q.r[k >> 2][k & 3];
        else if (lane < 32)
            raw[halo_w] = q.halo;
    };
    auto W_rows = [&](int stage, const Rows &q) {
#pragma unroll
        for (int k = 0; k < 9; ++k) W_row1(stage, q, k);
    };
    const int ubase4 = ubase + 4 * 256 * 4;                   // (the instruction offset has 12 bits: q = 4 .. 7 from a second base)
    auto G_u1 = [&](int ch, f32x4 (&u)[8], int q) {
        u[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_u, (q < 4 ? ubase : ubase4) + (q & 3) * 256 * 4, ch * (W1_U * 4), 0));
    };
    auto G_u = [&](int ch, f32x4 (&u)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) G_u1(ch, u, q);
    };
    // patch of (tile li, channel 2 lh + j) -> the 16 transform-domain values, in place
    auto T_read2 = [&](int stage, int j, float (&d)[16], int i0) {       // patch rows i0, i0 + 1
        const float *raw = smem + stage * W1_RAW + (2 * lh + j) * 4 * W1_ROW;
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            const f32x2 own = *reinterpret_cast<const f32x2 *>(raw + i * W1_ROW + (li + 1) * 2);
            d[i * 4 + 0] = raw[i * W1_ROW + lo];
            d[i * 4 + 1] = own[0];
            d[i * 4 + 2] = own[1];
            d[i * 4 + 3] = raw[i * W1_ROW + ro];
        }
    };
    auto T_read = [&](int stage, int j, float (&d)[16]) {
        const float *raw = smem + stage * W1_RAW + (2 * lh + j) * 4 * W1_ROW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 own = *reinterpret_cast<const f32x2 *>(raw + i * W1_ROW + (li + 1) * 2);
            d[i * 4 + 0] = raw[i * W1_ROW + lo];
            d[i * 4 + 1] = own[0];
            d[i * 4 + 2] = own[1];
            d[i * 4 + 3] = raw[i * W1_ROW + ro];
        }
    };
    auto T_col = [&](float (&d)[16], int j0) {
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) {
            const float d0 = d[0 * 4 + j], d1 = d[1 * 4 + j], d2 = d[2 * 4 + j], d3 = d[3 * 4 + j];
            d[0 * 4 + j] = d0 - d2;
            d[1 * 4 + j] = d1 + d2;
            d[2 * 4 + j] = d2 - d1;
            d[3 * 4 + j] = d1 - d3;
            asm volatile("" : "+v"(d[0 * 4 + j]), "+v"(d[1 * 4 + j]), "+v"(d[2 * 4 + j]), "+v"(d[3 * 4 + j]));
        }
    };
    // (the asm "uses" pin the results where they are computed: their only consumers are the NEXT iteration's MFMAs, and the
    //  optimizer otherwise sinks the whole transform -- 64 adds -- in front of them, where nothing runs in its shadow)
    auto T_rowp = [&](float (&d)[16], int i) {
        const float t0_ = d[i * 4 + 0], t1 = d[i * 4 + 1], t2 = d[i * 4 + 2], t3 = d[i * 4 + 3];
        d[i * 4 + 0] = t0_ - t2;
        d[i * 4 + 1] = t1 + t2;
        d[i * 4 + 2] = t2 - t1;
        d[i * 4 + 3] = t1 - t3;
        asm volatile("" : "+v"(d[i * 4 + 0]), "+v"(d[i * 4 + 1]), "+v"(d[i * 4 + 2]), "+v"(d[i * 4 + 3]));
    };


    int nch = g.nch;
    if (BNE && bn.live != nullptr) {                          // inference: skip what apply_mask killed (wave-uniform decisions)
        const int alive = (kb * 32 + li < g.M) ? bn.live[kb * 32 + li] : 0;
        const bool dead = __ballot(alive != 0) == 0ull;
        int lastc = g.nch;
        while (lastc > 0 && bn.live[bn.Mp + 4 + lastc - 1] == 0) --lastc;
        nch = dead ? 0 : lastc;
        if (lane == 0) {
            if (dead) atomicAdd(&bn.live[bn.Mp + 1], 1);
            if (blockIdx.x == 0 && wave == 0) bn.live[bn.Mp] = lastc * WG_CK;
        }
    }
    const int last = nch - 1;
    x_last = last;
    if (!BNE) x_last_off = __builtin_amdgcn_readfirstlane(min(last * WG_CK, g.C - WG_CK) * HW * 4);
    auto clampc = [&](int c) { return min(c, last); };
    f32x4 u0[8], u1[8], u2[8];             // U of chunks it, it + 1, it + 2 (three rotating sets: requested two iterations ahead)
    float b0[16], b1[16];                  // B operands of the current chunk: V of channels 2 lh, 2 lh + 1
    float n0v[16], n1v[16];                // ... of the next chunk, transformed while the current chunk's MFMAs run
    Rows rw0, rw1;                         // raw rows in flight: chunk c uses set c & 1, requested FOUR iterations before its MFMAs
    // With one wave per SIMD a wait is an idle MFMA pipe: the loads that miss L2 (the transformed filter of a 512 x 512 layer is
    // 16 MB, the input's first touch) take longer than one iteration (0.85 us), so everything is requested two iterations early.
    WG_STAMP(1);
    // prologue: chunk 0 operands, chunks 0 / 1 raw rows in LDS, chunks 2 / 3 rows and U(0), U(1) in flight
    if (nch > 0) {
    G_rows(0, rw0);
    G_rows(clampc(1), rw1);
    G_u(0, u0);
    G_u(clampc(1), u1);
    }
    // the accumulators are cleared while those loads fly (256 instructions, 0.4 us)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0" : : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    asm volatile("v_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0" : : : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    asm volatile("v_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0" : : : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    asm volatile("v_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0" : : : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    asm volatile("v_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0" : : : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    asm volatile("v_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0" : : : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    asm volatile("v_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0" : : : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    asm volatile("v_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" : : : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
    asm volatile("v_accvgpr_write_b32 a128, 0\n\tv_accvgpr_write_b32 a129, 0\n\tv_accvgpr_write_b32 a130, 0\n\tv_accvgpr_write_b32 a131, 0\n\tv_accvgpr_write_b32 a132, 0\n\tv_accvgpr_write_b32 a133, 0\n\tv_accvgpr_write_b32 a134, 0\n\tv_accvgpr_write_b32 a135, 0\n\tv_accvgpr_write_b32 a136, 0\n\tv_accvgpr_write_b32 a137, 0\n\tv_accvgpr_write_b32 a138, 0\n\tv_accvgpr_write_b32 a139, 0\n\tv_accvgpr_write_b32 a140, 0\n\tv_accvgpr_write_b32 a141, 0\n\tv_accvgpr_write_b32 a142, 0\n\tv_accvgpr_write_b32 a143, 0" : : : "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");
    asm volatile("v_accvgpr_write_b32 a144, 0\n\tv_accvgpr_write_b32 a145, 0\n\tv_accvgpr_write_b32 a146, 0\n\tv_accvgpr_write_b32 a147, 0\n\tv_accvgpr_write_b32 a148, 0\n\tv_accvgpr_write_b32 a149, 0\n\tv_accvgpr_write_b32 a150, 0\n\tv_accvgpr_write_b32 a151, 0\n\tv_accvgpr_write_b32 a152, 0\n\tv_accvgpr_write_b32 a153, 0\n\tv_accvgpr_write_b32 a154, 0\n\tv_accvgpr_write_b32 a155, 0\n\tv_accvgpr_write_b32 a156, 0\n\tv_accvgpr_write_b32 a157, 0\n\tv_accvgpr_write_b32 a158, 0\n\tv_accvgpr_write_b32 a159, 0" : : : "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159");
    asm volatile("v_accvgpr_write_b32 a160, 0\n\tv_accvgpr_write_b32 a161, 0\n\tv_accvgpr_write_b32 a162, 0\n\tv_accvgpr_write_b32 a163, 0\n\tv_accvgpr_write_b32 a164, 0\n\tv_accvgpr_write_b32 a165, 0\n\tv_accvgpr_write_b32 a166, 0\n\tv_accvgpr_write_b32 a167, 0\n\tv_accvgpr_write_b32 a168, 0\n\tv_accvgpr_write_b32 a169, 0\n\tv_accvgpr_write_b32 a170, 0\n\tv_accvgpr_write_b32 a171, 0\n\tv_accvgpr_write_b32 a172, 0\n\tv_accvgpr_write_b32 a173, 0\n\tv_accvgpr_write_b32 a174, 0\n\tv_accvgpr_write_b32 a175, 0" : : : "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175");
    asm volatile("v_accvgpr_write_b32 a176, 0\n\tv_accvgpr_write_b32 a177, 0\n\tv_accvgpr_write_b32 a178, 0\n\tv_accvgpr_write_b32 a179, 0\n\tv_accvgpr_write_b32 a180, 0\n\tv_accvgpr_write_b32 a181, 0\n\tv_accvgpr_write_b32 a182, 0\n\tv_accvgpr_write_b32 a183, 0\n\tv_accvgpr_write_b32 a184, 0\n\tv_accvgpr_write_b32 a185, 0\n\tv_accvgpr_write_b32 a186, 0\n\tv_accvgpr_write_b32 a187, 0\n\tv_accvgpr_write_b32 a188, 0\n\tv_accvgpr_write_b32 a189, 0\n\tv_accvgpr_write_b32 a190, 0\n\tv_accvgpr_write_b32 a191, 0" : : : "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191");
    asm volatile("v_accvgpr_write_b32 a192, 0\n\tv_accvgpr_write_b32 a193, 0\n\tv_accvgpr_write_b32 a194, 0\n\tv_accvgpr_write_b32 a195, 0\n\tv_accvgpr_write_b32 a196, 0\n\tv_accvgpr_write_b32 a197, 0\n\tv_accvgpr_write_b32 a198, 0\n\tv_accvgpr_write_b32 a199, 0\n\tv_accvgpr_write_b32 a200, 0\n\tv_accvgpr_write_b32 a201, 0\n\tv_accvgpr_write_b32 a202, 0\n\tv_accvgpr_write_b32 a203, 0\n\tv_accvgpr_write_b32 a204, 0\n\tv_accvgpr_write_b32 a205, 0\n\tv_accvgpr_write_b32 a206, 0\n\tv_accvgpr_write_b32 a207, 0" : : : "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207");
    asm volatile("v_accvgpr_write_b32 a208, 0\n\tv_accvgpr_write_b32 a209, 0\n\tv_accvgpr_write_b32 a210, 0\n\tv_accvgpr_write_b32 a211, 0\n\tv_accvgpr_write_b32 a212, 0\n\tv_accvgpr_write_b32 a213, 0\n\tv_accvgpr_write_b32 a214, 0\n\tv_accvgpr_write_b32 a215, 0\n\tv_accvgpr_write_b32 a216, 0\n\tv_accvgpr_write_b32 a217, 0\n\tv_accvgpr_write_b32 a218, 0\n\tv_accvgpr_write_b32 a219, 0\n\tv_accvgpr_write_b32 a220, 0\n\tv_accvgpr_write_b32 a221, 0\n\tv_accvgpr_write_b32 a222, 0\n\tv_accvgpr_write_b32 a223, 0" : : : "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223");
    asm volatile("v_accvgpr_write_b32 a224, 0\n\tv_accvgpr_write_b32 a225, 0\n\tv_accvgpr_write_b32 a226, 0\n\tv_accvgpr_write_b32 a227, 0\n\tv_accvgpr_write_b32 a228, 0\n\tv_accvgpr_write_b32 a229, 0\n\tv_accvgpr_write_b32 a230, 0\n\tv_accvgpr_write_b32 a231, 0\n\tv_accvgpr_write_b32 a232, 0\n\tv_accvgpr_write_b32 a233, 0\n\tv_accvgpr_write_b32 a234, 0\n\tv_accvgpr_write_b32 a235, 0\n\tv_accvgpr_write_b32 a236, 0\n\tv_accvgpr_write_b32 a237, 0\n\tv_accvgpr_write_b32 a238, 0\n\tv_accvgpr_write_b32 a239, 0" : : : "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239");
    asm volatile("v_accvgpr_write_b32 a240, 0\n\tv_accvgpr_write_b32 a241, 0\n\tv_accvgpr_write_b32 a242, 0\n\tv_accvgpr_write_b32 a243, 0\n\tv_accvgpr_write_b32 a244, 0\n\tv_accvgpr_write_b32 a245, 0\n\tv_accvgpr_write_b32 a246, 0\n\tv_accvgpr_write_b32 a247, 0\n\tv_accvgpr_write_b32 a248, 0\n\tv_accvgpr_write_b32 a249, 0\n\tv_accvgpr_write_b32 a250, 0\n\tv_accvgpr_write_b32 a251, 0\n\tv_accvgpr_write_b32 a252, 0\n\tv_accvgpr_write_b32 a253, 0\n\tv_accvgpr_write_b32 a254, 0\n\tv_accvgpr_write_b32 a255, 0" : : : "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    __builtin_amdgcn_sched_barrier(0);
    if (nch > 0) {
    W_rows(0, rw0);
    W_rows(1, rw1);
    G_rows(clampc(2), rw0);
    G_rows(clampc(3), rw1);
    T_read(0, 0, b0);
    T_read(0, 1, b1);
    T_col(b0, 0); T_col(b0, 2); T_col(b1, 0); T_col(b1, 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) { T_rowp(b0, i); T_rowp(b1, i); }
    }

    // The 256 accumulators live in FIXED accumulation registers a[16 p : 16 p + 15], named in the asm text and declared as clobbers.
    // (With the MFMA builtin -- or an "a" constraint on a variable -- the register allocator kept copies of them in VGPRs, shuffled
    //  v_accvgpr_read / write by the hundred inside the loop and spilled 250 registers.)
#define W1_ONE_0(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" : : "v"(A), "v"(B) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15")
#define W1_ONE_1(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[16:31], %0, %1, a[16:31]" : : "v"(A), "v"(B) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31")
#define W1_ONE_2(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[32:47], %0, %1, a[32:47]" : : "v"(A), "v"(B) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47")
#define W1_ONE_3(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[48:63], %0, %1, a[48:63]" : : "v"(A), "v"(B) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63")
#define W1_ONE_4(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[64:79], %0, %1, a[64:79]" : : "v"(A), "v"(B) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79")
#define W1_ONE_5(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[80:95], %0, %1, a[80:95]" : : "v"(A), "v"(B) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95")
#define W1_ONE_6(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[96:111], %0, %1, a[96:111]" : : "v"(A), "v"(B) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111")
#define W1_ONE_7(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[112:127], %0, %1, a[112:127]" : : "v"(A), "v"(B) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127")
#define W1_ONE_8(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[128:143], %0, %1, a[128:143]" : : "v"(A), "v"(B) : "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143")
#define W1_ONE_9(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[144:159], %0, %1, a[144:159]" : : "v"(A), "v"(B) : "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159")
#define W1_ONE_10(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[160:175], %0, %1, a[160:175]" : : "v"(A), "v"(B) : "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175")
#define W1_ONE_11(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[176:191], %0, %1, a[176:191]" : : "v"(A), "v"(B) : "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191")
#define W1_ONE_12(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[192:207], %0, %1, a[192:207]" : : "v"(A), "v"(B) : "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207")
#define W1_ONE_13(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[208:223], %0, %1, a[208:223]" : : "v"(A), "v"(B) : "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223")
#define W1_ONE_14(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[224:239], %0, %1, a[224:239]" : : "v"(A), "v"(B) : "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239")
#define W1_ONE_15(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[240:255], %0, %1, a[240:255]" : : "v"(A), "v"(B) : "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255")
// MFMA h (0: channels 0 / 2 of the chunk, 1: channels 1 / 3) of position p, then a slice of the staging work, then a fence.  ONE
// MFMA per slot: a wave issues in order, so whatever stands behind an MFMA pair only gets the second one's 64-cycle shadow --
// with 9 loads or 9 LDS writes in one slot the matrix pipe waited for the memory pipelines' issue
#define W1_SLOT(p, h, U, B, work)                                                      \
    W1_ONE_##p((U)[(p) >> 1][((p) & 1) * 2 + (h)], (B)[p]);                            \
    work;                                                                              \
    __builtin_amdgcn_sched_barrier(0)
#define W1_FENCE() __builtin_amdgcn_sched_barrier(0)
    // iteration it (par = it & 1): M(it) from (ucur, c0, c1); T(it + 1) from raw stage (it + 1) & 1 into (x0, x1); W(it + 2) from
    // the row set of this parity, which is then re-used for the loads of chunk it + 4; U(it + 2) is requested at the top
    auto iter = [&](int it, int par, f32x4 (&ucur)[8], f32x4 (&ufar)[8], Rows &rows, float (&c0)[16], float (&c1)[16],
                    float (&x0)[16], float (&x1)[16]) {
        const int cu = clampc(it + 2), cr = it + 4;
        W1_SLOT(0, 0, ucur, c0, T_read2(par ^ 1, 0, x0, 0));
        W1_SLOT(0, 1, ucur, c1, T_read2(par ^ 1, 0, x0, 2));
        W1_SLOT(1, 0, ucur, c0, T_read2(par ^ 1, 1, x1, 0));
        W1_SLOT(1, 1, ucur, c1, T_read2(par ^ 1, 1, x1, 2));
        W1_SLOT(2, 0, ucur, c0, T_col(x0, 0));
        W1_SLOT(2, 1, ucur, c1, T_col(x0, 2));
        W1_SLOT(3, 0, ucur, c0, T_col(x1, 0));
        W1_SLOT(3, 1, ucur, c1, T_col(x1, 2));
        W1_SLOT(4, 0, ucur, c0, T_rowp(x0, 0));
        W1_SLOT(4, 1, ucur, c1, T_rowp(x0, 1));
        W1_SLOT(5, 0, ucur, c0, T_rowp(x0, 2));
        W1_SLOT(5, 1, ucur, c1, T_rowp(x0, 3));
        W1_SLOT(6, 0, ucur, c0, T_rowp(x1, 0));
        W1_SLOT(6, 1, ucur, c1, T_rowp(x1, 1));
        W1_SLOT(7, 0, ucur, c0, T_rowp(x1, 2));
        W1_SLOT(7, 1, ucur, c1, T_rowp(x1, 3));
        W1_SLOT(8, 0, ucur, c0, G_u1(cu, ufar, 0));
        W1_SLOT(8, 1, ucur, c1, G_u1(cu, ufar, 1));
        W1_SLOT(9, 0, ucur, c0, G_u1(cu, ufar, 2));
        W1_SLOT(9, 1, ucur, c1, G_u1(cu, ufar, 3));
        W1_SLOT(10, 0, ucur, c0, G_u1(cu, ufar, 4));
        W1_SLOT(10, 1, ucur, c1, G_u1(cu, ufar, 5));
        W1_SLOT(11, 0, ucur, c0, G_u1(cu, ufar, 6));
        W1_SLOT(11, 1, ucur, c1, G_u1(cu, ufar, 7));
        // rows of chunk it + 2 -> raw stage (it + 2) & 1, each register pair re-requested for chunk it + 4 right after its store
        W1_SLOT(12, 0, ucur, c0, W_row1(par, rows, 0); W_row1(par, rows, 1); G_row1(cr, rows, 0));
        W1_SLOT(12, 1, ucur, c1, W_row1(par, rows, 2); G_row1(cr, rows, 1));
        W1_SLOT(13, 0, ucur, c0, W_row1(par, rows, 3); G_row1(cr, rows, 2));
        W1_SLOT(13, 1, ucur, c1, W_row1(par, rows, 4); G_row1(cr, rows, 3));
        W1_SLOT(14, 0, ucur, c0, W_row1(par, rows, 5); G_row1(cr, rows, 4));
        W1_SLOT(14, 1, ucur, c1, W_row1(par, rows, 6); G_row1(cr, rows, 5));
        W1_SLOT(15, 0, ucur, c0, W_row1(par, rows, 7); G_row1(cr, rows, 6));
        W1_SLOT(15, 1, ucur, c1, W_row1(par, rows, 8); G_row1(cr, rows, 7); G_row1(cr, rows, 8));
    };
    WG_STAMP(2);
    for (int it = 0; it < nch; it += 6) {
        iter(it, 0, u0, u2, rw0, b0, b1, n0v, n1v);
        if (it + 1 < nch) iter(it + 1, 1, u1, u0, rw1, n0v, n1v, b0, b1);
        if (it + 2 < nch) iter(it + 2, 0, u2, u1, rw0, b0, b1, n0v, n1v);
        if (it + 3 < nch) iter(it + 3, 1, u0, u2, rw1, n0v, n1v, b0, b1);
        if (it + 4 < nch) iter(it + 4, 0, u1, u0, rw0, b0, b1, n0v, n1v);
        if (it + 5 < nch) iter(it + 5, 1, u2, u1, rw1, n0v, n1v, b0, b1);
    }

    WG_STAMP(3);
    // ---- epilogue: Y = A^T M A in registers (M[i][j] = acc[4 i + j]),  A^T = [1 1 1 0; 0 1 -1 -1]
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (the last MFMA's result is read below: the compiler cannot see into the asm)
    const unsigned tg = t0 + li;
    const bool tv = tg < ttot;
    const int n = (int)(tg / timg), r = (int)(tg % timg);
    const int ty = (int)((unsigned)r / twu), tx = (int)((unsigned)r % twu);
    float *yout = y + ((int64_t)n * g.M) * HW + (2 * ty) * g.W + 2 * tx;
#define W1_RD_0(m) asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a16\n\tv_accvgpr_read_b32 %2, a32\n\tv_accvgpr_read_b32 %3, a48\n\tv_accvgpr_read_b32 %4, a64\n\tv_accvgpr_read_b32 %5, a80\n\tv_accvgpr_read_b32 %6, a96\n\tv_accvgpr_read_b32 %7, a112\n\tv_accvgpr_read_b32 %8, a128\n\tv_accvgpr_read_b32 %9, a144\n\tv_accvgpr_read_b32 %10, a160\n\tv_accvgpr_read_b32 %11, a176\n\tv_accvgpr_read_b32 %12, a192\n\tv_accvgpr_read_b32 %13, a208\n\tv_accvgpr_read_b32 %14, a224\n\tv_accvgpr_read_b32 %15, a240" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_1(m) asm volatile("v_accvgpr_read_b32 %0, a1\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a33\n\tv_accvgpr_read_b32 %3, a49\n\tv_accvgpr_read_b32 %4, a65\n\tv_accvgpr_read_b32 %5, a81\n\tv_accvgpr_read_b32 %6, a97\n\tv_accvgpr_read_b32 %7, a113\n\tv_accvgpr_read_b32 %8, a129\n\tv_accvgpr_read_b32 %9, a145\n\tv_accvgpr_read_b32 %10, a161\n\tv_accvgpr_read_b32 %11, a177\n\tv_accvgpr_read_b32 %12, a193\n\tv_accvgpr_read_b32 %13, a209\n\tv_accvgpr_read_b32 %14, a225\n\tv_accvgpr_read_b32 %15, a241" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_2(m) asm volatile("v_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a18\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a50\n\tv_accvgpr_read_b32 %4, a66\n\tv_accvgpr_read_b32 %5, a82\n\tv_accvgpr_read_b32 %6, a98\n\tv_accvgpr_read_b32 %7, a114\n\tv_accvgpr_read_b32 %8, a130\n\tv_accvgpr_read_b32 %9, a146\n\tv_accvgpr_read_b32 %10, a162\n\tv_accvgpr_read_b32 %11, a178\n\tv_accvgpr_read_b32 %12, a194\n\tv_accvgpr_read_b32 %13, a210\n\tv_accvgpr_read_b32 %14, a226\n\tv_accvgpr_read_b32 %15, a242" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_3(m) asm volatile("v_accvgpr_read_b32 %0, a3\n\tv_accvgpr_read_b32 %1, a19\n\tv_accvgpr_read_b32 %2, a35\n\tv_accvgpr_read_b32 %3, a51\n\tv_accvgpr_read_b32 %4, a67\n\tv_accvgpr_read_b32 %5, a83\n\tv_accvgpr_read_b32 %6, a99\n\tv_accvgpr_read_b32 %7, a115\n\tv_accvgpr_read_b32 %8, a131\n\tv_accvgpr_read_b32 %9, a147\n\tv_accvgpr_read_b32 %10, a163\n\tv_accvgpr_read_b32 %11, a179\n\tv_accvgpr_read_b32 %12, a195\n\tv_accvgpr_read_b32 %13, a211\n\tv_accvgpr_read_b32 %14, a227\n\tv_accvgpr_read_b32 %15, a243" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_4(m) asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a20\n\tv_accvgpr_read_b32 %2, a36\n\tv_accvgpr_read_b32 %3, a52\n\tv_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a84\n\tv_accvgpr_read_b32 %6, a100\n\tv_accvgpr_read_b32 %7, a116\n\tv_accvgpr_read_b32 %8, a132\n\tv_accvgpr_read_b32 %9, a148\n\tv_accvgpr_read_b32 %10, a164\n\tv_accvgpr_read_b32 %11, a180\n\tv_accvgpr_read_b32 %12, a196\n\tv_accvgpr_read_b32 %13, a212\n\tv_accvgpr_read_b32 %14, a228\n\tv_accvgpr_read_b32 %15, a244" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_5(m) asm volatile("v_accvgpr_read_b32 %0, a5\n\tv_accvgpr_read_b32 %1, a21\n\tv_accvgpr_read_b32 %2, a37\n\tv_accvgpr_read_b32 %3, a53\n\tv_accvgpr_read_b32 %4, a69\n\tv_accvgpr_read_b32 %5, a85\n\tv_accvgpr_read_b32 %6, a101\n\tv_accvgpr_read_b32 %7, a117\n\tv_accvgpr_read_b32 %8, a133\n\tv_accvgpr_read_b32 %9, a149\n\tv_accvgpr_read_b32 %10, a165\n\tv_accvgpr_read_b32 %11, a181\n\tv_accvgpr_read_b32 %12, a197\n\tv_accvgpr_read_b32 %13, a213\n\tv_accvgpr_read_b32 %14, a229\n\tv_accvgpr_read_b32 %15, a245" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_6(m) asm volatile("v_accvgpr_read_b32 %0, a6\n\tv_accvgpr_read_b32 %1, a22\n\tv_accvgpr_read_b32 %2, a38\n\tv_accvgpr_read_b32 %3, a54\n\tv_accvgpr_read_b32 %4, a70\n\tv_accvgpr_read_b32 %5, a86\n\tv_accvgpr_read_b32 %6, a102\n\tv_accvgpr_read_b32 %7, a118\n\tv_accvgpr_read_b32 %8, a134\n\tv_accvgpr_read_b32 %9, a150\n\tv_accvgpr_read_b32 %10, a166\n\tv_accvgpr_read_b32 %11, a182\n\tv_accvgpr_read_b32 %12, a198\n\tv_accvgpr_read_b32 %13, a214\n\tv_accvgpr_read_b32 %14, a230\n\tv_accvgpr_read_b32 %15, a246" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_7(m) asm volatile("v_accvgpr_read_b32 %0, a7\n\tv_accvgpr_read_b32 %1, a23\n\tv_accvgpr_read_b32 %2, a39\n\tv_accvgpr_read_b32 %3, a55\n\tv_accvgpr_read_b32 %4, a71\n\tv_accvgpr_read_b32 %5, a87\n\tv_accvgpr_read_b32 %6, a103\n\tv_accvgpr_read_b32 %7, a119\n\tv_accvgpr_read_b32 %8, a135\n\tv_accvgpr_read_b32 %9, a151\n\tv_accvgpr_read_b32 %10, a167\n\tv_accvgpr_read_b32 %11, a183\n\tv_accvgpr_read_b32 %12, a199\n\tv_accvgpr_read_b32 %13, a215\n\tv_accvgpr_read_b32 %14, a231\n\tv_accvgpr_read_b32 %15, a247" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_8(m) asm volatile("v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a24\n\tv_accvgpr_read_b32 %2, a40\n\tv_accvgpr_read_b32 %3, a56\n\tv_accvgpr_read_b32 %4, a72\n\tv_accvgpr_read_b32 %5, a88\n\tv_accvgpr_read_b32 %6, a104\n\tv_accvgpr_read_b32 %7, a120\n\tv_accvgpr_read_b32 %8, a136\n\tv_accvgpr_read_b32 %9, a152\n\tv_accvgpr_read_b32 %10, a168\n\tv_accvgpr_read_b32 %11, a184\n\tv_accvgpr_read_b32 %12, a200\n\tv_accvgpr_read_b32 %13, a216\n\tv_accvgpr_read_b32 %14, a232\n\tv_accvgpr_read_b32 %15, a248" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_9(m) asm volatile("v_accvgpr_read_b32 %0, a9\n\tv_accvgpr_read_b32 %1, a25\n\tv_accvgpr_read_b32 %2, a41\n\tv_accvgpr_read_b32 %3, a57\n\tv_accvgpr_read_b32 %4, a73\n\tv_accvgpr_read_b32 %5, a89\n\tv_accvgpr_read_b32 %6, a105\n\tv_accvgpr_read_b32 %7, a121\n\tv_accvgpr_read_b32 %8, a137\n\tv_accvgpr_read_b32 %9, a153\n\tv_accvgpr_read_b32 %10, a169\n\tv_accvgpr_read_b32 %11, a185\n\tv_accvgpr_read_b32 %12, a201\n\tv_accvgpr_read_b32 %13, a217\n\tv_accvgpr_read_b32 %14, a233\n\tv_accvgpr_read_b32 %15, a249" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_10(m) asm volatile("v_accvgpr_read_b32 %0, a10\n\tv_accvgpr_read_b32 %1, a26\n\tv_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a58\n\tv_accvgpr_read_b32 %4, a74\n\tv_accvgpr_read_b32 %5, a90\n\tv_accvgpr_read_b32 %6, a106\n\tv_accvgpr_read_b32 %7, a122\n\tv_accvgpr_read_b32 %8, a138\n\tv_accvgpr_read_b32 %9, a154\n\tv_accvgpr_read_b32 %10, a170\n\tv_accvgpr_read_b32 %11, a186\n\tv_accvgpr_read_b32 %12, a202\n\tv_accvgpr_read_b32 %13, a218\n\tv_accvgpr_read_b32 %14, a234\n\tv_accvgpr_read_b32 %15, a250" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_11(m) asm volatile("v_accvgpr_read_b32 %0, a11\n\tv_accvgpr_read_b32 %1, a27\n\tv_accvgpr_read_b32 %2, a43\n\tv_accvgpr_read_b32 %3, a59\n\tv_accvgpr_read_b32 %4, a75\n\tv_accvgpr_read_b32 %5, a91\n\tv_accvgpr_read_b32 %6, a107\n\tv_accvgpr_read_b32 %7, a123\n\tv_accvgpr_read_b32 %8, a139\n\tv_accvgpr_read_b32 %9, a155\n\tv_accvgpr_read_b32 %10, a171\n\tv_accvgpr_read_b32 %11, a187\n\tv_accvgpr_read_b32 %12, a203\n\tv_accvgpr_read_b32 %13, a219\n\tv_accvgpr_read_b32 %14, a235\n\tv_accvgpr_read_b32 %15, a251" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_12(m) asm volatile("v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a28\n\tv_accvgpr_read_b32 %2, a44\n\tv_accvgpr_read_b32 %3, a60\n\tv_accvgpr_read_b32 %4, a76\n\tv_accvgpr_read_b32 %5, a92\n\tv_accvgpr_read_b32 %6, a108\n\tv_accvgpr_read_b32 %7, a124\n\tv_accvgpr_read_b32 %8, a140\n\tv_accvgpr_read_b32 %9, a156\n\tv_accvgpr_read_b32 %10, a172\n\tv_accvgpr_read_b32 %11, a188\n\tv_accvgpr_read_b32 %12, a204\n\tv_accvgpr_read_b32 %13, a220\n\tv_accvgpr_read_b32 %14, a236\n\tv_accvgpr_read_b32 %15, a252" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_13(m) asm volatile("v_accvgpr_read_b32 %0, a13\n\tv_accvgpr_read_b32 %1, a29\n\tv_accvgpr_read_b32 %2, a45\n\tv_accvgpr_read_b32 %3, a61\n\tv_accvgpr_read_b32 %4, a77\n\tv_accvgpr_read_b32 %5, a93\n\tv_accvgpr_read_b32 %6, a109\n\tv_accvgpr_read_b32 %7, a125\n\tv_accvgpr_read_b32 %8, a141\n\tv_accvgpr_read_b32 %9, a157\n\tv_accvgpr_read_b32 %10, a173\n\tv_accvgpr_read_b32 %11, a189\n\tv_accvgpr_read_b32 %12, a205\n\tv_accvgpr_read_b32 %13, a221\n\tv_accvgpr_read_b32 %14, a237\n\tv_accvgpr_read_b32 %15, a253" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_14(m) asm volatile("v_accvgpr_read_b32 %0, a14\n\tv_accvgpr_read_b32 %1, a30\n\tv_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a62\n\tv_accvgpr_read_b32 %4, a78\n\tv_accvgpr_read_b32 %5, a94\n\tv_accvgpr_read_b32 %6, a110\n\tv_accvgpr_read_b32 %7, a126\n\tv_accvgpr_read_b32 %8, a142\n\tv_accvgpr_read_b32 %9, a158\n\tv_accvgpr_read_b32 %10, a174\n\tv_accvgpr_read_b32 %11, a190\n\tv_accvgpr_read_b32 %12, a206\n\tv_accvgpr_read_b32 %13, a222\n\tv_accvgpr_read_b32 %14, a238\n\tv_accvgpr_read_b32 %15, a254" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define W1_RD_15(m) asm volatile("v_accvgpr_read_b32 %0, a15\n\tv_accvgpr_read_b32 %1, a31\n\tv_accvgpr_read_b32 %2, a47\n\tv_accvgpr_read_b32 %3, a63\n\tv_accvgpr_read_b32 %4, a79\n\tv_accvgpr_read_b32 %5, a95\n\tv_accvgpr_read_b32 %6, a111\n\tv_accvgpr_read_b32 %7, a127\n\tv_accvgpr_read_b32 %8, a143\n\tv_accvgpr_read_b32 %9, a159\n\tv_accvgpr_read_b32 %10, a175\n\tv_accvgpr_read_b32 %11, a191\n\tv_accvgpr_read_b32 %12, a207\n\tv_accvgpr_read_b32 %13, a223\n\tv_accvgpr_read_b32 %14, a239\n\tv_accvgpr_read_b32 %15, a255" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
    float s1[16], s2[16];
    // hb: with / without a conv bias, decided ONCE outside (the two epilogues are separate code).  With the bias load behind a
    // branch inside the per-channel code the compiler's s_waitcnt vmcnt(0) for it sat in the common path: every channel waited for
    // the previous channel's stores to be acknowledged, 16 round trips = 3.9 us of a 29 us unit on the 64-channel layers
    // (tools/attic/diag_wg_timing.py); without it 2.7 us.
    auto out_e = [&](auto hb, int e, const float (&m)[16]) {       // m[4 i + j] = M[i][j] of output channel element e
        float r_[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r_[i][0] = m[i * 4 + 0] + m[i * 4 + 1] + m[i * 4 + 2];
            r_[i][1] = m[i * 4 + 1] - m[i * 4 + 2] - m[i * 4 + 3];
        }
        const int co = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float bv = 0.0f;
        if constexpr (decltype(hb)::value) bv = bias[co < g.M ? co : 0];
        float v00 = r_[0][0] + r_[1][0] + r_[2][0] + bv, v01 = r_[0][1] + r_[1][1] + r_[2][1] + bv;
        float v10 = r_[1][0] - r_[2][0] - r_[3][0] + bv, v11 = r_[1][1] - r_[2][1] - r_[3][1] + bv;
        if (BNE) {                             // y = [max(0,] (conv + bias - mean) * invstd * gamma + beta [)]
            const int cc = co < g.M ? co : 0;
            const float mu = bn.mean[cc], is = 1.0f / sqrtf(bn.var[cc] + bn.eps), ga = bn.gamma[cc], be = bn.beta[cc];
            v00 = (v00 - mu) * is * ga + be, v01 = (v01 - mu) * is * ga + be;
            v10 = (v10 - mu) * is * ga + be, v11 = (v11 - mu) * is * ga + be;
            if (bn.relu) v00 = fmaxf(v00, 0.0f), v01 = fmaxf(v01, 0.0f), v10 = fmaxf(v10, 0.0f), v11 = fmaxf(v11, 0.0f);
        }
        if (tv && co < g.M) {
            f32x2 o;
            o[0] = v00, o[1] = v01;
            *reinterpret_cast<f32x2 *>(yout + (int64_t)co * HW) = o;
            o[0] = v10, o[1] = v11;
            *reinterpret_cast<f32x2 *>(yout + (int64_t)co * HW + g.W) = o;
        }
        if (STATS) {
            s1[e] = tv ? (v00 + v01) + (v10 + v11) : 0.0f;
            s2[e] = tv ? (v00 * v00 + v01 * v01) + (v10 * v10 + v11 * v11) : 0.0f;
        }
    };
    auto out_all = [&](auto hb) {
        float m[16];
        W1_RD_0(m); out_e(hb, 0, m);
        W1_RD_1(m); out_e(hb, 1, m);
        W1_RD_2(m); out_e(hb, 2, m);
        W1_RD_3(m); out_e(hb, 3, m);
        W1_RD_4(m); out_e(hb, 4, m);
        W1_RD_5(m); out_e(hb, 5, m);
        W1_RD_6(m); out_e(hb, 6, m);
        W1_RD_7(m); out_e(hb, 7, m);
        W1_RD_8(m); out_e(hb, 8, m);
        W1_RD_9(m); out_e(hb, 9, m);
        W1_RD_10(m); out_e(hb, 10, m);
        W1_RD_11(m); out_e(hb, 11, m);
        W1_RD_12(m); out_e(hb, 12, m);
        W1_RD_13(m); out_e(hb, 13, m);
        W1_RD_14(m); out_e(hb, 14, m);
        W1_RD_15(m); out_e(hb, 15, m);
    };
    if (bias != nullptr) out_all(std::true_type{}); else out_all(std::false_type{});
    if (STATS) {
#pragma unroll
        for (int e = 0; e < 16; e += 8) half_wave_sum8(s1 + e), half_wave_sum8(s2 + e);
        if (li == kHalfSumLane) {
            const unsigned nruns = (ttot + W1_T - 1) / W1_T;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (co < g.M) {
                    float *dst = stats + ((int64_t)co * nruns + run) * 2;
                    f32x2 o;
                    o[0] = s1[e], o[1] = s2[e];
                    *reinterpret_cast<f32x2 *>(dst) = o;
                }
            }
        }
    }
    WG_STAMP(4);
    }   // next logical block
}

// ------------------------------------------------------------------------------ two waves = one unit ("k_wg2")
// k_wg1 split over the transform positions, built to test whether a SECOND wave per SIMD would fill the gaps the staging
// instructions leave in k_wg1's MFMA stream: a block is two waves that share a unit of 32 channels x 32 tiles, wave ph owns the
// positions of transform rows 2 ph, 2 ph + 1 -- 8 accumulators = 128 AGPRs, so two waves fit a SIMD (four independent blocks per
// CU).  Per chunk a wave fetches the three patch rows its positions need (ph .. ph + 2), does half of the input transform (16 adds
// per patch instead of 32: nothing is computed twice) and 16 MFMAs; the two waves meet once, in the epilogue, to swap the halves of
// the (linear) output transform through LDS.  RESULT: the same speed as k_wg1 within 1 % on every layer (34.5 vs 34.4 ms per VGG16
// pass) -- the time the other instructions take is not hidden by a second wave either; on this chip an fp32 MFMA and the fp32 vector
// adds / the memory instructions' register traffic share what they run on (the fp32 matrix and packed-vector peaks are the same
// number), so the cost is additive: MFMA time + ~4 cycles per vector add + ~8-16 per LDS / vector-memory instruction.  Kept behind
// CPG_WINO_KERNEL=pair (tested); not the default.
constexpr int W2_RAW = 4 * 3 * W1_ROW;                    // one stage of one wave: [channel][row 3][slot 34][2] = 816 floats

#define W2_ONE_0(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" : : "v"(A), "v"(B) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15")
#define W2_ONE_1(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[16:31], %0, %1, a[16:31]" : : "v"(A), "v"(B) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31")
#define W2_ONE_2(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[32:47], %0, %1, a[32:47]" : : "v"(A), "v"(B) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47")
#define W2_ONE_3(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[48:63], %0, %1, a[48:63]" : : "v"(A), "v"(B) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63")
#define W2_ONE_4(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[64:79], %0, %1, a[64:79]" : : "v"(A), "v"(B) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79")
#define W2_ONE_5(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[80:95], %0, %1, a[80:95]" : : "v"(A), "v"(B) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95")
#define W2_ONE_6(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[96:111], %0, %1, a[96:111]" : : "v"(A), "v"(B) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111")
#define W2_ONE_7(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[112:127], %0, %1, a[112:127]" : : "v"(A), "v"(B) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127")

#define W2_RD_0(m) asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a16\n\tv_accvgpr_read_b32 %2, a32\n\tv_accvgpr_read_b32 %3, a48\n\tv_accvgpr_read_b32 %4, a64\n\tv_accvgpr_read_b32 %5, a80\n\tv_accvgpr_read_b32 %6, a96\n\tv_accvgpr_read_b32 %7, a112" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_1(m) asm volatile("v_accvgpr_read_b32 %0, a1\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a33\n\tv_accvgpr_read_b32 %3, a49\n\tv_accvgpr_read_b32 %4, a65\n\tv_accvgpr_read_b32 %5, a81\n\tv_accvgpr_read_b32 %6, a97\n\tv_accvgpr_read_b32 %7, a113" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_2(m) asm volatile("v_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a18\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a50\n\tv_accvgpr_read_b32 %4, a66\n\tv_accvgpr_read_b32 %5, a82\n\tv_accvgpr_read_b32 %6, a98\n\tv_accvgpr_read_b32 %7, a114" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_3(m) asm volatile("v_accvgpr_read_b32 %0, a3\n\tv_accvgpr_read_b32 %1, a19\n\tv_accvgpr_read_b32 %2, a35\n\tv_accvgpr_read_b32 %3, a51\n\tv_accvgpr_read_b32 %4, a67\n\tv_accvgpr_read_b32 %5, a83\n\tv_accvgpr_read_b32 %6, a99\n\tv_accvgpr_read_b32 %7, a115" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_4(m) asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a20\n\tv_accvgpr_read_b32 %2, a36\n\tv_accvgpr_read_b32 %3, a52\n\tv_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a84\n\tv_accvgpr_read_b32 %6, a100\n\tv_accvgpr_read_b32 %7, a116" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_5(m) asm volatile("v_accvgpr_read_b32 %0, a5\n\tv_accvgpr_read_b32 %1, a21\n\tv_accvgpr_read_b32 %2, a37\n\tv_accvgpr_read_b32 %3, a53\n\tv_accvgpr_read_b32 %4, a69\n\tv_accvgpr_read_b32 %5, a85\n\tv_accvgpr_read_b32 %6, a101\n\tv_accvgpr_read_b32 %7, a117" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_6(m) asm volatile("v_accvgpr_read_b32 %0, a6\n\tv_accvgpr_read_b32 %1, a22\n\tv_accvgpr_read_b32 %2, a38\n\tv_accvgpr_read_b32 %3, a54\n\tv_accvgpr_read_b32 %4, a70\n\tv_accvgpr_read_b32 %5, a86\n\tv_accvgpr_read_b32 %6, a102\n\tv_accvgpr_read_b32 %7, a118" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_7(m) asm volatile("v_accvgpr_read_b32 %0, a7\n\tv_accvgpr_read_b32 %1, a23\n\tv_accvgpr_read_b32 %2, a39\n\tv_accvgpr_read_b32 %3, a55\n\tv_accvgpr_read_b32 %4, a71\n\tv_accvgpr_read_b32 %5, a87\n\tv_accvgpr_read_b32 %6, a103\n\tv_accvgpr_read_b32 %7, a119" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_8(m) asm volatile("v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a24\n\tv_accvgpr_read_b32 %2, a40\n\tv_accvgpr_read_b32 %3, a56\n\tv_accvgpr_read_b32 %4, a72\n\tv_accvgpr_read_b32 %5, a88\n\tv_accvgpr_read_b32 %6, a104\n\tv_accvgpr_read_b32 %7, a120" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_9(m) asm volatile("v_accvgpr_read_b32 %0, a9\n\tv_accvgpr_read_b32 %1, a25\n\tv_accvgpr_read_b32 %2, a41\n\tv_accvgpr_read_b32 %3, a57\n\tv_accvgpr_read_b32 %4, a73\n\tv_accvgpr_read_b32 %5, a89\n\tv_accvgpr_read_b32 %6, a105\n\tv_accvgpr_read_b32 %7, a121" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_10(m) asm volatile("v_accvgpr_read_b32 %0, a10\n\tv_accvgpr_read_b32 %1, a26\n\tv_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a58\n\tv_accvgpr_read_b32 %4, a74\n\tv_accvgpr_read_b32 %5, a90\n\tv_accvgpr_read_b32 %6, a106\n\tv_accvgpr_read_b32 %7, a122" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_11(m) asm volatile("v_accvgpr_read_b32 %0, a11\n\tv_accvgpr_read_b32 %1, a27\n\tv_accvgpr_read_b32 %2, a43\n\tv_accvgpr_read_b32 %3, a59\n\tv_accvgpr_read_b32 %4, a75\n\tv_accvgpr_read_b32 %5, a91\n\tv_accvgpr_read_b32 %6, a107\n\tv_accvgpr_read_b32 %7, a123" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_12(m) asm volatile("v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a28\n\tv_accvgpr_read_b32 %2, a44\n\tv_accvgpr_read_b32 %3, a60\n\tv_accvgpr_read_b32 %4, a76\n\tv_accvgpr_read_b32 %5, a92\n\tv_accvgpr_read_b32 %6, a108\n\tv_accvgpr_read_b32 %7, a124" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_13(m) asm volatile("v_accvgpr_read_b32 %0, a13\n\tv_accvgpr_read_b32 %1, a29\n\tv_accvgpr_read_b32 %2, a45\n\tv_accvgpr_read_b32 %3, a61\n\tv_accvgpr_read_b32 %4, a77\n\tv_accvgpr_read_b32 %5, a93\n\tv_accvgpr_read_b32 %6, a109\n\tv_accvgpr_read_b32 %7, a125" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_14(m) asm volatile("v_accvgpr_read_b32 %0, a14\n\tv_accvgpr_read_b32 %1, a30\n\tv_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a62\n\tv_accvgpr_read_b32 %4, a78\n\tv_accvgpr_read_b32 %5, a94\n\tv_accvgpr_read_b32 %6, a110\n\tv_accvgpr_read_b32 %7, a126" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W2_RD_15(m) asm volatile("v_accvgpr_read_b32 %0, a15\n\tv_accvgpr_read_b32 %1, a31\n\tv_accvgpr_read_b32 %2, a47\n\tv_accvgpr_read_b32 %3, a63\n\tv_accvgpr_read_b32 %4, a79\n\tv_accvgpr_read_b32 %5, a95\n\tv_accvgpr_read_b32 %6, a111\n\tv_accvgpr_read_b32 %7, a127" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))

template <bool DGRAD, bool STATS, bool BNE = false>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_wg2(WgGeom g, const float *__restrict__ x, const float *__restrict__ up, const float *__restrict__ bias,
           float *__restrict__ y, float *__restrict__ stats, WgBnEval bn) {
    __shared__ __attribute__((aligned(16))) float smem_all[2 * 32 * 64];       // 2 x 2 raw stages (3264 floats) / the exchange buffer
    const int tid = threadIdx.x, lane = tid & 63, ph = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int HW = g.H * g.W;
    float *smem = smem_all + ph * 2 * W2_RAW;                 // this wave's private raw stages

    unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int kb = lb % g.nkb;
    const unsigned ttot = (unsigned)g.tiles_total, timg = (unsigned)g.tiles_img, twu = (unsigned)g.tw;
    const unsigned run = lb / g.nkb;                          // tile run of 32
    const unsigned t0 = run * W1_T;
    if (t0 >= ttot) return;                                   // (uniform for the block)
    const int n0 = (int)(t0 / timg);

    constexpr int kOutOfRange = (int)0x80000000;
    int roff[3], hoff, lo, ro;
    {
        const unsigned tg = t0 + li;
        const bool tv = tg < ttot;
        const int n = (int)(tg / timg), r = (int)(tg % timg);
        const int ty = (int)((unsigned)r / twu), tx = (int)((unsigned)r % twu);
        const int cbase = ((n - n0) * g.C + 2 * lh) * HW;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int gh = 2 * ty - 1 + ph + i;
            roff[i] = (tv && (unsigned)gh < (unsigned)g.H) ? (cbase + gh * g.W + 2 * tx) * 4 : kOutOfRange;
        }
        lo = tx == 0 ? 0 : (li + 1) * 2 - 1;
        ro = tx == g.tw - 1 ? (W1_T + 1) * 2 + 1 : (li + 1) * 2 + 2;
        // halo: lanes 0-23 = (side, row, channel): the column left of tile t0 / right of tile t0 + 31
        const int side = lane >= 12 ? 1 : 0, hl = lane - 12 * side, hi = hl >> 2, hc = hl & 3;
        const unsigned th = side ? t0 + W1_T - 1 : t0;
        const int nh = (int)(th / timg), rh = (int)(th % timg);
        const int tyh = (int)((unsigned)rh / twu), txh = (int)((unsigned)rh % twu);
        const int ghh = 2 * tyh - 1 + ph + hi, gwh = side ? 2 * txh + 2 : 2 * txh - 1;
        const bool okh = lane < 24 && th < ttot && (unsigned)ghh < (unsigned)g.H && (unsigned)gwh < (unsigned)g.W;
        hoff = okh ? (((nh - n0) * g.C + hc) * HW + ghh * g.W + gwh) * 4 : kOutOfRange;
    }
    const int span = (W1_T + g.tiles_img - 1) / g.tiles_img + 1;
    const int nimg_here = min(span, g.N - n0);
    // byte offset of the LAST chunk's first channel, set once the chunk count is known: min(last chunk * 4, C - 4) channels.  Every row load
    // takes min(chunk * stride, x_last_off): chunks past the end re-read the last one (what clampc did for them), and with a channel count
    // that is no multiple of 4 the last chunk starts at C - 4, overlapping its neighbour (wg_chunk_base; the pack kernel zeroes its filters
    // for the channels the neighbour already contracted) -- one scalar multiply and one min per chunk, as before.
    int x_last_off = 0;
    const __amdgpu_buffer_rsrc_t srd_x =
        __builtin_amdgcn_make_buffer_rsrc((void *)(x + (int64_t)n0 * g.C * HW), 0, nimg_here * g.C * HW * 4, 0x00020000);
    const float *ubase = up + (int64_t)kb * g.nch * W1_U + ph * 1024 + lane * 4;     // this wave's four float4 of a chunk: q = 4 ph .. + 3

    // raw[c][row][slot][2] offsets of this lane (channels 2 lh + j)
    const int raw_own = (2 * lh) * 3 * W1_ROW + (li + 1) * 2;
    const int hside = lane >= 12 ? 1 : 0, hrem = lane - 12 * hside;
    const int halo_w = ((hrem & 3) * 3 + (hrem >> 2)) * W1_ROW + (hside ? (W1_T + 1) * 2 : 1);
    if (lane < 24) {                                          // zero slots of the 2 x 12 rows (never written)
        smem[lane * W1_ROW] = 0.0f;
        smem[lane * W1_ROW + (W1_T + 1) * 2 + 1] = 0.0f;
    }

    struct Rows {
        i32x2 r[2][3];
        float halo;
    };
    auto G_row1 = [&](int ch, Rows &q, int k) {             // k = 0..5: row k % 3 of channel 2 lh + k / 3;  k = 6: the halo values
        const int soff = min(ch * (WG_CK * HW * 4), x_last_off);      // (see x_last_off)
        if (k < 6)
            q.r[k / 3][k % 3] = __builtin_amdgcn_raw_buffer_load_b64(srd_x, roff[k % 3], soff + (k / 3) * HW * 4, 0);
        else
            q.halo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, hoff, soff, 0));
    };
    auto W_row1 = [&](int stage, const Rows &q, int k) {
        float *raw = smem + stage * W2_RAW;
        if (k < 6)
            *reinterpret_cast<i32x2 *>(raw + raw_own + k * W1_ROW) = q.r[k / 3][k % 3];
        else if (lane < 24)
            raw[halo_w] = q.halo;
    };
    auto G_u1 = [&](int ch, f32x4 (&u)[4], int q) { u[q] = *reinterpret_cast<const f32x4 *>(ubase + (int64_t)ch * W1_U + q * 256); };
    // half patch of (tile li, channel 2 lh + j): rows ph .. ph + 2 -> the 8 values of transform rows 2 ph, 2 ph + 1 in d[0..7]
    auto T_read1 = [&](int stage, int j, float (&d)[12], int i) {
        const float *raw = smem + stage * W2_RAW + ((2 * lh + j) * 3 + i) * W1_ROW;
        const f32x2 own = *reinterpret_cast<const f32x2 *>(raw + (li + 1) * 2);
        d[i * 4 + 0] = raw[lo], d[i * 4 + 1] = own[0], d[i * 4 + 2] = own[1], d[i * 4 + 3] = raw[ro];
    };
    // B^T d: ph = 0 holds patch rows 0, 1, 2 -> rows 0, 1 = d0 - d2, d1 + d2;  ph = 1 holds 1, 2, 3 -> rows 2, 3 = d2 - d1, d1 - d3
    auto T_col = [&](float (&d)[12], int j0) {
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) {
            const float e0 = d[0 * 4 + j], e1 = d[1 * 4 + j], e2 = d[2 * 4 + j];
            d[0 * 4 + j] = ph ? e1 - e0 : e0 - e2;
            d[1 * 4 + j] = ph ? e0 - e2 : e1 + e2;
            asm volatile("" : "+v"(d[0 * 4 + j]), "+v"(d[1 * 4 + j]));
        }
    };
    auto T_rowp = [&](float (&d)[12], int i) {
        const float t0_ = d[i * 4 + 0], t1 = d[i * 4 + 1], t2 = d[i * 4 + 2], t3 = d[i * 4 + 3];
        d[i * 4 + 0] = t0_ - t2, d[i * 4 + 1] = t1 + t2, d[i * 4 + 2] = t2 - t1, d[i * 4 + 3] = t1 - t3;
        asm volatile("" : "+v"(d[i * 4 + 0]), "+v"(d[i * 4 + 1]), "+v"(d[i * 4 + 2]), "+v"(d[i * 4 + 3]));
    };

    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0" : : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    asm volatile("v_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0" : : : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    asm volatile("v_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0" : : : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    asm volatile("v_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0" : : : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    asm volatile("v_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0" : : : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    asm volatile("v_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0" : : : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    asm volatile("v_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0" : : : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    asm volatile("v_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" : : : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");

    int nch = g.nch;
    if (BNE && bn.live != nullptr) {                          // inference: skip what apply_mask killed (wave-uniform decisions)
        const int alive = (kb * 32 + li < g.M) ? bn.live[kb * 32 + li] : 0;
        const bool dead = __ballot(alive != 0) == 0ull;
        int lastc = g.nch;
        while (lastc > 0 && bn.live[bn.Mp + 4 + lastc - 1] == 0) --lastc;
        nch = dead ? 0 : lastc;
        if (lane == 0 && ph == 0) {
            if (dead) atomicAdd(&bn.live[bn.Mp + 1], 1);
            if (blockIdx.x == 0) bn.live[bn.Mp] = lastc * WG_CK;
        }
    }
    const int last = nch - 1;
    x_last_off = __builtin_amdgcn_readfirstlane(min(last * WG_CK, g.C - WG_CK) * HW * 4);
    auto clampc = [&](int c) { return min(c, last); };
    f32x4 ua[4], ub[4];                    // U of the current / next chunk
    float c0[12], c1[12], x0[12], x1[12];  // B operands ([0..7]) of the current chunk for channels 2 lh, 2 lh + 1 / the next chunk's, being transformed
    Rows rows;
    if (nch > 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) G_row1(0, rows, k);
#pragma unroll
        for (int q = 0; q < 4; ++q) G_u1(0, ua, q);
#pragma unroll
        for (int k = 0; k < 7; ++k) W_row1(0, rows, k);
#pragma unroll
        for (int k = 0; k < 7; ++k) G_row1(clampc(1), rows, k);
#pragma unroll
        for (int i = 0; i < 3; ++i) { T_read1(0, 0, c0, i); T_read1(0, 1, c1, i); }
        T_col(c0, 0); T_col(c0, 2); T_col(c1, 0); T_col(c1, 2);
        T_rowp(c0, 0); T_rowp(c0, 1); T_rowp(c1, 0); T_rowp(c1, 1);
#pragma unroll
        for (int k = 0; k < 7; ++k) W_row1(1, rows, k);
#pragma unroll
        for (int k = 0; k < 7; ++k) G_row1(clampc(2), rows, k);
    }
#define W2_SLOT(p, h, U, B, work)                                                      \
    W2_ONE_##p((U)[(p) >> 1][((p) & 1) * 2 + (h)], (B)[p]);                            \
    work;                                                                              \
    __builtin_amdgcn_sched_barrier(0)
    // iteration it (par = it & 1): M(it); T(it + 1) from raw stage (it + 1) & 1; U(it + 1) requested; W(it + 2) stores the rows
    // requested one iteration ago, whose registers then take the loads of chunk it + 3
    auto iter = [&](int it, int par, f32x4 (&ucur)[4], f32x4 (&unext)[4], float (&b0)[12], float (&b1)[12], float (&n0v)[12], float (&n1v)[12]) {
        const int cu = clampc(it + 1), cr = it + 3;
        W2_SLOT(0, 0, ucur, b0, T_read1(par ^ 1, 0, n0v, 0); T_read1(par ^ 1, 0, n0v, 1));
        W2_SLOT(0, 1, ucur, b1, T_read1(par ^ 1, 0, n0v, 2); T_read1(par ^ 1, 1, n1v, 0));
        W2_SLOT(1, 0, ucur, b0, T_read1(par ^ 1, 1, n1v, 1); T_read1(par ^ 1, 1, n1v, 2));
        W2_SLOT(1, 1, ucur, b1, G_u1(cu, unext, 0));
        W2_SLOT(2, 0, ucur, b0, T_col(n0v, 0); G_u1(cu, unext, 1));
        W2_SLOT(2, 1, ucur, b1, T_col(n0v, 2); G_u1(cu, unext, 2));
        W2_SLOT(3, 0, ucur, b0, T_col(n1v, 0); G_u1(cu, unext, 3));
        W2_SLOT(3, 1, ucur, b1, T_col(n1v, 2));
        W2_SLOT(4, 0, ucur, b0, T_rowp(n0v, 0));
        W2_SLOT(4, 1, ucur, b1, T_rowp(n0v, 1));
        W2_SLOT(5, 0, ucur, b0, T_rowp(n1v, 0));
        W2_SLOT(5, 1, ucur, b1, T_rowp(n1v, 1));
        W2_SLOT(6, 0, ucur, b0, W_row1(par, rows, 0); W_row1(par, rows, 1); G_row1(cr, rows, 0));
        W2_SLOT(6, 1, ucur, b1, W_row1(par, rows, 2); W_row1(par, rows, 3); G_row1(cr, rows, 1); G_row1(cr, rows, 2));
        W2_SLOT(7, 0, ucur, b0, W_row1(par, rows, 4); W_row1(par, rows, 5); G_row1(cr, rows, 3); G_row1(cr, rows, 4));
        W2_SLOT(7, 1, ucur, b1, W_row1(par, rows, 6); G_row1(cr, rows, 5); G_row1(cr, rows, 6));
    };
    for (int it = 0; it < nch; it += 2) {
        iter(it, 0, ua, ub, c0, c1, x0, x1);
        if (it + 1 < nch) iter(it + 1, 1, ub, ua, x0, x1, c0, c1);
    }

    // ---- epilogue: this wave's 8 positions (transform rows i = 2 ph, 2 ph + 1) -> partial 2x2 outputs; the output transform is linear
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float own0[16], own1[16];                  // the output row this wave finishes (a = ph), columns 0 / 1, per accumulator element
    __syncthreads();                           // both waves are done with their raw stages: the LDS becomes the exchange buffer
    float *xch = smem_all;
    auto part_e = [&](int e, const float (&m)[8]) {
        const float r00 = m[0] + m[1] + m[2], r01 = m[1] - m[2] - m[3];       // R[il][b] = sum_j A^T[b][j] M[i][j]
        const float r10 = m[4] + m[5] + m[6], r11 = m[5] - m[6] - m[7];
        float g0, g1;
        if (ph == 0) {                         // i = 0, 1:  Y0 += R0 + R1 (own),  Y1 += R1 (given to the other wave)
            own0[e] = r00 + r10, own1[e] = r01 + r11, g0 = r10, g1 = r11;
        } else {                               // i = 2, 3:  Y1 += -R2 - R3 (own),  Y0 += R2 (given)
            own0[e] = -r00 - r10, own1[e] = -r01 - r11, g0 = r00, g1 = r01;
        }
        xch[((ph * 2 + 0) * 16 + e) * 64 + lane] = g0;
        xch[((ph * 2 + 1) * 16 + e) * 64 + lane] = g1;
    };
    {
        float m[8];
        W2_RD_0(m); part_e(0, m);
        W2_RD_1(m); part_e(1, m);
        W2_RD_2(m); part_e(2, m);
        W2_RD_3(m); part_e(3, m);
        W2_RD_4(m); part_e(4, m);
        W2_RD_5(m); part_e(5, m);
        W2_RD_6(m); part_e(6, m);
        W2_RD_7(m); part_e(7, m);
        W2_RD_8(m); part_e(8, m);
        W2_RD_9(m); part_e(9, m);
        W2_RD_10(m); part_e(10, m);
        W2_RD_11(m); part_e(11, m);
        W2_RD_12(m); part_e(12, m);
        W2_RD_13(m); part_e(13, m);
        W2_RD_14(m); part_e(14, m);
        W2_RD_15(m); part_e(15, m);

    }
    __syncthreads();
    const unsigned tg = t0 + li;
    const bool tv = tg < ttot;
    const int n = (int)(tg / timg), r = (int)(tg % timg);
    const int ty = (int)((unsigned)r / twu), tx = (int)((unsigned)r % twu);
    float *yout = y + ((int64_t)n * g.M) * HW + (2 * ty + ph) * g.W + 2 * tx;
    float s1[16], s2[16];
    auto out_all = [&](auto hb) {                // (hb: with / without a conv bias -- two separate epilogues, see k_wg1)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int co = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float v0 = own0[e] + xch[(((ph ^ 1) * 2 + 0) * 16 + e) * 64 + lane];
        float v1 = own1[e] + xch[(((ph ^ 1) * 2 + 1) * 16 + e) * 64 + lane];
        if constexpr (decltype(hb)::value) {
            const float bv = bias[co < g.M ? co : 0];
            v0 += bv, v1 += bv;
        }
        if (BNE) {                             // y = [max(0,] (conv + bias - mean) * invstd * gamma + beta [)]
            const int cc = co < g.M ? co : 0;
            const float mu = bn.mean[cc], is = 1.0f / sqrtf(bn.var[cc] + bn.eps), ga = bn.gamma[cc], be = bn.beta[cc];
            v0 = (v0 - mu) * is * ga + be, v1 = (v1 - mu) * is * ga + be;
            if (bn.relu) v0 = fmaxf(v0, 0.0f), v1 = fmaxf(v1, 0.0f);
        }
        if (tv && co < g.M) {
            f32x2 o;
            o[0] = v0, o[1] = v1;
            *reinterpret_cast<f32x2 *>(yout + (int64_t)co * HW) = o;
        }
        if (STATS) {
            s1[e] = tv ? v0 + v1 : 0.0f;
            s2[e] = tv ? v0 * v0 + v1 * v1 : 0.0f;
        }
    }
    };
    if (bias != nullptr) out_all(std::true_type{}); else out_all(std::false_type{});
    if (STATS) {                               // every wave is its own statistics tile: stats[k][2 run + ph][2]
#pragma unroll
        for (int e = 0; e < 16; e += 8) half_wave_sum8(s1 + e), half_wave_sum8(s2 + e);
        if (li == kHalfSumLane) {
            const unsigned ntile = 2 * ((ttot + W1_T - 1) / W1_T);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (co < g.M) {
                    float *dst = stats + ((int64_t)co * ntile + 2 * run + ph) * 2;
                    f32x2 o;
                    o[0] = s1[e], o[1] = s2[e];
                    *reinterpret_cast<f32x2 *>(dst) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------ two waves = one unit of 64 channels ("k_wg3")
// k_wg2 with 64 output channels per wave: the transformed input (whose adds and LDS round trip are paid in MFMA time, see k_wg2) feeds
// twice the MFMAs.  Wave ph owns the positions of transform rows 2 ph, 2 ph + 1 for the channel halves kq = 0, 1: 16 accumulators =
// 256 AGPRs, one wave per SIMD, two 2-wave blocks per CU.  Per chunk: 32 MFMAs against 32 (not 64) transform adds, 7 (not 9) row
// loads and 8 float4 of U.
#define W3_ONE_0(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" : : "v"(A), "v"(B) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15")
#define W3_ONE_1(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[16:31], %0, %1, a[16:31]" : : "v"(A), "v"(B) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31")
#define W3_ONE_2(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[32:47], %0, %1, a[32:47]" : : "v"(A), "v"(B) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47")
#define W3_ONE_3(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[48:63], %0, %1, a[48:63]" : : "v"(A), "v"(B) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63")
#define W3_ONE_4(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[64:79], %0, %1, a[64:79]" : : "v"(A), "v"(B) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79")
#define W3_ONE_5(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[80:95], %0, %1, a[80:95]" : : "v"(A), "v"(B) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95")
#define W3_ONE_6(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[96:111], %0, %1, a[96:111]" : : "v"(A), "v"(B) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111")
#define W3_ONE_7(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[112:127], %0, %1, a[112:127]" : : "v"(A), "v"(B) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127")
#define W3_ONE_8(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[128:143], %0, %1, a[128:143]" : : "v"(A), "v"(B) : "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143")
#define W3_ONE_9(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[144:159], %0, %1, a[144:159]" : : "v"(A), "v"(B) : "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159")
#define W3_ONE_10(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[160:175], %0, %1, a[160:175]" : : "v"(A), "v"(B) : "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175")
#define W3_ONE_11(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[176:191], %0, %1, a[176:191]" : : "v"(A), "v"(B) : "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191")
#define W3_ONE_12(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[192:207], %0, %1, a[192:207]" : : "v"(A), "v"(B) : "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207")
#define W3_ONE_13(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[208:223], %0, %1, a[208:223]" : : "v"(A), "v"(B) : "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223")
#define W3_ONE_14(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[224:239], %0, %1, a[224:239]" : : "v"(A), "v"(B) : "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239")
#define W3_ONE_15(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[240:255], %0, %1, a[240:255]" : : "v"(A), "v"(B) : "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255")

#define W3_RD_0_0(m) asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a16\n\tv_accvgpr_read_b32 %2, a32\n\tv_accvgpr_read_b32 %3, a48\n\tv_accvgpr_read_b32 %4, a64\n\tv_accvgpr_read_b32 %5, a80\n\tv_accvgpr_read_b32 %6, a96\n\tv_accvgpr_read_b32 %7, a112" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_1(m) asm volatile("v_accvgpr_read_b32 %0, a1\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a33\n\tv_accvgpr_read_b32 %3, a49\n\tv_accvgpr_read_b32 %4, a65\n\tv_accvgpr_read_b32 %5, a81\n\tv_accvgpr_read_b32 %6, a97\n\tv_accvgpr_read_b32 %7, a113" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_2(m) asm volatile("v_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a18\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a50\n\tv_accvgpr_read_b32 %4, a66\n\tv_accvgpr_read_b32 %5, a82\n\tv_accvgpr_read_b32 %6, a98\n\tv_accvgpr_read_b32 %7, a114" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_3(m) asm volatile("v_accvgpr_read_b32 %0, a3\n\tv_accvgpr_read_b32 %1, a19\n\tv_accvgpr_read_b32 %2, a35\n\tv_accvgpr_read_b32 %3, a51\n\tv_accvgpr_read_b32 %4, a67\n\tv_accvgpr_read_b32 %5, a83\n\tv_accvgpr_read_b32 %6, a99\n\tv_accvgpr_read_b32 %7, a115" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_4(m) asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a20\n\tv_accvgpr_read_b32 %2, a36\n\tv_accvgpr_read_b32 %3, a52\n\tv_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a84\n\tv_accvgpr_read_b32 %6, a100\n\tv_accvgpr_read_b32 %7, a116" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_5(m) asm volatile("v_accvgpr_read_b32 %0, a5\n\tv_accvgpr_read_b32 %1, a21\n\tv_accvgpr_read_b32 %2, a37\n\tv_accvgpr_read_b32 %3, a53\n\tv_accvgpr_read_b32 %4, a69\n\tv_accvgpr_read_b32 %5, a85\n\tv_accvgpr_read_b32 %6, a101\n\tv_accvgpr_read_b32 %7, a117" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_6(m) asm volatile("v_accvgpr_read_b32 %0, a6\n\tv_accvgpr_read_b32 %1, a22\n\tv_accvgpr_read_b32 %2, a38\n\tv_accvgpr_read_b32 %3, a54\n\tv_accvgpr_read_b32 %4, a70\n\tv_accvgpr_read_b32 %5, a86\n\tv_accvgpr_read_b32 %6, a102\n\tv_accvgpr_read_b32 %7, a118" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_7(m) asm volatile("v_accvgpr_read_b32 %0, a7\n\tv_accvgpr_read_b32 %1, a23\n\tv_accvgpr_read_b32 %2, a39\n\tv_accvgpr_read_b32 %3, a55\n\tv_accvgpr_read_b32 %4, a71\n\tv_accvgpr_read_b32 %5, a87\n\tv_accvgpr_read_b32 %6, a103\n\tv_accvgpr_read_b32 %7, a119" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_8(m) asm volatile("v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a24\n\tv_accvgpr_read_b32 %2, a40\n\tv_accvgpr_read_b32 %3, a56\n\tv_accvgpr_read_b32 %4, a72\n\tv_accvgpr_read_b32 %5, a88\n\tv_accvgpr_read_b32 %6, a104\n\tv_accvgpr_read_b32 %7, a120" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_9(m) asm volatile("v_accvgpr_read_b32 %0, a9\n\tv_accvgpr_read_b32 %1, a25\n\tv_accvgpr_read_b32 %2, a41\n\tv_accvgpr_read_b32 %3, a57\n\tv_accvgpr_read_b32 %4, a73\n\tv_accvgpr_read_b32 %5, a89\n\tv_accvgpr_read_b32 %6, a105\n\tv_accvgpr_read_b32 %7, a121" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_10(m) asm volatile("v_accvgpr_read_b32 %0, a10\n\tv_accvgpr_read_b32 %1, a26\n\tv_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a58\n\tv_accvgpr_read_b32 %4, a74\n\tv_accvgpr_read_b32 %5, a90\n\tv_accvgpr_read_b32 %6, a106\n\tv_accvgpr_read_b32 %7, a122" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_11(m) asm volatile("v_accvgpr_read_b32 %0, a11\n\tv_accvgpr_read_b32 %1, a27\n\tv_accvgpr_read_b32 %2, a43\n\tv_accvgpr_read_b32 %3, a59\n\tv_accvgpr_read_b32 %4, a75\n\tv_accvgpr_read_b32 %5, a91\n\tv_accvgpr_read_b32 %6, a107\n\tv_accvgpr_read_b32 %7, a123" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_12(m) asm volatile("v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a28\n\tv_accvgpr_read_b32 %2, a44\n\tv_accvgpr_read_b32 %3, a60\n\tv_accvgpr_read_b32 %4, a76\n\tv_accvgpr_read_b32 %5, a92\n\tv_accvgpr_read_b32 %6, a108\n\tv_accvgpr_read_b32 %7, a124" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_13(m) asm volatile("v_accvgpr_read_b32 %0, a13\n\tv_accvgpr_read_b32 %1, a29\n\tv_accvgpr_read_b32 %2, a45\n\tv_accvgpr_read_b32 %3, a61\n\tv_accvgpr_read_b32 %4, a77\n\tv_accvgpr_read_b32 %5, a93\n\tv_accvgpr_read_b32 %6, a109\n\tv_accvgpr_read_b32 %7, a125" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_14(m) asm volatile("v_accvgpr_read_b32 %0, a14\n\tv_accvgpr_read_b32 %1, a30\n\tv_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a62\n\tv_accvgpr_read_b32 %4, a78\n\tv_accvgpr_read_b32 %5, a94\n\tv_accvgpr_read_b32 %6, a110\n\tv_accvgpr_read_b32 %7, a126" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_0_15(m) asm volatile("v_accvgpr_read_b32 %0, a15\n\tv_accvgpr_read_b32 %1, a31\n\tv_accvgpr_read_b32 %2, a47\n\tv_accvgpr_read_b32 %3, a63\n\tv_accvgpr_read_b32 %4, a79\n\tv_accvgpr_read_b32 %5, a95\n\tv_accvgpr_read_b32 %6, a111\n\tv_accvgpr_read_b32 %7, a127" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_0(m) asm volatile("v_accvgpr_read_b32 %0, a128\n\tv_accvgpr_read_b32 %1, a144\n\tv_accvgpr_read_b32 %2, a160\n\tv_accvgpr_read_b32 %3, a176\n\tv_accvgpr_read_b32 %4, a192\n\tv_accvgpr_read_b32 %5, a208\n\tv_accvgpr_read_b32 %6, a224\n\tv_accvgpr_read_b32 %7, a240" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_1(m) asm volatile("v_accvgpr_read_b32 %0, a129\n\tv_accvgpr_read_b32 %1, a145\n\tv_accvgpr_read_b32 %2, a161\n\tv_accvgpr_read_b32 %3, a177\n\tv_accvgpr_read_b32 %4, a193\n\tv_accvgpr_read_b32 %5, a209\n\tv_accvgpr_read_b32 %6, a225\n\tv_accvgpr_read_b32 %7, a241" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_2(m) asm volatile("v_accvgpr_read_b32 %0, a130\n\tv_accvgpr_read_b32 %1, a146\n\tv_accvgpr_read_b32 %2, a162\n\tv_accvgpr_read_b32 %3, a178\n\tv_accvgpr_read_b32 %4, a194\n\tv_accvgpr_read_b32 %5, a210\n\tv_accvgpr_read_b32 %6, a226\n\tv_accvgpr_read_b32 %7, a242" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_3(m) asm volatile("v_accvgpr_read_b32 %0, a131\n\tv_accvgpr_read_b32 %1, a147\n\tv_accvgpr_read_b32 %2, a163\n\tv_accvgpr_read_b32 %3, a179\n\tv_accvgpr_read_b32 %4, a195\n\tv_accvgpr_read_b32 %5, a211\n\tv_accvgpr_read_b32 %6, a227\n\tv_accvgpr_read_b32 %7, a243" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_4(m) asm volatile("v_accvgpr_read_b32 %0, a132\n\tv_accvgpr_read_b32 %1, a148\n\tv_accvgpr_read_b32 %2, a164\n\tv_accvgpr_read_b32 %3, a180\n\tv_accvgpr_read_b32 %4, a196\n\tv_accvgpr_read_b32 %5, a212\n\tv_accvgpr_read_b32 %6, a228\n\tv_accvgpr_read_b32 %7, a244" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_5(m) asm volatile("v_accvgpr_read_b32 %0, a133\n\tv_accvgpr_read_b32 %1, a149\n\tv_accvgpr_read_b32 %2, a165\n\tv_accvgpr_read_b32 %3, a181\n\tv_accvgpr_read_b32 %4, a197\n\tv_accvgpr_read_b32 %5, a213\n\tv_accvgpr_read_b32 %6, a229\n\tv_accvgpr_read_b32 %7, a245" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_6(m) asm volatile("v_accvgpr_read_b32 %0, a134\n\tv_accvgpr_read_b32 %1, a150\n\tv_accvgpr_read_b32 %2, a166\n\tv_accvgpr_read_b32 %3, a182\n\tv_accvgpr_read_b32 %4, a198\n\tv_accvgpr_read_b32 %5, a214\n\tv_accvgpr_read_b32 %6, a230\n\tv_accvgpr_read_b32 %7, a246" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_7(m) asm volatile("v_accvgpr_read_b32 %0, a135\n\tv_accvgpr_read_b32 %1, a151\n\tv_accvgpr_read_b32 %2, a167\n\tv_accvgpr_read_b32 %3, a183\n\tv_accvgpr_read_b32 %4, a199\n\tv_accvgpr_read_b32 %5, a215\n\tv_accvgpr_read_b32 %6, a231\n\tv_accvgpr_read_b32 %7, a247" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_8(m) asm volatile("v_accvgpr_read_b32 %0, a136\n\tv_accvgpr_read_b32 %1, a152\n\tv_accvgpr_read_b32 %2, a168\n\tv_accvgpr_read_b32 %3, a184\n\tv_accvgpr_read_b32 %4, a200\n\tv_accvgpr_read_b32 %5, a216\n\tv_accvgpr_read_b32 %6, a232\n\tv_accvgpr_read_b32 %7, a248" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_9(m) asm volatile("v_accvgpr_read_b32 %0, a137\n\tv_accvgpr_read_b32 %1, a153\n\tv_accvgpr_read_b32 %2, a169\n\tv_accvgpr_read_b32 %3, a185\n\tv_accvgpr_read_b32 %4, a201\n\tv_accvgpr_read_b32 %5, a217\n\tv_accvgpr_read_b32 %6, a233\n\tv_accvgpr_read_b32 %7, a249" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_10(m) asm volatile("v_accvgpr_read_b32 %0, a138\n\tv_accvgpr_read_b32 %1, a154\n\tv_accvgpr_read_b32 %2, a170\n\tv_accvgpr_read_b32 %3, a186\n\tv_accvgpr_read_b32 %4, a202\n\tv_accvgpr_read_b32 %5, a218\n\tv_accvgpr_read_b32 %6, a234\n\tv_accvgpr_read_b32 %7, a250" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_11(m) asm volatile("v_accvgpr_read_b32 %0, a139\n\tv_accvgpr_read_b32 %1, a155\n\tv_accvgpr_read_b32 %2, a171\n\tv_accvgpr_read_b32 %3, a187\n\tv_accvgpr_read_b32 %4, a203\n\tv_accvgpr_read_b32 %5, a219\n\tv_accvgpr_read_b32 %6, a235\n\tv_accvgpr_read_b32 %7, a251" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_12(m) asm volatile("v_accvgpr_read_b32 %0, a140\n\tv_accvgpr_read_b32 %1, a156\n\tv_accvgpr_read_b32 %2, a172\n\tv_accvgpr_read_b32 %3, a188\n\tv_accvgpr_read_b32 %4, a204\n\tv_accvgpr_read_b32 %5, a220\n\tv_accvgpr_read_b32 %6, a236\n\tv_accvgpr_read_b32 %7, a252" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_13(m) asm volatile("v_accvgpr_read_b32 %0, a141\n\tv_accvgpr_read_b32 %1, a157\n\tv_accvgpr_read_b32 %2, a173\n\tv_accvgpr_read_b32 %3, a189\n\tv_accvgpr_read_b32 %4, a205\n\tv_accvgpr_read_b32 %5, a221\n\tv_accvgpr_read_b32 %6, a237\n\tv_accvgpr_read_b32 %7, a253" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_14(m) asm volatile("v_accvgpr_read_b32 %0, a142\n\tv_accvgpr_read_b32 %1, a158\n\tv_accvgpr_read_b32 %2, a174\n\tv_accvgpr_read_b32 %3, a190\n\tv_accvgpr_read_b32 %4, a206\n\tv_accvgpr_read_b32 %5, a222\n\tv_accvgpr_read_b32 %6, a238\n\tv_accvgpr_read_b32 %7, a254" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))
#define W3_RD_1_15(m) asm volatile("v_accvgpr_read_b32 %0, a143\n\tv_accvgpr_read_b32 %1, a159\n\tv_accvgpr_read_b32 %2, a175\n\tv_accvgpr_read_b32 %3, a191\n\tv_accvgpr_read_b32 %4, a207\n\tv_accvgpr_read_b32 %5, a223\n\tv_accvgpr_read_b32 %6, a239\n\tv_accvgpr_read_b32 %7, a255" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]))

// ODD: maps with an odd height and / or width (7 x 7: ResNet-50 layer4, SphereNet conv4_x) as ceil(H/2) x ceil(W/2) tiles -- the last tile
// row / column hangs over the edge by one pixel.  Rows past the map are range-checked to zero as ever; a tile in the last column of an
// odd-width map fetches the pair (W-2, W-1) instead of (W-1, W) -- nothing past the tensor is touched -- and stores (x[W-1], 0); its
// outputs past the edge are not stored (one dword instead of the pair; the odd row of the last tile row not at all) and stay out of
// the BatchNorm statistics.
struct Bop {                                  // k_wg3's B operands of one channel: patch rows as register pairs (see T_read1)
    f32x2 P[3], Q[3];
};
// SH (round 4; even maps, training launches, a multiple of 128 output channels): a block is FOUR waves = TWO units on the SAME tile run
// (output-channel blocks 2 j, 2 j + 1).  The B operands of a chunk -- the transformed patch of the lane's tile for channels 2 lh, 2 lh + 1 --
// are the same for both units, so the waves (unit 0, ph) and (unit 1, ph) split them: unit u fetches, stages and transforms channel
// 2 lh + u only (3 + 1 instead of 6 + 1 row loads and LDS stores, 9 instead of 18 raw reads, 8 instead of 16 packed transform
// instructions per chunk), leaves its eight operand values in a shared double-buffered LDS array (two 16-byte stores) and both read
// all sixteen back after a barrier that waits for LDS traffic only (four 16-byte reads).  Everything else -- filter operands,
// accumulators, the epilogue and its exchange between the two waves of a unit -- is per unit, as before; every sum is bit for bit
// what the two-wave block computes.
// SPLIT (round 5): the TAIL launch of a layer whose units do not fill their last round (SphereNet-20's 256 -> 256 @14 at batch 256: 784
// four-wave blocks on 256 CUs = 3.06 rounds -- the last 16 blocks ran alone for a whole round).  wino_run() launches the full rounds as
// before and the leftover logical blocks here, each cut into split_s pieces along the channel loop (piece s contracts chunks
// s * split_nch .. and stores its partial outputs -- the output transform is linear -- to its own slice of a workspace); k_wg_tail_reduce
// adds the slices in a fixed order (+ bias) into y.  No statistics / inference epilogue, even maps.
// ADD (round 5; input-gradient launches): y = result + addend, elementwise -- the gradient of the OTHER consumer of the conv's input (SphereNet's
// residual units, models/spherenet.py:121-131: `x = x + relu(conv(relu(conv(x))))`: x feeds the first conv and the sum), which autograd would
// otherwise add in a separate 3-pass kernel.  The addend comes in through the `bias` parameter (an input-gradient launch has no bias) and is
// shaped like y.
template <bool DGRAD, bool STATS, bool BNE = false, bool ODD = false, bool SH = false, bool SPLIT = false, bool ADD = false>
__global__ __launch_bounds__(SH ? 256 : 128) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_wg3(WgGeom g, const float *__restrict__ x, const float *__restrict__ up, const float *__restrict__ bias,
           float *__restrict__ y, float *__restrict__ stats, WgBnEval bn) {
    static_assert(!SH || (!BNE && !ODD), "shared B operands: training launches on even maps");
    static_assert(!SPLIT || (!STATS && !BNE && !ODD), "the tail launch has the plain epilogue only");
    static_assert(!ADD || (DGRAD && !STATS && !BNE && !SPLIT), "the addend rides in plain input-gradient launches");
    constexpr int UNITF = 2 * 64 * 64;                       // per unit: 2 x 2 raw stages (3264 floats) / the epilogue's exchange buffer (32 KB)
    constexpr int BXF = 2 * 2 * 2 * 2 * 64 * 4;              // SH: [buffer][ph][channel j][half][lane][4] transformed operands (16 KB)
    __shared__ __attribute__((aligned(16))) float smem_all[SH ? 2 * UNITF + BXF : UNITF];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int un = SH ? wv >> 1 : 0, ph = SH ? wv & 1 : wv;   // unit of the block, wave of the unit
    const int li = lane & 31, lh = lane >> 5;
    const int HW = g.H * g.W;
    float *unit_smem = smem_all + un * UNITF;
    float *smem = unit_smem + ph * 2 * W2_RAW;                // this wave's private raw stages
    float *bx = smem_all + 2 * UNITF;                         // (SH only)

    // persistent blocks, as in k_wg1 (two blocks of two waves per CU are resident)
    const unsigned nlb = SH ? g.nblocks >> 1 : g.nblocks;     // SH: pairs of logical blocks (the host only sends an even number of channel blocks)
    for (unsigned base = 0; base < nlb; base += gridDim.x) {
    // (virtual block v runs on XCD v % 8 = blockIdx % 8 -- the grid is a multiple of 8 blocks whenever there is more than one
    //  round -- and XCD x owns the x-th eighth of ALL logical blocks, walking through it round after round: consecutive tile runs,
    //  which share half of their input rows, stay in one XCD's L2 as with one block per logical block)
    const unsigned v = base + blockIdx.x;
    if (v >= nlb) break;
    // SPLIT: virtual block v = (leftover block or pair v / split_s, piece v % split_s); g.nblocks counts the pieces
    const int piece = SPLIT ? (int)(v % (unsigned)g.split_s) : 0;
    const unsigned lb = SPLIT ? g.split_first + (SH ? 2 * (v / (unsigned)g.split_s) + (unsigned)un : v / (unsigned)g.split_s)
                              : (SH ? 2 * xcd_remap(v, nlb) + (unsigned)un : xcd_remap(v, nlb));
    const int ch0 = SPLIT ? piece * g.split_nch : 0;          // first channel chunk of this unit's contraction
    float *const yy = SPLIT ? y + (int64_t)piece * g.split_stride : y;
    const int nkb64 = (g.nkb + 1) / 2;                        // (g.nkb counts blocks of 32 channels)
    const int kb = lb % nkb64;                                // block of 64 output channels
    const unsigned ttot = (unsigned)g.tiles_total, timg = (unsigned)g.tiles_img, twu = (unsigned)g.tw;
    const unsigned run = lb / nkb64;                          // tile run of 32
    const unsigned t0 = run * W1_T;
    if (t0 >= ttot) continue;                                 // (uniform for the block)
#ifdef WG_TIMING
    const unsigned dbg_u = lb * 2 + ph;
    WG_STAMP(0);
    if (lane == 0 && dbg_u < 65536) wg_dbg[dbg_u * 8 + 6] = __builtin_amdgcn_s_getreg(63492), wg_dbg[dbg_u * 8 + 7] = __builtin_amdgcn_s_getreg(63508);
#endif
    const unsigned n0u = t0 / timg, r0 = t0 - n0u * timg;      // (uniform: the one generic division of a unit)
    const int n0 = (int)n0u;
    // tile t0 + add (add < 64) -> image relative to n0, tile row, tile column: r0 + add < tiles_img + 64 < 2^24 (cpg_conv3x3_wino_ok)
    auto locate = [&](unsigned add, int &nrel, int &ty, int &tx) {
        unsigned dn, r, tyu, txu;
        wg_divmod(r0 + add, timg, g.inv_timg, dn, r);
        wg_divmod(r, twu, g.inv_tw, tyu, txu);
        nrel = (int)dn, ty = (int)tyu, tx = (int)txu;
    };

    constexpr int kOutOfRange = (int)0x80000000;
    int roff[3], hoff, lo, ro;
    bool oddcol = false;                                     // ODD: this lane's tile hangs over the right edge of an odd-width map
    {
        const unsigned tg = t0 + li;
        const bool tv = tg < ttot;
        int nrel, ty, tx;
        locate((unsigned)li, nrel, ty, tx);
        const int n = n0 + nrel;
        const int cbase = ((n - n0) * g.C + 2 * lh) * HW;
        if (ODD) oddcol = (g.W & 1) && tx == g.tw - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int gh = 2 * ty - 1 + ph + (ph ? 2 - i : i);          // wave 1 holds its rows 3, 2, 1 (see T_col)
            roff[i] = (tv && (unsigned)gh < (unsigned)g.H) ? (cbase + gh * g.W + 2 * tx - (oddcol ? 1 : 0)) * 4 : kOutOfRange;
        }
        lo = tx == 0 ? 0 : (li + 1) * 2 - 1;
        ro = tx == g.tw - 1 ? (W1_T + 1) * 2 + 1 : (li + 1) * 2 + 2;
        // halo: lanes 0-23 = (side, row, channel): the column left of tile t0 / right of tile t0 + 31
        const int side = lane >= 12 ? 1 : 0, hl = lane - 12 * side, hi = hl >> 2, hc = hl & 3;
        const unsigned th = side ? t0 + W1_T - 1 : t0;
        int nhrel, tyh, txh;
        locate(side ? W1_T - 1 : 0, nhrel, tyh, txh);
        const int nh = n0 + nhrel;
        const int ghh = 2 * tyh - 1 + ph + (ph ? 2 - hi : hi), gwh = side ? 2 * txh + 2 : 2 * txh - 1;
        const bool okh = lane < 24 && th < ttot && (unsigned)ghh < (unsigned)g.H && (unsigned)gwh < (unsigned)g.W;
        hoff = okh ? (((nh - n0) * g.C + hc) * HW + ghh * g.W + gwh) * 4 : kOutOfRange;
    }
    const int span = (W1_T + g.tiles_img - 1) / g.tiles_img + 1;
    const int nimg_here = min(span, g.N - n0);
    // byte offset of the LAST chunk's first channel, set once the chunk count is known: min(last chunk * 4, C - 4) channels.  Every row load
    // takes min(chunk * stride, x_last_off): chunks past the end re-read the last one (what clampc did for them), and with a channel count
    // that is no multiple of 4 the last chunk starts at C - 4, overlapping its neighbour (wg_chunk_base; the pack kernel zeroes its filters
    // for the channels the neighbour already contracted) -- one scalar multiply and one min per chunk, as before.
    int x_last_off = 0;
    const __amdgpu_buffer_rsrc_t srd_x =
        __builtin_amdgcn_make_buffer_rsrc((void *)(x + ((int64_t)n0 * g.C + ch0 * WG_CK) * HW), 0, (nimg_here * g.C - ch0 * WG_CK) * HW * 4, 0x00020000);
    // U records are per block of 32 channels: this wave reads float4 q = 4 ph .. + 3 of the records of blocks 2 kb, 2 kb + 1
    // (through a buffer descriptor: four per-lane byte offsets computed once per unit, the chunk in the scalar offset, the float4 pair
    //  in the instruction offset -- no vector instruction per load; 64-bit pointer arithmetic was four per chunk)
    const __amdgpu_buffer_rsrc_t srd_u = __builtin_amdgcn_make_buffer_rsrc((void *)up, 0, (int)(((int64_t)g.nkb + 1) / 2 * 2 * g.nch * W1_U * 4), 0x00020000);
    const int ubase = (((2 * kb) * g.nch + ch0) * W1_U + ph * 1024 + lane * 4) * 4;
    const int ukq = g.nch * W1_U * 4;

    // raw[c][row][slot][2] offsets of this lane (channels 2 lh + j)
    const int raw_own = (2 * lh) * 3 * W1_ROW + (li + 1) * 2;
    const int hside = lane >= 12 ? 1 : 0, hrem = lane - 12 * hside;
    const int halo_w = ((hrem & 3) * 3 + (hrem >> 2)) * W1_ROW + (hside ? (W1_T + 1) * 2 : 1);
    if (lane < 24) {                                          // zero slots of the 2 x 12 rows (never written)
        smem[lane * W1_ROW] = 0.0f;
        smem[lane * W1_ROW + (W1_T + 1) * 2 + 1] = 0.0f;
    }

    struct Rows {
        i32x2 r[2][3];
        float halo;
    };
    auto G_row1 = [&](int ch, Rows &q, int k) {             // k = 0..5: row k % 3 of channel 2 lh + k / 3;  k = 6: the halo values
        const int soff = min(ch * (WG_CK * HW * 4), x_last_off);      // (see x_last_off)
        if (k < 6)
            q.r[k / 3][k % 3] = __builtin_amdgcn_raw_buffer_load_b64(srd_x, roff[k % 3], soff + (k / 3) * HW * 4, 0);
        else
            q.halo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, hoff, soff, 0));
    };
    auto W_row1 = [&](int stage, const Rows &q, int k) {
        float *raw = smem + stage * W2_RAW;
        if (k < 6) {
            i32x2 v = q.r[k / 3][k % 3];
            if (ODD) {                                       // the pair fetched was (W-2, W-1): keep x[W-1], the column past the edge is 0
                v[0] = oddcol ? v[1] : v[0];
                v[1] = oddcol ? 0 : v[1];
            }
            *reinterpret_cast<i32x2 *>(raw + raw_own + k * W1_ROW) = v;
        } else if (lane < 24)
            raw[halo_w] = q.halo;
    };
    // SH: this wave stages channel 2 lh + un only (k = 0..2: its three patch rows; k = 3: the halo values, all four channels as before)
    const int un_goff = un * HW * 4;
    const int raw_own_s = raw_own + un * 3 * W1_ROW, tr_base = ((2 * lh + un) * 3) * W1_ROW;
    auto G_rows = [&](int ch, Rows &q, int k) {
        const int soff = min(ch * (WG_CK * HW * 4), x_last_off);      // (see x_last_off)
        if (k < 3)
            q.r[0][k] = __builtin_amdgcn_raw_buffer_load_b64(srd_x, roff[k], soff + un_goff, 0);
        else
            q.halo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, hoff, soff, 0));
    };
    auto W_rows = [&](int stage, const Rows &q, int k) {
        float *raw = smem + stage * W2_RAW;
        if (k < 3)
            *reinterpret_cast<i32x2 *>(raw + raw_own_s + k * W1_ROW) = q.r[0][k];
        else if (lane < 24)
            raw[halo_w] = q.halo;
    };
    // Wave ph = 1 keeps its two transform rows in REVERSE order (local row 0 = row 3, local row 1 = row 2) AND its three patch rows
    // in reverse order (e0, e1, e2 = patch rows 3, 2, 1; wave 0: 0, 1, 2): then local row 0 of B^T d is e0 - e2 for both waves
    // (= row 0 = d0 - d2 for wave 0, = -(row 3) = d3 - d1 for wave 1) and local row 1 is e1 + sgn e2 (row 1 = d1 + d2, row 2 =
    // d2 - d1) -- ONE fused multiply-add with the wave's sign where a select between two operands was two instructions: 2 instead
    // of 3 vector instructions per column.  The negated row 3 is put right where the two waves' rows meet in the epilogue (own =
    // r1 + sgn r0 instead of r0 + r1): every value is bit for bit what it was.  The row a wave GIVES the other is local row 1 for both.
    const float sgn = ph ? -1.0f : 1.0f;
    f32x2 sgn2;
    sgn2[0] = sgn, sgn2[1] = sgn;
    const int urow = ph * 512 * 4;                            // float4 q ^ 2 of the wave's four: + 512 floats for q = 0, 1, - 512 for q = 2, 3
    const int uoff[4] = {ubase + urow, ubase - urow + 512 * 4, ubase + ukq + urow, ubase + ukq - urow + 512 * 4};   // [(q >> 2) * 2 + ((q >> 1) & 1)]
    auto G_u1 = [&](int ch, f32x4 (&u)[8], int q) {
        u[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_u, uoff[(q >> 2) * 2 + ((q >> 1) & 1)] + (q & 1) * 256 * 4, ch * (W1_U * 4), 0));
    };
    // half patch of (tile li, channel 2 lh + j): the wave's three patch rows as register PAIRS, P[i] = columns (1, 2) -- the lane's own
    // pixel pair, one 8-byte LDS read -- and Q[i] = columns (0, 3).  The transforms below are packed fp32 instructions (two results per
    // vector instruction: v_pk_add_f32 / v_pk_fma_f32 with op_sel / neg picking the halves and signs), 16 per chunk where the scalar
    // form was 30 -- on this chip every vector instruction of the wave is MFMA time (profiles/r03_pmc_step_vgg16.md).
    // After T_col and T_rowp rows 0, 1 hold the B operands: position il * 4 + j -> j = 0: Q[il][0], 1: P[il][0], 2: P[il][1], 3: Q[il][1].
    auto T_read1 = [&](int stage, int j, Bop &d, int i) {
        const float *raw = smem + stage * W2_RAW + ((2 * lh + j) * 3 + i) * W1_ROW;
        d.P[i] = *reinterpret_cast<const f32x2 *>(raw + (li + 1) * 2);
        d.Q[i][0] = raw[lo], d.Q[i][1] = raw[ro];
    };
    auto T_reads = [&](int stage, Bop &d, int i) {            // SH: patch row i of this wave's channel 2 lh + un
        const float *raw = smem + stage * W2_RAW + tr_base + i * W1_ROW;
        d.P[i] = *reinterpret_cast<const f32x2 *>(raw + (li + 1) * 2);
        d.Q[i][0] = raw[lo], d.Q[i][1] = raw[ro];
    };
    // SH: the transformed operands of channel 2 lh + un (local rows 0, 1: P and Q pairs) go to bx[buffer][ph][un]; both waves with this ph
    // then read channel j's from bx[buffer][ph][j]
    auto X_put = [&](int buf, const Bop &d) {
        float *dst = bx + ((((buf * 2 + ph) * 2 + un) * 2) * 64 + lane) * 4;
        f32x4 a, b;
        a[0] = d.P[0][0], a[1] = d.P[0][1], a[2] = d.Q[0][0], a[3] = d.Q[0][1];
        b[0] = d.P[1][0], b[1] = d.P[1][1], b[2] = d.Q[1][0], b[3] = d.Q[1][1];
        *reinterpret_cast<f32x4 *>(dst) = a;
        *reinterpret_cast<f32x4 *>(dst + 256) = b;
    };
    auto X_get = [&](int buf, int j, Bop &d, int half) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(bx + ((((buf * 2 + ph) * 2 + j) * 2 + half) * 64 + lane) * 4);
        d.P[half][0] = v[0], d.P[half][1] = v[1], d.Q[half][0] = v[2], d.Q[half][1] = v[3];
    };
#define W3_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // B^T d on a pair type (half = 0: the P pairs, 1: the Q pairs): local rows (0, 1) = e0 - e2, e1 + sgn e2
    auto T_col = [&](Bop &d, int half) {
        f32x2 &e0 = half ? d.Q[0] : d.P[0], &e1 = half ? d.Q[1] : d.P[1], &e2 = half ? d.Q[2] : d.P[2];
        f32x2 r0, r1;
        asm volatile("v_pk_add_f32 %0, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_fma_f32 %1, %4, %3, %5"
                     : "=&v"(r0), "=&v"(r1) : "v"(e0), "v"(e2), "v"(sgn2), "v"(e1));
        e0 = r0, e1 = r1;
    };
    // (B^T d) B on local row i: P = (t1, t2), Q = (t0, t3) -> Q = (t0 - t2, t1 - t3), P = (t1 + t2, t2 - t1)
    auto T_rowp = [&](Bop &d, int i) {
        f32x2 nq, np;
        asm volatile("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,0]\n\t"
                     "v_pk_add_f32 %1, %3, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]"
                     : "=&v"(nq), "=&v"(np) : "v"(d.Q[i]), "v"(d.P[i]));
        d.Q[i] = nq, d.P[i] = np;
    };


    WG_STAMP(1);
    int nch = SPLIT ? min(g.split_nch, g.nch - ch0) : g.nch;
    if (BNE && bn.live != nullptr) {                          // inference: skip what apply_mask killed (wave-uniform decisions)
        const int alive = (kb * 64 + lane < g.M) ? bn.live[kb * 64 + lane] : 0;
        const bool dead = __ballot(alive != 0) == 0ull;
        int lastc = g.nch;
        while (lastc > 0 && bn.live[bn.Mp + 4 + lastc - 1] == 0) --lastc;
        nch = dead ? 0 : lastc;
        if (lane == 0 && ph == 0) {
            if (dead) atomicAdd(&bn.live[bn.Mp + 1], 1);
            if (blockIdx.x == 0) bn.live[bn.Mp] = lastc * WG_CK;
        }
    }
    const int last = nch - 1;
    x_last_off = __builtin_amdgcn_readfirstlane(min(last * WG_CK, g.C - WG_CK - ch0 * WG_CK) * HW * 4);
    auto clampc = [&](int c) { return min(c, last); };
    // U of the current / next chunk: [kq * 4 + q].  U3 (every variant but the one with the statistics epilogue, which has no registers
    // left): a third buffer -- the loads of chunk it + 2 go out during chunk it, two chunks (2 us) ahead of their first use instead of
    // one; the wait counters showed the waves stalled on vector-memory data 13 % of the time with one chunk of distance
    constexpr bool U3 = true;
    f32x4 ua[8], ub[8], uc[U3 ? 8 : 1];
    Bop c0, c1, x0, x1;                    // B operands of the current chunk for channels 2 lh, 2 lh + 1 / the next chunk's, being transformed
    Bop mine;                              // SH: the half this wave transforms
    Rows rows;
    if (nch > 0) {
        if constexpr (SH) {
#pragma unroll
            for (int k = 0; k < 4; ++k) G_rows(0, rows, k);
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) G_row1(0, rows, k);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) G_u1(0, ua, q);
        if constexpr (U3) {
#pragma unroll
            for (int q = 0; q < 8; ++q) G_u1(min(1, nch - 1), ub, q);
        }
    }
    // the accumulators are cleared while those loads fly (256 instructions, 0.4 us)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0" : : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    asm volatile("v_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0" : : : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    asm volatile("v_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0" : : : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    asm volatile("v_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0" : : : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    asm volatile("v_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0" : : : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    asm volatile("v_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0" : : : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    asm volatile("v_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0" : : : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    asm volatile("v_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" : : : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
    asm volatile("v_accvgpr_write_b32 a128, 0\n\tv_accvgpr_write_b32 a129, 0\n\tv_accvgpr_write_b32 a130, 0\n\tv_accvgpr_write_b32 a131, 0\n\tv_accvgpr_write_b32 a132, 0\n\tv_accvgpr_write_b32 a133, 0\n\tv_accvgpr_write_b32 a134, 0\n\tv_accvgpr_write_b32 a135, 0\n\tv_accvgpr_write_b32 a136, 0\n\tv_accvgpr_write_b32 a137, 0\n\tv_accvgpr_write_b32 a138, 0\n\tv_accvgpr_write_b32 a139, 0\n\tv_accvgpr_write_b32 a140, 0\n\tv_accvgpr_write_b32 a141, 0\n\tv_accvgpr_write_b32 a142, 0\n\tv_accvgpr_write_b32 a143, 0" : : : "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");
    asm volatile("v_accvgpr_write_b32 a144, 0\n\tv_accvgpr_write_b32 a145, 0\n\tv_accvgpr_write_b32 a146, 0\n\tv_accvgpr_write_b32 a147, 0\n\tv_accvgpr_write_b32 a148, 0\n\tv_accvgpr_write_b32 a149, 0\n\tv_accvgpr_write_b32 a150, 0\n\tv_accvgpr_write_b32 a151, 0\n\tv_accvgpr_write_b32 a152, 0\n\tv_accvgpr_write_b32 a153, 0\n\tv_accvgpr_write_b32 a154, 0\n\tv_accvgpr_write_b32 a155, 0\n\tv_accvgpr_write_b32 a156, 0\n\tv_accvgpr_write_b32 a157, 0\n\tv_accvgpr_write_b32 a158, 0\n\tv_accvgpr_write_b32 a159, 0" : : : "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159");
    asm volatile("v_accvgpr_write_b32 a160, 0\n\tv_accvgpr_write_b32 a161, 0\n\tv_accvgpr_write_b32 a162, 0\n\tv_accvgpr_write_b32 a163, 0\n\tv_accvgpr_write_b32 a164, 0\n\tv_accvgpr_write_b32 a165, 0\n\tv_accvgpr_write_b32 a166, 0\n\tv_accvgpr_write_b32 a167, 0\n\tv_accvgpr_write_b32 a168, 0\n\tv_accvgpr_write_b32 a169, 0\n\tv_accvgpr_write_b32 a170, 0\n\tv_accvgpr_write_b32 a171, 0\n\tv_accvgpr_write_b32 a172, 0\n\tv_accvgpr_write_b32 a173, 0\n\tv_accvgpr_write_b32 a174, 0\n\tv_accvgpr_write_b32 a175, 0" : : : "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175");
    asm volatile("v_accvgpr_write_b32 a176, 0\n\tv_accvgpr_write_b32 a177, 0\n\tv_accvgpr_write_b32 a178, 0\n\tv_accvgpr_write_b32 a179, 0\n\tv_accvgpr_write_b32 a180, 0\n\tv_accvgpr_write_b32 a181, 0\n\tv_accvgpr_write_b32 a182, 0\n\tv_accvgpr_write_b32 a183, 0\n\tv_accvgpr_write_b32 a184, 0\n\tv_accvgpr_write_b32 a185, 0\n\tv_accvgpr_write_b32 a186, 0\n\tv_accvgpr_write_b32 a187, 0\n\tv_accvgpr_write_b32 a188, 0\n\tv_accvgpr_write_b32 a189, 0\n\tv_accvgpr_write_b32 a190, 0\n\tv_accvgpr_write_b32 a191, 0" : : : "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191");
    asm volatile("v_accvgpr_write_b32 a192, 0\n\tv_accvgpr_write_b32 a193, 0\n\tv_accvgpr_write_b32 a194, 0\n\tv_accvgpr_write_b32 a195, 0\n\tv_accvgpr_write_b32 a196, 0\n\tv_accvgpr_write_b32 a197, 0\n\tv_accvgpr_write_b32 a198, 0\n\tv_accvgpr_write_b32 a199, 0\n\tv_accvgpr_write_b32 a200, 0\n\tv_accvgpr_write_b32 a201, 0\n\tv_accvgpr_write_b32 a202, 0\n\tv_accvgpr_write_b32 a203, 0\n\tv_accvgpr_write_b32 a204, 0\n\tv_accvgpr_write_b32 a205, 0\n\tv_accvgpr_write_b32 a206, 0\n\tv_accvgpr_write_b32 a207, 0" : : : "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207");
    asm volatile("v_accvgpr_write_b32 a208, 0\n\tv_accvgpr_write_b32 a209, 0\n\tv_accvgpr_write_b32 a210, 0\n\tv_accvgpr_write_b32 a211, 0\n\tv_accvgpr_write_b32 a212, 0\n\tv_accvgpr_write_b32 a213, 0\n\tv_accvgpr_write_b32 a214, 0\n\tv_accvgpr_write_b32 a215, 0\n\tv_accvgpr_write_b32 a216, 0\n\tv_accvgpr_write_b32 a217, 0\n\tv_accvgpr_write_b32 a218, 0\n\tv_accvgpr_write_b32 a219, 0\n\tv_accvgpr_write_b32 a220, 0\n\tv_accvgpr_write_b32 a221, 0\n\tv_accvgpr_write_b32 a222, 0\n\tv_accvgpr_write_b32 a223, 0" : : : "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223");
    asm volatile("v_accvgpr_write_b32 a224, 0\n\tv_accvgpr_write_b32 a225, 0\n\tv_accvgpr_write_b32 a226, 0\n\tv_accvgpr_write_b32 a227, 0\n\tv_accvgpr_write_b32 a228, 0\n\tv_accvgpr_write_b32 a229, 0\n\tv_accvgpr_write_b32 a230, 0\n\tv_accvgpr_write_b32 a231, 0\n\tv_accvgpr_write_b32 a232, 0\n\tv_accvgpr_write_b32 a233, 0\n\tv_accvgpr_write_b32 a234, 0\n\tv_accvgpr_write_b32 a235, 0\n\tv_accvgpr_write_b32 a236, 0\n\tv_accvgpr_write_b32 a237, 0\n\tv_accvgpr_write_b32 a238, 0\n\tv_accvgpr_write_b32 a239, 0" : : : "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239");
    asm volatile("v_accvgpr_write_b32 a240, 0\n\tv_accvgpr_write_b32 a241, 0\n\tv_accvgpr_write_b32 a242, 0\n\tv_accvgpr_write_b32 a243, 0\n\tv_accvgpr_write_b32 a244, 0\n\tv_accvgpr_write_b32 a245, 0\n\tv_accvgpr_write_b32 a246, 0\n\tv_accvgpr_write_b32 a247, 0\n\tv_accvgpr_write_b32 a248, 0\n\tv_accvgpr_write_b32 a249, 0\n\tv_accvgpr_write_b32 a250, 0\n\tv_accvgpr_write_b32 a251, 0\n\tv_accvgpr_write_b32 a252, 0\n\tv_accvgpr_write_b32 a253, 0\n\tv_accvgpr_write_b32 a254, 0\n\tv_accvgpr_write_b32 a255, 0" : : : "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SH) {
        if (nch > 0) {                        // (nch is the same for the two units of a block: the barriers pair up)
#pragma unroll
            for (int k = 0; k < 4; ++k) W_rows(0, rows, k);
#pragma unroll
            for (int k = 0; k < 4; ++k) G_rows(clampc(1), rows, k);
#pragma unroll
            for (int i = 0; i < 3; ++i) T_reads(0, mine, i);
            T_col(mine, 0); T_col(mine, 1);
            T_rowp(mine, 0); T_rowp(mine, 1);
            X_put(0, mine);
            W3_LDS_BARRIER();
            X_get(0, 0, c0, 0); X_get(0, 0, c0, 1); X_get(0, 1, c1, 0); X_get(0, 1, c1, 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) W_rows(1, rows, k);
#pragma unroll
            for (int k = 0; k < 4; ++k) G_rows(clampc(2), rows, k);
        }
    } else if (nch > 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) W_row1(0, rows, k);
#pragma unroll
        for (int k = 0; k < 7; ++k) G_row1(clampc(1), rows, k);
#pragma unroll
        for (int i = 0; i < 3; ++i) { T_read1(0, 0, c0, i); T_read1(0, 1, c1, i); }
        T_col(c0, 0); T_col(c0, 1); T_col(c1, 0); T_col(c1, 1);
        T_rowp(c0, 0); T_rowp(c0, 1); T_rowp(c1, 0); T_rowp(c1, 1);
#pragma unroll
        for (int k = 0; k < 7; ++k) W_row1(1, rows, k);
#pragma unroll
        for (int k = 0; k < 7; ++k) G_row1(clampc(2), rows, k);
    }
// accumulator a = kq * 8 + pp (channel half kq, local position pp = il * 4 + j: see T_read1 for where the operand sits)
#define W3_B(B, pp) (((pp) & 3) == 0 ? (B).Q[(pp) >> 2][0] : ((pp) & 3) == 1 ? (B).P[(pp) >> 2][0] : ((pp) & 3) == 2 ? (B).P[(pp) >> 2][1] : (B).Q[(pp) >> 2][1])
#define W3_SLOT(a, h, U, B, work)                                                                  \
    W3_ONE_##a((U)[((a) >> 3) * 4 + (((a) & 7) >> 1)][((a) & 1) * 2 + (h)], W3_B(B, (a) & 7));     \
    work;                                                                                          \
    __builtin_amdgcn_sched_barrier(0)
    // iteration it (par = it & 1): M(it); T(it + 1) from raw stage (it + 1) & 1; U(it + 1) requested; W(it + 2) stores the rows
    // requested one iteration ago, whose registers then take the loads of chunk it + 3
    auto iter = [&](int it, int par, f32x4 (&ucur)[8], f32x4 (&unext)[8], Bop &b0, Bop &b1, Bop &n0v, Bop &n1v) {
        const int cu = clampc(it + (U3 ? 2 : 1)), cr = it + 3;
        if constexpr (SH) {
            // the next chunk's operands: this wave's channel, the barrier, both channels read back; then the rows of chunk it + 2 / it + 3
            W3_SLOT(0, 0, ucur, b0, G_u1(cu, unext, 0));
            W3_SLOT(0, 1, ucur, b1, G_u1(cu, unext, 1));
            W3_SLOT(8, 0, ucur, b0, G_u1(cu, unext, 2));
            W3_SLOT(8, 1, ucur, b1, G_u1(cu, unext, 3));
            W3_SLOT(1, 0, ucur, b0, G_u1(cu, unext, 4));
            W3_SLOT(1, 1, ucur, b1, G_u1(cu, unext, 5));
            W3_SLOT(9, 0, ucur, b0, G_u1(cu, unext, 6));
            W3_SLOT(9, 1, ucur, b1, G_u1(cu, unext, 7));
            W3_SLOT(2, 0, ucur, b0, T_reads(par ^ 1, mine, 0));
            W3_SLOT(2, 1, ucur, b1, T_reads(par ^ 1, mine, 1));
            W3_SLOT(10, 0, ucur, b0, T_reads(par ^ 1, mine, 2));
            W3_SLOT(10, 1, ucur, b1, T_col(mine, 0));
            W3_SLOT(3, 0, ucur, b0, T_col(mine, 1));
            W3_SLOT(3, 1, ucur, b1, T_rowp(mine, 0));
            W3_SLOT(11, 0, ucur, b0, T_rowp(mine, 1));
            W3_SLOT(11, 1, ucur, b1, X_put(par ^ 1, mine));
            // (the barrier sits ten slots behind the stores it waits for: its s_waitcnt then finds them done -- right behind them the wave
            //  stood at the wait for an LDS latency per chunk while the MFMA pipe drained -- and the two units may drift by that much)
            W3_SLOT(4, 0, ucur, b0, W_rows(par, rows, 0); G_rows(cr, rows, 0));
            W3_SLOT(4, 1, ucur, b1, W_rows(par, rows, 1); G_rows(cr, rows, 1));
            W3_SLOT(12, 0, ucur, b0, W_rows(par, rows, 2); G_rows(cr, rows, 2));
            W3_SLOT(12, 1, ucur, b1, W_rows(par, rows, 3); G_rows(cr, rows, 3));
            W3_SLOT(5, 0, ucur, b0, );
            W3_SLOT(5, 1, ucur, b1, );
            W3_SLOT(13, 0, ucur, b0, );
            W3_SLOT(13, 1, ucur, b1, );
            W3_SLOT(6, 0, ucur, b0, );
            W3_SLOT(6, 1, ucur, b1, W3_LDS_BARRIER());
            W3_SLOT(14, 0, ucur, b0, X_get(par ^ 1, 0, n0v, 0));
            W3_SLOT(14, 1, ucur, b1, X_get(par ^ 1, 0, n0v, 1));
            W3_SLOT(7, 0, ucur, b0, X_get(par ^ 1, 1, n1v, 0));
            W3_SLOT(7, 1, ucur, b1, X_get(par ^ 1, 1, n1v, 1));
            W3_SLOT(15, 0, ucur, b0, );
            W3_SLOT(15, 1, ucur, b1, );
            return;
        }
        W3_SLOT(0, 0, ucur, b0, G_u1(cu, unext, 0));
        W3_SLOT(0, 1, ucur, b1, G_u1(cu, unext, 1));
        W3_SLOT(8, 0, ucur, b0, G_u1(cu, unext, 2));
        W3_SLOT(8, 1, ucur, b1, G_u1(cu, unext, 3));
        W3_SLOT(1, 0, ucur, b0, G_u1(cu, unext, 4));
        W3_SLOT(1, 1, ucur, b1, G_u1(cu, unext, 5));
        W3_SLOT(9, 0, ucur, b0, G_u1(cu, unext, 6));
        W3_SLOT(9, 1, ucur, b1, G_u1(cu, unext, 7));
        W3_SLOT(2, 0, ucur, b0, T_read1(par ^ 1, 0, n0v, 0));
        W3_SLOT(2, 1, ucur, b1, T_read1(par ^ 1, 0, n0v, 1));
        W3_SLOT(10, 0, ucur, b0, T_read1(par ^ 1, 0, n0v, 2));
        W3_SLOT(10, 1, ucur, b1, T_read1(par ^ 1, 1, n1v, 0));
        W3_SLOT(3, 0, ucur, b0, T_read1(par ^ 1, 1, n1v, 1));
        W3_SLOT(3, 1, ucur, b1, T_read1(par ^ 1, 1, n1v, 2));
        W3_SLOT(11, 0, ucur, b0, T_col(n0v, 0));
        W3_SLOT(11, 1, ucur, b1, T_col(n0v, 1));
        W3_SLOT(4, 0, ucur, b0, T_col(n1v, 0));
        W3_SLOT(4, 1, ucur, b1, T_col(n1v, 1));
        W3_SLOT(12, 0, ucur, b0, T_rowp(n0v, 0));
        W3_SLOT(12, 1, ucur, b1, T_rowp(n0v, 1));
        W3_SLOT(5, 0, ucur, b0, T_rowp(n1v, 0));
        W3_SLOT(5, 1, ucur, b1, T_rowp(n1v, 1));
        W3_SLOT(13, 0, ucur, b0, W_row1(par, rows, 0); G_row1(cr, rows, 0));
        W3_SLOT(13, 1, ucur, b1, W_row1(par, rows, 1); G_row1(cr, rows, 1));
        W3_SLOT(6, 0, ucur, b0, W_row1(par, rows, 2); G_row1(cr, rows, 2));
        W3_SLOT(6, 1, ucur, b1, W_row1(par, rows, 3); G_row1(cr, rows, 3));
        W3_SLOT(14, 0, ucur, b0, W_row1(par, rows, 4); G_row1(cr, rows, 4));
        W3_SLOT(14, 1, ucur, b1, W_row1(par, rows, 5); G_row1(cr, rows, 5));
        W3_SLOT(7, 0, ucur, b0, W_row1(par, rows, 6); G_row1(cr, rows, 6));
        W3_SLOT(7, 1, ucur, b1, );
        W3_SLOT(15, 0, ucur, b0, );
        W3_SLOT(15, 1, ucur, b1, );
    };
    WG_STAMP(2);
    if constexpr (U3) {                       // (U buffers rotate with period 3, the operand sets with period 2: six iterations per trip)
        for (int it = 0; it < nch; it += 6) {
            iter(it, 0, ua, uc, c0, c1, x0, x1);
            if (it + 1 < nch) iter(it + 1, 1, ub, ua, x0, x1, c0, c1);
            if (it + 2 < nch) iter(it + 2, 0, uc, ub, c0, c1, x0, x1);
            if (it + 3 < nch) iter(it + 3, 1, ua, uc, x0, x1, c0, c1);
            if (it + 4 < nch) iter(it + 4, 0, ub, ua, c0, c1, x0, x1);
            if (it + 5 < nch) iter(it + 5, 1, uc, ub, x0, x1, c0, c1);
        }
    } else {
        for (int it = 0; it < nch; it += 2) {
            iter(it, 0, ua, ub, c0, c1, x0, x1);
            if (it + 1 < nch) iter(it + 1, 1, ub, ua, x0, x1, c0, c1);
        }
    }

    WG_STAMP(3);
    // ---- epilogue: this wave's 8 positions (transform rows i = 2 ph, 2 ph + 1) -> partial 2x2 outputs; the output transform is linear
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float own0[32], own1[32];                  // the output row this wave finishes (a = ph), columns 0 / 1, per channel half and accumulator element
    // (the epilogue's lane constants are derived from an opaque copy: computed from `lh` itself they are invariant in the persistent loop,
    //  the compiler hoists all 32 channel offsets in front of it and spills them)
    int lh_e = lh;
    asm volatile("" : "+v"(lh_e));
    __syncthreads();                           // both waves are done with their raw stages: the LDS becomes the exchange buffer
    float *xch = unit_smem;
    // local rows: wave 0 holds R0, R1, wave 1 holds -R3, R2 (reversed; row 3 negated, see T_col).  Y0 = R0 + R1 + R2, Y1 = R1 - R2 - R3: a wave owns
    // +-(its two rows) and gives its local row 1 (R1 to Y1, R2 to Y0) to the other wave; the sign is applied when the two meet.
    auto part_e = [&](int kq, int e, const float (&m)[8]) {
        const float r00 = m[0] + m[1] + m[2], r01 = m[1] - m[2] - m[3];       // R[il][b] = sum_j A^T[b][j] M[i][j]
        const float r10 = m[4] + m[5] + m[6], r11 = m[5] - m[6] - m[7];
        own0[kq * 16 + e] = fmaf(sgn, r00, r10), own1[kq * 16 + e] = fmaf(sgn, r01, r11);   // wave 1's local row 0 is -R3 (T_col): R2 + R3
        f32x2 gv;                              // (one 8-byte LDS store / load per channel: xch[wave][kq][e][lane][2])
        gv[0] = r10, gv[1] = r11;
        *reinterpret_cast<f32x2 *>(xch + (((ph * 2 + kq) * 16 + e) * 64 + lane) * 2) = gv;
    };
    {
        float m[8];
        W3_RD_0_0(m); part_e(0, 0, m);
        W3_RD_0_1(m); part_e(0, 1, m);
        W3_RD_0_2(m); part_e(0, 2, m);
        W3_RD_0_3(m); part_e(0, 3, m);
        W3_RD_0_4(m); part_e(0, 4, m);
        W3_RD_0_5(m); part_e(0, 5, m);
        W3_RD_0_6(m); part_e(0, 6, m);
        W3_RD_0_7(m); part_e(0, 7, m);
        W3_RD_0_8(m); part_e(0, 8, m);
        W3_RD_0_9(m); part_e(0, 9, m);
        W3_RD_0_10(m); part_e(0, 10, m);
        W3_RD_0_11(m); part_e(0, 11, m);
        W3_RD_0_12(m); part_e(0, 12, m);
        W3_RD_0_13(m); part_e(0, 13, m);
        W3_RD_0_14(m); part_e(0, 14, m);
        W3_RD_0_15(m); part_e(0, 15, m);
        W3_RD_1_0(m); part_e(1, 0, m);
        W3_RD_1_1(m); part_e(1, 1, m);
        W3_RD_1_2(m); part_e(1, 2, m);
        W3_RD_1_3(m); part_e(1, 3, m);
        W3_RD_1_4(m); part_e(1, 4, m);
        W3_RD_1_5(m); part_e(1, 5, m);
        W3_RD_1_6(m); part_e(1, 6, m);
        W3_RD_1_7(m); part_e(1, 7, m);
        W3_RD_1_8(m); part_e(1, 8, m);
        W3_RD_1_9(m); part_e(1, 9, m);
        W3_RD_1_10(m); part_e(1, 10, m);
        W3_RD_1_11(m); part_e(1, 11, m);
        W3_RD_1_12(m); part_e(1, 12, m);
        W3_RD_1_13(m); part_e(1, 13, m);
        W3_RD_1_14(m); part_e(1, 14, m);
        W3_RD_1_15(m); part_e(1, 15, m);

    }
    __syncthreads();
    const unsigned tg = t0 + li;
    int nrel, ty, tx;
    locate((unsigned)li, nrel, ty, tx);
    const int n = n0 + nrel;
    // (ODD: the odd row of the last tile row of an odd-height map does not exist -- the wave that owns it stores nothing for that tile)
    const bool tv = tg < ttot && (!ODD || 2 * ty + ph < g.H);
    float *yout = yy + ((int64_t)n * g.M) * HW + (2 * ty + ph) * g.W + 2 * tx;
    // FAST (all 32 tiles of the unit exist, all 64 channels of the block exist, even map -- a wave-uniform test): plain stores at the unit's
    // (scalar) base + a 32-bit lane offset + the running scalar channel offset, one vector add per store.  The general path computes a
    // 64-bit address per store (a multiply-add and two shift-adds), compares and branches around it: 1 us of a unit for the few units
    // on the ragged end of a launch that need it.
    const bool fast_u = !ODD && t0 + W1_T <= ttot && g.M - kb * 64 >= 64;
    char *ybase = reinterpret_cast<char *>(yy + (int64_t)n0 * g.M * HW);
    // (ADD) the addend at the same offsets as y: `bias` IS the addend tensor
    const char *abase = reinterpret_cast<const char *>(bias + (ADD ? (int64_t)n0 * g.M * HW : 0));
    const float *aout = bias + (ADD ? ((int64_t)n * g.M) * HW + (2 * ty + ph) * g.W + 2 * tx : 0);
    const unsigned yoff = (unsigned)(((nrel * g.M + kb * 64 + 4 * lh_e) * HW + (2 * ty + ph) * g.W + 2 * tx) * 4);
    // (the scalar channel offsets are a running sum behind an opaque barrier: as multiples of HW they are invariant in the persistent
    //  loop, the compiler would compute all 32 in front of it and spill scalar registers into vector lanes)
    int HW4 = HW * 4;
    asm volatile("" : "+s"(HW4));
    // (the statistics go out per 32-channel half: with all 32 + 32 sums live next to own0 / own1 the kernel is past 256 vector registers and the
    //  allocator parks values in scratch and in an accumulation register -- tests/test_abi_and_host.py checks the compiled code for both)
    auto out_half = [&](auto hb, auto kqc, auto fastc) {     // (hb: with / without a conv bias -- two separate epilogues, see k_wg1)
    constexpr int kq = decltype(kqc)::value;
    constexpr bool FAST = decltype(fastc)::value;
    float s1[16], s2[16];
    constexpr int GR = STATS ? 2 : 4;          // (the statistics variant has no registers for four)
    f32x2 got4[GR];
    f32x2 ad4[(ADD && FAST) ? GR : 1];         // (ADD, fast path) the addend of the group's GR channels, requested with the exchange reads
    int soff = kq * 32 * HW4;                  // (kq * 32 + (e & 3) + 8 * (e >> 2)) * HW4, stepped
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ke = kq * 16 + e;
        const int co = kb * 64 + kq * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh_e;
        if ((e & (GR - 1)) == 0) {             // (the exchange reads of GR channels go out together: one LDS latency per GR stores)
#pragma unroll
            for (int j = 0; j < GR; ++j)
                got4[j] = *reinterpret_cast<const f32x2 *>(xch + ((((ph ^ 1) * 2 + kq) * 16 + e + j) * 64 + lane) * 2);
            if constexpr (ADD && FAST && decltype(hb)::value) {
                static_assert(!ADD || GR == 4, "a group is the four channels between two 5 HW jumps of the running offset");
#pragma unroll
                for (int j = 0; j < GR; ++j)
                    ad4[j] = *reinterpret_cast<const f32x2 *>(abase + (size_t)(yoff + (unsigned)(soff + j * HW4)));
            }
        }
        const f32x2 got = got4[e & (GR - 1)];
        float v0 = fmaf(own0[ke], sgn, got[0]);             // (+- own + got, exactly)
        float v1 = fmaf(own1[ke], sgn, got[1]);
        if constexpr (decltype(hb)::value) {
            if constexpr (ADD) {
                if constexpr (FAST) {
                    v0 += ad4[e & (GR - 1)][0], v1 += ad4[e & (GR - 1)][1];
                } else if (tv && co < g.M) {
                    if (ODD && oddcol) {
                        v0 += aout[(int64_t)co * HW];
                    } else {
                        const f32x2 ad = *reinterpret_cast<const f32x2 *>(aout + (int64_t)co * HW);
                        v0 += ad[0], v1 += ad[1];
                    }
                }
            } else {
                const float bv = bias[co < g.M ? co : 0];
                v0 += bv, v1 += bv;
            }
        }
        if (BNE) {                             // y = [max(0,] (conv + bias - mean) * invstd * gamma + beta [)]
            const int cc = co < g.M ? co : 0;
            const float mu = bn.mean[cc], is = 1.0f / sqrtf(bn.var[cc] + bn.eps), ga = bn.gamma[cc], be = bn.beta[cc];
            v0 = (v0 - mu) * is * ga + be, v1 = (v1 - mu) * is * ga + be;
            if (bn.relu) v0 = fmaxf(v0, 0.0f), v1 = fmaxf(v1, 0.0f);
        }
        if constexpr (FAST) {
            f32x2 o;
            o[0] = v0, o[1] = v1;
            *reinterpret_cast<f32x2 *>(ybase + (size_t)(yoff + (unsigned)soff)) = o;
            soff += (e & 3) == 3 ? 5 * HW4 : HW4;
            asm volatile("" : "+s"(soff));
            if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (with no branch between them the scheduler hoists every exchange read: spills)
            if (STATS) s1[e] = v0 + v1, s2[e] = v0 * v0 + v1 * v1;
        } else {
            if (ODD && oddcol) {               // the second column is past the edge: one dword, and it stays out of the statistics
                if (tv && co < g.M) yout[(int64_t)co * HW] = v0;
                v1 = 0.0f;
            } else if (tv && co < g.M) {
                f32x2 o;
                o[0] = v0, o[1] = v1;
                *reinterpret_cast<f32x2 *>(yout + (int64_t)co * HW) = o;
            }
            if (STATS) {
                s1[e] = tv ? v0 + v1 : 0.0f;
                s2[e] = tv ? v0 * v0 + v1 * v1 : 0.0f;
            }
        }
    }
    if (STATS) {                               // every wave is its own statistics tile: stats[k][2 run + ph][2]
#pragma unroll
        for (int e = 0; e < 16; e += 8) half_wave_sum8(s1 + e), half_wave_sum8(s2 + e);
        if (li == kHalfSumLane) {
            const unsigned ntile = 2 * ((ttot + W1_T - 1) / W1_T);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = kb * 64 + kq * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh_e;
                if (co < g.M) {
                    float *dst = stats + ((int64_t)co * ntile + 2 * run + ph) * 2;
                    f32x2 o;
                    o[0] = s1[e], o[1] = s2[e];
                    *reinterpret_cast<f32x2 *>(dst) = o;
                }
            }
        }
    }
    };
    auto out_all = [&](auto hb) {
        if constexpr (!ODD) {
            if (fast_u) {
                out_half(hb, std::integral_constant<int, 0>{}, std::true_type{}), out_half(hb, std::integral_constant<int, 1>{}, std::true_type{});
                return;
            }
        }
        out_half(hb, std::integral_constant<int, 0>{}, std::false_type{}), out_half(hb, std::integral_constant<int, 1>{}, std::false_type{});
    };
    if (bias != nullptr) out_all(std::true_type{}); else out_all(std::false_type{});
    WG_STAMP(4);
    __syncthreads();                           // the exchange buffer becomes the next unit's raw stages
    }   // next logical block
}

// y = bias + sum over the pieces of k_wg3<.., SPLIT>'s partial outputs, for the logical blocks lb_first .. lb_end - 1 of the layer (each: 64 output
// channels x 32 tiles x 2 x 2 pixels).  The pieces are added in index order: the result does not depend on which piece finished first.
// `part` is piece 0's slice, addressed like y (image n at (n * M + m) * H W); piece s lies s * stride floats further.
template <bool STATS>
__global__ __launch_bounds__(256) void k_wg_tail_reduce(WgGeom g, const float *__restrict__ part, const float *__restrict__ bias,
                                                        float *__restrict__ y, unsigned lb_first, unsigned lb_end, float *__restrict__ stats,
                                                        const float *__restrict__ addend) {
    // one half-wave = the 32 tiles of one (unit, channel): total is a multiple of 32, so a half-wave is never split by the loop bound
    const int64_t total = (int64_t)(lb_end - lb_first) * 64 * W1_T;
    const int nkb64 = (g.nkb + 1) / 2, HW = g.H * g.W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i & (W1_T - 1)), kl = (int)((i >> 5) & 63);
        const unsigned lb = lb_first + (unsigned)(i >> 11);
        const int m = (int)(lb % (unsigned)nkb64) * 64 + kl;
        const unsigned run = lb / (unsigned)nkb64;
        const int64_t tile = (int64_t)run * W1_T + t;
        const bool ok = m < g.M && tile < g.tiles_total;
        f32x2 a0, a1;
        a0[0] = a0[1] = a1[0] = a1[1] = 0.0f;
        int64_t off = 0;
        if (ok) {
            const int n = (int)(tile / g.tiles_img), r = (int)(tile % g.tiles_img);
            const int ty = r / g.tw, tx = r % g.tw;
            off = ((int64_t)n * g.M + m) * HW + (int64_t)(2 * ty) * g.W + 2 * tx;
            const float b = bias != nullptr ? bias[m] : 0.0f;
            a0[0] = a0[1] = a1[0] = a1[1] = b;
            for (int sidx = 0; sidx < g.split_s; ++sidx) {
                const float *p = part + (int64_t)sidx * g.split_stride + off;
                const f32x2 v0 = *reinterpret_cast<const f32x2 *>(p), v1 = *reinterpret_cast<const f32x2 *>(p + g.W);
                a0[0] += v0[0], a0[1] += v0[1], a1[0] += v1[0], a1[1] += v1[1];
            }
            if (addend != nullptr) {                                 // (the fused skip gradient of an input-gradient launch: k_wg3<.., ADD>)
                const f32x2 d0 = *reinterpret_cast<const f32x2 *>(addend + off), d1 = *reinterpret_cast<const f32x2 *>(addend + off + g.W);
                a0[0] += d0[0], a0[1] += d0[1], a1[0] += d1[0], a1[1] += d1[1];
            }
            *reinterpret_cast<f32x2 *>(y + off) = a0;
            *reinterpret_cast<f32x2 *>(y + off + g.W) = a1;
        }
        if constexpr (STATS) {
            // the BatchNorm partial sums of the unit, in k_wg3's layout: statistics tile 2 run + ph holds output row ph of the unit's 32 tiles
            // (a wave of the main kernel is its own tile).  Sum over the half-wave's 32 lanes (xor shuffles below 32 stay inside it); a
            // lane without a tile / channel contributes zeros.
            float s[4] = {a0[0] + a0[1], a0[0] * a0[0] + a0[1] * a0[1], a1[0] + a1[1], a1[0] * a1[0] + a1[1] * a1[1]};
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1)
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] += __shfl_xor(s[q], d, 64);
            if (t == 0 && m < g.M) {
                const unsigned ntile = 2 * (unsigned)((g.tiles_total + W1_T - 1) / W1_T);
                float *dst = stats + ((int64_t)m * ntile + 2 * run) * 2;
                dst[0] = s[0], dst[1] = s[1], dst[2] = s[2], dst[3] = s[3];
            }
        }
    }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// The tail of a k_wg3 launch (see k_wg3<.., SPLIT>): `units` blocks (SH: four-wave blocks of two units, one resident per CU; otherwise
// two-wave blocks, two per CU) = some full rounds + `left` blocks that would run a whole round alone.  When they are at most half a round,
// each is cut into S <= 16 pieces along the channel loop (>= 2 chunks per piece) so that the pieces fill (at most) one round of 1 / S length.
struct WgTail {
    bool on;
    unsigned first_lb, end_lb;    // logical blocks of the tail
    int64_t main_units, left;
    int S, nch_per, n_first;
    int64_t stride;               // floats per piece: images n_first .. N - 1 of the output
    size_t bytes;
};
static WgTail wino_tail_plan(const WgGeom &g0, int64_t nblocks, bool sh) {
    WgTail t{};
    if (opt_or(OPT_WINO_TAIL, 1) == 0) return t;
    const int64_t units = sh ? nblocks / 2 : nblocks, slots = sh ? kCUs : 2 * kCUs;
    const int64_t full = units / slots * slots, left = units - full;
    if (full == 0 || left == 0 || left * 2 > slots) return t;
    int S = (int)std::min<int64_t>(slots / left, 16);
    S = std::min(S, g0.nch / 2);
    if (S < 2) return t;
    const int per = (g0.nch + S - 1) / S;
    S = (g0.nch + per - 1) / per;
    if (S < 2) return t;
    t.first_lb = (unsigned)(sh ? 2 * full : full), t.end_lb = (unsigned)nblocks;
    t.main_units = full, t.left = left, t.S = S, t.nch_per = per;
    const int nkb64 = (g0.nkb + 1) / 2;
    t.n_first = (int)(((int64_t)(t.first_lb / (unsigned)nkb64) * W1_T) / g0.tiles_img);
    t.stride = (int64_t)(g0.N - t.n_first) * g0.M * g0.H * g0.W;
    t.bytes = (size_t)S * t.stride * sizeof(float);
    t.on = t.bytes <= ((size_t)64 << 20) && t.stride > 0;
    return t;
}

// waves per block for a launch.  The 8-wave / 64-channel block does half the staging work per MFMA, but its eight waves run in
// lock step behind one barrier and it measured 3-4 % SLOWER than two independent 4-wave blocks per CU on every VGG16 layer
// (docs/LAB_NOTEBOOK.md section 4.9), so it is only reachable through CPG_WINO_NW=8 (A/B experiments, tests).
inline int wino_nw(int c_read, int m) {
    return opt_or(OPT_WINO_NW, 4) == 8 ? 8 : 4;
}

template <int NW>
int wino_launch(bool dgrad, const WgGeom &g, int64_t tblocks, const float *x, const float *up, const float *bias, float *y, float *stats,
                hipStream_t stream) {
    const int64_t blocks = tblocks * g.nkb;
    if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "conv3x3 (winograd): grid too large");
    if (dgrad)
        hipLaunchKernelGGL((k_wg_fwd<NW, true, false>), dim3((unsigned)blocks), dim3(64 * NW), 0, stream, g, x, up, bias, y, nullptr);
    else if (stats != nullptr)
        hipLaunchKernelGGL((k_wg_fwd<NW, false, true>), dim3((unsigned)blocks), dim3(64 * NW), 0, stream, g, x, up, bias, y, stats);
    else
        hipLaunchKernelGGL((k_wg_fwd<NW, false, false>), dim3((unsigned)blocks), dim3(64 * NW), 0, stream, g, x, up, bias, y, nullptr);
    return CPG_OK;
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------
// eligibility of one launch (c_read channels contracted, m produced): even maps, the staging offsets of
// the images a block can touch fit 31 bits
// odd maps (7 x 7: ResNet-50 layer4, SphereNet conv4_x): only the two-wave kernel k_wg3 has the edge handling (its ODD instances), and it
// only pays with a long channel loop -- >= 128 channels read, >= 64 produced, no forced kernel choice
static inline bool wino_odd_ok(int c_read, int m, int H, int W) {
    return ((H | W) & 1) && H >= 3 && W >= 3 && c_read >= 128 && m >= 64 && cpg::opt(cpg::OPT_WINO_KERNEL) == cpg::OPT_UNSET && !cpg::opt_on(cpg::OPT_NO_WINO_ODD);
}
extern "C" int cpg_conv3x3_wino_ok(int N, int c_read, int m, int H, int W) {
    if (cpg::opt_on(cpg::OPT_NO_WINO)) return 0;
    if (c_read < 16 || m < 16 || N < 1) return 0;           // (any channel count: wg_chunk_base)
    if (((H | W) & 1) && !wino_odd_ok(c_read, m, H, W)) return 0;
    const int tiles_img = ((H + 1) / 2) * ((W + 1) / 2);
    const int span = (WG_T + tiles_img - 1) / tiles_img + 1;
    if ((int64_t)N * tiles_img + 64 >= (1ll << 31)) return 0;               // 32-bit tile indices in the kernels
    return (int64_t)span * std::max(c_read, m) * H * W * 4 < (1ll << 31);   // (a unit's images behind one buffer descriptor: read and written)
}

extern "C" size_t cpg_conv3x3_wino_pack_bytes(int c_read, int m) {      // (the 64-channel blocking pads m further: covers both)
    return (size_t)pad_to(m, 64) * pad_to(c_read, WG_CK) * 16 * sizeof(float);
}

static inline int wino_variant(int c_read, int m, bool stats = true);
static void wino_geom(WgGeom &g, int N, int c_read, int m, int H, int W) {
    g = WgGeom{};
    g.N = N, g.C = c_read, g.H = H, g.W = W, g.M = m;
    g.th = (H + 1) / 2, g.tw = (W + 1) / 2, g.tiles_img = g.th * g.tw;
    g.inv_timg = 1.0f / (float)g.tiles_img, g.inv_tw = 1.0f / (float)g.tw;
    g.tiles_total = (int64_t)N * g.tiles_img;
    g.nkb = pad_to(m, 32) / 32, g.nch = pad_to(c_read, WG_CK) / WG_CK;
    g.span = (W1_T + g.tiles_img - 1) / g.tiles_img + 1;
}
static inline bool wino_sh(const WgGeom &g) { return ((g.nkb + 1) / 2) % 2 == 0 && cpg::opt_or(cpg::OPT_WG3_SHARE, 1) != 0; }
// workspace behind the packed filter for the partial outputs of a tail launch (0: this launch has none) -- part of cpg_conv2d_workspace_bytes
extern "C" size_t cpg_conv3x3_wino_tail_bytes(int N, int c_read, int m, int H, int W) {
    if (((H | W) & 1) || !cpg_conv3x3_wino_ok(N, c_read, m, H, W) || wino_variant(c_read, m, false) != 3 /* WV_PAIR64 */) return 0;
    WgGeom g;
    wino_geom(g, N, c_read, m, H, W);
    const int64_t blocks = ((g.tiles_total + W1_T - 1) / W1_T) * ((g.nkb + 1) / 2);
    if (blocks > 0x7FFFFFFFll) return 0;
    const WgTail t = wino_tail_plan(g, blocks, wino_sh(g));
    return t.on ? t.bytes + 256 : 0;
}

// Which Winograd forward / input-gradient kernel a launch uses.  Default: k_wg3 (two waves per unit, 64 output channels each) when the
// layer reads and produces at least 64 channels, else k_wg1 (one wave per unit).  (Until round 4 the launches WITH the BatchNorm-statistics
// epilogue switched at 128 channels read: k_wg3's statistics epilogue was the heavier one.  After round 3's work on it -- sums per
// 32-channel half, scalar-base stores, the third filter buffer -- the interleaved A/B (tools/conv_bench.py --only fwdstats --ab
// CPG_WINO_KERNEL=-,64) reads 64 -> 64 @224 4.716 -> 4.546 ms and 64 -> 128 @112 2.277 -> 2.217 ms for k_wg3.)
// CPG_WINO_KERNEL = wave | pair | 64 | block forces k_wg1 / k_wg2 / k_wg3 / the cooperative block kernel (A/B experiments, tests).
enum { WV_BLOCK = 0, WV_WAVE = 1, WV_PAIR = 2, WV_PAIR64 = 3 };
static inline int wino_variant(int c_read, int m, bool stats) {
    if (const int forced = cpg::opt(cpg::OPT_WINO_KERNEL); forced != cpg::OPT_UNSET)      // (block | wave | pair | 64 -> WV_*)
        return forced >= WV_BLOCK && forced <= WV_PAIR64 ? forced : WV_WAVE;
    (void)stats;
    if (c_read < 64 || m < 64) return WV_WAVE;
    // Channel counts that are no multiple of 64 (the grown networks: 78 / 156 / 313 / 627 outputs): a unit of k_wg3 produces 64 channels, one
    // of k_wg1 32, and the padding of the last block is MFMA time.  At equal padding k_wg3 is the faster kernel by 5-8 % (shared transform),
    // so the one-wave kernel takes the layer when its blocks waste at least 10 % less: m = 78 (96 vs 128 computed) 9.88 -> 7.73 ms,
    // m = 156 (160 vs 192) 6.37 -> 5.81 ms; m = 313 / 627 (both 320 / 640) stay on k_wg3: 5.21 vs 5.49 ms  (conv_bench --width-multiplier 1.5).
    const int p64 = (m + 63) / 64 * 64, p32 = (m + 31) / 32 * 32;
    return p64 * 10 > p32 * 11 ? WV_WAVE : WV_PAIR64;
}

#ifdef WG_TIMING
extern "C" int cpg_debug_wg_timing(unsigned long long *dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wg_dbg), (size_t)n * 8 * sizeof(unsigned long long));
}
#endif
// CPG_WINO_PERSIST=0: one block per logical block instead of the persistent grid (A/B experiments)
static inline bool wino_persist() {
    return cpg::opt_or(cpg::OPT_WINO_PERSIST, 1) != 0;
}

// How many resident grids' worth of persistent blocks a launch gets: 8.  One grid (every block resident from the start, ~200 units
// per wave on the big layers) and 4, 8 or 16 grids measured the same within 0.5 % on every VGG16 layer -- a block that runs a
// dozen units has amortised its dispatch --, but with one exactly-resident grid a launch that finds some CUs taken (RCCL's
// kernels on the communication stream) would run its last blocks alone afterwards; with 8 the dispatcher balances whatever
// share of the chip is free to within 1/8 of a round.  CPG_WINO_GRIDS overrides (A/B experiments).
static inline int wino_grids() {
    return std::max(1, cpg::opt_or(cpg::OPT_WINO_GRIDS, 8));
}

// number of BatchNorm-statistics tiles per channel of a forward launch (stats[m][tiles][2])
extern "C" int cpg_conv3x3_wino_tiles(int N, int c_read, int m, int H, int W) {
    const int64_t tiles = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2);
    const int v = ((H | W) & 1) ? WV_PAIR64 : wino_variant(c_read, m);
    if (v == WV_PAIR || v == WV_PAIR64) return (int)(2 * ((tiles + W1_T - 1) / W1_T));     // every wave of a pair is its own tile
    const int per = v == WV_WAVE ? W1_T : WG_T;
    return (int)((tiles + per - 1) / per);
}

// y[N][m][H][W] = conv3x3(x[N][c_read][H][W], W .* bin(pm)) (+ bias); dgrad: x = gy, the filter transposed and flipped.
// w is the layer's [K][C][3][3] weight.  stats (forward only, may be null): [m][tiles][2] partial sums for the BatchNorm.
static int wino_run(int dgrad, int N, int c_read, int m, int H, int W, int K, int C, const float *x, const float *w, const float *pm,
                    float thr, const float *bias, float *y, float *stats, void *ws, size_t ws_bytes, hipStream_t stream, const WgBnEval *bne,
                    const float *addend = nullptr);

extern "C" int cpg_conv3x3_wino_run(int dgrad, int N, int c_read, int m, int H, int W, int K, int C, const float *x,
                                    const float *w, const float *pm, float thr, const float *bias, float *y, float *stats,
                                    void *ws, size_t ws_bytes, hipStream_t stream) {
    return wino_run(dgrad, N, c_read, m, H, W, K, C, x, w, pm, thr, bias, y, stats, ws, ws_bytes, stream, nullptr);
}

// input gradient + addend (gx = dgrad(gy) + addend): the two-wave kernel's launches (>= 64 channels on both sides, or an odd map)
extern "C" int cpg_conv3x3_wino_dgrad_add_ok(int N, int c_read, int m, int H, int W) {
    if (!cpg_conv3x3_wino_ok(N, c_read, m, H, W)) return 0;
    return (((H | W) & 1) || wino_variant(c_read, m, false) == WV_PAIR64) ? 1 : 0;
}
extern "C" int cpg_conv3x3_wino_dgrad_add(int N, int c_read, int m, int H, int W, int K, int C, const float *gy, const float *w, const float *pm,
                                          float thr, const float *addend, float *gx, void *ws, size_t ws_bytes, hipStream_t stream) {
    if (!cpg_conv3x3_wino_dgrad_add_ok(N, c_read, m, H, W) || addend == nullptr)
        return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_dgrad_add(winograd): shape not supported");
    return wino_run(1, N, c_read, m, H, W, K, C, gy, w, pm, thr, nullptr, gx, nullptr, ws, ws_bytes, stream, nullptr, addend);
}

// 1: the inference epilogue is available on the Winograd kernel (the one-wave kernel only)
extern "C" int cpg_conv3x3_wino_eval_ok(int N, int c_read, int m, int H, int W) {
    return wino_variant(c_read, m) != WV_BLOCK && cpg_conv3x3_wino_ok(N, c_read, m, H, W);
}

// forward with eval-mode BatchNorm (+ ReLU) in the epilogue; live (may be null): live_words ints, zeroed here, layout of k_c3_pack
extern "C" int cpg_conv3x3_wino_run_bn_eval(int N, int C, int K, int H, int W, const float *x, const float *w, const float *pm, float thr,
                                            const float *bias, const float *gamma, const float *beta, const float *mean, const float *var,
                                            float eps, int relu, int *live, size_t live_words, float *y, void *ws, size_t ws_bytes,
                                            hipStream_t stream) {
    if (live != nullptr) {
        hipError_t e = hipMemsetAsync(live, 0, live_words * sizeof(int), stream);
        if (e != hipSuccess) return hip_status(e, "cpg_conv2d_fwd_bn_eval(winograd)");
    }
    const WgBnEval bne{gamma, beta, mean, var, eps, relu, live, pad_to(K, 128)};
    return wino_run(0, N, C, K, H, W, K, C, x, w, pm, thr, bias, y, nullptr, ws, ws_bytes, stream, &bne);
}

static int wino_run(int dgrad, int N, int c_read, int m, int H, int W, int K, int C, const float *x, const float *w, const float *pm,
                    float thr, const float *bias, float *y, float *stats, void *ws, size_t ws_bytes, hipStream_t stream, const WgBnEval *bne,
                    const float *addend) {
    const char *what = dgrad ? "cpg_conv2d_dgrad(winograd)" : "cpg_conv2d_fwd(winograd)";
    // addend (input-gradient launches of the two-wave kernel only: cpg_conv3x3_wino_dgrad_add_ok): added to the result in the epilogue
    // (k_wg3<.., ADD>; the tail pieces' share in k_wg_tail_reduce)
    if (addend != nullptr && (!dgrad || bias != nullptr || bne != nullptr || stats != nullptr))
        return fail(CPG_E_INVALID, "%s: an addend rides in plain input-gradient launches only", what);
    const size_t need = cpg_conv3x3_wino_pack_bytes(c_read, m);
    if (ws == nullptr || ws_bytes < need) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    CPG_REQUIRE((((uintptr_t)ws) & 15) == 0, "%s: workspace must be 16-byte aligned", what);
    const bool odd = ((H | W) & 1) != 0;
    const int variant = odd ? WV_PAIR64 : wino_variant(c_read, m, stats != nullptr);
    if (variant != WV_BLOCK) {
        WgGeom g;
        wino_geom(g, N, c_read, m, H, W);
        float *up = (float *)ws;
        const WgBnEval none{nullptr, nullptr, nullptr, nullptr, 0.0f, 0, nullptr, 0};
        int ps = 0;
        if (bne == nullptr) {      // (the inference epilogue's pack also writes the liveness flags: never taken from a caller's operand)
            const float *pre = nullptr;
            ps = cpg::pack_site(cpg::PackJob{2, K, C, m, c_read, g.nch, dgrad ? 1 : 0, (long long)((m + 31) / 32) * g.nch * 32 * WG_CK, need}, &pre, what);
            if (ps == 1) return CPG_OK;
            if (ps < 0) return ps;
            if (ps == 2) up = const_cast<float *>(pre);
        } else if (cpg::pack_query()) {
            return CPG_OK;
        }
        if (ps != 2)
            hipLaunchKernelGGL(k_wg1_pack, dim3(stream_grid((int64_t)g.nkb * g.nch * 32 * WG_CK, 256)), dim3(256), 0, stream, w, pm, thr, up,
                               K, C, m, c_read, g.nch, dgrad ? 1 : 0, bne ? bne->live : nullptr, bne ? bne->Mp : 0);
        const int64_t runs = (g.tiles_total + W1_T - 1) / W1_T;
        const bool persist = wino_persist();
        if (variant == WV_PAIR64) {
            int64_t blocks = runs * ((g.nkb + 1) / 2);
            if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "%s: grid too large", what);
            g.nblocks = (unsigned)blocks;
            if (persist) blocks = std::min<int64_t>(blocks, (int64_t)wino_grids() * 2 * kCUs);
            if (odd) {
                if (bne != nullptr)
                    hipLaunchKernelGGL((k_wg3<false, false, true, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, nullptr, *bne);
                else if (dgrad && addend != nullptr)
                    hipLaunchKernelGGL((k_wg3<true, false, false, true, false, false, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, addend, y, nullptr, none);
                else if (dgrad)
                    hipLaunchKernelGGL((k_wg3<true, false, false, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, nullptr, none);
                else if (stats != nullptr)
                    hipLaunchKernelGGL((k_wg3<false, true, false, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, stats, none);
                else
                    hipLaunchKernelGGL((k_wg3<false, false, false, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, nullptr, none);
                CPG_CHECK_LAUNCH(what);
                return CPG_OK;
            }
            // The leftover units of the last round (k_wg3<.., SPLIT>): the full rounds run as before on logical blocks 0 .. first_lb - 1, the rest
            // as pieces of the channel loop into the workspace behind the packed filter, k_wg_tail_reduce adds the pieces (+ bias) into y.
            const bool sh = bne == nullptr && wino_sh(g);
            WgTail tail{};
            const size_t tail_off = (need + 255) / 256 * 256;
            if (bne == nullptr) tail = wino_tail_plan(g, g.nblocks, sh);      // (with the BatchNorm statistics too: k_wg_tail_reduce<true> sums them)
            if (tail.on && ws_bytes < tail_off + tail.bytes) tail.on = false;       // (a caller with the round-4 workspace: one launch, as before)
            WgGeom gt = g;
            float *part0 = nullptr;
            if (tail.on) {
                g.nblocks = tail.first_lb;
                gt.split_first = tail.first_lb, gt.split_s = tail.S, gt.split_nch = tail.nch_per, gt.split_stride = tail.stride;
                gt.nblocks = (unsigned)((sh ? 2 : 1) * tail.left * tail.S);
                // piece 0's slice, addressed like y: its first image is n_first
                part0 = reinterpret_cast<float *>(reinterpret_cast<uintptr_t>(ws) + tail_off) - (int64_t)tail.n_first * g.M * H * W;
            }
            auto finish_tail = [&]() {
                const dim3 rg(stream_grid((int64_t)(tail.end_lb - tail.first_lb) * 64 * W1_T, 256));
                if (stats != nullptr)
                    hipLaunchKernelGGL(k_wg_tail_reduce<true>, rg, dim3(256), 0, stream, gt, part0, bias, y, tail.first_lb, tail.end_lb, stats, (const float *)nullptr);
                else
                    hipLaunchKernelGGL(k_wg_tail_reduce<false>, rg, dim3(256), 0, stream, gt, part0, bias, y, tail.first_lb, tail.end_lb, (float *)nullptr, addend);
            };
            // two units per block sharing the input transform (k_wg3<..., SH>): training launches whose 64-channel blocks pair up
            if (sh) {
                int64_t pairs = (int64_t)g.nblocks / 2;
                if (persist) pairs = std::min<int64_t>(pairs, (int64_t)wino_grids() * kCUs);       // (one four-wave block per CU is resident)
                if (dgrad && addend != nullptr)
                    hipLaunchKernelGGL((k_wg3<true, false, false, false, true, false, true>), dim3((unsigned)pairs), dim3(256), 0, stream, g, x, up, addend, y, nullptr, none);
                else if (dgrad)
                    hipLaunchKernelGGL((k_wg3<true, false, false, false, true>), dim3((unsigned)pairs), dim3(256), 0, stream, g, x, up, bias, y, nullptr, none);
                else if (stats != nullptr)
                    hipLaunchKernelGGL((k_wg3<false, true, false, false, true>), dim3((unsigned)pairs), dim3(256), 0, stream, g, x, up, bias, y, stats, none);
                else
                    hipLaunchKernelGGL((k_wg3<false, false, false, false, true>), dim3((unsigned)pairs), dim3(256), 0, stream, g, x, up, bias, y, nullptr, none);
                if (tail.on) {
                    const unsigned tb = (unsigned)(tail.left * tail.S);
                    if (dgrad)
                        hipLaunchKernelGGL((k_wg3<true, false, false, false, true, true>), dim3(tb), dim3(256), 0, stream, gt, x, up, nullptr, part0, nullptr, none);
                    else
                        hipLaunchKernelGGL((k_wg3<false, false, false, false, true, true>), dim3(tb), dim3(256), 0, stream, gt, x, up, nullptr, part0, nullptr, none);
                    finish_tail();
                }
                CPG_CHECK_LAUNCH(what);
                return CPG_OK;
            }
            if (tail.on) blocks = std::min<int64_t>(blocks, (int64_t)g.nblocks);
            if (bne != nullptr)
                hipLaunchKernelGGL((k_wg3<false, false, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, nullptr, *bne);
            else if (dgrad && addend != nullptr)
                hipLaunchKernelGGL((k_wg3<true, false, false, false, false, false, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, addend, y, nullptr, none);
            else if (dgrad)
                hipLaunchKernelGGL((k_wg3<true, false>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, nullptr, none);
            else if (stats != nullptr)
                hipLaunchKernelGGL((k_wg3<false, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, stats, none);
            else
                hipLaunchKernelGGL((k_wg3<false, false>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, nullptr, none);
            if (tail.on) {
                const unsigned tb = (unsigned)(tail.left * tail.S);
                if (dgrad)
                    hipLaunchKernelGGL((k_wg3<true, false, false, false, false, true>), dim3(tb), dim3(128), 0, stream, gt, x, up, nullptr, part0, nullptr, none);
                else
                    hipLaunchKernelGGL((k_wg3<false, false, false, false, false, true>), dim3(tb), dim3(128), 0, stream, gt, x, up, nullptr, part0, nullptr, none);
                finish_tail();
            }
            CPG_CHECK_LAUNCH(what);
            return CPG_OK;
        }
        if (variant == WV_PAIR) {
            const int64_t blocks = runs * g.nkb;
            if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "%s: grid too large", what);
            if (bne != nullptr)
                hipLaunchKernelGGL((k_wg2<false, false, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, nullptr, *bne);
            else if (dgrad)
                hipLaunchKernelGGL((k_wg2<true, false>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, nullptr, none);
            else if (stats != nullptr)
                hipLaunchKernelGGL((k_wg2<false, true>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, stats, none);
            else
                hipLaunchKernelGGL((k_wg2<false, false>), dim3((unsigned)blocks), dim3(128), 0, stream, g, x, up, bias, y, nullptr, none);
            CPG_CHECK_LAUNCH(what);
            return CPG_OK;
        }
        int64_t blocks = (runs + 3) / 4 * g.nkb;
        if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "%s: grid too large", what);
        g.nblocks = (unsigned)blocks;
        if (persist) blocks = std::min<int64_t>(blocks, (int64_t)wino_grids() * kCUs);
        if (bne != nullptr)
            hipLaunchKernelGGL((k_wg1<false, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, up, bias, y, nullptr, *bne);
        else if (dgrad)
            hipLaunchKernelGGL((k_wg1<true, false>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, up, bias, y, nullptr, none);
        else if (stats != nullptr)
            hipLaunchKernelGGL((k_wg1<false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, up, bias, y, stats, none);
        else
            hipLaunchKernelGGL((k_wg1<false, false>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, up, bias, y, nullptr, none);
        CPG_CHECK_LAUNCH(what);
        return CPG_OK;
    }
    if (bne != nullptr) return fail(CPG_E_UNSUPPORTED, "%s: the block kernels have no inference epilogue", what);
    if (cpg::pack_query()) return CPG_OK;        // (the block kernels pack for themselves: no job recorded)
    const int nw = wino_nw(c_read, m), BK = 8 * nw;
    WgGeom g;
    g.N = N, g.C = c_read, g.H = H, g.W = W, g.M = m;
    g.th = H / 2, g.tw = W / 2, g.tiles_img = g.th * g.tw;
    g.inv_timg = 1.0f / (float)g.tiles_img, g.inv_tw = 1.0f / (float)g.tw;
    g.tiles_total = (int64_t)N * g.tiles_img;
    g.nkb = pad_to(m, BK) / BK, g.nch = pad_to(c_read, WG_CK) / WG_CK;
    g.span = (WG_T + g.tiles_img - 1) / g.tiles_img + 1;
    float *up = (float *)ws;
    hipLaunchKernelGGL(k_wg_pack, dim3(stream_grid((int64_t)g.nkb * g.nch * BK * WG_CK, 256)), dim3(256), 0, stream, w, pm, thr, up,
                       K, C, m, c_read, g.nch, dgrad ? 1 : 0, BK);
    const int64_t tblocks = cpg_conv3x3_wino_tiles(N, c_read, m, H, W);
    const int rc = nw == 8 ? wino_launch<8>(dgrad != 0, g, tblocks, x, up, bias, y, stats, stream)
                           : wino_launch<4>(dgrad != 0, g, tblocks, x, up, bias, y, stats, stream);
    if (rc != CPG_OK) return rc;
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}

// cpg_conv2d_pack's launch (declared in cpg_common.h): job b may be null
int cpg::pack_jobs_launch(const cpg::PackJob *ja, float *dst_a, const cpg::PackJob *jb, float *dst_b, const float *w, const float *pm, float thr,
                          hipStream_t stream) {
    auto args = [](const cpg::PackJob *j, float *dst) {
        if (j == nullptr) return PackArgs{0, 0, 0, 0, 0, 0, 0, 0, nullptr};
        return PackArgs{j->family, j->K, j->C, j->a, j->b, j->c, j->d, j->total, dst};
    };
    const PackArgs a = args(ja, dst_a), b = args(jb, dst_b);
    if (a.total + b.total <= 0) return CPG_OK;
    hipLaunchKernelGGL(k_pack_jobs, dim3(stream_grid(a.total + b.total, 256)), dim3(256), 0, stream, a, b, w, pm, thr);
    CPG_CHECK_LAUNCH("cpg_conv2d_pack");
    return CPG_OK;
}
