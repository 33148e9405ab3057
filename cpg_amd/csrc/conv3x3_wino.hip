// Masked 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2, 3x3) on fp32 MFMA: forward and input gradient.
//
//   Y(2x2 tile) = A^T [ (G g G^T) .* (B^T d B) ] A        (d: 4x4 input patch, g: 3x3 filter; Lavin & Gray 2015)
//
// 16 multiplies per 2x2 output tile and (c, k) pair instead of 36: the contraction over input channels becomes 16
// independent GEMMs  M_p[k][t] = sum_c U_p[k][c] * V_p[c][t]  (p = transform-domain position, t = tile), 2.25x fewer MFMAs
// than the direct kernels of conv3x3.hip for the same result up to fp32 rounding (the transforms only add, subtract and
// halve; measured 1-4e-6 of the output scale against fp64, the direct kernels 0.5-1e-6).  DESIGN.md section 4.9 has the
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
};

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
        const int m = kb * BK + kl, c = ch * WG_CK + cl;     // produced / read channel
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) g[r][s] = 0.0f;
        if (m < M && c < Cin) {
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
        const int soff = ch * WG_CK * HW * 4;
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

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// waves per block for a launch.  The 8-wave / 64-channel block does half the staging work per MFMA, but its eight waves run in
// lock step behind one barrier and it measured 3-4 % SLOWER than two independent 4-wave blocks per CU on every VGG16 layer
// (DESIGN.md section 4.9), so it is only reachable through CPG_WINO_NW=8 (A/B experiments, tests).
inline int wino_nw(int c_read, int m) {
    if (const char *f = getenv("CPG_WINO_NW")) return atoi(f) == 8 ? 8 : 4;
    return 4;
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
// eligibility of one launch (c_read channels contracted, m produced): even maps, channel chunks of 4, the staging offsets of
// the images a block can touch fit 31 bits
extern "C" int cpg_conv3x3_wino_ok(int N, int c_read, int m, int H, int W) {
    if (getenv("CPG_NO_WINO")) return 0;
    if (H % 2 || W % 2 || c_read % 4 || c_read < 16 || m < 16 || N < 1) return 0;
    const int tiles_img = (H / 2) * (W / 2);
    const int span = (WG_T + tiles_img - 1) / tiles_img + 1;
    return (int64_t)span * c_read * H * W * 4 < (1ll << 31);
}

extern "C" size_t cpg_conv3x3_wino_pack_bytes(int c_read, int m) {      // (the 64-channel blocking pads m further: covers both)
    return (size_t)pad_to(m, 64) * pad_to(c_read, WG_CK) * 16 * sizeof(float);
}

extern "C" int cpg_conv3x3_wino_tiles(int N, int H, int W) {
    return (int)(((int64_t)N * (H / 2) * (W / 2) + WG_T - 1) / WG_T);
}

// y[N][m][H][W] = conv3x3(x[N][c_read][H][W], W .* bin(pm)) (+ bias); dgrad: x = gy, the filter transposed and flipped.
// w is the layer's [K][C][3][3] weight.  stats (forward only, may be null): [m][tiles][2] partial sums for the BatchNorm.
extern "C" int cpg_conv3x3_wino_run(int dgrad, int N, int c_read, int m, int H, int W, int K, int C, const float *x,
                                    const float *w, const float *pm, float thr, const float *bias, float *y, float *stats,
                                    void *ws, size_t ws_bytes, hipStream_t stream) {
    const char *what = dgrad ? "cpg_conv2d_dgrad(winograd)" : "cpg_conv2d_fwd(winograd)";
    const size_t need = cpg_conv3x3_wino_pack_bytes(c_read, m);
    if (ws == nullptr || ws_bytes < need) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    CPG_REQUIRE((((uintptr_t)ws) & 15) == 0, "%s: workspace must be 16-byte aligned", what);
    const int nw = wino_nw(c_read, m), BK = 8 * nw;
    WgGeom g;
    g.N = N, g.C = c_read, g.H = H, g.W = W, g.M = m;
    g.th = H / 2, g.tw = W / 2, g.tiles_img = g.th * g.tw;
    g.tiles_total = (int64_t)N * g.tiles_img;
    g.nkb = pad_to(m, BK) / BK, g.nch = pad_to(c_read, WG_CK) / WG_CK;
    g.span = (WG_T + g.tiles_img - 1) / g.tiles_img + 1;
    float *up = (float *)ws;
    hipLaunchKernelGGL(k_wg_pack, dim3(stream_grid((int64_t)g.nkb * g.nch * BK * WG_CK, 256)), dim3(256), 0, stream, w, pm, thr, up,
                       K, C, m, c_read, g.nch, dgrad ? 1 : 0, BK);
    const int64_t tblocks = cpg_conv3x3_wino_tiles(N, H, W);
    const int rc = nw == 8 ? wino_launch<8>(dgrad != 0, g, tblocks, x, up, bias, y, stats, stream)
                           : wino_launch<4>(dgrad != 0, g, tblocks, x, up, bias, y, stats, stream);
    if (rc != CPG_OK) return rc;
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}
