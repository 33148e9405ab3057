// Masked R x R / stride 2 / pad R/2 convolution of an image (<= 3 channels) to 64 channels -- the stems of the other two topologies:
//   ResNet-50   conv1: 7x7 s2 p3, 3 -> 64 @224 -> 112  (models/resnet.py:126)        118 M MACs / image, 3.2 MB of output / image
//   SphereNet   conv1_1: 3x3 s2 p1 (+ bias), 3 -> 64 @112 -> 56 (models/spherenet.py:203)
// (SharableConv2d.forward, models/layers.py:98-109; the input gradient does not exist -- the network input needs none.)
//
// The generic gather kernel spent 1.42 ms forward and 2.63 ms in the weight gradient on the ResNet stem at batch 256; the layer's
// floor is the MFMA time of a K = 147 contraction (0.39 ms) beside 0.82 GB of output (0.13 ms at the copy rate).
//
// Forward, k_stem2_fwd<R>: conv3x3_stem.hip's design -- one WAVE owns (a tile of TR x 32 output pixels, one block of 32 output
// channels) at a time, no barriers, its A operands (W .* bin(pm)) live in registers for the whole launch, the input patch in
// wave-private LDS.  What the stride changes: output column j, tap s reads patch column 2 j + s, which as a lane stride of two words
// would put two lanes on every LDS bank; the patch rows are therefore stored DE-INTERLEAVED (even columns, then odd columns) and the
// taps of a row are taken in pairs (s = 2 sp + lh: the two k values of an MFMA step are the even / odd tap of pair sp) -- the B
// operand of step (c, r, sp) is then one ds_read_b32 at lane base (li + HALF lh) + an immediate.  R is odd, so every row has one
// padding tap with a zero weight (7 -> 8, 3 -> 4): K = 3 x 7 x 8 = 168 for the ResNet stem (84 steps instead of the minimal 74).
//
// Weight gradient, k_stem2_wgrad<R>: gW[co][(c, r, s)] = sum over pixels of gy[co][pix] x[c][2 oy + r - p][2 ox + s - p]: 64 x 147,
// K = 3.2 M pixels per image batch -- bound by streaming gy once.  Block = a range of tiles; all four waves hold the full 64 x
// (C R R padded to 32-column fragments) result for their quarter of a tile's pixel pairs; A = gy through LDS ([co][pixel]), B = the
// de-interleaved x patch read at a per-lane (c, r, s) offset + the pixel pair's immediate.  Partials per block, reduced by
// k_split_reduce (with the autograd epilogue gW = g bin(pm), gPM = g W).
#include <algorithm>
#include <type_traits>
#include "igemm_core.h"

using namespace cpg;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int R_>
struct S2Cfg {
    static constexpr int R = R_, PAD = R_ / 2, CMAX = 3;
    static constexpr int TR = 4, TW = 32;                        // output rows / columns of a tile
    static constexpr int PROWS = 2 * TR + R - 2;                 // input rows a tile reads
    static constexpr int PCOLS = 2 * TW + R - 2;                 // input columns
    static constexpr int HALF = 48, PW = 2 * HALF;               // even columns at [0, HALF), odd ones at [HALF, PW); HALF = 16 (mod 32):
    static_assert((PCOLS + 1) / 2 <= HALF && HALF % 32 == 16, "the two column parities must land on disjoint bank halves");
    static constexpr int SP = (R + 1) / 2;                       // tap pairs per filter row (the last pair's odd tap is padding)
    static constexpr int KS = CMAX * R * SP;                     // MFMA steps
    static constexpr int ITEMS = CMAX * PROWS;                   // (channel, patch row) staging items
    static constexpr int PATCH = ITEMS * PW;
    static constexpr int XTRA = PCOLS - 64;                      // columns past the first 64 (one more, sparsely used, load per item)
    static_assert(XTRA > 0 && XTRA <= 64, "two loads per item");
};

struct Stem2Geom {
    int N, C, K, H, W, OH, OW;
    int tiles_x, tiles_y;
    unsigned ntiles;
};

template <int R, bool STATS>
__global__ __launch_bounds__(256, 2)
void k_stem2_fwd(Stem2Geom g, const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ pm, float thr,
                 const float *__restrict__ bias, float *__restrict__ y, float *__restrict__ stats) {
    using Cfg = S2Cfg<R>;
    __shared__ float smem_all[4 * Cfg::PATCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    float *smem = smem_all + wave * Cfg::PATCH;
    const int HW = g.H * g.W, HWo = g.OH * g.OW, CRR = g.C * R * R;

    // work item = (tile, block of 32 output channels); the wave stride is even, so a wave keeps ONE channel block: its A operands
    const unsigned nwaves = gridDim.x * 4, wid = blockIdx.x * 4 + wave, nitems = g.ntiles * 2;
    if (wid >= nitems) return;                  // (no barriers anywhere: a wave may leave)
    const int mb = wid & 1;
    float A[Cfg::KS];
#pragma unroll
    for (int t = 0; t < Cfg::KS; ++t) {
        const int c = t / (R * Cfg::SP), r = (t / Cfg::SP) % R, s = 2 * (t % Cfg::SP) + lh;
        const int co = mb * 32 + li;
        float v = 0.0f;
        if (s < R && c < g.C && co < g.K) {
            const int off = co * CRR + (c * R + r) * R + s;
            v = w[off];
            if (pm != nullptr) v *= binarize(pm[off], thr);
        }
        A[t] = v;
    }
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int co = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        bv[e] = (bias != nullptr && co < g.K) ? bias[co] : 0.0f;
    }
    for (int i = lane; i < Cfg::PATCH; i += 64) smem[i] = 0.0f;      // the padding taps read slots no load ever writes: keep them finite

    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, g.N * g.C * HW * 4, 0x00020000);
    constexpr int kOOR = (int)0x80000000;
    // columns 0..63 of an item: one load per item (lane = column); the XTRA columns past 64 of ALL items: lane + 64 q -> (item, column)
    constexpr int NXB = (Cfg::ITEMS * Cfg::XTRA + 63) / 64;
    float pa[Cfg::ITEMS], pb[NXB];
    auto tile_coords = [&](unsigned tile, int &n, int &y0, int &x0) {
        const unsigned per_img = (unsigned)(g.tiles_x * g.tiles_y);
        n = (int)(tile / per_img);
        const unsigned r = tile % per_img;
        y0 = (int)(r / (unsigned)g.tiles_x) * Cfg::TR, x0 = (int)(r % (unsigned)g.tiles_x) * Cfg::TW;
    };
    auto issue_loads = [&](unsigned tile) {
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
        const int cola = 2 * x0 - Cfg::PAD + lane;
        const bool oka = (unsigned)cola < (unsigned)g.W;
#pragma unroll
        for (int q = 0; q < Cfg::ITEMS; ++q) {
            const int c = q / Cfg::PROWS, row = 2 * y0 - Cfg::PAD + q % Cfg::PROWS;
            const bool rok = c < g.C && (unsigned)row < (unsigned)g.H;
            const int base = ((n * g.C + c) * g.H + row) * g.W;
            pa[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, (rok && oka) ? (base + cola) * 4 : kOOR, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < NXB; ++q) {
            const int e = lane + 64 * q, it = e / Cfg::XTRA, xc = e % Cfg::XTRA;
            const int c = it / Cfg::PROWS, row = 2 * y0 - Cfg::PAD + it % Cfg::PROWS, col = 2 * x0 - Cfg::PAD + 64 + xc;
            const bool ok = it < Cfg::ITEMS && c < g.C && (unsigned)row < (unsigned)g.H && (unsigned)col < (unsigned)g.W;
            pb[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, ok ? (((n * g.C + c) * g.H + row) * g.W + col) * 4 : kOOR, 0, 0));
        }
    };
    // patch column pc -> slot (pc & 1) * HALF + (pc >> 1)
    const int slot_a = (lane & 1) * Cfg::HALF + (lane >> 1);
    auto write_patch = [&]() {
#pragma unroll
        for (int q = 0; q < Cfg::ITEMS; ++q) smem[q * Cfg::PW + slot_a] = pa[q];
#pragma unroll
        for (int q = 0; q < NXB; ++q) {
            const int e = lane + 64 * q, it = e / Cfg::XTRA, pc = 64 + e % Cfg::XTRA;
            if (it < Cfg::ITEMS) smem[it * Cfg::PW + (pc & 1) * Cfg::HALF + (pc >> 1)] = pb[q];
        }
    };

    const int b_lane = lh * Cfg::HALF + li;
    issue_loads(wid >> 1);
    for (unsigned item = wid; item < nitems; item += nwaves) {
        const unsigned tile = item >> 1;
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
        write_patch();
        if (item + nwaves < nitems) issue_loads((item + nwaves) >> 1);
        float s1[16], s2[16];
        if (STATS) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s1[e] = s2[e] = 0.0f;
        }
        const __amdgpu_buffer_rsrc_t srd_y =
            __builtin_amdgcn_make_buffer_rsrc((void *)(y + (int64_t)n * g.K * HWo), 0, g.K * HWo * 4, 0x00020000);
        const bool cok = x0 + li < g.OW;
        const int pix0 = (y0 * g.OW + x0 + li) * 4 + (lh * 4 + mb * 32) * HWo * 4;
#pragma unroll 1
        for (int j = 0; j < Cfg::TR; ++j) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
            const float *prow = smem + 2 * j * Cfg::PW + b_lane;
#pragma unroll
            for (int t = 0; t < Cfg::KS; ++t) {
                const int c = t / (R * Cfg::SP), r = (t / Cfg::SP) % R, sp = t % Cfg::SP;
                const float b = prow[(c * Cfg::PROWS + r) * Cfg::PW + sp];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t], b, acc, 0, 0, 0);
            }
            const bool pok = cok && y0 + j < g.OH;
            const int voff = pok ? pix0 + j * g.OW * 4 : kOOR;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cu = (e & 3) + 8 * (e >> 2);                  // + 4 lh + 32 mb: in voff
                const float v = acc[e] + bv[e];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), srd_y, voff, cu * HWo * 4, 0);
                if (STATS) {
                    const float vm = pok ? v : 0.0f;
                    s1[e] += vm;
                    s2[e] = fmaf(vm, vm, s2[e]);
                }
            }
        }
        if (STATS) {
            half_wave_sum8(s1), half_wave_sum8(s1 + 8), half_wave_sum8(s2), half_wave_sum8(s2 + 8);
            if (li == kHalfSumLane) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    f32x2 o;
                    o[0] = s1[e], o[1] = s2[e];
                    if (co < g.K) *reinterpret_cast<f32x2 *>(stats + ((int64_t)co * g.ntiles + tile) * 2) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------ weight gradient
// part[split][co][j], j = (c R + r) R + s.  Tile = TR x 32 pixels of gy (= 64 pixel pairs per wave quarter ... see below).
template <int R_>
struct S2WCfg {
    using F = S2Cfg<R_>;
    static constexpr int R = R_, J = 3 * R * R, NF = (J + 31) / 32;          // fragment columns (147 -> 5 fragments)
    static constexpr int TR = F::TR, TW = F::TW, NPIX = TR * TW, LDG = NPIX + 1;
    static constexpr int G_ELEMS = 64 * LDG;
    static constexpr int X_ELEMS = F::PATCH + 2 * TR * F::PW;                  // + all-zero rows for the unused fragment columns (read at every pixel offset)
    static constexpr int SMEM = G_ELEMS + X_ELEMS;
};

template <int R>
__global__ __launch_bounds__(256, 2)
void k_stem2_wgrad(Stem2Geom g, int tiles_per_split, const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ part) {
    using Cfg = S2WCfg<R>;
    using F = typename Cfg::F;
    __shared__ float smem[Cfg::SMEM];
    float *gs = smem, *xs = smem + Cfg::G_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wco = sub & 1, khalf = sub >> 1;                   // waves 0/1: output channels 0-31 / 32-63 for tile rows 0-1; waves 2/3: rows 2-3
    const int HW = g.H * g.W, HWo = g.OH * g.OW;
    const unsigned t0 = blockIdx.x * (unsigned)tiles_per_split, t1 = min(g.ntiles, t0 + (unsigned)tiles_per_split);
    const int J = g.C * R * R;

    for (int i = tid; i < Cfg::X_ELEMS; i += 256) xs[i] = 0.0f;
    // fragment column j = (c, r, s) -> fixed patch offset (de-interleaved columns); unused columns read the zero row
    int joff[Cfg::NF];
#pragma unroll
    for (int f = 0; f < Cfg::NF; ++f) {
        const int j = f * 32 + li;
        const int c = j / (R * R), r = (j / R) % R, s = j % R;
        joff[f] = j < J ? (c * F::PROWS + r) * F::PW + (s & 1) * F::HALF + (s >> 1) : F::PATCH;
    }
    const int a_base = (wco * 32 + li) * Cfg::LDG + lh;

    f32x16 acc[Cfg::NF];
#pragma unroll
    for (int f = 0; f < Cfg::NF; ++f)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[f][e] = 0.0f;

    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, g.N * g.C * HW * 4, 0x00020000);
    constexpr int kOOR = (int)0x80000000;
    for (unsigned tile = t0; tile < t1; ++tile) {
        const unsigned per_img = (unsigned)(g.tiles_x * g.tiles_y);
        const int n = (int)(tile / per_img);
        const unsigned rr = tile % per_img;
        const int y0 = (int)(rr / (unsigned)g.tiles_x) * Cfg::TR, x0 = (int)(rr % (unsigned)g.tiles_x) * Cfg::TW;
        __syncthreads();                                     // previous tile fully consumed
        // gy: wave `sub` stages channels sub + 4 i (i < 16), lanes own pixels lane and lane + 64 of the 4 x 32 tile
        const float *gimg = gy + (int64_t)n * g.K * HWo;
        float rg[2][16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pix = lane + 64 * h, r = pix / Cfg::TW, c = pix % Cfg::TW;
            const bool pv = y0 + r < g.OH && x0 + c < g.OW;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = sub + 4 * i;
                rg[h][i] = (pv && co < g.K) ? gimg[(int64_t)co * HWo + (y0 + r) * g.OW + x0 + c] : 0.0f;
            }
        }
        // x patch: items (c, patch row) x 69 columns; thread t takes item t / 4 ... simple strided sweep, range-checked loads
        constexpr int NXE = F::ITEMS * F::PCOLS, NXL = (NXE + 255) / 256;
        float rx[NXL];
        int xd[NXL];
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int e = tid + 256 * i;
            const int q = e / F::PCOLS, pc = e % F::PCOLS;
            const int c = q / F::PROWS, row = 2 * y0 - F::PAD + q % F::PROWS, col = 2 * x0 - F::PAD + pc;
            const bool ok = e < NXE && c < g.C && (unsigned)row < (unsigned)g.H && (unsigned)col < (unsigned)g.W;
            rx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, ok ? (((n * g.C + c) * g.H + row) * g.W + col) * 4 : kOOR, 0, 0));
            xd[i] = e < NXE ? q * F::PW + (pc & 1) * F::HALF + (pc >> 1) : -1;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 16; ++i) gs[(sub + 4 * i) * Cfg::LDG + lane + 64 * h] = rg[h][i];
#pragma unroll
        for (int i = 0; i < NXL; ++i)
            if (xd[i] >= 0) xs[xd[i]] = rx[i];
        __syncthreads();
        // this wave's half of the tile's rows; k of the MFMA = a pair of horizontally adjacent pixels (2 c2 + lh)
#pragma unroll
        for (int r2 = 0; r2 < Cfg::TR / 2; ++r2) {
            const int r = khalf * (Cfg::TR / 2) + r2;
#pragma unroll
            for (int c2 = 0; c2 < Cfg::TW / 2; ++c2) {
                const float a = gs[a_base + r * Cfg::TW + 2 * c2];
                // pixel (r, 2 c2 + lh): patch row 2 r (+ tap row, in joff), patch column 2 (2 c2 + lh) + s -> slot + 2 c2 + lh
                const float *xb = xs + 2 * r * F::PW + 2 * c2 + lh;
#pragma unroll
                for (int f = 0; f < Cfg::NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[joff[f]], acc[f], 0, 0, 0);
            }
        }
    }
    // combine the two row halves through LDS, then write part[split][co][j]
    __syncthreads();
    float *red = smem;                                       // [2 channel blocks][NF][16][64]
    if (khalf == 1) {
#pragma unroll
        for (int f = 0; f < Cfg::NF; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[((wco * Cfg::NF + f) * 16 + e) * 64 + lane] = acc[f][e];
    }
    __syncthreads();
    if (khalf == 0) {
        float *dst = part + (int64_t)blockIdx.x * g.K * J;
#pragma unroll
        for (int f = 0; f < Cfg::NF; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = acc[f][e] + red[((wco * Cfg::NF + f) * 16 + e) * 64 + lane];
                const int co = wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, j = f * 32 + li;
                if (co < g.K && j < J) dst[(int64_t)co * J + j] = v;
            }
    }
}

bool stem2_geom(const cpg_conv_desc *d, Stem2Geom &g) {
    if (opt_on(OPT_NO_STEM)) return false;                       // (A/B experiments, tests)
    if (!(d->R == d->S && (d->R == 7 || d->R == 3) && d->stride_h == 2 && d->stride_w == 2 && d->pad_h == d->R / 2 && d->pad_w == d->R / 2 &&
          d->dil_h == 1 && d->dil_w == 1 && d->groups == 1))
        return false;
    if (d->N < 1 || d->C < 1 || d->C > 3 || d->K != 64 || d->H < 2 || d->W < 2) return false;
    const int OH = (d->H + 2 * d->pad_h - d->R) / 2 + 1, OW = (d->W + 2 * d->pad_w - d->S) / 2 + 1;
    if ((int64_t)d->N * d->C * d->H * d->W * 4 >= (1ll << 31) || (int64_t)d->K * OH * OW * 4 >= (1ll << 31)) return false;
    g.N = d->N, g.C = d->C, g.K = d->K, g.H = d->H, g.W = d->W, g.OH = OH, g.OW = OW;
    g.tiles_x = (OW + 31) / 32, g.tiles_y = (OH + 3) / 4;
    const int64_t nt = (int64_t)d->N * g.tiles_x * g.tiles_y;
    if (nt >= (1ll << 30)) return false;
    g.ntiles = (unsigned)nt;
    return true;
}

int wgrad_splits(const Stem2Geom &g, int &tiles_per_split) {
    int64_t want = 4 * kCUs;                                     // ~2 rounds of 2 blocks per CU
    if (want > g.ntiles) want = g.ntiles;
    tiles_per_split = (int)((g.ntiles + want - 1) / want);
    return (int)((g.ntiles + tiles_per_split - 1) / tiles_per_split);
}

}  // namespace

// 1: cpg_conv2d_fwd / cpg_conv2d_fwd_bnstats / cpg_conv2d_wgrad run this layer on the strided stem kernels (CPG_NO_STEM: never)
extern "C" int cpg_conv_stem2_ok(const cpg_conv_desc *d) {
    Stem2Geom g;
    return stem2_geom(d, g) ? 1 : 0;
}
int cpg_conv_stem2_tiles(const cpg_conv_desc *d) {
    Stem2Geom g;
    return stem2_geom(d, g) ? (int)g.ntiles : 0;
}
int cpg_conv_stem2_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias, float *y,
                       float *stats, hipStream_t stream) {
    Stem2Geom g;
    if (!stem2_geom(d, g)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_fwd(strided stem): shape not supported");
    CPG_REQUIRE(x && w && y, "cpg_conv2d_fwd: null pointer");
    const unsigned blocks = (unsigned)std::min<int64_t>(((int64_t)g.ntiles * 2 + 3) / 4, 2 * kCUs);
    if (d->R == 7) {
        if (stats) hipLaunchKernelGGL((k_stem2_fwd<7, true>), dim3(blocks), dim3(256), 0, stream, g, x, w, pm, thr, bias, y, stats);
        else hipLaunchKernelGGL((k_stem2_fwd<7, false>), dim3(blocks), dim3(256), 0, stream, g, x, w, pm, thr, bias, y, nullptr);
    } else {
        if (stats) hipLaunchKernelGGL((k_stem2_fwd<3, true>), dim3(blocks), dim3(256), 0, stream, g, x, w, pm, thr, bias, y, stats);
        else hipLaunchKernelGGL((k_stem2_fwd<3, false>), dim3(blocks), dim3(256), 0, stream, g, x, w, pm, thr, bias, y, nullptr);
    }
    CPG_CHECK_LAUNCH("cpg_conv2d_fwd(strided stem)");
    return CPG_OK;
}
size_t cpg_conv_stem2_wgrad_workspace(const cpg_conv_desc *d) {
    Stem2Geom g;
    if (!stem2_geom(d, g)) return 0;
    int per;
    return (size_t)wgrad_splits(g, per) * d->K * d->C * d->R * d->S * sizeof(float);
}
int cpg_conv_stem2_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr, float *gw,
                         float *gpm, void *ws, size_t ws_bytes, hipStream_t stream) {
    Stem2Geom g;
    if (!stem2_geom(d, g)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_wgrad(strided stem): shape not supported");
    int per;
    const int nsplit = wgrad_splits(g, per);
    const int64_t out_elems = (int64_t)d->K * d->C * d->R * d->S;
    if (ws == nullptr || ws_bytes < (size_t)nsplit * out_elems * sizeof(float))
        return fail(CPG_E_WORKSPACE, "cpg_conv2d_wgrad(strided stem): workspace %zu < %zu bytes", ws_bytes, (size_t)nsplit * out_elems * sizeof(float));
    if (d->R == 7) hipLaunchKernelGGL(k_stem2_wgrad<7>, dim3((unsigned)nsplit), dim3(256), 0, stream, g, per, x, gy, (float *)ws);
    else hipLaunchKernelGGL(k_stem2_wgrad<3>, dim3((unsigned)nsplit), dim3(256), 0, stream, g, per, x, gy, (float *)ws);
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    launch_split_reduce((const float *)ws, nsplit, out_elems, 0, ep, stream);
    CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(strided stem)");
    return CPG_OK;
}
