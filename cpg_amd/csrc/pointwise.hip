// Masked 1x1 convolution (any stride, no padding) on fp32 MFMA: forward and input-gradient of
// models/layers.py:108-109 for the pointwise layers -- 36 of ResNet-50's 53 convs (33 1x1 s1 + 3
// 1x1 s2 downsample shortcuts, models/resnet.py), ~55 % of its FLOPs.
//
// A 1x1 conv over NCHW is a GEMM whose B operand is already k-major in memory:
//     D[m][g] = sum_c Wp[c][m] * X[n_g][c][pix_g],     g = flattened (image, output pixel)
// so no patch / halo is needed and the pixel tile can run ACROSS image boundaries: every tile is
// BN = 224 flattened pixels = 7 MFMA fragments, with zero tile waste for 56x56, 28x28, 14x14 and 7x7
// maps alike (a per-image tile would waste 12.5 % at 28x28 and 56 % at 7x7).
//
// Same loop discipline as conv3x3.hip's k_c3_fwd (see docs/LAB_NOTEBOOK.md section 4.1): packed K-major weights,
// two LDS stages with one barrier per chunk, operands of k-step s+1 read while the MFMAs of step s
// run, a branch-free chunk body with pinned instruction order in which the staging of later chunks
// rides between the MFMAs (registers holding chunk ch+1 are stored to the other LDS stage and
// refilled with chunk ch+2), activations fetched with range-checked buffer loads (out-of-range =>
// 0, no selects).  LDS row strides are = 32 (mod 64) floats so that the two half-waves (channel
// 2p / 2p+1 of a k-step) hit disjoint bank halves.
//
//   fwd  : reads x  [N][C][H][W]  at (oh*s, ow*s), writes y [N][K][OH][OW] densely
//   dgrad: reads gy [N][K][OH][OW] densely, writes gx [N][C][H][W] at (oh*s, ow*s)
//          (s > 1: the other positions of gx receive no gradient and are zeroed by a memset first)
//   wgrad: gW[k][c] = sum_g gy[k][g] * x[c][g] for dense (stride 1), 4 | pixels-per-image layers: k_pw_wgrad below;
//          strided / odd-sized ones stay on the generic split-K kernel of igemm_conv.hip.
#include <algorithm>
#include <type_traits>
#include "igemm_core.h"

using namespace cpg;

namespace {

struct PwGeom {
    int N, C, M;               // images; channels read; channels produced
    int Mp;                    // row stride of the packed weights (M rounded up to 128)
    int OW, HWo;               // pixel grid the GEMM runs over (the conv's OUTPUT grid in both passes)
    int in_plane, in_sy, in_sx;     // channel plane size of the tensor read, and its row / column pitch per grid step
    int out_plane, out_sy, out_sx;  // same for the tensor written
    int tiles_m;
    long long G;               // N * HWo (< 2^31 - 512: the kernels index the grid with 32-bit integers)
    float inv_hwo, inv_ow;     // 1 / HWo, 1 / OW (divmod_small)
    int flags;                 // kPwDenseIn: the tensor read is the grid itself (position = q); kPwDenseOut: same for the tensor written;
};                             // kPwSingle: one "image" (the plain GEMMs): no image index at all
enum { kPwDenseIn = 1, kPwDenseOut = 2, kPwSingle = 4 };
inline void pw_finish_geom(PwGeom &g) {
    g.inv_hwo = 1.0f / (float)g.HWo;
    g.inv_ow = 1.0f / (float)g.OW;
    g.flags = (g.N == 1 ? kPwSingle : 0) | ((g.in_sx == 1 && (g.in_sy == g.OW || g.N == 1 && g.in_sy == 0)) ? kPwDenseIn : 0) |
              ((g.out_sx == 1 && (g.out_sy == g.OW || g.N == 1 && g.out_sy == 0)) ? kPwDenseOut : 0);
}

// rel / d and rel % d for rel < 2^24 (exact in fp32) and a quotient of a few hundred at most: one multiply by the reciprocal and one
// correction step -- ~10 vector instructions where the compiler's 64-bit division is ~150.  On this chip every vector instruction of
// a wave costs its SIMD 4 cycles of fp32 MFMA issue (the two share the fp32 datapath: MFMA-busy + 4 x VALU instructions = 0.95-0.98 of
// the kernel's cycles in every counter pass, profiles/r03_pmc_pointwise.md), so the tile prologue / epilogue arithmetic is not free.
__device__ __forceinline__ void divmod_small(unsigned rel, unsigned d, float inv, unsigned &qt, unsigned &rm) {
    unsigned n = (unsigned)((float)rel * inv);
    int r = (int)(rel - n * d);
    if (r < 0) {
        n -= 1;
        r += (int)d;
    } else if (r >= (int)d) {
        n += 1;
        r -= (int)d;
    }
    qt = n;
    rm = (unsigned)r;
}

__device__ __forceinline__ f32x4 ld_sv4(const float *sbase, unsigned byte_off) {
    return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(sbase) + byte_off);
}

template <int BM_, int WM_, int WN_, int FN_, int CK_, bool VEC_, int MINW_>
struct PwCfg {
    static constexpr int BM = BM_, WM = WM_, WN = WN_, FN = FN_, CK = CK_, MINW = MINW_;
    static constexpr bool VEC = VEC_;                        // activations staged as float4 (dense, 4 | HWo)
    static_assert(WM * WN == 4 && BM % (32 * WM) == 0 && CK % 2 == 0, "bad pointwise config");
    static constexpr int FM = BM / 32 / WM;
    static constexpr int BN = 32 * FN * WN;
    static constexpr int LDW = BM + 32, LDX = (BN % 64 == 32) ? BN : BN + 32;      // = 32 (mod 64)
    static_assert(LDW % 64 == 32 && LDX % 64 == 32 && BM % 64 == 0, "row strides must be 32 mod 64");
    static constexpr int W4 = BM / 4, WROWS = 256 / W4, NW4 = (CK + WROWS - 1) / WROWS;
    static constexpr int XE = VEC ? CK * BN / 4 : CK * BN;   // staged activation items (float4 or float) per chunk
    static constexpr int NX = (XE + 255) / 256;
    // both LDS regions are padded to whole staging passes: every staging store is unconditional
    static constexpr int W_ELEMS = NW4 * WROWS * LDW;
    static constexpr int XS_ELEMS = CK * LDX + LDX;          // + one spare row: the idle threads of the last staging pass write there
    static constexpr int STAGE = W_ELEMS + XS_ELEMS;
    static constexpr int SMEM_FLOATS = 2 * STAGE;
    static constexpr int NS = CK / 2;
    static constexpr int NITEMS = NW4 + NX;
    static constexpr int PER_STEP = (NITEMS + NS - 1) / NS;  // staged items handled per k-step
};

// Wp[c][m] (row stride Mp, zero padded to whole CK x 128 blocks + WROWS slack rows):
//   fwd  : Wp[ci][co] = W[co][ci] * bin(pm)        dgrad: Wp[co][ci] = W[co][ci] * bin(pm)
__global__ __launch_bounds__(256) void k_pw_pack(const float *__restrict__ w, const float *__restrict__ pm, float thr,
                                                 float *__restrict__ out, int K, int C, int rows, int Mp, int dgrad) {
    const int64_t total = (int64_t)rows * Mp, nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += nthreads) {
        const int m = (int)(o % Mp), c = (int)(o / Mp);
        const int co = dgrad ? c : m, ci = dgrad ? m : c;
        float v = 0.0f;
        if (co < K && ci < C) {
            const int64_t off = (int64_t)co * C + ci;
            v = w[off];
            if (pm != nullptr) v *= binarize(pm[off], thr);
        }
        out[o] = v;
    }
}

// STATS (forward only): the block also writes, per output channel, the sum and the sum of squares of its 224 outputs ->
// stats[channel][pixel tile][2], the partial sums of the training-mode BatchNorm2d that follows every pointwise conv of the
// ResNet topologies (models/resnet.py:86-98) -- cpg_bn_stats_finalize merges them, no statistics pass over y (cf. k_c3_fwd).
// ADD (input-gradient launches): y = result + addend, elementwise at the stored positions -- the gradient of the OTHER consumer of the
// conv's input (a residual block's identity branch, models/resnet.py:84,104: `out += identity`), which autograd would otherwise add
// in a separate 3-pass kernel.
// MASK (round 5; the plain-GEMM form: the weight gradient of a masked linear layer WITH a piggymask, models/layers.py:190 -- the autograd of
// bin(pm) * W): the accumulator tile is g = gW_eff; the epilogue reads pm and W at the output's own positions and stores
// y = gW = g * bin(pm) and mk.gpm = g * W -- both gradients from one tile, W and pm read once.  Until round 5 these launches ran on the
// generic k_gemm (0.30 of the MFMA peak on features.45, a 16-step contraction in front of a 1.6 GB epilogue).
struct PwMask {
    const float *pm, *w;
    float *gpm;
    float thr;
};
// MASKX (round 5; plain-GEMM form: the INPUT gradient of a masked linear layer with a piggymask, gx = gy . (W * bin(pm))): the K-major operand
// X is the weight itself; its piggymask (mk.pm, laid out like X) is fetched beside it and the binarised product goes to LDS -- W and pm are
// read once, nothing is materialised.
template <class Cfg, bool DGRAD, bool STATS = false, bool ADD = false, bool MASK = false, bool MASKX = false>
__global__ __launch_bounds__(256, Cfg::MINW) void k_pw(PwGeom g, const float *__restrict__ x, const float *__restrict__ wp,
                                                       const float *__restrict__ bias, float *__restrict__ y,
                                                       float *__restrict__ stats = nullptr, const float *__restrict__ addend = nullptr,
                                                       PwMask mk = PwMask{nullptr, nullptr, nullptr, 0.0f}) {
    static_assert(!ADD || (DGRAD && !STATS), "the addend rides in plain input-gradient launches");
    static_assert(!STATS || !DGRAD, "statistics ride in forward launches");
    static_assert(!MASK || (!DGRAD && !STATS && !ADD), "the piggymask epilogue rides in plain forward-form launches");
    static_assert(!MASKX || (Cfg::VEC && !DGRAD && !STATS && !ADD && !MASK), "the masked operand: float4 staging, plain forward-form launches");
    __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int li = lane & 31, lh = lane >> 5;

    unsigned lb = xcd_remap(blockIdx.x, gridDim.x);      // m tile fastest: blocks sharing a pixel tile share an L2
    // (the integer divisions of uniform values are expanded into vector code: pin the results to scalar registers, or everything
    // derived from them -- descriptors, channel tests -- is computed per lane)
    const int tm = __builtin_amdgcn_readfirstlane((int)(lb % (unsigned)g.tiles_m));
    const unsigned g0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((lb / (unsigned)g.tiles_m) * Cfg::BN));   // first grid position of the tile (G < 2^31)
    const unsigned Gu = (unsigned)g.G, HWo = (unsigned)g.HWo;
    const int m0 = tm * Cfg::BM;
    const bool single = g.flags & kPwSingle;
    const int n_first = single ? 0 : __builtin_amdgcn_readfirstlane((int)(g0 / HWo));
    const unsigned rel0 = g0 - (unsigned)n_first * HWo;  // the tile's first position within image n_first
    // grid position rel0 + j -> (image - n_first, position in the tensor's plane given its row / column pitch)
    auto locate = [&](unsigned j, bool dense, int sy, int sx, unsigned &n_rel, unsigned &pos) {
        unsigned q = rel0 + j;
        n_rel = 0;
        if (!single) divmod_small(q, HWo, g.inv_hwo, n_rel, q);
        pos = q;
        if (!dense) {
            unsigned row, col;
            divmod_small(q, (unsigned)g.OW, g.inv_ow, row, col);
            pos = row * (unsigned)sy + col * (unsigned)sx;
        }
    };

    // ---- staging descriptors (fixed for the life of the block) ----
    const int wcol = (tid % Cfg::W4) * 4, wrow0 = tid / Cfg::W4;
    const int wdst = wrow0 * Cfg::LDW + wcol;
    const unsigned wbyte = (unsigned)(wrow0 * g.Mp + m0 + wcol) * 4u;
    // activations: byte offset from image n_first's channel 0 (of the current chunk), or 0x80000000 for grid
    // positions past the last image -- the buffer unit's range check returns 0 for those
    constexpr int kOutOfRange = (int)0x80000000;
    int xbyte[Cfg::NX], xdst[Cfg::NX];
    const bool dense_in = g.flags & kPwDenseIn;
#pragma unroll
    for (int i = 0; i < Cfg::NX; ++i) {
        const int e = tid + 256 * i;
        constexpr int PER_ROW = Cfg::VEC ? Cfg::BN / 4 : Cfg::BN;
        const int cl = e / PER_ROW, j = (e - cl * PER_ROW) * (Cfg::VEC ? 4 : 1);
        const bool ok = e < Cfg::XE && g0 + (unsigned)j < Gu;
        unsigned n_rel, pos;
        locate((unsigned)j, dense_in, g.in_sy, g.in_sx, n_rel, pos);
        xbyte[i] = ok ? (int)((((n_rel * (unsigned)g.C + (unsigned)cl) * (unsigned)g.in_plane) + pos) * 4u) : kOutOfRange;
        xdst[i] = e < Cfg::XE ? cl * Cfg::LDX + j : Cfg::CK * Cfg::LDX;            // spare row for the idle threads of the last pass
    }
    const long long remaining = (long long)(g.N - n_first) * g.C * g.in_plane * 4;
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(x + (long long)n_first * g.C * g.in_plane), 0, (int)std::min<long long>(remaining, 0x7FFFFFFFll), 0x00020000);

    // packed weights: (C rounded up to 16, + 16 slack rows) x Mp floats (pack_bytes)
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, (int)std::min<long long>(((long long)g.C + 32) * g.Mp * 4, 0x7FFFFFFFll), 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_pm = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(MASKX ? mk.pm + (long long)n_first * g.C * g.in_plane : x), 0, MASKX ? (int)std::min<long long>(remaining, 0x7FFFFFFFll) : 0, 0x00020000);
    f32x4 rw[Cfg::NW4];
    f32x4 rxv[Cfg::VEC ? Cfg::NX : 1];
    f32x4 rpm[MASKX ? Cfg::NX : 1];
    float rxs[Cfg::VEC ? 1 : Cfg::NX];
    auto load_item = [&](int k, int c0) {
        // (the chunk's share of every address is uniform: it rides in the loads' scalar offset, no vector add per load)
        if (k < Cfg::NW4) {
            rw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_w, (int)wbyte, (c0 + Cfg::WROWS * k) * g.Mp * 4, 0));
        } else if (Cfg::VEC) {
            rxv[k - Cfg::NW4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, xbyte[k - Cfg::NW4], c0 * g.in_plane * 4, 0));
            if constexpr (MASKX)
                rpm[k - Cfg::NW4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_pm, xbyte[k - Cfg::NW4], c0 * g.in_plane * 4, 0));
        } else {
            rxs[k - Cfg::NW4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, xbyte[k - Cfg::NW4], c0 * g.in_plane * 4, 0));
        }
    };
    auto store_item = [&](int k, float *stage) {
        if (k < Cfg::NW4)
            *reinterpret_cast<f32x4 *>(stage + wdst + Cfg::WROWS * k * Cfg::LDW) = rw[k];
        else if (Cfg::VEC) {
            f32x4 v = rxv[k - Cfg::NW4];
            if constexpr (MASKX) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] *= binarize(rpm[k - Cfg::NW4][q], mk.thr);
            }
            *reinterpret_cast<f32x4 *>(stage + Cfg::W_ELEMS + xdst[k - Cfg::NW4]) = v;
        }
        else
            stage[Cfg::W_ELEMS + xdst[k - Cfg::NW4]] = rxs[k - Cfg::NW4];
    };

    const int a_base = lh * Cfg::LDW + wm * Cfg::FM * 32 + li;
    const int b_base = Cfg::W_ELEMS + lh * Cfg::LDX + wn * Cfg::FN * 32 + li;

    f32x16 acc[Cfg::FM][Cfg::FN];
#pragma unroll
    for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[fm][fn][e] = 0.0f;

    const int nch = g.C / Cfg::CK;                 // the host guarantees C % CK == 0
#pragma unroll
    for (int k = 0; k < Cfg::NITEMS; ++k) load_item(k, 0);
#pragma unroll
    for (int k = 0; k < Cfg::NITEMS; ++k) store_item(k, smem);
#pragma unroll
    for (int k = 0; k < Cfg::NITEMS; ++k) load_item(k, min(1, nch - 1) * Cfg::CK);
    __syncthreads();
    // (two copies of the chunk body, one per LDS stage: with the stage a compile-time constant every LDS address is a per-lane base
    //  + an immediate -- selecting the stage at run time cost 19 vector address instructions per chunk)
    auto chunk = [&](int ch, auto stage_c) {
        constexpr int kPar = decltype(stage_c)::value;
        const float *cur = smem + kPar * Cfg::STAGE;
        float *other = smem + (kPar ^ 1) * Cfg::STAGE;
        const int c_next2 = min(ch + 2, nch - 1) * Cfg::CK;        // clamped: the tail re-stages data nobody reads
        float a[2][Cfg::FM], b[2][Cfg::FN];
        auto lds_operands = [&](int st, int set) {
#pragma unroll
            for (int fm = 0; fm < Cfg::FM; ++fm) a[set][fm] = cur[a_base + 2 * st * Cfg::LDW + fm * 32];
#pragma unroll
            for (int fn = 0; fn < Cfg::FN; ++fn) b[set][fn] = cur[b_base + 2 * st * Cfg::LDX + fn * 32];
        };
        lds_operands(0, 0);
#pragma unroll
        for (int st = 0; st < Cfg::NS; ++st) {
            if (st + 1 < Cfg::NS) lds_operands(st + 1, (st + 1) & 1);
#pragma unroll
            for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < Cfg::FN; ++fn)
                    acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][fm], b[st & 1][fn], acc[fm][fn], 0, 0, 0);
            // this step's share of the staging: registers hold chunk ch+1 -> other stage, then refill with chunk ch+2
#pragma unroll
            for (int k = st * Cfg::PER_STEP; k < (st + 1) * Cfg::PER_STEP && k < Cfg::NITEMS; ++k) {
                store_item(k, other);
                load_item(k, c_next2);
            }
#pragma unroll
            for (int i = 0; i < Cfg::FM * Cfg::FN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (i >= 1 && i <= Cfg::PER_STEP) {
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
        __syncthreads();
    };
    for (int ch = 0; ch < nch; ch += 2) {
        chunk(ch, std::integral_constant<int, 0>{});
        if (ch + 1 < nch) chunk(ch + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: D col = grid position (lane & 31), D row = channel ----
    // Addressing is 32-bit and mostly scalar: per fragment column a lane computes ONE byte offset (its grid position in the output
    // tensor, + 4 lh channel planes); the channel of accumulator element e is uniform -- it goes into the buffer descriptor's base
    // (one descriptor per group of 8 channels, built with scalar arithmetic) and the store's scalar offset.  Invalid grid positions
    // carry the out-of-range offset, whole invalid channel groups are skipped by a scalar branch (the hosts only send channel
    // counts that are multiples of 8 here: 16 for the convs, cpg_pw_gemm_nn_ok for the plain GEMMs).
    float s1[STATS ? Cfg::FM : 1][16], s2[STATS ? Cfg::FM : 1][16];
    if (STATS) {
#pragma unroll
        for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
            for (int e = 0; e < 16; ++e) s1[fm][e] = s2[fm][e] = 0.0f;
    }
    const bool dense_out = g.flags & kPwDenseOut;
    const long long img_base = (long long)n_first * g.M * g.out_plane;             // (elements) image n_first, channel 0
    const long long y_total = (long long)g.N * g.M * g.out_plane;
    const unsigned plane4 = (unsigned)g.out_plane * 4u;
    auto write_out = [&](auto has_bias) {        // two copies of the loop, picked by ONE uniform branch (the ResNet convs have no bias)
#pragma unroll
    for (int fn = 0; fn < Cfg::FN; ++fn) {
        const unsigned pix = (unsigned)((wn * Cfg::FN + fn) * 32 + li);
        const bool pok = g0 + pix < Gu;
        unsigned n_rel, pos;
        locate(pix, dense_out, g.out_sy, g.out_sx, n_rel, pos);
        // byte offset of (image n_first + n_rel, channel 4 lh, position) from (image n_first, channel 0)
        const int voff = pok ? (int)((((n_rel * (unsigned)g.M + 4u * (unsigned)lh) * (unsigned)g.out_plane) + pos) * 4u) : kOutOfRange;
#pragma unroll
        for (int fm = 0; fm < Cfg::FM; ++fm) {
            const int rowf = m0 + (wm * Cfg::FM + fm) * 32;                         // channels rowf + 8 eg + (e & 3) + 4 lh
            auto group_srd = [&](const float *base, int eg) {
                const long long gbase = img_base + (long long)(rowf + 8 * eg) * g.out_plane;
                const int records = (int)std::max<long long>(0, std::min<long long>((y_total - gbase) * 4, 0x7FFFFFFFll));
                return __builtin_amdgcn_make_buffer_rsrc((void *)(base + gbase), 0, records, 0x00020000);
            };
            // (Round 5 tried requesting the NEXT fragment's addend before this fragment's stores -- two register sets, loads one fragment ahead:
            //  no change, 217.6 vs 218.1 ms over ResNet-50's input-gradient family.  The launch is bound by bytes, not by a late load:
            //  256 <- 64 channels @56 x 56 moves gy + addend + gx = 1.85 GB (14 flops per byte) in 0.447 ms -- the 0.29 ms those bytes take at
            //  6.3 TB/s plus the 0.17 ms of its MFMA work; profiles/r05_pmc_wait_resnet50.md.)
            float av[16];
            if (ADD) {                                                              // the fragment's 16 addend loads go out together
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[4 * eg + r] = 0.0f;
                    if (rowf + 8 * eg >= g.M) continue;                             // (uniform) a group past the last channel has no descriptor
                    const __amdgpu_buffer_rsrc_t srd_a = group_srd(addend, eg);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        av[4 * eg + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_a, voff, r * plane4, 0));
                }
            }
            if constexpr (MASK) {
                // the piggymask and the weight of the fragment's 16 outputs: 32 loads out together, then two stores per output
                float pv[16], wv[16];
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[4 * eg + r] = wv[4 * eg + r] = 0.0f;
                    if (rowf + 8 * eg >= g.M) continue;
                    const __amdgpu_buffer_rsrc_t srd_p = group_srd(mk.pm, eg), srd_w = group_srd(mk.w, eg);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pv[4 * eg + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_p, voff, r * plane4, 0));
                        wv[4 * eg + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_w, voff, r * plane4, 0));
                    }
                }
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {
                    if (rowf + 8 * eg >= g.M) continue;
                    const __amdgpu_buffer_rsrc_t srd_y = group_srd(y, eg), srd_g = group_srd(mk.gpm, eg);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = 4 * eg + r;
                        const float v = acc[fm][fn][e];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v * binarize(pv[e], mk.thr)), srd_y, voff, r * plane4, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v * wv[e]), srd_g, voff, r * plane4, 0);
                    }
                }
                continue;
            }
#pragma unroll
            for (int eg = 0; eg < 4; ++eg) {
                const int row0 = rowf + 8 * eg;
                if (row0 >= g.M) continue;                                          // (uniform; M is a multiple of 8: whole groups)
                const __amdgpu_buffer_rsrc_t srd_y = group_srd(y, eg);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = 4 * eg + r;
                    float v = acc[fm][fn][e];
                    if (ADD) v += av[e];
                    if (decltype(has_bias)::value) {
                        const int co = row0 + r + 4 * lh;
                        v += bias[co < g.M ? co : 0];
                        if (STATS && !pok) v = 0.0f;                               // (without a bias a padded position is an exact 0)
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), srd_y, voff, r * plane4, 0);
                    if (STATS) {
                        s1[fm][e] += v;
                        s2[fm][e] = fmaf(v, v, s2[fm][e]);
                    }
                }
            }
        }
    }
    };
    if (bias != nullptr) write_out(std::true_type{});
    else write_out(std::false_type{});
    if (STATS) {
        // sum over the 32 pixel lanes of each half-wave (DPP adds; valid in lanes 16-31 / 48-63)
        const unsigned ntiles = gridDim.x / g.tiles_m, tile_n = xcd_remap(blockIdx.x, gridDim.x) / g.tiles_m;
        float *red = smem;                               // WN > 1: [WN][BM][2] (the main loop's last barrier freed the LDS)
#pragma unroll
        for (int fm = 0; fm < Cfg::FM; ++fm) {
            half_wave_sum8(&s1[fm][0]);
            half_wave_sum8(&s1[fm][8]);
            half_wave_sum8(&s2[fm][0]);
            half_wave_sum8(&s2[fm][8]);
            if (li == 31) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ch = (wm * Cfg::FM + fm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (Cfg::WN == 1) {                  // the wave owns its channel rows
                        if (m0 + ch < g.M) {
                            float *dst = stats + ((int64_t)(m0 + ch) * ntiles + tile_n) * 2;
                            dst[0] = s1[fm][e];
                            dst[1] = s2[fm][e];
                        }
                    } else {
                        red[(wn * Cfg::BM + ch) * 2 + 0] = s1[fm][e];
                        red[(wn * Cfg::BM + ch) * 2 + 1] = s2[fm][e];
                    }
                }
            }
        }
        if (Cfg::WN > 1) {                               // fixed-order merge of the WN waves that share the channels
            __syncthreads();
            if (tid < Cfg::BM && m0 + tid < g.M) {
                float a = 0.0f, b = 0.0f;
#pragma unroll
                for (int w2 = 0; w2 < Cfg::WN; ++w2) {
                    a += red[(w2 * Cfg::BM + tid) * 2 + 0];
                    b += red[(w2 * Cfg::BM + tid) * 2 + 1];
                }
                float *dst = stats + ((int64_t)(m0 + tid) * ntiles + tile_n) * 2;
                dst[0] = a;
                dst[1] = b;
            }
        }
    }
}

// ------------------------------------------------------------------------------ weight gradient
// D[co][ci] += sum over flattened pixels g = (image, pixel) of gy[n][co][q] * x[n][ci][q]; split-K over ranges of
// 32-pixel units.  Both operands are pixel-contiguous in memory, so the LDS tiles are [channel][34]: a row stride of
// 34 floats sends lane li (= channel) to bank 2*li and the other half-wave (pixel + 1) to the odd banks -- conflict-free
// operand reads -- and keeps rows 8-byte aligned, so a float4 of 4 consecutive pixels is staged with one
// buffer_load_dwordx4 and two ds_write_b64.  Two LDS stages, one barrier per unit; the 8 loads + 16 LDS stores of
// unit u+2 / u+1 ride between the 64 MFMAs of unit u (conv3x3.hip's k_c3_wgrad scheme).  Block = BCO x BCI
// channels, 2 x 2 waves, each wave (BCO/2) x (BCI/2).
// WCO_ x (4 / WCO_) waves: 2 x 2 for the conv layers; 1 x 4 with BCO = 32 for the plain-GEMM form with <= 32 rows (the forward of a linear
// layer at <= 32 images per GPU, the reference's own 256 / 8 split: a 128-row tile spent 3/4 of its MFMAs on rows that do not exist).
template <int BCO_, int BCI_, int WCO_ = 2>
struct PwWCfg {
    static constexpr int BCO = BCO_, BCI = BCI_, PIX = 32, LD = PIX + 2;
    static constexpr int WCO = WCO_, WCI = 4 / WCO_;
    static constexpr int FM = BCO / (32 * WCO), FN = BCI / (32 * WCI);
    static_assert(BCO % (32 * WCO) == 0 && BCI % (32 * WCI) == 0 && WCO * WCI == 4, "wave tiles are multiples of 32 x 32");
    static constexpr int NA = BCO * (PIX / 4) / 256, NB = BCI * (PIX / 4) / 256;     // float4 per thread per unit
    static constexpr int A_ELEMS = BCO * LD, STAGE = (BCO + BCI) * LD;
    static constexpr int NITEMS = NA + NB, NS = PIX / 2;
};

// SCALAR: every staged pixel is fetched on its own (four dword loads instead of one dwordx4 per item) -- planes whose size is not
// a multiple of 4 (ResNet layer4's 7 x 7 maps: a float4 would straddle images and lose its alignment) and STRIDED layers (the 1x1 s2
// downsample shortcuts, models/resnet.py:189-193), where pixel q of the output grid pairs with x at (s q / OW, s (q % OW)).
// xg = {x plane size, OW, row pitch, column pitch} of the tensor read as x (SCALAR only).
struct PwWX {
    int plane, OW, sy, sx;
};
// MASKB (round 5; the plain-GEMM form = the FORWARD of a masked linear layer with a piggymask, y = x . (W * bin(pm))^T): the operand read as
// "x" is the weight; its piggymask (pmb, laid out like it) is fetched beside it and the binarised product goes to LDS.
template <class Cfg, bool SCALAR = false, bool MASKB = false>
__global__ __launch_bounds__(256, 2) void k_pw_wgrad(int M, int C, int HWo, long long G, int tiles_co, int tiles_ci,
                                                     int units_per_split, const float *__restrict__ x,
                                                     const float *__restrict__ gy, float *__restrict__ part, PwWX xg = PwWX{0, 0, 0, 0},
                                                     const float *__restrict__ pmb = nullptr, float thr = 0.0f) {
    static_assert(!MASKB || !SCALAR, "the masked operand: float4 staging");
    __shared__ __attribute__((aligned(16))) float smem[2 * Cfg::STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = sub / Cfg::WCI, wci = sub % Cfg::WCI;
    const int li = lane & 31, lh = lane >> 5;
    // all (co, ci) tiles of one split on one XCD (block b runs on XCD b % 8): the split's units are read once per L2
    const int tiles = tiles_co * tiles_ci;
    const int xcd = blockIdx.x % kXCDs, jb = blockIdx.x / kXCDs;
    const int split = (jb / tiles) * kXCDs + xcd, tile = jb % tiles;
    const int co0 = (tile / tiles_ci) * Cfg::BCO, ci0 = (tile % tiles_ci) * Cfg::BCI;
    const long long total_units = (G + Cfg::PIX - 1) / Cfg::PIX;
    const long long u0 = std::min<long long>(total_units, (long long)split * units_per_split);
    const long long u1 = std::min<long long>(total_units, u0 + units_per_split);

    f32x16 acc[Cfg::FM][Cfg::FN];
#pragma unroll
    for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[fm][fn][e] = 0.0f;

    // staged float4 e = tid + 256*i: channel e / 8 of the tile, pixels 4*(e % 8) .. +3 of the unit.  Channels past the
    // tensor are clamped (their accumulator rows / columns are never stored).
    const int j4 = tid & 7, chl = tid >> 3;             // 32 channels per pass of the block
    constexpr int kOutOfRange = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t srd_g = __builtin_amdgcn_make_buffer_rsrc((void *)gy, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_pb = __builtin_amdgcn_make_buffer_rsrc((void *)(MASKB ? pmb : x), 0, MASKB ? 0x7FFFFFFF : 0, 0x00020000);
    int a_chan[Cfg::NA], b_chan[Cfg::NB];               // byte offset of the channel plane
#pragma unroll
    for (int i = 0; i < Cfg::NA; ++i) a_chan[i] = min(co0 + chl + 32 * i, M - 1) * HWo * 4;
#pragma unroll
    for (int i = 0; i < Cfg::NB; ++i) b_chan[i] = min(ci0 + chl + 32 * i, C - 1) * (SCALAR ? xg.plane : HWo) * 4;
    const int dst = chl * Cfg::LD + 4 * j4;
    int pos_g = kOutOfRange, pos_x = kOutOfRange;       // byte offset of (image, pixel) in gy / x for the unit being loaded
    int pgs[SCALAR ? 4 : 1], pxs[SCALAR ? 4 : 1];       // SCALAR: one position per pixel of the float4
    // Units are described in order u0, u0 + 1, ... (clamped at the split's last unit), so the dense variant keeps (pixel within the
    // image, byte offsets) per lane and ADVANCES them by one unit -- a dozen vector instructions where the division by the plane
    // size and the two multiplications were ~30 (vector instructions are MFMA time on this chip, see k_pw).
    int d_q = 0, d_g = 0;                               // dense: pixel within its image / flattened pixel of the lane's float4
    int d_pg = 0, d_px = 0;                             // dense: byte offsets of that pixel in gy / x (valid or not)
    long long d_u = -1;                                 // dense: the unit those describe
    const int wrap_g = (M - 1) * HWo * 4, wrap_x = (C - 1) * HWo * 4;     // extra bytes when a lane's pixel moves on to the next image
    auto describe = [&](long long u) {      // G < 2^29 (host check): 32-bit arithmetic
        if (!SCALAR) {
            if (d_u < 0) {                                       // first unit of the split: one division
                d_g = (int)u * Cfg::PIX + 4 * j4;
                const int n = d_g / HWo;
                d_q = d_g - n * HWo;
                d_pg = (n * M * HWo + d_q) * 4, d_px = (n * C * HWo + d_q) * 4;
            } else if (u != d_u) {                               // (uniform) the next unit; u == d_u: clamped at the last one
                d_g += Cfg::PIX, d_q += Cfg::PIX, d_pg += Cfg::PIX * 4, d_px += Cfg::PIX * 4;
                while (d_q >= HWo) d_q -= HWo, d_pg += wrap_g, d_px += wrap_x;    // (planes of >= 32 pixels: at most once)
            }
            d_u = u;
            const bool ok = d_g < (int)G;
            pos_g = ok ? d_pg : kOutOfRange;
            pos_x = ok ? d_px : kOutOfRange;
            return;
        }
        const int g = (int)u * Cfg::PIX + 4 * j4;
        int n = g / HWo, q = g - n * HWo;
        {
            int oy = q / xg.OW, ox = q - oy * xg.OW;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = g + e < (int)G;
                pgs[e] = ok ? (n * M * HWo + q) * 4 : kOutOfRange;
                pxs[e] = ok ? (n * C * xg.plane + oy * xg.sy + ox * xg.sx) * 4 : kOutOfRange;
                ++q, ++ox;                                       // next pixel: carry into the row / the image
                if (ox == xg.OW) ox = 0, ++oy;
                if (q == HWo) q = 0, oy = 0, ++n;
            }
        }
    };
    f32x4 st[Cfg::NITEMS];
    f32x4 stm[MASKB ? Cfg::NB : 1];            // (MASKB) the piggymask of the staged weight rows
    auto load_item = [&](int k) {
        if (SCALAR) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                st[k][e] = __builtin_bit_cast(float, k < Cfg::NA ? __builtin_amdgcn_raw_buffer_load_b32(srd_g, pgs[e] + a_chan[k], 0, 0)
                                                                 : __builtin_amdgcn_raw_buffer_load_b32(srd_x, pxs[e] + b_chan[k - Cfg::NA], 0, 0));
            return;
        }
        if (k < Cfg::NA)
            st[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_g, pos_g + a_chan[k], 0, 0));
        else {
            st[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, pos_x + b_chan[k - Cfg::NA], 0, 0));
            if constexpr (MASKB)
                stm[k - Cfg::NA] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_pb, pos_x + b_chan[k - Cfg::NA], 0, 0));
        }
    };
    auto store_item = [&](int k, float *stage) {
        float *p = stage + (k < Cfg::NA ? 32 * k * Cfg::LD : Cfg::A_ELEMS + 32 * (k - Cfg::NA) * Cfg::LD) + dst;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x4 v = st[k];
        if constexpr (MASKB) {
            if (k >= Cfg::NA) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] *= binarize(stm[k - Cfg::NA][q], thr);
            }
        }
        *reinterpret_cast<f32x2 *>(p) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2 *>(p + 2) = f32x2{v[2], v[3]};
    };
    const int a_base = (wco * (Cfg::BCO / Cfg::WCO) + li) * Cfg::LD + lh;
    const int b_base = Cfg::A_ELEMS + (wci * (Cfg::BCI / Cfg::WCI) + li) * Cfg::LD + lh;

    if (u0 < u1) {
        describe(u0);
#pragma unroll
        for (int k = 0; k < Cfg::NITEMS; ++k) load_item(k);
#pragma unroll
        for (int k = 0; k < Cfg::NITEMS; ++k) store_item(k, smem);
        describe(std::min(u0 + 1, u1 - 1));
#pragma unroll
        for (int k = 0; k < Cfg::NITEMS; ++k) load_item(k);
        __syncthreads();
        auto unit = [&](long long u, auto stage_c) {             // (one copy per LDS stage: immediate LDS offsets, as in k_pw)
            constexpr int cur_i = decltype(stage_c)::value;
            const float *cur = smem + cur_i * Cfg::STAGE;
            float *other = smem + (cur_i ^ 1) * Cfg::STAGE;
            describe(std::min(u + 2, u1 - 1));          // clamped: the tail re-stages data nobody reads
            float a[2][Cfg::FM], b[2][Cfg::FN];
            auto rd = [&](int s, int set) {
#pragma unroll
                for (int fm = 0; fm < Cfg::FM; ++fm) a[set][fm] = cur[a_base + fm * 32 * Cfg::LD + 2 * s];
#pragma unroll
                for (int fn = 0; fn < Cfg::FN; ++fn) b[set][fn] = cur[b_base + fn * 32 * Cfg::LD + 2 * s];
            };
            rd(0, 0);
#pragma unroll
            for (int s = 0; s < Cfg::NS; ++s) {
                if (s + 1 < Cfg::NS) rd(s + 1, (s + 1) & 1);
#pragma unroll
                for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
                    for (int fn = 0; fn < Cfg::FN; ++fn)
                        acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][fm], b[s & 1][fn], acc[fm][fn], 0, 0, 0);
                const int k0 = s * Cfg::NITEMS / Cfg::NS, k1 = (s + 1) * Cfg::NITEMS / Cfg::NS;
#pragma unroll
                for (int k = k0; k < k1; ++k) {
                    store_item(k, other);
                    load_item(k);
                }
#pragma unroll
                for (int i = 0; i < Cfg::FM * Cfg::FN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (i < k1 - k0) {
                        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, SCALAR ? 4 : 1, 0);
                    }
                }
            }
            __syncthreads();
        };
        for (long long u = u0; u < u1; u += 2) {
            unit(u, std::integral_constant<int, 0>{});
            if (u + 1 < u1) unit(u + 1, std::integral_constant<int, 1>{});
        }
    }
    // partial result part[split][co][ci]: lanes 0-31 of a store cover 32 consecutive ci
    float *dstp = part + (int64_t)split * M * C;
#pragma unroll
    for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wco * (Cfg::BCO / Cfg::WCO) + fm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int ci = ci0 + wci * (Cfg::BCI / Cfg::WCI) + fn * 32 + li;
                if (co < M && ci < C) dstp[(int64_t)co * C + ci] = acc[fm][fn][e];
            }
}

struct PwWPlan {
    int tiles_co, tiles_ci, nsplit, units_per_split;
    size_t ws_bytes;
};
template <class Cfg>
PwWPlan pw_wgrad_plan(const cpg_conv_desc *d) {
    PwWPlan p;
    p.tiles_co = (d->K + Cfg::BCO - 1) / Cfg::BCO;
    p.tiles_ci = (d->C + Cfg::BCI - 1) / Cfg::BCI;
    const int OH = (d->H - 1) / d->stride_h + 1, OW = (d->W - 1) / d->stride_w + 1;       // the gy grid
    const int64_t units = ((int64_t)d->N * OH * OW + Cfg::PIX - 1) / Cfg::PIX;
    const int64_t tiles = (int64_t)p.tiles_co * p.tiles_ci;
    // split blocks per CU: every split writes a full set of partial sums that k_split_reduce reads back; 2 (one round of the two
    // resident blocks) instead of round 2's 4: ResNet-50 73.83 -> 72.98 ms per step (A/B through CPG_PWW_BPC)
    // (the shared-chip hint no longer changes this: see conv3x3.hip's weight-gradient planner)
    const int bpc = std::max(1, opt_or(OPT_PWW_BPC, 2));
    int64_t want = ((int64_t)bpc * kCUs + tiles - 1) / tiles;
    want = std::max<int64_t>(1, std::min<int64_t>(want, (units + 7) / 8));      // at least 8 units per split
    want = (want + kXCDs - 1) / kXCDs * kXCDs;
    p.units_per_split = (int)((units + want - 1) / want);
    p.nsplit = (int)want;                                       // trailing splits may be empty: they write zeros
    p.ws_bytes = (size_t)p.nsplit * d->K * d->C * sizeof(float);
    return p;
}
using PwW128 = PwWCfg<128, 128>;
using PwW64o = PwWCfg<64, 128>;      // <= 64 output channels
using PwW64i = PwWCfg<128, 64>;      // <= 64 input channels
using PwW64 = PwWCfg<64, 64>;
inline int pw_wgrad_pick(const cpg_conv_desc *d) { return (d->K <= 64 ? 1 : 0) + (d->C <= 64 ? 2 : 0); }

template <class Cfg>
int pw_wgrad_launch(const cpg_conv_desc *d, const float *x, const float *gy, const Epilogue &ep, void *ws, size_t ws_bytes,
                    hipStream_t stream) {
    const PwWPlan p = pw_wgrad_plan<Cfg>(d);
    if (ws == nullptr || ws_bytes < p.ws_bytes)
        return fail(CPG_E_WORKSPACE, "cpg_conv2d_wgrad(1x1): workspace %zu < %zu bytes", ws_bytes, p.ws_bytes);
    const int OH = (d->H - 1) / d->stride_h + 1, OW = (d->W - 1) / d->stride_w + 1;
    const bool vec = d->stride_h == 1 && d->stride_w == 1 && (OH * OW) % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)gy) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL((k_pw_wgrad<Cfg, false>), dim3((unsigned)(p.tiles_co * p.tiles_ci * p.nsplit)), dim3(256), 0, stream, d->K, d->C,
                           OH * OW, (long long)d->N * OH * OW, p.tiles_co, p.tiles_ci, p.units_per_split, x, gy, (float *)ws, PwWX{0, 0, 0, 0});
    else
        hipLaunchKernelGGL((k_pw_wgrad<Cfg, true>), dim3((unsigned)(p.tiles_co * p.tiles_ci * p.nsplit)), dim3(256), 0, stream, d->K, d->C,
                           OH * OW, (long long)d->N * OH * OW, p.tiles_co, p.tiles_ci, p.units_per_split, x, gy, (float *)ws,
                           PwWX{d->H * d->W, OW, d->stride_h * d->W, d->stride_w});
    launch_split_reduce((const float *)ws, p.nsplit, (int64_t)d->K * d->C, 0, ep, stream);
    CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(1x1)");
    return CPG_OK;
}

//                BM  WM WN FN CK  VEC  MINW
#ifndef PW_MINW
#define PW_MINW 3
#endif
#ifndef PW_CK
#define PW_CK 16
#endif
#ifndef PW_TILE_DEFAULT
#define PW_TILE_DEFAULT 0
#endif
using PwV = PwCfg<128, 4, 1, 7, PW_CK, true, PW_MINW>;       // dense reads, 4 | pixels per image: float4 staging
using PwS = PwCfg<128, 4, 1, 7, 16, false, 2>;      // strided reads (1x1 s2 forward) or odd plane sizes (7x7 maps)
// round 4: 128 rows x 256 flattened pixels as 2 x 2 waves of 2 x 4 fragments -- 6 LDS operand reads per 8 MFMAs where the 4 x 1 / 7-fragment
// tile has 8 per 7 (the four waves of that tile all read the same seven B operands)
using PwV2 = PwCfg<128, 2, 2, 4, PW_CK, true, 2>;
inline bool pw_wide() { return cpg::opt_or(cpg::OPT_PW_TILE, PW_TILE_DEFAULT) != 0; }
// <= 64 channels produced (ResNet layer1: conv1 forward, conv3 input gradient): a 128-row tile would run half of its MFMAs on
// rows that do not exist.  64 rows x 256 flattened pixels (2 x 2 waves, 4 fragments each); 56 x 56 maps x any batch divide by 256.
using PwV64 = PwCfg<64, 2, 2, 4, 16, true, 3>;
using PwS64 = PwCfg<64, 2, 2, 4, 16, false, 3>;

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }
inline size_t pack_bytes(int c_read, int m) { return ((size_t)pad_to(c_read, 16) + 16) * pad_to(m, 128) * sizeof(float); }

template <class Cfg, bool DGRAD>
int launch(PwGeom g, const float *x, const float *wp, const float *bias, float *y, hipStream_t stream, const char *what,
           float *stats = nullptr, const float *addend = nullptr, const PwMask *mask = nullptr) {
    g.tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM;
    pw_finish_geom(g);
    if (g.G >= (1ll << 31) - 512 || (g.N > 1 && (int64_t)g.HWo + 1024 >= (1 << 24)))
        return fail(CPG_E_UNSUPPORTED, "conv1x1: grid too large for 32-bit tile arithmetic");
    const int64_t blocks = (g.G + Cfg::BN - 1) / Cfg::BN * g.tiles_m;
    if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "conv1x1: grid too large");
    if constexpr (!DGRAD) {
        if (mask != nullptr) {
            hipLaunchKernelGGL((k_pw<Cfg, false, false, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, (float *)nullptr,
                               (const float *)nullptr, *mask);
            CPG_CHECK_LAUNCH(what);
            return CPG_OK;
        }
        if (stats != nullptr) {
            hipLaunchKernelGGL((k_pw<Cfg, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, stats, (const float *)nullptr);
            CPG_CHECK_LAUNCH(what);
            return CPG_OK;
        }
    }
    if constexpr (DGRAD) {
        if (addend != nullptr) {
            hipLaunchKernelGGL((k_pw<Cfg, true, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, (float *)nullptr, addend);
            CPG_CHECK_LAUNCH(what);
            return CPG_OK;
        }
    }
    hipLaunchKernelGGL((k_pw<Cfg, DGRAD>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, (float *)nullptr, (const float *)nullptr);
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}

}  // namespace

extern "C" int cpg_conv1x1_supported(const cpg_conv_desc *d) {
    if (cpg::opt_on(cpg::OPT_DISABLE_CONV1X1)) return 0;
    if (!(d->R == 1 && d->S == 1 && d->pad_h == 0 && d->pad_w == 0 && d->groups == 1 && d->stride_h >= 1 && d->stride_w >= 1 &&
          d->N > 0 && d->C > 0 && d->K > 0 && d->H > 0 && d->W > 0))
        return 0;
    const int OH = (d->H - 1) / d->stride_h + 1, OW = (d->W - 1) / d->stride_w + 1;
    // whole 16-channel chunks on both sides (forward contracts over C, the input gradient over K); a tile's images are
    // addressed with 31-bit byte offsets
    const int64_t span = (int64_t)(256 / (OH * OW) + 2) * std::max(d->C, d->K) * d->H * d->W * 4;
    // (32-bit tile arithmetic: grid positions below 2^31, a plane's positions exact in fp32 -- divmod_small)
    if ((int64_t)d->N * OH * OW >= (1ll << 31) - 512 || (int64_t)OH * OW + 1024 >= (1 << 24)) return 0;
    return d->C % 16 == 0 && d->K % 16 == 0 && span < (1ll << 31) && (int64_t)std::max(d->C, d->K) * d->H * d->W < (1ll << 28);
}

size_t cpg_conv1x1_pack_workspace(const cpg_conv_desc *d) { return std::max(pack_bytes(d->C, d->K), pack_bytes(d->K, d->C)); }

// pixel tiles of the forward launch = rows of the [K][tiles][2] statistics buffer of cpg_conv2d_fwd_bnstats.  The launch below derives
// the same tile from the same inputs (K, stride, plane size, CPG_PW_TILE); where the wide tile additionally needs a 16-byte aligned input
// it REFUSES a fused-statistics launch on an unaligned one instead of falling back to another tile size.  CPG_PW_TILE is A/B tooling:
// like every planner switch it must not change between a query and the launch it sized (include/cpg_hip.h).
int cpg_conv1x1_bnstats_tiles(const cpg_conv_desc *d) {
    const int OH = (d->H - 1) / d->stride_h + 1, OW = (d->W - 1) / d->stride_w + 1;
    const bool wide = pw_wide() && d->stride_h == 1 && d->stride_w == 1 && (OH * OW) % 4 == 0;
    const int bn = d->K <= 64 ? PwV64::BN : (wide ? PwV2::BN : PwV::BN);
    return (int)(((int64_t)d->N * OH * OW + bn - 1) / bn);
}

int cpg_conv1x1_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                    float *y, void *ws, size_t ws_bytes, hipStream_t stream, float *stats) {
    const char *what = "cpg_conv2d_fwd(1x1)";
    CPG_REQUIRE(x && w && y, "%s: null pointer", what);
    const size_t need = pack_bytes(d->C, d->K);
    if (ws == nullptr || ws_bytes < need) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    CPG_REQUIRE((((uintptr_t)ws) & 15) == 0, "%s: workspace must be 16-byte aligned", what);
    float *wp = (float *)ws;
    const int rows = pad_to(d->C, 16), Mp = pad_to(d->K, 128);
    {
        const float *pre = nullptr;
        const int ps = cpg::pack_site(cpg::PackJob{1, d->K, d->C, rows, Mp, 0, 0, (long long)rows * Mp, need}, &pre, what);
        if (ps == 1) return CPG_OK;
        if (ps < 0) return ps;
        if (ps == 2)
            wp = const_cast<float *>(pre);
        else
            hipLaunchKernelGGL(k_pw_pack, dim3(stream_grid((int64_t)rows * Mp, 256)), dim3(256), 0, stream, w, pm, thr, wp, d->K, d->C, rows, Mp, 0);
    }
    const int OH = (d->H - 1) / d->stride_h + 1, OW = (d->W - 1) / d->stride_w + 1;
    PwGeom g{d->N, d->C, d->K, Mp, OW, OH * OW, d->H * d->W, d->stride_h * d->W, d->stride_w, OH * OW, OW, 1, 0, (long long)d->N * OH * OW};
    const bool dense = d->stride_h == 1 && d->stride_w == 1;
    static_assert(PwV::BN == PwS::BN && PwV64::BN == PwS64::BN, "cpg_conv1x1_bnstats_tiles counts tiles of either staging flavour");
    const bool vec = dense && (OH * OW) % 4 == 0 && (((uintptr_t)x) & 15) == 0;
    if (d->K <= 64) return vec ? launch<PwV64, false>(g, x, wp, bias, y, stream, what, stats) : launch<PwS64, false>(g, x, wp, bias, y, stream, what, stats);
    if (pw_wide() && dense && (OH * OW) % 4 == 0) {
        CPG_REQUIRE(vec || stats == nullptr, "%s: the fused-statistics launch needs a 16-byte aligned input", what);
        if (vec) return launch<PwV2, false>(g, x, wp, bias, y, stream, what, stats);
    }
    if (vec) return launch<PwV, false>(g, x, wp, bias, y, stream, what, stats);
    return launch<PwS, false>(g, x, wp, bias, y, stream, what, stats);
}

int cpg_conv1x1_dgrad(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, float *gx, void *ws,
                      size_t ws_bytes, hipStream_t stream, const float *addend) {
    const char *what = "cpg_conv2d_dgrad(1x1)";
    CPG_REQUIRE(gy && w && gx, "%s: null pointer", what);
    const size_t need = pack_bytes(d->K, d->C);
    if (ws == nullptr || ws_bytes < need) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    CPG_REQUIRE((((uintptr_t)ws) & 15) == 0, "%s: workspace must be 16-byte aligned", what);
    float *wp = (float *)ws;
    const int rows = pad_to(d->K, 16), Mp = pad_to(d->C, 128);
    {
        const float *pre = nullptr;
        const int ps = cpg::pack_site(cpg::PackJob{1, d->K, d->C, rows, Mp, 1, 0, (long long)rows * Mp, need}, &pre, what);
        if (ps == 1) return CPG_OK;
        if (ps < 0) return ps;
        if (ps == 2)
            wp = const_cast<float *>(pre);
        else
            hipLaunchKernelGGL(k_pw_pack, dim3(stream_grid((int64_t)rows * Mp, 256)), dim3(256), 0, stream, w, pm, thr, wp, d->K, d->C, rows, Mp, 1);
    }
    const int OH = (d->H - 1) / d->stride_h + 1, OW = (d->W - 1) / d->stride_w + 1;
    const bool dense = d->stride_h == 1 && d->stride_w == 1;
    if (!dense) {       // positions the strided conv never read receive no gradient
        hipError_t e = hipMemsetAsync(gx, 0, (size_t)d->N * d->C * d->H * d->W * sizeof(float), stream);
        if (e != hipSuccess) return hip_status(e, what);
    }
    // reads gy (K channels, dense over the output grid), produces gx (C channels) at the strided positions
    PwGeom g{d->N, d->K, d->C, Mp, OW, OH * OW, OH * OW, OW, 1, d->H * d->W, d->stride_h * d->W, d->stride_w, 0, (long long)d->N * OH * OW};
    if (addend != nullptr && !dense) return fail(CPG_E_UNSUPPORTED, "%s: the fused addend needs a dense (stride 1) layer", what);
    const bool vec = (OH * OW) % 4 == 0 && (((uintptr_t)gy) & 15) == 0;
    if (d->C <= 64) return vec ? launch<PwV64, true>(g, gy, wp, nullptr, gx, stream, what, nullptr, addend) : launch<PwS64, true>(g, gy, wp, nullptr, gx, stream, what, nullptr, addend);
    if (vec && pw_wide()) return launch<PwV2, true>(g, gy, wp, nullptr, gx, stream, what, nullptr, addend);
    if (vec) return launch<PwV, true>(g, gy, wp, nullptr, gx, stream, what, nullptr, addend);
    return launch<PwS, true>(g, gy, wp, nullptr, gx, stream, what, nullptr, addend);
}

// dense pointwise layers whose activations can be staged as aligned float4 and addressed with 31-bit byte offsets
extern "C" int cpg_conv1x1_wgrad_supported(const cpg_conv_desc *d) {
    if (cpg::opt_on(cpg::OPT_DISABLE_CONV1X1_WGRAD) || !cpg_conv1x1_supported(d)) return 0;
    const int64_t hw = (int64_t)d->H * d->W;
    // (strided layers and planes that are not multiples of 4 pixels take the kernel's per-pixel staging)
    return (int64_t)d->N * std::max(d->C, d->K) * hw * 4 < (1ll << 31) && (int64_t)d->N * hw < (1ll << 29);
}

size_t cpg_conv1x1_wgrad_workspace(const cpg_conv_desc *d) {
    switch (pw_wgrad_pick(d)) {
        case 1: return pw_wgrad_plan<PwW64o>(d).ws_bytes;
        case 2: return pw_wgrad_plan<PwW64i>(d).ws_bytes;
        case 3: return pw_wgrad_plan<PwW64>(d).ws_bytes;
        default: return pw_wgrad_plan<PwW128>(d).ws_bytes;
    }
}

int cpg_conv1x1_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr,
                      float *gw, float *gpm, void *ws, size_t ws_bytes, hipStream_t stream) {
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    switch (pw_wgrad_pick(d)) {
        case 1: return pw_wgrad_launch<PwW64o>(d, x, gy, ep, ws, ws_bytes, stream);
        case 2: return pw_wgrad_launch<PwW64i>(d, x, gy, ep, ws, ws_bytes, stream);
        case 3: return pw_wgrad_launch<PwW64>(d, x, gy, ep, ws, ws_bytes, stream);
        default: return pw_wgrad_launch<PwW128>(d, x, gy, ep, ws, ws_bytes, stream);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same two kernels as plain GEMMs (N = 1 "image"): the masked linear layers without a piggymask (task 1) run on them.
//   nt: D[M][C] = A[M][K] . B[C][K]^T (both operands K-contiguous)      = k_pw_wgrad with HWo = K    (linear forward)
//   nn: D[M][G] = Wp[Kd][Mp]^T . X[Kd][G] (both operands K-major)       = k_pw with one image        (linear dgrad, wgrad)
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct NtPlan {
    int tiles_co, tiles_ci, nsplit, units_per_split;
    size_t ws_bytes;
};
template <class Cfg>
NtPlan nt_plan(int M, int C, int64_t K) {
    NtPlan p;
    p.tiles_co = (M + Cfg::BCO - 1) / Cfg::BCO;
    p.tiles_ci = (C + Cfg::BCI - 1) / Cfg::BCI;
    const int64_t units = (K + Cfg::PIX - 1) / Cfg::PIX, tiles = (int64_t)p.tiles_co * p.tiles_ci;
    // split blocks per CU: 4; the 32-row tiles (an HBM-bound stream of the other operand) 2 = one round of the two resident blocks,
    // half the partial sums
    const int bpc = std::max(1, opt_or(OPT_PWW_BPC, Cfg::BCO == 32 ? 2 : 4));
    int64_t want = ((int64_t)bpc * kCUs + tiles - 1) / tiles;
    want = std::max<int64_t>(1, std::min<int64_t>(want, (units + 7) / 8));
    want = (want + kXCDs - 1) / kXCDs * kXCDs;
    p.units_per_split = (int)((units + want - 1) / want);
    p.nsplit = (int)want;
    p.ws_bytes = (size_t)p.nsplit * M * C * sizeof(float);
    return p;
}
}  // namespace

bool cpg_pw_gemm_nt_ok(const float *A, const float *B, int M, int C, int64_t K) {
    return !cpg::opt_on(cpg::OPT_DISABLE_PW_GEMM) && K % 4 == 0 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0 &&
           (int64_t)std::max(M, C) * K * 4 < (1ll << 31) && K < (1ll << 29);
}
// the tile by the row count: 128 rows; 64 (<= 64 rows); 32 x 128 on 1 x 4 waves (<= 32 rows: the linear forward at <= 32 images per GPU;
// features.45 at batch 32: 0.245 -> 0.081 ms, 5.1 TB/s of weight bytes; the 32 x 256 tile: 0.087)
using PwW32 = PwWCfg<32, 128, 1>;
using PwW32n = PwWCfg<32, 256, 1>;       // (development: CPG_FC_SMALL bit 8)
namespace {
template <class Cfg, bool MASKB>
int nt_launch(const float *A, const float *B, const float *pmB, float thr, int M, int C, int64_t K, const Epilogue &ep, void *ws, size_t ws_bytes,
              hipStream_t stream, const char *what) {
    const NtPlan p = nt_plan<Cfg>(M, C, K);
    if (ws == nullptr || ws_bytes < p.ws_bytes) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, p.ws_bytes);
    // k_pw_wgrad(M, C, HWo, G, ...): "gy" = A with M rows, "x" = B with C rows, one image of HWo = G = K pixels; MASKB: B * bin(pmB) in staging
    hipLaunchKernelGGL((k_pw_wgrad<Cfg, false, MASKB>), dim3((unsigned)(p.tiles_co * p.tiles_ci * p.nsplit)), dim3(256), 0, stream, M, C, (int)K,
                       (long long)K, p.tiles_co, p.tiles_ci, p.units_per_split, B, A, (float *)ws, PwWX{0, 0, 0, 0}, pmB, thr);
    launch_split_reduce((const float *)ws, p.nsplit, (int64_t)M * C, 0, ep, stream);
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}
inline int nt_rows(int M) {
    const int o = cpg::opt_or(cpg::OPT_FC_SMALL, 1);
    return o == 0 ? 128 : M <= 32 ? ((o & 256) ? 33 : 32) : M <= 64 ? 64 : 128;
}
}  // namespace
size_t cpg_pw_gemm_nt_workspace(int M, int C, int64_t K) {      // (of every tile a switch could select: options may change between query and call)
    return std::max(std::max(nt_plan<PwW128>(M, C, K).ws_bytes, nt_plan<PwW32n>(M, C, K).ws_bytes),
                    std::max(nt_plan<PwW32>(M, C, K).ws_bytes, nt_plan<PwW64o>(M, C, K).ws_bytes));
}
int cpg_pw_gemm_nt(const float *A, const float *B, int M, int C, int64_t K, const Epilogue &ep, void *ws, size_t ws_bytes,
                   hipStream_t stream, const char *what) {
    const int r = nt_rows(M);
    if (r == 32) return nt_launch<PwW32, false>(A, B, nullptr, 0.0f, M, C, K, ep, ws, ws_bytes, stream, what);
    if (r == 33) return nt_launch<PwW32n, false>(A, B, nullptr, 0.0f, M, C, K, ep, ws, ws_bytes, stream, what);
    if (r == 64) return nt_launch<PwW64o, false>(A, B, nullptr, 0.0f, M, C, K, ep, ws, ws_bytes, stream, what);
    return nt_launch<PwW128, false>(A, B, nullptr, 0.0f, M, C, K, ep, ws, ws_bytes, stream, what);
}
// the same product with B masked in staging: D[M][C] = A[M][K] . (B * bin(pmB))[C][K]^T
int cpg_pw_gemm_nt_maskb(const float *A, const float *B, const float *pmB, float thr, int M, int C, int64_t K, const Epilogue &ep, void *ws,
                         size_t ws_bytes, hipStream_t stream, const char *what) {
    const int r = nt_rows(M);
    if (r == 32) return nt_launch<PwW32, true>(A, B, pmB, thr, M, C, K, ep, ws, ws_bytes, stream, what);
    if (r == 33) return nt_launch<PwW32n, true>(A, B, pmB, thr, M, C, K, ep, ws, ws_bytes, stream, what);
    if (r == 64) return nt_launch<PwW64o, true>(A, B, pmB, thr, M, C, K, ep, ws, ws_bytes, stream, what);
    return nt_launch<PwW128, true>(A, B, pmB, thr, M, C, K, ep, ws, ws_bytes, stream, what);
}

bool cpg_pw_gemm_nn_ok(const float *X, int M, int Mp, int Kd, int64_t G, bool masked) {
    // no split-K here: the output tiles alone must fill the chip (4096 -> 4096 dgrad at batch 256 has 38 of them: 4x slower)
    // (the pixel tile of the launch that follows: cpg_pw_gemm_nn takes the wide tile under pw_wide(); the masked forms -- cpg_pw_gemm_nn_masked /
    //  _maskx, `masked` -- always run the 224-pixel PwV / PwVM tiles)
    const int64_t bn = (!masked && G % 4 == 0 && pw_wide()) ? PwV2::BN : PwV::BN;
    if (((G + bn - 1) / bn) * ((M + 127) / 128) < 192) return false;
    return !cpg::opt_on(cpg::OPT_DISABLE_PW_GEMM) && Kd % 16 == 0 && M % 8 == 0 && Mp % 128 == 0 && Mp >= M && G < (1ll << 28) &&
           (int64_t)Kd * G * 4 < (1ll << 31) && (int64_t)M * G < (1ll << 31) && (((uintptr_t)X) & 15) == 0;
}
// y[M][G] (+ bias[m]) from K-major Wp (row stride Mp >= M, a multiple of 128; rows beyond Kd are never read)
int cpg_pw_gemm_nn(const float *wp, int Mp, const float *X, int M, int Kd, int64_t G, const float *bias, float *y, hipStream_t stream,
                   const char *what) {
    PwGeom g{1, Kd, M, Mp, (int)G, (int)G, (int)G, 0, 1, (int)G, 0, 1, 0, (long long)G};
    if (G % 4 == 0 && pw_wide()) return launch<PwV2, false>(g, X, wp, bias, y, stream, what);
    if (G % 4 == 0) return launch<PwV, false>(g, X, wp, bias, y, stream, what);
    return launch<PwS, false>(g, X, wp, bias, y, stream, what);
}
// the same GEMM with the K-major operand masked in staging: y[M][G] = Wp^T . (X * bin(pmX)), pmX laid out like X (two resident blocks
// per CU: the piggymask's staging registers)
using PwVM = PwCfg<128, 4, 1, 7, PW_CK, true, 2>;
int cpg_pw_gemm_nn_maskx(const float *wp, int Mp, const float *X, const float *pmX, float thr, int M, int Kd, int64_t G, float *y,
                         hipStream_t stream, const char *what) {
    PwGeom g{1, Kd, M, Mp, (int)G, (int)G, (int)G, 0, 1, (int)G, 0, 1, 0, (long long)G};
    const PwMask mk{pmX, nullptr, nullptr, thr};
    g.tiles_m = (g.M + PwVM::BM - 1) / PwVM::BM;
    pw_finish_geom(g);
    if (g.G >= (1ll << 31) - 512) return fail(CPG_E_UNSUPPORTED, "%s: grid too large for 32-bit tile arithmetic", what);
    const int64_t blocks = (g.G + PwVM::BN - 1) / PwVM::BN * g.tiles_m;
    if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "%s: grid too large", what);
    hipLaunchKernelGGL((k_pw<PwVM, false, false, false, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, X, wp, (const float *)nullptr, y,
                       (float *)nullptr, (const float *)nullptr, mk);
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}
// the same GEMM with the autograd epilogue of bin(pm) * W: gw[M][G] = D * bin(pm), gpm[M][G] = D * w (pm, w, gpm laid out like gw)
int cpg_pw_gemm_nn_masked(const float *wp, int Mp, const float *X, int M, int Kd, int64_t G, const float *pm, const float *w, float thr,
                          float *gw, float *gpm, hipStream_t stream, const char *what) {
    PwGeom g{1, Kd, M, Mp, (int)G, (int)G, (int)G, 0, 1, (int)G, 0, 1, 0, (long long)G};
    const PwMask mk{pm, w, gpm, thr};
    if (G % 4 == 0) return launch<PwV, false>(g, X, wp, nullptr, gw, stream, what, nullptr, nullptr, &mk);
    return launch<PwS, false>(g, X, wp, nullptr, gw, stream, what, nullptr, nullptr, &mk);
}
// K-major transpose of a row-major [R][Cc] matrix into wp[Cc (padded to 16)][R (padded to 128)]
void cpg_pw_pack_transpose(const float *a, int R, int Cc, float *wp, hipStream_t stream) {
    const int rows = pad_to(Cc, 16), Mp = pad_to(R, 128);
    hipLaunchKernelGGL(k_pw_pack, dim3(stream_grid((int64_t)rows * Mp, 256)), dim3(256), 0, stream, a, (const float *)nullptr, 0.f, wp, R, Cc,
                       rows, Mp, 0);
}
size_t cpg_pw_pack_transpose_bytes(int R, int Cc) { return (size_t)(pad_to(Cc, 16) + 16) * pad_to(R, 128) * sizeof(float); }
