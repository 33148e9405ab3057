// Specialised masked 3x3 / stride 1 / pad 1 convolution on fp32 MFMA -- the shape class of every
// VGG16 conv (100 % of config 1-3 conv FLOPs), 16 of SphereNet-20's 20 convs and ResNet's 3x3 s1.
// (docs/LAB_NOTEBOOK.md section 4.1 has the measurements behind every choice below.)
//
// forward + input-gradient (one kernel, k_c3_fwd; only the packed weights differ):
//   k_c3_pack writes Wp[(c*9 + tap)][m] = W * bin(piggymask), K-major and zero padded (binarise +
//   transpose fused in one tiny pass).  Block = BM output channels x a pixel tile (TH x TW of one
//   image, 4 x 28 of two images, or 32 "virtual rows" of 7-wide images); loop over input channels in
//   chunks of 4.  Per chunk the block stages
//     Ws[36][BM+4]            rows of Wp (float4 global loads, ds_write_b128)
//     Xs[4][TH+2][TW+2]       the zero-padded input patch -- every element is reused by 9 taps x BM
//                             channels out of LDS ("im2col in LDS", no HBM im2col buffer); fetched
//                             with range-checked buffer loads, so padding costs no select
//   into the OTHER LDS stage while each wave runs 18 k-steps of v_mfma_f32_32x32x2_f32 on the current
//   one; operand addresses are lane_base + compile-time immediates.  K order inside a chunk is
//   (channel pair, tap): lanes 0-31 take channel 2p, lanes 32-63 channel 2p+1, same tap.  The chunk
//   body is branch free with a pinned MFMA / LDS / VMEM interleave (sched_group_barrier).
//   dgrad is the same contraction with roles swapped: input = gy, output channels = ci, and the
//   weights packed as Wp[(co, 8-tap)][ci] (spatially flipped taps).
//
// weight-gradient (k_c3_wgrad):
//   block = 64 co x 64 ci x 9 taps, K = pixels.  Each wave owns a 32 co x 32 ci fragment for ALL 9
//   taps (9 accumulators): one gy operand read feeds 9 MFMAs against a sliding window over the LDS
//   input patch.  Staging = one buffer_load_dword + one ds_write_b32 per element, spread between the
//   MFMAs.  Split-K over (image, pixel-tile) units; tap-major partials reduced by k_split_reduce
//   (igemm_core.h), which also applies the autograd epilogue gW = g*bin(pm), gPM = g*W.
#include <algorithm>
#include <cmath>
#include "igemm_core.h"

using namespace cpg;

namespace {

// Launder a value through an empty asm so the compiler cannot hoist the staging index arithmetic that
// depends on it out of the chunk loop: the hoisted form costs 1-2 VGPRs per staged element (40-100
// registers) and with it a wave per SIMD; recomputing costs VALU cycles the MFMA-bound loop has spare.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// same, and additionally a compiler-level memory barrier (orders the staging batches of k_c3_wgrad)
__device__ __forceinline__ int opaque_mem(int v) {
    asm volatile("" : "+v"(v) : : "memory");
    return v;
}

// uniform base + 32-bit per-lane BYTE offset: the shape LLVM folds into `global_load_dword v, v_off, s[base:base+1]`
// (one VGPR of address per load instead of a 64-bit pair computed with VALU adds)
__device__ __forceinline__ float ld_sv(const float *sbase, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sbase) + byte_off);
}
__device__ __forceinline__ f32x4 ld_sv4(const float *sbase, unsigned byte_off) {
    return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(sbase) + byte_off);
}


// Inference-mode BatchNorm (+ ReLU) folded into the forward epilogue: y = [max(0,] (acc + bias - mean) * invstd * gamma + beta [)]
// with invstd = 1 / sqrt(var + eps) -- the expression of bn_kernels.hip's apply pass.  gamma == nullptr: plain conv.
// `live` (may be null; inference only): liveness of the effective weights, written by k_c3_pack -- live[m] != 0 when output
// channel m has a non-zero weight, live[Mp + 4 + q] != 0 when input-channel chunk q (4 channels) has one; live[Mp] receives
// 4 * (number of chunks up to the last live one), live[Mp + 1] counts the output tiles that were skipped (diagnostics).  A block whose BM output channels are all dead skips its MFMA loop (their
// conv output is exactly 0, the epilogue still writes bias / BatchNorm of 0), and every block stops after the last live
// input-channel chunk.  Whole channels die when a model that was GROWN for later tasks serves an earlier, narrower task
// after apply_mask (every slot with owner > task is zero): the kernel then does the work of the cropped model.
struct C3BnEval {
    const float *gamma, *beta, *mean, *var;
    float eps;
    int relu;
    int *live;
};

// Input-gradient launches whose result is the gradient w.r.t. a = relu(bn(ypre)) of the PREVIOUS layer can do that BatchNorm's
// backward reduction in their epilogue (BRED): per element g' = g * [bn(ypre) > 0] is what gets stored, and sum g', sum g' * xhat
// (xhat = (ypre - mean) * invstd) are accumulated per channel and pixel tile -> partials[channel][tile][2], exactly like the
// forward statistics (STATS).  The BatchNorm backward then needs no reduction pass over ypre and g (cpg_bn_bwd_from_partials).
struct C3BnBwd {
    const float *ypre, *gamma, *beta, *mean, *invstd;
};

struct C3Geom {
    int N, C, H, W, M;        // C: channels of the tensor being read, M: channels being produced
    int Mp;                   // row stride of the packed weights (M rounded up to 128)
    int tiles_x, tiles_y, tiles_m;
    int OH, OW;               // output map (= H, W for the stride-1 tiles; the strided forward: (H - 1) / 2 + 1)
    int dgrad;                // host side only: which instantiation to launch
    int ksplit;               // 1, or 2: the channel chunks of a tile are shared by two blocks that atomically add into a zeroed y
};

// S2 = 1: the STRIDE-2 forward (3x3, pad 1: ResNet's conv2 of a down-sampling block, SphereNet's conv{s}_1).  TH x TW is the tile of
// the OUTPUT map; its input patch is (2 TH + 1) x (2 TW + 1), staged with the columns de-interleaved -- a patch row holds its
// TW + 1 even columns, then its TW odd ones -- so that the B operand of a tap is 32 CONSECUTIVE words for 32 consecutive output
// columns (a stride-2 read would put two lanes on every LDS bank): tap column s = 0 / 1 / 2 of output column c is slot c,
// TW + 1 + c, c + 1.  Everything else (weights, k order, MFMA loop, epilogues) is the stride-1 kernel's.
template <int BM_, int TH_, int TW_, int WM_, int WN_, int CK_, int MINW_, int NIMG_ = 1, int VROWS_ = 0, int S2_ = 0>
struct C3Cfg {
    static constexpr int S2 = S2_;
    static_assert(S2_ == 0 || VROWS_ == 0, "no virtual-row tiles for the strided forward");
    // VROWS > 0 ("virtual rows", for maps that are exactly TH x TW -- smaller than any sensible tile): the rows of ALL images
    // are numbered consecutively (R = n * TH + h) and a tile is VROWS consecutive rows, straddling images.  7 x 7 maps: 32
    // virtual rows x 7 columns = 224 pixels = 7 fragments exactly, where the 14 x 16 single-image tile wastes 78 %.
    static constexpr int VROWS = VROWS_;
    static_assert(VROWS == 0 || NIMG_ >= (TH_ - 1 + VROWS_ + TH_ - 1) / TH_, "a virtual-row tile must fit in NIMG images");
    static constexpr int NIMG = NIMG_;                      // images per tile (2: a 4x28 strip of two images = 7 fragments)
    static constexpr int MINW = MINW_;                      // waves per SIMD the register allocator must leave room for
    static constexpr int BM = BM_, TH = TH_, TW = TW_, WM = WM_, WN = WN_, CK = CK_;
    static constexpr int TPIX = TH * TW;                    // pixels of one image in the tile
    static constexpr int BN = VROWS ? VROWS * TW : NIMG * TPIX;
    static_assert(WM * WN == 4 && BN % (32 * WN) == 0 && BM % (32 * WM) == 0 && CK % 2 == 0, "bad conv3x3 config");
    static constexpr int FM = BM / 32 / WM, FN = BN / 32 / WN;
    static constexpr int PH = S2 ? 2 * TH + 1 : TH + 2, PW = S2 ? 2 * TW + 1 : TW + 2, IPLANE = PH * PW, PLANE = NIMG * IPLANE;
    static constexpr int RSTEP = S2 ? 2 : 1;                // patch rows per output row
    // patch offset of tap (r, s) relative to the lane's output pixel
    static constexpr int tap_off(int tap) { return (tap / 3) * PW + (S2 ? (tap % 3 == 0 ? 0 : tap % 3 == 1 ? TW + 1 : 1) : tap % 3); }
    static constexpr int LDW = BM + 4;                      // +4: rows stay 16-byte aligned for ds_write_b128
    static constexpr int KC = CK * 9;                       // k extent of one chunk
    static constexpr int W4 = BM / 4;                       // float4 per weight row
    static constexpr int WROWS = 256 / W4;                  // weight rows staged per pass of the block
    static constexpr int NW4 = (KC + WROWS - 1) / WROWS;    // float4 loads per thread per chunk
    static constexpr int X_ELEMS = CK * PLANE;
    static constexpr int NXL = (X_ELEMS + 255) / 256;       // patch elements per thread per chunk
    // both LDS regions are padded to whole staging passes, so that every staging store is unconditional
    static constexpr int W_ELEMS = NW4 * WROWS * LDW, XS_ELEMS = NXL * 256;
    static constexpr int STAGE = W_ELEMS + XS_ELEMS;
    static constexpr int SMEM_FLOATS = 2 * STAGE;
    static constexpr int NS = CK / 2 * 9;                   // k-steps (of 2 channels x 1 tap) per chunk
    static constexpr int NITEMS = NW4 + NXL;                // staging loads (= staging stores) per thread per chunk
    // staging items per k-step: 1 for every stride-1 tile; the strided forward's patch is 4x the pixels per output and rides 2 per step
    static constexpr int IPS = (NITEMS + NS - 1) / NS;
    static constexpr int LSTEPS = (NITEMS + IPS - 1) / IPS; // k-steps that carry loads (the first LSTEPS) / stores (the last LSTEPS)
    static_assert(IPS <= 2 && LSTEPS <= NS, "at most two staging loads and two stores per k-step");
};

// ------------------------------------------------------------------------------ weight pack
// Wp[(c*9 + tap')][m]  (row stride Mp, zero padded to CK channels x 128 columns)
//   fwd  : c = ci, m = co, tap' = tap        value = W[co][ci][tap] * bin(pm)
//   dgrad: c = co, m = ci, tap' = 8 - tap    (spatially flipped: conv of gy with the transposed filter)
// One fused pass: binarise + mask + transpose into the K-major layout the conv kernel streams with
// float4 loads.  Conv weights are 1.7 k ... 2.4 M elements per layer (59 MB for all of VGG16), so this
// costs microseconds; rocprof PMC showed the alternative -- gathering W[co][ci][tap] inside the conv
// kernel -- at 13.4 VALU instructions per MFMA and 57 % MFMA utilisation (profiles/r01c_pmc.md).
__global__ __launch_bounds__(256) void k_c3_pack(const float *__restrict__ w, const float *__restrict__ pm, float thr,
                                                 float *__restrict__ out, int K, int C, int rows_c, int Mp, int dgrad,
                                                 int *__restrict__ live) {
    // out index o = (c*9 + tp) * Mp + m ; consecutive threads -> consecutive m (coalesced writes)
    const int64_t total = (int64_t)rows_c * 9 * Mp;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += nthreads) {
        const int m = (int)(o % Mp);
        const int r = (int)(o / Mp);
        const int c = r / 9, tp = r - c * 9;
        const int co = dgrad ? c : m, ci = dgrad ? m : c, tap = dgrad ? 8 - tp : tp;
        float v = 0.0f;
        if (co < K && ci < C) {
            const int64_t off = ((int64_t)co * C + ci) * 9 + tap;
            v = w[off];
            if (pm != nullptr) v *= binarize(pm[off], thr);
        }
        out[o] = v;
        // liveness (forward flavour, zeroed by the caller): plain stores of 1 by the lanes that hold a non-zero -- one flag per
        // output channel, one per 4-channel input chunk (same-value races are benign; no atomics: a single-address atomicMax
        // per wave serialised at L2 and cost the validate pass 4 ms)
        if (live != nullptr && v != 0.0f) {
            live[m] = 1;
            live[Mp + 4 + c / 4] = 1;
        }
    }
}

// ------------------------------------------------------------------------------ fwd / dgrad
// y[n][m][h][w] = sum_{c,tap} Wp[(c*9+tap)][m] * x[n][c][h + tap/3 - 1][w + tap%3 - 1]
// (DGRAD changes nothing in the code: the two passes differ only in the packed weights.  It gives the input-gradient
// launches their own kernel name, so that a rocprofv3 kernel summary separates the conv_fwd and conv_dgrad families.)
// STATS: the forward of a conv that feeds a training-mode BatchNorm also emits, per block, the sum and the sum of squares of
// its outputs per channel -- stats[channel][pixel tile][2] -- so the BatchNorm needs no statistics pass over y
// (cpg_conv2d_fwd_bnstats + cpg_bn_stats_finalize; deterministic: fixed-order merges only).
// SPLITK: g.ksplit blocks share a tile's channel chunks and atomically add their accumulators into a zeroed y.  Two addends
// per element commute, so the result does not depend on which block arrives first (0 + a + b = 0 + b + a bitwise).
template <class Cfg, bool DGRAD, bool STATS = false, bool SPLITK = false, bool BRED = false>
__global__ __launch_bounds__(256, Cfg::MINW) void k_c3_fwd(C3Geom g, const float *__restrict__ x, const float *__restrict__ wp,
                                                           const float *__restrict__ bias, float *__restrict__ y,
                                                           float *__restrict__ stats, C3BnEval bn, C3BnBwd bb) {
    static_assert(!BRED || (DGRAD && !STATS && !SPLITK), "the BatchNorm-backward reduction rides in plain input-gradient launches");
    static_assert(!Cfg::S2 || (!DGRAD && !SPLITK && !BRED), "the strided tiles are forward-only (its input gradient is k_c3s2_dgrad)");
    __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int li = lane & 31, lh = lane >> 5;

    // block -> (m tile, x tile, y tile, image); m fastest so co-resident blocks of an XCD share the patch
    unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lb % g.tiles_m; lb /= g.tiles_m;
    int ksp = 0;
    if (SPLITK) {
        ksp = lb % g.ksplit;
        lb /= g.ksplit;
    }
    const unsigned tile_n = lb;                          // pixel-tile index (STATS)
    const int tx = lb % g.tiles_x; lb /= g.tiles_x;
    const int ty = lb % g.tiles_y;
    // first image of the tile; virtual rows: tile k starts at row vr0 of image n (tiles_x = tiles_y = 1, h0 = w0 = 0)
    const int n = Cfg::VROWS ? (int)(lb * Cfg::VROWS) / Cfg::TH : (int)(lb / g.tiles_y) * Cfg::NIMG;
    const int vr0 = Cfg::VROWS ? (int)(lb * Cfg::VROWS) % Cfg::TH : 0;
    const int m0 = tm * Cfg::BM, h0 = ty * Cfg::TH, w0 = tx * Cfg::TW;
    const int HW = g.H * g.W;
    // tile pixel t -> (image of the tile, row, column inside that image's patch window)
    auto pixel = [&](int t, int &img, int &r, int &c) {
        if (Cfg::VROWS) {
            const int R = vr0 + t / Cfg::TW;
            img = R / Cfg::TH, r = R % Cfg::TH, c = t % Cfg::TW;
        } else {
            const int tt = t % Cfg::TPIX;
            img = t / Cfg::TPIX, r = tt / Cfg::TW, c = tt % Cfg::TW;
        }
    };

    // ---- staging descriptors, all fixed for the life of the block (a handful of registers) ----
    // weights: float4 (row = wrow0 + WROWS*i, 4 columns at wcol); Wp is zero padded (and carries WROWS rows of
    // slack for the last pass), so no guards
    const int wcol = (tid % Cfg::W4) * 4, wrow0 = tid / Cfg::W4;
    const int wdst = wrow0 * Cfg::LDW + wcol;
    const unsigned wbyte = (unsigned)(wrow0 * g.Mp + m0 + wcol) * 4u;
    // patch: element e = tid + 256*i of [CK][NIMG][PH][PW], fetched with buffer loads whose range check does the
    // zero padding: byte offset from image n's channel 0, or 0x80000000 (>= num_records -> the load returns 0
    // without touching memory) for halo positions outside the image.  Channels past C (ragged last chunk) fall
    // out of range by themselves because num_records ends at the image's (NIMG = 2: the image pair's) last channel.
    constexpr int kOutOfRange = (int)0x80000000;
    int xbyte[Cfg::NXL];
#pragma unroll
    for (int i = 0; i < Cfg::NXL; ++i) {
        const int e = tid + 256 * i;
        const int cl = e / Cfg::PLANE, rem0 = e - cl * Cfg::PLANE;
        const int img = rem0 / Cfg::IPLANE, rem = rem0 - img * Cfg::IPLANE;
        const int pr = rem / Cfg::PW, slot = rem - pr * Cfg::PW;
        // strided forward: slots 0..TW of a patch row are its even columns, slots TW+1..2TW the odd ones (see C3Cfg)
        const int pc = Cfg::S2 ? (slot <= Cfg::TW ? 2 * slot : 2 * (slot - Cfg::TW - 1) + 1) : slot;
        const int gh = Cfg::RSTEP * h0 - 1 + pr, gw = Cfg::RSTEP * w0 - 1 + pc;
        const bool ok = e < Cfg::X_ELEMS && n + img < g.N && (unsigned)gh < (unsigned)g.H && (unsigned)gw < (unsigned)g.W;
        xbyte[i] = ok ? ((img * g.C + cl) * HW + gh * g.W + gw) * 4 : kOutOfRange;
    }
    const int nimg_here = min(Cfg::NIMG, g.N - n);
    const __amdgpu_buffer_rsrc_t srd_x =
        __builtin_amdgcn_make_buffer_rsrc((void *)(x + (int64_t)n * g.C * HW), 0, nimg_here * g.C * HW * 4, 0x00020000);

    // staging item k < NW4: weight float4 k; else patch element k - NW4.  Results stay untouched in registers
    // until store_item(): arithmetic on a loaded value would drag its s_waitcnt in front of the MFMAs.
    f32x4 rw[Cfg::NW4];             // native vector type: HIP's float4 struct kept this array in scratch memory
    float rx[Cfg::NXL];
    auto load_item = [&](int k, int c0) {
        if (k < Cfg::NW4)
            rw[k] = ld_sv4(wp + ((int64_t)c0 * 9 + Cfg::WROWS * k) * g.Mp, wbyte);
        else
            rx[k - Cfg::NW4] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, xbyte[k - Cfg::NW4] + c0 * HW * 4, 0, 0));
    };
    auto store_item = [&](int k, float *stage) {
        if (k < Cfg::NW4)
            *reinterpret_cast<f32x4 *>(stage + wdst + Cfg::WROWS * k * Cfg::LDW) = rw[k];
        else
            stage[Cfg::W_ELEMS + tid + 256 * (k - Cfg::NW4)] = rx[k - Cfg::NW4];
    };

    // ---- operand lane bases ----
    const int a_base = lh * 9 * Cfg::LDW + wm * Cfg::FM * 32 + li;
    int b_base[Cfg::FN];
#pragma unroll
    for (int fn = 0; fn < Cfg::FN; ++fn) {
        int img, r, c;
        pixel((wn * Cfg::FN + fn) * 32 + li, img, r, c);
        b_base[fn] = lh * Cfg::PLANE + img * Cfg::IPLANE + Cfg::RSTEP * r * Cfg::PW + c;
    }

    f32x16 acc[Cfg::FM][Cfg::FN];
#pragma unroll
    for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[fm][fn][e] = 0.0f;

    int nch_all = (g.C + Cfg::CK - 1) / Cfg::CK;
    if (!DGRAD && !STATS && !SPLITK && bn.live != nullptr) {                      // inference: skip what apply_mask killed
        int alive = 0;
        for (int i = lane; i < Cfg::BM; i += 64) alive |= (m0 + i < g.M) ? bn.live[m0 + i] : 0;
        const bool dead = __ballot(alive != 0) == 0ull;                          // wave-uniform; same answer in all 4 waves
        static_assert(Cfg::CK == 4, "the liveness chunks of k_c3_pack are 4 channels");
        int last = nch_all;                                                      // trailing dead input chunks (uniform scalar loop:
        while (last > 0 && bn.live[g.Mp + 4 + last - 1] == 0) --last;            //  one read when nothing is dead)
        nch_all = dead ? 0 : last;
        if (tid == 0) {
            if (dead) atomicAdd(&bn.live[g.Mp + 1], 1);
            if (blockIdx.x == 0) bn.live[g.Mp] = last * Cfg::CK;
        }
    }
    const int per_split = SPLITK ? (nch_all + g.ksplit - 1) / g.ksplit : nch_all;
    const int ch0 = ksp * per_split, nch = min(nch_all, ch0 + per_split);        // this block's chunks: [ch0, nch)
#pragma unroll
    for (int k = 0; k < Cfg::NITEMS; ++k) load_item(k, ch0 * Cfg::CK);
#pragma unroll
    for (int k = 0; k < Cfg::NITEMS; ++k) store_item(k, smem + (ch0 & 1) * Cfg::STAGE);
    __syncthreads();
    // One chunk = NS k-steps; k-step s = (channel pair p = s / 9, tap = s % 9): lanes 0-31 hold channel 2p, lanes
    // 32-63 channel 2p+1.  The body is branch free and its instruction order pinned (sched_group_barrier):
    //   * the operands of step s+1 are read from LDS while the MFMAs of step s run (two register sets);
    //   * the next chunk's staging loads ride in the first NITEMS steps, its LDS stores (into the OTHER stage)
    //     in the last NITEMS steps -- never a staging-only phase, which the block's waves (two per SIMD, in
    //     lock step) could not hide from one another.  The chunk index is clamped, so the last chunk re-stages
    //     itself instead of branching.
    for (int ch = ch0; ch < nch; ++ch) {
        const float *ws = smem + (ch & 1) * Cfg::STAGE;
        const float *xs = ws + Cfg::W_ELEMS;
        float *other = smem + ((ch + 1) & 1) * Cfg::STAGE;
        const int c_next = min(ch + 1, nch - 1) * Cfg::CK;
        float a[2][Cfg::FM], b[2][Cfg::FN];
        auto lds_operands = [&](int st, int set) {
            const int p = st / 9, tap = st % 9;
#pragma unroll
            for (int fm = 0; fm < Cfg::FM; ++fm) a[set][fm] = ws[a_base + (2 * p * 9 + tap) * Cfg::LDW + fm * 32];
#pragma unroll
            for (int fn = 0; fn < Cfg::FN; ++fn)
                b[set][fn] = xs[b_base[fn] + 2 * p * Cfg::PLANE + Cfg::tap_off(tap)];
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
#pragma unroll
            for (int j = 0; j < Cfg::IPS; ++j) {
                const int kl = st * Cfg::IPS + j, ks = (st - (Cfg::NS - Cfg::LSTEPS)) * Cfg::IPS + j;
                if (st < Cfg::LSTEPS && kl < Cfg::NITEMS) load_item(kl, c_next);
                if (st >= Cfg::NS - Cfg::LSTEPS && ks < Cfg::NITEMS) store_item(ks, other);
            }
#pragma unroll
            for (int i = 0; i < Cfg::FM * Cfg::FN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (i >= 1 && i <= Cfg::IPS && st < Cfg::LSTEPS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (i >= 1 && i <= Cfg::IPS && st >= Cfg::NS - Cfg::LSTEPS) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D col = pixel (lane & 31), D row = channel ----
    const int HWo = g.OH * g.OW;                         // (the map being written; = HW except for the strided forward)
    float s1[(STATS || BRED) ? Cfg::FM : 1][16], s2[(STATS || BRED) ? Cfg::FM : 1][16];
    if (STATS || BRED) {
#pragma unroll
        for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
            for (int e = 0; e < 16; ++e) s1[fm][e] = s2[fm][e] = 0.0f;
    }
    if (BRED) {
        // Input gradient w.r.t. a = relu(bn(ypre)): store g' = g * [bn(ypre) > 0] (bn_kernels.hip's bn_affine expression, so the
        // mask is the forward's) and accumulate sum g', sum g' * xhat.  Loop order (channel quad, pixel fragment): the 16
        // per-channel scalars of a quad stay in registers across the FN fragments and only 4 ypre values are in flight, which
        // keeps the epilogue inside the main loop's register budget (the fragment-major order spilled 250-470 VGPRs).
        int poffs[Cfg::FN];                                  // element offset of the lane's pixel inside a channel plane, or -1
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn) {
            int img, r, c;
            pixel((wn * Cfg::FN + fn) * 32 + li, img, r, c);
            const int oh = h0 + r, ow = w0 + c;
            const bool pok = oh < g.H && ow < g.W && n + img < g.N;
            poffs[fn] = pok ? img * g.M * HW + oh * g.W + ow : -1;
        }
        const float *ypre = bb.ypre + (int64_t)n * g.M * HW;
        float *yout = y + (int64_t)n * g.M * HW;
        // software pipeline over the FM * 4 channel quads: the 4 x FN loads of quad i + 1 are in flight while quad i is processed
        // (one HBM round trip per quad would otherwise be exposed: 4 * FM of them per block)
        float yp[2][Cfg::FN][4];
        auto quad_loads = [&](int idx, float (&dst)[Cfg::FN][4]) {
            const int fm = idx >> 2, q = idx & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = m0 + (wm * Cfg::FM + fm) * 32 + j + 8 * q + 4 * lh;
                const int cbase = co < g.M ? co * HW : -1;
#pragma unroll
                for (int fn = 0; fn < Cfg::FN; ++fn) dst[fn][j] = ypre[(poffs[fn] >= 0 && cbase >= 0) ? cbase + poffs[fn] : 0];
            }
        };
        quad_loads(0, yp[0]);
#pragma unroll
        for (int idx = 0; idx < Cfg::FM * 4; ++idx) {
            const int fm = idx >> 2, q = idx & 3;
            if (idx + 1 < Cfg::FM * 4) quad_loads(idx + 1, yp[(idx + 1) & 1]);
            float mu[4], is[4], ga[4], be[4];
            int coff[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = m0 + (wm * Cfg::FM + fm) * 32 + j + 8 * q + 4 * lh;
                const int cc = co < g.M ? co : 0;
                mu[j] = bb.mean[cc], is[j] = bb.invstd[cc], ga[j] = bb.gamma[cc], be[j] = bb.beta[cc];
                coff[j] = co < g.M ? co * HW : -1;
            }
#pragma unroll
            for (int fn = 0; fn < Cfg::FN; ++fn)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * q + j;
                    const float v = yp[idx & 1][fn][j];
                    const float xh = (v - mu[j]) * is[j];
                    const bool on = ((v - mu[j]) * is[j] * ga[j] + be[j]) > 0.0f;
                    const float gm = on ? acc[fm][fn][e] : 0.0f;
                    if (poffs[fn] >= 0 && coff[j] >= 0) {
                        yout[coff[j] + poffs[fn]] = gm;
                        s1[fm][e] += gm;
                        s2[fm][e] += gm * xh;
                    }
                }
        }
    }
#pragma unroll
    for (int fn = 0; fn < (BRED ? 0 : Cfg::FN); ++fn) {
        int img, r, c;
        pixel((wn * Cfg::FN + fn) * 32 + li, img, r, c);
        const int oh = h0 + r, ow = w0 + c;
        const bool pok = oh < g.OH && ow < g.OW && n + img < g.N;
        const int poff = oh * g.OW + ow;
        float *yout = y + (int64_t)(n + img) * g.M * HWo;
#pragma unroll
        for (int fm = 0; fm < Cfg::FM; ++fm) {
            float bv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) bv[e] = 0.0f;
            if (bias != nullptr && ksp == 0) {          // one uniform branch per fragment, 16 loads issued together
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = m0 + (wm * Cfg::FM + fm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    bv[e] = bias[co < g.M ? co : 0];
                }
            }
            if (!DGRAD && !STATS && !SPLITK && bn.gamma != nullptr) {       // eval-mode BatchNorm (+ ReLU) in the epilogue (uniform branch)
                float ga[16], be[16], mu[16], is[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = m0 + (wm * Cfg::FM + fm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const int cc = co < g.M ? co : 0;
                    ga[e] = bn.gamma[cc], be[e] = bn.beta[cc], mu[e] = bn.mean[cc], is[e] = bn.var[cc];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    is[e] = 1.0f / sqrtf(is[e] + bn.eps);
                    float v = ((acc[fm][fn][e] + bv[e]) - mu[e]) * is[e] * ga[e] + be[e];
                    if (bn.relu) v = fmaxf(v, 0.0f);
                    acc[fm][fn][e] = v;
                    bv[e] = 0.0f;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = m0 + (wm * Cfg::FM + fm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float v = acc[fm][fn][e] + bv[e];
                if (pok && co < g.M) {
                    if (SPLITK) atomicAdd(&yout[(int64_t)co * HWo + poff], v);
                    else yout[(int64_t)co * HWo + poff] = v;
                }
                if (STATS && pok) {
                    s1[fm][e] += v;
                    s2[fm][e] += v * v;
                }
            }
        }
    }
    if (STATS || BRED) {
        // per channel row: sum over the 32 pixel lanes of this half-wave, then over the WN waves sharing the channels
#pragma unroll
        for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) {
                    s1[fm][e] += __shfl_xor(s1[fm][e], off);
                    s2[fm][e] += __shfl_xor(s2[fm][e], off);
                }
        float *red = smem;                               // [WN][BM][2]; the main loop's last barrier freed the LDS
        if (li == 0) {
#pragma unroll
            for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ch = (wm * Cfg::FM + fm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    red[(wn * Cfg::BM + ch) * 2 + 0] = s1[fm][e];
                    red[(wn * Cfg::BM + ch) * 2 + 1] = s2[fm][e];
                }
        }
        __syncthreads();
        if (tid < Cfg::BM && m0 + tid < g.M) {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w2 = 0; w2 < Cfg::WN; ++w2) {
                a += red[(w2 * Cfg::BM + tid) * 2 + 0];
                b += red[(w2 * Cfg::BM + tid) * 2 + 1];
            }
            const unsigned ntiles = gridDim.x / g.tiles_m;
            float *dst = stats + ((int64_t)(m0 + tid) * ntiles + tile_n) * 2;
            dst[0] = a;
            dst[1] = b;
        }
    }
}


// ------------------------------------------------------------------------------ input gradient of the 3x3 / stride 2 / pad 1 conv
// gx[n][ci][h][w] = sum_{co, r, s : h = 2 oh + r - 1, w = 2 ow + s - 1} W[co][ci][r][s] * gy[n][co][oh][ow]
// A position (h, w) = (2 i + a, 2 j + b) only receives the taps with r odd-ness = !a, s odd-ness = !b: class (0,0) one tap, (0,1) and
// (1,0) two, (1,1) four -- nine taps per 2 x 2 group of gx, the forward's multiply-adds and not one more.  (The generic kernel ran one
// gather launch per class: 26-43 TFLOP/s.)  Here ONE block computes all four classes of a tile of (i, j): its gy patch
// (TH + 1) x (TW + 1) (rows i .. i + TH, no flip: tap r = 0 reads row i + 1, r = 1 and r = 2 read row i) is staged once per chunk
// of CK output channels exactly like k_c3_fwd's, a k-step is (channel pair, tap) and the tap picks which of the lane's four
// accumulators the MFMA adds to.  The B operand only depends on (r == 0, s == 0): 4 LDS reads serve the 9 taps.  Weights: k_c3_pack's
// input-gradient layout Wp[(co * 9 + 8 - tap)][ci].  Epilogue: classes (a, 0) and (a, 1) of a lane are neighbours in memory -> one
// 8-byte store per row.
template <int BM_, int WM_, int WN_, int TH_, int TW_, int NIMG_>
struct D2Cfg {
    static constexpr int BM = BM_, WM = WM_, WN = WN_, TH = TH_, TW = TW_, NIMG = NIMG_, CK = 4;
    static constexpr int TPIX = TH * TW, BN = NIMG * TPIX;
    static_assert(WM * WN == 4 && BM == 32 * WM && BN % (32 * WN) == 0, "bad strided-dgrad config");
    static constexpr int FN = BN / 32 / WN;
    static constexpr int PH = TH + 1, PW = TW + 1, IPLANE = PH * PW, PLANE = NIMG * IPLANE;
    static constexpr int LDW = BM + 4, KC = CK * 9, W4 = BM / 4, WROWS = 256 / W4, NW4 = (KC + WROWS - 1) / WROWS;
    static constexpr int X_ELEMS = CK * PLANE, NXL = (X_ELEMS + 255) / 256;
    static constexpr int W_ELEMS = NW4 * WROWS * LDW, XS_ELEMS = NXL * 256, STAGE = W_ELEMS + XS_ELEMS, SMEM_FLOATS = 2 * STAGE;
    static constexpr int NITEMS = NW4 + NXL;
};

template <class Cfg>
__global__ __launch_bounds__(256, 2) void k_c3s2_dgrad(C3Geom g, const float *__restrict__ gy, const float *__restrict__ wp,
                                                       float *__restrict__ gx) {
    // g: N, C = channels of gy (contracted), H x W = the gy map, M = channels of gx, OH x OW = the gx map
    __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int li = lane & 31, lh = lane >> 5;
    unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lb % g.tiles_m; lb /= g.tiles_m;
    const int tx = lb % g.tiles_x; lb /= g.tiles_x;
    const int ty = lb % g.tiles_y;
    const int n = (int)(lb / g.tiles_y) * Cfg::NIMG;
    const int m0 = tm * Cfg::BM, i0 = ty * Cfg::TH, j0 = tx * Cfg::TW;
    const int HW = g.H * g.W;

    const int wcol = (tid % Cfg::W4) * 4, wrow0 = tid / Cfg::W4;
    const int wdst = wrow0 * Cfg::LDW + wcol;
    const unsigned wbyte = (unsigned)(wrow0 * g.Mp + m0 + wcol) * 4u;
    constexpr int kOutOfRange = (int)0x80000000;
    int xbyte[Cfg::NXL];
#pragma unroll
    for (int i = 0; i < Cfg::NXL; ++i) {
        const int e = tid + 256 * i;
        const int cl = e / Cfg::PLANE, rem0 = e - cl * Cfg::PLANE;
        const int img = rem0 / Cfg::IPLANE, rem = rem0 - img * Cfg::IPLANE;
        const int pr = rem / Cfg::PW, pc = rem - pr * Cfg::PW;
        const int gh = i0 + pr, gw = j0 + pc;
        const bool ok = e < Cfg::X_ELEMS && n + img < g.N && gh < g.H && gw < g.W;
        xbyte[i] = ok ? ((img * g.C + cl) * HW + gh * g.W + gw) * 4 : kOutOfRange;
    }
    const int nimg_here = min(Cfg::NIMG, g.N - n);
    const __amdgpu_buffer_rsrc_t srd_x =
        __builtin_amdgcn_make_buffer_rsrc((void *)(gy + (int64_t)n * g.C * HW), 0, nimg_here * g.C * HW * 4, 0x00020000);
    f32x4 rw[Cfg::NW4];
    float rx[Cfg::NXL];
    auto load_item = [&](int k, int c0) {
        if (k < Cfg::NW4)
            rw[k] = ld_sv4(wp + ((int64_t)c0 * 9 + Cfg::WROWS * k) * g.Mp, wbyte);
        else
            rx[k - Cfg::NW4] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, xbyte[k - Cfg::NW4] + c0 * HW * 4, 0, 0));
    };
    auto store_item = [&](int k, float *stage) {
        if (k < Cfg::NW4)
            *reinterpret_cast<f32x4 *>(stage + wdst + Cfg::WROWS * k * Cfg::LDW) = rw[k];
        else
            stage[Cfg::W_ELEMS + tid + 256 * (k - Cfg::NW4)] = rx[k - Cfg::NW4];
    };

    const int a_base = lh * 9 * Cfg::LDW + wm * 32 + li;
    int b_base[Cfg::FN];
#pragma unroll
    for (int fn = 0; fn < Cfg::FN; ++fn) {
        const int t = (wn * Cfg::FN + fn) * 32 + li;
        const int img = t / Cfg::TPIX, tt = t % Cfg::TPIX;
        b_base[fn] = lh * Cfg::PLANE + img * Cfg::IPLANE + (tt / Cfg::TW) * Cfg::PW + tt % Cfg::TW;
    }
    f32x16 acc[4][Cfg::FN];                               // [class 2 a + b][pixel fragment]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][fn][e] = 0.0f;

    const int nch = (g.C + Cfg::CK - 1) / Cfg::CK;
#pragma unroll
    for (int k = 0; k < Cfg::NITEMS; ++k) load_item(k, 0);
#pragma unroll
    for (int k = 0; k < Cfg::NITEMS; ++k) store_item(k, smem);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const float *ws = smem + (ch & 1) * Cfg::STAGE;
        const float *xs = ws + Cfg::W_ELEMS;
        float *other = smem + ((ch + 1) & 1) * Cfg::STAGE;
        const int c_next = min(ch + 1, nch - 1) * Cfg::CK;
#pragma unroll
        for (int k = 0; k < Cfg::NITEMS; ++k) load_item(k, c_next);
#pragma unroll
        for (int p = 0; p < Cfg::CK / 2; ++p) {
            float b[4][Cfg::FN];                           // [2 * (r == 0) + (s == 0)]
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int fn = 0; fn < Cfg::FN; ++fn) b[v][fn] = xs[b_base[fn] + 2 * p * Cfg::PLANE + (v >> 1) * Cfg::PW + (v & 1)];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int tap = 8 - tp, r = tap / 3, q = tap % 3;
                const int cls = 2 * (r != 1) + (q != 1), v = 2 * (r == 0) + (q == 0);
                const float a = ws[a_base + (2 * p * 9 + tp) * Cfg::LDW];
#pragma unroll
                for (int fn = 0; fn < Cfg::FN; ++fn)
                    acc[cls][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[v][fn], acc[cls][fn], 0, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < Cfg::NITEMS; ++k) store_item(k, other);
        __syncthreads();
    }

    // epilogue: lane li = pixel (i, j) of the class grid, rows of D = channels; classes (a, 0) / (a, 1) -> gx[2 i + a][2 j], [2 j + 1]
    const int HWo = g.OH * g.OW;
#pragma unroll
    for (int fn = 0; fn < Cfg::FN; ++fn) {
        const int t = (wn * Cfg::FN + fn) * 32 + li;
        const int img = t / Cfg::TPIX, tt = t % Cfg::TPIX;
        const int i = i0 + tt / Cfg::TW, j = j0 + tt % Cfg::TW;
        float *out = gx + (int64_t)(n + img) * g.M * HWo;
        const bool iok = n + img < g.N;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int h = 2 * i + a, w = 2 * j;
            const bool ok0 = iok && h < g.OH && w < g.OW, ok1 = ok0 && w + 1 < g.OW;
            const bool pair = ok1 && (g.OW % 2 == 0);      // 8-byte aligned pair
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (ci >= g.M) continue;
                float *dst = out + (int64_t)ci * HWo + h * g.OW + w;
                const float v0 = acc[2 * a + 0][fn][e], v1 = acc[2 * a + 1][fn][e];
                if (pair) {
                    *reinterpret_cast<float2 *>(dst) = make_float2(v0, v1);
                } else {
                    if (ok0) dst[0] = v0;
                    if (ok1) dst[1] = v1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------ wgrad
// D[co][ci](tap) += sum_pix gy[co][pix] * x[ci][pix + tap offset]
//
// A "unit" is one TH x TW tile of one image; the block walks `units_per_split` units, contracting over their
// pixels (k of the MFMA = a pair of horizontally adjacent pixels, lanes 0-31 / 32-63).
//
// Staging map (zero per-element index arithmetic): wave `sub` (0..3) owns channels sub + 4*i, i < 16; its 64
// lanes own tile positions lane + 64*g.  One staged element = ONE `buffer_load_dword v, v_pos, s[srd], s_chan offen`
// and ONE `ds_write_b32 v_dst, v offset:imm`:
//   * v_pos (per thread and position group, recomputed per unit) carries the position inside the channel plane, or
//     0x80000000 for zero padding / beyond-the-image lanes: the buffer unit range-checks it against num_records and
//     returns 0 without touching memory, so no select is needed before the LDS write;
//   * s_chan (wave-uniform) is the channel's plane offset, clamped to the tensor's last channel -- rows of channels
//     >= M / C hold duplicates, which only reach accumulator rows / columns that are never stored;
//   * lanes of the last position group that have no position write to the row's padding column.
// The loads and LDS writes are spread between the MFMAs of a unit (sched_group_barrier), so the chip never sees a
// staging-only phase: before this, with the block's two waves per SIMD in lock step, wgrad ran at 74-77 % MFMA
// utilisation against 95 % for the same loop with staging removed (profiles/r01j_*.md).
// S2 = 1: the weight gradient of the 3x3 / STRIDE 2 / pad 1 conv.  TH x TW tiles the gy map (OH x OW); the x patch of a unit is
// (2 TH + 1) x (2 TW + 1), pixel (r, c) of the tile pairs with patch element (2 r + kh, 2 c + kw): no sliding window (consecutive
// pixel pairs share no column), nine LDS reads of x per nine MFMAs instead of six.
template <int TH_, int TW_, bool DB_ = false, int S2_ = 0>
struct W3Cfg {
    static constexpr bool DB = DB_;                             // two LDS stages: one barrier per unit instead of two
    static constexpr int S2 = S2_, SS = S2_ ? 2 : 1;
    static constexpr int TH = TH_, TW = TW_, NPIX = TH * TW;
    static_assert(TW % 2 == 0, "pixel pairs must not straddle rows");
    static constexpr int BMC = 64, BCI = 64;                    // block tile: 64 co x 64 ci, 2 x 2 waves
    static constexpr int PH = S2 ? 2 * TH + 1 : TH + 2, PW = S2 ? 2 * TW + 1 : TW + 2, PHW = PH * PW;
    static_assert(S2 || PHW % 2 == 0, "the patch plane needs a padding column");
    static constexpr int PLANE = PHW + 1 + (PHW % 2);           // odd strides: conflict-free lane = channel reads,
    static constexpr int LDG = NPIX + 1;                        // and one padding column per row
    static_assert(PLANE % 2 == 1 && LDG % 2 == 1, "row strides must be odd");
    static constexpr int G_ELEMS = BMC * LDG, X_ELEMS = BCI * PLANE;
    static constexpr int STAGE = G_ELEMS + X_ELEMS;
    static constexpr int SMEM_FLOATS = (DB ? 2 : 1) * STAGE;
    static constexpr int GP = (NPIX + 63) / 64, XP = (PHW + 63) / 64;      // 64-lane position groups
    static constexpr int NI = 16;                                           // channels per wave
    static constexpr int NITEMS = (GP + XP) * NI;                           // staged elements per thread per unit
    static constexpr int NC = TW / 2, NSTEP = TH * NC;                      // pixel-pair steps per row / per unit
};

// (OH x OW: the gy map -- = H x W for the stride-1 tiles)
template <class Cfg>
__global__ __launch_bounds__(256, 2) void k_c3_wgrad(int N, int C, int H, int W, int M, int OH, int OW, int tiles_x, int tiles_y, int tiles_co,
                                                     int tiles_ci, int units_per_split, const float *__restrict__ x,
                                                     const float *__restrict__ gy, float *__restrict__ part) {
    __shared__ float smem[Cfg::SMEM_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave id, provably uniform
    const int wco = sub >> 1, wci = sub & 1;
    const int li = lane & 31, lh = lane >> 5;
    // Block -> (split, channel tile).  All (co, ci) tiles of one split read the SAME gy / x units, so they are
    // placed on one XCD (block b runs on XCD b % 8) and dispatched back to back: the units are then fetched
    // from HBM once into that XCD's L2 instead of once per XCD.  PMC before this mapping: FETCH_SIZE 4-10x the
    // algorithmic bytes (profiles/r01_traffic.json).  nsplit is a multiple of 8; speed only, never correctness.
    const int tiles = tiles_co * tiles_ci;
    const int xcd = blockIdx.x % kXCDs, j = blockIdx.x / kXCDs;
    const int split = (j / tiles) * kXCDs + xcd, tile = j % tiles;
    const int tci = tile % tiles_ci, tco = tile / tiles_ci;
    const int co0 = tco * Cfg::BMC, ci0 = tci * Cfg::BCI;
    const int HW = H * W, HWg = OH * OW;
    const int units_per_img = tiles_x * tiles_y;
    const int total_units = N * units_per_img;
    const int u0 = min(total_units, split * units_per_split);
    const int u1 = min(total_units, u0 + units_per_split);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;

    // ---- fixed per-thread position descriptors (one small set per 64-lane position group) ----
    constexpr int kOutOfRange = (int)0x80000000;
    int g_fix[Cfg::GP], g_dst[Cfg::GP], x_fix[Cfg::XP], x_dst[Cfg::XP];
    unsigned g_rc[Cfg::GP], x_rc[Cfg::XP];      // (row << 8) | col inside the tile / patch, 0xFFFF: no position
#pragma unroll
    for (int g = 0; g < Cfg::GP; ++g) {
        const int pix = lane + 64 * g, r = pix / Cfg::TW, c = pix % Cfg::TW;
        const bool has = pix < Cfg::NPIX;
        g_fix[g] = 4 * (r * OW + c);                                      // bytes from the tile origin
        g_rc[g] = has ? (unsigned)((r << 8) | c) : 0xFFFFu;
        g_dst[g] = sub * Cfg::LDG + (has ? pix : Cfg::NPIX);              // LDS float index (padding column if none)
    }
#pragma unroll
    for (int g = 0; g < Cfg::XP; ++g) {
        const int q = lane + 64 * g, r = q / Cfg::PW, c = q % Cfg::PW;
        const bool has = q < Cfg::PHW;
        x_fix[g] = 4 * (r * W + c);                                       // bytes from the patch origin (h0-1, w0-1)
        x_rc[g] = has ? (unsigned)((r << 8) | c) : 0xFFFFu;
        x_dst[g] = Cfg::G_ELEMS + sub * Cfg::PLANE + (has ? q : Cfg::PHW);
    }
    // channel plane offsets (bytes, wave-uniform), clamped to the last channel of the tensor
    auto g_chan = [&](int i) { return min(sub + 4 * i, M - 1 - co0) * HWg * 4; };
    auto x_chan = [&](int i) { return min(sub + 4 * i, C - 1 - ci0) * HW * 4; };

    // ---- descriptor of the unit being loaded: two buffer resources + per-group byte offsets ----
    __amdgpu_buffer_rsrc_t srd_g, srd_x;
    int gv[Cfg::GP], xv[Cfg::XP];
    auto describe = [&](int u) {
        const int n = u / units_per_img, rr = u - n * units_per_img;
        const int ty = rr / tiles_x, tx = rr - ty * tiles_x;
        const int h0 = ty * Cfg::TH, w0 = tx * Cfg::TW;
        srd_g = __builtin_amdgcn_make_buffer_rsrc((void *)(gy + ((int64_t)n * M + co0) * HWg), 0, 0x7FFFFFFF, 0x00020000);
        srd_x = __builtin_amdgcn_make_buffer_rsrc((void *)(x + ((int64_t)n * C + ci0) * HW), 0, 0x7FFFFFFF, 0x00020000);
        const int go = 4 * (h0 * OW + w0), xo = 4 * ((Cfg::SS * h0 - 1) * W + (Cfg::SS * w0 - 1));
#pragma unroll
        for (int g = 0; g < Cfg::GP; ++g) {
            const int r = g_rc[g] >> 8, c = g_rc[g] & 255;
            const bool pv = g_rc[g] != 0xFFFFu && h0 + r < OH && w0 + c < OW;
            gv[g] = pv ? go + g_fix[g] : kOutOfRange;
        }
#pragma unroll
        for (int g = 0; g < Cfg::XP; ++g) {
            const int r = x_rc[g] >> 8, c = x_rc[g] & 255;
            const int gh = Cfg::SS * h0 - 1 + r, gw = Cfg::SS * w0 - 1 + c;
            const bool pv = x_rc[g] != 0xFFFFu && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            xv[g] = pv ? xo + x_fix[g] : kOutOfRange;
        }
    };
    // staged element k: position group k / NI (patch groups first), channel slot k % NI
    float st[Cfg::GP + Cfg::XP][Cfg::NI];
    auto load_item = [&](int k) {
        const int grp = k / Cfg::NI, i = k % Cfg::NI;
        if (grp < Cfg::XP)
            st[grp][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, xv[grp], x_chan(i), 0));
        else
            st[grp][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_g, gv[grp - Cfg::XP], g_chan(i), 0));
    };
    auto write_item = [&](int k, float *stage) {
        const int grp = k / Cfg::NI, i = k % Cfg::NI;
        if (grp < Cfg::XP)
            stage[x_dst[grp] + 4 * i * Cfg::PLANE] = st[grp][i];
        else
            stage[g_dst[grp - Cfg::XP] + 4 * i * Cfg::LDG] = st[grp][i];
    };

    const int a_base = (wco * 32 + li) * Cfg::LDG + lh;                                  // gy[co][pix], pix = 2s + lh
    const int b_base = Cfg::G_ELEMS + (wci * 32 + li) * Cfg::PLANE + Cfg::SS * lh;       // x[ci][(S r + kh) * PW + S c + kw]

    // One unit: NSTEP steps of 9 MFMAs.  Sliding window along a row: step c2 needs patch columns 2*c2 + lh + {0,1,2};
    // column +2 of one step is column +0 of the next, so a step reads 1 + 6 new LDS values, one step ahead of the
    // MFMAs that consume them.  Each step also carries its share of the unit's staging: element k's LDS write (two
    // stages: into the other stage; the value was loaded one unit ago) and the buffer load that refills its register.
    auto unit_s2 = [&](const float *cur, float *other, bool stage_writes) {
        float a[2], bs[2][3][3];
        auto rd = [&](int q, int set) {
            const int r = q / Cfg::NC, c2 = q % Cfg::NC;
            a[set] = cur[a_base + r * Cfg::TW + 2 * c2];
            const float *xb = cur + b_base + 2 * r * Cfg::PW + 4 * c2;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) bs[set][kh][kw] = xb[kh * Cfg::PW + kw];
        };
        rd(0, 0);
#pragma unroll
        for (int q = 0; q < Cfg::NSTEP; ++q) {
            const int set = q & 1;
            if (q + 1 < Cfg::NSTEP) rd(q + 1, set ^ 1);
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set], bs[set][t / 3][t % 3], acc[t], 0, 0, 0);
            const int k0 = q * Cfg::NITEMS / Cfg::NSTEP, k1 = (q + 1) * Cfg::NITEMS / Cfg::NSTEP;
#pragma unroll
            for (int k = k0; k < k1; ++k) {
                if (stage_writes) write_item(k, other);
                load_item(k);
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (i < k1 - k0) {
                    if (stage_writes) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
    };
    auto unit_s1 = [&](const float *cur, float *other, bool stage_writes) {
        float a[2], bn[2][3][2], b0[3], b0n[3];
        auto rd = [&](int q, int set) {
            const int r = q / Cfg::NC, c2 = q % Cfg::NC;
            const float *ga = cur + a_base + r * Cfg::TW, *xb = cur + b_base + r * Cfg::PW;
            a[set] = ga[2 * c2];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if (c2 == 0) b0n[kh] = xb[kh * Cfg::PW];
                bn[set][kh][0] = xb[kh * Cfg::PW + 2 * c2 + 1];
                bn[set][kh][1] = xb[kh * Cfg::PW + 2 * c2 + 2];
            }
        };
        rd(0, 0);
#pragma unroll
        for (int q = 0; q < Cfg::NSTEP; ++q) {
            const int set = q & 1;
            if (q % Cfg::NC == 0) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) b0[kh] = b0n[kh];
            }
            if (q + 1 < Cfg::NSTEP) rd(q + 1, set ^ 1);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set], b0[kh], acc[kh * 3 + 0], 0, 0, 0);
                acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set], bn[set][kh][0], acc[kh * 3 + 1], 0, 0, 0);
                acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set], bn[set][kh][1], acc[kh * 3 + 2], 0, 0, 0);
                b0[kh] = bn[set][kh][1];
            }
            const int k0 = q * Cfg::NITEMS / Cfg::NSTEP, k1 = (q + 1) * Cfg::NITEMS / Cfg::NSTEP;
#pragma unroll
            for (int k = k0; k < k1; ++k) {
                if (stage_writes) write_item(k, other);
                load_item(k);
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i < k1 - k0) {
                    if (stage_writes) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
    };

    auto unit = [&](const float *cur, float *other, bool stage_writes) {
        if constexpr (Cfg::S2) unit_s2(cur, other, stage_writes);
        else unit_s1(cur, other, stage_writes);
    };
    if (u0 < u1) {      // (an empty trailing split just writes zeros)
        describe(u0);
#pragma unroll
        for (int k = 0; k < Cfg::NITEMS; ++k) load_item(k);
        if (Cfg::DB) {
            // Two LDS stages, ONE barrier per unit.  During unit u the registers (holding unit u+1) are written to the
            // other stage and refilled with unit u+2; unit indices are clamped to the split's last unit so the tail needs
            // no branches (it re-stages data nobody reads).
#pragma unroll
            for (int k = 0; k < Cfg::NITEMS; ++k) write_item(k, smem);
            describe(min(u0 + 1, u1 - 1));
#pragma unroll
            for (int k = 0; k < Cfg::NITEMS; ++k) load_item(k);
            __syncthreads();
            for (int u = u0; u < u1; ++u) {
                const int cur = (u - u0) & 1;
                describe(min(u + 2, u1 - 1));
                unit(smem + cur * Cfg::STAGE, smem + (cur ^ 1) * Cfg::STAGE, true);
                __syncthreads();
            }
        } else {
            // One stage: barrier, registers (unit u) -> LDS, barrier; the loads of unit u+1 ride inside unit u's MFMAs.
            for (int u = u0; u < u1; ++u) {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < Cfg::NITEMS; ++k) write_item(k, smem);
                __syncthreads();
                describe(min(u + 1, u1 - 1));
                unit(smem, smem, false);
            }
        }
    }
    // partial result, tap-major: part[split][tap][co][ci] -- lanes 0-31 of a store cover 32 consecutive ci (128 bytes).
    // (The [co][ci][tap] order of the final gradient would make every lane's 4-byte store its own 32-byte sector:
    // WRITE_SIZE showed 1.2 GB per launch for 0.15 GB of partials.)  k_split_reduce transposes while it sums.
    float *dst = part + (int64_t)split * M * C * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const int ci = ci0 + wci * 32 + li;
            if (co < M && ci < C) dst[((int64_t)t * M + co) * C + ci] = acc[t][e];
        }
}

// ------------------------------------------------------------------------------ wgrad, <= 3 input channels
// The network stem (3 -> 64 @224x224): all (ci, tap) pairs fit ONE 32-wide fragment column (27 <= 32), the
// contraction runs over 12.8 M pixels and the kernel is bound by streaming gy (3.3 GB) from HBM, not by MFMA.
// Block = 64 co; waves 0/1 own the two 32-channel fragments for the first half of a unit's pixel pairs,
// waves 2/3 for the second half (combined through LDS at the end).  x patch: C planes + one all-zero plane
// that the 5 unused fragment columns read.
struct WSCfg {
    static constexpr int TH = 4, TW = 32, NPIX = TH * TW, PH = TH + 2, PW = TW + 2, IPLANE = PH * PW;
    static constexpr int CMAX = 3, LDG = NPIX + 1;
    static constexpr int G_ELEMS = 64 * LDG, X_ELEMS = (CMAX + 1) * IPLANE;
    static constexpr int NX = (CMAX * IPLANE + 255) / 256;
};

__global__ __launch_bounds__(256) void k_c3_wgrad_smallc(int N, int C, int H, int W, int M, int tiles_x, int tiles_y,
                                                         int units_per_split, const float *__restrict__ x,
                                                         const float *__restrict__ gy, float *__restrict__ part) {
    using Cfg = WSCfg;
    __shared__ float smem[Cfg::G_ELEMS + Cfg::X_ELEMS];
    float *gs = smem, *xs = smem + Cfg::G_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wco = sub & 1, khalf = sub >> 1;
    const int co0 = blockIdx.x * 64;
    const int HW = H * W;
    const int units_per_img = tiles_x * tiles_y;
    const int total_units = N * units_per_img;
    const int u0 = blockIdx.y * units_per_split;
    const int u1 = min(total_units, u0 + units_per_split);
    const int J = C * 9;

    for (int i = tid; i < Cfg::IPLANE; i += 256) xs[Cfg::CMAX * Cfg::IPLANE + i] = 0.0f;     // the zero plane
    // fragment column j = (ci, tap) -> fixed offset into the patch; unused columns read the zero plane
    const int jci = li / 9, jt = li - jci * 9;
    const int joff = li < J ? jci * Cfg::IPLANE + (jt / 3) * Cfg::PW + (jt % 3) : Cfg::CMAX * Cfg::IPLANE;
    const int a_base = (wco * 32 + li) * Cfg::LDG + lh;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

    // gy staging map: wave `sub` owns channels sub + 4*i (i < 16), lanes own pixels lane and lane + 64
    for (int u = u0; u < u1; ++u) {
        const int n = u / units_per_img, rr = u - n * units_per_img;
        const int ty = rr / tiles_x, tx = rr - ty * tiles_x;
        const int h0 = ty * Cfg::TH, w0 = tx * Cfg::TW;
        __syncthreads();                                     // previous unit fully consumed
        const float *gimg = gy + ((int64_t)n * M + co0) * HW + h0 * W + w0;
        float rg[2][16];
        bool pv[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int pix = lane + 64 * g, r = pix / Cfg::TW, c = pix % Cfg::TW;
            pv[g] = h0 + r < H && w0 + c < W;
            const unsigned off = 4u * (unsigned)(sub * HW + r * W + c);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool cv = co0 + sub + 4 * i < M;
                const float *base = gimg + (int64_t)(cv ? 4 * i : 0) * HW;
                rg[g][i] = ld_sv(base, (pv[g] && cv) ? off : 0u);
            }
        }
        const float *ximg = x + (int64_t)n * C * HW;
        float rx[Cfg::NX];
        bool xv[Cfg::NX];
#pragma unroll
        for (int i = 0; i < Cfg::NX; ++i) {
            const int e = tid + 256 * i;
            const int ci = e / Cfg::IPLANE, rem = e - ci * Cfg::IPLANE;
            const int gh = h0 - 1 + rem / Cfg::PW, gw = w0 - 1 + rem % Cfg::PW;
            xv[i] = ci < C && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            rx[i] = ximg[xv[i] ? (int64_t)ci * HW + gh * W + gw : 0];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                gs[(sub + 4 * i) * Cfg::LDG + lane + 64 * g] = (pv[g] && co0 + sub + 4 * i < M) ? rg[g][i] : 0.0f;
#pragma unroll
        for (int i = 0; i < Cfg::NX; ++i) {
            const int e = tid + 256 * i;
            if (e < Cfg::CMAX * Cfg::IPLANE) xs[e] = xv[i] ? rx[i] : 0.0f;
        }
        __syncthreads();
        // this wave's half of the pixel pairs: rows khalf*2, khalf*2 + 1
#pragma unroll
        for (int r2 = 0; r2 < Cfg::TH / 2; ++r2) {
            const int r = khalf * (Cfg::TH / 2) + r2;
#pragma unroll
            for (int c2 = 0; c2 < Cfg::TW / 2; ++c2) {
                const float a = gs[a_base + r * Cfg::TW + 2 * c2];
                const float b = xs[joff + r * Cfg::PW + 2 * c2 + lh];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
    }
    // combine the two pixel halves through LDS, then write part[split][co][j]
    __syncthreads();
    float *red = smem;                                       // [2 fragments][16][64]
    if (khalf == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(wco * 16 + e) * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (khalf == 0) {
        float *dst = part + (int64_t)blockIdx.y * M * J;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = acc[e] + red[(wco * 16 + e) * 64 + lane];
            const int co = co0 + wco * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (co < M && li < J) dst[(int64_t)co * J + li] = v;
        }
    }
}

// ------------------------------------------------------------------------------ dispatch
//                 BM  TH  TW  WM WN CK MINW
using CfgM128 = C3Cfg<128, 4, 32, 2, 2, 4, 3>;     // >= 128 output channels, wide images
using CfgM64 = C3Cfg<64, 8, 32, 1, 4, 4, 3>;       // <= 64 output channels (VGG 224x224 layers)
using CfgS16 = C3Cfg<128, 14, 16, 4, 1, 4, 2>;     // 14x14 (and <= 16 wide) feature maps: whole image, 7 fragments
using CfgD128 = C3Cfg<128, 4, 56, 4, 1, 4, 2>;     // 56 / 112 wide maps: 4 x 56 = 7 fragments per wave, zero tile waste
using CfgD64 = C3Cfg<64, 8, 56, 2, 2, 4, 2>;       // same for <= 64 output channels
using CfgV14 = C3Cfg<128, 14, 14, 4, 1, 4, 2, 3, 16>;   // 14 x 14 maps: 16 virtual rows = 224 pixels, zero tile waste (with ksplit = 2)
using CfgV7 = C3Cfg<128, 7, 7, 4, 1, 4, 2, 6, 32>;     // 7 x 7 maps (ResNet layer4): 32 virtual rows over up to 6 images
using CfgP28 = C3Cfg<128, 4, 28, 4, 1, 4, 2, 2>;   // 28-wide maps: a 4 x 28 strip of TWO images = 7 fragments, zero tile waste

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }
// packed-weight workspace: [roundup(C_read, 4) * 9 (+ 16 rows of slack the last float4 staging pass may read)][roundup(M, 128)] floats
inline size_t pack_floats(int c_read, int m) { return ((size_t)pad_to(c_read, 4) * 9 + 16) * pad_to(m, 128); }
// ... followed by the liveness words of the inference path: live[Mp] column flags, 4 words (live input channels, skipped-tile
// counter, pad), one flag per 4-channel input chunk
inline size_t live_words(int c_read, int m) { return (size_t)pad_to(m, 128) + 4 + pad_to(c_read, 4) / 4; }
inline size_t pack_bytes(int c_read, int m) { return (pack_floats(c_read, m) + live_words(c_read, m)) * sizeof(float); }

// stats != nullptr: forward with fused BatchNorm statistics.  tiles_out (optional) receives the number of pixel tiles;
// dry: only compute it.
}  // namespace
extern "C" int cpg_conv3x3_wino_ok(int N, int c_read, int m, int H, int W);
extern "C" int cpg_conv3x3_stem_ok(int N, int C, int K, int H, int W);
extern "C" int cpg_conv3x3_stem_tiles(int N, int C, int K, int H, int W);
extern "C" int cpg_conv3x3_stem_run(int N, int C, int K, int H, int W, const float *x, const float *w, const float *pm, float thr,
                                    const float *bias, float *y, float *stats, hipStream_t stream);
extern "C" size_t cpg_conv3x3_wino_pack_bytes(int c_read, int m);
extern "C" size_t cpg_conv3x3_wino_tail_bytes(int N, int c_read, int m, int H, int W);
extern "C" int cpg_conv3x3_wino_tiles(int N, int c_read, int m, int H, int W);
extern "C" int cpg_conv3x3_wino_eval_ok(int N, int c_read, int m, int H, int W);
extern "C" int cpg_conv3x3_wino_run_bn_eval(int N, int C, int K, int H, int W, const float *x, const float *w, const float *pm, float thr,
                                            const float *bias, const float *gamma, const float *beta, const float *mean, const float *var,
                                            float eps, int relu, int *live, size_t live_words, float *y, void *ws, size_t ws_bytes,
                                            hipStream_t stream);
extern "C" int cpg_conv3x3_wino_run(int dgrad, int N, int c_read, int m, int H, int W, int K, int C, const float *x, const float *w,
                                    const float *pm, float thr, const float *bias, float *y, float *stats, void *ws, size_t ws_bytes,
                                    hipStream_t stream);
namespace {

template <class Cfg>
int launch_fwd(C3Geom g, const float *x, const float *wp, const float *bias, float *y, hipStream_t stream, const char *what,
               float *stats = nullptr, int *tiles_out = nullptr, bool dry = false, const C3BnEval *bnp = nullptr,
               const C3BnBwd *bbp = nullptr) {
    const C3BnEval bn = bnp ? *bnp : C3BnEval{nullptr, nullptr, nullptr, nullptr, 0.0f, 0, nullptr};
    const C3BnBwd bb = bbp ? *bbp : C3BnBwd{nullptr, nullptr, nullptr, nullptr, nullptr};
    g.tiles_x = (g.OW + Cfg::TW - 1) / Cfg::TW;           // (tiles cover the OUTPUT map; = the input map for the stride-1 tiles)
    g.tiles_y = (g.OH + Cfg::TH - 1) / Cfg::TH;
    g.tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM;
    const int64_t blocks = Cfg::VROWS ? (int64_t)(((int64_t)g.N * Cfg::TH + Cfg::VROWS - 1) / Cfg::VROWS) * g.tiles_m
                                      : (int64_t)((g.N + Cfg::NIMG - 1) / Cfg::NIMG) * g.tiles_x * g.tiles_y * g.tiles_m;
    if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "conv3x3: grid too large");
    if (tiles_out) *tiles_out = g.ksplit > 1 ? 0 : (int)(blocks / g.tiles_m);      // no fused statistics on split tiles
    if (dry) return CPG_OK;
    if (g.ksplit > 1) {
        if (stats != nullptr || bnp != nullptr || bbp != nullptr) return fail(CPG_E_UNSUPPORTED, "conv3x3: no fused epilogue on channel-split tiles");
        hipError_t e = hipMemsetAsync(y, 0, (size_t)g.N * g.M * g.OH * g.OW * sizeof(float), stream);
        if (e != hipSuccess) return hip_status(e, what);
        if (g.dgrad)
            hipLaunchKernelGGL((k_c3_fwd<Cfg, true, false, true>), dim3((unsigned)(blocks * g.ksplit)), dim3(256), 0, stream, g, x, wp, bias, y, nullptr, bn, bb);
        else
            hipLaunchKernelGGL((k_c3_fwd<Cfg, false, false, true>), dim3((unsigned)(blocks * g.ksplit)), dim3(256), 0, stream, g, x, wp, bias, y, nullptr, bn, bb);
        CPG_CHECK_LAUNCH(what);
        return CPG_OK;
    }
    if (g.dgrad && bbp != nullptr) {
        if (stats == nullptr) return fail(CPG_E_INVALID, "%s: no buffer for the BatchNorm-backward partial sums", what);
        hipLaunchKernelGGL((k_c3_fwd<Cfg, true, false, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, stats, bn, bb);
    } else if (g.dgrad)
        hipLaunchKernelGGL((k_c3_fwd<Cfg, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, nullptr, bn, bb);
    else if (stats != nullptr)
        hipLaunchKernelGGL((k_c3_fwd<Cfg, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, stats, bn, bb);
    else
        hipLaunchKernelGGL((k_c3_fwd<Cfg, false>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, nullptr, bn, bb);
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}

// c_read / m: channels contracted over / produced.  w is the layer's [K][C][3][3] weight.
int run_fwd(bool dgrad, int N, int c_read, int m, int H, int W, int K, int C, const float *x, const float *w, const float *pm,
            float thr, const float *bias, float *y, void *ws, size_t ws_bytes, hipStream_t stream, float *stats = nullptr,
            int *tiles_out = nullptr, bool dry = false, const C3BnEval *bn = nullptr, const C3BnBwd *bb = nullptr) {
    const char *what = dgrad ? "cpg_conv2d_dgrad(3x3)" : "cpg_conv2d_fwd(3x3)";
    // Winograd F(2x2, 3x3) (conv3x3_wino.hip) takes the training forward (with or without the BatchNorm statistics) and the plain
    // input gradient of every even-sized map with >= 16 channels on both sides: 2.25x fewer MFMAs, 1.36-1.47x the speed of the
    // direct kernels below.  The inference epilogues (eval BatchNorm, dead-channel skip) and BRED stay on the direct kernels.
    if (bn == nullptr && bb == nullptr && cpg_conv3x3_wino_ok(N, c_read, m, H, W)) {
        if (tiles_out) *tiles_out = cpg_conv3x3_wino_tiles(N, c_read, m, H, W);
        if (dry) return CPG_OK;
        return cpg_conv3x3_wino_run(dgrad ? 1 : 0, N, c_read, m, H, W, K, C, x, w, pm, thr, bias, y, stats, ws, ws_bytes, stream);
    }
    if (!dry && cpg::pack_query()) return CPG_OK;       // (cpg_conv2d_pack's query: only the Winograd route above records a job)
    // ... and the inference forward with the eval-mode BatchNorm epilogue and the dead-channel skip (k_wg1<.., BNE>).  Workspace layout:
    // [the direct kernels' packed-weight region (unused) | liveness words, where cpg_conv3x3_fwd_bn_eval looks for them | U]
    if (bn != nullptr && bb == nullptr && !dgrad && stats == nullptr && !dry && cpg_conv3x3_wino_eval_ok(N, c_read, m, H, W)) {
        const size_t off = (pack_bytes(c_read, m) + 15) / 16 * 16;
        if (ws == nullptr || ws_bytes < off) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, off);
        int *live = bn->live != nullptr ? reinterpret_cast<int *>((float *)ws + pack_floats(c_read, m)) : nullptr;
        return cpg_conv3x3_wino_run_bn_eval(N, c_read, m, H, W, x, w, pm, thr, bias, bn->gamma, bn->beta, bn->mean, bn->var, bn->eps, bn->relu,
                                            live, live_words(c_read, m), y, (char *)ws + off, ws_bytes - off, stream);
    }
    // the <= 3-channel stem (conv3x3_stem.hip: one persistent wave per tile, weights in registers, HBM-bound)
    if (!dgrad && bn == nullptr && bb == nullptr && cpg_conv3x3_stem_ok(N, c_read, m, H, W)) {
        if (tiles_out) *tiles_out = cpg_conv3x3_stem_tiles(N, c_read, m, H, W);
        if (dry) return CPG_OK;
        return cpg_conv3x3_stem_run(N, c_read, m, H, W, x, w, pm, thr, bias, y, stats, stream);
    }
    float *wp = (float *)ws;
    const int rows_c = pad_to(c_read, 4), Mp = pad_to(m, 128);
    if (!dry) {
        const size_t need = pack_bytes(c_read, m);
        if (ws == nullptr || ws_bytes < need) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
        CPG_REQUIRE((((uintptr_t)ws) & 15) == 0, "%s: workspace must be 16-byte aligned", what);
        int *live = nullptr;
        if (bn != nullptr && bn->live != nullptr) {          // (bn->live is only a request flag here; the words live in the workspace)
            live = reinterpret_cast<int *>(wp + pack_floats(c_read, m));
            hipError_t e = hipMemsetAsync(live, 0, live_words(c_read, m) * sizeof(int), stream);
            if (e != hipSuccess) return hip_status(e, what);
        }
        hipLaunchKernelGGL(k_c3_pack, dim3(stream_grid((int64_t)rows_c * 9 * Mp, 256)), dim3(256), 0, stream, w, pm, thr, wp, K, C,
                           rows_c, Mp, dgrad ? 1 : 0, live);
    }
    C3BnEval bn_local;
    if (bn != nullptr) {
        bn_local = *bn;
        bn_local.live = (bn->live != nullptr && !dry) ? reinterpret_cast<int *>(wp + pack_floats(c_read, m)) : nullptr;
        bn = &bn_local;
    }
    C3Geom g{N, c_read, H, W, m, Mp, 0, 0, 0, H, W, dgrad ? 1 : 0, 1};
    if (const int force = opt(OPT_C3_FORCE); force != OPT_UNSET) {        // A/B experiments only (tools/conv_bench.py --ab)
        switch (force) {
            case 0: return launch_fwd<CfgM128>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
            case 1: return launch_fwd<CfgM64>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
            case 2: return launch_fwd<CfgS16>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
            case 3: return launch_fwd<CfgD128>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
            case 4: return launch_fwd<CfgD64>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
            case 7: return launch_fwd<CfgV14>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
            case 8: g.ksplit = 2; return launch_fwd<CfgV14>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
            default: if (c_read % 4 == 0) return launch_fwd<CfgP28>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
        }
    }
    if (W == 7 && H == 7 && c_read % 4 == 0) return launch_fwd<CfgV7>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
    // 14 x 14 maps: the 14 x 16 single-image tile wastes 1/8 of its MFMAs on two padding columns; the zero-waste virtual-row
    // tile alone measured the same, because its N*14/16 tiles put 3.5 block-equivalents on each CU, which rounds up to 4
    // (at batch 256 the layer is too small for 256 CUs).  Halving the blocks (two per tile, each half of the channel chunks,
    // atomically added into a zeroed y) makes it 7 half-blocks per CU.
    if (W == 14 && H == 14 && m > 64 && c_read % 8 == 0 && bn == nullptr && !opt_on(OPT_NO_V14)) {
        // ... when that balances: per-CU MFMA time in block-equivalents of either tiling (the split pays a memset, atomics
        // and a second prologue; at 256 channels and batch 256 -- 3.5 half-blocks per CU -- it measured no gain)
        const int tm = (m + 127) / 128;
        const double t_single = std::ceil((double)N * tm / kCUs) * (16.0 / 14.0);
        const double t_split = std::ceil(2.0 * (((int64_t)N * 14 + 15) / 16) * tm / kCUs) * 0.5 * 1.04;
        if (t_split < 0.9 * t_single) {
            g.ksplit = 2;
            return launch_fwd<CfgV14>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
        }
    }
    if (W <= 16 && H <= 16 && m > 64) return launch_fwd<CfgS16>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
    if (W == 28 && H % 4 == 0 && m > 64 && c_read % 4 == 0)
        return launch_fwd<CfgP28>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
    // 56, 112, 168 ...: a 32-wide tile would waste 12.5 % of the MFMAs.  The 64-channel 8 x 56 tile (2 x 2 waves) measured
    // 1-2 % faster than the 128-channel 4 x 56 tile (4 x 1 waves) on every 56- and 112-wide VGG layer, also for m > 64
    // (interleaved in-process A/B, tools/conv_bench.py --ab CPG_C3_FORCE=3,4).
    // (224-wide maps divide by 32 too; the 8 x 56 tile measured 0.8 % faster there as well -- except for the HBM-bound 3-channel
    // stem, which prefers the 8 x 32 tile by 9 %)
    if (W % 56 == 0 && (W % 32 != 0 || c_read >= 16)) return launch_fwd<CfgD64>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
    if (m <= 64) return launch_fwd<CfgM64>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
    return launch_fwd<CfgM128>(g, x, wp, bias, y, stream, what, stats, tiles_out, dry, bn, bb);
}

}  // namespace

extern "C" int cpg_conv3x3_supported(const cpg_conv_desc *d) {
    if (cpg::opt_on(cpg::OPT_DISABLE_CONV3X3)) return 0;
    return d->R == 3 && d->S == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1 &&
           d->dil_h == 1 && d->dil_w == 1 && d->groups == 1 && d->N > 0 && d->C > 0 && d->K > 0 && d->H > 0 && d->W > 0 &&
           // staging uses 32-bit byte offsets inside a tile's (up to two) images
           (int64_t)d->C * d->H * d->W < (1ll << 28) && (int64_t)d->K * d->H * d->W < (1ll << 28) &&
           (int64_t)d->H * d->W <= (1ll << 22);       // wgrad: 64 channel planes addressed with a 31-bit byte offset
}

size_t cpg_conv3x3_pack_workspace(const cpg_conv_desc *d) {
    return std::max(pack_bytes(d->C, d->K), pack_bytes(d->K, d->C)) + 16 +
           std::max(cpg_conv3x3_wino_pack_bytes(d->C, d->K), cpg_conv3x3_wino_pack_bytes(d->K, d->C)) + 256 +
           // (the partial outputs of a Winograd launch's tail pieces, behind its packed filter: conv3x3_wino.hip, k_wg3<.., SPLIT>)
           std::max(cpg_conv3x3_wino_tail_bytes(d->N, d->C, d->K, d->H, d->W), cpg_conv3x3_wino_tail_bytes(d->N, d->K, d->C, d->H, d->W));
}

int cpg_conv3x3_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                    float *y, void *ws, size_t ws_bytes, hipStream_t stream) {
    CPG_REQUIRE(x && w && y, "cpg_conv2d_fwd: null pointer");
    return run_fwd(false, d->N, d->C, d->K, d->H, d->W, d->K, d->C, x, w, pm, thr, bias, y, ws, ws_bytes, stream);
}

// forward that also writes the per-(channel, pixel tile) BatchNorm partial sums; tiles = cpg_conv3x3_bnstats_tiles(d)
int cpg_conv3x3_bnstats_tiles(const cpg_conv_desc *d) {
    int tiles = 0;
    if (run_fwd(false, d->N, d->C, d->K, d->H, d->W, d->K, d->C, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, 0, nullptr,
                nullptr, &tiles, true) != CPG_OK)
        return 0;
    return tiles;
}
int cpg_conv3x3_fwd_bnstats(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                            float *y, float *stats, void *ws, size_t ws_bytes, hipStream_t stream) {
    CPG_REQUIRE(x && w && y && stats, "cpg_conv2d_fwd_bnstats: null pointer");
    return run_fwd(false, d->N, d->C, d->K, d->H, d->W, d->K, d->C, x, w, pm, thr, bias, y, ws, ws_bytes, stream, stats);
}

// forward with the inference-mode BatchNorm (+ ReLU) that follows the conv folded into the epilogue (Manager.validate's path)
int cpg_conv3x3_fwd_bn_eval(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                            const float *gamma, const float *beta, const float *mean, const float *var, float eps, int relu, float *y,
                            int32_t *skip_stats, void *ws, size_t ws_bytes, hipStream_t stream) {
    CPG_REQUIRE(x && w && y && gamma && beta && mean && var, "cpg_conv2d_fwd_bn_eval: null pointer");
    static int dummy;
    const bool skip = !opt_on(OPT_NO_DEAD_SKIP);
    const C3BnEval bn{gamma, beta, mean, var, eps, relu, skip ? &dummy : nullptr};
    int rc = run_fwd(false, d->N, d->C, d->K, d->H, d->W, d->K, d->C, x, w, pm, thr, bias, y, ws, ws_bytes, stream, nullptr, nullptr, false,
                     &bn);
    if (rc == CPG_OK && skip_stats != nullptr) {
        // {1 + last live input channel, output tiles skipped}: device-to-device copy of the two words behind live[Mp]
        if (skip) {
            const int *live = reinterpret_cast<const int *>((const float *)ws + pack_floats(d->C, d->K));
            hipError_t e = hipMemcpyAsync(skip_stats, live + pad_to(d->K, 128), 2 * sizeof(int), hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return hip_status(e, "cpg_conv2d_fwd_bn_eval");
        } else {
            hipError_t e = hipMemsetAsync(skip_stats, 0, 2 * sizeof(int), stream);
            if (e != hipSuccess) return hip_status(e, "cpg_conv2d_fwd_bn_eval");
        }
    }
    return rc;
}

int cpg_conv3x3_dgrad(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, float *gx, void *ws,
                      size_t ws_bytes, hipStream_t stream) {
    CPG_REQUIRE(gy && w && gx, "cpg_conv2d_dgrad: null pointer");
    // reads gy (K channels), produces gx (C channels)
    return run_fwd(true, d->N, d->K, d->C, d->H, d->W, d->K, d->C, gy, w, pm, thr, nullptr, gx, ws, ws_bytes, stream);
}

// input gradient whose epilogue also does the BatchNorm-backward reduction of the layer below (see C3BnBwd).  tiles = 0: this
// shape has no such path (channel-split 14 x 14 tiles).
int cpg_conv3x3_dgrad_bnbwd_tiles(const cpg_conv_desc *d) {
    int tiles = 0;
    const C3BnBwd probe{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (run_fwd(true, d->N, d->K, d->C, d->H, d->W, d->K, d->C, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                &tiles, true, nullptr, &probe) != CPG_OK)
        return 0;
    return tiles;
}
int cpg_conv3x3_dgrad_bnbwd(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, const float *ypre,
                            const float *gamma, const float *beta, const float *mean, const float *invstd, float *gx, float *partials,
                            void *ws, size_t ws_bytes, hipStream_t stream) {
    CPG_REQUIRE(gy && w && gx && ypre && gamma && beta && mean && invstd && partials, "cpg_conv2d_dgrad_bnbwd: null pointer");
    const C3BnBwd bb{ypre, gamma, beta, mean, invstd};
    return run_fwd(true, d->N, d->K, d->C, d->H, d->W, d->K, d->C, gy, w, pm, thr, nullptr, gx, ws, ws_bytes, stream, partials, nullptr, false,
                   nullptr, &bb);
}

// ---- wgrad host side -------------------------------------------------------------------------
namespace {
struct W3Plan {
    int tiles_x, tiles_y, tiles_co, tiles_ci, nsplit, units_per_split;
    size_t ws_bytes;
};
template <class Cfg>
W3Plan w3_plan(const cpg_conv_desc *d) {
    W3Plan p;
    const int OH = (d->H + 2 * d->pad_h - 3) / d->stride_h + 1, OW = (d->W + 2 * d->pad_w - 3) / d->stride_w + 1;   // the gy map
    p.tiles_x = (OW + Cfg::TW - 1) / Cfg::TW;
    p.tiles_y = (OH + Cfg::TH - 1) / Cfg::TH;
    p.tiles_co = (d->K + Cfg::BMC - 1) / Cfg::BMC;
    p.tiles_ci = (d->C + Cfg::BCI - 1) / Cfg::BCI;
    const int64_t units = (int64_t)d->N * p.tiles_x * p.tiles_y;
    const int64_t tiles = (int64_t)p.tiles_co * p.tiles_ci;
    // split blocks per CU (each split writes a full set of partial sums): 2 instead of round 2's 4 -- SphereNet-20 20.96 -> 20.78,
    // ResNet-50 73.83 -> 73.49 ms per step (A/B through CPG_C3W_BPC; 8: 21.45 / 74.11)
    // (until round 4 the shared-chip hint of a multi-GPU rank doubled this; measured beside RCCL's kernels -- a world-1 group, every hook
    // and message of the N > 1 path -- 4 blocks per CU cost ResNet-50 1.4 ms and SphereNet-20 0.9 ms per step, more than the whole
    // all-reduce of their gradients lasts: profiles/r04_ab_shared_chip_plans.txt)
    const int bpc = std::max(1, opt_or(OPT_C3W_BPC, 2));
    int64_t want = ((int64_t)bpc * kCUs + tiles - 1) / tiles;
    if (want > units) want = units;
    if (want < 1) want = 1;
    if (want > 4096) want = 4096;
    want = (want + kXCDs - 1) / kXCDs * kXCDs;                 // one split group per XCD (see k_c3_wgrad)
    p.units_per_split = (int)((units + want - 1) / want);
    p.nsplit = (int)want;                                      // trailing splits may be empty: they write zeros
    p.ws_bytes = (size_t)p.nsplit * d->K * d->C * 9 * sizeof(float);
    return p;
}
using W3Wide = W3Cfg<2, 28>;      // 112 / 224 wide feature maps: long contiguous rows, 2 x 64-lane patch groups exactly
using W3Mid = W3Cfg<4, 14, true>;  // 14 / 28 / 56 wide feature maps: zero column waste; two stages (2 x 77 KB per CU)
using W3Nar = W3Cfg<4, 16>;       // everything else that is narrow
using W3Sev = W3Cfg<7, 8, true>;   // 7 x 7 maps: one unit = one image
using W3Tiny = W3Cfg<2, 14, true>; // 14-wide maps whose height is not a multiple of 4 (14 x 14: zero row waste)
inline int w3_pick(const cpg_conv_desc *d) {
    if (const int pick = opt(OPT_W3_PICK); pick != OPT_UNSET) return pick;
    if (d->W <= 8 && d->H <= 7) return 4;
    if (d->W % 28 == 0 && d->W >= 112) return 0;
    if (d->W % 14 == 0) return (d->H % 4 != 0 && d->H % 2 == 0) ? 3 : 1;
    return 2;
}
}  // namespace

extern "C" int cpg_conv3x3_wino_wgrad_ok(const cpg_conv_desc *d);
extern "C" size_t cpg_conv3x3_wino_wgrad_workspace(const cpg_conv_desc *d);
extern "C" int cpg_conv3x3_wino_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr,
                                      float *gw, float *gpm, void *ws, size_t ws_bytes, hipStream_t stream);

size_t cpg_conv3x3_wgrad_workspace(const cpg_conv_desc *d) {
    if (cpg_conv3x3_wino_wgrad_ok(d)) return cpg_conv3x3_wino_wgrad_workspace(d);
    if (d->C <= 3) {            // stem kernel: same formula as ws_plan() below
        const int tiles = ((d->W + 31) / 32) * ((d->H + 3) / 4);
        const int64_t units = (int64_t)d->N * tiles;
        const int blocks_co = (d->K + 63) / 64;
        int64_t want = (8 * kCUs + blocks_co - 1) / blocks_co;
        if (want > units) want = units;
        if (want < 1) want = 1;
        const int64_t per = (units + want - 1) / want;
        return (size_t)((units + per - 1) / per) * d->K * d->C * 9 * sizeof(float);
    }
    switch (w3_pick(d)) {
        case 1: return w3_plan<W3Mid>(d).ws_bytes;
        case 2: return w3_plan<W3Nar>(d).ws_bytes;
        case 3: return w3_plan<W3Tiny>(d).ws_bytes;
        case 4: return w3_plan<W3Sev>(d).ws_bytes;
        default: return w3_plan<W3Wide>(d).ws_bytes;
    }
}

template <class Cfg>
static int w3_launch(const cpg_conv_desc *d, const float *x, const float *gy, const Epilogue &ep, void *ws, size_t ws_bytes,
                     hipStream_t stream) {
    const W3Plan p = w3_plan<Cfg>(d);
    if (ws_bytes < p.ws_bytes) return fail(CPG_E_WORKSPACE, "cpg_conv2d_wgrad(3x3): workspace %zu < %zu bytes", ws_bytes, p.ws_bytes);
    const int OH = (d->H + 2 * d->pad_h - 3) / d->stride_h + 1, OW = (d->W + 2 * d->pad_w - 3) / d->stride_w + 1;
    hipLaunchKernelGGL(k_c3_wgrad<Cfg>, dim3((unsigned)(p.tiles_co * p.tiles_ci * p.nsplit)), dim3(256), 0, stream, d->N, d->C, d->H,
                       d->W, d->K, OH, OW, p.tiles_x, p.tiles_y, p.tiles_co, p.tiles_ci, p.units_per_split, x, gy, (float *)ws);
    const int64_t out_elems = (int64_t)d->K * d->C * 9;
    launch_split_reduce((const float *)ws, p.nsplit, out_elems, (int64_t)d->K * d->C, ep, stream);
    CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(3x3)");
    return CPG_OK;
}

namespace {
struct WSPlan {
    int tiles_x, tiles_y, blocks_co, nsplit, units_per_split;
    size_t ws_bytes;
};
WSPlan ws_plan(const cpg_conv_desc *d) {
    WSPlan p;
    p.tiles_x = (d->W + WSCfg::TW - 1) / WSCfg::TW;
    p.tiles_y = (d->H + WSCfg::TH - 1) / WSCfg::TH;
    p.blocks_co = (d->K + 63) / 64;
    const int64_t units = (int64_t)d->N * p.tiles_x * p.tiles_y;
    int64_t want = (8 * kCUs + p.blocks_co - 1) / p.blocks_co;     // HBM-streaming kernel: ~8 blocks per CU
    if (want > units) want = units;
    if (want < 1) want = 1;
    p.units_per_split = (int)((units + want - 1) / want);
    p.nsplit = (int)((units + p.units_per_split - 1) / p.units_per_split);
    p.ws_bytes = (size_t)p.nsplit * d->K * d->C * 9 * sizeof(float);
    return p;
}
}  // namespace

int cpg_conv3x3_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr,
                      float *gw, float *gpm, void *ws, size_t ws_bytes, hipStream_t stream) {
    // Winograd F(2x2, 3x3) weight gradient (conv3x3_wino_wgrad.hip): maps 14 or a multiple of 28 wide, channel counts multiples of 32
    if (cpg_conv3x3_wino_wgrad_ok(d)) return cpg_conv3x3_wino_wgrad(d, x, gy, w, pm, thr, gw, gpm, ws, ws_bytes, stream);
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    if (d->C <= WSCfg::CMAX) {
        const WSPlan p = ws_plan(d);
        if (ws_bytes < p.ws_bytes) return fail(CPG_E_WORKSPACE, "cpg_conv2d_wgrad(3x3 stem): workspace %zu < %zu bytes", ws_bytes, p.ws_bytes);
        hipLaunchKernelGGL(k_c3_wgrad_smallc, dim3((unsigned)p.blocks_co, (unsigned)p.nsplit), dim3(256), 0, stream, d->N, d->C, d->H,
                           d->W, d->K, p.tiles_x, p.tiles_y, p.units_per_split, x, gy, (float *)ws);
        const int64_t out_elems = (int64_t)d->K * d->C * 9;
        launch_split_reduce((const float *)ws, p.nsplit, out_elems, 0, ep, stream);
        CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(3x3 stem)");
        return CPG_OK;
    }
    switch (w3_pick(d)) {
        case 1: return w3_launch<W3Mid>(d, x, gy, ep, ws, ws_bytes, stream);
        case 2: return w3_launch<W3Nar>(d, x, gy, ep, ws, ws_bytes, stream);
        case 3: return w3_launch<W3Tiny>(d, x, gy, ep, ws, ws_bytes, stream);
        case 4: return w3_launch<W3Sev>(d, x, gy, ep, ws, ws_bytes, stream);
        default: return w3_launch<W3Wide>(d, x, gy, ep, ws, ws_bytes, stream);
    }
}

// ---- the 3x3 / stride 2 / pad 1 class (ResNet: conv2 of the first block of layer2-4; SphereNet: conv{2,3,4}_1) -------------------
namespace {
using S2P28 = C3Cfg<128, 4, 28, 4, 1, 4, 2, 2, 0, 1>;   // 28-wide outputs: a 4 x 28 strip of two images, 7 fragments, zero waste
using S2S16 = C3Cfg<128, 14, 16, 4, 1, 4, 2, 1, 0, 1>;  // <= 16-wide outputs (14 x 14): one image
using S2M8 = C3Cfg<128, 8, 8, 4, 1, 4, 2, 2, 0, 1>;     // <= 8-wide outputs (7 x 7): two images of 8 x 8 = 4 fragments (four would not leave LDS for a second block)
using S2G128 = C3Cfg<128, 4, 32, 2, 2, 4, 2, 1, 0, 1>;  // anything else
using S2G64 = C3Cfg<64, 8, 32, 1, 4, 4, 2, 1, 0, 1>;    // ... with <= 64 output channels

template <class Cfg>
int launch_fwd_s2(C3Geom g, const float *x, const float *wp, const float *bias, float *y, hipStream_t stream, float *stats,
                  int *tiles_out, bool dry, const C3BnEval *bnp) {
    const C3BnEval bn = bnp ? *bnp : C3BnEval{nullptr, nullptr, nullptr, nullptr, 0.0f, 0, nullptr};
    const C3BnBwd bb{nullptr, nullptr, nullptr, nullptr, nullptr};
    g.tiles_x = (g.OW + Cfg::TW - 1) / Cfg::TW;
    g.tiles_y = (g.OH + Cfg::TH - 1) / Cfg::TH;
    g.tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM;
    const int64_t blocks = (int64_t)((g.N + Cfg::NIMG - 1) / Cfg::NIMG) * g.tiles_x * g.tiles_y * g.tiles_m;
    if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "conv3x3 s2: grid too large");
    if (tiles_out) *tiles_out = (int)(blocks / g.tiles_m);
    if (dry) return CPG_OK;
    if (stats != nullptr)
        hipLaunchKernelGGL((k_c3_fwd<Cfg, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, stats, bn, bb);
    else
        hipLaunchKernelGGL((k_c3_fwd<Cfg, false>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, wp, bias, y, nullptr, bn, bb);
    CPG_CHECK_LAUNCH("cpg_conv2d_fwd(3x3 s2)");
    return CPG_OK;
}

inline int s2_out(int v) { return (v - 1) / 2 + 1; }

int run_fwd_s2(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias, float *y, void *ws,
               size_t ws_bytes, hipStream_t stream, float *stats, int *tiles_out, bool dry) {
    const char *what = "cpg_conv2d_fwd(3x3 s2)";
    const int OH = s2_out(d->H), OW = s2_out(d->W);
    float *wp = (float *)ws;
    const int rows_c = pad_to(d->C, 4), Mp = pad_to(d->K, 128);
    if (!dry) {
        const size_t need = pack_bytes(d->C, d->K);
        if (ws == nullptr || ws_bytes < need) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
        CPG_REQUIRE((((uintptr_t)ws) & 15) == 0, "%s: workspace must be 16-byte aligned", what);
        hipLaunchKernelGGL(k_c3_pack, dim3(stream_grid((int64_t)rows_c * 9 * Mp, 256)), dim3(256), 0, stream, w, pm, thr, wp, d->K, d->C,
                           rows_c, Mp, 0, (int *)nullptr);
    }
    C3Geom g{d->N, d->C, d->H, d->W, d->K, Mp, 0, 0, 0, OH, OW, 0, 1};
    if (OW == 28 && OH % 4 == 0 && d->K > 64) return launch_fwd_s2<S2P28>(g, x, wp, bias, y, stream, stats, tiles_out, dry, nullptr);
    if (OW <= 8 && OH <= 8 && d->K > 64) return launch_fwd_s2<S2M8>(g, x, wp, bias, y, stream, stats, tiles_out, dry, nullptr);
    if (OW <= 16 && OH <= 16 && d->K > 64) return launch_fwd_s2<S2S16>(g, x, wp, bias, y, stream, stats, tiles_out, dry, nullptr);
    if (d->K <= 64) return launch_fwd_s2<S2G64>(g, x, wp, bias, y, stream, stats, tiles_out, dry, nullptr);
    return launch_fwd_s2<S2G128>(g, x, wp, bias, y, stream, stats, tiles_out, dry, nullptr);
}

//                  BM  WM WN TH TW NIMG
using D2a = D2Cfg<128, 4, 1, 4, 16, 1>;      // class grids >= 17 wide (gx 56 x 56 -> 28 x 28 classes)
using D2b = D2Cfg<128, 4, 1, 2, 16, 2>;      // class grids <= 16 wide (gx 28 x 28 -> 14 x 14): two rows of two images, no row waste
using D2c = D2Cfg<128, 4, 1, 1, 8, 8>;       // class grids <= 8 wide (gx 14 x 14 -> 7 x 7): one row of eight images
using D2a64 = D2Cfg<64, 2, 2, 4, 16, 1>;     // the same for <= 64 input channels
using D2b64 = D2Cfg<64, 2, 2, 2, 16, 2>;
using D2c64 = D2Cfg<64, 2, 2, 1, 8, 8>;

template <class Cfg>
int launch_dgrad_s2(C3Geom g, const float *gy, const float *wp, float *gx, hipStream_t stream) {
    const int GH = (g.OH + 1) / 2, GW = (g.OW + 1) / 2;           // the class grid (i, j) = (h >> 1, w >> 1)
    g.tiles_x = (GW + Cfg::TW - 1) / Cfg::TW;
    g.tiles_y = (GH + Cfg::TH - 1) / Cfg::TH;
    g.tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM;
    const int64_t blocks = (int64_t)((g.N + Cfg::NIMG - 1) / Cfg::NIMG) * g.tiles_x * g.tiles_y * g.tiles_m;
    if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "conv3x3 s2 dgrad: grid too large");
    hipLaunchKernelGGL(k_c3s2_dgrad<Cfg>, dim3((unsigned)blocks), dim3(256), 0, stream, g, gy, wp, gx);
    CPG_CHECK_LAUNCH("cpg_conv2d_dgrad(3x3 s2)");
    return CPG_OK;
}

using W3S2a = W3Cfg<2, 14, false, 1>;        // gy maps 14 / 28 / 56 wide
using W3S2b = W3Cfg<4, 8, false, 1>;         // everything else (7 x 7: 8-wide tiles)
}  // namespace

extern "C" int cpg_conv3x3s2_supported(const cpg_conv_desc *d) {
    if (cpg::opt_on(cpg::OPT_DISABLE_CONV3X3) || cpg::opt_on(cpg::OPT_NO_S2)) return 0;
    return d->R == 3 && d->S == 3 && d->stride_h == 2 && d->stride_w == 2 && d->pad_h == 1 && d->pad_w == 1 && d->dil_h == 1 &&
           d->dil_w == 1 && d->groups == 1 && d->N > 0 && d->C >= 16 && d->K >= 16 && d->H > 1 && d->W > 1 &&
           (int64_t)d->C * d->H * d->W < (1ll << 27) && (int64_t)d->K * d->H * d->W < (1ll << 27) && (int64_t)d->H * d->W <= (1ll << 22);
}
size_t cpg_conv3x3s2_pack_workspace(const cpg_conv_desc *d) { return std::max(pack_bytes(d->C, d->K), pack_bytes(d->K, d->C)) + 16; }
int cpg_conv3x3s2_bnstats_tiles(const cpg_conv_desc *d) {
    int tiles = 0;
    if (run_fwd_s2(d, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, 0, nullptr, nullptr, &tiles, true) != CPG_OK) return 0;
    return tiles;
}
int cpg_conv3x3s2_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias, float *y,
                      float *stats, void *ws, size_t ws_bytes, hipStream_t stream) {
    CPG_REQUIRE(x && w && y, "cpg_conv2d_fwd: null pointer");
    return run_fwd_s2(d, x, w, pm, thr, bias, y, ws, ws_bytes, stream, stats, nullptr, false);
}
int cpg_conv3x3s2_dgrad(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, float *gx, void *ws,
                        size_t ws_bytes, hipStream_t stream) {
    CPG_REQUIRE(gy && w && gx, "cpg_conv2d_dgrad: null pointer");
    const char *what = "cpg_conv2d_dgrad(3x3 s2)";
    const int OH = s2_out(d->H), OW = s2_out(d->W);
    float *wp = (float *)ws;
    const int rows_c = pad_to(d->K, 4), Mp = pad_to(d->C, 128);
    const size_t need = pack_bytes(d->K, d->C);
    if (ws == nullptr || ws_bytes < need) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    hipLaunchKernelGGL(k_c3_pack, dim3(stream_grid((int64_t)rows_c * 9 * Mp, 256)), dim3(256), 0, stream, w, pm, thr, wp, d->K, d->C,
                       rows_c, Mp, 1, (int *)nullptr);
    // reads gy (K channels, OH x OW), produces gx (C channels, H x W)
    C3Geom g{d->N, d->K, OH, OW, d->C, Mp, 0, 0, 0, d->H, d->W, 1, 1};
    const int GW = (d->W + 1) / 2;
    if (d->C <= 64) {
        if (GW <= 8) return launch_dgrad_s2<D2c64>(g, gy, wp, gx, stream);
        if (GW <= 16) return launch_dgrad_s2<D2b64>(g, gy, wp, gx, stream);
        return launch_dgrad_s2<D2a64>(g, gy, wp, gx, stream);
    }
    if (GW <= 8) return launch_dgrad_s2<D2c>(g, gy, wp, gx, stream);
    if (GW <= 16) return launch_dgrad_s2<D2b>(g, gy, wp, gx, stream);
    return launch_dgrad_s2<D2a>(g, gy, wp, gx, stream);
}
static inline bool s2_wide(const cpg_conv_desc *d) { return s2_out(d->W) % 14 == 0; }
size_t cpg_conv3x3s2_wgrad_workspace(const cpg_conv_desc *d) {
    return s2_wide(d) ? w3_plan<W3S2a>(d).ws_bytes : w3_plan<W3S2b>(d).ws_bytes;
}
int cpg_conv3x3s2_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr, float *gw,
                        float *gpm, void *ws, size_t ws_bytes, hipStream_t stream) {
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    return s2_wide(d) ? w3_launch<W3S2a>(d, x, gy, ep, ws, ws_bytes, stream) : w3_launch<W3S2b>(d, x, gy, ep, ws, ws_bytes, stream);
}
