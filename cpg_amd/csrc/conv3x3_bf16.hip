// OPT-IN bf16 MFMA path of the masked 3x3 / stride 1 / pad 1 convolution (north_star: "MFMA fp32/bf16 GEMM").
//
// Never the default and never the headline number: the reference computes in fp32 and north_star's parity bar (logits
// within 1e-4) is an fp32 bar, which the fp32 kernels of conv3x3.hip meet.  This path trades it for the 16x higher
// MFMA rate of v_mfma_f32_32x32x16_bf16: activations and effective weights are ROUNDED TO bf16 (8-bit mantissa, round to
// nearest even) on their way into LDS, products are exact, accumulation is fp32 -- the arithmetic of torch autocast(bf16)
// convolutions.  Its own tolerance (tests: 2e-2 of the output scale against the fp32 oracle, 1e-6 against an oracle fed
// the same bf16-rounded operands) and its own roofline (2.5 PFLOP/s dense bf16) apply; tensors in HBM stay fp32 NCHW, so
// nothing else in the framework changes and the two paths can be mixed per call.
//
// forward + input gradient (one kernel; as in conv3x3.hip only the packed weights differ):
//   k_c3b_pack writes the effective weights W * bin(piggymask) as bf16 in the order the kernel's LDS wants them,
//       Wp[chunk of 16 channels][tap][half h][m][8 channels 16*chunk + 8*h ...]      (one uint4 = 8 bf16 per (.., m))
//   Block = BM output channels x (TH x TW) pixels of one image; loop over input channels in chunks of 16.  Per chunk the
//   block stages
//       Ws[9][2][BM]  uint4      a straight copy of the packed rows (global_load_dwordx4 -> ds_write_b128)
//       Xs[2][PH*PW]  uint4      the zero-padded (TH+2) x (TW+2) patch, channel-last in groups of 8: a thread loads the 8
//                                channels of one (pixel, half) with range-checked buffer loads (zero padding and the
//                                channel tail cost no select), converts (v_cvt_pk_bf16_f32) and writes ONE ds_write_b128
//   into the other LDS stage while the MFMAs of the current chunk run.  One k-step = (tap, 16 channels): lanes 0-31 hold
//   channels 0-7, lanes 32-63 channels 8-15 of A (row = output channel) and B (column = pixel); both operand reads are one
//   ds_read_b128 per fragment whose 16 consecutive lanes cover 256 contiguous bytes (conflict free).
//   Work per staged byte is 16x lower than in the fp32 kernel, so this kernel is bound by L2 / HBM delivery of the patch
//   and the packed weights on most VGG layers, not by the matrix pipe (docs/LAB_NOTEBOOK.md section 4.6 has the measurements).
#include <algorithm>
#include "igemm_core.h"

using namespace cpg;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;          // round to nearest even; the compiler pairs the two conversions into one v_cvt_pk_bf16_f32
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// lo part of the two-term split v = hi + lo (hi = bf16(v), lo = bf16(v - hi)): with NP = 2 planes the kernels accumulate
// a_hi b_hi + a_hi b_lo + a_lo b_hi ("bf16x3"), which restores ~16 mantissa bits per product
__device__ __forceinline__ float lo_part(float v) { return v - (float)(__bf16)v; }

struct B16Geom {
    int N, C, H, W, M;        // C: channels read, M: channels produced
    int Mp;                   // M rounded up to 128 (row length of the packed weights, in uint4)
    int tiles_x, tiles_y, tiles_m, nchunks;
};

template <int BM_, int TH_, int TW_, int WM_, int WN_, int MINB_, int NP_ = 1, int NSTAGE_ = 2>
struct B16Cfg {
    static constexpr int NSTAGE = NSTAGE_;                    // LDS stages: 2, or 1 (the next chunk waits in registers until the MFMAs are done)
    static constexpr int BM = BM_, TH = TH_, TW = TW_, WM = WM_, WN = WN_, MINB = MINB_;
    static constexpr int NP = NP_;                            // operand planes: 1 = bf16, 2 = hi + lo ("bf16x3")
    static constexpr int BN = TH * TW;
    static_assert(WM * WN == 4 && BN % (32 * WN) == 0 && BM % (32 * WM) == 0, "bad bf16 conv config");
    static constexpr int FM = BM / 32 / WM, FN = BN / 32 / WN;
    static constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW;
    static constexpr int NPIXP = (NPIX + 15) / 16 * 16;       // plane of one channel half, multiple of 256 bytes
    static constexpr int WQ = 9 * 2 * BM;                     // uint4 per stage: weights
    static constexpr int XQ = 2 * NPIXP;                      //                  patch
    static constexpr int STAGE1 = WQ + XQ;                    // one plane; the lo plane (NP = 2) repeats the layout behind it
    static constexpr int STAGEQ = NP * STAGE1;
    static constexpr int NWI = (WQ + 255) / 256;              // weight uint4 per thread per chunk and plane
    static constexpr int NXI = (2 * NPIX + 255) / 256;        // (pixel, half) items per thread per chunk
};

// ------------------------------------------------------------------------------ weight pack
// out[((chunk * 9 + tap') * 2 + h) * Mp + m] = 8 bf16: channels 16 * chunk + 8 * h + j (j = 0..7) of
//   fwd  : W[co = m][ci = c][tap = tap'] * bin(pm)
//   dgrad: W[co = c][ci = m][tap = 8 - tap'] * bin(pm)        (conv of gy with the spatially flipped, transposed filter)
// np = 2: every chunk holds two such [tap][half][m] blocks, hi then lo.
__global__ __launch_bounds__(256) void k_c3b_pack(const float *__restrict__ w, const float *__restrict__ pm, float thr,
                                                  u32x4 *__restrict__ out, int K, int C, int nchunks, int Mp, int dgrad, int np) {
    const int64_t total = (int64_t)nchunks * 9 * 2 * Mp;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += nthreads) {
        const int m = (int)(o % Mp);
        int r = (int)(o / Mp);
        const int h = r & 1;
        r >>= 1;
        const int tp = r % 9, chunk = r / 9;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chunk * 16 + h * 8 + j;
            const int co = dgrad ? c : m, ci = dgrad ? m : c, tap = dgrad ? 8 - tp : tp;
            v[j] = 0.0f;
            if (co < K && ci < C) {
                const int64_t off = ((int64_t)co * C + ci) * 9 + tap;
                v[j] = w[off];
                if (pm != nullptr) v[j] *= binarize(pm[off], thr);
            }
        }
        // o = (chunk * 18 + row) * Mp + m  ->  chunk-major with np planes per chunk
        const int64_t dst = ((int64_t)chunk * np * 18 + (tp * 2 + h)) * Mp + m;
        u32x4 q = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        out[dst] = q;
        if (np == 2) {
            u32x4 l = {pack2(lo_part(v[0]), lo_part(v[1])), pack2(lo_part(v[2]), lo_part(v[3])), pack2(lo_part(v[4]), lo_part(v[5])),
                       pack2(lo_part(v[6]), lo_part(v[7]))};
            out[dst + (int64_t)18 * Mp] = l;
        }
    }
}

// ------------------------------------------------------------------------------ fwd / dgrad
template <class Cfg>
__global__ __launch_bounds__(256, Cfg::MINB) void k_c3b_fwd(B16Geom g, const float *__restrict__ x, const u32x4 *__restrict__ wp,
                                                            const float *__restrict__ bias, float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4 *smem = reinterpret_cast<u32x4 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int li = lane & 31, lh = lane >> 5;

    unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lb % g.tiles_m; lb /= g.tiles_m;
    const int tx = lb % g.tiles_x; lb /= g.tiles_x;
    const int ty = lb % g.tiles_y;
    const int n = lb / g.tiles_y;
    const int m0 = tm * Cfg::BM, h0 = ty * Cfg::TH, w0 = tx * Cfg::TW;
    const int HW = g.H * g.W;

    // ---- staging descriptors (fixed for the life of the block) ----
    // weights: uint4 item e = tid + 256 i of [9][2][BM]: row (tap, half) = e / BM, column e % BM
    int wsrc[Cfg::NWI];
#pragma unroll
    for (int i = 0; i < Cfg::NWI; ++i) {
        const int e = min(tid + 256 * i, Cfg::WQ - 1);                 // (clamped duplicates re-write the same value)
        wsrc[i] = (e / Cfg::BM) * g.Mp + m0 + (e % Cfg::BM);
    }
    // patch: item e = tid + 256 i of [2 halves][NPIX]: byte offset of (channel 8 * half, pixel) from the image's channel 0,
    // or an out-of-range offset for halo positions outside the image (the buffer unit then returns 0)
    constexpr int kOutOfRange = (int)0x80000000;
    int xoff[Cfg::NXI], xdst[Cfg::NXI];
#pragma unroll
    for (int i = 0; i < Cfg::NXI; ++i) {
        const int e = tid + 256 * i;
        const int half = e / Cfg::NPIX, p = e - half * Cfg::NPIX;
        const int pr = p / Cfg::PW, pc = p - pr * Cfg::PW;
        const int gh = h0 - 1 + pr, gw = w0 - 1 + pc;
        const bool ok = e < 2 * Cfg::NPIX && (unsigned)gh < (unsigned)g.H && (unsigned)gw < (unsigned)g.W;
        xoff[i] = ok ? (half * 8 * HW + gh * g.W + gw) * 4 : kOutOfRange;
        xdst[i] = e < 2 * Cfg::NPIX ? Cfg::WQ + half * Cfg::NPIXP + p : -1;
    }
    const __amdgpu_buffer_rsrc_t srd_x =
        __builtin_amdgcn_make_buffer_rsrc((void *)(x + (int64_t)n * g.C * HW), 0, g.C * HW * 4, 0x00020000);

    u32x4 rw[Cfg::NP][Cfg::NWI];
    float rx[Cfg::NXI][8];
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int pl = 0; pl < Cfg::NP; ++pl) {
            const u32x4 *wsrc_ch = wp + ((int64_t)ch * Cfg::NP + pl) * 18 * g.Mp;
#pragma unroll
            for (int i = 0; i < Cfg::NWI; ++i) rw[pl][i] = wsrc_ch[wsrc[i]];
        }
        const int cbyte = ch * 16 * HW * 4;
#pragma unroll
        for (int i = 0; i < Cfg::NXI; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                // (channel offset in the VECTOR offset: only that one is range-checked, and the check is what zero-fills
                // channels past C in the last chunk)
                rx[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, xoff[i] + cbyte + j * HW * 4, 0, 0));
    };
    auto store_chunk = [&](u32x4 *stage) {
#pragma unroll
        for (int pl = 0; pl < Cfg::NP; ++pl)
#pragma unroll
            for (int i = 0; i < Cfg::NWI; ++i) stage[pl * Cfg::STAGE1 + min(tid + 256 * i, Cfg::WQ - 1)] = rw[pl][i];
#pragma unroll
        for (int i = 0; i < Cfg::NXI; ++i) {
            if (xdst[i] >= 0) {
                u32x4 q = {pack2(rx[i][0], rx[i][1]), pack2(rx[i][2], rx[i][3]), pack2(rx[i][4], rx[i][5]), pack2(rx[i][6], rx[i][7])};
                stage[xdst[i]] = q;
                if (Cfg::NP == 2) {
                    u32x4 l = {pack2(lo_part(rx[i][0]), lo_part(rx[i][1])), pack2(lo_part(rx[i][2]), lo_part(rx[i][3])),
                               pack2(lo_part(rx[i][4]), lo_part(rx[i][5])), pack2(lo_part(rx[i][6]), lo_part(rx[i][7]))};
                    stage[Cfg::STAGE1 + xdst[i]] = l;
                }
            }
        }
    };

    // ---- operand lane bases (uint4 index inside a stage) ----
    const int a_base = lh * Cfg::BM + wm * Cfg::FM * 32 + li;                              // + tap * 2 * BM + fm * 32
    int b_base[Cfg::FN];
#pragma unroll
    for (int fn = 0; fn < Cfg::FN; ++fn) {
        const int t = (wn * Cfg::FN + fn) * 32 + li;
        b_base[fn] = Cfg::WQ + lh * Cfg::NPIXP + (t / Cfg::TW) * Cfg::PW + (t % Cfg::TW);   // + kh * PW + kw
    }

    f32x16 acc[Cfg::FM][Cfg::FN];
#pragma unroll
    for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[fm][fn][e] = 0.0f;

    load_chunk(0);
    store_chunk(smem);
    __syncthreads();
    for (int ch = 0; ch < g.nchunks; ++ch) {
        const u32x4 *cur = smem + (Cfg::NSTAGE == 2 ? (ch & 1) : 0) * Cfg::STAGEQ;
        u32x4 *other = smem + (Cfg::NSTAGE == 2 ? ((ch + 1) & 1) : 0) * Cfg::STAGEQ;
        const bool more = ch + 1 < g.nchunks;                      // block-uniform
        if (more) load_chunk(ch + 1);                              // in flight under the MFMAs below
        // Operands of tap t + 1 are read from LDS while the MFMAs of tap t run (two register sets), and the MFMAs of a tap are
        // ordered product-major, so consecutive MFMAs never share an accumulator (the compiler's own order was: per fragment
        // ds_read -> s_waitcnt lgkmcnt(0) -> 3 dependent MFMAs on one accumulator: the pipe idled on both the LDS latency and
        // the accumulate dependency).  sched_group_barrier pins one LDS read behind each MFMA.
        bf16x8 a[2][Cfg::NP][Cfg::FM], b[2][Cfg::NP][Cfg::FN];
        auto lds_operands = [&](int tap, int set) {
#pragma unroll
            for (int pl = 0; pl < Cfg::NP; ++pl) {
#pragma unroll
                for (int fm = 0; fm < Cfg::FM; ++fm)
                    a[set][pl][fm] = __builtin_bit_cast(bf16x8, cur[pl * Cfg::STAGE1 + a_base + tap * 2 * Cfg::BM + fm * 32]);
#pragma unroll
                for (int fn = 0; fn < Cfg::FN; ++fn)
                    b[set][pl][fn] = __builtin_bit_cast(bf16x8, cur[pl * Cfg::STAGE1 + b_base[fn] + (tap / 3) * Cfg::PW + (tap % 3)]);
            }
        };
        // (one-plane kernels: measured 5 % slower with the pinned order -- they are bound by L2 delivery of the next chunk, not
        // by operand latency -- so they keep read-then-multiply per tap and the compiler's schedule)
        constexpr bool kPipelined = Cfg::NP == 2;
        if (kPipelined) lds_operands(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int set = kPipelined ? (tap & 1) : 0;
            if (!kPipelined) lds_operands(tap, 0);
            else if (tap + 1 < 9) lds_operands(tap + 1, set ^ 1);
            constexpr int NPROD = Cfg::NP == 2 ? 3 : 1;
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr) {
                // bf16x3: small terms first (a_lo b_hi, a_hi b_lo), then a_hi b_hi; a_lo b_lo (~2^-16 of it) is dropped
                const int pa = (Cfg::NP == 2 && pr == 0) ? 1 : 0, pb = (Cfg::NP == 2 && pr == 1) ? 1 : 0;
#pragma unroll
                for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
                    for (int fn = 0; fn < Cfg::FN; ++fn)
                        acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[set][pa][fm], b[set][pb][fn], acc[fm][fn], 0, 0, 0);
            }
            constexpr int NREADS = Cfg::NP * (Cfg::FM + Cfg::FN), NMFMA = NPROD * Cfg::FM * Cfg::FN;
            if (kPipelined) {
#pragma unroll
                for (int i = 0; i < NMFMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // one MFMA
                    if (tap + 1 < 9 && i < NREADS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one LDS read of the next tap
                }
                if (tap + 1 < 9 && NREADS > NMFMA) __builtin_amdgcn_sched_group_barrier(0x100, NREADS - NMFMA, 0);
            }
        }
        if (Cfg::NSTAGE == 1) __syncthreads();                     // one stage: everybody is done reading it before it is overwritten
        if (more) store_chunk(other);
        __syncthreads();
    }

    // ---- epilogue: D col = pixel (lane & 31), D row = channel ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) ----
#pragma unroll
    for (int fn = 0; fn < Cfg::FN; ++fn) {
        const int t = (wn * Cfg::FN + fn) * 32 + li;
        const int oh = h0 + t / Cfg::TW, ow = w0 + t % Cfg::TW;
        const bool pok = oh < g.H && ow < g.W;
        float *yout = y + (int64_t)n * g.M * HW + oh * g.W + ow;
#pragma unroll
        for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = m0 + (wm * Cfg::FM + fm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (pok && co < g.M) yout[(int64_t)co * HW] = acc[fm][fn][e] + (bias != nullptr ? bias[co] : 0.0f);
            }
    }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }
inline size_t packb_bytes(int c_read, int m, int np = 2) { return (size_t)((c_read + 15) / 16) * np * 18 * pad_to(m, 128) * sizeof(u32x4); }

template <class Cfg>
int launch(B16Geom g, const float *x, const u32x4 *wp, const float *bias, float *y, hipStream_t stream, const char *what) {
    g.tiles_x = (g.W + Cfg::TW - 1) / Cfg::TW;
    g.tiles_y = (g.H + Cfg::TH - 1) / Cfg::TH;
    g.tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM;
    const int64_t blocks = (int64_t)g.N * g.tiles_x * g.tiles_y * g.tiles_m;
    if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "%s: grid too large", what);
    constexpr size_t smem = (size_t)Cfg::NSTAGE * Cfg::STAGEQ * sizeof(u32x4);
    static_assert(smem <= 160 * 1024, "LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_c3b_fwd<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_status(e, what);
    hipLaunchKernelGGL((k_c3b_fwd<Cfg>), dim3((unsigned)blocks), dim3(256), smem, stream, g, x, wp, bias, y);
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}

//                    BM  TH  TW  WM WN MINB
using B16W128 = B16Cfg<128, 8, 56, 2, 2, 1>;      // 56 / 112 / 224 wide maps, >= 128 channels: 7 pixel fragments, zero tile waste
using B16W64 = B16Cfg<64, 8, 56, 2, 2, 1>;        // same maps, <= 64 channels
using B16N128 = B16Cfg<128, 8, 32, 2, 2, 1>;      // everything else
using B16N64 = B16Cfg<64, 8, 32, 2, 2, 2>;
using B16P28 = B16Cfg<128, 7, 32, 4, 1, 1>;       // 28-high maps: 4 tiles of 7 rows (only the 4 padding columns are wasted)
using B16S16 = B16Cfg<128, 14, 16, 4, 1, 1>;      // 14 x 14 (<= 16 wide) maps: the whole image, 7 fragments
// two-plane ("bf16x3") variants: twice the LDS per chunk, so 64 output channels per block
using X3W = B16Cfg<64, 8, 56, 2, 2, 1, 2>;
using X3N = B16Cfg<64, 8, 32, 2, 2, 1, 2>;
using X3S = B16Cfg<64, 8, 16, 2, 2, 1, 2>;        // <= 16 wide maps
// ... and, for >= 128 output channels on <= 32 wide maps, a 128-channel tile on ONE LDS stage (the two-plane kernels are bound by
// L2 delivery: a block that produces twice the channels from the same patch needs 25 % fewer bytes per MFMA): 512 -> 512 @28 went
// 278 -> 322 TFLOP/s.  The same with the 8 x 56 tile spills (71 VGPRs) and measured 10 % slower.
using X3N1 = B16Cfg<128, 8, 32, 2, 2, 1, 2, 1>;

int run(bool dgrad, int N, int c_read, int m, int H, int W, int K, int C, const float *x, const float *w, const float *pm, float thr,
        const float *bias, float *y, void *ws, size_t ws_bytes, hipStream_t stream, int np) {
    const char *what = dgrad ? "cpg_conv2d_dgrad_bf16" : "cpg_conv2d_fwd_bf16";
    const size_t need = packb_bytes(c_read, m, np);
    if (ws == nullptr || ws_bytes < need) return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    CPG_REQUIRE((((uintptr_t)ws) & 15) == 0, "%s: workspace must be 16-byte aligned", what);
    u32x4 *wp = (u32x4 *)ws;
    const int nchunks = (c_read + 15) / 16, Mp = pad_to(m, 128);
    hipLaunchKernelGGL(k_c3b_pack, dim3(stream_grid((int64_t)nchunks * 18 * Mp, 256)), dim3(256), 0, stream, w, pm, thr, wp, K, C, nchunks,
                       Mp, dgrad ? 1 : 0, np);
    B16Geom g{N, c_read, H, W, m, Mp, 0, 0, 0, nchunks};
    if (np == 2) {
        if (m > 64 && W % 56 != 0 && W > 16) return launch<X3N1>(g, x, wp, bias, y, stream, what);
        if (W % 56 == 0) return launch<X3W>(g, x, wp, bias, y, stream, what);
        if (W <= 16) return launch<X3S>(g, x, wp, bias, y, stream, what);
        return launch<X3N>(g, x, wp, bias, y, stream, what);
    }
    const bool wide = W % 56 == 0;
    if (wide) return m > 64 ? launch<B16W128>(g, x, wp, bias, y, stream, what) : launch<B16W64>(g, x, wp, bias, y, stream, what);
    if (m > 64 && W <= 16 && H <= 14) return launch<B16S16>(g, x, wp, bias, y, stream, what);
    if (m > 64 && W <= 32 && H % 7 == 0) return launch<B16P28>(g, x, wp, bias, y, stream, what);
    return m > 64 ? launch<B16N128>(g, x, wp, bias, y, stream, what) : launch<B16N64>(g, x, wp, bias, y, stream, what);
}


// ------------------------------------------------------------------------------ weight gradient
// D[co][ci](tap) = sum over images and pixels of gy[co][pix] * x[ci][pix + tap offset], k of the MFMA = 16 consecutive pixels
// of one image row (lanes 0-31: pixels 0-7, lanes 32-63: pixels 8-15 of the k-step).
//
// A "unit" is TH rows x TW pixels (TW a multiple of 16) of one image.  LDS holds it channel-fastest in groups of 8 pixels,
//     Gs[row][TW/8 groups][64 co]      uint4 (8 bf16)         gy
//     Xs[row + halo][TW/8 + 2 groups][64 ci]   uint4          x, one group of left and right halo per row (w0 - 8 ... w0 + TW + 7)
// so that an operand is ONE ds_read_b128 with consecutive lanes (channels) 16 bytes apart.  The three horizontal taps of a row
// need the same 8 pixels shifted by -1 / 0 / +1 pixel: kw = 1 is the group itself, kw = 0 and kw = 2 are assembled in registers
// from the group and one dword of its left / right neighbour with five v_alignbit (three of them shared by the two shifted
// fragments).  Each wave owns a 32 co x 32 ci fragment for all 9 taps (9 accumulators); per k-step and row offset kh it issues 3
// MFMAs against one b128 + two b32 reads.
// Staging: thread = (channel, group): eight range-checked dword buffer loads (8 consecutive pixels = one 32-byte sector; groups
// outside the image get an out-of-range offset and read as 0; a 16-lane group of a wave covers 16 channels x 4 consecutive
// groups = one 128-byte line per channel), four v_cvt_pk_bf16_f32, one ds_write_b128.  (buffer_load_dwordx4 through the
// same raw descriptor returned its first dword four times on gfx950 -- tools/attic/diag_wgrad16.py -- and bought 4 % at best.)  The loads of unit u+1 are issued before the
// MFMAs of unit u and stored into the other LDS stage after them (two stages, one barrier per unit, one block per CU).
// MASKW: W is not a multiple of 8 (28-wide maps): the group that straddles the right image border is masked per element.
// Split-K over units exactly like the fp32 kernel (k_c3_wgrad): tap-major partials, reduced (with the autograd epilogue
// gW = g * bin(pm), gPM = g * W) by k_split_reduce.
template <int TH_, int TW_, bool MASKW_, int NP_ = 1, int NSTAGE_ = 2>
struct B16WCfg {
    static constexpr int TH = TH_, TW = TW_, NP = NP_, NSTAGE = NSTAGE_;
    static constexpr bool MASKW = MASKW_;
    static_assert(TW % 16 == 0, "k-steps are 16 pixels of one row");
    static constexpr int GG = TW / 8, XG = GG + 2, XR = TH + 2;          // groups per gy row / x row, x rows
    static constexpr int GQ = TH * GG * 64, XQ = XR * XG * 64, STAGE1 = GQ + XQ;      // uint4 per plane
    static constexpr int STAGEQ = NP * STAGE1;                                        // (the lo plane repeats the layout)
    static constexpr int NGI = GQ / 256, NXI = (XQ + 255) / 256;          // staging items (channel, group) per thread
    static_assert(GQ % 256 == 0, "gy items must fill the block");
};

template <class Cfg>
__global__ __launch_bounds__(256, 1) void k_c3b_wgrad(int N, int C, int H, int W, int M, int tiles_x, int tiles_y, int tiles_co,
                                                      int tiles_ci, int units_per_split, const float *__restrict__ x,
                                                      const float *__restrict__ gy, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4 *smem = reinterpret_cast<u32x4 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave >> 1, wci = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // block -> (split, channel tile): all tiles of one split on one XCD, as in k_c3_wgrad
    const int tiles = tiles_co * tiles_ci;
    const int xcd = blockIdx.x % kXCDs, j = blockIdx.x / kXCDs;
    const int split = (j / tiles) * kXCDs + xcd, tile = j % tiles;
    const int tci = tile % tiles_ci, tco = tile / tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int HW = H * W;
    const int units_per_img = tiles_x * tiles_y;
    const int total_units = N * units_per_img;
    const int u0 = min(total_units, split * units_per_split);
    const int u1 = min(total_units, u0 + units_per_split);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;

    // ---- staging items: 64-lane blocks of (16 channels x 4 consecutive groups); lane l: channel l % 16, group l / 16 ----
    // item i of this thread covers block b = wave + 4 i:  channels 16 * (b % 4) ..., groups 4 * (b / 4) ...
    constexpr int kOutOfRange = (int)0x80000000;
    int gch[Cfg::NGI], ggrp[Cfg::NGI], xch[Cfg::NXI], xgrp[Cfg::NXI];
    int grow[Cfg::NGI], gcol[Cfg::NGI], xrow[Cfg::NXI], xcol[Cfg::NXI];      // row and first pixel of the item's group inside the unit
#pragma unroll
    for (int i = 0; i < Cfg::NGI; ++i) {
        const int b = wave + 4 * i;
        gch[i] = 16 * (b & 3) + (lane & 15);
        ggrp[i] = 4 * (b >> 2) + (lane >> 4);                       // < TH * GG
        grow[i] = ggrp[i] / Cfg::GG, gcol[i] = 8 * (ggrp[i] % Cfg::GG);
    }
#pragma unroll
    for (int i = 0; i < Cfg::NXI; ++i) {
        const int b = wave + 4 * i;
        xch[i] = 16 * (b & 3) + (lane & 15);
        xgrp[i] = 4 * (b >> 2) + (lane >> 4);                       // may exceed XR * XG in the last item: skipped
        xrow[i] = xgrp[i] / Cfg::XG - 1, xcol[i] = 8 * (xgrp[i] % Cfg::XG) - 8;
    }
    auto clamp_g = [&](int c) { return min(c, M - 1 - co0); };       // channels past M / C: duplicates of the last one, they only
    auto clamp_x = [&](int c) { return min(c, C - 1 - ci0); };       // reach accumulator rows / columns that are never stored

    __amdgpu_buffer_rsrc_t srd_g, srd_x;
    float rg[Cfg::NGI][8], rx[Cfg::NXI][8];
    int gnv[Cfg::NGI], xnv[Cfg::NXI];                                // MASKW: number of valid pixels in the group (0..8)
    auto load_unit = [&](int u) {
        const int n = u / units_per_img, rr = u - n * units_per_img;
        const int ty = rr / tiles_x, tx = rr - ty * tiles_x;
        const int h0 = ty * Cfg::TH, w0 = tx * Cfg::TW;
        // num_records = what is left of the tensor: a 16-byte load that starts inside the last row never reads past the allocation
        const int64_t g_left = ((int64_t)(N - n) * M - co0) * HW * 4, x_left = ((int64_t)(N - n) * C - ci0) * HW * 4;
        srd_g = __builtin_amdgcn_make_buffer_rsrc((void *)(gy + ((int64_t)n * M + co0) * HW), 0, (int)min(g_left, (int64_t)0x7FFFFFFF), 0x00020000);
        srd_x = __builtin_amdgcn_make_buffer_rsrc((void *)(x + ((int64_t)n * C + ci0) * HW), 0, (int)min(x_left, (int64_t)0x7FFFFFFF), 0x00020000);
#pragma unroll
        for (int i = 0; i < Cfg::NGI; ++i) {
            const int gh = h0 + grow[i], gw = w0 + gcol[i];
            const bool ok = gh < H && gw < W;
            const int off = ok ? (clamp_g(gch[i]) * HW + gh * W + gw) * 4 : kOutOfRange;
            if (Cfg::MASKW) gnv[i] = ok ? min(8, W - gw) : 0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                rg[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_g, off + 4 * e, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < Cfg::NXI; ++i) {
            const int gh = h0 + xrow[i], gw = w0 + xcol[i];
            const bool ok = xgrp[i] < Cfg::XR * Cfg::XG && (unsigned)gh < (unsigned)H && gw >= 0 && gw < W;
            const int off = ok ? (clamp_x(xch[i]) * HW + gh * W + gw) * 4 : kOutOfRange;
            if (Cfg::MASKW) xnv[i] = ok ? min(8, W - gw) : 0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                rx[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, off + 4 * e, 0, 0));
        }
    };
    auto put = [&](u32x4 *stage, int idx, const float (&v)[8]) {
        u32x4 q = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        stage[idx] = q;
        if (Cfg::NP == 2) {
            u32x4 l = {pack2(lo_part(v[0]), lo_part(v[1])), pack2(lo_part(v[2]), lo_part(v[3])), pack2(lo_part(v[4]), lo_part(v[5])),
                       pack2(lo_part(v[6]), lo_part(v[7]))};
            stage[Cfg::STAGE1 + idx] = l;
        }
    };
    auto store_unit = [&](u32x4 *stage) {
#pragma unroll
        for (int i = 0; i < Cfg::NGI; ++i) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (Cfg::MASKW && e >= gnv[i]) ? 0.0f : rg[i][e];
            put(stage, ggrp[i] * 64 + gch[i], v);
        }
#pragma unroll
        for (int i = 0; i < Cfg::NXI; ++i) {
            if (xgrp[i] < Cfg::XR * Cfg::XG) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (Cfg::MASKW && e >= xnv[i]) ? 0.0f : rx[i][e];
                put(stage, Cfg::GQ + xgrp[i] * 64 + xch[i], v);
            }
        }
    };

    const int a_lane = wco * 32 + li, b_lane = wci * 32 + li;
    auto compute = [&](const u32x4 *cur) {
        const unsigned *cur32 = reinterpret_cast<const unsigned *>(cur);
#pragma unroll
        for (int r = 0; r < Cfg::TH; ++r)
#pragma unroll
            for (int k = 0; k < Cfg::TW / 16; ++k) {
                bf16x8 a[Cfg::NP];
#pragma unroll
                for (int pl = 0; pl < Cfg::NP; ++pl)
                    a[pl] = __builtin_bit_cast(bf16x8, cur[pl * Cfg::STAGE1 + (r * Cfg::GG + 2 * k + lh) * 64 + a_lane]);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    bf16x8 b[Cfg::NP][3];                                   // [plane][kw]
#pragma unroll
                    for (int pl = 0; pl < Cfg::NP; ++pl) {
                        const int gi = pl * Cfg::STAGE1 + Cfg::GQ + ((r + kh) * Cfg::XG + 1 + 2 * k + lh) * 64 + b_lane;   // centre group
                        const u32x4 c = cur[gi];
                        const unsigned left = cur32[(gi - 64) * 4 + 3], right = cur32[(gi + 64) * 4 + 0];
                        const unsigned s01 = __builtin_amdgcn_alignbit(c[1], c[0], 16), s12 = __builtin_amdgcn_alignbit(c[2], c[1], 16),
                                       s23 = __builtin_amdgcn_alignbit(c[3], c[2], 16);
                        const u32x4 bl = {__builtin_amdgcn_alignbit(c[0], left, 16), s01, s12, s23};      // pixels p0 - 1 ... p0 + 6
                        const u32x4 br = {s01, s12, s23, __builtin_amdgcn_alignbit(right, c[3], 16)};     // pixels p0 + 1 ... p0 + 8
                        b[pl][0] = __builtin_bit_cast(bf16x8, bl);
                        b[pl][1] = __builtin_bit_cast(bf16x8, c);
                        b[pl][2] = __builtin_bit_cast(bf16x8, br);
                    }
                    // product-major: consecutive MFMAs go to different accumulators (small terms first in bf16x3)
                    constexpr int NPROD = Cfg::NP == 2 ? 3 : 1;
#pragma unroll
                    for (int pr = 0; pr < NPROD; ++pr) {
                        const int pa = (Cfg::NP == 2 && pr == 0) ? 1 : 0, pb = (Cfg::NP == 2 && pr == 1) ? 1 : 0;
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[pb][kw], acc[kh * 3 + kw], 0, 0, 0);
                    }
                }
            }
    };

    if (u0 < u1) {
        load_unit(u0);
        store_unit(smem);
        __syncthreads();
        for (int u = u0; u < u1; ++u) {
            const int cs = Cfg::NSTAGE == 2 ? ((u - u0) & 1) : 0;
            const bool more = u + 1 < u1;                             // block-uniform
            if (more) load_unit(u + 1);
            compute(smem + cs * Cfg::STAGEQ);
            if (Cfg::NSTAGE == 1) __syncthreads();                    // one stage: all reads done before it is overwritten
            if (more) store_unit(smem + (Cfg::NSTAGE == 2 ? (cs ^ 1) : 0) * Cfg::STAGEQ);
            __syncthreads();
        }
    }
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

struct B16WPlan {
    int tiles_x, tiles_y, tiles_co, tiles_ci, nsplit, units_per_split;
    size_t ws_bytes;
};
template <class Cfg>
B16WPlan wplan(const cpg_conv_desc *d) {
    B16WPlan p;
    p.tiles_x = (d->W + Cfg::TW - 1) / Cfg::TW;
    p.tiles_y = (d->H + Cfg::TH - 1) / Cfg::TH;
    p.tiles_co = (d->K + 63) / 64;
    p.tiles_ci = (d->C + 63) / 64;
    const int64_t units = (int64_t)d->N * p.tiles_x * p.tiles_y;
    const int64_t tiles = (int64_t)p.tiles_co * p.tiles_ci;
    int64_t want = (3 * kCUs + tiles - 1) / tiles;               // ~3 rounds of one block per CU
    if (want > units) want = units;
    if (want < 1) want = 1;
    if (want > 4096) want = 4096;
    want = (want + kXCDs - 1) / kXCDs * kXCDs;
    p.units_per_split = (int)((units + want - 1) / want);
    p.nsplit = (int)want;
    p.ws_bytes = (size_t)p.nsplit * d->K * d->C * 9 * sizeof(float);
    return p;
}
using B16G64 = B16WCfg<2, 64, false>;       // 56 / 112 / 224 wide maps (W a multiple of 8): 128 pixels per unit
using B16G32 = B16WCfg<4, 32, true>;        // any other width > 16 (28-wide maps): the group at the right border is masked per element
using B16G16 = B16WCfg<8, 16, true>;        // <= 16 wide maps (14 x 14)
// two-plane ("bf16x3") variants: half the rows per unit so that two stages still fit in LDS
using X3G64 = B16WCfg<1, 64, false, 2>;
// (two rows per unit on ONE stage -- 2x instead of 3x re-read of the x rows -- spilled 61 VGPRs and ran at half the speed)
using X3G32 = B16WCfg<2, 32, true, 2>;
using X3G16 = B16WCfg<4, 16, true, 2>;
inline int wpick(const cpg_conv_desc *d) { return d->W % 8 == 0 && d->W >= 56 ? 0 : d->W > 16 ? 1 : 2; }

template <class Cfg>
int wlaunch(const cpg_conv_desc *d, const float *x, const float *gy, const Epilogue &ep, void *ws, size_t ws_bytes, hipStream_t stream) {
    const B16WPlan p = wplan<Cfg>(d);
    if (ws == nullptr || ws_bytes < p.ws_bytes) return fail(CPG_E_WORKSPACE, "cpg_conv2d_wgrad_bf16: workspace %zu < %zu bytes", ws_bytes, p.ws_bytes);
    constexpr size_t smem = (size_t)Cfg::NSTAGE * Cfg::STAGEQ * sizeof(u32x4);
    static_assert(smem <= 160 * 1024, "LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_c3b_wgrad<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return hip_status(e, "cpg_conv2d_wgrad_bf16");
    hipLaunchKernelGGL(k_c3b_wgrad<Cfg>, dim3((unsigned)(p.tiles_co * p.tiles_ci * p.nsplit)), dim3(256), smem, stream, d->N, d->C, d->H, d->W,
                       d->K, p.tiles_x, p.tiles_y, p.tiles_co, p.tiles_ci, p.units_per_split, x, gy, (float *)ws);
    launch_split_reduce((const float *)ws, p.nsplit, (int64_t)d->K * d->C * 9, (int64_t)d->K * d->C, ep, stream);
    CPG_CHECK_LAUNCH("cpg_conv2d_wgrad_bf16");
    return CPG_OK;
}

}  // namespace

extern "C" int cpg_conv3x3_supported(const cpg_conv_desc *d);

// (fewer than 16 channels on either side -- the 3 -> 64 stem -- would fill most of a 16-channel k-step with zeros and is
// HBM-bound anyway: it stays on the fp32 kernels)
extern "C" int32_t cpg_conv2d_bf16_supported(const cpg_conv_desc *d) {
    return d != nullptr && cpg_conv3x3_supported(d) && d->C >= 16 && d->K >= 16 ? 1 : 0;
}

extern "C" size_t cpg_conv2d_bf16_workspace_bytes(const cpg_conv_desc *d) {       // (sized for the two-plane variant)
    if (!cpg_conv2d_bf16_supported(d)) return 0;
    return std::max(packb_bytes(d->C, d->K), packb_bytes(d->K, d->C));
}

static int fwd_any(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias, float *y,
                   void *ws, size_t ws_bytes, void *stream, int np) {
    if (!cpg_conv2d_bf16_supported(d)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_fwd_bf16: only 3x3 / stride 1 / pad 1 convolutions with >= 16 channels");
    CPG_REQUIRE(x && w && y, "cpg_conv2d_fwd_bf16: null pointer");
    return run(false, d->N, d->C, d->K, d->H, d->W, d->K, d->C, x, w, pm, thr, bias, y, ws, ws_bytes, (hipStream_t)stream, np);
}
static int dgrad_any(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, float *gx, void *ws,
                     size_t ws_bytes, void *stream, int np) {
    if (!cpg_conv2d_bf16_supported(d)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_dgrad_bf16: only 3x3 / stride 1 / pad 1 convolutions with >= 16 channels");
    CPG_REQUIRE(gy && w && gx, "cpg_conv2d_dgrad_bf16: null pointer");
    return run(true, d->N, d->K, d->C, d->H, d->W, d->K, d->C, gy, w, pm, thr, nullptr, gx, ws, ws_bytes, (hipStream_t)stream, np);
}
extern "C" int cpg_conv2d_fwd_bf16(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                                   float *y, void *ws, size_t ws_bytes, void *stream) {
    return fwd_any(d, x, w, pm, thr, bias, y, ws, ws_bytes, stream, 1);
}
extern "C" int cpg_conv2d_fwd_bf16x3(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr,
                                     const float *bias, float *y, void *ws, size_t ws_bytes, void *stream) {
    return fwd_any(d, x, w, pm, thr, bias, y, ws, ws_bytes, stream, 2);
}
extern "C" int cpg_conv2d_dgrad_bf16(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, float *gx,
                                     void *ws, size_t ws_bytes, void *stream) {
    return dgrad_any(d, gy, w, pm, thr, gx, ws, ws_bytes, stream, 1);
}
extern "C" int cpg_conv2d_dgrad_bf16x3(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, float *gx,
                                       void *ws, size_t ws_bytes, void *stream) {
    return dgrad_any(d, gy, w, pm, thr, gx, ws, ws_bytes, stream, 2);
}

// weight gradient on bf16 MFMA: every 3x3 s1 p1 layer with >= 16 channels on both sides (the 3 -> 64 stem stays on fp32)
extern "C" int32_t cpg_conv2d_wgrad_bf16_supported(const cpg_conv_desc *d) {
    return cpg_conv2d_bf16_supported(d) && (int64_t)d->H * d->W <= (1ll << 22) ? 1 : 0;
}

extern "C" size_t cpg_conv2d_wgrad_bf16_workspace_bytes(const cpg_conv_desc *d) {  // (the larger of the one- and two-plane plans)
    if (!cpg_conv2d_wgrad_bf16_supported(d)) return 0;
    switch (wpick(d)) {
        case 0: return std::max(wplan<B16G64>(d).ws_bytes, wplan<X3G64>(d).ws_bytes);
        case 1: return std::max(wplan<B16G32>(d).ws_bytes, wplan<X3G32>(d).ws_bytes);
        default: return std::max(wplan<B16G16>(d).ws_bytes, wplan<X3G16>(d).ws_bytes);
    }
}

static int wgrad_any(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr, float *gw,
                     float *gpm, void *ws, size_t ws_bytes, void *stream, int np) {
    if (!cpg_conv2d_wgrad_bf16_supported(d)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_wgrad_bf16: shape not supported");
    CPG_REQUIRE(x && gy && gw && w, "cpg_conv2d_wgrad_bf16: null pointer");
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    hipStream_t st = (hipStream_t)stream;
    switch (wpick(d) * 2 + (np == 2)) {
        case 0: return wlaunch<B16G64>(d, x, gy, ep, ws, ws_bytes, st);
        case 1: return wlaunch<X3G64>(d, x, gy, ep, ws, ws_bytes, st);
        case 2: return wlaunch<B16G32>(d, x, gy, ep, ws, ws_bytes, st);
        case 3: return wlaunch<X3G32>(d, x, gy, ep, ws, ws_bytes, st);
        case 4: return wlaunch<B16G16>(d, x, gy, ep, ws, ws_bytes, st);
        default: return wlaunch<X3G16>(d, x, gy, ep, ws, ws_bytes, st);
    }
}
extern "C" int cpg_conv2d_wgrad_bf16(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr,
                                     float *gw, float *gpm, void *ws, size_t ws_bytes, void *stream) {
    return wgrad_any(d, x, gy, w, pm, thr, gw, gpm, ws, ws_bytes, stream, 1);
}
extern "C" int cpg_conv2d_wgrad_bf16x3(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr,
                                       float *gw, float *gpm, void *ws, size_t ws_bytes, void *stream) {
    return wgrad_any(d, x, gy, w, pm, thr, gw, gpm, ws, ws_bytes, stream, 2);
}
