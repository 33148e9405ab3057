// Masked 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2, 3x3) on fp32 MFMA: forward and input gradient.
//
//   Y(2x2 tile) = A^T [ (G g G^T) .* (B^T d B) ] A        (d: 4x4 input patch, g: 3x3 filter; Lavin & Gray 2015)
//
// 16 multiplies per 2x2 output tile and (c, k) pair instead of 36: the contraction over input channels becomes 16
// independent GEMMs  M_p[k][t] = sum_c U_p[k][c] * V_p[c][t]  (p = transform-domain position, t = tile), 2.25x fewer MFMAs
// than the direct kernels of conv3x3.hip for the same result up to fp32 rounding (the transforms only add, subtract and
// halve).  DESIGN.md section 4.9 has the measurements.
//
//   k_wg_pack   U = G (W .* bin(piggymask)) G^T per (k, c), written in the order the conv kernel streams it:
//               Up[k block of 32][channel chunk of 4][p][k][c]  (one contiguous 8 KB record per block and chunk)
//   k_wg_fwd    block = 4 waves = 32 output channels x 64 tiles (256 output pixels), two blocks per CU.  Wave (tq, ph) owns
//               tiles tq*32..+31 and positions ph*8..+7: 8 accumulators of 32 x 32.  Per chunk of 4 input channels the
//               block stages U (two float4 per thread) and V: every thread gathers ONE 4x4 patch (tile = tid / 4,
//               channel = tid % 4) with 16 range-checked buffer loads (zero padding for free), transforms it in
//               registers (32 adds) and writes the 16 positions to LDS -- V[p][t][c], consecutive threads to consecutive
//               words.  Operands are ds_read_b64: lanes 0-31 take channels (0, 1), lanes 32-63 channels (2, 3) of the
//               chunk -- two k-steps of v_mfma_f32_32x32x2_f32 per read pair.  Two LDS stages, one barrier per chunk.
//               Epilogue: the output transform is linear, so each wave reduces its 8 positions to partial 2x2 outputs,
//               the two waves of a tile swap halves through LDS (row 0 of every tile is finished by ph = 0, row 1 by
//               ph = 1) and store float2 per lane (256 contiguous bytes per channel row and half-wave).
//   dgrad       the same kernel on gy with the filter transposed and spatially flipped (k_wg_pack's dgrad flavour).
#include <algorithm>
#include "igemm_core.h"

using namespace cpg;

namespace {

constexpr int WG_BK = 32;                     // output channels per block
constexpr int WG_T = 64;                      // tiles per block
constexpr int WG_CK = 4;                      // input channels per chunk
constexpr int WG_U = 16 * WG_BK * WG_CK;      // floats of U per chunk
constexpr int WG_V = 16 * WG_T * WG_CK;
constexpr int WG_STAGE = WG_U + WG_V;         // 6144 floats = 24 KB

struct WgGeom {
    int N, C, H, W, M;        // C: channels read, M: channels produced
    int th, tw;               // tiles per image column / row (H / 2, W / 2)
    int tiles_img;            // th * tw
    int64_t tiles_total;      // N * th * tw
    int nkb, nch;             // blocks of 32 output channels, chunks of 4 input channels
    int span;                 // images a block's 64 consecutive tiles can touch
};

// ------------------------------------------------------------------------------ weight transform
__global__ __launch_bounds__(256) void k_wg_pack(const float *__restrict__ w, const float *__restrict__ pm, float thr,
                                                 float *__restrict__ up, int K, int C, int M, int Cin, int nch, int dgrad) {
    // one thread per (kb, ch, kl, cl); writes 16 values at stride 128 floats
    const int64_t total = (int64_t)((M + WG_BK - 1) / WG_BK) * nch * WG_BK * WG_CK;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int cl = (int)(o % WG_CK);
        const int kl = (int)((o / WG_CK) % WG_BK);
        const int64_t rec = o / (WG_CK * WG_BK);            // kb * nch + ch
        const int ch = (int)(rec % nch), kb = (int)(rec / nch);
        const int m = kb * WG_BK + kl, c = ch * WG_CK + cl;  // produced / read channel
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
        float *dst = up + rec * WG_U + kl * WG_CK + cl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[(i * 4 + 0) * (WG_BK * WG_CK)] = t[i][0];
            dst[(i * 4 + 1) * (WG_BK * WG_CK)] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
            dst[(i * 4 + 2) * (WG_BK * WG_CK)] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
            dst[(i * 4 + 3) * (WG_BK * WG_CK)] = t[i][2];
        }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------ forward / input gradient
template <bool DGRAD, bool STATS>
__global__ __launch_bounds__(256, 2) void k_wg_fwd(WgGeom g, const float *__restrict__ x, const float *__restrict__ up,
                                                   const float *__restrict__ bias, float *__restrict__ y,
                                                   float *__restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float smem[2 * WG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tq = wave & 1, ph = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int HW = g.H * g.W;

    unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int kb = lb % g.nkb;
    const unsigned tb = lb / g.nkb;
    const int64_t t0 = (int64_t)tb * WG_T;                  // first tile of the block
    const int n0 = (int)(t0 / g.tiles_img);                 // first image the block touches

    // ---- staging item of this thread: the 4x4 patch of tile (tid / 4), channel (tid % 4) of the chunk ----
    constexpr int kOutOfRange = (int)0x80000000;
    int xoff[16];
    {
        const int64_t tg = t0 + (tid >> 2);
        const bool tv = tg < g.tiles_total;
        const int n = (int)(tg / g.tiles_img), r = (int)(tg % g.tiles_img);
        const int ty = r / g.tw, tx = r % g.tw;
        const int cbase = ((n - n0) * g.C + (tid & 3)) * HW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gh = 2 * ty - 1 + i, gw = 2 * tx - 1 + j;
                const bool ok = tv && (unsigned)gh < (unsigned)g.H && (unsigned)gw < (unsigned)g.W;
                xoff[i * 4 + j] = ok ? (cbase + gh * g.W + gw) * 4 : kOutOfRange;
            }
    }
    const int nimg_here = min(g.span, g.N - n0);
    const __amdgpu_buffer_rsrc_t srd_x =
        __builtin_amdgcn_make_buffer_rsrc((void *)(x + (int64_t)n0 * g.C * HW), 0, nimg_here * g.C * HW * 4, 0x00020000);
    const float *ubase = up + (int64_t)kb * g.nch * WG_U + tid * 4;

    // two register sets: the loads of chunk ch + 2 are issued at the top of chunk ch and consumed (transformed, written to LDS)
    // at the bottom of chunk ch + 1 -- a chunk is only 16 MFMAs per wave, one chunk of distance left the HBM / L2 latency exposed
    f32x4 ru[2][2];
    float rx[2][16];
    auto load_chunk = [&](int ch, f32x4 (&u)[2], float (&r)[16]) {
        u[0] = *reinterpret_cast<const f32x4 *>(ubase + (int64_t)ch * WG_U);
        u[1] = *reinterpret_cast<const f32x4 *>(ubase + (int64_t)ch * WG_U + 1024);
        const int soff = ch * WG_CK * HW * 4;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            r[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, xoff[e], soff, 0));
    };
    auto store_chunk = [&](float *stage, const f32x4 (&u)[2], const float (&r)[16]) {
        *reinterpret_cast<f32x4 *>(stage + tid * 4) = u[0];
        *reinterpret_cast<f32x4 *>(stage + tid * 4 + 1024) = u[1];
        // V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        float t[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0 * 4 + j] = r[0 * 4 + j] - r[2 * 4 + j];
            t[1 * 4 + j] = r[1 * 4 + j] + r[2 * 4 + j];
            t[2 * 4 + j] = r[2 * 4 + j] - r[1 * 4 + j];
            t[3 * 4 + j] = r[1 * 4 + j] - r[3 * 4 + j];
        }
        float *v = stage + WG_U + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[(i * 4 + 0) * (WG_T * WG_CK)] = t[i * 4 + 0] - t[i * 4 + 2];
            v[(i * 4 + 1) * (WG_T * WG_CK)] = t[i * 4 + 1] + t[i * 4 + 2];
            v[(i * 4 + 2) * (WG_T * WG_CK)] = t[i * 4 + 2] - t[i * 4 + 1];
            v[(i * 4 + 3) * (WG_T * WG_CK)] = t[i * 4 + 1] - t[i * 4 + 3];
        }
    };

    // operand lane bases (floats): position p = ph * 8 + pp
    const int a_base = (ph * 8) * (WG_BK * WG_CK) + li * WG_CK + lh * 2;
    const int b_base = WG_U + (ph * 8) * (WG_T * WG_CK) + (tq * 32 + li) * WG_CK + lh * 2;

    f32x16 acc[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pp][e] = 0.0f;

    const int last = g.nch - 1;
    load_chunk(0, ru[0], rx[0]);
    load_chunk(min(1, last), ru[1], rx[1]);
    store_chunk(smem, ru[0], rx[0]);
    __syncthreads();
    // chunk ch computes from stage ch & 1; register set (ch + 1) & 1 holds chunk ch + 1, set ch & 1 receives chunk ch + 2
    auto chunk = [&](int ch, f32x4 (&u_in)[2], float (&r_in)[16], f32x4 (&u_next)[2], float (&r_next)[16]) {
        const float *st = smem + (ch & 1) * WG_STAGE;
        float *other = smem + ((ch + 1) & 1) * WG_STAGE;
        load_chunk(min(ch + 2, last), u_in, r_in);
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const f32x2 a = *reinterpret_cast<const f32x2 *>(st + a_base + pp * (WG_BK * WG_CK));
            const f32x2 b = *reinterpret_cast<const f32x2 *>(st + b_base + pp * (WG_T * WG_CK));
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc[pp], 0, 0, 0);
        }
        store_chunk(other, u_next, r_next);
        __syncthreads();
    };
    for (int ch = 0; ch < g.nch; ch += 2) {
        chunk(ch, ru[0], rx[0], ru[1], rx[1]);
        if (ch + 1 < g.nch) chunk(ch + 1, ru[1], rx[1], ru[0], rx[0]);
    }

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
    // (the main loop's last barrier freed the LDS) exchange buffer [tq][writer ph][b * 16 + e][lane]
    float *xch = smem;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[((tq * 2 + ph) * 32 + b * 16 + e) * 64 + lane] = give[b][e];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) own[b][e] += xch[((tq * 2 + (ph ^ 1)) * 32 + b * 16 + e) * 64 + lane];

    const int64_t tg = t0 + tq * 32 + li;
    const bool tv = tg < g.tiles_total;
    const int n = (int)(tg / g.tiles_img), r = (int)(tg % g.tiles_img);
    const int ty = r / g.tw, tx = r % g.tw;
    float *yout = y + ((int64_t)n * g.M) * HW + (2 * ty + ph) * g.W + 2 * tx;
    float s1[16], s2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int co = kb * WG_BK + (e & 3) + 8 * (e >> 2) + 4 * lh;
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
                red[(wave * WG_BK + cl) * 2 + 0] = s1[e];
                red[(wave * WG_BK + cl) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        if (tid < WG_BK && kb * WG_BK + tid < g.M) {
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                a += red[(w2 * WG_BK + tid) * 2 + 0];
                b += red[(w2 * WG_BK + tid) * 2 + 1];
            }
            const unsigned ntb = gridDim.x / g.nkb;
            float *dst = stats + ((int64_t)(kb * WG_BK + tid) * ntb + tb) * 2;
            dst[0] = a;
            dst[1] = b;
        }
    }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

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

extern "C" size_t cpg_conv3x3_wino_pack_bytes(int c_read, int m) {
    return (size_t)pad_to(m, WG_BK) / WG_BK * (pad_to(c_read, WG_CK) / WG_CK) * WG_U * sizeof(float);
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
    WgGeom g;
    g.N = N, g.C = c_read, g.H = H, g.W = W, g.M = m;
    g.th = H / 2, g.tw = W / 2, g.tiles_img = g.th * g.tw;
    g.tiles_total = (int64_t)N * g.tiles_img;
    g.nkb = pad_to(m, WG_BK) / WG_BK, g.nch = pad_to(c_read, WG_CK) / WG_CK;
    g.span = (WG_T + g.tiles_img - 1) / g.tiles_img + 1;
    float *up = (float *)ws;
    hipLaunchKernelGGL(k_wg_pack, dim3(stream_grid((int64_t)g.nkb * g.nch * WG_BK * WG_CK, 256)), dim3(256), 0, stream, w, pm, thr, up,
                       K, C, m, c_read, g.nch, dgrad ? 1 : 0);
    const int64_t blocks = (int64_t)cpg_conv3x3_wino_tiles(N, H, W) * g.nkb;
    if (blocks > 0x7FFFFFFFll) return fail(CPG_E_UNSUPPORTED, "%s: grid too large", what);
    if (dgrad)
        hipLaunchKernelGGL((k_wg_fwd<true, false>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, up, bias, y, nullptr);
    else if (stats != nullptr)
        hipLaunchKernelGGL((k_wg_fwd<false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, up, bias, y, stats);
    else
        hipLaunchKernelGGL((k_wg_fwd<false, false>), dim3((unsigned)blocks), dim3(256), 0, stream, g, x, up, bias, y, nullptr);
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}
