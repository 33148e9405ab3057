// Generic masked conv2d on fp32 MFMA: forward, input-gradient and weight-gradient of
// models/layers.py:108-109 for ANY kernel size / stride / padding / dilation (groups == 1).
// This is the shape-complete path (ResNet 1x1, 7x7 s2, 3x3 s2; SphereNet 3x3 s2) and the
// fallback for the specialised 3x3 s1 p1 kernels of conv3x3.hip.
//
// Implicit GEMM, no im2col buffer: the B operand loader gathers the input window straight
// from the NCHW tensor into the k-major LDS tile (lanes run along output pixels, so the
// gathers are coalesced for stride 1); the A loader applies W * bin(piggymask) on the fly.
//
//   fwd  : D[co][p]        = sum_{ci,r,s} Weff[co][ci][r][s] * x[n_p][ci][ih][iw]
//   dgrad: D[ci][q]        = sum_{co,r,s} Weff[co][ci][r][s] * gy[n_q][co][oh][ow]
//   wgrad: D[co][(ci,r,s)] = sum_{p}      gy[n_p][co][p]      * x[n_p][ci][ih][iw]     (split-K over p)
#include <algorithm>
#include "igemm_core.h"

using namespace cpg;

namespace {

struct ConvGeom {
    int N, C, H, W, K, R, S, sh, sw, ph, pw, dh, dw, OH, OW;
};

// Input gradient of a STRIDED convolution, one residue class at a time.  gx[h][w] only receives taps r = (h + ph) mod sh
// (+ sh, + 2 sh, ...): the positions of one residue class (rho, sigma) form the sub-grid h = h_start + sh*i, w = w_start +
// sw*j, and on it the gradient is a dense, stride-1 transposed conv of gy with the sub-kernel W[..][rho + sh*r'][sigma +
// sw*s'].  k_conv_dgrad runs that sub-problem with ConvGeom = {H, W := sub-grid extent; R, S := sub-kernel extent; stride
// 1; ph, pw := the class's offset} and this struct says where its weights and outputs really live.  (The single-launch
// form evaluated all R*S taps at every position and threw away (1 - 1/(sh*sw)) of the MFMAs and gathers: 8-13 TFLOP/s.)
struct DgradSub {
    int Hfull, Wfull;          // gx plane
    int h_start, w_start, sh, sw;
    int RSfull, Sfull, r0, s0; // weight layout [K][C][R][S] of the full kernel; first tap of the class
};

// ---------------------------------------------------------------------------------- loaders
// conv fwd B: B[k=(ci,r,s)][j=p]
template <int BN, int BK>
struct FwdBLoader {
    static constexpr int N = BN * BK / 256;
    static constexpr int KSTEP = 256 / BN;
    const float *x;
    int C, H, W, R, S, dh, dw, Kg;
    int ih0, iw0, t_j, t_k;
    int64_t xbase;
    bool jvalid;
    __device__ __forceinline__ void init(const float *x_, const ConvGeom &g, int64_t p0, int64_t P) {
        x = x_; C = g.C; H = g.H; W = g.W; R = g.R; S = g.S; dh = g.dh; dw = g.dw; Kg = g.C * g.R * g.S;
        t_j = threadIdx.x % BN;
        t_k = threadIdx.x / BN;
        const int64_t p = p0 + t_j;
        jvalid = p < P;
        const int ohw = g.OH * g.OW;
        const int n = jvalid ? (int)(p / ohw) : 0;
        const int q = jvalid ? (int)(p % ohw) : 0;
        const int oh = q / g.OW, ow = q % g.OW;
        ih0 = oh * g.sh - g.ph;
        iw0 = ow * g.sw - g.pw;
        xbase = (int64_t)n * C * H * W;
    }
    unsigned okmask;
    __device__ __forceinline__ void fetch(int kt, float (&r)[N]) {
        const int RS = R * S;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int k = kt * BK + t_k + KSTEP * i;
            const int ci = k / RS, rs = k - ci * RS;
            const int rr = rs / S, ss = rs - rr * S;
            const int ih = ih0 + rr * dh, iw = iw0 + ss * dw;
            const bool ok = jvalid && k < Kg && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            okmask |= (ok ? 1u : 0u) << i;
            r[i] = x[ok ? xbase + ((int64_t)ci * H + ih) * W + iw : 0];       // unconditional load, select in put()
        }
    }
    __device__ __forceinline__ void put(const float (&r)[N], float *lds) {
#pragma unroll
        for (int i = 0; i < N; ++i) lds[(t_k + KSTEP * i) * (BN + 1) + t_j] = ((okmask >> i) & 1u) ? r[i] : 0.0f;
    }
};

// conv dgrad B: B[k=(co,r,s)][j=q=(n,h,w)] = gy[n][co][(h+ph-r*dh)/sh][(w+pw-s*dw)/sw] when divisible & in range
template <int BN, int BK>
struct DgradBLoader {
    static constexpr int N = BN * BK / 256;
    static constexpr int KSTEP = 256 / BN;
    const float *gy;
    int K, OH, OW, R, S, dh, dw, sh, sw, Kg;
    int th0, tw0, t_j, t_k;
    int64_t base;
    bool jvalid;
    __device__ __forceinline__ void init(const float *gy_, const ConvGeom &g, int64_t q0, int64_t Q) {
        gy = gy_; K = g.K; OH = g.OH; OW = g.OW; R = g.R; S = g.S; dh = g.dh; dw = g.dw; sh = g.sh; sw = g.sw;
        Kg = g.K * g.R * g.S;
        t_j = threadIdx.x % BN;
        t_k = threadIdx.x / BN;
        const int64_t q = q0 + t_j;
        jvalid = q < Q;
        const int hw = g.H * g.W;
        const int n = jvalid ? (int)(q / hw) : 0;
        const int rem = jvalid ? (int)(q % hw) : 0;
        th0 = rem / g.W + g.ph;
        tw0 = rem % g.W + g.pw;
        base = (int64_t)n * K * OH * OW;
    }
    unsigned okmask;
    __device__ __forceinline__ void fetch(int kt, float (&r)[N]) {
        const int RS = R * S;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int k = kt * BK + t_k + KSTEP * i;
            const int co = k / RS, rs = k - co * RS;
            const int rr = rs / S, ss = rs - rr * S;
            const int th = th0 - rr * dh, tw = tw0 - ss * dw;
            bool ok = jvalid && k < Kg && th >= 0 && tw >= 0;
            int oh = th, ow = tw;
            if (sh != 1) { oh = th / sh; ok = ok && (oh * sh == th); }
            if (sw != 1) { ow = tw / sw; ok = ok && (ow * sw == tw); }
            ok = ok && oh < OH && ow < OW;
            okmask |= (ok ? 1u : 0u) << i;
            r[i] = gy[ok ? base + ((int64_t)co * OH + oh) * OW + ow : 0];
        }
    }
    __device__ __forceinline__ void put(const float (&r)[N], float *lds) {
#pragma unroll
        for (int i = 0; i < N; ++i) lds[(t_k + KSTEP * i) * (BN + 1) + t_j] = ((okmask >> i) & 1u) ? r[i] : 0.0f;
    }
};

// conv dgrad A: A[k=(co,r,s)][m=ci] = Weff[co][ci][r][s]
template <int BM, int BK>
struct DgradALoader {
    static constexpr int N = BM * BK / 256;
    static constexpr int MSTEP = 256 / BK;
    const float *w, *pm;
    float thr;
    int C, RS, S, Kg, m0, t_k, t_m;
    DgradSub sub;
    unsigned okmask;
    float rp[N];
    __device__ __forceinline__ void init(const float *w_, const float *pm_, float thr_, const ConvGeom &g, const DgradSub &sub_,
                                         int m0_) {
        w = w_; pm = pm_; thr = thr_; C = g.C; RS = g.R * g.S; S = g.S; Kg = g.K * RS; m0 = m0_; sub = sub_;
        t_k = threadIdx.x % BK;
        t_m = threadIdx.x / BK;
    }
    __device__ __forceinline__ void fetch(int kt, float (&r)[N]) {
        const int k = kt * BK + t_k;
        const int co = k / RS, rs_sub = k - co * RS;
        const int rr = rs_sub / S, ss = rs_sub - rr * S;
        const int rs = (sub.r0 + sub.sh * rr) * sub.Sfull + sub.s0 + sub.sw * ss;      // tap of the full kernel
        okmask = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int ci = m0 + t_m + MSTEP * i;
            const bool ok = k < Kg && ci < C;
            const int64_t off = ok ? ((int64_t)co * C + ci) * sub.RSfull + rs : 0;
            okmask |= (ok ? 1u : 0u) << i;
            r[i] = w[off];
            if (pm != nullptr) rp[i] = pm[off];            // wave-uniform condition
        }
    }
    __device__ __forceinline__ void put(const float (&r)[N], float *lds) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float v = r[i];
            if (pm != nullptr) v *= binarize(rp[i], thr);
            lds[t_k * (BM + 1) + t_m + MSTEP * i] = ((okmask >> i) & 1u) ? v : 0.0f;
        }
    }
};

// conv wgrad A: A[k=p][m=co] = gy[n_p][co][q_p]
template <int BM, int BK>
struct WgradALoader {
    static constexpr int N = BM * BK / 256;
    static constexpr int MSTEP = 256 / BK;
    const float *gy;
    int K, OHW, m0, t_k, t_m;
    unsigned okmask;
    int64_t P;
    __device__ __forceinline__ void init(const float *gy_, const ConvGeom &g, int m0_) {
        gy = gy_; K = g.K; OHW = g.OH * g.OW; m0 = m0_; P = (int64_t)g.N * OHW;
        t_k = threadIdx.x % BK;
        t_m = threadIdx.x / BK;
    }
    __device__ __forceinline__ void fetch(int kt, float (&r)[N]) {
        const int64_t p = (int64_t)kt * BK + t_k;
        const bool pv = p < P;
        const int n = pv ? (int)(p / OHW) : 0;
        const int q = pv ? (int)(p % OHW) : 0;
        const int64_t b = (int64_t)n * K * OHW + q;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int co = m0 + t_m + MSTEP * i;
            const bool ok = pv && co < K;
            okmask |= (ok ? 1u : 0u) << i;
            r[i] = gy[ok ? b + (int64_t)co * OHW : 0];
        }
    }
    __device__ __forceinline__ void put(const float (&r)[N], float *lds) {
#pragma unroll
        for (int i = 0; i < N; ++i) lds[t_k * (BM + 1) + t_m + MSTEP * i] = ((okmask >> i) & 1u) ? r[i] : 0.0f;
    }
};

// conv wgrad B: B[k=p][j=(ci,r,s)] = x[n_p][ci][oh*sh-ph+r*dh][ow*sw-pw+s*dw]
template <int BN, int BK>
struct WgradBLoader {
    static constexpr int N = BN * BK / 256;
    static constexpr int JSTEP = 256 / BK;
    const float *x;
    int C, H, W, OW, OHW, sh, sw, ph, pw, t_k, t_j;
    int64_t P;
    unsigned okmask;
    int joff[N];      // ci*H*W + r*dh*W + s*dw, or -1 when j is out of range
    int jrs[N];       // (r*dh) << 16 | (s*dw)
    __device__ __forceinline__ void init(const float *x_, const ConvGeom &g, int j0) {
        x = x_; C = g.C; H = g.H; W = g.W; OW = g.OW; OHW = g.OH * g.OW; sh = g.sh; sw = g.sw; ph = g.ph; pw = g.pw;
        P = (int64_t)g.N * OHW;
        t_k = threadIdx.x % BK;
        t_j = threadIdx.x / BK;
        const int RS = g.R * g.S, J = g.C * RS;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = j0 + t_j + JSTEP * i;
            if (j < J) {
                const int ci = j / RS, rs = j - ci * RS;
                const int rr = (rs / g.S) * g.dh, ss = (rs % g.S) * g.dw;
                joff[i] = ci * H * W + rr * W + ss;
                jrs[i] = (rr << 16) | ss;
            } else {
                joff[i] = -1;
                jrs[i] = 0;
            }
        }
    }
    __device__ __forceinline__ void fetch(int kt, float (&r)[N]) {
        const int64_t p = (int64_t)kt * BK + t_k;
        const bool pv = p < P;
        const int n = pv ? (int)(p / OHW) : 0;
        const int q = pv ? (int)(p % OHW) : 0;
        const int oh = q / OW, ow = q - oh * OW;
        const int ih0 = oh * sh - ph, iw0 = ow * sw - pw;
        const int64_t b = (int64_t)n * C * H * W + (int64_t)ih0 * W + iw0;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int ih = ih0 + (jrs[i] >> 16), iw = iw0 + (jrs[i] & 0xFFFF);
            const bool ok = pv && joff[i] >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            okmask |= (ok ? 1u : 0u) << i;
            r[i] = x[ok ? b + joff[i] : 0];
        }
    }
    __device__ __forceinline__ void put(const float (&r)[N], float *lds) {
#pragma unroll
        for (int i = 0; i < N; ++i) lds[t_k * (BN + 1) + t_j + JSTEP * i] = ((okmask >> i) & 1u) ? r[i] : 0.0f;
    }
};

// ---------------------------------------------------------------------------------- kernels
// out-of-tile guards live in the loaders (zeros) and in the store lambdas.
template <class Cfg>
__global__ __launch_bounds__(256) void k_conv_fwd(ConvGeom g, const float *__restrict__ x, const float *__restrict__ w,
                                                  const float *__restrict__ pm, float thr, Epilogue ep, int tiles_m) {
    __shared__ float smem[Cfg::SMEM_FLOATS];
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lb % tiles_m, tn = lb / tiles_m;
    const int Kg = g.C * g.R * g.S;
    const int ohw = g.OH * g.OW;
    const int64_t P = (int64_t)g.N * ohw;
    const int m0 = tm * Cfg::BM;
    const int64_t p0 = (int64_t)tn * Cfg::BN;
    DenseLoader<Cfg::BM, Cfg::BK, true, true> la;      // pm == nullptr handled below
    la.init(w, pm, thr, Kg, m0, g.K, Kg);
    FwdBLoader<Cfg::BN, Cfg::BK> lbB;
    lbB.init(x, g, p0, P);
    f32x16 acc[Cfg::FM][Cfg::FN];
    const int nkt = (Kg + Cfg::BK - 1) / Cfg::BK;
    if (pm != nullptr) {
        igemm_mainloop<Cfg>(la, lbB, 0, nkt, smem, acc);
    } else {
        DenseLoader<Cfg::BM, Cfg::BK, true, false> la0;
        la0.init(w, nullptr, thr, Kg, m0, g.K, Kg);
        igemm_mainloop<Cfg>(la0, lbB, 0, nkt, smem, acc);
    }
    const int K = g.K;
    int64_t colbase[Cfg::FN];           // n*K*ohw + q per fragment column, -1 when past the end
    col_setup<Cfg>(colbase, [&](int j) -> int64_t {
        const int64_t p = p0 + j;
        if (p >= P) return -1;
        const int n = (int)(p / ohw), q = (int)(p - (int64_t)n * ohw);
        return (int64_t)n * K * ohw + q;
    });
    for_each_acc<Cfg>(acc, [&](int m, int j, int fn, float v) {
        const int co = m0 + m;
        if (co < K && colbase[fn] >= 0) epilogue_store(ep, colbase[fn] + (int64_t)co * ohw, v);
    });
}

template <class Cfg>
__global__ __launch_bounds__(256) void k_conv_dgrad(ConvGeom g, DgradSub sub, const float *__restrict__ gy,
                                                    const float *__restrict__ w, const float *__restrict__ pm, float thr,
                                                    Epilogue ep, int tiles_m) {
    __shared__ float smem[Cfg::SMEM_FLOATS];
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lb % tiles_m, tn = lb / tiles_m;
    const int Kg = g.K * g.R * g.S;
    const int hw = g.H * g.W;
    const int64_t Q = (int64_t)g.N * hw;
    const int m0 = tm * Cfg::BM;
    const int64_t q0 = (int64_t)tn * Cfg::BN;
    DgradALoader<Cfg::BM, Cfg::BK> la;
    la.init(w, pm, thr, g, sub, m0);
    DgradBLoader<Cfg::BN, Cfg::BK> lbB;
    lbB.init(gy, g, q0, Q);
    f32x16 acc[Cfg::FM][Cfg::FN];
    igemm_mainloop<Cfg>(la, lbB, 0, (Kg + Cfg::BK - 1) / Cfg::BK, smem, acc);
    const int C = g.C;
    const int64_t hw_full = (int64_t)sub.Hfull * sub.Wfull;
    int64_t colbase[Cfg::FN];
    col_setup<Cfg>(colbase, [&](int j) -> int64_t {
        const int64_t q = q0 + j;
        if (q >= Q) return -1;
        const int n = (int)(q / hw), rem = (int)(q - (int64_t)n * hw);
        const int i = rem / g.W, j2 = rem - i * g.W;
        return (int64_t)n * C * hw_full + (int64_t)(sub.h_start + sub.sh * i) * sub.Wfull + sub.w_start + sub.sw * j2;
    });
    for_each_acc<Cfg>(acc, [&](int m, int j, int fn, float v) {
        const int ci = m0 + m;
        if (ci < C && colbase[fn] >= 0) epilogue_store(ep, colbase[fn] + (int64_t)ci * hw_full, v);
    });
}

// split-K over output pixels; blockIdx.y = split.  With nsplit > 1 raw partials go to `part`
// ([nsplit][K*C*R*S]) and k_split_reduce (igemm_core.h) applies the epilogue.
template <class Cfg>
__global__ __launch_bounds__(256) void k_conv_wgrad(ConvGeom g, const float *__restrict__ x, const float *__restrict__ gy,
                                                    Epilogue ep, float *__restrict__ part, int tiles_m, int kt_per_split) {
    __shared__ float smem[Cfg::SMEM_FLOATS];
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lb % tiles_m, tn = lb / tiles_m;
    const int J = g.C * g.R * g.S;
    const int64_t P = (int64_t)g.N * g.OH * g.OW;
    const int nkt = (int)((P + Cfg::BK - 1) / Cfg::BK);
    const int m0 = tm * Cfg::BM, j0 = tn * Cfg::BN;
    WgradALoader<Cfg::BM, Cfg::BK> la;
    la.init(gy, g, m0);
    WgradBLoader<Cfg::BN, Cfg::BK> lbB;
    lbB.init(x, g, j0);
    f32x16 acc[Cfg::FM][Cfg::FN];
    const int kt0 = blockIdx.y * kt_per_split;
    const int kt1 = min(nkt, kt0 + kt_per_split);
    igemm_mainloop<Cfg>(la, lbB, kt0, kt1, smem, acc);
    const int K = g.K;
    const int64_t out_elems = (int64_t)K * J;
    float *dst = part ? part + (int64_t)blockIdx.y * out_elems : nullptr;
    for_each_acc<Cfg>(acc, [&](int m, int j, int, float v) {
        const int co = m0 + m, jj = j0 + j;
        if (co < K && jj < J) {
            const int64_t e = (int64_t)co * J + jj;
            if (dst) dst[e] = v;
            else epilogue_store(ep, e, v);
        }
    });
}

// ---------------------------------------------------------------------------------- linear
// fwd  : y[b][o]  = sum_i x[b][i] * Weff[o][i]       A = x (KC), B = W (KC)      split-K over i
// dgrad: gx[b][i] = sum_o gy[b][o] * Weff[o][i]      A = gy (KC), B = W (RC)     split-K over o
// wgrad: gw[o][i] = sum_b gy[b][o] * x[b][i]         A = gy (RC), B = x (RC)     split-K over b
template <class Cfg, bool A_KC, bool B_KC, int MASK_SIDE /*0 none,2 B*/>
__global__ __launch_bounds__(256) void k_gemm(const float *__restrict__ A, int64_t lda, const float *__restrict__ B,
                                              int64_t ldb, const float *__restrict__ pm, float thr, int M, int Nn, int Kd,
                                              Epilogue ep, float *__restrict__ part, int tiles_m, int kt_per_split) {
    __shared__ float smem[Cfg::SMEM_FLOATS];
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lb % tiles_m, tn = lb / tiles_m;
    const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
    const int nkt = (Kd + Cfg::BK - 1) / Cfg::BK;
    const int kt0 = blockIdx.y * kt_per_split;
    const int kt1 = min(nkt, kt0 + kt_per_split);
    DenseLoader<Cfg::BM, Cfg::BK, A_KC, false> la;
    la.init(A, nullptr, thr, lda, m0, M, Kd);
    f32x16 acc[Cfg::FM][Cfg::FN];
    if (MASK_SIDE == 2 && pm != nullptr) {
        DenseLoader<Cfg::BN, Cfg::BK, B_KC, true> lbm;
        lbm.init(B, pm, thr, ldb, n0, Nn, Kd);
        igemm_mainloop<Cfg>(la, lbm, kt0, kt1, smem, acc);
    } else {
        DenseLoader<Cfg::BN, Cfg::BK, B_KC, false> lb0;
        lb0.init(B, nullptr, thr, ldb, n0, Nn, Kd);
        igemm_mainloop<Cfg>(la, lb0, kt0, kt1, smem, acc);
    }
    const int64_t out_elems = (int64_t)M * Nn;
    float *dst = part ? part + (int64_t)blockIdx.y * out_elems : nullptr;
    for_each_acc<Cfg>(acc, [&](int m, int j, int, float v) {
        const int mm = m0 + m, jj = n0 + j;
        if (mm < M && jj < Nn) {
            const int64_t e = (int64_t)mm * Nn + jj;
            if (dst) dst[e] = v;
            else epilogue_store(ep, e, v);
        }
    });
}

// gb[c] = sum over n, q of gy[n][c][q]   (conv: rows = N, inner = OH*OW; linear: inner = 1 with stride C)
__global__ __launch_bounds__(256) void k_bias_grad(const float *__restrict__ gy, float *__restrict__ gb, int rows, int C,
                                                   int inner) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float s = 0.0f;
    const int64_t total = (int64_t)rows * inner;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int64_t n = i / inner, q = i - n * inner;
        s += gy[(n * C + c) * inner + q];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) gb[c] = red[0];
}

// conv bias gradient in two deterministic stages: grid (K, slices) partial sums over image slices, then one thread per
// channel.  (One block per channel with a 64-bit (n, q) split per element took 340 us per SphereNet layer -- 10 % of its
// training step -- with 64..512 blocks on 256 CUs.)
__global__ __launch_bounds__(256) void k_conv_bias_partial(const float *__restrict__ gy, float *__restrict__ partial, int N, int C,
                                                           int inner, int imgs_per_slice) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const int n0 = s * imgs_per_slice, n1 = min(N, n0 + imgs_per_slice);
    const bool vec = (inner & 3) == 0 && (((uintptr_t)gy) & 15) == 0;
    float acc = 0.0f;
    for (int n = n0; n < n1; ++n) {
        const float *plane = gy + ((int64_t)n * C + c) * inner;
        if (vec) {
            for (int j = threadIdx.x; j < (inner >> 2); j += 256) {
                const float4 v = reinterpret_cast<const float4 *>(plane)[j];
                acc += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int j = threadIdx.x; j < inner; j += 256) acc += plane[j];
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)c * gridDim.y + s] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void k_conv_bias_final(const float *__restrict__ partial, float *__restrict__ gb, int C, int slices) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.0f;
    for (int k = 0; k < slices; ++k) s += partial[(int64_t)c * slices + k];
    gb[c] = s;
}
inline int bias_slices(int N, int K) { return std::max(1, std::min(N, (8 * kCUs + K - 1) / K)); }
inline size_t bias_ws_bytes(int N, int K) { return (size_t)K * bias_slices(N, K) * sizeof(float); }
// the workspace is free again once the weight-gradient reduce has been enqueued (same stream)
static void launch_conv_bias_grad(const float *gy, float *gb, int N, int K, int inner, void *ws, hipStream_t stream) {
    const int slices = bias_slices(N, K), ips = (N + slices - 1) / slices;
    const int used = (N + ips - 1) / ips;
    hipLaunchKernelGGL(k_conv_bias_partial, dim3((unsigned)K, (unsigned)used), dim3(256), 0, stream, gy, (float *)ws, N, K, inner, ips);
    hipLaunchKernelGGL(k_conv_bias_final, dim3((unsigned)((K + 63) / 64)), dim3(64), 0, stream, (const float *)ws, gb, K, used);
}

using CfgA = TileCfg<128, 128, 16, 2, 2>;
using CfgB = TileCfg<64, 256, 16, 1, 4>;

int make_geom(const cpg_conv_desc *d, ConvGeom &g) {
    CPG_REQUIRE(d != nullptr, "conv: null descriptor");
    CPG_REQUIRE(d->N > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->K > 0 && d->R > 0 && d->S > 0, "conv: non-positive dimension");
    CPG_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
                "conv: bad stride/dilation/padding");
    if (d->groups != 1) return fail(CPG_E_UNSUPPORTED, "conv: groups=%d not implemented (all CPG configs use 1)", d->groups);
    g = ConvGeom{d->N, d->C, d->H, d->W, d->K, d->R, d->S, d->stride_h, d->stride_w, d->pad_h, d->pad_w, d->dil_h, d->dil_w, 0, 0};
    g.OH = (d->H + 2 * d->pad_h - d->dil_h * (d->R - 1) - 1) / d->stride_h + 1;
    g.OW = (d->W + 2 * d->pad_w - d->dil_w * (d->S - 1) - 1) / d->stride_w + 1;
    CPG_REQUIRE(g.OH > 0 && g.OW > 0, "conv: empty output");
    CPG_REQUIRE((int64_t)d->C * d->H * d->W < (1ll << 31) && (int64_t)d->K * g.OH * g.OW < (1ll << 31) &&
                    (int64_t)d->C * d->R * d->S < (1ll << 31),
                "conv: per-image tensor exceeds 2^31 elements");
    return CPG_OK;
}

// wgrad split-K plan: enough blocks to fill the chip (~4 per CU) without shrinking a split
// below 8 K tiles; the partial buffer is nsplit * K*C*R*S floats.
void wgrad_plan(int64_t tiles, int nkt, int &nsplit, int &kt_per_split) {
    int64_t want = (4 * kCUs + tiles - 1) / tiles;
    int64_t max_by_k = (nkt + 7) / 8;
    if (want > max_by_k) want = max_by_k;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    kt_per_split = (int)((nkt + want - 1) / want);
    nsplit = (nkt + kt_per_split - 1) / kt_per_split;
}

template <class Cfg>
void conv_wgrad_tiles(const ConvGeom &g, int &tiles_m, int &tiles_n) {
    tiles_m = (g.K + Cfg::BM - 1) / Cfg::BM;
    tiles_n = (g.C * g.R * g.S + Cfg::BN - 1) / Cfg::BN;
}

}  // namespace

// The specialised 3x3 kernels (conv3x3.hip) take over when they support the shape.
extern "C" int cpg_conv3x3_supported(const cpg_conv_desc *d);
int cpg_conv3x3_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                    float *y, void *ws, size_t ws_bytes, hipStream_t stream);
int cpg_conv3x3_dgrad(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, float *gx, void *ws,
                      size_t ws_bytes, hipStream_t stream);
size_t cpg_conv3x3_pack_workspace(const cpg_conv_desc *d);
size_t cpg_conv3x3_wgrad_workspace(const cpg_conv_desc *d);
int cpg_conv3x3_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr,
                      float *gw, float *gpm, void *ws, size_t ws_bytes, hipStream_t stream);
// the 3x3 weight-gradient kernel owns a 64-wide input-channel tile; <= 3 channels (the VGG stem) have their own
// HBM-streaming kernel, 4..15 channels go to the generic kernel whose (ci, tap) column packing wastes less MFMA
int cpg_conv3x3_bnstats_tiles(const cpg_conv_desc *d);
int cpg_conv3x3_fwd_bnstats(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                            float *y, float *stats, void *ws, size_t ws_bytes, hipStream_t stream);
int cpg_conv3x3_fwd_bn_eval(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                            const float *gamma, const float *beta, const float *mean, const float *var, float eps, int relu, float *y,
                            int32_t *skip_stats, void *ws, size_t ws_bytes, hipStream_t stream);
int cpg_conv3x3_dgrad_bnbwd_tiles(const cpg_conv_desc *d);
int cpg_conv3x3_dgrad_bnbwd(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, const float *ypre,
                            const float *gamma, const float *beta, const float *mean, const float *invstd, float *gx, float *partials,
                            void *ws, size_t ws_bytes, hipStream_t stream);
// ... and the pointwise kernels (pointwise.hip) for 1x1 convolutions (forward and input gradient)
extern "C" int cpg_conv1x1_supported(const cpg_conv_desc *d);
size_t cpg_conv1x1_pack_workspace(const cpg_conv_desc *d);
int cpg_conv1x1_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias,
                    float *y, void *ws, size_t ws_bytes, hipStream_t stream, float *stats = nullptr);
int cpg_conv1x1_bnstats_tiles(const cpg_conv_desc *d);
int cpg_conv1x1_dgrad(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, float *gx, void *ws,
                      size_t ws_bytes, hipStream_t stream, const float *addend = nullptr);
extern "C" int cpg_conv1x1_wgrad_supported(const cpg_conv_desc *d);
size_t cpg_conv1x1_wgrad_workspace(const cpg_conv_desc *d);
int cpg_conv1x1_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr,
                      float *gw, float *gpm, void *ws, size_t ws_bytes, hipStream_t stream);
// ... the 3x3 / stride 2 / pad 1 class (conv3x3.hip: k_c3_fwd's strided tiles, k_c3s2_dgrad, k_c3_wgrad's strided units)
extern "C" int cpg_conv3x3s2_supported(const cpg_conv_desc *d);
size_t cpg_conv3x3s2_pack_workspace(const cpg_conv_desc *d);
int cpg_conv3x3s2_bnstats_tiles(const cpg_conv_desc *d);
int cpg_conv3x3s2_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias, float *y,
                      float *stats, void *ws, size_t ws_bytes, hipStream_t stream);
int cpg_conv3x3s2_dgrad(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, float *gx, void *ws,
                        size_t ws_bytes, hipStream_t stream);
size_t cpg_conv3x3s2_wgrad_workspace(const cpg_conv_desc *d);
int cpg_conv3x3s2_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr, float *gw,
                        float *gpm, void *ws, size_t ws_bytes, hipStream_t stream);
// ... and the strided image stems (conv_stem_s2.hip: 7x7 s2 p3 and 3x3 s2 p1 from <= 3 channels to 64)
extern "C" int cpg_conv_stem2_ok(const cpg_conv_desc *d);
int cpg_conv_stem2_tiles(const cpg_conv_desc *d);
int cpg_conv_stem2_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr, const float *bias, float *y,
                       float *stats, hipStream_t stream);
size_t cpg_conv_stem2_wgrad_workspace(const cpg_conv_desc *d);
int cpg_conv_stem2_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr, float *gw,
                         float *gpm, void *ws, size_t ws_bytes, hipStream_t stream);
static inline bool use_c3_wgrad(const cpg_conv_desc *d) { return cpg_conv3x3_supported(d) && (d->C >= 16 || d->C <= 3); }

extern "C" size_t cpg_conv2d_workspace_bytes(const cpg_conv_desc *d) {
    ConvGeom g;
    if (make_geom(d, g) != CPG_OK) return 0;
    size_t pack = cpg_conv3x3_supported(d) ? cpg_conv3x3_pack_workspace(d) : cpg_conv1x1_supported(d) ? cpg_conv1x1_pack_workspace(d) : 0;
    pack = std::max(pack, bias_ws_bytes(g.N, g.K));          // the bias gradient's partial sums reuse the workspace
    if (use_c3_wgrad(d)) return std::max(pack, cpg_conv3x3_wgrad_workspace(d));
    if (cpg_conv3x3s2_supported(d)) return std::max(std::max(pack, cpg_conv3x3s2_pack_workspace(d)), cpg_conv3x3s2_wgrad_workspace(d));
    if (cpg_conv_stem2_ok(d)) return std::max(pack, cpg_conv_stem2_wgrad_workspace(d));
    if (cpg_conv1x1_wgrad_supported(d)) return std::max(pack, cpg_conv1x1_wgrad_workspace(d));
    int tm, tn, nsplit, per;
    conv_wgrad_tiles<CfgA>(g, tm, tn);
    const int64_t P = (int64_t)g.N * g.OH * g.OW;
    wgrad_plan((int64_t)tm * tn, (int)((P + CfgA::BK - 1) / CfgA::BK), nsplit, per);
    return std::max(pack, nsplit > 1 ? (size_t)nsplit * g.K * g.C * g.R * g.S * sizeof(float) : (size_t)0);
}

extern "C" int cpg_conv2d_fwd_generic(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr,
                                      const float *bias, float *y, void *stream_v) {
    ConvGeom g;
    int rc = make_geom(d, g);
    if (rc) return rc;
    CPG_REQUIRE(x && w && y, "cpg_conv2d_fwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_v;
    const int64_t P = (int64_t)g.N * g.OH * g.OW;
    Epilogue ep{y, bias, bias ? BIAS_OUTER : BIAS_NONE, (int64_t)g.OH * g.OW, g.K, nullptr, nullptr, nullptr, thr};
    if (g.K <= 64) {
        const int tm = (g.K + CfgB::BM - 1) / CfgB::BM;
        const int64_t tn = (P + CfgB::BN - 1) / CfgB::BN;
        hipLaunchKernelGGL(k_conv_fwd<CfgB>, dim3((unsigned)(tm * tn)), dim3(256), 0, stream, g, x, w, pm, thr, ep, tm);
    } else {
        const int tm = (g.K + CfgA::BM - 1) / CfgA::BM;
        const int64_t tn = (P + CfgA::BN - 1) / CfgA::BN;
        hipLaunchKernelGGL(k_conv_fwd<CfgA>, dim3((unsigned)(tm * tn)), dim3(256), 0, stream, g, x, w, pm, thr, ep, tm);
    }
    CPG_CHECK_LAUNCH("cpg_conv2d_fwd");
    return CPG_OK;
}

static int launch_dgrad_sub(const ConvGeom &g, const DgradSub &sub, const float *gy, const float *w, const float *pm, float thr,
                            float *gx, hipStream_t stream) {
    const int64_t Q = (int64_t)g.N * g.H * g.W;
    Epilogue ep{gx, nullptr, BIAS_NONE, 1, 1, nullptr, nullptr, nullptr, thr};
    if (g.C <= 64) {
        const int tm = (g.C + CfgB::BM - 1) / CfgB::BM;
        const int64_t tn = (Q + CfgB::BN - 1) / CfgB::BN;
        hipLaunchKernelGGL(k_conv_dgrad<CfgB>, dim3((unsigned)(tm * tn)), dim3(256), 0, stream, g, sub, gy, w, pm, thr, ep, tm);
    } else {
        const int tm = (g.C + CfgA::BM - 1) / CfgA::BM;
        const int64_t tn = (Q + CfgA::BN - 1) / CfgA::BN;
        hipLaunchKernelGGL(k_conv_dgrad<CfgA>, dim3((unsigned)(tm * tn)), dim3(256), 0, stream, g, sub, gy, w, pm, thr, ep, tm);
    }
    CPG_CHECK_LAUNCH("cpg_conv2d_dgrad");
    return CPG_OK;
}

extern "C" int cpg_conv2d_dgrad_generic(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr,
                                        float *gx, void *stream_v) {
    ConvGeom g;
    int rc = make_geom(d, g);
    if (rc) return rc;
    CPG_REQUIRE(gy && w && gx, "cpg_conv2d_dgrad: null pointer");
    hipStream_t stream = (hipStream_t)stream_v;
    const DgradSub whole{g.H, g.W, 0, 0, 1, 1, g.R * g.S, g.S, 0, 0};
    if ((g.sh == 1 && g.sw == 1) || g.dh != 1 || g.dw != 1)      // dense, or strided + dilated (no CPG topology): one launch
        return launch_dgrad_sub(g, whole, gy, w, pm, thr, gx, stream);
    // strided: one dense sub-problem per residue class of (h + ph, w + pw) modulo the stride (see DgradSub)
    bool need_zero = false;
    for (int rho = 0; rho < g.sh; ++rho)
        for (int sig = 0; sig < g.sw; ++sig) need_zero = need_zero || rho >= g.R || sig >= g.S;     // class without taps
    if (need_zero) {
        hipError_t e = hipMemsetAsync(gx, 0, (size_t)g.N * g.C * g.H * g.W * sizeof(float), stream);
        if (e != hipSuccess) return hip_status(e, "cpg_conv2d_dgrad(memset)");
    }
    for (int rho = 0; rho < g.sh && rho < g.R; ++rho) {
        const int h_start = ((rho - g.ph) % g.sh + g.sh) % g.sh;
        if (h_start >= g.H) { need_zero = true; continue; }
        for (int sig = 0; sig < g.sw && sig < g.S; ++sig) {
            const int w_start = ((sig - g.pw) % g.sw + g.sw) % g.sw;
            if (w_start >= g.W) continue;
            ConvGeom s = g;
            s.H = (g.H - h_start + g.sh - 1) / g.sh;
            s.W = (g.W - w_start + g.sw - 1) / g.sw;
            s.R = (g.R - rho + g.sh - 1) / g.sh;
            s.S = (g.S - sig + g.sw - 1) / g.sw;
            s.ph = (h_start + g.ph - rho) / g.sh;
            s.pw = (w_start + g.pw - sig) / g.sw;
            s.sh = s.sw = 1;
            const DgradSub sub{g.H, g.W, h_start, w_start, g.sh, g.sw, g.R * g.S, g.S, rho, sig};
            rc = launch_dgrad_sub(s, sub, gy, w, pm, thr, gx, stream);
            if (rc) return rc;
        }
    }
    return CPG_OK;
}

// Drops the calling thread's packed-operand context when a conv entry point returns, whether or not the launch consumed it
// (cpg_conv2d_use_packed is one-shot).  QUERY mode is cpg_conv2d_pack's own and is cleared by it.
namespace {
struct PackScope {
    ~PackScope() {
        cpg::PackCtx &c = cpg::pack_ctx();
        if (c.mode == cpg::PACK_USE) c = cpg::PackCtx{};
    }
};
}  // namespace

extern "C" int cpg_conv2d_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr,
                              const float *bias, float *y, void *ws, size_t ws_bytes, void *stream) {
    PackScope scope;
    if (d && cpg_conv3x3_supported(d)) return cpg_conv3x3_fwd(d, x, w, pm, thr, bias, y, ws, ws_bytes, (hipStream_t)stream);
    if (d && cpg_conv1x1_supported(d)) return cpg_conv1x1_fwd(d, x, w, pm, thr, bias, y, ws, ws_bytes, (hipStream_t)stream);
    if (cpg::pack_query()) return CPG_OK;
    if (d && cpg_conv3x3s2_supported(d)) return cpg_conv3x3s2_fwd(d, x, w, pm, thr, bias, y, nullptr, ws, ws_bytes, (hipStream_t)stream);
    if (d && cpg_conv_stem2_ok(d)) return cpg_conv_stem2_fwd(d, x, w, pm, thr, bias, y, nullptr, (hipStream_t)stream);
    return cpg_conv2d_fwd_generic(d, x, w, pm, thr, bias, y, stream);
}

// Forward that also emits the BatchNorm partial sums of its output (3x3 s1 p1 shapes; 0 tiles = not available).
extern "C" int32_t cpg_conv2d_bnstats_tiles(const cpg_conv_desc *d) {
    if (d && cpg_conv3x3s2_supported(d)) return cpg_conv3x3s2_bnstats_tiles(d);
    if (d && cpg_conv_stem2_ok(d)) return cpg_conv_stem2_tiles(d);
    if (d && !cpg_conv3x3_supported(d) && cpg_conv1x1_supported(d)) return cpg_conv1x1_bnstats_tiles(d);
    return (d && cpg_conv3x3_supported(d)) ? cpg_conv3x3_bnstats_tiles(d) : 0;
}
extern "C" int cpg_conv2d_fwd_bnstats(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr,
                                      const float *bias, float *y, float *stats, size_t stats_bytes, void *ws, size_t ws_bytes,
                                      void *stream) {
    PackScope scope;
    const int tiles = cpg_conv2d_bnstats_tiles(d);
    if (tiles <= 0) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_fwd_bnstats: no fused-statistics kernel for this shape");
    const size_t need = (size_t)d->K * tiles * 2 * sizeof(float);
    if (stats == nullptr || stats_bytes < need)
        return fail(CPG_E_WORKSPACE, "cpg_conv2d_fwd_bnstats: statistics buffer %zu < %zu bytes", stats_bytes, need);
    if (cpg::pack_query() && (cpg_conv3x3s2_supported(d) || cpg_conv_stem2_ok(d))) return CPG_OK;
    if (cpg_conv3x3s2_supported(d)) return cpg_conv3x3s2_fwd(d, x, w, pm, thr, bias, y, stats, ws, ws_bytes, (hipStream_t)stream);
    if (cpg_conv_stem2_ok(d)) return cpg_conv_stem2_fwd(d, x, w, pm, thr, bias, y, stats, (hipStream_t)stream);
    if (!cpg_conv3x3_supported(d)) return cpg_conv1x1_fwd(d, x, w, pm, thr, bias, y, ws, ws_bytes, (hipStream_t)stream, stats);
    return cpg_conv3x3_fwd_bnstats(d, x, w, pm, thr, bias, y, stats, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int cpg_conv3x3_wino_ok(int N, int c_read, int m, int H, int W);
extern "C" int cpg_conv3x3_wino_wgrad_ok(const cpg_conv_desc *d);
extern "C" int cpg_conv3x3_wino_eval_ok(int N, int c_read, int m, int H, int W);
extern "C" int32_t cpg_conv2d_winograd(const cpg_conv_desc *d, int32_t dgrad) {
    ConvGeom g;
    if (d == nullptr || make_geom(d, g) != CPG_OK || !cpg_conv3x3_supported(d)) return 0;
    if (dgrad == 2) return use_c3_wgrad(d) ? cpg_conv3x3_wino_wgrad_ok(d) : 0;
    if (dgrad == 3) return cpg_conv3x3_wino_eval_ok(d->N, d->C, d->K, d->H, d->W);      // the dispatch rule of run_fwd(bn != nullptr)
    return dgrad ? cpg_conv3x3_wino_ok(d->N, d->K, d->C, d->H, d->W) : cpg_conv3x3_wino_ok(d->N, d->C, d->K, d->H, d->W);
}

// conv -> eval-mode BatchNorm2d (-> ReLU) in one kernel; 0 from the _supported query: use cpg_conv2d_fwd + cpg_bn_relu_fwd_eval
extern "C" int32_t cpg_conv2d_fwd_bn_eval_supported(const cpg_conv_desc *d) {
    ConvGeom g;
    return make_geom(d, g) == CPG_OK && cpg_conv3x3_supported(d) ? 1 : 0;
}
extern "C" int cpg_conv2d_fwd_bn_eval(const cpg_conv_desc *d, const float *x, const float *w, const float *pm, float thr,
                                      const float *bias, const float *gamma, const float *beta, const float *running_mean,
                                      const float *running_var, float eps, int32_t relu, float *y, int32_t *skip_stats, void *ws,
                                      size_t ws_bytes, void *stream) {
    ConvGeom g;
    int rc = make_geom(d, g);
    if (rc) return rc;
    if (!cpg_conv3x3_supported(d)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_fwd_bn_eval: only the 3x3 s1 p1 kernels fuse the epilogue");
    return cpg_conv3x3_fwd_bn_eval(d, x, w, pm, thr, bias, gamma, beta, running_mean, running_var, eps, relu, y, skip_stats, ws,
                                   ws_bytes, (hipStream_t)stream);
}

// input gradient + the BatchNorm-backward reduction of the layer below in its epilogue (3x3 s1 p1 kernels only)
extern "C" int32_t cpg_conv2d_dgrad_bnbwd_tiles(const cpg_conv_desc *d) {
    ConvGeom g;
    if (make_geom(d, g) != CPG_OK || !cpg_conv3x3_supported(d)) return 0;
    return cpg_conv3x3_dgrad_bnbwd_tiles(d);
}
extern "C" int cpg_conv2d_dgrad_bnbwd(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr,
                                      const float *ypre, const float *gamma, const float *beta, const float *mean, const float *invstd,
                                      float *gm, float *partials, size_t partial_bytes, void *ws, size_t ws_bytes, void *stream) {
    PackScope scope;
    ConvGeom g;
    int rc = make_geom(d, g);
    if (rc) return rc;
    const int tiles = cpg_conv2d_dgrad_bnbwd_tiles(d);
    if (tiles <= 0) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_dgrad_bnbwd: this shape has no fused path");
    if (partial_bytes < (size_t)d->C * tiles * 2 * sizeof(float)) return fail(CPG_E_WORKSPACE, "cpg_conv2d_dgrad_bnbwd: partial-sum buffer too small");
    return cpg_conv3x3_dgrad_bnbwd(d, gy, w, pm, thr, ypre, gamma, beta, mean, invstd, gm, partials, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int cpg_conv2d_dgrad(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr,
                                float *gx, void *ws, size_t ws_bytes, void *stream) {
    PackScope scope;
    if (d && cpg_conv3x3_supported(d)) return cpg_conv3x3_dgrad(d, gy, w, pm, thr, gx, ws, ws_bytes, (hipStream_t)stream);
    if (d && cpg_conv1x1_supported(d)) return cpg_conv1x1_dgrad(d, gy, w, pm, thr, gx, ws, ws_bytes, (hipStream_t)stream);
    if (cpg::pack_query()) return CPG_OK;
    if (d && cpg_conv3x3s2_supported(d)) return cpg_conv3x3s2_dgrad(d, gy, w, pm, thr, gx, ws, ws_bytes, (hipStream_t)stream);
    return cpg_conv2d_dgrad_generic(d, gy, w, pm, thr, gx, stream);
}

// input gradient + addend (the gradient of the input's other consumer): dense pointwise layers only
extern "C" int cpg_conv3x3_wino_dgrad_add_ok(int N, int c_read, int m, int H, int W);
extern "C" int cpg_conv3x3_wino_dgrad_add(int N, int c_read, int m, int H, int W, int K, int C, const float *gy, const float *w, const float *pm,
                                          float thr, const float *addend, float *gx, void *ws, size_t ws_bytes, hipStream_t stream);
extern "C" int32_t cpg_conv2d_dgrad_add_supported(const cpg_conv_desc *d) {
    if (d == nullptr) return 0;
    // (round 5) the 3x3 s1 p1 layers whose input gradient runs the two-wave Winograd kernel: SphereNet's residual units
    if (cpg_conv3x3_supported(d)) return cpg_conv3x3_wino_dgrad_add_ok(d->N, d->K, d->C, d->H, d->W) ? 1 : 0;
    return cpg_conv1x1_supported(d) && d->stride_h == 1 && d->stride_w == 1 ? 1 : 0;
}
extern "C" int cpg_conv2d_dgrad_add(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm, float thr, const float *addend,
                                    float *gx, void *ws, size_t ws_bytes, void *stream) {
    PackScope scope;
    if (!cpg_conv2d_dgrad_add_supported(d))
        return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_dgrad_add: dense 1x1 layers and the 3x3 layers of the two-wave Winograd kernel fuse the addend");
    CPG_REQUIRE(addend != nullptr && gy && w && gx, "cpg_conv2d_dgrad_add: null pointer");
    if (cpg_conv3x3_supported(d))
        return cpg_conv3x3_wino_dgrad_add(d->N, d->K, d->C, d->H, d->W, d->K, d->C, gy, w, pm, thr, addend, gx, ws, ws_bytes, (hipStream_t)stream);
    return cpg_conv1x1_dgrad(d, gy, w, pm, thr, gx, ws, ws_bytes, (hipStream_t)stream, addend);
}

// ---- caller-owned packed weight operands (include/cpg_hip.h, ABI 3) ------------------------------------------------------------------
namespace {
// What a call of this pass would pack: run the REAL dispatch in query mode -- it stops at the pack site.  Only shape classes whose every
// route ends at a hooked site may enter (3x3 s1 p1 layers on the Winograd kernels, pointwise layers): the pointers below are never
// dereferenced, but a route without a site would launch on them.
bool query_pack_job(const cpg_conv_desc *d, int pass, cpg::PackJob *out) {
    ConvGeom g;
    if (d == nullptr || pass < 0 || pass > 2 || make_geom(d, g) != CPG_OK) return false;
    if (d->N <= 0 || d->K <= 0 || d->C <= 0) return false;
    bool eligible = false;
    if (cpg_conv3x3_supported(d))
        eligible = cpg_conv2d_winograd(d, pass == 1 ? 1 : 0) != 0;
    else if (cpg_conv1x1_supported(d))
        eligible = true;
    if (!eligible || (pass == 2 && cpg_conv2d_bnstats_tiles(d) <= 0)) return false;
    float *const fake = reinterpret_cast<float *>((uintptr_t)1 << 20);
    const size_t big = (size_t)1 << 46;
    cpg::PackCtx &c = cpg::pack_ctx();
    c = cpg::PackCtx{};
    c.mode = cpg::PACK_QUERY;
    const int rc = pass == 1   ? cpg_conv2d_dgrad(d, fake, fake, nullptr, 0.0f, fake, fake, big, nullptr)
                   : pass == 2 ? cpg_conv2d_fwd_bnstats(d, fake, fake, nullptr, 0.0f, nullptr, fake, fake, big, fake, big, nullptr)
                               : cpg_conv2d_fwd(d, fake, fake, nullptr, 0.0f, nullptr, fake, fake, big, nullptr);
    const bool hit = c.hit;
    const cpg::PackJob job = c.job;
    c = cpg::PackCtx{};
    if (rc != CPG_OK || !hit || job.family == 0) return false;
    *out = job;
    return true;
}
}  // namespace

extern "C" size_t cpg_conv2d_pack_bytes(const cpg_conv_desc *d, int32_t pass) {
    cpg::PackJob j;
    return query_pack_job(d, pass, &j) ? j.bytes : 0;
}

extern "C" int cpg_conv2d_pack(const cpg_conv_desc *d, const float *w, const float *pm, float thr, int32_t pass_a, void *packed_a,
                               size_t bytes_a, int32_t pass_b, void *packed_b, size_t bytes_b, void *stream) {
    CPG_REQUIRE(d && w && packed_a, "cpg_conv2d_pack: null pointer");
    cpg::PackJob ja, jb;
    if (!query_pack_job(d, pass_a, &ja)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_pack: pass %d of this shape streams no packed operand", pass_a);
    if (bytes_a != ja.bytes) return fail(CPG_E_WORKSPACE, "cpg_conv2d_pack: buffer of %zu bytes for a %zu-byte operand (pass %d)", bytes_a, ja.bytes, pass_a);
    CPG_REQUIRE((((uintptr_t)packed_a) & 15) == 0 && (((uintptr_t)packed_b) & 15) == 0, "cpg_conv2d_pack: packed operands must be 16-byte aligned");
    if (packed_b != nullptr) {
        if (!query_pack_job(d, pass_b, &jb)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_pack: pass %d of this shape streams no packed operand", pass_b);
        if (bytes_b != jb.bytes) return fail(CPG_E_WORKSPACE, "cpg_conv2d_pack: buffer of %zu bytes for a %zu-byte operand (pass %d)", bytes_b, jb.bytes, pass_b);
    }
    return cpg::pack_jobs_launch(&ja, (float *)packed_a, packed_b ? &jb : nullptr, (float *)packed_b, w, pm, thr, (hipStream_t)stream);
}

extern "C" int cpg_conv2d_use_packed(const void *packed, size_t bytes) {
    cpg::PackCtx &c = cpg::pack_ctx();
    c = cpg::PackCtx{};
    if (packed == nullptr) return CPG_OK;        // (disarms)
    c.mode = cpg::PACK_USE;
    c.use = (const float *)packed;
    c.use_bytes = bytes;
    return CPG_OK;
}

extern "C" int cpg_conv2d_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm,
                                float thr, float *gw, float *gpm, float *gb, void *ws, size_t ws_bytes, void *stream_v) {
    ConvGeom g;
    int rc = make_geom(d, g);
    if (rc) return rc;
    CPG_REQUIRE(x && gy && gw, "cpg_conv2d_wgrad: null pointer");
    CPG_REQUIRE((pm == nullptr) == (gpm == nullptr), "cpg_conv2d_wgrad: pm and gpm must both be given or both be NULL");
    CPG_REQUIRE(pm == nullptr || w != nullptr, "cpg_conv2d_wgrad: w is required to form the piggymask gradient");
    if (gb != nullptr && (ws == nullptr || ws_bytes < bias_ws_bytes(g.N, g.K)))
        return fail(CPG_E_WORKSPACE, "cpg_conv2d_wgrad: workspace %zu < %zu bytes (bias gradient)", ws_bytes, bias_ws_bytes(g.N, g.K));
    hipStream_t stream = (hipStream_t)stream_v;
    if (use_c3_wgrad(d)) {
        rc = cpg_conv3x3_wgrad(d, x, gy, w, pm, thr, gw, gpm, ws, ws_bytes, stream);
        if (rc) return rc;
        if (gb) launch_conv_bias_grad(gy, gb, g.N, g.K, g.OH * g.OW, ws, stream);
        CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(bias)");
        return CPG_OK;
    }
    if (cpg_conv_stem2_ok(d)) {
        rc = cpg_conv_stem2_wgrad(d, x, gy, w, pm, thr, gw, gpm, ws, ws_bytes, stream);
        if (rc) return rc;
        if (gb) launch_conv_bias_grad(gy, gb, g.N, g.K, g.OH * g.OW, ws, stream);
        CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(bias)");
        return CPG_OK;
    }
    if (cpg_conv3x3s2_supported(d)) {
        rc = cpg_conv3x3s2_wgrad(d, x, gy, w, pm, thr, gw, gpm, ws, ws_bytes, stream);
        if (rc) return rc;
        if (gb) launch_conv_bias_grad(gy, gb, g.N, g.K, g.OH * g.OW, ws, stream);
        CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(bias)");
        return CPG_OK;
    }
    if (cpg_conv1x1_wgrad_supported(d)) {
        rc = cpg_conv1x1_wgrad(d, x, gy, w, pm, thr, gw, gpm, ws, ws_bytes, stream);
        if (rc) return rc;
        if (gb) launch_conv_bias_grad(gy, gb, g.N, g.K, g.OH * g.OW, ws, stream);
        CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(bias)");
        return CPG_OK;
    }
    int tm, tn, nsplit, per;
    conv_wgrad_tiles<CfgA>(g, tm, tn);
    const int64_t P = (int64_t)g.N * g.OH * g.OW;
    const int nkt = (int)((P + CfgA::BK - 1) / CfgA::BK);
    wgrad_plan((int64_t)tm * tn, nkt, nsplit, per);
    const int64_t out_elems = (int64_t)g.K * g.C * g.R * g.S;
    if (nsplit > 1 && ws_bytes < (size_t)nsplit * out_elems * sizeof(float))
        return fail(CPG_E_WORKSPACE, "cpg_conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes,
                    (size_t)nsplit * out_elems * sizeof(float));
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    float *part = nsplit > 1 ? (float *)ws : nullptr;
    hipLaunchKernelGGL(k_conv_wgrad<CfgA>, dim3((unsigned)(tm * tn), (unsigned)nsplit), dim3(256), 0, stream, g, x, gy, ep,
                       part, tm, per);
    if (nsplit > 1)
        launch_split_reduce(part, nsplit, out_elems, 0, ep, stream);
    if (gb) launch_conv_bias_grad(gy, gb, g.N, g.K, g.OH * g.OW, ws, stream);
    CPG_CHECK_LAUNCH("cpg_conv2d_wgrad");
    return CPG_OK;
}

// ---------------------------------------------------------------------------------- linear host side
namespace {
void gemm_plan(int64_t tiles, int nkt, int &nsplit, int &per) {
    int64_t want = (2 * kCUs + tiles - 1) / tiles;      // aim for >= 2 blocks per CU
    if (tiles >= 2 * kCUs) want = 1;
    int64_t max_by_k = (nkt + 15) / 16;                  // >= 16 K tiles per split
    if (want > max_by_k) want = max_by_k;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    per = (int)((nkt + want - 1) / want);
    nsplit = (nkt + per - 1) / per;
}
}  // namespace
// plain-GEMM entry points of pointwise.hip (used when the layer has no piggymask)
bool cpg_pw_gemm_nt_ok(const float *A, const float *B, int M, int C, int64_t K);
size_t cpg_pw_gemm_nt_workspace(int M, int C, int64_t K);
int cpg_pw_gemm_nt(const float *A, const float *B, int M, int C, int64_t K, const Epilogue &ep, void *ws, size_t ws_bytes,
                   hipStream_t stream, const char *what);
int cpg_pw_gemm_nt_maskb(const float *A, const float *B, const float *pmB, float thr, int M, int C, int64_t K, const Epilogue &ep, void *ws,
                         size_t ws_bytes, hipStream_t stream, const char *what);
bool cpg_pw_gemm_nn_ok(const float *X, int M, int Mp, int Kd, int64_t G, bool masked = false);
int cpg_pw_gemm_nn(const float *wp, int Mp, const float *X, int M, int Kd, int64_t G, const float *bias, float *y, hipStream_t stream,
                   const char *what);
int cpg_pw_gemm_nn_masked(const float *wp, int Mp, const float *X, int M, int Kd, int64_t G, const float *pm, const float *w, float thr,
                          float *gw, float *gpm, hipStream_t stream, const char *what);
int cpg_pw_gemm_nn_maskx(const float *wp, int Mp, const float *X, const float *pmX, float thr, int M, int Kd, int64_t G, float *y,
                         hipStream_t stream, const char *what);
void cpg_pw_pack_transpose(const float *a, int R, int Cc, float *wp, hipStream_t stream);
size_t cpg_pw_pack_transpose_bytes(int R, int Cc);
// fc_small.hip: the weight-streaming input gradient at <= 64 rows
bool cpg_fc_small_dgrad_ok(const float *w, const float *pm, const float *gx, int batch, int in_f, int out_f);
size_t cpg_fc_small_dgrad_workspace(int batch, int in_f, int out_f);
int cpg_fc_small_dgrad(const float *gy, const float *w, const float *pm, float thr, float *gx, int batch, int in_f, int out_f, void *ws,
                       size_t ws_bytes, hipStream_t stream, const char *what);
namespace {
size_t linear_ws(int batch, int in_f, int out_f) {
    size_t best = std::max(cpg_pw_gemm_nt_workspace(batch, out_f, in_f), cpg_pw_pack_transpose_bytes(batch, out_f));
    best = std::max(best, cpg_fc_small_dgrad_workspace(batch, in_f, out_f));
    int ns, per;
    {   // fwd: M=batch N=out K=in
        int64_t tiles = (int64_t)((batch + 127) / 128) * ((out_f + 127) / 128);
        gemm_plan(tiles, (in_f + 15) / 16, ns, per);
        if (ns > 1) best = std::max(best, (size_t)ns * batch * out_f * sizeof(float));
    }
    {   // dgrad: M=batch N=in K=out
        int64_t tiles = (int64_t)((batch + 127) / 128) * ((in_f + 127) / 128);
        gemm_plan(tiles, (out_f + 15) / 16, ns, per);
        if (ns > 1) best = std::max(best, (size_t)ns * batch * in_f * sizeof(float));
    }
    {   // wgrad: M=out N=in K=batch
        int64_t tiles = (int64_t)((out_f + 127) / 128) * ((in_f + 127) / 128);
        gemm_plan(tiles, (batch + 15) / 16, ns, per);
        if (ns > 1) best = std::max(best, (size_t)ns * out_f * in_f * sizeof(float));
    }
    return best;
}

template <bool A_KC, bool B_KC, int MASK_SIDE>
int launch_gemm(const float *A, int64_t lda, const float *B, int64_t ldb, const float *pm, float thr, int M, int Nn, int Kd,
                const Epilogue &ep, void *ws, size_t ws_bytes, hipStream_t stream, const char *what) {
    const int tm = (M + CfgA::BM - 1) / CfgA::BM, tn = (Nn + CfgA::BN - 1) / CfgA::BN;
    int nsplit, per;
    gemm_plan((int64_t)tm * tn, (Kd + CfgA::BK - 1) / CfgA::BK, nsplit, per);
    const int64_t out_elems = (int64_t)M * Nn;
    if (nsplit > 1 && ws_bytes < (size_t)nsplit * out_elems * sizeof(float))
        return fail(CPG_E_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, (size_t)nsplit * out_elems * sizeof(float));
    float *part = nsplit > 1 ? (float *)ws : nullptr;
    hipLaunchKernelGGL((k_gemm<CfgA, A_KC, B_KC, MASK_SIDE>), dim3((unsigned)(tm * tn), (unsigned)nsplit), dim3(256), 0, stream,
                       A, lda, B, ldb, pm, thr, M, Nn, Kd, ep, part, tm, per);
    if (nsplit > 1)
        launch_split_reduce(part, nsplit, out_elems, 0, ep, stream);
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}
}  // namespace

extern "C" size_t cpg_linear_workspace_bytes(int32_t batch, int32_t in_f, int32_t out_f) {
    if (batch <= 0 || in_f <= 0 || out_f <= 0) return 0;
    return linear_ws(batch, in_f, out_f);
}

extern "C" int cpg_linear_fwd(const float *x, const float *w, const float *pm, float thr, const float *bias, float *y,
                              int32_t batch, int32_t in_f, int32_t out_f, void *ws, size_t ws_bytes, void *stream) {
    CPG_REQUIRE(x && w && y && batch > 0 && in_f > 0 && out_f > 0, "cpg_linear_fwd: bad argument");
    Epilogue ep{y, bias, bias ? BIAS_INNER : BIAS_NONE, out_f, 1, nullptr, nullptr, nullptr, thr};
    if (pm == nullptr && cpg_pw_gemm_nt_ok(x, w, batch, out_f, in_f))          // y[b][o] = x[b][:] . W[o][:]
        return cpg_pw_gemm_nt(x, w, batch, out_f, in_f, ep, ws, ws_bytes, (hipStream_t)stream, "cpg_linear_fwd");
    if (pm != nullptr && (((uintptr_t)pm) & 15) == 0 && cpg_pw_gemm_nt_ok(x, w, batch, out_f, in_f))      // (round 5) ... x[b][:] . (W * bin(pm))[o][:]
        return cpg_pw_gemm_nt_maskb(x, w, pm, thr, batch, out_f, in_f, ep, ws, ws_bytes, (hipStream_t)stream, "cpg_linear_fwd");
    return launch_gemm<true, true, 2>(x, in_f, w, in_f, pm, thr, batch, out_f, in_f, ep, ws, ws_bytes, (hipStream_t)stream,
                                      "cpg_linear_fwd");
}

extern "C" int cpg_linear_dgrad(const float *gy, const float *w, const float *pm, float thr, float *gx, int32_t batch,
                                int32_t in_f, int32_t out_f, void *ws, size_t ws_bytes, void *stream) {
    CPG_REQUIRE(gy && w && gx && batch > 0 && in_f > 0 && out_f > 0, "cpg_linear_dgrad: bad argument");
    Epilogue ep{gx, nullptr, BIAS_NONE, 1, 1, nullptr, nullptr, nullptr, thr};
    // (round 5) <= 64 rows: the weight is STREAMED once, operands straight from global memory (fc_small.hip)
    if (batch <= 64 && ws != nullptr && ws_bytes >= cpg_fc_small_dgrad_workspace(batch, in_f, out_f) && (((uintptr_t)ws) & 15) == 0 &&
        (((uintptr_t)gy) & 3) == 0 && cpg_fc_small_dgrad_ok(w, pm, gx, batch, in_f, out_f))
        return cpg_fc_small_dgrad(gy, w, pm, thr, gx, batch, in_f, out_f, ws, ws_bytes, (hipStream_t)stream, "cpg_linear_dgrad");
    const int Mp_b = (batch + 127) / 128 * 128;
    if (ws != nullptr && ws_bytes >= cpg_pw_pack_transpose_bytes(batch, out_f) && (((uintptr_t)ws) & 15) == 0 &&
        cpg_pw_gemm_nn_ok(w, batch, Mp_b, out_f, in_f, pm != nullptr) && (pm == nullptr || (in_f % 4 == 0 && (((uintptr_t)pm) & 15) == 0))) {
        // gx[b][i] = sum_o gy[b][o] W[o][i]: gy^T packed K-major (4 MB) is the "weight", W[o][:] the K-major operand -- with a piggymask
        // (round 5) the operand is W * bin(pm), formed in the kernel's staging (k_pw<.., MASKX>): W and pm are read once
        cpg_pw_pack_transpose(gy, batch, out_f, (float *)ws, (hipStream_t)stream);
        if (pm != nullptr)
            return cpg_pw_gemm_nn_maskx((const float *)ws, Mp_b, w, pm, thr, batch, out_f, in_f, gx, (hipStream_t)stream, "cpg_linear_dgrad");
        return cpg_pw_gemm_nn((const float *)ws, Mp_b, w, batch, out_f, in_f, nullptr, gx, (hipStream_t)stream, "cpg_linear_dgrad");
    }
    return launch_gemm<true, false, 2>(gy, out_f, w, in_f, pm, thr, batch, in_f, out_f, ep, ws, ws_bytes, (hipStream_t)stream,
                                       "cpg_linear_dgrad");
}

extern "C" int cpg_linear_wgrad(const float *x, const float *gy, const float *w, const float *pm, float thr, float *gw,
                                float *gpm, float *gb, int32_t batch, int32_t in_f, int32_t out_f, void *ws, size_t ws_bytes,
                                void *stream) {
    CPG_REQUIRE(x && gy && gw && batch > 0 && in_f > 0 && out_f > 0, "cpg_linear_wgrad: bad argument");
    CPG_REQUIRE((pm == nullptr) == (gpm == nullptr), "cpg_linear_wgrad: pm and gpm must both be given or both be NULL");
    CPG_REQUIRE(pm == nullptr || w != nullptr, "cpg_linear_wgrad: w is required to form the piggymask gradient");
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    int rc;
    if (pm == nullptr && (((uintptr_t)gy) & 15) == 0 && cpg_pw_gemm_nn_ok(x, out_f, out_f, batch, in_f))
        // gW[o][i] = sum_b gy[b][o] x[b][i]: gy is already K-major ([b][o], o a multiple of 128), x[b][:] the other operand
        rc = cpg_pw_gemm_nn(gy, out_f, x, out_f, batch, in_f, nullptr, gw, (hipStream_t)stream, "cpg_linear_wgrad");
    else if (pm != nullptr && (((uintptr_t)gy) & 15) == 0 && cpg_pw_gemm_nn_ok(x, out_f, out_f, batch, in_f, true))
        // ... and with a piggymask the same GEMM with the autograd epilogue of bin(pm) * W (gW = g bin(pm), gPM = g W from ONE accumulator tile)
        rc = cpg_pw_gemm_nn_masked(gy, out_f, x, out_f, batch, in_f, pm, w, thr, gw, gpm, (hipStream_t)stream, "cpg_linear_wgrad");
    else
        rc = launch_gemm<false, false, 0>(gy, out_f, x, in_f, nullptr, thr, out_f, in_f, batch, ep, ws, ws_bytes,
                                          (hipStream_t)stream, "cpg_linear_wgrad");
    if (rc) return rc;
    if (gb) {
        hipLaunchKernelGGL(k_bias_grad, dim3((unsigned)out_f), dim3(256), 0, (hipStream_t)stream, gy, gb, batch, out_f, 1);
        CPG_CHECK_LAUNCH("cpg_linear_wgrad(bias)");
    }
    return CPG_OK;
}
