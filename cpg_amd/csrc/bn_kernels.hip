// Fused BatchNorm2d (+ ReLU) forward / backward for NCHW fp32 -- SURVEY.md section 8(f) item 2: the
// BN -> ReLU pairs that follow every masked conv in the VGG topologies (models/vgg.py:137-141).
// All four kernels are HBM-bound streaming passes (16 B per lane):
//
//   stats      : 1 read            per-(channel, slice) sum / sum-of-squares in fp64, fixed-order finalize
//                                  (mean, biased var, invstd; running stats updated in the same kernel)
//   apply      : 1 read + 1 write  y = max(0, (x - mean) * invstd * gamma + beta)
//   bwd reduce : 2 reads           sum(g), sum(g * xhat) with g = gy * [y > 0]; the ReLU mask is RECOMPUTED
//                                  from x (same expression as apply), so y is never re-read
//   bwd apply  : 2 reads + 1 write dx = (g - mean(g) - xhat * mean(g * xhat)) * invstd * gamma
//
// i.e. 3 passes forward and 5 backward over the activation, against 5 + 8 for the stock
// BatchNorm -> ReLU(inplace) pair (and MIOpen's BN forward itself ran ~4x off its roofline here).
// Two-stage reductions are deterministic (no float atomics): results do not depend on block scheduling.
#include "cpg_common.h"

using namespace cpg;

namespace {

constexpr int kThreads = 256;

struct BnDims {
    int N, C, HW;
    int slices;          // reduction slices per channel (over images)
    int imgs_per_slice;
};

__device__ __forceinline__ float bn_affine(float x, float mean, float invstd, float gamma, float beta) {
    return (x - mean) * invstd * gamma + beta;
}

__device__ __forceinline__ double block_sum(double v, double *red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];       // fixed order
}

// Visit item j < P of every image plane n in [n0, n1): f(n, j), with flush() at least every 64 visits of a thread (the
// callers keep fp32 running sums and fold them into fp64 there).  No per-element division: large planes are walked
// image by image (4 independent visits per loop trip, so 4 loads are in flight per thread -- the one-visit-per-trip
// version with a 64-bit (i / P, i % P) per element ran k_bn_stats at 4.9 TB/s and k_bnp_bwd_reduce at 3.3 TB/s);
// small planes are walked flattened.
template <class F, class FL>
__device__ __forceinline__ void walk_planes(int P, int n0, int n1, F &&f, FL &&flush) {
    const int tid = threadIdx.x;
    if (P >= 16 * kThreads) {
        for (int n = n0; n < n1; ++n) {
            int j = tid, trips = 0;
            for (; j + 3 * kThreads < P; j += 4 * kThreads) {
                f(n, j);
                f(n, j + kThreads);
                f(n, j + 2 * kThreads);
                f(n, j + 3 * kThreads);
                if (++trips == 16) {
                    flush();
                    trips = 0;
                }
            }
            for (; j < P; j += kThreads) f(n, j);
            flush();
        }
    } else {
        // small planes: the (image, item) pairs of the slice are flattened so that every thread stays busy (a per-image
        // walk would idle 23 % of the block on 784-item planes); 32-bit index arithmetic
        const unsigned total = (unsigned)(n1 - n0) * (unsigned)P;
        int cnt = 0;
        for (unsigned i = tid; i < total; i += kThreads) {
            const unsigned dn = i / (unsigned)P;
            f(n0 + (int)dn, (int)(i - dn * (unsigned)P));
            if (++cnt == 64) {
                flush();
                cnt = 0;
            }
        }
        flush();
    }
}

// grid (C, slices): partial[c][slice] = {sum, sumsq} over images [s*ips, (s+1)*ips)
__global__ __launch_bounds__(kThreads) void k_bn_stats(const float *__restrict__ x, BnDims d, double *__restrict__ partial) {
    __shared__ double red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const int n0 = s * d.imgs_per_slice, n1 = min(d.N, n0 + d.imgs_per_slice);
    double ds = 0.0, dss = 0.0;
    float fs = 0.f, fss = 0.f;
    const bool vec = (d.HW & 3) == 0 && (((uintptr_t)x) & 15) == 0;
    auto flush = [&]() {          // bounds the fp32 partials to <= 256 terms
        ds += (double)fs; dss += (double)fss; fs = 0.f; fss = 0.f;
    };
    if (vec)
        walk_planes(d.HW >> 2, n0, n1, [&](int n, int j) {
            const float4 v = *reinterpret_cast<const float4 *>(x + ((int64_t)n * d.C + c) * d.HW + 4 * j);
            fs += (v.x + v.y) + (v.z + v.w);
            fss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }, flush);
    else
        walk_planes(d.HW, n0, n1, [&](int n, int j) {
            const float v = x[((int64_t)n * d.C + c) * d.HW + j];
            fs += v;
            fss += v * v;
        }, flush);
    const double ts = block_sum(ds, red);
    const double tss = block_sum(dss, red);
    if (threadIdx.x == 0) {
        partial[((int64_t)c * d.slices + s) * 2 + 0] = ts;
        partial[((int64_t)c * d.slices + s) * 2 + 1] = tss;
    }
}

// one thread per channel: fixed-order merge, mean / biased var / invstd, running statistics
// (torch.nn.BatchNorm2d semantics: running_var uses the unbiased variance, momentum m)
__global__ void k_bn_finalize(const double *__restrict__ partial, BnDims d, float eps, float momentum, float *__restrict__ mean,
                              float *__restrict__ invstd, float *__restrict__ running_mean, float *__restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.C) return;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < d.slices; ++k) {
        s += partial[((int64_t)c * d.slices + k) * 2 + 0];
        ss += partial[((int64_t)c * d.slices + k) * 2 + 1];
    }
    const double n = (double)d.N * d.HW;
    const double m = s / n;
    double var = ss / n - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// GROUP threads (a whole block, or one wave for small feature maps) own one (n, c) plane at a time, so the
// four per-channel scalars sit in registers and every access is a contiguous 16 B per lane.
// `res` (may be null): a residual added before the ReLU -- the tail of a ResNet block, relu(bn3(conv3) + identity).
// `mask` (may be null; needs 4 | HW and 16-byte aligned tensors): one byte per float4 of y whose bits 0-3 say which of its four
// elements are > 0 -- the ReLU mask the backward of relu(bn(x) + res) needs (it cannot be recomputed from x alone), at 1/16 of the
// bytes of the output it would otherwise re-read.
template <bool RELU, int GROUP>
__global__ __launch_bounds__(kThreads) void k_bn_apply(const float *__restrict__ x, const float *__restrict__ res,
                                                       float *__restrict__ y, BnDims d, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, uint8_t *__restrict__ mask = nullptr) {
    const int64_t planes = (int64_t)d.N * d.C;
    const bool vec = (d.HW & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 15) == 0 && (((uintptr_t)res) & 15) == 0;
    const int gl = threadIdx.x % GROUP;
    const int64_t g0 = (int64_t)blockIdx.x * (kThreads / GROUP) + threadIdx.x / GROUP;
    const int64_t gstride = (int64_t)gridDim.x * (kThreads / GROUP);
    for (int64_t pl = g0; pl < planes; pl += gstride) {
        const int c = (int)(pl % d.C);
        const float m = mean[c], is = invstd[c], g = gamma[c], b = beta[c];
        const float *p = x + pl * d.HW;
        float *q = y + pl * d.HW;
        if (vec) {
            for (int i = gl; i < d.HW / 4; i += GROUP) {
                float4 v = reinterpret_cast<const float4 *>(p)[i];
                v.x = bn_affine(v.x, m, is, g, b);
                v.y = bn_affine(v.y, m, is, g, b);
                v.z = bn_affine(v.z, m, is, g, b);
                v.w = bn_affine(v.w, m, is, g, b);
                if (res != nullptr) {
                    const float4 r = reinterpret_cast<const float4 *>(res + pl * d.HW)[i];
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                reinterpret_cast<float4 *>(q)[i] = v;
                if (mask != nullptr)
                    mask[pl * (d.HW >> 2) + i] = (uint8_t)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
            }
        } else {
            for (int i = gl; i < d.HW; i += GROUP) {
                float v = bn_affine(p[i], m, is, g, b);
                if (res != nullptr) v += res[pl * d.HW + i];
                q[i] = RELU ? fmaxf(v, 0.f) : v;
            }
        }
    }
}

// grid (C, slices): partial[c][slice] = {sum g, sum g*xhat}
// MASKY (the tail of a residual block, out = relu(bn(x) + identity)): the ReLU mask cannot be recomputed from x alone, so it is
// taken from the saved output (`out > 0`), and the masked gradient gz = g * [out > 0] -- which is also the identity branch's
// gradient -- is WRITTEN here, so that neither a separate threshold pass nor a second read of `out` in the apply pass is needed.
template <bool RELU, bool MASKY = false>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_reduce(const float *__restrict__ x, const float *__restrict__ gy, BnDims d,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            double *__restrict__ partial, const float *__restrict__ out = nullptr,
                                                            float *__restrict__ gz = nullptr, const uint8_t *__restrict__ mask = nullptr) {
    __shared__ double red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const int n0 = s * d.imgs_per_slice, n1 = min(d.N, n0 + d.imgs_per_slice);
    const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    double dsg = 0.0, dsgx = 0.0;
    float sg = 0.f, sgx = 0.f;
    const bool vec = (d.HW & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)gy) & 15) == 0;
    auto visit = [&](float xv, float gv) {
        if (RELU && !(bn_affine(xv, m, is, ga, be) > 0.f)) gv = 0.f;
        sg += gv;
        sgx += gv * ((xv - m) * is);
    };
    auto flush = [&]() {
        dsg += (double)sg; dsgx += (double)sgx; sg = 0.f; sgx = 0.f;
    };
    if (MASKY) {
        const bool vec3 = vec && (((uintptr_t)out) & 15) == 0 && (((uintptr_t)gz) & 15) == 0;
        if (vec && mask != nullptr && (((uintptr_t)gz) & 15) == 0)
            // the forward left one byte per float4 of `out` (bits 0-3: element > 0): 1/16 of the bytes of re-reading `out`
            walk_planes(d.HW >> 2, n0, n1, [&](int n, int j) {
                const int64_t q4 = ((int64_t)n * d.C + c) * (d.HW >> 2) + j, off = 4 * q4;
                const float4 xv = *reinterpret_cast<const float4 *>(x + off);
                float4 gv = *reinterpret_cast<const float4 *>(gy + off);
                const unsigned mb = mask[q4];
                gv.x = (mb & 1u) ? gv.x : 0.f; gv.y = (mb & 2u) ? gv.y : 0.f;
                gv.z = (mb & 4u) ? gv.z : 0.f; gv.w = (mb & 8u) ? gv.w : 0.f;
                *reinterpret_cast<float4 *>(gz + off) = gv;
                visit(xv.x, gv.x); visit(xv.y, gv.y); visit(xv.z, gv.z); visit(xv.w, gv.w);
            }, flush);
        else if (vec3)
            walk_planes(d.HW >> 2, n0, n1, [&](int n, int j) {
                const int64_t off = ((int64_t)n * d.C + c) * d.HW + 4 * j;
                const float4 xv = *reinterpret_cast<const float4 *>(x + off);
                float4 gv = *reinterpret_cast<const float4 *>(gy + off);
                const float4 ov = *reinterpret_cast<const float4 *>(out + off);
                gv.x = ov.x > 0.f ? gv.x : 0.f; gv.y = ov.y > 0.f ? gv.y : 0.f;
                gv.z = ov.z > 0.f ? gv.z : 0.f; gv.w = ov.w > 0.f ? gv.w : 0.f;
                *reinterpret_cast<float4 *>(gz + off) = gv;
                visit(xv.x, gv.x); visit(xv.y, gv.y); visit(xv.z, gv.z); visit(xv.w, gv.w);
            }, flush);
        else
            walk_planes(d.HW, n0, n1, [&](int n, int j) {
                const int64_t off = ((int64_t)n * d.C + c) * d.HW + j;
                const float gv = out[off] > 0.f ? gy[off] : 0.f;
                gz[off] = gv;
                visit(x[off], gv);
            }, flush);
    } else if (vec)
        walk_planes(d.HW >> 2, n0, n1, [&](int n, int j) {
            const int64_t off = ((int64_t)n * d.C + c) * d.HW + 4 * j;
            const float4 xv = *reinterpret_cast<const float4 *>(x + off);
            const float4 gv = *reinterpret_cast<const float4 *>(gy + off);
            visit(xv.x, gv.x); visit(xv.y, gv.y); visit(xv.z, gv.z); visit(xv.w, gv.w);
        }, flush);
    else
        walk_planes(d.HW, n0, n1, [&](int n, int j) {
            const int64_t off = ((int64_t)n * d.C + c) * d.HW + j;
            visit(x[off], gy[off]);
        }, flush);
    const double t0 = block_sum(dsg, red);
    const double t1 = block_sum(dsgx, red);
    if (threadIdx.x == 0) {
        partial[((int64_t)c * d.slices + s) * 2 + 0] = t0;
        partial[((int64_t)c * d.slices + s) * 2 + 1] = t1;
    }
}

// one thread per channel: dgamma = sum g*xhat, dbeta = sum g; coefficients for the apply pass
__global__ void k_bn_bwd_finalize(const double *__restrict__ partial, BnDims d, float *__restrict__ dgamma,
                                  float *__restrict__ dbeta, float *__restrict__ coef /* [C][2]: mean(g), mean(g*xhat) */) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.C) return;
    double sg = 0.0, sgx = 0.0;
    for (int k = 0; k < d.slices; ++k) {
        sg += partial[((int64_t)c * d.slices + k) * 2 + 0];
        sgx += partial[((int64_t)c * d.slices + k) * 2 + 1];
    }
    const double n = (double)d.N * d.HW;
    dbeta[c] = (float)sg;
    dgamma[c] = (float)sgx;
    coef[2 * c + 0] = (float)(sg / n);
    coef[2 * c + 1] = (float)(sgx / n);
}

// TRAIN: dx = (g - mean(g) - xhat * mean(g xhat)) * invstd * gamma ; EVAL (fixed statistics): dx = g * invstd * gamma
template <bool RELU, bool TRAIN, int GROUP>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_apply(const float *__restrict__ x, const float *__restrict__ gy,
                                                           float *__restrict__ gx, BnDims d, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const float *__restrict__ coef) {
    const int64_t planes = (int64_t)d.N * d.C;
    const bool vec = (d.HW & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)gy) & 15) == 0 && (((uintptr_t)gx) & 15) == 0;
    const int gl = threadIdx.x % GROUP;
    const int64_t g0 = (int64_t)blockIdx.x * (kThreads / GROUP) + threadIdx.x / GROUP;
    const int64_t gstride = (int64_t)gridDim.x * (kThreads / GROUP);
    for (int64_t pl = g0; pl < planes; pl += gstride) {
        const int c = (int)(pl % d.C);
        const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
        const float mg = TRAIN ? coef[2 * c] : 0.f, mgx = TRAIN ? coef[2 * c + 1] : 0.f;
        const float scale = is * ga;
        const int64_t off = pl * d.HW;
        auto one = [&](float xv, float gv) {
            if (RELU && !(bn_affine(xv, m, is, ga, be) > 0.f)) gv = 0.f;
            return TRAIN ? (gv - mg - ((xv - m) * is) * mgx) * scale : gv * scale;
        };
        if (vec) {
            for (int i = gl; i < d.HW / 4; i += GROUP) {
                const float4 xv = reinterpret_cast<const float4 *>(x + off)[i];
                const float4 gv = reinterpret_cast<const float4 *>(gy + off)[i];
                float4 r = {one(xv.x, gv.x), one(xv.y, gv.y), one(xv.z, gv.z), one(xv.w, gv.w)};
                reinterpret_cast<float4 *>(gx + off)[i] = r;
            }
        } else {
            for (int i = gl; i < d.HW; i += GROUP) gx[off + i] = one(x[off + i], gy[off + i]);
        }
    }
}

int make_dims(int N, int C, int HW, BnDims &d) {
    CPG_REQUIRE(N > 0 && C > 0 && HW > 0, "bn: non-positive dimension");
    d.N = N; d.C = C; d.HW = HW;
    // enough (channel, slice) blocks to fill the chip ~8x (measured flat from 2x to 16x on 224x224 maps, 5 % better than 4x
    // on 56x56), at least one image per slice
    int want = (8 * kCUs + C - 1) / C;
    if (want > N) want = N;
    if (want < 1) want = 1;
    d.imgs_per_slice = (N + want - 1) / want;
    d.slices = (N + d.imgs_per_slice - 1) / d.imgs_per_slice;
    return CPG_OK;
}

// small feature maps: one wave per plane (4 planes per block); otherwise a block per plane
inline bool wave_planes(const BnDims &d) { return d.HW < 4096; }
unsigned plane_grid(const BnDims &d) {
    int64_t groups = (int64_t)d.N * d.C;
    if (wave_planes(d)) groups = (groups + 3) / 4;
    const int64_t cap = (int64_t)kCUs * 16;
    return (unsigned)(groups < cap ? groups : cap);
}
template <bool RELU>
void launch_apply(const BnDims &d, const float *x, float *y, const float *mean, const float *invstd, const float *gamma,
                  const float *beta, hipStream_t stream, const float *res = nullptr, uint8_t *mask = nullptr) {
    if (wave_planes(d))
        hipLaunchKernelGGL((k_bn_apply<RELU, 64>), dim3(plane_grid(d)), dim3(kThreads), 0, stream, x, res, y, d, mean, invstd, gamma, beta, mask);
    else
        hipLaunchKernelGGL((k_bn_apply<RELU, 256>), dim3(plane_grid(d)), dim3(kThreads), 0, stream, x, res, y, d, mean, invstd, gamma, beta, mask);
}
template <bool RELU, bool TRAIN>
void launch_bwd_apply(const BnDims &d, const float *x, const float *gy, float *gx, const float *mean, const float *invstd,
                      const float *gamma, const float *beta, const float *coef, hipStream_t stream) {
    if (wave_planes(d))
        hipLaunchKernelGGL((k_bn_bwd_apply<RELU, TRAIN, 64>), dim3(plane_grid(d)), dim3(kThreads), 0, stream, x, gy, gx, d, mean, invstd, gamma, beta, coef);
    else
        hipLaunchKernelGGL((k_bn_bwd_apply<RELU, TRAIN, 256>), dim3(plane_grid(d)), dim3(kThreads), 0, stream, x, gy, gx, d, mean, invstd, gamma, beta, coef);
}

}  // namespace

extern "C" size_t cpg_bn_workspace_bytes(int32_t N, int32_t C, int32_t HW) {
    BnDims d;
    if (make_dims(N, C, HW, d) != CPG_OK) return 0;
    return (size_t)C * d.slices * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float);
}

// training forward: batch statistics (mean / invstd out, running stats updated in place when given), then y
extern "C" int cpg_bn_relu_fwd_train(const float *x, const float *gamma, const float *beta, float eps, float momentum,
                                     float *running_mean, float *running_var, float *mean, float *invstd, float *y, int32_t N,
                                     int32_t C, int32_t HW, int32_t relu, void *ws, size_t ws_bytes, void *stream_v) {
    BnDims d;
    int rc = make_dims(N, C, HW, d);
    if (rc) return rc;
    CPG_REQUIRE(x && gamma && beta && mean && invstd && y && ws, "cpg_bn_relu_fwd_train: null pointer");
    CPG_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "cpg_bn_relu_fwd_train: running stats must come as a pair");
    if (ws_bytes < cpg_bn_workspace_bytes(N, C, HW)) return fail(CPG_E_WORKSPACE, "cpg_bn_relu_fwd_train: workspace too small");
    hipStream_t stream = (hipStream_t)stream_v;
    double *partial = (double *)ws;
    hipLaunchKernelGGL(k_bn_stats, dim3(C, d.slices), dim3(kThreads), 0, stream, x, d, partial);
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 63) / 64), dim3(64), 0, stream, partial, d, eps, momentum, mean, invstd,
                       running_mean, running_var);
    if (relu) launch_apply<true>(d, x, y, mean, invstd, gamma, beta, stream);
    else launch_apply<false>(d, x, y, mean, invstd, gamma, beta, stream);
    CPG_CHECK_LAUNCH("cpg_bn_relu_fwd_train");
    return CPG_OK;
}

// inference forward with given statistics (mean, invstd = 1/sqrt(running_var + eps) prepared by the caller)
extern "C" int cpg_bn_relu_fwd_eval(const float *x, const float *gamma, const float *beta, const float *mean,
                                    const float *invstd, float *y, int32_t N, int32_t C, int32_t HW, int32_t relu,
                                    void *stream_v) {
    BnDims d;
    int rc = make_dims(N, C, HW, d);
    if (rc) return rc;
    CPG_REQUIRE(x && gamma && beta && mean && invstd && y, "cpg_bn_relu_fwd_eval: null pointer");
    hipStream_t stream = (hipStream_t)stream_v;
    if (relu) launch_apply<true>(d, x, y, mean, invstd, gamma, beta, stream);
    else launch_apply<false>(d, x, y, mean, invstd, gamma, beta, stream);
    CPG_CHECK_LAUNCH("cpg_bn_relu_fwd_eval");
    return CPG_OK;
}

// Batch statistics from the per-(channel, pixel tile) partial sums a conv forward already produced
// (cpg_conv2d_fwd_bnstats): partials[c][tile] = {sum y, sum y^2} in fp32 over <= 448 outputs each; merged in fp64 in a
// fixed order.  Writes mean / invstd (and updates the running statistics) exactly like k_bn_finalize, so the apply
// pass is cpg_bn_relu_fwd_eval / cpg_bn_relu_pool_fwd(train = 0) and BatchNorm never re-reads y for statistics.
namespace {
__global__ __launch_bounds__(kThreads) void k_bn_finalize_tiles(const float *__restrict__ partials, int tiles, double count, float eps,
                                                                float momentum, float *__restrict__ mean, float *__restrict__ invstd,
                                                                float *__restrict__ running_mean, float *__restrict__ running_var,
                                                                long long *__restrict__ num_batches_tracked) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    if (num_batches_tracked != nullptr && c == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const float2 *p = reinterpret_cast<const float2 *>(partials) + (int64_t)c * tiles;
    // four loads in flight per thread (a 224 x 224 x 256-image layer has 28 672 tiles per channel and only 64 channels = 64 blocks:
    // the loop is latency-bound); four partial sums merged in a fixed order
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    int t = threadIdx.x;
    for (; t + 3 * kThreads < tiles; t += 4 * kThreads) {
        const float2 a = p[t], b = p[t + kThreads], c2 = p[t + 2 * kThreads], d = p[t + 3 * kThreads];
        s0 += (double)a.x;
        q0 += (double)a.y;
        s1 += (double)b.x;
        q1 += (double)b.y;
        s2 += (double)c2.x;
        q2 += (double)c2.y;
        s3 += (double)d.x;
        q3 += (double)d.y;
    }
    for (; t < tiles; t += kThreads) {
        const float2 v = p[t];
        s0 += (double)v.x;
        q0 += (double)v.y;
    }
    const double s = (s0 + s1) + (s2 + s3), ss = (q0 + q1) + (q2 + q3);
    const double ts = block_sum(s, red);
    const double tss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        const double m = ts / count;
        double var = tss / count - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean != nullptr) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
}
}  // namespace

static int bn_stats_finalize(const char *what, const float *partials, int32_t tiles, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                             float *running_mean, float *running_var, float *mean, float *invstd, int64_t *nbt, void *stream_v) {
    CPG_REQUIRE(partials && mean && invstd, "%s: null pointer", what);
    CPG_REQUIRE(tiles > 0 && N > 0 && C > 0 && HW > 0, "%s: non-positive dimension", what);
    CPG_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "%s: running stats must come as a pair", what);
    CPG_REQUIRE((((uintptr_t)partials) & 7) == 0, "%s: partials must be 8-byte aligned", what);
    CPG_REQUIRE((((uintptr_t)nbt) & 7) == 0, "%s: num_batches_tracked must be 8-byte aligned", what);
    hipLaunchKernelGGL(k_bn_finalize_tiles, dim3(C), dim3(kThreads), 0, (hipStream_t)stream_v, partials, tiles, (double)N * HW, eps,
                       momentum, mean, invstd, running_mean, running_var, (long long *)nbt);
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}

extern "C" int cpg_bn_stats_finalize(const float *partials, int32_t tiles, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                     float *running_mean, float *running_var, float *mean, float *invstd, void *stream_v) {
    return bn_stats_finalize("cpg_bn_stats_finalize", partials, tiles, N, C, HW, eps, momentum, running_mean, running_var, mean, invstd, nullptr, stream_v);
}

extern "C" int cpg_bn_stats_finalize_count(const float *partials, int32_t tiles, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                           float *running_mean, float *running_var, float *mean, float *invstd, int64_t *num_batches_tracked,
                                           void *stream_v) {
    return bn_stats_finalize("cpg_bn_stats_finalize_count", partials, tiles, N, C, HW, eps, momentum, running_mean, running_var, mean, invstd,
                             num_batches_tracked, stream_v);
}

// Backward of y = relu(bn(x)) whose reduction already happened in the epilogue of the NEXT layer's input-gradient kernel
// (cpg_conv2d_dgrad_bnbwd): gm = g * [y > 0] and partials[c][tile] = {sum gm, sum gm * xhat} in fp32 per pixel tile.  Merge in fp64
// in a fixed order -> dgamma, dbeta and the two means, then the plain apply pass dx = (gm - mean(gm) - xhat * mean(gm xhat)) *
// invstd * gamma.  2 reads + 1 write instead of 4 reads + 1 write.
namespace {
__global__ __launch_bounds__(kThreads) void k_bn_bwd_finalize_tiles(const float *__restrict__ partials, int tiles, double count,
                                                                    float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                    float *__restrict__ coef) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    const float2 *p = reinterpret_cast<const float2 *>(partials) + (int64_t)c * tiles;
    double s = 0.0, sx = 0.0;
    for (int t = threadIdx.x; t < tiles; t += kThreads) {
        const float2 v = p[t];
        s += (double)v.x;
        sx += (double)v.y;
    }
    const double ts = block_sum(s, red);
    const double tsx = block_sum(sx, red);
    if (threadIdx.x == 0) {
        dbeta[c] = (float)ts;
        dgamma[c] = (float)tsx;
        coef[2 * c + 0] = (float)(ts / count);
        coef[2 * c + 1] = (float)(tsx / count);
    }
}
}  // namespace

extern "C" int cpg_bn_bwd_from_partials(const float *partials, int32_t tiles, const float *x, const float *gm, const float *gamma,
                                        const float *beta, const float *mean, const float *invstd, float *gx, float *dgamma,
                                        float *dbeta, int32_t N, int32_t C, int32_t HW, void *ws, size_t ws_bytes, void *stream_v) {
    BnDims d;
    int rc = make_dims(N, C, HW, d);
    if (rc) return rc;
    CPG_REQUIRE(partials && x && gm && gamma && beta && mean && invstd && gx && dgamma && dbeta && ws, "cpg_bn_bwd_from_partials: null pointer");
    CPG_REQUIRE(tiles > 0 && (((uintptr_t)partials) & 7) == 0, "cpg_bn_bwd_from_partials: bad partial sums");
    if (ws_bytes < (size_t)C * 2 * sizeof(float)) return fail(CPG_E_WORKSPACE, "cpg_bn_bwd_from_partials: workspace too small");
    hipStream_t stream = (hipStream_t)stream_v;
    float *coef = (float *)ws;
    hipLaunchKernelGGL(k_bn_bwd_finalize_tiles, dim3(C), dim3(kThreads), 0, stream, partials, tiles, (double)N * HW, dgamma, dbeta, coef);
    launch_bwd_apply<false, true>(d, x, gm, gx, mean, invstd, gamma, beta, coef, stream);
    CPG_CHECK_LAUNCH("cpg_bn_bwd_from_partials");
    return CPG_OK;
}

// the merge step of cpg_bn_bwd_from_partials on its own (the fused stem applies the result in its own kernel): partials[C][tiles][2]
// = {sum gm, sum gm xhat} -> dbeta, dgamma and coef[2 c] = mean(gm), coef[2 c + 1] = mean(gm xhat) over N * HW elements
extern "C" int cpg_bn_bwd_finalize_partials(const float *partials, int32_t tiles, int32_t N, int32_t C, int32_t HW, float *dgamma,
                                            float *dbeta, float *coef, void *stream_v) {
    CPG_REQUIRE(partials && dgamma && dbeta && coef, "cpg_bn_bwd_finalize_partials: null pointer");
    CPG_REQUIRE(tiles > 0 && N > 0 && C > 0 && HW > 0 && (((uintptr_t)partials) & 7) == 0, "cpg_bn_bwd_finalize_partials: bad partial sums");
    hipLaunchKernelGGL(k_bn_bwd_finalize_tiles, dim3(C), dim3(kThreads), 0, (hipStream_t)stream_v, partials, tiles, (double)N * HW, dgamma,
                       dbeta, coef);
    CPG_CHECK_LAUNCH("cpg_bn_bwd_finalize_partials");
    return CPG_OK;
}

// y = relu(bn(x) + res): the tail of a residual block (models/resnet.py: `out = bn3(conv3(out)); out += identity;
// relu(out)`) in the same two passes as plain BN -- 3 activation passes instead of 8 for the stock bn / add_ / relu_.
// relu_mask (may be NULL; cpg_bn_add_relu_mask_bytes(N, C, HW) bytes, 0 = this shape has no mask): receives the ReLU mask of y, one byte
// per four elements, for cpg_bn_add_relu_bwd.
extern "C" size_t cpg_bn_add_relu_mask_bytes(int32_t N, int32_t C, int32_t HW) {
    return (N > 0 && C > 0 && HW > 0 && HW % 4 == 0) ? (size_t)N * C * (HW / 4) : 0;
}
extern "C" int cpg_bn_add_relu_fwd(const float *x, const float *res, const float *gamma, const float *beta, float eps, float momentum,
                                   float *running_mean, float *running_var, float *mean, float *invstd, float *y, int32_t N,
                                   int32_t C, int32_t HW, int32_t train, void *ws, size_t ws_bytes, void *stream_v, uint8_t *relu_mask) {
    BnDims d;
    int rc = make_dims(N, C, HW, d);
    if (rc) return rc;
    CPG_REQUIRE(x && res && gamma && beta && mean && invstd && y, "cpg_bn_add_relu_fwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_v;
    if (train) {
        CPG_REQUIRE(ws != nullptr, "cpg_bn_add_relu_fwd: null workspace");
        CPG_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "cpg_bn_add_relu_fwd: running stats must come as a pair");
        if (ws_bytes < cpg_bn_workspace_bytes(N, C, HW)) return fail(CPG_E_WORKSPACE, "cpg_bn_add_relu_fwd: workspace too small");
        double *partial = (double *)ws;
        hipLaunchKernelGGL(k_bn_stats, dim3(C, d.slices), dim3(kThreads), 0, stream, x, d, partial);
        hipLaunchKernelGGL(k_bn_finalize, dim3((C + 63) / 64), dim3(64), 0, stream, partial, d, eps, momentum, mean, invstd,
                           running_mean, running_var);
    }
    if (relu_mask != nullptr)
        CPG_REQUIRE(HW % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res)) & 15) == 0,
                    "cpg_bn_add_relu_fwd: the ReLU mask needs 4 | HW and 16-byte aligned tensors");
    launch_apply<true>(d, x, y, mean, invstd, gamma, beta, stream, res, relu_mask);
    CPG_CHECK_LAUNCH("cpg_bn_add_relu_fwd");
    return CPG_OK;
}

// backward of y = [relu](bn(x)); train != 0: batch statistics were used (full BN gradient), else fixed statistics
extern "C" int cpg_bn_relu_bwd(const float *x, const float *gy, const float *gamma, const float *beta, const float *mean,
                               const float *invstd, float *gx, float *dgamma, float *dbeta, int32_t N, int32_t C, int32_t HW,
                               int32_t relu, int32_t train, void *ws, size_t ws_bytes, void *stream_v) {
    BnDims d;
    int rc = make_dims(N, C, HW, d);
    if (rc) return rc;
    CPG_REQUIRE(x && gy && gamma && beta && mean && invstd && gx && dgamma && dbeta && ws, "cpg_bn_relu_bwd: null pointer");
    if (ws_bytes < cpg_bn_workspace_bytes(N, C, HW)) return fail(CPG_E_WORKSPACE, "cpg_bn_relu_bwd: workspace too small");
    hipStream_t stream = (hipStream_t)stream_v;
    double *partial = (double *)ws;
    float *coef = (float *)(partial + (size_t)C * d.slices * 2);
    if (relu)
        hipLaunchKernelGGL(k_bn_bwd_reduce<true>, dim3(C, d.slices), dim3(kThreads), 0, stream, x, gy, d, mean, invstd, gamma, beta, partial);
    else
        hipLaunchKernelGGL(k_bn_bwd_reduce<false>, dim3(C, d.slices), dim3(kThreads), 0, stream, x, gy, d, mean, invstd, gamma, beta, partial);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 63) / 64), dim3(64), 0, stream, partial, d, dgamma, dbeta, coef);
    if (relu && train) launch_bwd_apply<true, true>(d, x, gy, gx, mean, invstd, gamma, beta, coef, stream);
    else if (relu) launch_bwd_apply<true, false>(d, x, gy, gx, mean, invstd, gamma, beta, coef, stream);
    else if (train) launch_bwd_apply<false, true>(d, x, gy, gx, mean, invstd, gamma, beta, coef, stream);
    else launch_bwd_apply<false, false>(d, x, gy, gx, mean, invstd, gamma, beta, coef, stream);
    CPG_CHECK_LAUNCH("cpg_bn_relu_bwd");
    return CPG_OK;
}

// backward of out = relu(bn(x) + res) (cpg_bn_add_relu_fwd; models/resnet.py:62-71,96-104): gz = gy * [out > 0] is the gradient of
// the residual branch AND of bn(x); one pass reads x, gy, out, writes gz and reduces {sum gz, sum gz * xhat}, the second applies
// the BatchNorm gradient from x and gz -- 7 activation passes where threshold_backward + cpg_bn_relu_bwd made 8 (and one launch
// of a stock elementwise kernel less).  gz must NOT alias gy (the kernel declares both __restrict__).
// relu_mask (round 4, may be NULL): the byte mask cpg_bn_add_relu_fwd wrote; with it `out` is not read (and may be NULL) -- 6 1/16 passes.
extern "C" int cpg_bn_add_relu_bwd(const float *x, const float *out, const float *gy, const float *gamma, const float *beta,
                                   const float *mean, const float *invstd, float *gx, float *gz, float *dgamma, float *dbeta, int32_t N,
                                   int32_t C, int32_t HW, int32_t train, void *ws, size_t ws_bytes, void *stream_v, const uint8_t *relu_mask) {
    BnDims d;
    int rc = make_dims(N, C, HW, d);
    if (rc) return rc;
    CPG_REQUIRE(x && (out || relu_mask) && gy && gamma && beta && mean && invstd && gx && gz && dgamma && dbeta && ws, "cpg_bn_add_relu_bwd: null pointer");
    if (relu_mask != nullptr)
        CPG_REQUIRE(HW % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)gz)) & 15) == 0,
                    "cpg_bn_add_relu_bwd: the ReLU mask needs 4 | HW and 16-byte aligned tensors");
    if (ws_bytes < cpg_bn_workspace_bytes(N, C, HW)) return fail(CPG_E_WORKSPACE, "cpg_bn_add_relu_bwd: workspace too small");
    hipStream_t stream = (hipStream_t)stream_v;
    double *partial = (double *)ws;
    float *coef = (float *)(partial + (size_t)C * d.slices * 2);
    hipLaunchKernelGGL((k_bn_bwd_reduce<false, true>), dim3(C, d.slices), dim3(kThreads), 0, stream, x, gy, d, mean, invstd, gamma, beta,
                       partial, out, gz, relu_mask);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 63) / 64), dim3(64), 0, stream, partial, d, dgamma, dbeta, coef);
    if (train) launch_bwd_apply<false, true>(d, x, gz, gx, mean, invstd, gamma, beta, coef, stream);
    else launch_bwd_apply<false, false>(d, x, gz, gx, mean, invstd, gamma, beta, coef, stream);
    CPG_CHECK_LAUNCH("cpg_bn_add_relu_bwd");
    return CPG_OK;
}

// =====================================================================================================
// BatchNorm2d -> ReLU -> MaxPool2d(2, 2) in one op (5 of the 13 VGG blocks, models/vgg.py:131-141).
// The un-pooled activation never exists in HBM: forward reads x once (after the statistics pass) and
// writes the pooled tensor (1/4 size); backward recomputes the four ReLU(BN(x)) values of each window
// from x, routes the pooled gradient to the first maximum (torch's max_pool2d tie rule) if it is
// positive, and continues with the BatchNorm gradient.  H and W must be even.
// =====================================================================================================
namespace {

struct PoolDims {
    int H, W, OW, wins;          // wins = (H/2) * (W/2) windows per plane
};

// the four values of window (pr, pc) of a plane and their common processing
struct Win {
    float v[4];
};
__device__ __forceinline__ Win load_win(const float *plane, int W, int pr, int pc) {
    const float2 a = *reinterpret_cast<const float2 *>(plane + (2 * pr) * W + 2 * pc);
    const float2 b = *reinterpret_cast<const float2 *>(plane + (2 * pr + 1) * W + 2 * pc);
    return Win{{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ int win_argmax(const float (&y)[4]) {
    int idx = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (y[k] > y[idx]) idx = k;      // strict: first maximum wins, as torch's max_pool2d
    return idx;
}

template <int GROUP>
__global__ __launch_bounds__(kThreads) void k_bnp_apply(const float *__restrict__ x, float *__restrict__ yp, BnDims d, PoolDims p,
                                                        const float *__restrict__ mean, const float *__restrict__ invstd,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta) {
    const int64_t planes = (int64_t)d.N * d.C;
    const int gl = threadIdx.x % GROUP;
    const int64_t g0 = (int64_t)blockIdx.x * (kThreads / GROUP) + threadIdx.x / GROUP;
    const int64_t gstride = (int64_t)gridDim.x * (kThreads / GROUP);
    for (int64_t pl = g0; pl < planes; pl += gstride) {
        const int c = (int)(pl % d.C);
        const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
        const float *plane = x + pl * d.HW;
        float *out = yp + pl * p.wins;
        for (int i = gl; i < p.wins; i += GROUP) {
            const int pr = i / p.OW, pc = i - pr * p.OW;
            const Win w = load_win(plane, p.W, pr, pc);
            float best = 0.0f;                       // ReLU floor
#pragma unroll
            for (int k = 0; k < 4; ++k) best = fmaxf(best, bn_affine(w.v[k], m, is, ga, be));
            out[i] = best;
        }
    }
}

// grid (C, slices): partial = {sum g, sum g * xhat} with g scattered from the pooled gradient
__global__ __launch_bounds__(kThreads) void k_bnp_bwd_reduce(const float *__restrict__ x, const float *__restrict__ gp, BnDims d,
                                                             PoolDims p, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, double *__restrict__ partial) {
    __shared__ double red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const int n0 = s * d.imgs_per_slice, n1 = min(d.N, n0 + d.imgs_per_slice);
    const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    double dsg = 0.0, dsgx = 0.0;
    float sg = 0.f, sgx = 0.f;
    walk_planes(p.wins, n0, n1, [&](int n, int wi) {
        const int pr = wi / p.OW, pc = wi - pr * p.OW;
        const int64_t pl = (int64_t)n * d.C + c;
        const Win w = load_win(x + pl * d.HW, p.W, pr, pc);
        const float g = gp[pl * p.wins + wi];
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = bn_affine(w.v[k], m, is, ga, be);
        const int idx = win_argmax(y);
        if (y[idx] > 0.f) {
            sg += g;
            sgx += g * ((w.v[idx] - m) * is);
        }
    }, [&]() {
        dsg += (double)sg; dsgx += (double)sgx; sg = 0.f; sgx = 0.f;
    });
    const double t0 = block_sum(dsg, red);
    const double t1 = block_sum(dsgx, red);
    if (threadIdx.x == 0) {
        partial[((int64_t)c * d.slices + s) * 2 + 0] = t0;
        partial[((int64_t)c * d.slices + s) * 2 + 1] = t1;
    }
}

template <bool TRAIN, int GROUP>
__global__ __launch_bounds__(kThreads) void k_bnp_bwd_apply(const float *__restrict__ x, const float *__restrict__ gp,
                                                            float *__restrict__ gx, BnDims d, PoolDims p,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ coef) {
    const int64_t planes = (int64_t)d.N * d.C;
    const int gl = threadIdx.x % GROUP;
    const int64_t g0 = (int64_t)blockIdx.x * (kThreads / GROUP) + threadIdx.x / GROUP;
    const int64_t gstride = (int64_t)gridDim.x * (kThreads / GROUP);
    for (int64_t pl = g0; pl < planes; pl += gstride) {
        const int c = (int)(pl % d.C);
        const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
        const float mg = TRAIN ? coef[2 * c] : 0.f, mgx = TRAIN ? coef[2 * c + 1] : 0.f;
        const float scale = is * ga;
        const float *plane = x + pl * d.HW;
        float *gplane = gx + pl * d.HW;
        for (int i = gl; i < p.wins; i += GROUP) {
            const int pr = i / p.OW, pc = i - pr * p.OW;
            const Win w = load_win(plane, p.W, pr, pc);
            float y[4], r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = bn_affine(w.v[k], m, is, ga, be);
            const int idx = win_argmax(y);
            const float g = (y[idx] > 0.f) ? gp[pl * p.wins + i] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = (k == idx) ? g : 0.f;
                r[k] = TRAIN ? (gk - mg - ((w.v[k] - m) * is) * mgx) * scale : gk * scale;
            }
            *reinterpret_cast<float2 *>(gplane + (2 * pr) * p.W + 2 * pc) = make_float2(r[0], r[1]);
            *reinterpret_cast<float2 *>(gplane + (2 * pr + 1) * p.W + 2 * pc) = make_float2(r[2], r[3]);
        }
    }
}

int make_pool(int H, int W, PoolDims &p) {
    CPG_REQUIRE(H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0, "bn_relu_pool: H and W must be even (got %d x %d)", H, W);
    p.H = H; p.W = W; p.OW = W / 2; p.wins = (H / 2) * (W / 2);
    return CPG_OK;
}
unsigned pool_grid(const BnDims &d, const PoolDims &p, bool wave) {
    int64_t groups = (int64_t)d.N * d.C;
    if (wave) groups = (groups + 3) / 4;
    const int64_t cap = (int64_t)kCUs * 16;
    return (unsigned)(groups < cap ? groups : cap);
}

}  // namespace

extern "C" int cpg_bn_relu_pool_fwd(const float *x, const float *gamma, const float *beta, float eps, float momentum,
                                    float *running_mean, float *running_var, float *mean, float *invstd, float *y_pooled,
                                    int32_t N, int32_t C, int32_t H, int32_t W, int32_t train, void *ws, size_t ws_bytes,
                                    void *stream_v) {
    BnDims d;
    PoolDims p;
    int rc = make_dims(N, C, H * W, d);
    if (rc) return rc;
    rc = make_pool(H, W, p);
    if (rc) return rc;
    CPG_REQUIRE(x && gamma && beta && mean && invstd && y_pooled, "cpg_bn_relu_pool_fwd: null pointer");
    CPG_REQUIRE((((uintptr_t)x) & 7) == 0, "cpg_bn_relu_pool_fwd: x must be 8-byte aligned");
    hipStream_t stream = (hipStream_t)stream_v;
    if (train) {
        CPG_REQUIRE(ws != nullptr, "cpg_bn_relu_pool_fwd: workspace required in training mode");
        if (ws_bytes < cpg_bn_workspace_bytes(N, C, H * W)) return fail(CPG_E_WORKSPACE, "cpg_bn_relu_pool_fwd: workspace too small");
        double *partial = (double *)ws;
        hipLaunchKernelGGL(k_bn_stats, dim3(C, d.slices), dim3(kThreads), 0, stream, x, d, partial);
        hipLaunchKernelGGL(k_bn_finalize, dim3((C + 63) / 64), dim3(64), 0, stream, partial, d, eps, momentum, mean, invstd,
                           running_mean, running_var);
    }
    const bool wave = p.wins < 1024;
    if (wave)
        hipLaunchKernelGGL(k_bnp_apply<64>, dim3(pool_grid(d, p, true)), dim3(kThreads), 0, stream, x, y_pooled, d, p, mean, invstd, gamma, beta);
    else
        hipLaunchKernelGGL(k_bnp_apply<256>, dim3(pool_grid(d, p, false)), dim3(kThreads), 0, stream, x, y_pooled, d, p, mean, invstd, gamma, beta);
    CPG_CHECK_LAUNCH("cpg_bn_relu_pool_fwd");
    return CPG_OK;
}

extern "C" int cpg_bn_relu_pool_bwd(const float *x, const float *g_pooled, const float *gamma, const float *beta,
                                    const float *mean, const float *invstd, float *gx, float *dgamma, float *dbeta, int32_t N,
                                    int32_t C, int32_t H, int32_t W, int32_t train, void *ws, size_t ws_bytes, void *stream_v) {
    BnDims d;
    PoolDims p;
    int rc = make_dims(N, C, H * W, d);
    if (rc) return rc;
    rc = make_pool(H, W, p);
    if (rc) return rc;
    CPG_REQUIRE(x && g_pooled && gamma && beta && mean && invstd && gx && dgamma && dbeta && ws, "cpg_bn_relu_pool_bwd: null pointer");
    CPG_REQUIRE((((uintptr_t)x) & 7) == 0 && (((uintptr_t)gx) & 7) == 0, "cpg_bn_relu_pool_bwd: x / gx must be 8-byte aligned");
    if (ws_bytes < cpg_bn_workspace_bytes(N, C, H * W)) return fail(CPG_E_WORKSPACE, "cpg_bn_relu_pool_bwd: workspace too small");
    hipStream_t stream = (hipStream_t)stream_v;
    double *partial = (double *)ws;
    float *coef = (float *)(partial + (size_t)C * d.slices * 2);
    hipLaunchKernelGGL(k_bnp_bwd_reduce, dim3(C, d.slices), dim3(kThreads), 0, stream, x, g_pooled, d, p, mean, invstd, gamma, beta, partial);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 63) / 64), dim3(64), 0, stream, partial, d, dgamma, dbeta, coef);
    const bool wave = p.wins < 1024;
    const dim3 grid(pool_grid(d, p, wave)), block(kThreads);
    if (wave && train) hipLaunchKernelGGL((k_bnp_bwd_apply<true, 64>), grid, block, 0, stream, x, g_pooled, gx, d, p, mean, invstd, gamma, beta, coef);
    else if (wave) hipLaunchKernelGGL((k_bnp_bwd_apply<false, 64>), grid, block, 0, stream, x, g_pooled, gx, d, p, mean, invstd, gamma, beta, coef);
    else if (train) hipLaunchKernelGGL((k_bnp_bwd_apply<true, 256>), grid, block, 0, stream, x, g_pooled, gx, d, p, mean, invstd, gamma, beta, coef);
    else hipLaunchKernelGGL((k_bnp_bwd_apply<false, 256>), grid, block, 0, stream, x, g_pooled, gx, d, p, mean, invstd, gamma, beta, coef);
    CPG_CHECK_LAUNCH("cpg_bn_relu_pool_bwd");
    return CPG_OK;
}


// =====================================================================================================
// BatchNorm2d -> ReLU -> MaxPool2d(kernel 3, stride 2, padding 1): the ResNet stem's tail (models/resnet.py:127-129,208-211),
// the largest activation of the network (64 x 112 x 112 per image).  Stock torch: BN apply (2 passes), max-pool forward with an
// int64 index tensor, max_pool_backward (1.6 ms at batch 256: a gather over the indices), then the BatchNorm backward (5 passes).
// Here a block owns a strip of one (n, c) plane at a time: its rows go through LDS as z = relu(bn(x)) (windows overlap, so neighbours are
// needed), the pooled maxima are written; backward recomputes z the same way, finds every window's arg-max (first maximum in
// row-major order = torch's rule; padding never wins: z >= 0 and every window has a valid element), routes the pooled gradient to
// it without atomics (each element looks at the <= 4 windows that contain it), masks by z > 0 and continues with the BatchNorm
// gradient -- a reduction launch and an apply launch that each read x and the pooled gradient once.
// Work unit = a strip of 8 window rows of one (n, c) plane (19 input rows in LDS: a dozen blocks per CU).
// =====================================================================================================
namespace {

struct Pool3Dims {
    int H, W, OH, OW;
    int strips;                 // strips of kPool3SR window rows per plane
};
constexpr int kPool3SR = 8;     // window (= quad) rows per work unit; a unit stages 2 SR + 3 input rows and SR + 1 window rows

// LDS layout of a unit: zs[(2 SR + 3) * W] = relu(bn(x)) of input rows 2 r0 - 1 ... 2 r0 + 2 SR + 1 (r0 = first window row of the
// strip; rows outside the plane are never read), idx[(SR + 1) * OW] = arg-max (flat plane index) of window rows r0 ... r0 + SR
__device__ __forceinline__ void pool3_stage(const float *__restrict__ plane, float *zs, const Pool3Dims &p, int r0, int nrows, float m,
                                            float is, float ga, float be) {
    const int h0 = 2 * r0 - 1;                                   // input row of LDS row 0
    const int lo = max(h0, 0), hi = min(h0 + nrows, p.H);        // valid input rows [lo, hi)
    const float *src = plane + lo * p.W;
    float *dst = zs + (lo - h0) * p.W;
    const int n = (hi - lo) * p.W;
    if ((n & 3) == 0 && (((uintptr_t)src) & 15) == 0 && (((lo - h0) * p.W) & 3) == 0) {
        for (int i = threadIdx.x; i < n / 4; i += kThreads) {
            float4 v = reinterpret_cast<const float4 *>(src)[i];
            v.x = fmaxf(bn_affine(v.x, m, is, ga, be), 0.f); v.y = fmaxf(bn_affine(v.y, m, is, ga, be), 0.f);
            v.z = fmaxf(bn_affine(v.z, m, is, ga, be), 0.f); v.w = fmaxf(bn_affine(v.w, m, is, ga, be), 0.f);
            reinterpret_cast<float4 *>(dst)[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += kThreads) dst[i] = fmaxf(bn_affine(src[i], m, is, ga, be), 0.f);
    }
}

// arg-max of window (ph, pw): flat PLANE index h * W + w of the first maximum (row-major scan over the valid positions; padding never
// takes part, as in torch); zs row 0 = input row h0
__device__ __forceinline__ int pool3_argmax(const float *zs, const Pool3Dims &p, int h0, int ph, int pw, float *best_out = nullptr) {
    float best = -1.0f;
    int arg = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int h = 2 * ph - 1 + r;
        if ((unsigned)h >= (unsigned)p.H) continue;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int w = 2 * pw - 1 + q;
            if ((unsigned)w >= (unsigned)p.W) continue;
            const float v = zs[(h - h0) * p.W + w];
            if (v > best || v != v) best = v, arg = h * p.W + w;      // (a NaN wins and stays: torch's max_pool2d propagates it)
        }
    }
    if (best_out) *best_out = best;
    return arg;
}

__global__ __launch_bounds__(kThreads) void k_bnp3_apply(const float *__restrict__ x, float *__restrict__ y, BnDims d, Pool3Dims p,
                                                         const float *__restrict__ mean, const float *__restrict__ invstd,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float *zs = dyn;
    const int64_t units = (int64_t)d.N * d.C * p.strips;
    const int wins = p.OH * p.OW;
    for (int64_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int64_t pl = u / p.strips;
        const int r0 = (int)(u % p.strips) * kPool3SR, r1 = min(p.OH, r0 + kPool3SR);
        const int c = (int)(pl % d.C);
        __syncthreads();
        pool3_stage(x + pl * d.HW, zs, p, r0, 2 * (r1 - r0) + 1, mean[c], invstd[c], gamma[c], beta[c]);
        __syncthreads();
        for (int i = threadIdx.x; i < (r1 - r0) * p.OW; i += kThreads) {
            float best;
            pool3_argmax(zs, p, 2 * r0 - 1, r0 + i / p.OW, i % p.OW, &best);
            y[pl * wins + r0 * p.OW + i] = best;
        }
    }
}

// One thread = one 2 x 2 QUAD of the plane, quad (a, b) = elements (2a + dy, 2b + dx): element (0,0) lies in window (a, b) only,
// (0,1) in (a, b) and (a, b+1), (1,0) in (a, b) and (a+1, b), (1,1) in all four -- so the four windows' arg-max entries and pooled
// gradients are fetched once per quad (pooled gradients straight from global memory: consecutive quads read consecutive words).
// dz[2 dy + dx] = the pooled gradients routed to the element, masked by z > 0 (elements past the plane's edge get 0).
__device__ __forceinline__ void pool3_quad(const float *zs, const int *idx, const float *__restrict__ gp, const Pool3Dims &p, int r0, int a,
                                           int b, float (&dz)[4]) {
    const int h0 = 2 * r0 - 1;
    const int *ix = idx + (a - r0) * p.OW + b;
    const float *g = gp + a * p.OW + b;
    const bool hb = b + 1 < p.OW, ha = a + 1 < p.OH;
    const int i00 = ix[0], i01 = hb ? ix[1] : -1, i10 = ha ? ix[p.OW] : -1, i11 = (ha && hb) ? ix[p.OW + 1] : -1;
    const float g00 = g[0], g01 = hb ? g[1] : 0.f, g10 = ha ? g[p.OW] : 0.f, g11 = (ha && hb) ? g[p.OW + 1] : 0.f;
    const int e00 = 2 * a * p.W + 2 * b;
    const float *z = zs + (2 * a - h0) * p.W + 2 * b;
    const bool hx = 2 * b + 1 < p.W, hy = 2 * a + 1 < p.H;
    dz[0] = z[0] > 0.f ? (i00 == e00 ? g00 : 0.f) : 0.f;
    dz[1] = (hx && z[1] > 0.f) ? (i00 == e00 + 1 ? g00 : 0.f) + (i01 == e00 + 1 ? g01 : 0.f) : 0.f;
    dz[2] = (hy && z[p.W] > 0.f) ? (i00 == e00 + p.W ? g00 : 0.f) + (i10 == e00 + p.W ? g10 : 0.f) : 0.f;
    const int e11 = e00 + p.W + 1;
    dz[3] = (hx && hy && z[p.W + 1] > 0.f)
                ? ((i00 == e11 ? g00 : 0.f) + (i01 == e11 ? g01 : 0.f)) + ((i10 == e11 ? g10 : 0.f) + (i11 == e11 ? g11 : 0.f))
                : 0.f;
}

// stage a strip for the backward kernels: z rows and the arg-max table of window rows r0 ... min(OH, r1 + 1) - 1
__device__ __forceinline__ void pool3_stage_bwd(const float *__restrict__ plane, float *zs, int *idx, const Pool3Dims &p, int r0, int r1,
                                                float m, float is, float ga, float be) {
    const int wr = min(p.OH, r1 + 1) - r0;                       // window rows needed (one past the strip for its last quads)
    __syncthreads();
    pool3_stage(plane, zs, p, r0, 2 * wr + 1, m, is, ga, be);
    __syncthreads();
    for (int i = threadIdx.x; i < wr * p.OW; i += kThreads) idx[i] = pool3_argmax(zs, p, 2 * r0 - 1, r0 + i / p.OW, i % p.OW);
    __syncthreads();
}

// grid (C, slices): partial[c][slice] = {sum dz, sum dz * xhat}
__global__ __launch_bounds__(kThreads) void k_bnp3_bwd_reduce(const float *__restrict__ x, const float *__restrict__ gp, BnDims d, Pool3Dims p,
                                                              const float *__restrict__ mean, const float *__restrict__ invstd,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta,
                                                              double *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    __shared__ double red[4];
    float *zs = dyn;
    int *idx = reinterpret_cast<int *>(dyn + (2 * kPool3SR + 3) * p.W);
    const int c = blockIdx.x, s = blockIdx.y;
    const int n0 = s * d.imgs_per_slice, n1 = min(d.N, n0 + d.imgs_per_slice);
    const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    const int wins = p.OH * p.OW;
    double dsg = 0.0, dsgx = 0.0;
    for (int n = n0; n < n1; ++n) {
        const int64_t pl = (int64_t)n * d.C + c;
        const float *src = x + pl * d.HW;
        float sg = 0.f, sgx = 0.f;
        for (int st = 0; st < p.strips; ++st) {
            const int r0 = st * kPool3SR, r1 = min(p.OH, r0 + kPool3SR);
            pool3_stage_bwd(src, zs, idx, p, r0, r1, m, is, ga, be);
            // x of the quad is read again unconditionally (the strip was staged a moment ago: L2 hits, coalesced 8-byte pairs) --
            // a load under "dz != 0" is four divergent round trips per quad
            const bool pairs = (p.W & 1) == 0 && (((uintptr_t)src) & 7) == 0;
            for (int i = threadIdx.x; i < (r1 - r0) * p.OW; i += kThreads) {
                const int a = r0 + i / p.OW, b = i % p.OW;
                const int e00 = 2 * a * p.W + 2 * b;
                const bool hx = 2 * b + 1 < p.W, hy = 2 * a + 1 < p.H;
                float2 x0, x1 = make_float2(m, m);
                if (pairs) {
                    x0 = *reinterpret_cast<const float2 *>(src + e00);
                    if (hy) x1 = *reinterpret_cast<const float2 *>(src + e00 + p.W);
                } else {
                    x0 = make_float2(src[e00], hx ? src[e00 + 1] : m);
                    if (hy) x1 = make_float2(src[e00 + p.W], hx ? src[e00 + p.W + 1] : m);
                }
                float dz[4];
                pool3_quad(zs, idx, gp + pl * wins, p, r0, a, b, dz);
                sg += (dz[0] + dz[1]) + (dz[2] + dz[3]);
                sgx += (dz[0] * ((x0.x - m) * is) + dz[1] * ((x0.y - m) * is)) + (dz[2] * ((x1.x - m) * is) + dz[3] * ((x1.y - m) * is));
            }
        }
        dsg += (double)sg;
        dsgx += (double)sgx;
    }
    const double t0 = block_sum(dsg, red);
    const double t1 = block_sum(dsgx, red);
    if (threadIdx.x == 0) {
        partial[((int64_t)c * d.slices + s) * 2 + 0] = t0;
        partial[((int64_t)c * d.slices + s) * 2 + 1] = t1;
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(kThreads) void k_bnp3_bwd_apply(const float *__restrict__ x, const float *__restrict__ gp, float *__restrict__ gx,
                                                             BnDims d, Pool3Dims p, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, const float *__restrict__ coef) {
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float *zs = dyn;
    int *idx = reinterpret_cast<int *>(dyn + (2 * kPool3SR + 3) * p.W);
    const int64_t units = (int64_t)d.N * d.C * p.strips;
    const int wins = p.OH * p.OW;
    for (int64_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int64_t pl = u / p.strips;
        const int r0 = (int)(u % p.strips) * kPool3SR, r1 = min(p.OH, r0 + kPool3SR);
        const int c = (int)(pl % d.C);
        const float m = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
        const float mg = TRAIN ? coef[2 * c] : 0.f, mgx = TRAIN ? coef[2 * c + 1] : 0.f, scale = is * ga;
        const float *src = x + pl * d.HW;
        float *dst = gx + pl * d.HW;
        pool3_stage_bwd(src, zs, idx, p, r0, r1, m, is, ga, be);
        const bool pairs = (p.W & 1) == 0 && (((uintptr_t)src) & 7) == 0 && (((uintptr_t)dst) & 7) == 0;   // rows as 8-byte pairs
        auto out = [&](float xv, float g) { return TRAIN ? (g - mg - ((xv - m) * is) * mgx) * scale : g * scale; };
        for (int i = threadIdx.x; i < (r1 - r0) * p.OW; i += kThreads) {
            const int a = r0 + i / p.OW, b = i % p.OW;
            float dz[4];
            pool3_quad(zs, idx, gp + pl * wins, p, r0, a, b, dz);
            const int e0 = 2 * a * p.W + 2 * b;
            const bool hx = 2 * b + 1 < p.W, hy = 2 * a + 1 < p.H;
            if (pairs) {
                const float2 x0 = *reinterpret_cast<const float2 *>(src + e0);
                *reinterpret_cast<float2 *>(dst + e0) = make_float2(out(x0.x, dz[0]), out(x0.y, dz[1]));
                if (hy) {
                    const float2 x1 = *reinterpret_cast<const float2 *>(src + e0 + p.W);
                    *reinterpret_cast<float2 *>(dst + e0 + p.W) = make_float2(out(x1.x, dz[2]), out(x1.y, dz[3]));
                }
            } else {
                dst[e0] = out(src[e0], dz[0]);
                if (hx) dst[e0 + 1] = out(src[e0 + 1], dz[1]);
                if (hy) dst[e0 + p.W] = out(src[e0 + p.W], dz[2]);
                if (hx && hy) dst[e0 + p.W + 1] = out(src[e0 + p.W + 1], dz[3]);
            }
        }
    }
}

bool make_pool3(int H, int W, Pool3Dims &p) {
    if (H < 1 || W < 1 || W > 512) return false;                 // (a strip of 2 SR + 3 rows must fit LDS)
    p.H = H, p.W = W, p.OH = (H - 1) / 2 + 1, p.OW = (W - 1) / 2 + 1;
    p.strips = (p.OH + kPool3SR - 1) / kPool3SR;
    return true;
}
inline size_t pool3_lds(const Pool3Dims &p) { return ((size_t)(2 * kPool3SR + 3) * p.W + (size_t)(kPool3SR + 1) * p.OW) * sizeof(float); }

}  // namespace

extern "C" int32_t cpg_bn_relu_pool3_supported(int32_t H, int32_t W) {
    Pool3Dims p;
    return make_pool3(H, W, p) ? 1 : 0;
}

// statistics (mean / invstd) are given: from cpg_bn_stats_finalize (training, after cpg_conv2d_fwd_bnstats) or the running ones
extern "C" int cpg_bn_relu_pool3_fwd(const float *x, const float *gamma, const float *beta, const float *mean, const float *invstd,
                                     float *y_pooled, int32_t N, int32_t C, int32_t H, int32_t W, void *stream_v) {
    BnDims d;
    Pool3Dims p;
    int rc = make_dims(N, C, H * W, d);
    if (rc) return rc;
    if (!make_pool3(H, W, p)) return fail(CPG_E_UNSUPPORTED, "cpg_bn_relu_pool3_fwd: plane %d x %d does not fit", H, W);
    CPG_REQUIRE(x && gamma && beta && mean && invstd && y_pooled, "cpg_bn_relu_pool3_fwd: null pointer");
    const int64_t units = (int64_t)N * C * p.strips;
    hipLaunchKernelGGL(k_bnp3_apply, dim3((unsigned)std::min<int64_t>(units, (int64_t)kCUs * 32)), dim3(kThreads), pool3_lds(p),
                       (hipStream_t)stream_v, x, y_pooled, d, p, mean, invstd, gamma, beta);
    CPG_CHECK_LAUNCH("cpg_bn_relu_pool3_fwd");
    return CPG_OK;
}

extern "C" int cpg_bn_relu_pool3_bwd(const float *x, const float *g_pooled, const float *gamma, const float *beta, const float *mean,
                                     const float *invstd, float *gx, float *dgamma, float *dbeta, int32_t N, int32_t C, int32_t H, int32_t W,
                                     int32_t train, void *ws, size_t ws_bytes, void *stream_v) {
    BnDims d;
    Pool3Dims p;
    int rc = make_dims(N, C, H * W, d);
    if (rc) return rc;
    if (!make_pool3(H, W, p)) return fail(CPG_E_UNSUPPORTED, "cpg_bn_relu_pool3_bwd: plane %d x %d does not fit", H, W);
    CPG_REQUIRE(x && g_pooled && gamma && beta && mean && invstd && gx && dgamma && dbeta && ws, "cpg_bn_relu_pool3_bwd: null pointer");
    if (ws_bytes < cpg_bn_workspace_bytes(N, C, H * W)) return fail(CPG_E_WORKSPACE, "cpg_bn_relu_pool3_bwd: workspace too small");
    hipStream_t stream = (hipStream_t)stream_v;
    double *partial = (double *)ws;
    float *coef = (float *)(partial + (size_t)C * d.slices * 2);
    const size_t lds = pool3_lds(p);
    hipLaunchKernelGGL(k_bnp3_bwd_reduce, dim3(C, d.slices), dim3(kThreads), lds, stream, x, g_pooled, d, p, mean, invstd, gamma, beta, partial);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 63) / 64), dim3(64), 0, stream, partial, d, dgamma, dbeta, coef);
    const dim3 grid((unsigned)std::min<int64_t>((int64_t)N * C * p.strips, (int64_t)kCUs * 32));
    if (train) hipLaunchKernelGGL(k_bnp3_bwd_apply<true>, grid, dim3(kThreads), lds, stream, x, g_pooled, gx, d, p, mean, invstd, gamma, beta, coef);
    else hipLaunchKernelGGL(k_bnp3_bwd_apply<false>, grid, dim3(kThreads), lds, stream, x, g_pooled, gx, d, p, mean, invstd, gamma, beta, coef);
    CPG_CHECK_LAUNCH("cpg_bn_relu_pool3_bwd");
    return CPG_OK;
}

// ------------------------------------------------------------------------------------------
// PReLU backward (SphereNet-20's activation, models/spherenet.py: nn.PReLU(channels) after every conv).
//   gx = x > 0 ? g : a[c] * g          ga[c] = sum over (n, pixels) of (x > 0 ? 0 : g * x)
// One pass: the (channel, image-slice) block that streams x and g writes gx and keeps the slope-gradient partial
// (fp32 running sums folded into fp64, fixed-order merge -> deterministic).  torch's prelu_backward materialises a
// full-size per-element slope gradient and reduces it afterwards: 1.2 ms per layer, 36 % of a SphereNet-20 step.
// ------------------------------------------------------------------------------------------
namespace {
// BIAS: also sum gx per channel -- x is the output of a conv WITH bias (every SphereNet conv, models/spherenet.py:203-217), gx is that
// conv's output gradient, and its per-channel sum is the conv's bias gradient: no separate reduction pass over gx (k_conv_bias_*).
template <bool BIAS>
__global__ __launch_bounds__(kThreads) void k_prelu_bwd(const float *__restrict__ x, const float *__restrict__ g,
                                                        const float *__restrict__ slope, float *__restrict__ gx, BnDims d,
                                                        int slope_stride, double *__restrict__ partial) {
    __shared__ double red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const int n0 = s * d.imgs_per_slice, n1 = min(d.N, n0 + d.imgs_per_slice);
    const float a = slope[c * slope_stride];
    double dacc = 0.0, dbias = 0.0;
    float acc = 0.f, accb = 0.f;
    const bool vec = (d.HW & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)g) & 15) == 0 && (((uintptr_t)gx) & 15) == 0;
    auto flush = [&]() {
        dacc += (double)acc; acc = 0.f;
        if (BIAS) dbias += (double)accb, accb = 0.f;
    };
    auto one = [&](float xv, float gv, float &out) {
        const bool pos = xv > 0.f;
        out = pos ? gv : a * gv;
        acc += pos ? 0.f : gv * xv;
        if (BIAS) accb += out;
    };
    if (vec)
        walk_planes(d.HW >> 2, n0, n1, [&](int n, int j) {
            const int64_t off = ((int64_t)n * d.C + c) * d.HW + 4 * j;
            const float4 xv = *reinterpret_cast<const float4 *>(x + off);
            const float4 gv = *reinterpret_cast<const float4 *>(g + off);
            float4 o;
            one(xv.x, gv.x, o.x); one(xv.y, gv.y, o.y); one(xv.z, gv.z, o.z); one(xv.w, gv.w, o.w);
            *reinterpret_cast<float4 *>(gx + off) = o;
        }, flush);
    else
        walk_planes(d.HW, n0, n1, [&](int n, int j) {
            const int64_t off = ((int64_t)n * d.C + c) * d.HW + j;
            float o;
            one(x[off], g[off], o);
            gx[off] = o;
        }, flush);
    const double t = block_sum(dacc, red);
    if (threadIdx.x == 0) partial[(int64_t)c * d.slices + s] = t;
    if (BIAS) {
        const double tb = block_sum(dbias, red);
        if (threadIdx.x == 0) partial[(int64_t)d.C * d.slices + (int64_t)c * d.slices + s] = tb;
    }
}
__global__ void k_prelu_bwd_finalize(const double *__restrict__ partial, BnDims d, int per_channel, float *__restrict__ gslope,
                                     float *__restrict__ gbias) {
    if (gbias != nullptr) {                                   // bias gradient of the conv below: per channel, fixed-order merge
        const int c = blockIdx.x * blockDim.x + threadIdx.x;
        if (c < d.C) {
            double s = 0.0;
            for (int k = 0; k < d.slices; ++k) s += partial[(int64_t)d.C * d.slices + (int64_t)c * d.slices + k];
            gbias[c] = (float)s;
        }
    }
    if (per_channel) {
        const int c = blockIdx.x * blockDim.x + threadIdx.x;
        if (c >= d.C) return;
        double s = 0.0;
        for (int k = 0; k < d.slices; ++k) s += partial[(int64_t)c * d.slices + k];
        gslope[c] = (float)s;
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {        // nn.PReLU(1): one slope shared by all channels
        double s = 0.0;
        for (int64_t k = 0; k < (int64_t)d.C * d.slices; ++k) s += partial[k];
        gslope[0] = (float)s;
    }
}
}  // namespace

extern "C" size_t cpg_prelu_workspace_bytes(int32_t N, int32_t C, int32_t HW) {
    BnDims d;
    if (make_dims(N, C, HW, d) != CPG_OK) return 0;
    return (size_t)2 * C * d.slices * sizeof(double);        // slope-gradient partials, then (cpg_prelu_bwd_bias) the bias-gradient ones
}

// n_slopes: C (one slope per channel) or 1 (shared).  gbias (cpg_prelu_bwd_bias; may be NULL): per-channel sum of gx
static int prelu_bwd(const float *x, const float *gy, const float *slope, float *gx, float *gslope, float *gbias, int32_t N, int32_t C,
                     int32_t HW, int32_t n_slopes, void *ws, size_t ws_bytes, void *stream_v, const char *what) {
    BnDims d;
    int rc = make_dims(N, C, HW, d);
    if (rc) return rc;
    CPG_REQUIRE(x && gy && slope && gx && gslope && ws, "%s: null pointer", what);
    CPG_REQUIRE(n_slopes == C || n_slopes == 1, "%s: n_slopes must be C or 1", what);
    if (ws_bytes < cpg_prelu_workspace_bytes(N, C, HW)) return fail(CPG_E_WORKSPACE, "%s: workspace too small", what);
    hipStream_t stream = (hipStream_t)stream_v;
    double *partial = (double *)ws;
    if (gbias != nullptr)
        hipLaunchKernelGGL(k_prelu_bwd<true>, dim3(C, d.slices), dim3(kThreads), 0, stream, x, gy, slope, gx, d, n_slopes == C ? 1 : 0, partial);
    else
        hipLaunchKernelGGL(k_prelu_bwd<false>, dim3(C, d.slices), dim3(kThreads), 0, stream, x, gy, slope, gx, d, n_slopes == C ? 1 : 0, partial);
    hipLaunchKernelGGL(k_prelu_bwd_finalize, dim3((C + 63) / 64), dim3(64), 0, stream, partial, d, n_slopes == C ? 1 : 0, gslope, gbias);
    CPG_CHECK_LAUNCH(what);
    return CPG_OK;
}
extern "C" int cpg_prelu_bwd(const float *x, const float *gy, const float *slope, float *gx, float *gslope, int32_t N, int32_t C,
                             int32_t HW, int32_t n_slopes, void *ws, size_t ws_bytes, void *stream_v) {
    return prelu_bwd(x, gy, slope, gx, gslope, nullptr, N, C, HW, n_slopes, ws, ws_bytes, stream_v, "cpg_prelu_bwd");
}
extern "C" int cpg_prelu_bwd_bias(const float *x, const float *gy, const float *slope, float *gx, float *gslope, float *gbias, int32_t N,
                                  int32_t C, int32_t HW, int32_t n_slopes, void *ws, size_t ws_bytes, void *stream_v) {
    CPG_REQUIRE(gbias != nullptr, "cpg_prelu_bwd_bias: null bias-gradient pointer");
    return prelu_bwd(x, gy, slope, gx, gslope, gbias, N, C, HW, n_slopes, ws, ws_bytes, stream_v, "cpg_prelu_bwd_bias");
}

// PReLU forward with the residual add of SphereNet's units folded in (models/spherenet.py:219-247: x = x + relu_b(conv_b(relu_a(conv_a(x))))):
// y = res + (v > 0 ? v : a[c] v), res may be null.  One pass (stock: prelu kernel, then an add kernel).
namespace {
template <int GROUP>
__global__ __launch_bounds__(kThreads) void k_prelu_fwd(const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ slope,
                                                        float *__restrict__ y, BnDims d, int per_channel) {
    const int64_t planes = (int64_t)d.N * d.C;
    const bool vec = (d.HW & 3) == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 15) == 0 && (((uintptr_t)res) & 15) == 0;
    const int gl = threadIdx.x % GROUP;
    const int64_t g0 = (int64_t)blockIdx.x * (kThreads / GROUP) + threadIdx.x / GROUP;
    const int64_t gstride = (int64_t)gridDim.x * (kThreads / GROUP);
    for (int64_t pl = g0; pl < planes; pl += gstride) {
        const float a = slope[per_channel ? (int)(pl % d.C) : 0];
        const float *p = x + pl * d.HW;
        const float *r = res ? res + pl * d.HW : nullptr;
        float *q = y + pl * d.HW;
        if (vec) {
            for (int i = gl; i < d.HW / 4; i += GROUP) {
                float4 v = reinterpret_cast<const float4 *>(p)[i];
                v.x = v.x > 0.f ? v.x : a * v.x; v.y = v.y > 0.f ? v.y : a * v.y;
                v.z = v.z > 0.f ? v.z : a * v.z; v.w = v.w > 0.f ? v.w : a * v.w;
                if (r != nullptr) {
                    const float4 t = reinterpret_cast<const float4 *>(r)[i];
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                reinterpret_cast<float4 *>(q)[i] = v;
            }
        } else {
            for (int i = gl; i < d.HW; i += GROUP) {
                const float v = p[i];
                q[i] = (v > 0.f ? v : a * v) + (r ? r[i] : 0.f);
            }
        }
    }
}
}  // namespace

extern "C" int cpg_prelu_fwd(const float *x, const float *res, const float *slope, float *y, int32_t N, int32_t C, int32_t HW,
                             int32_t n_slopes, void *stream_v) {
    BnDims d;
    int rc = make_dims(N, C, HW, d);
    if (rc) return rc;
    CPG_REQUIRE(x && slope && y, "cpg_prelu_fwd: null pointer");
    CPG_REQUIRE(n_slopes == C || n_slopes == 1, "cpg_prelu_fwd: n_slopes must be C or 1");
    hipStream_t stream = (hipStream_t)stream_v;
    if (wave_planes(d))
        hipLaunchKernelGGL(k_prelu_fwd<64>, dim3(plane_grid(d)), dim3(kThreads), 0, stream, x, res, slope, y, d, n_slopes == C ? 1 : 0);
    else
        hipLaunchKernelGGL(k_prelu_fwd<256>, dim3(plane_grid(d)), dim3(kThreads), 0, stream, x, res, slope, y, d, n_slopes == C ? 1 : 0);
    CPG_CHECK_LAUNCH("cpg_prelu_fwd");
    return CPG_OK;
}
