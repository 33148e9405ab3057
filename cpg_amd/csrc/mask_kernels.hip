// HBM-bound mask / gradient-routing kernels (K1, K4e, K7, K8 of SURVEY.md section 2).
// Each is a single streaming pass: 16 B per lane per access for fp32, 4 owner bytes as one
// dword, grid capped at 8 blocks per CU with a grid-stride loop.  gfx950 only.
#include <cmath>
#include "cpg_common.h"

using namespace cpg;

namespace {

constexpr int kThreads = 256;

struct alignas(16) F4 { float x, y, z, w; };

__device__ __forceinline__ bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------- K1
__global__ __launch_bounds__(kThreads) void k_binarize_mul(const float *__restrict__ w,
                                                           const float *__restrict__ pm, float thr,
                                                           float *__restrict__ out, int64_t n, int vec_ok) {
    // w == nullptr: plain Binarizer (out = bin(pm)); else out = w * bin(pm)
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    const bool has_w = (w != nullptr);
    if (vec_ok) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nthreads) {
            const F4 m = reinterpret_cast<const F4 *>(pm)[i];
            F4 r = {binarize(m.x, thr), binarize(m.y, thr), binarize(m.z, thr), binarize(m.w, thr)};
            if (has_w) {
                const F4 a = reinterpret_cast<const F4 *>(w)[i];
                r = F4{a.x * r.x, a.y * r.y, a.z * r.z, a.w * r.w};
            }
            reinterpret_cast<F4 *>(out)[i] = r;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) out[i] = (has_w ? w[i] : 1.0f) * binarize(pm[i], thr);
    } else {
        for (int64_t i = tid; i < n; i += nthreads) out[i] = (has_w ? w[i] : 1.0f) * binarize(pm[i], thr);
    }
}

// ---------------------------------------------------------------- K4e
// utils/prune.py:203-210 in one pass.  MODE: 0 finetune, 1 prune.
template <bool HAS_PM>
__global__ __launch_bounds__(kThreads) void k_route(float *__restrict__ gw, const float *__restrict__ w,
                                                    const uint8_t *__restrict__ owner, int cur, float wd,
                                                    float *__restrict__ gpm, int mode, int64_t n, int vec_ok) {
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    auto route1 = [&](int64_t i) {
        const int o = owner[i];
        gw[i] = (o == cur) ? fmaf(wd, w[i], gw[i]) : 0.0f;
        if (HAS_PM) {
            if (mode == CPG_MODE_PRUNE || o == 0 || o >= cur) gpm[i] = 0.0f;
        }
    };
    if (vec_ok) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nthreads) {
            const uint32_t o4 = reinterpret_cast<const uint32_t *>(owner)[i];
            const int o0 = o4 & 255, o1 = (o4 >> 8) & 255, o2 = (o4 >> 16) & 255, o3 = o4 >> 24;
            F4 g = {0.f, 0.f, 0.f, 0.f};
            if (o0 == cur || o1 == cur || o2 == cur || o3 == cur) {      // frozen quads: no gw / w read at all
                const F4 gi = reinterpret_cast<const F4 *>(gw)[i];
                const F4 wi = reinterpret_cast<const F4 *>(w)[i];
                g.x = (o0 == cur) ? fmaf(wd, wi.x, gi.x) : 0.0f;
                g.y = (o1 == cur) ? fmaf(wd, wi.y, gi.y) : 0.0f;
                g.z = (o2 == cur) ? fmaf(wd, wi.z, gi.z) : 0.0f;
                g.w = (o3 == cur) ? fmaf(wd, wi.w, gi.w) : 0.0f;
            }
            reinterpret_cast<F4 *>(gw)[i] = g;
            if (HAS_PM) {
                if (mode == CPG_MODE_PRUNE) {
                    reinterpret_cast<F4 *>(gpm)[i] = F4{0.f, 0.f, 0.f, 0.f};
                } else {
                    const bool z0 = (o0 == 0 || o0 >= cur), z1 = (o1 == 0 || o1 >= cur);
                    const bool z2 = (o2 == 0 || o2 >= cur), z3 = (o3 == 0 || o3 >= cur);
                    if (z0 || z1 || z2 || z3) {
                        F4 p = {0.f, 0.f, 0.f, 0.f};
                        if (!(z0 && z1 && z2 && z3)) {
                            p = reinterpret_cast<const F4 *>(gpm)[i];
                            if (z0) p.x = 0.f;
                            if (z1) p.y = 0.f;
                            if (z2) p.z = 0.f;
                            if (z3) p.w = 0.f;
                        }
                        reinterpret_cast<F4 *>(gpm)[i] = p;
                    }
                }
            }
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) route1(i);
    } else {
        for (int64_t i = tid; i < n; i += nthreads) route1(i);
    }
}

// ---------------------------------------------------------------- K7
// Owner-id histogram.  Owner ids take very few distinct values per layer, so plain LDS atomics
// would serialise 64-way on one bin; instead each wave peels off one distinct value per
// iteration with readfirstlane + ballot + popcount and issues ONE LDS atomic for it.
__device__ __forceinline__ void wave_count_byte(unsigned v, unsigned *hist_lds) {
    for (;;) {
        const unsigned u = __builtin_amdgcn_readfirstlane(v);
        const unsigned long long m = __ballot(v == u);
        if (v == u) {
            const int first = __ffsll((long long)m) - 1;
            if ((int)(threadIdx.x & 63) == first) atomicAdd(&hist_lds[u], (unsigned)__popcll(m));
            break;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_mask_hist(const uint8_t *__restrict__ owner,
                                                        const float *__restrict__ pm, int idx, int64_t n,
                                                        unsigned long long *__restrict__ hist, int vec_ok) {
    __shared__ unsigned h[257];
    for (int i = threadIdx.x; i < 257; i += kThreads) h[i] = 0;
    __syncthreads();
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    unsigned shared_cnt = 0;
    const float pick_thr = 0.005f;                 // literal of utils/prune.py:188
    if (vec_ok) {
        const int64_t n4 = n >> 2;
        // all lanes of a wave iterate together (uniform trip count) so ballots see whole waves
        const int64_t iters = (n4 + nthreads - 1) / nthreads;
        for (int64_t it = 0; it < iters; ++it) {
            const int64_t i = tid + it * nthreads;
            if (i < n4) {
                const uint32_t o4 = reinterpret_cast<const uint32_t *>(owner)[i];
                if (pm != nullptr) {
                    const bool a0 = (o4 & 255) > 0 && (int)(o4 & 255) < idx;
                    const bool a1 = ((o4 >> 8) & 255) > 0 && (int)((o4 >> 8) & 255) < idx;
                    const bool a2 = ((o4 >> 16) & 255) > 0 && (int)((o4 >> 16) & 255) < idx;
                    const bool a3 = (o4 >> 24) > 0 && (int)(o4 >> 24) < idx;
                    if (a0 || a1 || a2 || a3) {
                        const F4 p = reinterpret_cast<const F4 *>(pm)[i];
                        shared_cnt += (a0 && p.x > pick_thr) + (a1 && p.y > pick_thr) + (a2 && p.z > pick_thr) +
                                      (a3 && p.w > pick_thr);
                    }
                }
                wave_count_byte(o4 & 255, h);
                wave_count_byte((o4 >> 8) & 255, h);
                wave_count_byte((o4 >> 16) & 255, h);
                wave_count_byte(o4 >> 24, h);
            }
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) {
            const unsigned o = owner[i];
            atomicAdd(&h[o], 1u);
            if (pm != nullptr && o > 0 && (int)o < idx && pm[i] > pick_thr) shared_cnt++;
        }
    } else {
        for (int64_t i = tid; i < n; i += nthreads) {
            const unsigned o = owner[i];
            atomicAdd(&h[o], 1u);
            if (pm != nullptr && o > 0 && (int)o < idx && pm[i] > pick_thr) shared_cnt++;
        }
    }
    if (shared_cnt) atomicAdd(&h[256], shared_cnt);
    __syncthreads();
    for (int i = threadIdx.x; i < 257; i += kThreads)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// ---------------------------------------------------------------- K8
// zero w where owner == 0 (zero_pruned) or additionally owner > idx (apply_mask, idx < 256)
__global__ __launch_bounds__(kThreads) void k_zero_by_owner(float *__restrict__ w, const uint8_t *__restrict__ owner,
                                                            int idx, int64_t n, int vec_ok) {
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    auto dead = [idx](int o) { return o == 0 || o > idx; };
    if (vec_ok) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nthreads) {
            const uint32_t o4 = reinterpret_cast<const uint32_t *>(owner)[i];
            const bool d0 = dead(o4 & 255), d1 = dead((o4 >> 8) & 255), d2 = dead((o4 >> 16) & 255), d3 = dead(o4 >> 24);
            if (d0 && d1 && d2 && d3) {
                reinterpret_cast<F4 *>(w)[i] = F4{0.f, 0.f, 0.f, 0.f};
            } else if (d0 || d1 || d2 || d3) {
                F4 v = reinterpret_cast<const F4 *>(w)[i];
                if (d0) v.x = 0.f;
                if (d1) v.y = 0.f;
                if (d2) v.z = 0.f;
                if (d3) v.w = 0.f;
                reinterpret_cast<F4 *>(w)[i] = v;
            }
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads)
            if (dead(owner[i])) w[i] = 0.0f;
    } else {
        for (int64_t i = tid; i < n; i += nthreads)
            if (dead(owner[i])) w[i] = 0.0f;
    }
}

__global__ __launch_bounds__(kThreads) void k_claim_free(uint8_t *__restrict__ owner, unsigned new_idx, int64_t n,
                                                         int vec_ok) {
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    if (vec_ok) {
        const int64_t n16 = n >> 4;
        for (int64_t i = tid; i < n16; i += nthreads) {
            uint4 v = reinterpret_cast<const uint4 *>(owner)[i];
            auto fix = [new_idx](uint32_t x) {
                // bytes that are zero -> new_idx.  exact zero-byte detect (no borrow artefacts)
                uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
                t = ~(t | x | 0x7F7F7F7Fu);            // 0x80 in every zero byte
                return x | ((t >> 7) * new_idx);
            };
            const uint4 r = {fix(v.x), fix(v.y), fix(v.z), fix(v.w)};
            if (r.x != v.x || r.y != v.y || r.z != v.z || r.w != v.w) reinterpret_cast<uint4 *>(owner)[i] = r;
        }
        for (int64_t i = (n16 << 4) + tid; i < n; i += nthreads)
            if (owner[i] == 0) owner[i] = (uint8_t)new_idx;
    } else {
        for (int64_t i = tid; i < n; i += nthreads)
            if (owner[i] == 0) owner[i] = (uint8_t)new_idx;
    }
}

// ---------------------------------------------------------------- fused masked SGD
// utils/prune.py:203-205 + torch.optim.SGD(momentum, nesterov, dampening 0, weight_decay 0) in ONE pass:
//   g   = owner == cur ? gw + wd * w : 0            (gradient routing; written back so .grad is what the reference leaves)
//   buf = first ? g : momentum * buf + g            (two roundings, as torch's _foreach_mul_ / _foreach_add_)
//   d   = nesterov ? g + momentum * buf : buf
//   w  -= lr * d
// 17 B read + 12 B written per element instead of the 13 + 36 B of the routing pass plus torch's three foreach passes.
__global__ __launch_bounds__(kThreads) void k_sgd_route(float *__restrict__ w, float *__restrict__ gw, float *__restrict__ buf,
                                                        const uint8_t *__restrict__ owner, int cur, float wd, float lr,
                                                        float momentum, int nesterov, int first, int64_t n, int vec_ok) {
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    auto one = [&](float &wv, float &gv, float &bv, int o) {
        const float g = (o == cur) ? fmaf(wd, wv, gv) : 0.0f;
        const float b = first ? g : __fadd_rn(__fmul_rn(momentum, bv), g);
        const float d = nesterov ? fmaf(momentum, b, g) : b;
        wv = fmaf(-lr, d, wv);
        gv = g;
        bv = b;
    };
    if (vec_ok) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nthreads) {
            const uint32_t o4 = reinterpret_cast<const uint32_t *>(owner)[i];
            F4 wv = reinterpret_cast<F4 *>(w)[i], gv = reinterpret_cast<F4 *>(gw)[i];
            F4 bv = first ? F4{0.f, 0.f, 0.f, 0.f} : reinterpret_cast<F4 *>(buf)[i];
            one(wv.x, gv.x, bv.x, o4 & 255);
            one(wv.y, gv.y, bv.y, (o4 >> 8) & 255);
            one(wv.z, gv.z, bv.z, (o4 >> 16) & 255);
            one(wv.w, gv.w, bv.w, o4 >> 24);
            reinterpret_cast<F4 *>(w)[i] = wv;
            reinterpret_cast<F4 *>(gw)[i] = gv;
            reinterpret_cast<F4 *>(buf)[i] = bv;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) {
            float bv = first ? 0.f : buf[i];
            one(w[i], gw[i], bv, owner[i]);
            buf[i] = bv;
        }
    } else {
        for (int64_t i = tid; i < n; i += nthreads) {
            float bv = first ? 0.f : buf[i];
            one(w[i], gw[i], bv, owner[i]);
            buf[i] = bv;
        }
    }
}

// Fused piggymask step (CPG_cifar100_main_normal.py:342-346 Adam on the piggymasks + utils/prune.py:206-210 routing):
//   g   = finetune: (owner == 0 || owner >= cur) ? 0 : gpm ; prune: 0        (only older tasks' slots learn to be picked)
//   m   = m + (1 - beta1) * (g - m);   v = beta2 * v + (1 - beta2) * g * g    (torch.optim.Adam, amsgrad = False, wd = 0)
//   pm -= step_size * m / (sqrt(v) / sqrt(bias_correction2) + eps),  step_size = lr / bias_correction1
// and the routed g is written back to .grad.  21 B read + 16 B written per element in one pass.
__global__ __launch_bounds__(kThreads) void k_adam_route(float *__restrict__ pm, float *__restrict__ gpm, float *__restrict__ m1,
                                                         float *__restrict__ m2, const uint8_t *__restrict__ owner, int cur, int mode,
                                                         float step_size, float omb1, float beta2, float omb2, float eps, float bc2_sqrt,
                                                         int64_t n, int vec_ok) {
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    auto one = [&](float &p, float &g, float &a, float &b, int o) {
        const bool keep = mode == CPG_MODE_FINETUNE && o != 0 && o < cur;
        const float gr = keep ? g : 0.0f;
        a = fmaf(omb1, gr - a, a);                 // exp_avg.lerp_(grad, 1 - beta1)
        b = fmaf(omb2, gr * gr, beta2 * b);        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(b) / bc2_sqrt + eps;
        p = fmaf(-step_size, a / denom, p);
        g = gr;
    };
    if (vec_ok) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nthreads) {
            const uint32_t o4 = reinterpret_cast<const uint32_t *>(owner)[i];
            F4 pv = reinterpret_cast<F4 *>(pm)[i], gv = reinterpret_cast<F4 *>(gpm)[i];
            F4 av = reinterpret_cast<F4 *>(m1)[i], bv = reinterpret_cast<F4 *>(m2)[i];
            one(pv.x, gv.x, av.x, bv.x, o4 & 255);
            one(pv.y, gv.y, av.y, bv.y, (o4 >> 8) & 255);
            one(pv.z, gv.z, av.z, bv.z, (o4 >> 16) & 255);
            one(pv.w, gv.w, av.w, bv.w, o4 >> 24);
            reinterpret_cast<F4 *>(pm)[i] = pv;
            reinterpret_cast<F4 *>(gpm)[i] = gv;
            reinterpret_cast<F4 *>(m1)[i] = av;
            reinterpret_cast<F4 *>(m2)[i] = bv;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nthreads) one(pm[i], gpm[i], m1[i], m2[i], owner[i]);
    } else {
        for (int64_t i = tid; i < n; i += nthreads) one(pm[i], gpm[i], m1[i], m2[i], owner[i]);
    }
}

inline int is16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------- multi-tensor forms of the two fused optimizer passes (ABI 3)
// One launch for up to kMultiMax layers: the per-layer pointers travel BY VALUE in the kernel arguments (no device table, no copy),
// a block owns one kChunk-element piece of one layer and finds it by walking the (block-uniform, scalar) prefix of pieces per layer.
// The arithmetic per element is k_sgd_route's / k_adam_route's own `one`: results do not depend on how the work is cut.
constexpr int kMultiMax = 48;
constexpr int kChunk = kThreads * 16;           // elements per block: 4 float4 per thread
struct SgdItems {
    float *w[kMultiMax], *gw[kMultiMax], *buf[kMultiMax];
    const uint8_t *owner[kMultiMax];
    long long n[kMultiMax];
    int first_block[kMultiMax + 1];             // pieces before layer i; [count] = grid size
    unsigned char vec[kMultiMax];
    int count;
};
struct AdamItems {
    float *pm[kMultiMax], *gpm[kMultiMax], *m1[kMultiMax], *m2[kMultiMax];
    const uint8_t *owner[kMultiMax];
    long long n[kMultiMax];
    int first_block[kMultiMax + 1];
    unsigned char vec[kMultiMax];
    int count;
};
static_assert(sizeof(SgdItems) <= 4096 && sizeof(AdamItems) <= 4096, "kernel arguments are limited to 4 KB");

template <class Items>
__device__ __forceinline__ int find_item(const Items &it, int block) {
    int i = 0;
    while (i + 1 < it.count && it.first_block[i + 1] <= block) ++i;
    return i;
}

__global__ __launch_bounds__(kThreads) void k_sgd_route_multi(const SgdItems it, int cur, float wd, float lr, float momentum, int nesterov,
                                                              int first) {
    const int item = find_item(it, (int)blockIdx.x);
    const int64_t base = (int64_t)((int)blockIdx.x - it.first_block[item]) * kChunk;
    const int64_t n = it.n[item];
    const int64_t end = base + kChunk < n ? base + kChunk : n;
    float *__restrict__ w = it.w[item], *__restrict__ gw = it.gw[item], *__restrict__ buf = it.buf[item];
    const uint8_t *__restrict__ owner = it.owner[item];
    auto one = [&](float &wv, float &gv, float &bv, int o) {
        const float g = (o == cur) ? fmaf(wd, wv, gv) : 0.0f;
        const float b = first ? g : __fadd_rn(__fmul_rn(momentum, bv), g);
        const float d = nesterov ? fmaf(momentum, b, g) : b;
        wv = fmaf(-lr, d, wv);
        gv = g;
        bv = b;
    };
    if (it.vec[item]) {
        const int64_t e4 = end >> 2;                // (base is a multiple of kChunk, hence of 4)
        for (int64_t i = (base >> 2) + threadIdx.x; i < e4; i += kThreads) {
            const uint32_t o4 = reinterpret_cast<const uint32_t *>(owner)[i];
            F4 wv = reinterpret_cast<F4 *>(w)[i], gv = reinterpret_cast<F4 *>(gw)[i];
            F4 bv = first ? F4{0.f, 0.f, 0.f, 0.f} : reinterpret_cast<F4 *>(buf)[i];
            one(wv.x, gv.x, bv.x, o4 & 255);
            one(wv.y, gv.y, bv.y, (o4 >> 8) & 255);
            one(wv.z, gv.z, bv.z, (o4 >> 16) & 255);
            one(wv.w, gv.w, bv.w, o4 >> 24);
            reinterpret_cast<F4 *>(w)[i] = wv;
            reinterpret_cast<F4 *>(gw)[i] = gv;
            reinterpret_cast<F4 *>(buf)[i] = bv;
        }
        for (int64_t i = (e4 << 2) + threadIdx.x; i < end; i += kThreads) {
            float bv = first ? 0.f : buf[i];
            one(w[i], gw[i], bv, owner[i]);
            buf[i] = bv;
        }
    } else {
        for (int64_t i = base + threadIdx.x; i < end; i += kThreads) {
            float bv = first ? 0.f : buf[i];
            one(w[i], gw[i], bv, owner[i]);
            buf[i] = bv;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_adam_route_multi(const AdamItems it, int cur, int mode, float step_size, float omb1, float beta2,
                                                               float omb2, float eps, float bc2_sqrt) {
    const int item = find_item(it, (int)blockIdx.x);
    const int64_t base = (int64_t)((int)blockIdx.x - it.first_block[item]) * kChunk;
    const int64_t n = it.n[item];
    const int64_t end = base + kChunk < n ? base + kChunk : n;
    float *__restrict__ pm = it.pm[item], *__restrict__ gpm = it.gpm[item], *__restrict__ m1 = it.m1[item], *__restrict__ m2 = it.m2[item];
    const uint8_t *__restrict__ owner = it.owner[item];
    auto one = [&](float &p, float &g, float &a, float &b, int o) {
        const bool keep = mode == CPG_MODE_FINETUNE && o != 0 && o < cur;
        const float gr = keep ? g : 0.0f;
        a = fmaf(omb1, gr - a, a);
        b = fmaf(omb2, gr * gr, beta2 * b);
        const float denom = sqrtf(b) / bc2_sqrt + eps;
        p = fmaf(-step_size, a / denom, p);
        g = gr;
    };
    if (it.vec[item]) {
        const int64_t e4 = end >> 2;
        for (int64_t i = (base >> 2) + threadIdx.x; i < e4; i += kThreads) {
            const uint32_t o4 = reinterpret_cast<const uint32_t *>(owner)[i];
            F4 pv = reinterpret_cast<F4 *>(pm)[i], gv = reinterpret_cast<F4 *>(gpm)[i];
            F4 av = reinterpret_cast<F4 *>(m1)[i], bv = reinterpret_cast<F4 *>(m2)[i];
            one(pv.x, gv.x, av.x, bv.x, o4 & 255);
            one(pv.y, gv.y, av.y, bv.y, (o4 >> 8) & 255);
            one(pv.z, gv.z, av.z, bv.z, (o4 >> 16) & 255);
            one(pv.w, gv.w, av.w, bv.w, o4 >> 24);
            reinterpret_cast<F4 *>(pm)[i] = pv;
            reinterpret_cast<F4 *>(gpm)[i] = gv;
            reinterpret_cast<F4 *>(m1)[i] = av;
            reinterpret_cast<F4 *>(m2)[i] = bv;
        }
        for (int64_t i = (e4 << 2) + threadIdx.x; i < end; i += kThreads) one(pm[i], gpm[i], m1[i], m2[i], owner[i]);
    } else {
        for (int64_t i = base + threadIdx.x; i < end; i += kThreads) one(pm[i], gpm[i], m1[i], m2[i], owner[i]);
    }
}

}  // namespace

extern "C" int32_t cpg_multi_tensor_max(void) { return kMultiMax; }

extern "C" int cpg_sgd_route_step_multi(const cpg_sgd_item *items_host, int32_t n_items, int32_t cur, float wd, float lr, float momentum,
                                        int32_t nesterov, int32_t first_step, void *stream) {
    CPG_REQUIRE(n_items >= 0 && (n_items == 0 || items_host), "cpg_sgd_route_step_multi: null item table or negative count");
    CPG_REQUIRE(cur >= 0 && cur <= 255, "cpg_sgd_route_step_multi: owner id %d out of uint8 range", cur);
    for (int32_t at = 0; at < n_items;) {
        SgdItems it;
        int blocks = 0;
        it.count = 0;
        for (; at < n_items && it.count < kMultiMax; ++at) {
            const cpg_sgd_item &s = items_host[at];
            CPG_REQUIRE(s.n >= 0 && (s.n == 0 || (s.w && s.gw && s.momentum_buf && s.owner)), "cpg_sgd_route_step_multi: item %d: null pointer or negative n", at);
            if (s.n == 0) continue;
            const int64_t pieces = (s.n + kChunk - 1) / kChunk;
            CPG_REQUIRE(blocks + pieces < (1ll << 30), "cpg_sgd_route_step_multi: item %d is too large for one launch", at);
            const int i = it.count++;
            it.w[i] = s.w, it.gw[i] = s.gw, it.buf[i] = s.momentum_buf, it.owner[i] = s.owner, it.n[i] = s.n;
            it.vec[i] = is16(s.w) && is16(s.gw) && is16(s.momentum_buf) && (((uintptr_t)s.owner) & 3) == 0;
            it.first_block[i] = blocks;
            blocks += (int)pieces;
        }
        if (it.count == 0) continue;
        it.first_block[it.count] = blocks;
        hipLaunchKernelGGL(k_sgd_route_multi, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, it, cur, wd, lr, momentum, nesterov, first_step);
        CPG_CHECK_LAUNCH("cpg_sgd_route_step_multi");
    }
    return CPG_OK;
}

extern "C" int cpg_adam_route_step_multi(const cpg_adam_item *items_host, int32_t n_items, int32_t cur, int32_t mode, double lr, double beta1,
                                         double beta2, double eps, int32_t step, void *stream) {
    CPG_REQUIRE(n_items >= 0 && (n_items == 0 || items_host), "cpg_adam_route_step_multi: null item table or negative count");
    CPG_REQUIRE(mode == CPG_MODE_FINETUNE || mode == CPG_MODE_PRUNE, "cpg_adam_route_step_multi: unknown mode %d", mode);
    CPG_REQUIRE(cur >= 0 && cur <= 255 && step >= 1, "cpg_adam_route_step_multi: owner id %d / step %d out of range", cur, step);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    for (int32_t at = 0; at < n_items;) {
        AdamItems it;
        int blocks = 0;
        it.count = 0;
        for (; at < n_items && it.count < kMultiMax; ++at) {
            const cpg_adam_item &s = items_host[at];
            CPG_REQUIRE(s.n >= 0 && (s.n == 0 || (s.pm && s.gpm && s.exp_avg && s.exp_avg_sq && s.owner)),
                        "cpg_adam_route_step_multi: item %d: null pointer or negative n", at);
            if (s.n == 0) continue;
            const int64_t pieces = (s.n + kChunk - 1) / kChunk;
            CPG_REQUIRE(blocks + pieces < (1ll << 30), "cpg_adam_route_step_multi: item %d is too large for one launch", at);
            const int i = it.count++;
            it.pm[i] = s.pm, it.gpm[i] = s.gpm, it.m1[i] = s.exp_avg, it.m2[i] = s.exp_avg_sq, it.owner[i] = s.owner, it.n[i] = s.n;
            it.vec[i] = is16(s.pm) && is16(s.gpm) && is16(s.exp_avg) && is16(s.exp_avg_sq) && (((uintptr_t)s.owner) & 3) == 0;
            it.first_block[i] = blocks;
            blocks += (int)pieces;
        }
        if (it.count == 0) continue;
        it.first_block[it.count] = blocks;
        hipLaunchKernelGGL(k_adam_route_multi, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, it, cur, mode, (float)(lr / bc1),
                           (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)sqrt(bc2));
        CPG_CHECK_LAUNCH("cpg_adam_route_step_multi");
    }
    return CPG_OK;
}

extern "C" int cpg_adam_route_step(float *pm, float *gpm, float *exp_avg, float *exp_avg_sq, const uint8_t *owner, int32_t cur,
                                   int32_t mode, double lr, double beta1, double beta2, double eps, int32_t step, int64_t n, void *stream) {
    CPG_REQUIRE(n >= 0 && (n == 0 || (pm && gpm && exp_avg && exp_avg_sq && owner)), "cpg_adam_route_step: null pointer or negative n");
    CPG_REQUIRE(mode == CPG_MODE_FINETUNE || mode == CPG_MODE_PRUNE, "cpg_adam_route_step: unknown mode %d", mode);
    CPG_REQUIRE(cur >= 0 && cur <= 255 && step >= 1, "cpg_adam_route_step: owner id %d / step %d out of range", cur, step);
    if (n == 0) return CPG_OK;
    // every derived constant is formed in double and rounded once, as torch does with its python-float hyper-parameters
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const int vec = is16(pm) && is16(gpm) && is16(exp_avg) && is16(exp_avg_sq) && (((uintptr_t)owner) & 3) == 0;
    hipLaunchKernelGGL(k_adam_route, dim3(stream_grid(n, kThreads * 4)), dim3(kThreads), 0, (hipStream_t)stream, pm, gpm, exp_avg,
                       exp_avg_sq, owner, cur, mode, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)sqrt(bc2), n, vec);
    CPG_CHECK_LAUNCH("cpg_adam_route_step");
    return CPG_OK;
}

extern "C" int cpg_sgd_route_step(float *w, float *gw, float *momentum_buf, const uint8_t *owner, int32_t cur, float wd, float lr,
                                  float momentum, int32_t nesterov, int32_t first_step, int64_t n, void *stream) {
    CPG_REQUIRE(n >= 0 && (n == 0 || (w && gw && momentum_buf && owner)), "cpg_sgd_route_step: null pointer or negative n");
    CPG_REQUIRE(cur >= 0 && cur <= 255, "cpg_sgd_route_step: owner id %d out of uint8 range", cur);
    if (n == 0) return CPG_OK;
    const int vec = is16(w) && is16(gw) && is16(momentum_buf) && (((uintptr_t)owner) & 3) == 0;
    hipLaunchKernelGGL(k_sgd_route, dim3(stream_grid(n, kThreads * 4)), dim3(kThreads), 0, (hipStream_t)stream, w, gw, momentum_buf,
                       owner, cur, wd, lr, momentum, nesterov, first_step, n, vec);
    CPG_CHECK_LAUNCH("cpg_sgd_route_step");
    return CPG_OK;
}

extern "C" int cpg_binarize_mask_weight(const float *w, const float *pm, float thr, float *w_eff, int64_t n,
                                        void *stream) {
    CPG_REQUIRE(n >= 0 && (n == 0 || (pm && w_eff)), "cpg_binarize_mask_weight: null pointer or negative n");
    if (n == 0) return CPG_OK;
    const int vec = (!w || is16(w)) && is16(pm) && is16(w_eff);
    hipLaunchKernelGGL(k_binarize_mul, dim3(stream_grid(n, kThreads * 4)), dim3(kThreads), 0, (hipStream_t)stream, w, pm,
                       thr, w_eff, n, vec);
    CPG_CHECK_LAUNCH("cpg_binarize_mask_weight");
    return CPG_OK;
}

extern "C" int cpg_route_grads(float *gw, const float *w, const uint8_t *owner, int32_t cur, float wd, float *gpm,
                               int32_t mode, int64_t n, void *stream) {
    CPG_REQUIRE(n >= 0 && (n == 0 || (gw && w && owner)), "cpg_route_grads: null pointer or negative n");
    CPG_REQUIRE(mode == CPG_MODE_FINETUNE || mode == CPG_MODE_PRUNE, "cpg_route_grads: unknown mode %d", mode);
    CPG_REQUIRE(cur >= 0 && cur <= 255, "cpg_route_grads: owner id %d out of uint8 range", cur);
    if (n == 0) return CPG_OK;
    const int vec = is16(gw) && is16(w) && (((uintptr_t)owner) & 3) == 0 && (!gpm || is16(gpm));
    const dim3 grid(stream_grid(n, kThreads * 4)), block(kThreads);
    if (gpm)
        hipLaunchKernelGGL(k_route<true>, grid, block, 0, (hipStream_t)stream, gw, w, owner, cur, wd, gpm, mode, n, vec);
    else
        hipLaunchKernelGGL(k_route<false>, grid, block, 0, (hipStream_t)stream, gw, w, owner, cur, wd, gpm, mode, n, vec);
    CPG_CHECK_LAUNCH("cpg_route_grads");
    return CPG_OK;
}

extern "C" int cpg_mask_hist(const uint8_t *owner, const float *pm, int32_t inference_idx, int64_t n, uint64_t *hist,
                             void *stream) {
    CPG_REQUIRE(n >= 0 && hist && (n == 0 || owner), "cpg_mask_hist: null pointer or negative n");
    if (n == 0) return CPG_OK;
    const int vec = (((uintptr_t)owner) & 3) == 0 && (!pm || is16(pm));
    hipLaunchKernelGGL(k_mask_hist, dim3(stream_grid(n, kThreads * 16)), dim3(kThreads), 0, (hipStream_t)stream, owner, pm,
                       inference_idx, n, (unsigned long long *)hist, vec);
    CPG_CHECK_LAUNCH("cpg_mask_hist");
    return CPG_OK;
}

extern "C" int cpg_apply_mask(float *w, const uint8_t *owner, int32_t inference_idx, int64_t n, void *stream) {
    CPG_REQUIRE(n >= 0 && (n == 0 || (w && owner)), "cpg_apply_mask: null pointer or negative n");
    if (n == 0) return CPG_OK;
    const int vec = is16(w) && (((uintptr_t)owner) & 3) == 0;
    hipLaunchKernelGGL(k_zero_by_owner, dim3(stream_grid(n, kThreads * 4)), dim3(kThreads), 0, (hipStream_t)stream, w,
                       owner, inference_idx, n, vec);
    CPG_CHECK_LAUNCH("cpg_apply_mask");
    return CPG_OK;
}

extern "C" int cpg_zero_pruned(float *w, const uint8_t *owner, int64_t n, void *stream) {
    return cpg_apply_mask(w, owner, 255, n, stream);      // no uint8 owner exceeds 255: only owner == 0 dies
}

extern "C" int cpg_claim_free(uint8_t *owner, int32_t new_idx, int64_t n, void *stream) {
    CPG_REQUIRE(n >= 0 && (n == 0 || owner), "cpg_claim_free: null pointer or negative n");
    CPG_REQUIRE(new_idx >= 1 && new_idx <= 255, "cpg_claim_free: owner id %d out of uint8 range", new_idx);
    if (n == 0) return CPG_OK;
    const int vec = is16(owner);
    hipLaunchKernelGGL(k_claim_free, dim3(stream_grid(n, kThreads * 16)), dim3(kThreads), 0, (hipStream_t)stream, owner,
                       (unsigned)new_idx, n, vec);
    CPG_CHECK_LAUNCH("cpg_claim_free");
    return CPG_OK;
}
