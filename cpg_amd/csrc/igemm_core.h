// Shared main loop of the fp32 MFMA implicit-GEMM kernels (conv fwd / dgrad / wgrad, linear).
//
//   D[m][j] += sum_k A[k][m] * B[k][j]          (m: BM rows, j: BN cols, k: BK per step)
//
// Both operands are staged in LDS k-major ([BK][BM+1] and [BK][BN+1] floats) so that the
// v_mfma_f32_32x32x2_f32 operand fetch -- lane l reads A[k = l>>5][m = l&31] and
// B[k = l>>5][j = l&31] (one VGPR each) -- is a conflict-free ds_read_b32 of 32 consecutive
// floats per half-wave.  The effective weight W * bin(piggymask) is formed by the operand
// loaders while they stage the tile (never materialised in HBM).
//
// Pipeline: two LDS stages; the global loads of tile t+1 are issued into registers before the
// MFMAs of tile t and written to the other stage after them => one __syncthreads per K tile
// and HBM/L2 latency hidden under 2*BK... MFMA issue cycles.  fp32-in MFMA is 64 FLOP/clk/SIMD
// (1/16 of bf16), so a 2x2 register block (4 ds_read_b32 per 4 MFMAs = 256 cycles) leaves the
// LDS pipe ~95 % idle; the kernel is MFMA-issue bound by construction.
#pragma once
#include <algorithm>
#include "cpg_common.h"

namespace cpg {

template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
    static constexpr int THREADS = 256;
    static_assert(WM * WN == 4, "4 waves per block");
    static constexpr int FM = BM / 32 / WM;       // 32x32 fragments per wave along m
    static constexpr int FN = BN / 32 / WN;
    static_assert(FM >= 1 && FN >= 1 && BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "tile/wave mismatch");
    static constexpr int LDA = BM + 1, LDB = BN + 1;
    static constexpr int A_ELEMS = BK * LDA, B_ELEMS = BK * LDB;
    static constexpr int STAGE_ELEMS = A_ELEMS + B_ELEMS;
    static constexpr int SMEM_FLOATS = 2 * STAGE_ELEMS;
    static constexpr int NA = BM * BK / THREADS;  // staged elements per thread
    static constexpr int NB = BN * BK / THREADS;
    static_assert((BM * BK) % THREADS == 0 && (BN * BK) % THREADS == 0, "staging must divide evenly");
};

// XCD-aware, bijective remap of the linear block id: the dispatcher places block b on XCD b % 8;
// hand each XCD a contiguous run of logical tiles so neighbours (same pixel tile, next channel
// tile; adjacent pixel tiles) share that XCD's 4 MiB L2.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned q = nblocks / kXCDs, r = nblocks % kXCDs;
    const unsigned xcd = bid % kXCDs, idx = bid / kXCDs;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Sum over the 32 lanes of each half-wave by DPP (quad swaps, half-row / row mirrors, then lane 15 of rows 0 / 2 into rows 1 / 3):
// five v_add_f32_dpp per value, the result valid in lanes 16-31 and 48-63.  The ds_bpermute butterfly this replaces cost the
// statistics epilogue of the Winograd kernel k_wg3 4.7 us per unit (tools/attic/diag_wg_timing.py --stats): 320 LDS-crossbar round trips.  Eight values per
// asm block, step by step across the eight: a DPP operand must not be read within two instructions of the VALU write that produced
// it, and neither the assembler nor the compiler looks into inline asm for that.
#define CPG_DPP8(OP)                                                                                                                \
    "v_add_f32_dpp %0, %0, %0 " OP "\n\tv_add_f32_dpp %1, %1, %1 " OP "\n\tv_add_f32_dpp %2, %2, %2 " OP "\n\tv_add_f32_dpp %3, %3, %3 " OP "\n\t" \
    "v_add_f32_dpp %4, %4, %4 " OP "\n\tv_add_f32_dpp %5, %5, %5 " OP "\n\tv_add_f32_dpp %6, %6, %6 " OP "\n\tv_add_f32_dpp %7, %7, %7 " OP "\n\t"
__device__ __forceinline__ void half_wave_sum8(float *v) {
    asm volatile("s_nop 1\n\t"                                         // (the values may have been written just before)
                 CPG_DPP8("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 CPG_DPP8("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 CPG_DPP8("row_half_mirror row_mask:0xf bank_mask:0xf")
                 CPG_DPP8("row_mirror row_mask:0xf bank_mask:0xf")      // every lane of a 16-lane row holds the row's sum
                 CPG_DPP8("row_bcast:15 row_mask:0xa bank_mask:0xf")    // rows 1 / 3 += lane 15 of rows 0 / 2
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
constexpr int kHalfSumLane = 16;       // li of a lane that holds wg_half_sum's result

template <class Cfg>
__device__ __forceinline__ void mma_stage(const float *__restrict__ As, const float *__restrict__ Bs,
                                          f32x16 (&acc)[Cfg::FM][Cfg::FN], int a_off, int b_off) {
    // a_off = (lane>>5)*LDA + wave_m*FM*32 + (lane&31) ; b_off likewise
#pragma unroll
    for (int s = 0; s < Cfg::BK / 2; ++s) {
        float a[Cfg::FM], b[Cfg::FN];
#pragma unroll
        for (int fm = 0; fm < Cfg::FM; ++fm) a[fm] = As[a_off + 2 * s * Cfg::LDA + fm * 32];
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn) b[fn] = Bs[b_off + 2 * s * Cfg::LDB + fn * 32];
#pragma unroll
        for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < Cfg::FN; ++fn)
                acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fm], b[fn], acc[fm][fn], 0, 0, 0);
    }
}

// ALoader / BLoader concept:
//   void fetch(int kt, float (&r)[N])            issue the global loads of K tile kt into registers
//   void put(const float (&r)[N], float *lds)    write them to an LDS stage
template <class Cfg, class ALoader, class BLoader>
__device__ __forceinline__ void igemm_mainloop(ALoader &la, BLoader &lb, int kt_begin, int kt_end, float *smem,
                                               f32x16 (&acc)[Cfg::FM][Cfg::FN]) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int a_off = (lane >> 5) * Cfg::LDA + wm * Cfg::FM * 32 + (lane & 31);
    const int b_off = (lane >> 5) * Cfg::LDB + wn * Cfg::FN * 32 + (lane & 31);
#pragma unroll
    for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[fm][fn][e] = 0.0f;
    if (kt_begin >= kt_end) return;

    float ra[Cfg::NA], rb[Cfg::NB];
    la.fetch(kt_begin, ra);
    lb.fetch(kt_begin, rb);
    la.put(ra, smem);
    lb.put(rb, smem + Cfg::A_ELEMS);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        float *stage = smem + cur * Cfg::STAGE_ELEMS;
        float *next = smem + (cur ^ 1) * Cfg::STAGE_ELEMS;
        const bool more = (kt + 1 < kt_end);
        if (more) {
            la.fetch(kt + 1, ra);
            lb.fetch(kt + 1, rb);
        }
        mma_stage<Cfg>(stage, stage + Cfg::A_ELEMS, acc, a_off, b_off);
        if (more) {
            la.put(ra, next);
            lb.put(rb, next + Cfg::A_ELEMS);
        }
        __syncthreads();
    }
}

// Walk the accumulator fragments: f(m_local, j_local, fn, value).  C/D layout of 32x32 MFMA:
// col j = lane & 31, row m = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <class Cfg, class F>
__device__ __forceinline__ void for_each_acc(const f32x16 (&acc)[Cfg::FM][Cfg::FN], F &&f) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
    for (int fm = 0; fm < Cfg::FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < Cfg::FN; ++fn)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (wm * Cfg::FM + fm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int j = (wn * Cfg::FN + fn) * 32 + (lane & 31);
                f(m, j, fn, acc[fm][fn][e]);
            }
}

// per-fragment-column setup: out[fn] = f(j_local of this lane in fragment column fn)
template <class Cfg, class F>
__device__ __forceinline__ void col_setup(int64_t (&out)[Cfg::FN], F &&f) {
    const int lane = threadIdx.x & 63;
    const int wn = (threadIdx.x >> 6) % Cfg::WN;
#pragma unroll
    for (int fn = 0; fn < Cfg::FN; ++fn) out[fn] = f((wn * Cfg::FN + fn) * 32 + (lane & 31));
}

// ------------------------------------------------------------------------------------------
// Dense operand loaders (linear layers; weights of conv fwd)
// ------------------------------------------------------------------------------------------
// "KC": element (k, row) at base[row * ld + k]  -- K contiguous in memory (lanes run along k)
// "RC": element (k, row) at base[k * ld + row]  -- row contiguous in memory (lanes run along row)
// ROWS = BM or BN, LD_LDS = ROWS + 1.  Optional piggymask with the same indexing.
template <int ROWS, int BK, bool KC, bool HAS_PM>
struct DenseLoader {
    static constexpr int N = ROWS * BK / 256;
    static constexpr int LDS_LD = ROWS + 1;
    const float *base;
    const float *pm;
    float thr;
    int64_t ld;
    int row0, rows_total, k_total;
    // per-thread mapping
    int t_k, t_row;          // KC: k = t_k, row = t_row + (256/BK)*i ; RC: row = t_row, k = t_k + (256/ROWS)*i

    __device__ __forceinline__ void init(const float *b, const float *p, float th, int64_t ld_, int row0_, int rows_total_,
                                         int k_total_) {
        base = b; pm = p; thr = th; ld = ld_; row0 = row0_; rows_total = rows_total_; k_total = k_total_;
        if (KC) {
            t_k = threadIdx.x % BK;
            t_row = threadIdx.x / BK;
        } else {
            t_row = threadIdx.x % ROWS;
            t_k = threadIdx.x / ROWS;
        }
    }
    // Loads are unconditional (clamped address) and nothing touches the loaded registers before put():
    // a predicated load becomes an exec-masked branch with its own s_waitcnt vmcnt(0) (one serialised
    // round trip per element), and arithmetic on a loaded value inside fetch() would drag the wait in
    // front of the MFMAs of the current tile.  Validity is carried as a bitmask, piggymask values raw.
    unsigned okmask;
    float rp[HAS_PM ? N : 1];
    __device__ __forceinline__ void fetch(int kt, float (&r)[N]) {
        static_assert(N <= 32, "validity bitmask is 32 bits");
        okmask = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int k = kt * BK + (KC ? t_k : t_k + (256 / ROWS) * i);
            const int row = row0 + (KC ? t_row + (256 / BK) * i : t_row);
            const bool ok = k < k_total && row < rows_total;
            int64_t off = KC ? (int64_t)row * ld + k : (int64_t)k * ld + row;
            off = ok ? off : 0;
            okmask |= (ok ? 1u : 0u) << i;
            r[i] = base[off];
            if (HAS_PM) rp[i] = pm[off];
        }
    }
    __device__ __forceinline__ void put(const float (&r)[N], float *lds) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int kk = KC ? t_k : t_k + (256 / ROWS) * i;
            const int rr = KC ? t_row + (256 / BK) * i : t_row;
            float v = r[i];
            if (HAS_PM) v *= binarize(rp[i], thr);
            lds[kk * LDS_LD + rr] = ((okmask >> i) & 1u) ? v : 0.0f;
        }
    }
};

// ------------------------------------------------------------------------------------------
// Output description shared by the direct epilogue and the split-K reduction.
// ------------------------------------------------------------------------------------------
enum BiasMode { BIAS_NONE = 0, BIAS_OUTER = 1, BIAS_INNER = 2 };
// flat output index e:   BIAS_OUTER: channel = (e / inner) % outer   (conv: inner = OH*OW, outer = K)
//                        BIAS_INNER: channel = e % inner             (linear: inner = out_features)
struct Epilogue {
    float *out;            // y / gx / gw
    const float *bias;     // or nullptr
    int bias_mode;
    int64_t inner, outer;
    const float *pm;       // wgrad: piggymask (nullptr -> plain copy)
    const float *w;        // wgrad: weight, for gpm
    float *gpm;            // wgrad: piggymask grad out (nullptr if pm == nullptr)
    float thr;
};

__device__ __forceinline__ void epilogue_store(const Epilogue &ep, int64_t e, float v) {
    if (ep.bias_mode == BIAS_OUTER) v += ep.bias[(e / ep.inner) % ep.outer];
    else if (ep.bias_mode == BIAS_INNER) v += ep.bias[e % ep.inner];
    if (ep.pm != nullptr) {
        // autograd of `bin(pm) * W` (models/layers.py:103): gW = g * bin(pm), gPM = g * W (straight-through)
        ep.gpm[e] = v * ep.w[e];
        v *= binarize(ep.pm[e], ep.thr);
    }
    ep.out[e] = v;
}

// ------------------------------------------------------------------------------------------
// Split-K reduction shared by every weight-gradient / split GEMM path.
// ------------------------------------------------------------------------------------------
namespace {
// Sums the split partials in a fixed order (deterministic for a given shape) and applies the autograd epilogue.
// tap_plane = 0: the partials are laid out like the output; tap_plane = M * C: they are tap-major,
// part[split][tap][co][ci] (what conv3x3.hip's k_c3_wgrad writes with 128-byte coalesced stores), and position p is scattered into
// [co][ci][tap].  `ks` (a power of two <= 64) threads share one output, each summing every ks-th split, combined
// through LDS: layers with few outputs have the most splits (64 -> 64 channels: 36 864 outputs x 1024 splits), and one
// thread per output left that sum latency bound (764 us for the stem, 320 us for features.3).
__global__ __launch_bounds__(256) void k_split_reduce(const float *__restrict__ part, int nsplit, int64_t out_elems,
                                                         int64_t tap_plane, int ks, Epilogue ep) {
    __shared__ float red[256];
    const int tid = threadIdx.x, P = 256 / ks, pl = tid % P, sl = tid / P;
    for (int64_t base = (int64_t)blockIdx.x * P; base < out_elems; base += (int64_t)gridDim.x * P) {
        const int64_t p = base + pl;
        float s = 0.0f;
        if (p < out_elems)
            for (int k = sl; k < nsplit; k += ks) s += part[(int64_t)k * out_elems + p];
        if (ks > 1) {
            red[tid] = s;
            __syncthreads();
            if (sl == 0)
                for (int j = 1; j < ks; ++j) s += red[j * P + pl];
            __syncthreads();
        }
        if (sl == 0 && p < out_elems) {
            const int64_t e = tap_plane ? (p % tap_plane) * 9 + p / tap_plane : p;
            epilogue_store(ep, e, s);
        }
    }
}

inline void launch_split_reduce(const float *part, int nsplit, int64_t out_elems, int64_t tap_plane, const Epilogue &ep,
                                hipStream_t stream) {
    int ks = 1;
    while (ks < 64 && ks * 2 <= nsplit && out_elems * ks < 262144) ks *= 2;
    const int64_t groups = (out_elems + 256 / ks - 1) / (256 / ks);
    hipLaunchKernelGGL(k_split_reduce, dim3((unsigned)std::min<int64_t>(groups, 16384)), dim3(256), 0, stream, part, nsplit,
                       out_elems, tap_plane, ks, ep);
}
}  // namespace

}  // namespace cpg
