// Masked 3x3 / stride 1 / pad 1 convolution: WEIGHT gradient by Winograd F(2x2, 3x3) on fp32 MFMA.
//
//   dg = G^T [ sum over tiles (A dY A^T) .* (B^T d B) ] G        (dY: 2x2 tile of gy, d: the 4x4 input patch of that tile)
//
// the adjoint of conv3x3_wino.hip's forward: per transform position p one GEMM  M_p[k][c] = sum_t P_p[k][t] * V_p[c][t]  over the
// tiles t -- 16 multiplies per tile and channel pair instead of 36 -- and a 4x4 -> 3x3 output transform per (k, c) at the end.
// Same design as k_wg1: ONE WAVE = ONE UNIT (32 output channels x 32 input channels x all 16 positions, 256 accumulators in fixed
// AGPRs, one wave per SIMD), no barriers.  The MFMA's k dimension is a PAIR OF TILES (lanes 0-31: even tile, 32-63: odd tile):
// lane (li, lh) transforms the gy tile of output channel li and the input patch of input channel li for its tile of the pair --
// those 16 + 16 values are its A and B operands.  Channels are lanes here, pixels are what the global loads coalesce over, so
// the raw rows are transposed through (wave-private) LDS:
//   stage = 14 consecutive tiles of one tile row (every VGG16 map is a multiple of 14 tiles wide, so a stage never straddles a row
//           end and the only halo is the one at its two ends): 7 k-steps of 16 MFMAs.
//   G  global -> registers: 16 byte per lane (two tiles' column pairs), lanes = (8 quads x 8 (channel, row) items): 16 loads for x,
//      8 for gy, 4 dword loads for the halo columns; all offsets are a per-lane constant + a scalar stage base.
//   W  registers -> LDS  x_raw[c][row][16 slots][2], gy_raw[k][row][14][2]  (row / channel strides chosen for the LDS banks, see WW_XR)
//   T  per k-step: lane reads its patch (three 8-byte reads per row) and gy tile, 32 + 12 adds -> B and A operands
//   M  16 MFMAs per k-step, one per schedule slot, with one slice of T / W / G behind each (sched_barrier fences).
// Maps 14 pixels wide (VGG16's last block) run the NARROW instance: a stage is tile row ty of TWO images, 7 + 7 tiles, the two
// halves of the wave take one image each; everything after the staging loads is the same code with other address constants.
// ONE LDS buffer per wave: a wave's LDS operations complete in order, so the last k-step of a stage first stores the next
// stage's rows (loaded five k-steps earlier) and then reads the next stage's first operands.
// Split over tile ranges; the partial sums land in part[split][tap][k][c], which k_split_reduce (igemm_core.h) adds up and passes
// through the autograd epilogue (gW = g * bin(pm), gPM = g * W) exactly as for the direct kernels.
// The sign of the transform rows / columns with a -1 (A's last row) is applied to M in the epilogue instead of to the operands.
#include <algorithm>
#include "igemm_core.h"

using namespace cpg;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// LDS layout of one stage (floats).  x_raw: 32 channels x 4 rows of 32 floats, row stride 34, channel stride 138; gy_raw: 32 channels
// x 2 rows of 28, row stride 30, channel stride 62.  The strides are what the LDS banking wants (MI355X guide, LDS table):
// ds_write_b64 goes out in groups of 16 lanes over 32 banks, and 16 lanes are two rows x 8 quads here, so the second row must
// sit 2 banks off the first (34 = 2, 30 = -2 mod 32) for the two to interleave; the b64 reads of a half-wave have the channel as
// the lane index, and 138 / 62 step through all the even banks.  With 32 / 130 / 28 / 58 the stores were 2-way conflicted and
// the kernel 3% slower (profiles/r02j_pmc_wino.md: 44% of the LDS cycles were conflict cycles).
constexpr int WW_XR = 34, WW_XC = 138, WW_GR = 30, WW_GC = 62;
constexpr int WW_XS = 32 * WW_XC;
constexpr int WW_STAGE = WW_XS + 32 * WW_GC;        // 6400 floats = 25 KB per wave

struct WwGeom {
    int N, C, K, H, W;
    int th, tw, nseg;         // tile rows / tiles per row / 14-tile segments per row
    unsigned nstages;         // N * th * nseg
    int nkb, ncb, nsplit;
    unsigned su;              // stages per unit
    int span;                 // images a unit can touch
    int xcd;                  // blocks in XCD-contiguous order (block b runs on XCD b % 8: each XCD walks one eighth of the units)
};

@@MMA@@
@@RD@@
#define WW_FENCE() __builtin_amdgcn_sched_barrier(0)

// -DWG_TIMING (development builds, tools/attic/diag_wg_timing.py --wgrad): per-wave constant-clock stamps, as in conv3x3_wino.hip
#ifdef WG_TIMING
__device__ unsigned long long ww_dbg[65536 * 8];
#define WW_STAMP(i) do { if (lane == 0 && u < 65536) ww_dbg[u * 8 + (i)] = wall_clock64(); } while (0)
#else
#define WW_STAMP(i)
#endif
// (blocks of one or two waves and an XCD-contiguous unit order were measured: within 0.5 % on every layer, a single wave without the
//  reorder 6 % slower on the 64-channel layers)
// GS > 0 (round 4; wide maps only): the waves of a block that work on the SAME input-channel block cb -- GS = 4: all four (output-channel
// blocks 4 j .. 4 j + 3), GS = 2: two pairs -- stage that block's x rows ONCE, each wave loading and storing 1 / GS of them, into a shared,
// double-buffered LDS region (one workgroup barrier per stage: the next stage's rows are stored into the OTHER buffer in the last k-step of
// a stage, the barrier sits between those stores and the first operand reads of the next stage; every wave has finished reading a buffer
// before it arrives at the barrier after which that buffer is written again).  gy rows stay wave-private.  Per stage and wave: 13 (GS = 4)
// or 18 (GS = 2) vector-memory instructions and as many LDS store groups instead of 28 -- and the x rows are requested from L2 once per
// GS units instead of once per unit.
template <bool NARROW, int GS = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_wgw(WwGeom g, const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ part) {
    static_assert(GS == 0 || GS == 4 || (!NARROW && GS == 2), "shared staging: groups of 4 waves, or 2 on wide maps");
    constexpr int NG = GS ? 4 / GS : 0;                          // x-sharing groups per block
    // TS (GS == 4): the four waves of a group also SHARE THE INPUT TRANSFORM.  Their B operands are the same 16 values per lane (the
    // transformed patch of input channel li for the lane's tile): wave w computes transform row w only -- two patch rows, 4 + 4 adds
    // instead of four rows and 32 --, leaves its four values in a shared, double-buffered LDS array (one 16-byte store) and reads
    // all sixteen back after a barrier (four 16-byte reads): 20 instead of 44 vector-ALU instructions per k-step (on this chip
    // each one is fp32-MFMA issue time), the same sums bit for bit.
    constexpr bool TS = GS == 4;
    constexpr int WW_BS = 2 * 16 * 64;                           // floats of the shared-B array of one group: [buffer][position quad][lane][4]
    __shared__ __attribute__((aligned(16))) float smem_all[GS ? NG * 2 * WW_XS + 4 * 32 * WW_GC + (TS ? WW_BS : 0) : 4 * WW_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int gw = GS ? wave % GS : 0;                           // this wave's index within its x-sharing group
    float *smem = GS ? smem_all : smem_all + wave * WW_STAGE;
    // x rows of a stage: buffer 0 / 1 (the same private buffer without sharing); gy rows: wave-private in both designs
    float *xb0 = GS ? smem_all + (wave / (GS ? GS : 1)) * 2 * WW_XS : smem;
    float *xb1 = GS ? xb0 + WW_XS : smem;
    float *gyp = GS ? smem_all + NG * 2 * WW_XS + wave * 32 * WW_GC : smem + WW_XS;
    float *bsh = smem_all + NG * 2 * WW_XS + 4 * 32 * WW_GC;      // (TS only)
    const int HW = g.H * g.W;
    const unsigned npairs = (unsigned)(g.nkb * g.ncb);
    const unsigned u = (g.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * 4 + wave;
    if (u >= npairs * (unsigned)g.nsplit) return;                // (no barriers anywhere: a wave may leave)
    const unsigned pair = u % npairs, split = u / npairs;
    const int kb = (int)(pair % (unsigned)g.nkb), cb = (int)(pair / (unsigned)g.nkb);
    // first channel of the unit's blocks.  Channel counts that are not multiples of 32 (the reference's grown networks: 78 / 156 / 313 / 627,
    // models/vgg.py:124-154 with sqrt(1.5)): the LAST block starts at K - 32 / C - 32 and overlaps its neighbour -- every lane works on a
    // channel that exists, nothing in the main loop is predicated, and the overlapped (k, c) entries are computed twice from the same
    // operands in the same order: both units store the same bits.
    // (readfirstlane: without it the compiler carried the two minima in vector registers and every staging load of the wide instances got
    //  a waterfall loop around its scalar offset -- 3-5 % of the kernel)
    const int k0 = __builtin_amdgcn_readfirstlane(min(kb * 32, g.K - 32)), c0 = __builtin_amdgcn_readfirstlane(min(cb * 32, g.C - 32));
    const unsigned s_begin = split * g.su;
    const int nst = (int)min(g.su, g.nstages - s_begin);
    WW_STAMP(0);
#ifdef WG_TIMING
    if (lane == 0 && u < 65536) ww_dbg[u * 8 + 6] = __builtin_amdgcn_s_getreg(63492), ww_dbg[u * 8 + 7] = __builtin_amdgcn_s_getreg(63508);
#endif
    // loader coordinates (uniform): the stage whose loads are issued next -> image n, tile row ty, segment tseg
    // (NARROW: image PAIR n, n + 1 and tile row ty -- see below)
    const unsigned per_img = (unsigned)(g.th * g.nseg);
    int n = (int)(s_begin / per_img) * (NARROW ? 2 : 1);
    const unsigned r0 = s_begin % per_img;
    int ty = (int)(r0 / (unsigned)g.nseg), tseg = (int)(r0 % (unsigned)g.nseg);
    const int n0 = n;
    const int nimg_here = min(g.span, g.N - n0);
    // x descriptor: base one row and four pixels (16 bytes: the base stays 16-byte aligned for the dwordx4 loads) BELOW the first image, so that the row / column "- 1" of the patch never makes a
    // per-lane offset negative (only the per-lane part of an offset is range-checked; out-of-image elements get 0x80000000)
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(x + (int64_t)n0 * g.C * HW - (g.W + 4)), 0, nimg_here * g.C * HW * 4 + (g.W + 4) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_g =
        __builtin_amdgcn_make_buffer_rsrc((void *)(gy + (int64_t)n0 * g.K * HW), 0, nimg_here * g.K * HW * 4, 0x00020000);

    // ---- per-lane constants of G and W ----
    constexpr int kOOR = (int)0x80000000;
    // quad (tiles 2 qd, 2 qd + 1) and (channel, row) item of this lane.  Seven quads per 14-tile row: the eighth lane of a group
    // repeats the seventh (same address, same data -- a dump address would need the per-instruction offset removed again)
    const int qd = min(lane & 7, 6), rem = lane >> 3;
    // NARROW (maps 14 pixels = 7 tiles wide): the stage is tile row ty of TWO images, 7 + 7 tiles (half h = lh of the k-step's tile
    // pair); an LDS row holds both halves side by side, x: [0 0 | 14 px of image n | 0 0 | 14 px of image n + 1 | 0 0] (the zeros
    // are the left / right padding of the image: written once, there are no halo loads), gy: [14 | 14].  A row is 7 column PAIRS:
    // 8-byte loads, lanes = (8 pairs x 8 items), item = (row, half) for x -> one channel per load (32 loads), (channel, row, half)
    // for gy (16 loads).  The half is the low bit of the item so that the 16 lanes of a ds_write_b64 group fill 28 different banks.
    const int xrow = NARROW ? rem >> 1 : rem & 3, half = rem & 1;
    // (shared staging: wave gw of a group takes the x items j GS + gw and the halo items j GS + gw -- its share of the channel offset is
    //  part of the per-lane constants, so that the item index of a load / store stays a compile-time constant)
    const int vx_const = NARROW ? (half * g.C * HW + xrow * g.W + gw * HW) * 4 + qd * 8 + 16
                                : ((rem >> 2) * HW + xrow * g.W + gw * 2 * HW) * 4 + qd * 16 + 16;   // x item (c = 2 j + rem / 4, row = rem % 4)
    const int vg_const = NARROW ? ((rem >> 2) * HW + half * g.K * HW + ((rem >> 1) & 1) * g.W) * 4 + qd * 8
                                : ((rem >> 1) * HW + (rem & 1) * g.W) * 4 + qd * 16;   // gy item (k = 4 j + rem / 2, row = rem % 2)
    const int hside = lane & 1, hrow = (lane >> 1) & 3;                             // halo item (c = 8 j + lane / 8, row, side)
    const int vh_const = ((lane >> 3) * HW + hrow * g.W + gw * 8 * HW) * 4 + (hside ? 28 * 4 + 16 : 12);
    const int xw_addr = NARROW ? xrow * WW_XR + 16 * half + 2 + 2 * qd + gw * WW_XC : (rem >> 2) * WW_XC + xrow * WW_XR + 2 + 4 * qd + gw * 2 * WW_XC;
    const int gw_addr = NARROW ? (rem >> 2) * WW_GC + ((rem >> 1) & 1) * WW_GR + 14 * half + 2 * qd          // (relative to gyp)
                               : (rem >> 1) * WW_GC + (rem & 1) * WW_GR + 4 * qd;
    const int hw_addr = (lane >> 3) * WW_XC + hrow * WW_XR + (hside ? 30 : 1) + gw * 8 * WW_XC;
    if constexpr (NARROW) {                                      // the padding columns 0, 1, 16, 17, 32, 33 of every x row
        for (int i = lane; i < 32 * 4 * 6; i += 64) {                // (shared staging: of both buffers; every wave of the group writes the same zeros)
            const int rowi = i / 6, col = i % 6;
            xb0[(rowi >> 2) * WW_XC + (rowi & 3) * WW_XR + (col >> 1) * 16 + (col & 1)] = 0.0f;
            if (GS) xb1[(rowi >> 2) * WW_XC + (rowi & 3) * WW_XR + (col >> 1) * 16 + (col & 1)] = 0.0f;
        }
    }
    int sx, sgo, vx, vg, vh;                                     // stage part of the offsets (scalar) / per-lane part with validity
    auto stage_offsets = [&]() {
        const bool top = ty == 0, bot = ty == g.th - 1;
        if constexpr (NARROW) {
            sx = (((n - n0) * g.C + c0) * HW + 2 * ty * g.W) * 4;
            sgo = (((n - n0) * g.K + k0) * HW + 2 * ty * g.W) * 4;
            const bool gone = half && n + 1 >= g.N;              // an odd batch: the last stage has one image only
            vx = ((xrow == 0 && top) || (xrow == 3 && bot) || gone) ? kOOR : vx_const;
            vg = gone ? kOOR : vg_const;
            vh = kOOR;
        } else {
            sx = (((n - n0) * g.C + c0) * HW + 2 * ty * g.W + 28 * tseg) * 4;
            sgo = (((n - n0) * g.K + k0) * HW + 2 * ty * g.W + 28 * tseg) * 4;
            vx = ((xrow == 0 && top) || (xrow == 3 && bot)) ? kOOR : vx_const;
            vg = vg_const;
            vh = ((hrow == 0 && top) || (hrow == 3 && bot) || (hside == 0 && tseg == 0) || (hside == 1 && tseg == g.nseg - 1)) ? kOOR : vh_const;
        }
    };
    auto advance_stage = [&]() {
        if (++tseg == g.nseg) {
            tseg = 0;
            if (++ty == g.th) ty = 0, n += NARROW ? 2 : 1;
        }
        stage_offsets();
    };
    // staging registers: 16 + 8 quads (+ 4 halo words), or NARROW 32 + 16 pairs: 96 registers either way
    i32x4 rx[16], rg[8];
    float rh[4];
    auto g_load = [&](int idx, int) {
        if constexpr (NARROW) {
            i32x2 v;
            if (idx < 32)
                v = __builtin_amdgcn_raw_buffer_load_b64(srd_x, vx, sx + idx * HW * 4, 0);
            else
                v = __builtin_amdgcn_raw_buffer_load_b64(srd_g, vg, sgo + (idx - 32) * 2 * HW * 4, 0);
            if (idx < 32)
                rx[idx >> 1][2 * (idx & 1)] = v[0], rx[idx >> 1][2 * (idx & 1) + 1] = v[1];
            else
                rg[(idx - 32) >> 1][2 * (idx & 1)] = v[0], rg[(idx - 32) >> 1][2 * (idx & 1) + 1] = v[1];
        } else {
            if (idx < 16)
                rx[idx] = __builtin_amdgcn_raw_buffer_load_b128(srd_x, vx, sx + idx * 2 * HW * 4, 0);
            else if (idx < 24)
                rg[idx - 16] = __builtin_amdgcn_raw_buffer_load_b128(srd_g, vg, sgo + (idx - 16) * 4 * HW * 4, 0);
            else
                rh[idx - 24] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, vh, sx + (idx - 24) * 8 * HW * 4, 0));
        }
    };
    auto w_store = [&](int idx) {
        if constexpr (NARROW) {
            if (idx < 32) {
                i32x2 v;
                v[0] = rx[idx >> 1][2 * (idx & 1)], v[1] = rx[idx >> 1][2 * (idx & 1) + 1];
                *reinterpret_cast<i32x2 *>(smem + xw_addr + idx * WW_XC) = v;
            } else if (idx < 48) {
                i32x2 v;
                v[0] = rg[(idx - 32) >> 1][2 * (idx & 1)], v[1] = rg[(idx - 32) >> 1][2 * (idx & 1) + 1];
                *reinterpret_cast<i32x2 *>(gyp + gw_addr + (idx - 32) * 2 * WW_GC) = v;
            }
        } else if (idx < 16) {
            i32x2 *d = reinterpret_cast<i32x2 *>(smem + xw_addr + idx * 2 * WW_XC);
            i32x2 lo, hi;
            lo[0] = rx[idx][0], lo[1] = rx[idx][1], hi[0] = rx[idx][2], hi[1] = rx[idx][3];
            d[0] = lo, d[1] = hi;
        } else if (idx < 24) {
            i32x2 *d = reinterpret_cast<i32x2 *>(gyp + gw_addr + (idx - 16) * 4 * WW_GC);
            i32x2 lo, hi;
            lo[0] = rg[idx - 16][0], lo[1] = rg[idx - 16][1], hi[0] = rg[idx - 16][2], hi[1] = rg[idx - 16][3];
            d[0] = lo, d[1] = hi;
        } else if (idx < 28) {
            smem[hw_addr + (idx - 24) * 8 * WW_XC] = rh[idx - 24];
        }
    };
    // shared staging (GS > 0): item i of this wave = x item j GS + gw (i < XL), gy item i - XL (own 8), halo item j GS + gw
    // (NARROW: 32 / GS eight-byte x items -- one channel each -- and the wave's own 16 gy items, no halo)
    constexpr int XL = GS ? (NARROW ? 32 : 16) / GS : 0, GL = NARROW ? 16 : 8, HL = (GS && !NARROW) ? 4 / GS : 0, NSI = XL + GL + HL;
    auto g_load_s = [&](int i) {
        if constexpr (NARROW) {
            i32x2 v;
            if (i < XL)
                v = __builtin_amdgcn_raw_buffer_load_b64(srd_x, vx, sx + i * GS * HW * 4, 0);
            else
                v = __builtin_amdgcn_raw_buffer_load_b64(srd_g, vg, sgo + (i - XL) * 2 * HW * 4, 0);
            if (i < XL)
                rx[i >> 1][2 * (i & 1)] = v[0], rx[i >> 1][2 * (i & 1) + 1] = v[1];
            else
                rg[(i - XL) >> 1][2 * ((i - XL) & 1)] = v[0], rg[(i - XL) >> 1][2 * ((i - XL) & 1) + 1] = v[1];
        } else {
            if (i < XL)
                rx[i] = __builtin_amdgcn_raw_buffer_load_b128(srd_x, vx, sx + i * GS * 2 * HW * 4, 0);
            else if (i < XL + 8)
                rg[i - XL] = __builtin_amdgcn_raw_buffer_load_b128(srd_g, vg, sgo + (i - XL) * 4 * HW * 4, 0);
            else if (i < NSI)
                rh[i - XL - 8] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, vh, sx + (i - XL - 8) * GS * 8 * HW * 4, 0));
        }
    };
    auto w_store_s = [&](int i, float *xb) {
        if constexpr (NARROW) {
            i32x2 v;
            if (i < XL) {
                v[0] = rx[i >> 1][2 * (i & 1)], v[1] = rx[i >> 1][2 * (i & 1) + 1];
                *reinterpret_cast<i32x2 *>(xb + xw_addr + i * GS * WW_XC) = v;
            } else if (i < NSI) {
                v[0] = rg[(i - XL) >> 1][2 * ((i - XL) & 1)], v[1] = rg[(i - XL) >> 1][2 * ((i - XL) & 1) + 1];
                *reinterpret_cast<i32x2 *>(gyp + gw_addr + (i - XL) * 2 * WW_GC) = v;
            }
        } else if (i < XL) {
            i32x2 *d = reinterpret_cast<i32x2 *>(xb + xw_addr + i * GS * 2 * WW_XC);
            i32x2 lo, hi;
            lo[0] = rx[i][0], lo[1] = rx[i][1], hi[0] = rx[i][2], hi[1] = rx[i][3];
            d[0] = lo, d[1] = hi;
        } else if (i < XL + 8) {
            i32x2 *d = reinterpret_cast<i32x2 *>(gyp + gw_addr + (i - XL) * 4 * WW_GC);
            i32x2 lo, hi;
            lo[0] = rg[i - XL][0], lo[1] = rg[i - XL][1], hi[0] = rg[i - XL][2], hi[1] = rg[i - XL][3];
            d[0] = lo, d[1] = hi;
        } else if (i < NSI) {
            xb[hw_addr + (i - XL - 8) * GS * 8 * WW_XC] = rh[i - XL - 8];
        }
    };
    // ---- T: operands of k-step ks (tile 2 ks + lh of the stage in LDS), in 14 micro steps ----
    // tile of (k-step ks, half-wave lh): wide 2 ks + lh of the 14-tile segment, NARROW tile ks of image n + lh
    constexpr int kLhX = NARROW ? 16 : 2, kLhG = NARROW ? 14 : 2, kKs = NARROW ? 2 : 4;
    const int xr_base = li * WW_XC + 2 + lh * kLhX, gr_base = li * WW_GC + lh * kLhG;
    auto t_micro = [&](int m, int ks, float (&A)[16], float (&B)[16], const float *xb) {
        if (m < 4) {                                             // patch row m: own pair + the neighbours' halves
            const float *r = xb + xr_base + kKs * ks + m * WW_XR;
            const f32x2 own = *reinterpret_cast<const f32x2 *>(r);
            // the neighbours' halves as 8-byte reads too: a 4-byte read of 32 lanes at an even channel stride is 2-way conflicted
            const f32x2 lo = *reinterpret_cast<const f32x2 *>(r - 2), hi = *reinterpret_cast<const f32x2 *>(r + 2);
            B[m * 4 + 0] = lo[1], B[m * 4 + 1] = own[0], B[m * 4 + 2] = own[1], B[m * 4 + 3] = hi[0];
        } else if (m == 4) {                                     // the gy tile: A[0] = y00, A[3] = y01, A[12] = y10, A[15] = y11
            const f32x2 y0 = *reinterpret_cast<const f32x2 *>(gyp + gr_base + kKs * ks);
            const f32x2 y1 = *reinterpret_cast<const f32x2 *>(gyp + gr_base + kKs * ks + WW_GR);
            A[0] = y0[0], A[3] = y0[1], A[12] = y1[0], A[15] = y1[1];
        } else if (m < 7) {                                      // V = B^T d B: column pass of columns 2 (m - 5), + 1
#pragma unroll
            for (int j = 2 * (m - 5); j < 2 * (m - 5) + 2; ++j) {
                const float d0 = B[0 * 4 + j], d1 = B[1 * 4 + j], d2 = B[2 * 4 + j], d3 = B[3 * 4 + j];
                B[0 * 4 + j] = d0 - d2, B[1 * 4 + j] = d1 + d2, B[2 * 4 + j] = d2 - d1, B[3 * 4 + j] = d1 - d3;
                asm volatile("" : "+v"(B[0 * 4 + j]), "+v"(B[1 * 4 + j]), "+v"(B[2 * 4 + j]), "+v"(B[3 * 4 + j]));
            }
        } else if (m < 11) {                                     // ... row pass of row m - 7
            const int i = m - 7;
            const float t0 = B[i * 4 + 0], t1 = B[i * 4 + 1], t2 = B[i * 4 + 2], t3 = B[i * 4 + 3];
            B[i * 4 + 0] = t0 - t2, B[i * 4 + 1] = t1 + t2, B[i * 4 + 2] = t2 - t1, B[i * 4 + 3] = t1 - t3;
            asm volatile("" : "+v"(B[i * 4 + 0]), "+v"(B[i * 4 + 1]), "+v"(B[i * 4 + 2]), "+v"(B[i * 4 + 3]));
        } else if (m == 11) {                                    // P' = A dY A^T without the signs: rows 1, 2 of A dY
            A[4] = A[0] + A[12], A[7] = A[3] + A[15], A[8] = A[0] - A[12], A[11] = A[3] - A[15];
            asm volatile("" : "+v"(A[4]), "+v"(A[7]), "+v"(A[8]), "+v"(A[11]));
        } else if (m == 12) {                                    // ... columns 1, 2 of rows 0, 1
            A[1] = A[0] + A[3], A[2] = A[0] - A[3], A[5] = A[4] + A[7], A[6] = A[4] - A[7];
            asm volatile("" : "+v"(A[1]), "+v"(A[2]), "+v"(A[5]), "+v"(A[6]));
        } else if (m == 13) {                                    // ... of rows 2, 3
            A[9] = A[8] + A[11], A[10] = A[8] - A[11], A[13] = A[12] + A[15], A[14] = A[12] - A[15];
            asm volatile("" : "+v"(A[9]), "+v"(A[10]), "+v"(A[13]), "+v"(A[14]));
        }
    };

    // ---- TS: this wave's share of the input transform, in 7 micro steps (0, 1: read the two patch rows the wave's transform row needs;
    // 2: column + row pass of that row, store; [barrier]; 3-6: read back the sixteen values) ----
    // transform row w of B^T d: w = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3  ->  d[ra] + sg * d[rb]
    const int b_ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1), b_rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
    const float b_sg = wave == 1 ? 1.0f : -1.0f;
    float ta[4], tb[4];
    auto b_micro = [&](int m, int ks, int buf, float (&B)[16], const float *xb) {
        if (m < 2) {
            const float *r = xb + xr_base + kKs * ks + (m == 0 ? b_ra : b_rb) * WW_XR;
            const f32x2 own = *reinterpret_cast<const f32x2 *>(r);
            const f32x2 lo = *reinterpret_cast<const f32x2 *>(r - 2), hi = *reinterpret_cast<const f32x2 *>(r + 2);
            float (&t)[4] = m == 0 ? ta : tb;
            t[0] = lo[1], t[1] = own[0], t[2] = own[1], t[3] = hi[0];
        } else if (m == 2) {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = __builtin_fmaf(b_sg, tb[j], ta[j]);       // (sg = +-1: exactly the add / subtract)
            f32x4 v;
            v[0] = e[0] - e[2], v[1] = e[1] + e[2], v[2] = e[2] - e[1], v[3] = e[1] - e[3];
            *reinterpret_cast<f32x4 *>(bsh + ((buf * 4 + wave) * 64 + lane) * 4) = v;
        } else {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(bsh + ((buf * 4 + (m - 3)) * 64 + lane) * 4);
            B[(m - 3) * 4 + 0] = v[0], B[(m - 3) * 4 + 1] = v[1], B[(m - 3) * 4 + 2] = v[2], B[(m - 3) * 4 + 3] = v[3];
        }
    };
    // workgroup barrier that waits for this wave's LDS traffic only (__syncthreads would also wait for the stage loads in flight)
#define WW_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    WW_STAMP(1);
@@ZERO@@
    float A0[16], B0[16], A1[16], B1[16];
    // prologue: stage 0 into LDS, its first operands, stage 1's coordinates ready
    stage_offsets();
    if constexpr (GS > 0) {
#pragma unroll
        for (int i = 0; i < NSI; ++i) g_load_s(i);
#pragma unroll
        for (int i = 0; i < NSI; ++i) w_store_s(i, xb0);
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < (NARROW ? 48 : 28); ++i) g_load(i, 0);
#pragma unroll
        for (int i = 0; i < (NARROW ? 48 : 28); ++i) w_store(i);
    }
    advance_stage();
    if constexpr (TS) {
#pragma unroll
        for (int m = 0; m < 3; ++m) b_micro(m, 0, 0, B0, xb0);
        WW_LDS_BARRIER();
#pragma unroll
        for (int m = 3; m < 7; ++m) b_micro(m, 0, 0, B0, xb0);
        t_micro(4, 0, A0, B0, xb0), t_micro(11, 0, A0, B0, xb0), t_micro(12, 0, A0, B0, xb0), t_micro(13, 0, A0, B0, xb0);
    } else {
#pragma unroll
        for (int m = 0; m < 14; ++m) t_micro(m, 0, A0, B0, xb0);
    }

    WW_STAMP(2);
    for (int st = 0; st < nst; st += 2) {
        if constexpr (NARROW && GS == 4) {
@@BODY_NS4@@
        } else if constexpr (NARROW) {
@@BODY_N@@
        } else if constexpr (GS == 4) {
@@BODY_S4@@
        } else if constexpr (GS == 2) {
@@BODY_S2@@
        } else {
@@BODY@@
        }
    }

    WW_STAMP(3);
    // ---- epilogue: dg = G^T M G per (k, c); M[i][j] = sigma_i sigma_j acc[4 i + j], sigma = (1, 1, 1, -1) ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float *pout = part + ((int64_t)split * 9 * g.K + k0) * g.C + c0 + li;
    const int64_t tap_plane = (int64_t)g.K * g.C;
    auto out_e = [&](int e, float (&m)[16]) {
        m[3] = -m[3], m[7] = -m[7], m[11] = -m[11], m[12] = -m[12], m[13] = -m[13], m[14] = -m[14];
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = 0.5f * (m[4 + j] + m[8 + j]), d = 0.5f * (m[4 + j] - m[8 + j]);
            t[0][j] = m[j] + s, t[1][j] = d, t[2][j] = s + m[12 + j];
        }
        float *dst = pout + (int64_t)((e & 3) + 8 * (e >> 2) + 4 * lh) * g.C;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float s = 0.5f * (t[r][1] + t[r][2]), d = 0.5f * (t[r][1] - t[r][2]);
            dst[(r * 3 + 0) * tap_plane] = t[r][0] + s;
            dst[(r * 3 + 1) * tap_plane] = d;
            dst[(r * 3 + 2) * tap_plane] = s + t[r][3];
        }
    };
    {
        float m[16];
@@OUT@@
    }
    WW_STAMP(4);
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

struct WwPlan {
    WwGeom g;
    size_t ws_bytes;
    int64_t blocks;
    bool narrow;              // maps 14 pixels wide: a stage is one tile row of two images
    int gs;                   // waves of a block that share the staging of their x rows (0: none; k_wgw<false, gs>)
};

bool ww_plan(const cpg_conv_desc *d, WwPlan &p) {
    if (opt_on(OPT_NO_WINO) || opt_on(OPT_NO_WINO_WGRAD)) return false;
    if (d->R != 3 || d->S != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1 || d->dil_h != 1 ||
        d->dil_w != 1 || d->groups != 1)
        return false;
    p.narrow = d->W == 14;
    if (d->H % 2 || (d->W % 28 && !p.narrow) || d->C < 32 || d->K < 32 || d->N < 1) return false;
    WwGeom &g = p.g;
    g.N = d->N, g.C = d->C, g.K = d->K, g.H = d->H, g.W = d->W;
    g.th = d->H / 2, g.tw = d->W / 2, g.nseg = p.narrow ? 1 : g.tw / 14;
    const int64_t nstages = (int64_t)(p.narrow ? (d->N + 1) / 2 : d->N) * g.th * g.nseg;
    if (nstages >= (1ll << 28)) return false;
    g.nstages = (unsigned)nstages;
    g.nkb = (d->K + 31) / 32, g.ncb = (d->C + 31) / 32;      // (any channel count >= 32: the last block overlaps, see k0 / c0 in k_wgw)
    const int64_t npairs = (int64_t)g.nkb * g.ncb;
    // units per wave slot.  Every unit costs a prologue / epilogue and a 9 x 32 x 32 partial sum (36.9 KB) that it writes and
    // k_split_reduce reads: 4 units per slot are 4096 units = 151 MB of partials per launch whatever the layer's size -- the PMC
    // passes over the bench (profiles/r03_traffic_bench.json) showed SphereNet-20's weight-gradient launches moving 4 x their
    // algorithmic bytes, 22 % of them in the reduce.  One unit per slot (the units of a launch are equally long: nothing to
    // balance on an idle chip): SphereNet-20 20.96 -> 19.92 ms per step, ResNet-50 73.83 -> 72.69, VGG16 121.10 -> 120.44.
    // When another stream's kernels (RCCL) hold some CUs a one-round launch grows by a whole round (x 2); with two rounds of
    // half-length units it grows by one of them (x 1.5) at 0.6 % of the idle-chip time (4 rounds: x 1.25 at 1.5-3 %, 9 % on the
    // 14-pixel maps).  cpg_amd.dist raises the hint (cpg_set_shared_chip_hint) when a rank's gradients keep RCCL busy for a
    // noticeable share of the backward (VGG16's 537 MB); until round 4 the hint meant 4 rounds for every multi-GPU rank, which
    // beside RCCL's real kernels cost more than it can save (profiles/r04_ab_shared_chip_plans.txt).  CPG_WW_UNITS overrides.
    // Channel-block pairs that do not divide the wave slots (the grown networks: 627 -> 627 has 400 pairs, 2 splits = 800 units on 1024 slots,
    // 78 % of one round): up to two more rounds are tried and the plan with the best slot use wins, an extra round priced at 1.5 % (the
    // figures above) -- 400 pairs: 5 splits = 2000 units = 97.7 % of two rounds.  The width-1.0 layers keep their one-round plans.
    int upw = shared_chip_hint() ? 2 : 1;
    const int64_t slots = 4 * kCUs;
    if (const int forced = opt(OPT_WW_UNITS); forced != OPT_UNSET) {
        upw = std::max(1, forced);
    } else {
        double best = -1.0;
        int best_r = upw;
        for (int r = upw; r <= upw + 2; ++r) {
            const int64_t w = std::min<int64_t>(std::max<int64_t>(1, r * slots / npairs), nstages);
            const int64_t units = npairs * w, rounds = (units + slots - 1) / slots;
            const double eff = (double)units / (double)(rounds * slots) - 0.015 * (double)(rounds - upw);
            if (eff > best + 1e-9) best = eff, best_r = r;
        }
        upw = best_r;
    }
    int64_t want = std::max<int64_t>(1, ((int64_t)upw * slots) / npairs);
    want = std::min<int64_t>(want, nstages);
    g.su = (unsigned)((nstages + want - 1) / want);
    g.nsplit = (int)((nstages + g.su - 1) / g.su);
    const int64_t per_img = (int64_t)g.th * g.nseg;
    g.span = ((int)((g.su + per_img - 1) / per_img) + 1) * (p.narrow ? 2 : 1);
    const int64_t HW = (int64_t)d->H * d->W;
    if ((int64_t)g.span * std::max(d->C, d->K) * HW * 4 + (d->W + 4) * 4 >= (1ll << 31)) return false;
    p.ws_bytes = (size_t)g.nsplit * 9 * d->K * d->C * sizeof(float);
    p.blocks = (npairs * g.nsplit + 3) / 4;
    g.xcd = opt_or(OPT_WW_XCD, 1) != 0;
    // shared staging of the x rows (wide maps): the four units of a block are consecutive (split, cb, kb) with kb fastest -- all four
    // share cb when there are >= 4 output-channel blocks (a multiple of 4), pairs of them when there are 2 (mod 4); every block must be
    // whole (no wave may leave early: the variant has a barrier per stage).  CPG_WW_SHARE = 0 / 2 / 4 overrides.
    p.gs = 0;
    if (npairs % 4 == 0) p.gs = g.nkb % 4 == 0 ? 4 : ((g.nkb % 2 == 0 && !p.narrow) ? 2 : 0);
    if (const int f = opt(OPT_WW_SHARE); f != OPT_UNSET && (f == 0 || (p.gs != 0 && (f == 2 || (f == 4 && p.gs == 4))))) p.gs = f;
    return p.blocks <= 0x7FFFFFFFll;
}

}  // namespace

#ifdef WG_TIMING
extern "C" int cpg_debug_ww_timing(unsigned long long *dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ww_dbg), (size_t)n * 8 * sizeof(unsigned long long));
}
#endif
extern "C" int cpg_conv3x3_wino_wgrad_ok(const cpg_conv_desc *d) {
    WwPlan p;
    return ww_plan(d, p) ? 1 : 0;
}

extern "C" size_t cpg_conv3x3_wino_wgrad_workspace(const cpg_conv_desc *d) {
    WwPlan p;
    return ww_plan(d, p) ? p.ws_bytes : 0;
}

extern "C" int cpg_conv3x3_wino_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w, const float *pm, float thr,
                                      float *gw, float *gpm, void *ws, size_t ws_bytes, hipStream_t stream) {
    WwPlan p;
    if (!ww_plan(d, p)) return fail(CPG_E_UNSUPPORTED, "cpg_conv2d_wgrad(winograd): shape not supported");
    if (ws == nullptr || ws_bytes < p.ws_bytes)
        return fail(CPG_E_WORKSPACE, "cpg_conv2d_wgrad(winograd): workspace %zu < %zu bytes", ws_bytes, p.ws_bytes);
    if (p.narrow && p.gs == 4)
        hipLaunchKernelGGL((k_wgw<true, 4>), dim3((unsigned)p.blocks), dim3(256), 0, stream, p.g, x, gy, (float *)ws);
    else if (p.narrow)
        hipLaunchKernelGGL(k_wgw<true>, dim3((unsigned)p.blocks), dim3(256), 0, stream, p.g, x, gy, (float *)ws);
    else if (p.gs == 4)
        hipLaunchKernelGGL((k_wgw<false, 4>), dim3((unsigned)p.blocks), dim3(256), 0, stream, p.g, x, gy, (float *)ws);
    else if (p.gs == 2)
        hipLaunchKernelGGL((k_wgw<false, 2>), dim3((unsigned)p.blocks), dim3(256), 0, stream, p.g, x, gy, (float *)ws);
    else
        hipLaunchKernelGGL(k_wgw<false>, dim3((unsigned)p.blocks), dim3(256), 0, stream, p.g, x, gy, (float *)ws);
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    const int64_t out_elems = (int64_t)d->K * d->C * 9;
    launch_split_reduce((const float *)ws, p.g.nsplit, out_elems, (int64_t)d->K * d->C, ep, stream);
    CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(winograd)");
    return CPG_OK;
}
