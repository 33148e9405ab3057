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
//   W  registers -> LDS  x_raw[c][row][16 slots][2], gy_raw[k][row][14][2]  (channel strides 130 / 58 words: conflict-free reads)
//   T  per k-step: lane reads its patch (ds_read_b64 + ds_read2_b32 per row) and gy tile, 32 + 12 adds -> B and A operands
//   M  16 MFMAs per k-step, one per schedule slot, with one slice of T / W / G behind each (sched_barrier fences).
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

constexpr int WW_XC = 130, WW_GC = 58;              // channel strides (floats) of the raw rows in LDS
constexpr int WW_XS = 32 * WW_XC;                   // 4160
constexpr int WW_STAGE = WW_XS + 32 * WW_GC;        // 6016 floats = 23.5 KB

struct WwGeom {
    int N, C, K, H, W;
    int th, tw, nseg;         // tile rows / tiles per row / 14-tile segments per row
    unsigned nstages;         // N * th * nseg
    int nkb, ncb, nsplit;
    unsigned su;              // stages per unit
    int span;                 // images a unit can touch
};

#define WW_ONE_0(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" : : "v"(A), "v"(B) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15")
#define WW_ONE_1(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[16:31], %0, %1, a[16:31]" : : "v"(A), "v"(B) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31")
#define WW_ONE_2(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[32:47], %0, %1, a[32:47]" : : "v"(A), "v"(B) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47")
#define WW_ONE_3(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[48:63], %0, %1, a[48:63]" : : "v"(A), "v"(B) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63")
#define WW_ONE_4(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[64:79], %0, %1, a[64:79]" : : "v"(A), "v"(B) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79")
#define WW_ONE_5(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[80:95], %0, %1, a[80:95]" : : "v"(A), "v"(B) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95")
#define WW_ONE_6(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[96:111], %0, %1, a[96:111]" : : "v"(A), "v"(B) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111")
#define WW_ONE_7(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[112:127], %0, %1, a[112:127]" : : "v"(A), "v"(B) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127")
#define WW_ONE_8(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[128:143], %0, %1, a[128:143]" : : "v"(A), "v"(B) : "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143")
#define WW_ONE_9(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[144:159], %0, %1, a[144:159]" : : "v"(A), "v"(B) : "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159")
#define WW_ONE_10(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[160:175], %0, %1, a[160:175]" : : "v"(A), "v"(B) : "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175")
#define WW_ONE_11(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[176:191], %0, %1, a[176:191]" : : "v"(A), "v"(B) : "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191")
#define WW_ONE_12(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[192:207], %0, %1, a[192:207]" : : "v"(A), "v"(B) : "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207")
#define WW_ONE_13(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[208:223], %0, %1, a[208:223]" : : "v"(A), "v"(B) : "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223")
#define WW_ONE_14(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[224:239], %0, %1, a[224:239]" : : "v"(A), "v"(B) : "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239")
#define WW_ONE_15(A, B) asm volatile("v_mfma_f32_32x32x2_f32 a[240:255], %0, %1, a[240:255]" : : "v"(A), "v"(B) : "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255")

#define WW_RD_0(m) asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a16\n\tv_accvgpr_read_b32 %2, a32\n\tv_accvgpr_read_b32 %3, a48\n\tv_accvgpr_read_b32 %4, a64\n\tv_accvgpr_read_b32 %5, a80\n\tv_accvgpr_read_b32 %6, a96\n\tv_accvgpr_read_b32 %7, a112\n\tv_accvgpr_read_b32 %8, a128\n\tv_accvgpr_read_b32 %9, a144\n\tv_accvgpr_read_b32 %10, a160\n\tv_accvgpr_read_b32 %11, a176\n\tv_accvgpr_read_b32 %12, a192\n\tv_accvgpr_read_b32 %13, a208\n\tv_accvgpr_read_b32 %14, a224\n\tv_accvgpr_read_b32 %15, a240" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_1(m) asm volatile("v_accvgpr_read_b32 %0, a1\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a33\n\tv_accvgpr_read_b32 %3, a49\n\tv_accvgpr_read_b32 %4, a65\n\tv_accvgpr_read_b32 %5, a81\n\tv_accvgpr_read_b32 %6, a97\n\tv_accvgpr_read_b32 %7, a113\n\tv_accvgpr_read_b32 %8, a129\n\tv_accvgpr_read_b32 %9, a145\n\tv_accvgpr_read_b32 %10, a161\n\tv_accvgpr_read_b32 %11, a177\n\tv_accvgpr_read_b32 %12, a193\n\tv_accvgpr_read_b32 %13, a209\n\tv_accvgpr_read_b32 %14, a225\n\tv_accvgpr_read_b32 %15, a241" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_2(m) asm volatile("v_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a18\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a50\n\tv_accvgpr_read_b32 %4, a66\n\tv_accvgpr_read_b32 %5, a82\n\tv_accvgpr_read_b32 %6, a98\n\tv_accvgpr_read_b32 %7, a114\n\tv_accvgpr_read_b32 %8, a130\n\tv_accvgpr_read_b32 %9, a146\n\tv_accvgpr_read_b32 %10, a162\n\tv_accvgpr_read_b32 %11, a178\n\tv_accvgpr_read_b32 %12, a194\n\tv_accvgpr_read_b32 %13, a210\n\tv_accvgpr_read_b32 %14, a226\n\tv_accvgpr_read_b32 %15, a242" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_3(m) asm volatile("v_accvgpr_read_b32 %0, a3\n\tv_accvgpr_read_b32 %1, a19\n\tv_accvgpr_read_b32 %2, a35\n\tv_accvgpr_read_b32 %3, a51\n\tv_accvgpr_read_b32 %4, a67\n\tv_accvgpr_read_b32 %5, a83\n\tv_accvgpr_read_b32 %6, a99\n\tv_accvgpr_read_b32 %7, a115\n\tv_accvgpr_read_b32 %8, a131\n\tv_accvgpr_read_b32 %9, a147\n\tv_accvgpr_read_b32 %10, a163\n\tv_accvgpr_read_b32 %11, a179\n\tv_accvgpr_read_b32 %12, a195\n\tv_accvgpr_read_b32 %13, a211\n\tv_accvgpr_read_b32 %14, a227\n\tv_accvgpr_read_b32 %15, a243" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_4(m) asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a20\n\tv_accvgpr_read_b32 %2, a36\n\tv_accvgpr_read_b32 %3, a52\n\tv_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a84\n\tv_accvgpr_read_b32 %6, a100\n\tv_accvgpr_read_b32 %7, a116\n\tv_accvgpr_read_b32 %8, a132\n\tv_accvgpr_read_b32 %9, a148\n\tv_accvgpr_read_b32 %10, a164\n\tv_accvgpr_read_b32 %11, a180\n\tv_accvgpr_read_b32 %12, a196\n\tv_accvgpr_read_b32 %13, a212\n\tv_accvgpr_read_b32 %14, a228\n\tv_accvgpr_read_b32 %15, a244" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_5(m) asm volatile("v_accvgpr_read_b32 %0, a5\n\tv_accvgpr_read_b32 %1, a21\n\tv_accvgpr_read_b32 %2, a37\n\tv_accvgpr_read_b32 %3, a53\n\tv_accvgpr_read_b32 %4, a69\n\tv_accvgpr_read_b32 %5, a85\n\tv_accvgpr_read_b32 %6, a101\n\tv_accvgpr_read_b32 %7, a117\n\tv_accvgpr_read_b32 %8, a133\n\tv_accvgpr_read_b32 %9, a149\n\tv_accvgpr_read_b32 %10, a165\n\tv_accvgpr_read_b32 %11, a181\n\tv_accvgpr_read_b32 %12, a197\n\tv_accvgpr_read_b32 %13, a213\n\tv_accvgpr_read_b32 %14, a229\n\tv_accvgpr_read_b32 %15, a245" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_6(m) asm volatile("v_accvgpr_read_b32 %0, a6\n\tv_accvgpr_read_b32 %1, a22\n\tv_accvgpr_read_b32 %2, a38\n\tv_accvgpr_read_b32 %3, a54\n\tv_accvgpr_read_b32 %4, a70\n\tv_accvgpr_read_b32 %5, a86\n\tv_accvgpr_read_b32 %6, a102\n\tv_accvgpr_read_b32 %7, a118\n\tv_accvgpr_read_b32 %8, a134\n\tv_accvgpr_read_b32 %9, a150\n\tv_accvgpr_read_b32 %10, a166\n\tv_accvgpr_read_b32 %11, a182\n\tv_accvgpr_read_b32 %12, a198\n\tv_accvgpr_read_b32 %13, a214\n\tv_accvgpr_read_b32 %14, a230\n\tv_accvgpr_read_b32 %15, a246" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_7(m) asm volatile("v_accvgpr_read_b32 %0, a7\n\tv_accvgpr_read_b32 %1, a23\n\tv_accvgpr_read_b32 %2, a39\n\tv_accvgpr_read_b32 %3, a55\n\tv_accvgpr_read_b32 %4, a71\n\tv_accvgpr_read_b32 %5, a87\n\tv_accvgpr_read_b32 %6, a103\n\tv_accvgpr_read_b32 %7, a119\n\tv_accvgpr_read_b32 %8, a135\n\tv_accvgpr_read_b32 %9, a151\n\tv_accvgpr_read_b32 %10, a167\n\tv_accvgpr_read_b32 %11, a183\n\tv_accvgpr_read_b32 %12, a199\n\tv_accvgpr_read_b32 %13, a215\n\tv_accvgpr_read_b32 %14, a231\n\tv_accvgpr_read_b32 %15, a247" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_8(m) asm volatile("v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a24\n\tv_accvgpr_read_b32 %2, a40\n\tv_accvgpr_read_b32 %3, a56\n\tv_accvgpr_read_b32 %4, a72\n\tv_accvgpr_read_b32 %5, a88\n\tv_accvgpr_read_b32 %6, a104\n\tv_accvgpr_read_b32 %7, a120\n\tv_accvgpr_read_b32 %8, a136\n\tv_accvgpr_read_b32 %9, a152\n\tv_accvgpr_read_b32 %10, a168\n\tv_accvgpr_read_b32 %11, a184\n\tv_accvgpr_read_b32 %12, a200\n\tv_accvgpr_read_b32 %13, a216\n\tv_accvgpr_read_b32 %14, a232\n\tv_accvgpr_read_b32 %15, a248" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_9(m) asm volatile("v_accvgpr_read_b32 %0, a9\n\tv_accvgpr_read_b32 %1, a25\n\tv_accvgpr_read_b32 %2, a41\n\tv_accvgpr_read_b32 %3, a57\n\tv_accvgpr_read_b32 %4, a73\n\tv_accvgpr_read_b32 %5, a89\n\tv_accvgpr_read_b32 %6, a105\n\tv_accvgpr_read_b32 %7, a121\n\tv_accvgpr_read_b32 %8, a137\n\tv_accvgpr_read_b32 %9, a153\n\tv_accvgpr_read_b32 %10, a169\n\tv_accvgpr_read_b32 %11, a185\n\tv_accvgpr_read_b32 %12, a201\n\tv_accvgpr_read_b32 %13, a217\n\tv_accvgpr_read_b32 %14, a233\n\tv_accvgpr_read_b32 %15, a249" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_10(m) asm volatile("v_accvgpr_read_b32 %0, a10\n\tv_accvgpr_read_b32 %1, a26\n\tv_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a58\n\tv_accvgpr_read_b32 %4, a74\n\tv_accvgpr_read_b32 %5, a90\n\tv_accvgpr_read_b32 %6, a106\n\tv_accvgpr_read_b32 %7, a122\n\tv_accvgpr_read_b32 %8, a138\n\tv_accvgpr_read_b32 %9, a154\n\tv_accvgpr_read_b32 %10, a170\n\tv_accvgpr_read_b32 %11, a186\n\tv_accvgpr_read_b32 %12, a202\n\tv_accvgpr_read_b32 %13, a218\n\tv_accvgpr_read_b32 %14, a234\n\tv_accvgpr_read_b32 %15, a250" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_11(m) asm volatile("v_accvgpr_read_b32 %0, a11\n\tv_accvgpr_read_b32 %1, a27\n\tv_accvgpr_read_b32 %2, a43\n\tv_accvgpr_read_b32 %3, a59\n\tv_accvgpr_read_b32 %4, a75\n\tv_accvgpr_read_b32 %5, a91\n\tv_accvgpr_read_b32 %6, a107\n\tv_accvgpr_read_b32 %7, a123\n\tv_accvgpr_read_b32 %8, a139\n\tv_accvgpr_read_b32 %9, a155\n\tv_accvgpr_read_b32 %10, a171\n\tv_accvgpr_read_b32 %11, a187\n\tv_accvgpr_read_b32 %12, a203\n\tv_accvgpr_read_b32 %13, a219\n\tv_accvgpr_read_b32 %14, a235\n\tv_accvgpr_read_b32 %15, a251" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_12(m) asm volatile("v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a28\n\tv_accvgpr_read_b32 %2, a44\n\tv_accvgpr_read_b32 %3, a60\n\tv_accvgpr_read_b32 %4, a76\n\tv_accvgpr_read_b32 %5, a92\n\tv_accvgpr_read_b32 %6, a108\n\tv_accvgpr_read_b32 %7, a124\n\tv_accvgpr_read_b32 %8, a140\n\tv_accvgpr_read_b32 %9, a156\n\tv_accvgpr_read_b32 %10, a172\n\tv_accvgpr_read_b32 %11, a188\n\tv_accvgpr_read_b32 %12, a204\n\tv_accvgpr_read_b32 %13, a220\n\tv_accvgpr_read_b32 %14, a236\n\tv_accvgpr_read_b32 %15, a252" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_13(m) asm volatile("v_accvgpr_read_b32 %0, a13\n\tv_accvgpr_read_b32 %1, a29\n\tv_accvgpr_read_b32 %2, a45\n\tv_accvgpr_read_b32 %3, a61\n\tv_accvgpr_read_b32 %4, a77\n\tv_accvgpr_read_b32 %5, a93\n\tv_accvgpr_read_b32 %6, a109\n\tv_accvgpr_read_b32 %7, a125\n\tv_accvgpr_read_b32 %8, a141\n\tv_accvgpr_read_b32 %9, a157\n\tv_accvgpr_read_b32 %10, a173\n\tv_accvgpr_read_b32 %11, a189\n\tv_accvgpr_read_b32 %12, a205\n\tv_accvgpr_read_b32 %13, a221\n\tv_accvgpr_read_b32 %14, a237\n\tv_accvgpr_read_b32 %15, a253" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_14(m) asm volatile("v_accvgpr_read_b32 %0, a14\n\tv_accvgpr_read_b32 %1, a30\n\tv_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a62\n\tv_accvgpr_read_b32 %4, a78\n\tv_accvgpr_read_b32 %5, a94\n\tv_accvgpr_read_b32 %6, a110\n\tv_accvgpr_read_b32 %7, a126\n\tv_accvgpr_read_b32 %8, a142\n\tv_accvgpr_read_b32 %9, a158\n\tv_accvgpr_read_b32 %10, a174\n\tv_accvgpr_read_b32 %11, a190\n\tv_accvgpr_read_b32 %12, a206\n\tv_accvgpr_read_b32 %13, a222\n\tv_accvgpr_read_b32 %14, a238\n\tv_accvgpr_read_b32 %15, a254" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))
#define WW_RD_15(m) asm volatile("v_accvgpr_read_b32 %0, a15\n\tv_accvgpr_read_b32 %1, a31\n\tv_accvgpr_read_b32 %2, a47\n\tv_accvgpr_read_b32 %3, a63\n\tv_accvgpr_read_b32 %4, a79\n\tv_accvgpr_read_b32 %5, a95\n\tv_accvgpr_read_b32 %6, a111\n\tv_accvgpr_read_b32 %7, a127\n\tv_accvgpr_read_b32 %8, a143\n\tv_accvgpr_read_b32 %9, a159\n\tv_accvgpr_read_b32 %10, a175\n\tv_accvgpr_read_b32 %11, a191\n\tv_accvgpr_read_b32 %12, a207\n\tv_accvgpr_read_b32 %13, a223\n\tv_accvgpr_read_b32 %14, a239\n\tv_accvgpr_read_b32 %15, a255" : "=v"(m[0]), "=v"(m[1]), "=v"(m[2]), "=v"(m[3]), "=v"(m[4]), "=v"(m[5]), "=v"(m[6]), "=v"(m[7]), "=v"(m[8]), "=v"(m[9]), "=v"(m[10]), "=v"(m[11]), "=v"(m[12]), "=v"(m[13]), "=v"(m[14]), "=v"(m[15]))

#define WW_FENCE() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_wgw(WwGeom g, const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ part) {
    __shared__ __attribute__((aligned(16))) float smem_all[4 * WW_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    float *smem = smem_all + wave * WW_STAGE;
    const int HW = g.H * g.W;
    const unsigned npairs = (unsigned)(g.nkb * g.ncb);
    const unsigned u = blockIdx.x * 4 + wave;
    if (u >= npairs * (unsigned)g.nsplit) return;                // (no barriers anywhere: a wave may leave)
    const unsigned pair = u % npairs, split = u / npairs;
    const int kb = (int)(pair % (unsigned)g.nkb), cb = (int)(pair / (unsigned)g.nkb);
    const unsigned s_begin = split * g.su;
    const int nst = (int)min(g.su, g.nstages - s_begin);
    // loader coordinates (uniform): the stage whose loads are issued next -> image n, tile row ty, segment tseg
    const unsigned per_img = (unsigned)(g.th * g.nseg);
    int n = (int)(s_begin / per_img);
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
    const int xrow = rem & 3;
    const int vx_const = ((rem >> 2) * HW + xrow * g.W) * 4 + qd * 16 + 16;         // x item (c = 2 j + rem / 4, row = rem % 4)
    const int vg_const = ((rem >> 1) * HW + (rem & 1) * g.W) * 4 + qd * 16;         // gy item (k = 4 j + rem / 2, row = rem % 2)
    const int hside = lane & 1, hrow = (lane >> 1) & 3;                             // halo item (c = 8 j + lane / 8, row, side)
    const int vh_const = ((lane >> 3) * HW + hrow * g.W) * 4 + (hside ? 28 * 4 + 16 : 12);
    const int xw_addr = (rem >> 2) * WW_XC + xrow * 32 + 2 + 4 * qd;
    const int gw_addr = WW_XS + (rem >> 1) * WW_GC + (rem & 1) * 28 + 4 * qd;
    const int hw_addr = (lane >> 3) * WW_XC + hrow * 32 + (hside ? 30 : 1);
    int sx, sgo, vx, vg, vh;                                     // stage part of the offsets (scalar) / per-lane part with validity
    auto stage_offsets = [&]() {
        sx = (((n - n0) * g.C + cb * 32) * HW + 2 * ty * g.W + 28 * tseg) * 4;
        sgo = (((n - n0) * g.K + kb * 32) * HW + 2 * ty * g.W + 28 * tseg) * 4;
        const bool top = ty == 0, bot = ty == g.th - 1;
        vx = ((xrow == 0 && top) || (xrow == 3 && bot)) ? kOOR : vx_const;
        vg = vg_const;
        vh = ((hrow == 0 && top) || (hrow == 3 && bot) || (hside == 0 && tseg == 0) || (hside == 1 && tseg == g.nseg - 1)) ? kOOR : vh_const;
    };
    auto advance_stage = [&]() {
        if (++tseg == g.nseg) {
            tseg = 0;
            if (++ty == g.th) ty = 0, ++n;
        }
        stage_offsets();
    };
    i32x4 rx[16], rg[8];
    float rh[4];
    auto g_load = [&](int idx, int) {
        if (idx < 16)
            rx[idx] = __builtin_amdgcn_raw_buffer_load_b128(srd_x, vx, sx + idx * 2 * HW * 4, 0);
        else if (idx < 24)
            rg[idx - 16] = __builtin_amdgcn_raw_buffer_load_b128(srd_g, vg, sgo + (idx - 16) * 4 * HW * 4, 0);
        else
            rh[idx - 24] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_x, vh, sx + (idx - 24) * 8 * HW * 4, 0));
    };
    auto w_store = [&](int idx) {
        if (idx < 16) {
            i32x2 *d = reinterpret_cast<i32x2 *>(smem + xw_addr + idx * 2 * WW_XC);
            i32x2 lo, hi;
            lo[0] = rx[idx][0], lo[1] = rx[idx][1], hi[0] = rx[idx][2], hi[1] = rx[idx][3];
            d[0] = lo, d[1] = hi;
        } else if (idx < 24) {
            i32x2 *d = reinterpret_cast<i32x2 *>(smem + gw_addr + (idx - 16) * 4 * WW_GC);
            i32x2 lo, hi;
            lo[0] = rg[idx - 16][0], lo[1] = rg[idx - 16][1], hi[0] = rg[idx - 16][2], hi[1] = rg[idx - 16][3];
            d[0] = lo, d[1] = hi;
        } else {
            smem[hw_addr + (idx - 24) * 8 * WW_XC] = rh[idx - 24];
        }
    };
    // ---- T: operands of k-step ks (tile 2 ks + lh of the stage in LDS), in 14 micro steps ----
    const int xr_base = li * WW_XC + (lh + 1) * 2, gr_base = WW_XS + li * WW_GC + lh * 2;
    auto t_micro = [&](int m, int ks, float (&A)[16], float (&B)[16]) {
        if (m < 4) {                                             // patch row m: own pair + the neighbours' halves
            const float *r = smem + xr_base + 4 * ks + m * 32;
            const f32x2 own = *reinterpret_cast<const f32x2 *>(r);
            B[m * 4 + 0] = r[-1], B[m * 4 + 1] = own[0], B[m * 4 + 2] = own[1], B[m * 4 + 3] = r[2];
        } else if (m == 4) {                                     // the gy tile: A[0] = y00, A[3] = y01, A[12] = y10, A[15] = y11
            const f32x2 y0 = *reinterpret_cast<const f32x2 *>(smem + gr_base + 4 * ks);
            const f32x2 y1 = *reinterpret_cast<const f32x2 *>(smem + gr_base + 4 * ks + 28);
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

    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0" : : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    asm volatile("v_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0" : : : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    asm volatile("v_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0" : : : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    asm volatile("v_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0" : : : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    asm volatile("v_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0" : : : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    asm volatile("v_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0" : : : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    asm volatile("v_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0" : : : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    asm volatile("v_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" : : : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
    asm volatile("v_accvgpr_write_b32 a128, 0\n\tv_accvgpr_write_b32 a129, 0\n\tv_accvgpr_write_b32 a130, 0\n\tv_accvgpr_write_b32 a131, 0\n\tv_accvgpr_write_b32 a132, 0\n\tv_accvgpr_write_b32 a133, 0\n\tv_accvgpr_write_b32 a134, 0\n\tv_accvgpr_write_b32 a135, 0\n\tv_accvgpr_write_b32 a136, 0\n\tv_accvgpr_write_b32 a137, 0\n\tv_accvgpr_write_b32 a138, 0\n\tv_accvgpr_write_b32 a139, 0\n\tv_accvgpr_write_b32 a140, 0\n\tv_accvgpr_write_b32 a141, 0\n\tv_accvgpr_write_b32 a142, 0\n\tv_accvgpr_write_b32 a143, 0" : : : "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143");
    asm volatile("v_accvgpr_write_b32 a144, 0\n\tv_accvgpr_write_b32 a145, 0\n\tv_accvgpr_write_b32 a146, 0\n\tv_accvgpr_write_b32 a147, 0\n\tv_accvgpr_write_b32 a148, 0\n\tv_accvgpr_write_b32 a149, 0\n\tv_accvgpr_write_b32 a150, 0\n\tv_accvgpr_write_b32 a151, 0\n\tv_accvgpr_write_b32 a152, 0\n\tv_accvgpr_write_b32 a153, 0\n\tv_accvgpr_write_b32 a154, 0\n\tv_accvgpr_write_b32 a155, 0\n\tv_accvgpr_write_b32 a156, 0\n\tv_accvgpr_write_b32 a157, 0\n\tv_accvgpr_write_b32 a158, 0\n\tv_accvgpr_write_b32 a159, 0" : : : "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159");
    asm volatile("v_accvgpr_write_b32 a160, 0\n\tv_accvgpr_write_b32 a161, 0\n\tv_accvgpr_write_b32 a162, 0\n\tv_accvgpr_write_b32 a163, 0\n\tv_accvgpr_write_b32 a164, 0\n\tv_accvgpr_write_b32 a165, 0\n\tv_accvgpr_write_b32 a166, 0\n\tv_accvgpr_write_b32 a167, 0\n\tv_accvgpr_write_b32 a168, 0\n\tv_accvgpr_write_b32 a169, 0\n\tv_accvgpr_write_b32 a170, 0\n\tv_accvgpr_write_b32 a171, 0\n\tv_accvgpr_write_b32 a172, 0\n\tv_accvgpr_write_b32 a173, 0\n\tv_accvgpr_write_b32 a174, 0\n\tv_accvgpr_write_b32 a175, 0" : : : "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175");
    asm volatile("v_accvgpr_write_b32 a176, 0\n\tv_accvgpr_write_b32 a177, 0\n\tv_accvgpr_write_b32 a178, 0\n\tv_accvgpr_write_b32 a179, 0\n\tv_accvgpr_write_b32 a180, 0\n\tv_accvgpr_write_b32 a181, 0\n\tv_accvgpr_write_b32 a182, 0\n\tv_accvgpr_write_b32 a183, 0\n\tv_accvgpr_write_b32 a184, 0\n\tv_accvgpr_write_b32 a185, 0\n\tv_accvgpr_write_b32 a186, 0\n\tv_accvgpr_write_b32 a187, 0\n\tv_accvgpr_write_b32 a188, 0\n\tv_accvgpr_write_b32 a189, 0\n\tv_accvgpr_write_b32 a190, 0\n\tv_accvgpr_write_b32 a191, 0" : : : "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191");
    asm volatile("v_accvgpr_write_b32 a192, 0\n\tv_accvgpr_write_b32 a193, 0\n\tv_accvgpr_write_b32 a194, 0\n\tv_accvgpr_write_b32 a195, 0\n\tv_accvgpr_write_b32 a196, 0\n\tv_accvgpr_write_b32 a197, 0\n\tv_accvgpr_write_b32 a198, 0\n\tv_accvgpr_write_b32 a199, 0\n\tv_accvgpr_write_b32 a200, 0\n\tv_accvgpr_write_b32 a201, 0\n\tv_accvgpr_write_b32 a202, 0\n\tv_accvgpr_write_b32 a203, 0\n\tv_accvgpr_write_b32 a204, 0\n\tv_accvgpr_write_b32 a205, 0\n\tv_accvgpr_write_b32 a206, 0\n\tv_accvgpr_write_b32 a207, 0" : : : "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207");
    asm volatile("v_accvgpr_write_b32 a208, 0\n\tv_accvgpr_write_b32 a209, 0\n\tv_accvgpr_write_b32 a210, 0\n\tv_accvgpr_write_b32 a211, 0\n\tv_accvgpr_write_b32 a212, 0\n\tv_accvgpr_write_b32 a213, 0\n\tv_accvgpr_write_b32 a214, 0\n\tv_accvgpr_write_b32 a215, 0\n\tv_accvgpr_write_b32 a216, 0\n\tv_accvgpr_write_b32 a217, 0\n\tv_accvgpr_write_b32 a218, 0\n\tv_accvgpr_write_b32 a219, 0\n\tv_accvgpr_write_b32 a220, 0\n\tv_accvgpr_write_b32 a221, 0\n\tv_accvgpr_write_b32 a222, 0\n\tv_accvgpr_write_b32 a223, 0" : : : "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223");
    asm volatile("v_accvgpr_write_b32 a224, 0\n\tv_accvgpr_write_b32 a225, 0\n\tv_accvgpr_write_b32 a226, 0\n\tv_accvgpr_write_b32 a227, 0\n\tv_accvgpr_write_b32 a228, 0\n\tv_accvgpr_write_b32 a229, 0\n\tv_accvgpr_write_b32 a230, 0\n\tv_accvgpr_write_b32 a231, 0\n\tv_accvgpr_write_b32 a232, 0\n\tv_accvgpr_write_b32 a233, 0\n\tv_accvgpr_write_b32 a234, 0\n\tv_accvgpr_write_b32 a235, 0\n\tv_accvgpr_write_b32 a236, 0\n\tv_accvgpr_write_b32 a237, 0\n\tv_accvgpr_write_b32 a238, 0\n\tv_accvgpr_write_b32 a239, 0" : : : "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239");
    asm volatile("v_accvgpr_write_b32 a240, 0\n\tv_accvgpr_write_b32 a241, 0\n\tv_accvgpr_write_b32 a242, 0\n\tv_accvgpr_write_b32 a243, 0\n\tv_accvgpr_write_b32 a244, 0\n\tv_accvgpr_write_b32 a245, 0\n\tv_accvgpr_write_b32 a246, 0\n\tv_accvgpr_write_b32 a247, 0\n\tv_accvgpr_write_b32 a248, 0\n\tv_accvgpr_write_b32 a249, 0\n\tv_accvgpr_write_b32 a250, 0\n\tv_accvgpr_write_b32 a251, 0\n\tv_accvgpr_write_b32 a252, 0\n\tv_accvgpr_write_b32 a253, 0\n\tv_accvgpr_write_b32 a254, 0\n\tv_accvgpr_write_b32 a255, 0" : : : "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");

    float A0[16], B0[16], A1[16], B1[16];
    // prologue: stage 0 into LDS, its first operands, stage 1's coordinates ready
    stage_offsets();
#pragma unroll
    for (int i = 0; i < 28; ++i) g_load(i, 0);
#pragma unroll
    for (int i = 0; i < 28; ++i) w_store(i);
    advance_stage();
#pragma unroll
    for (int m = 0; m < 14; ++m) t_micro(m, 0, A0, B0);

    for (int st = 0; st < nst; st += 2) {
        {   // stage A of the pair
            const bool more = st + 0 + 1 < nst; const int nst_idx = st + 0 + 1;
            // k-step 0
            WW_ONE_0(A0[0], B0[0]); t_micro(0, 1, A1, B1); WW_FENCE();
            WW_ONE_1(A0[1], B0[1]); t_micro(1, 1, A1, B1); WW_FENCE();
            WW_ONE_2(A0[2], B0[2]); t_micro(2, 1, A1, B1); WW_FENCE();
            WW_ONE_3(A0[3], B0[3]); t_micro(3, 1, A1, B1); WW_FENCE();
            WW_ONE_4(A0[4], B0[4]); t_micro(4, 1, A1, B1); WW_FENCE();
            WW_ONE_5(A0[5], B0[5]); t_micro(5, 1, A1, B1); WW_FENCE();
            WW_ONE_6(A0[6], B0[6]); t_micro(6, 1, A1, B1); WW_FENCE();
            WW_ONE_7(A0[7], B0[7]); t_micro(7, 1, A1, B1); WW_FENCE();
            WW_ONE_8(A0[8], B0[8]); t_micro(8, 1, A1, B1); WW_FENCE();
            WW_ONE_9(A0[9], B0[9]); t_micro(9, 1, A1, B1); WW_FENCE();
            WW_ONE_10(A0[10], B0[10]); t_micro(10, 1, A1, B1); WW_FENCE();
            WW_ONE_11(A0[11], B0[11]); t_micro(11, 1, A1, B1); WW_FENCE();
            WW_ONE_12(A0[12], B0[12]); t_micro(12, 1, A1, B1); WW_FENCE();
            WW_ONE_13(A0[13], B0[13]); t_micro(13, 1, A1, B1); WW_FENCE();
            WW_ONE_14(A0[14], B0[14]); ; WW_FENCE();
            WW_ONE_15(A0[15], B0[15]); ; WW_FENCE();
            // k-step 1
            WW_ONE_0(A1[0], B1[0]); t_micro(0, 2, A0, B0); g_load(0, nst_idx); WW_FENCE();
            WW_ONE_1(A1[1], B1[1]); t_micro(1, 2, A0, B0); g_load(1, nst_idx); WW_FENCE();
            WW_ONE_2(A1[2], B1[2]); t_micro(2, 2, A0, B0); g_load(2, nst_idx); WW_FENCE();
            WW_ONE_3(A1[3], B1[3]); t_micro(3, 2, A0, B0); g_load(3, nst_idx); WW_FENCE();
            WW_ONE_4(A1[4], B1[4]); t_micro(4, 2, A0, B0); g_load(4, nst_idx); WW_FENCE();
            WW_ONE_5(A1[5], B1[5]); t_micro(5, 2, A0, B0); g_load(5, nst_idx); WW_FENCE();
            WW_ONE_6(A1[6], B1[6]); t_micro(6, 2, A0, B0); g_load(6, nst_idx); WW_FENCE();
            WW_ONE_7(A1[7], B1[7]); t_micro(7, 2, A0, B0); g_load(7, nst_idx); WW_FENCE();
            WW_ONE_8(A1[8], B1[8]); t_micro(8, 2, A0, B0); g_load(8, nst_idx); WW_FENCE();
            WW_ONE_9(A1[9], B1[9]); t_micro(9, 2, A0, B0); g_load(9, nst_idx); WW_FENCE();
            WW_ONE_10(A1[10], B1[10]); t_micro(10, 2, A0, B0); g_load(10, nst_idx); WW_FENCE();
            WW_ONE_11(A1[11], B1[11]); t_micro(11, 2, A0, B0); g_load(11, nst_idx); WW_FENCE();
            WW_ONE_12(A1[12], B1[12]); t_micro(12, 2, A0, B0); g_load(12, nst_idx); WW_FENCE();
            WW_ONE_13(A1[13], B1[13]); t_micro(13, 2, A0, B0); g_load(13, nst_idx); WW_FENCE();
            WW_ONE_14(A1[14], B1[14]); ; WW_FENCE();
            WW_ONE_15(A1[15], B1[15]); ; WW_FENCE();
            // k-step 2
            WW_ONE_0(A0[0], B0[0]); t_micro(0, 3, A1, B1); g_load(14, nst_idx); WW_FENCE();
            WW_ONE_1(A0[1], B0[1]); t_micro(1, 3, A1, B1); g_load(15, nst_idx); WW_FENCE();
            WW_ONE_2(A0[2], B0[2]); t_micro(2, 3, A1, B1); g_load(16, nst_idx); WW_FENCE();
            WW_ONE_3(A0[3], B0[3]); t_micro(3, 3, A1, B1); g_load(17, nst_idx); WW_FENCE();
            WW_ONE_4(A0[4], B0[4]); t_micro(4, 3, A1, B1); g_load(18, nst_idx); WW_FENCE();
            WW_ONE_5(A0[5], B0[5]); t_micro(5, 3, A1, B1); g_load(19, nst_idx); WW_FENCE();
            WW_ONE_6(A0[6], B0[6]); t_micro(6, 3, A1, B1); g_load(20, nst_idx); WW_FENCE();
            WW_ONE_7(A0[7], B0[7]); t_micro(7, 3, A1, B1); g_load(21, nst_idx); WW_FENCE();
            WW_ONE_8(A0[8], B0[8]); t_micro(8, 3, A1, B1); g_load(22, nst_idx); WW_FENCE();
            WW_ONE_9(A0[9], B0[9]); t_micro(9, 3, A1, B1); g_load(23, nst_idx); WW_FENCE();
            WW_ONE_10(A0[10], B0[10]); t_micro(10, 3, A1, B1); g_load(24, nst_idx); WW_FENCE();
            WW_ONE_11(A0[11], B0[11]); t_micro(11, 3, A1, B1); g_load(25, nst_idx); WW_FENCE();
            WW_ONE_12(A0[12], B0[12]); t_micro(12, 3, A1, B1); g_load(26, nst_idx); WW_FENCE();
            WW_ONE_13(A0[13], B0[13]); t_micro(13, 3, A1, B1); g_load(27, nst_idx); WW_FENCE();
            WW_ONE_14(A0[14], B0[14]); ; WW_FENCE();
            WW_ONE_15(A0[15], B0[15]); ; WW_FENCE();
            // k-step 3
            WW_ONE_0(A1[0], B1[0]); t_micro(0, 4, A0, B0); WW_FENCE();
            WW_ONE_1(A1[1], B1[1]); t_micro(1, 4, A0, B0); WW_FENCE();
            WW_ONE_2(A1[2], B1[2]); t_micro(2, 4, A0, B0); WW_FENCE();
            WW_ONE_3(A1[3], B1[3]); t_micro(3, 4, A0, B0); WW_FENCE();
            WW_ONE_4(A1[4], B1[4]); t_micro(4, 4, A0, B0); WW_FENCE();
            WW_ONE_5(A1[5], B1[5]); t_micro(5, 4, A0, B0); WW_FENCE();
            WW_ONE_6(A1[6], B1[6]); t_micro(6, 4, A0, B0); WW_FENCE();
            WW_ONE_7(A1[7], B1[7]); t_micro(7, 4, A0, B0); WW_FENCE();
            WW_ONE_8(A1[8], B1[8]); t_micro(8, 4, A0, B0); WW_FENCE();
            WW_ONE_9(A1[9], B1[9]); t_micro(9, 4, A0, B0); WW_FENCE();
            WW_ONE_10(A1[10], B1[10]); t_micro(10, 4, A0, B0); WW_FENCE();
            WW_ONE_11(A1[11], B1[11]); t_micro(11, 4, A0, B0); WW_FENCE();
            WW_ONE_12(A1[12], B1[12]); t_micro(12, 4, A0, B0); WW_FENCE();
            WW_ONE_13(A1[13], B1[13]); t_micro(13, 4, A0, B0); WW_FENCE();
            WW_ONE_14(A1[14], B1[14]); ; WW_FENCE();
            WW_ONE_15(A1[15], B1[15]); ; WW_FENCE();
            // k-step 4
            WW_ONE_0(A0[0], B0[0]); t_micro(0, 5, A1, B1); WW_FENCE();
            WW_ONE_1(A0[1], B0[1]); t_micro(1, 5, A1, B1); WW_FENCE();
            WW_ONE_2(A0[2], B0[2]); t_micro(2, 5, A1, B1); WW_FENCE();
            WW_ONE_3(A0[3], B0[3]); t_micro(3, 5, A1, B1); WW_FENCE();
            WW_ONE_4(A0[4], B0[4]); t_micro(4, 5, A1, B1); WW_FENCE();
            WW_ONE_5(A0[5], B0[5]); t_micro(5, 5, A1, B1); WW_FENCE();
            WW_ONE_6(A0[6], B0[6]); t_micro(6, 5, A1, B1); WW_FENCE();
            WW_ONE_7(A0[7], B0[7]); t_micro(7, 5, A1, B1); WW_FENCE();
            WW_ONE_8(A0[8], B0[8]); t_micro(8, 5, A1, B1); WW_FENCE();
            WW_ONE_9(A0[9], B0[9]); t_micro(9, 5, A1, B1); WW_FENCE();
            WW_ONE_10(A0[10], B0[10]); t_micro(10, 5, A1, B1); WW_FENCE();
            WW_ONE_11(A0[11], B0[11]); t_micro(11, 5, A1, B1); WW_FENCE();
            WW_ONE_12(A0[12], B0[12]); t_micro(12, 5, A1, B1); WW_FENCE();
            WW_ONE_13(A0[13], B0[13]); t_micro(13, 5, A1, B1); WW_FENCE();
            WW_ONE_14(A0[14], B0[14]); ; WW_FENCE();
            WW_ONE_15(A0[15], B0[15]); ; WW_FENCE();
            // k-step 5
            WW_ONE_0(A1[0], B1[0]); t_micro(0, 6, A0, B0); WW_FENCE();
            WW_ONE_1(A1[1], B1[1]); t_micro(1, 6, A0, B0); WW_FENCE();
            WW_ONE_2(A1[2], B1[2]); t_micro(2, 6, A0, B0); WW_FENCE();
            WW_ONE_3(A1[3], B1[3]); t_micro(3, 6, A0, B0); WW_FENCE();
            WW_ONE_4(A1[4], B1[4]); t_micro(4, 6, A0, B0); WW_FENCE();
            WW_ONE_5(A1[5], B1[5]); t_micro(5, 6, A0, B0); WW_FENCE();
            WW_ONE_6(A1[6], B1[6]); t_micro(6, 6, A0, B0); WW_FENCE();
            WW_ONE_7(A1[7], B1[7]); t_micro(7, 6, A0, B0); WW_FENCE();
            WW_ONE_8(A1[8], B1[8]); t_micro(8, 6, A0, B0); WW_FENCE();
            WW_ONE_9(A1[9], B1[9]); t_micro(9, 6, A0, B0); WW_FENCE();
            WW_ONE_10(A1[10], B1[10]); t_micro(10, 6, A0, B0); WW_FENCE();
            WW_ONE_11(A1[11], B1[11]); t_micro(11, 6, A0, B0); WW_FENCE();
            WW_ONE_12(A1[12], B1[12]); t_micro(12, 6, A0, B0); WW_FENCE();
            WW_ONE_13(A1[13], B1[13]); t_micro(13, 6, A0, B0); WW_FENCE();
            WW_ONE_14(A1[14], B1[14]); ; WW_FENCE();
            WW_ONE_15(A1[15], B1[15]); ; WW_FENCE();
            // k-step 6
            WW_ONE_0(A0[0], B0[0]); w_store(0); w_store(1); w_store(2); w_store(3); WW_FENCE();
            WW_ONE_1(A0[1], B0[1]); w_store(4); w_store(5); w_store(6); w_store(7); WW_FENCE();
            WW_ONE_2(A0[2], B0[2]); w_store(8); w_store(9); w_store(10); w_store(11); WW_FENCE();
            WW_ONE_3(A0[3], B0[3]); w_store(12); w_store(13); w_store(14); w_store(15); WW_FENCE();
            WW_ONE_4(A0[4], B0[4]); w_store(16); w_store(17); w_store(18); w_store(19); WW_FENCE();
            WW_ONE_5(A0[5], B0[5]); w_store(20); w_store(21); w_store(22); w_store(23); WW_FENCE();
            WW_ONE_6(A0[6], B0[6]); w_store(24); w_store(25); w_store(26); w_store(27); WW_FENCE();
            WW_ONE_7(A0[7], B0[7]); if (more) t_micro(0, 0, A1, B1); if (more) t_micro(1, 0, A1, B1); WW_FENCE();
            WW_ONE_8(A0[8], B0[8]); if (more) t_micro(2, 0, A1, B1); if (more) t_micro(3, 0, A1, B1); WW_FENCE();
            WW_ONE_9(A0[9], B0[9]); if (more) t_micro(4, 0, A1, B1); WW_FENCE();
            WW_ONE_10(A0[10], B0[10]); if (more) t_micro(5, 0, A1, B1); if (more) t_micro(6, 0, A1, B1); WW_FENCE();
            WW_ONE_11(A0[11], B0[11]); if (more) t_micro(7, 0, A1, B1); if (more) t_micro(8, 0, A1, B1); WW_FENCE();
            WW_ONE_12(A0[12], B0[12]); if (more) t_micro(9, 0, A1, B1); if (more) t_micro(10, 0, A1, B1); WW_FENCE();
            WW_ONE_13(A0[13], B0[13]); if (more) t_micro(11, 0, A1, B1); WW_FENCE();
            WW_ONE_14(A0[14], B0[14]); if (more) t_micro(12, 0, A1, B1); WW_FENCE();
            WW_ONE_15(A0[15], B0[15]); if (more) t_micro(13, 0, A1, B1); WW_FENCE();
            advance_stage();
        }
        {   // stage B of the pair
            const bool more = st + 1 + 1 < nst; const int nst_idx = st + 1 + 1;
            if (st + 1 >= nst) break;
            // k-step 0
            WW_ONE_0(A1[0], B1[0]); t_micro(0, 1, A0, B0); WW_FENCE();
            WW_ONE_1(A1[1], B1[1]); t_micro(1, 1, A0, B0); WW_FENCE();
            WW_ONE_2(A1[2], B1[2]); t_micro(2, 1, A0, B0); WW_FENCE();
            WW_ONE_3(A1[3], B1[3]); t_micro(3, 1, A0, B0); WW_FENCE();
            WW_ONE_4(A1[4], B1[4]); t_micro(4, 1, A0, B0); WW_FENCE();
            WW_ONE_5(A1[5], B1[5]); t_micro(5, 1, A0, B0); WW_FENCE();
            WW_ONE_6(A1[6], B1[6]); t_micro(6, 1, A0, B0); WW_FENCE();
            WW_ONE_7(A1[7], B1[7]); t_micro(7, 1, A0, B0); WW_FENCE();
            WW_ONE_8(A1[8], B1[8]); t_micro(8, 1, A0, B0); WW_FENCE();
            WW_ONE_9(A1[9], B1[9]); t_micro(9, 1, A0, B0); WW_FENCE();
            WW_ONE_10(A1[10], B1[10]); t_micro(10, 1, A0, B0); WW_FENCE();
            WW_ONE_11(A1[11], B1[11]); t_micro(11, 1, A0, B0); WW_FENCE();
            WW_ONE_12(A1[12], B1[12]); t_micro(12, 1, A0, B0); WW_FENCE();
            WW_ONE_13(A1[13], B1[13]); t_micro(13, 1, A0, B0); WW_FENCE();
            WW_ONE_14(A1[14], B1[14]); ; WW_FENCE();
            WW_ONE_15(A1[15], B1[15]); ; WW_FENCE();
            // k-step 1
            WW_ONE_0(A0[0], B0[0]); t_micro(0, 2, A1, B1); g_load(0, nst_idx); WW_FENCE();
            WW_ONE_1(A0[1], B0[1]); t_micro(1, 2, A1, B1); g_load(1, nst_idx); WW_FENCE();
            WW_ONE_2(A0[2], B0[2]); t_micro(2, 2, A1, B1); g_load(2, nst_idx); WW_FENCE();
            WW_ONE_3(A0[3], B0[3]); t_micro(3, 2, A1, B1); g_load(3, nst_idx); WW_FENCE();
            WW_ONE_4(A0[4], B0[4]); t_micro(4, 2, A1, B1); g_load(4, nst_idx); WW_FENCE();
            WW_ONE_5(A0[5], B0[5]); t_micro(5, 2, A1, B1); g_load(5, nst_idx); WW_FENCE();
            WW_ONE_6(A0[6], B0[6]); t_micro(6, 2, A1, B1); g_load(6, nst_idx); WW_FENCE();
            WW_ONE_7(A0[7], B0[7]); t_micro(7, 2, A1, B1); g_load(7, nst_idx); WW_FENCE();
            WW_ONE_8(A0[8], B0[8]); t_micro(8, 2, A1, B1); g_load(8, nst_idx); WW_FENCE();
            WW_ONE_9(A0[9], B0[9]); t_micro(9, 2, A1, B1); g_load(9, nst_idx); WW_FENCE();
            WW_ONE_10(A0[10], B0[10]); t_micro(10, 2, A1, B1); g_load(10, nst_idx); WW_FENCE();
            WW_ONE_11(A0[11], B0[11]); t_micro(11, 2, A1, B1); g_load(11, nst_idx); WW_FENCE();
            WW_ONE_12(A0[12], B0[12]); t_micro(12, 2, A1, B1); g_load(12, nst_idx); WW_FENCE();
            WW_ONE_13(A0[13], B0[13]); t_micro(13, 2, A1, B1); g_load(13, nst_idx); WW_FENCE();
            WW_ONE_14(A0[14], B0[14]); ; WW_FENCE();
            WW_ONE_15(A0[15], B0[15]); ; WW_FENCE();
            // k-step 2
            WW_ONE_0(A1[0], B1[0]); t_micro(0, 3, A0, B0); g_load(14, nst_idx); WW_FENCE();
            WW_ONE_1(A1[1], B1[1]); t_micro(1, 3, A0, B0); g_load(15, nst_idx); WW_FENCE();
            WW_ONE_2(A1[2], B1[2]); t_micro(2, 3, A0, B0); g_load(16, nst_idx); WW_FENCE();
            WW_ONE_3(A1[3], B1[3]); t_micro(3, 3, A0, B0); g_load(17, nst_idx); WW_FENCE();
            WW_ONE_4(A1[4], B1[4]); t_micro(4, 3, A0, B0); g_load(18, nst_idx); WW_FENCE();
            WW_ONE_5(A1[5], B1[5]); t_micro(5, 3, A0, B0); g_load(19, nst_idx); WW_FENCE();
            WW_ONE_6(A1[6], B1[6]); t_micro(6, 3, A0, B0); g_load(20, nst_idx); WW_FENCE();
            WW_ONE_7(A1[7], B1[7]); t_micro(7, 3, A0, B0); g_load(21, nst_idx); WW_FENCE();
            WW_ONE_8(A1[8], B1[8]); t_micro(8, 3, A0, B0); g_load(22, nst_idx); WW_FENCE();
            WW_ONE_9(A1[9], B1[9]); t_micro(9, 3, A0, B0); g_load(23, nst_idx); WW_FENCE();
            WW_ONE_10(A1[10], B1[10]); t_micro(10, 3, A0, B0); g_load(24, nst_idx); WW_FENCE();
            WW_ONE_11(A1[11], B1[11]); t_micro(11, 3, A0, B0); g_load(25, nst_idx); WW_FENCE();
            WW_ONE_12(A1[12], B1[12]); t_micro(12, 3, A0, B0); g_load(26, nst_idx); WW_FENCE();
            WW_ONE_13(A1[13], B1[13]); t_micro(13, 3, A0, B0); g_load(27, nst_idx); WW_FENCE();
            WW_ONE_14(A1[14], B1[14]); ; WW_FENCE();
            WW_ONE_15(A1[15], B1[15]); ; WW_FENCE();
            // k-step 3
            WW_ONE_0(A0[0], B0[0]); t_micro(0, 4, A1, B1); WW_FENCE();
            WW_ONE_1(A0[1], B0[1]); t_micro(1, 4, A1, B1); WW_FENCE();
            WW_ONE_2(A0[2], B0[2]); t_micro(2, 4, A1, B1); WW_FENCE();
            WW_ONE_3(A0[3], B0[3]); t_micro(3, 4, A1, B1); WW_FENCE();
            WW_ONE_4(A0[4], B0[4]); t_micro(4, 4, A1, B1); WW_FENCE();
            WW_ONE_5(A0[5], B0[5]); t_micro(5, 4, A1, B1); WW_FENCE();
            WW_ONE_6(A0[6], B0[6]); t_micro(6, 4, A1, B1); WW_FENCE();
            WW_ONE_7(A0[7], B0[7]); t_micro(7, 4, A1, B1); WW_FENCE();
            WW_ONE_8(A0[8], B0[8]); t_micro(8, 4, A1, B1); WW_FENCE();
            WW_ONE_9(A0[9], B0[9]); t_micro(9, 4, A1, B1); WW_FENCE();
            WW_ONE_10(A0[10], B0[10]); t_micro(10, 4, A1, B1); WW_FENCE();
            WW_ONE_11(A0[11], B0[11]); t_micro(11, 4, A1, B1); WW_FENCE();
            WW_ONE_12(A0[12], B0[12]); t_micro(12, 4, A1, B1); WW_FENCE();
            WW_ONE_13(A0[13], B0[13]); t_micro(13, 4, A1, B1); WW_FENCE();
            WW_ONE_14(A0[14], B0[14]); ; WW_FENCE();
            WW_ONE_15(A0[15], B0[15]); ; WW_FENCE();
            // k-step 4
            WW_ONE_0(A1[0], B1[0]); t_micro(0, 5, A0, B0); WW_FENCE();
            WW_ONE_1(A1[1], B1[1]); t_micro(1, 5, A0, B0); WW_FENCE();
            WW_ONE_2(A1[2], B1[2]); t_micro(2, 5, A0, B0); WW_FENCE();
            WW_ONE_3(A1[3], B1[3]); t_micro(3, 5, A0, B0); WW_FENCE();
            WW_ONE_4(A1[4], B1[4]); t_micro(4, 5, A0, B0); WW_FENCE();
            WW_ONE_5(A1[5], B1[5]); t_micro(5, 5, A0, B0); WW_FENCE();
            WW_ONE_6(A1[6], B1[6]); t_micro(6, 5, A0, B0); WW_FENCE();
            WW_ONE_7(A1[7], B1[7]); t_micro(7, 5, A0, B0); WW_FENCE();
            WW_ONE_8(A1[8], B1[8]); t_micro(8, 5, A0, B0); WW_FENCE();
            WW_ONE_9(A1[9], B1[9]); t_micro(9, 5, A0, B0); WW_FENCE();
            WW_ONE_10(A1[10], B1[10]); t_micro(10, 5, A0, B0); WW_FENCE();
            WW_ONE_11(A1[11], B1[11]); t_micro(11, 5, A0, B0); WW_FENCE();
            WW_ONE_12(A1[12], B1[12]); t_micro(12, 5, A0, B0); WW_FENCE();
            WW_ONE_13(A1[13], B1[13]); t_micro(13, 5, A0, B0); WW_FENCE();
            WW_ONE_14(A1[14], B1[14]); ; WW_FENCE();
            WW_ONE_15(A1[15], B1[15]); ; WW_FENCE();
            // k-step 5
            WW_ONE_0(A0[0], B0[0]); t_micro(0, 6, A1, B1); WW_FENCE();
            WW_ONE_1(A0[1], B0[1]); t_micro(1, 6, A1, B1); WW_FENCE();
            WW_ONE_2(A0[2], B0[2]); t_micro(2, 6, A1, B1); WW_FENCE();
            WW_ONE_3(A0[3], B0[3]); t_micro(3, 6, A1, B1); WW_FENCE();
            WW_ONE_4(A0[4], B0[4]); t_micro(4, 6, A1, B1); WW_FENCE();
            WW_ONE_5(A0[5], B0[5]); t_micro(5, 6, A1, B1); WW_FENCE();
            WW_ONE_6(A0[6], B0[6]); t_micro(6, 6, A1, B1); WW_FENCE();
            WW_ONE_7(A0[7], B0[7]); t_micro(7, 6, A1, B1); WW_FENCE();
            WW_ONE_8(A0[8], B0[8]); t_micro(8, 6, A1, B1); WW_FENCE();
            WW_ONE_9(A0[9], B0[9]); t_micro(9, 6, A1, B1); WW_FENCE();
            WW_ONE_10(A0[10], B0[10]); t_micro(10, 6, A1, B1); WW_FENCE();
            WW_ONE_11(A0[11], B0[11]); t_micro(11, 6, A1, B1); WW_FENCE();
            WW_ONE_12(A0[12], B0[12]); t_micro(12, 6, A1, B1); WW_FENCE();
            WW_ONE_13(A0[13], B0[13]); t_micro(13, 6, A1, B1); WW_FENCE();
            WW_ONE_14(A0[14], B0[14]); ; WW_FENCE();
            WW_ONE_15(A0[15], B0[15]); ; WW_FENCE();
            // k-step 6
            WW_ONE_0(A1[0], B1[0]); w_store(0); w_store(1); w_store(2); w_store(3); WW_FENCE();
            WW_ONE_1(A1[1], B1[1]); w_store(4); w_store(5); w_store(6); w_store(7); WW_FENCE();
            WW_ONE_2(A1[2], B1[2]); w_store(8); w_store(9); w_store(10); w_store(11); WW_FENCE();
            WW_ONE_3(A1[3], B1[3]); w_store(12); w_store(13); w_store(14); w_store(15); WW_FENCE();
            WW_ONE_4(A1[4], B1[4]); w_store(16); w_store(17); w_store(18); w_store(19); WW_FENCE();
            WW_ONE_5(A1[5], B1[5]); w_store(20); w_store(21); w_store(22); w_store(23); WW_FENCE();
            WW_ONE_6(A1[6], B1[6]); w_store(24); w_store(25); w_store(26); w_store(27); WW_FENCE();
            WW_ONE_7(A1[7], B1[7]); if (more) t_micro(0, 0, A0, B0); if (more) t_micro(1, 0, A0, B0); WW_FENCE();
            WW_ONE_8(A1[8], B1[8]); if (more) t_micro(2, 0, A0, B0); if (more) t_micro(3, 0, A0, B0); WW_FENCE();
            WW_ONE_9(A1[9], B1[9]); if (more) t_micro(4, 0, A0, B0); WW_FENCE();
            WW_ONE_10(A1[10], B1[10]); if (more) t_micro(5, 0, A0, B0); if (more) t_micro(6, 0, A0, B0); WW_FENCE();
            WW_ONE_11(A1[11], B1[11]); if (more) t_micro(7, 0, A0, B0); if (more) t_micro(8, 0, A0, B0); WW_FENCE();
            WW_ONE_12(A1[12], B1[12]); if (more) t_micro(9, 0, A0, B0); if (more) t_micro(10, 0, A0, B0); WW_FENCE();
            WW_ONE_13(A1[13], B1[13]); if (more) t_micro(11, 0, A0, B0); WW_FENCE();
            WW_ONE_14(A1[14], B1[14]); if (more) t_micro(12, 0, A0, B0); WW_FENCE();
            WW_ONE_15(A1[15], B1[15]); if (more) t_micro(13, 0, A0, B0); WW_FENCE();
            advance_stage();
        }

    }

    // ---- epilogue: dg = G^T M G per (k, c); M[i][j] = sigma_i sigma_j acc[4 i + j], sigma = (1, 1, 1, -1) ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float *pout = part + ((int64_t)split * 9 * g.K + kb * 32) * g.C + cb * 32 + li;
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
        WW_RD_0(m); out_e(0, m);
        WW_RD_1(m); out_e(1, m);
        WW_RD_2(m); out_e(2, m);
        WW_RD_3(m); out_e(3, m);
        WW_RD_4(m); out_e(4, m);
        WW_RD_5(m); out_e(5, m);
        WW_RD_6(m); out_e(6, m);
        WW_RD_7(m); out_e(7, m);
        WW_RD_8(m); out_e(8, m);
        WW_RD_9(m); out_e(9, m);
        WW_RD_10(m); out_e(10, m);
        WW_RD_11(m); out_e(11, m);
        WW_RD_12(m); out_e(12, m);
        WW_RD_13(m); out_e(13, m);
        WW_RD_14(m); out_e(14, m);
        WW_RD_15(m); out_e(15, m);

    }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

struct WwPlan {
    WwGeom g;
    size_t ws_bytes;
    int64_t blocks;
};

bool ww_plan(const cpg_conv_desc *d, WwPlan &p) {
    if (getenv("CPG_NO_WINO") || getenv("CPG_NO_WINO_WGRAD")) return false;
    if (d->R != 3 || d->S != 3 || d->stride_h != 1 || d->stride_w != 1 || d->pad_h != 1 || d->pad_w != 1 || d->dil_h != 1 ||
        d->dil_w != 1 || d->groups != 1)
        return false;
    if (d->H % 2 || d->W % 28 || d->C % 32 || d->K % 32 || d->N < 1) return false;
    WwGeom &g = p.g;
    g.N = d->N, g.C = d->C, g.K = d->K, g.H = d->H, g.W = d->W;
    g.th = d->H / 2, g.tw = d->W / 2, g.nseg = g.tw / 14;
    const int64_t nstages = (int64_t)d->N * g.th * g.nseg;
    if (nstages >= (1ll << 28)) return false;
    g.nstages = (unsigned)nstages;
    g.nkb = d->K / 32, g.ncb = d->C / 32;
    const int64_t npairs = (int64_t)g.nkb * g.ncb;
    int64_t want = std::max<int64_t>(1, (6 * 4 * kCUs) / npairs);              // ~6 units per wave slot
    want = std::min<int64_t>(want, nstages);
    g.su = (unsigned)((nstages + want - 1) / want);
    g.nsplit = (int)((nstages + g.su - 1) / g.su);
    const int64_t per_img = (int64_t)g.th * g.nseg;
    g.span = (int)((g.su + per_img - 1) / per_img) + 1;
    const int64_t HW = (int64_t)d->H * d->W;
    if ((int64_t)g.span * std::max(d->C, d->K) * HW * 4 + (d->W + 4) * 4 >= (1ll << 31)) return false;
    p.ws_bytes = (size_t)g.nsplit * 9 * d->K * d->C * sizeof(float);
    p.blocks = (npairs * g.nsplit + 3) / 4;
    return p.blocks <= 0x7FFFFFFFll;
}

}  // namespace

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
    hipLaunchKernelGGL(k_wgw, dim3((unsigned)p.blocks), dim3(256), 0, stream, p.g, x, gy, (float *)ws);
    Epilogue ep{gw, nullptr, BIAS_NONE, 1, 1, pm, w, gpm, thr};
    const int64_t out_elems = (int64_t)d->K * d->C * 9;
    launch_split_reduce((const float *)ws, p.g.nsplit, out_elems, (int64_t)d->K * d->C, ep, stream);
    CPG_CHECK_LAUNCH("cpg_conv2d_wgrad(winograd)");
    return CPG_OK;
}
