/*
 * cpg_hip.h -- C ABI of libcpg_hip.so, the MI355X (gfx950) implementation of the
 * ivclab/CPG masked-CNN train / prune / retrain hot path.
 *
 * The reference has no FFI layer: its boundary is the Python class contract of
 * models/layers.py + utils/prune.py.  Every entry point below
 * replaces one stock-PyTorch call sequence of the reference and cites it; the
 * reference-side binding (a ctypes stub inside SharableConv2d / SharableLinear /
 * SparsePruner) is shown in INTEGRATION.md, and cpg_amd/_lib.py is that binding.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HBM) unless named *_host.  The caller (PyTorch)
 *     owns every buffer; kernels borrow them for the duration of the call.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued there and the call
 *     returns without synchronising.  No global mutable state: the library is re-entrant
 *     and may be called concurrently from several host threads on different streams.
 *   - Tensors are dense, contiguous fp32 (NCHW activations, [Cout][Cin/g][R][S] conv
 *     weights, [out][in] linear weights) and uint8 owner ids shaped like the weight
 *     (reference: torch.ByteTensor masks, CPG_cifar100_main_normal.py:204).
 *   - Element-wise entry points accept n == 0 (and then NULL data pointers, as an empty
 *     tensor has no storage) and return CPG_OK without launching anything.
 *   - `pm` is the real-valued piggymask (same shape as the weight) or NULL (task 1,
 *     CPG_cifar100_main_normal.py:263-270).  When non-NULL the effective weight is
 *     W * (pm > thr ? 1 : 0) -- models/layers.py:11-23,99-105.  The linear and generic conv kernels
 *     form it inside their LDS staging pass (never materialised); the 3x3 s1 p1 conv kernels stream it
 *     from a K-major packed copy produced by a fused binarise+mask+transpose pass into the call's
 *     workspace (conv weights are <= 9.4 MB per layer; the 411 MB linear weights are never copied).
 *   - Return value: 0 ok; <0 invalid argument / unsupported; CPG_E_KRANGE is rank-prune's
 *     "not enough weights" (reference: sys.exit(2), utils/prune.py:38-42);
 *     >= CPG_E_HIP_BASE is CPG_E_HIP_BASE + hipError_t.
 *   - Workspaces: query with the *_workspace_bytes function, pass any buffer at least
 *     that large (contents undefined on entry and exit).
 */
#ifndef CPG_HIP_H
#define CPG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPG_ABI_VERSION 3

#define CPG_OK 0
#define CPG_E_INVALID (-1)
#define CPG_E_UNSUPPORTED (-2)
#define CPG_E_WORKSPACE (-3)
#define CPG_E_KRANGE 2
#define CPG_E_HIP_BASE 1000

/* gradient-routing modes (args.mode of utils/prune.py:206-210) */
#define CPG_MODE_FINETUNE 0
#define CPG_MODE_PRUNE 1

/* geometry of one SharableConv2d call (models/layers.py:46-49,108-109) */
typedef struct cpg_conv_desc {
    int32_t N, C, H, W;       /* input  NCHW                                  */
    int32_t K;                /* out_channels                                 */
    int32_t R, S;             /* kernel_size                                  */
    int32_t stride_h, stride_w;
    int32_t pad_h, pad_w;
    int32_t dil_h, dil_w;
    int32_t groups;           /* only 1 is implemented (all CPG configs)      */
} cpg_conv_desc;

/* result record of cpg_rank_prune, written to DEVICE memory (one per call) */
typedef struct cpg_prune_result {
    int64_t n_candidates;     /* #(owner == cur || owner == 0)                */
    int64_t k;                /* round-half-even(ratio * n_candidates)        */
    int64_t n_released;       /* #slots whose owner went cur -> 0             */
    float cutoff;             /* k-th smallest |w| among the candidates       */
    int32_t status;           /* CPG_OK or CPG_E_KRANGE (owner left untouched)*/
} cpg_prune_result;

int cpg_version(void);
/* PROCESS-wide scheduling hint (an atomic; default 0): 1 = other streams' kernels share the chip with this process's launches
 * (a data-parallel run: RCCL's all-reduce kernels hold some CUs during the backward pass).  Only changes scheduling and the
 * summation order of split partial sums.  Since round 4 ONE planner reads it: the Winograd weight gradient runs 2 rounds of half-length
 * units per wave slot instead of 1 (a launch that finds CUs taken then grows by half a round instead of a whole one); the pointwise and
 * direct 3x3 weight-gradient planners ignore it (beside RCCL's real kernels their finer splits cost more than they saved).  It changes
 * the workspace a weight-gradient call needs, and it is PROCESS-wide (the planners run inside autograd's backward, on the engine's
 * worker thread, not on the thread that set it): set it once, before the first workspace query, and do not change it while ANY thread
 * is between a workspace query and the launch it sized -- a launch whose plan outgrew its workspace fails with CPG_E_WORKSPACE, it
 * never overruns.  cpg_amd.dist.DataParallel raises it only for >= 256 MB of gradients per step (VGG16).  Replaces nn.DataParallel's
 * implicit "all GPUs are mine" (CPG_cifar100_main_normal.py:199-200). */
int cpg_set_shared_chip_hint(int32_t shared);
int32_t cpg_get_shared_chip_hint(void);
/* Library options.  Every CPG_* switch of the dispatch code (INTEGRATION.md lists them) lives in one process-wide table that is
 * filled from the environment ONCE, when the library is loaded; no launch path calls getenv().  After loading, the table changes
 * only through these two calls.  `name` is the switch's documented name ("CPG_NO_WINO", "CPG_WW_UNITS", ...); `value`: boolean
 * switches 0 / 1, integer switches their number, CPG_WINO_KERNEL 0 = block, 1 = wave, 2 = pair, 3 = 64; CPG_OPT_UNSET = "never
 * given" (the built-in default applies).  The policy switches -- the only ones an integrator should touch -- choose the arithmetic
 * of the 3x3 convolutions: CPG_NO_WINO (direct fp32 MFMA instead of Winograd F(2x2,3x3) in all three passes), CPG_NO_WINO_WGRAD,
 * CPG_NO_WINO_ODD, CPG_NO_STEM, CPG_NO_STEM_FUSE, CPG_NO_DEAD_SKIP; everything else is A/B tooling.  Unknown name: CPG_E_INVALID.
 * There is no counterpart in the reference: F.conv2d takes whichever algorithm cuDNN / MIOpen selects (models/layers.py:106-109).
 *
 * WHICH SWITCHES CHANGE RESULTS.  Every path computes the same fp32 convolution / GEMM; what a switch can change is the rounding
 * (always inside the 1e-4 parity bar, every combination is covered by tests/test_hip_parity.py):
 *   arithmetic (Winograd transforms vs direct taps: differences of 1e-6 of the output scale):
 *       CPG_NO_WINO, CPG_NO_WINO_WGRAD, CPG_NO_WINO_ODD, CPG_DISABLE_CONV3X3, CPG_C3_FORCE (tile shape of the direct kernel = its
 *       channel-split accumulation), CPG_NO_S2, CPG_NO_V14, CPG_W3_PICK, CPG_DISABLE_CONV1X1, CPG_DISABLE_CONV1X1_WGRAD,
 *       CPG_DISABLE_PW_GEMM, CPG_NO_STEM (stem kernel vs general kernel: other k order of the 27 taps);
 *   summation order only (how a reduction over pixels / tiles / channel blocks is split and added: last-bit differences, mostly in
 *   weight gradients and BatchNorm statistics):
 *       CPG_WW_UNITS, CPG_WW_SHARE, CPG_C3W_BPC, CPG_PWW_BPC, CPG_PW_TILE, CPG_WINO_KERNEL, CPG_WINO_NW, CPG_STEM_BLOCKS, CPG_WINO_TAIL,
 *       CPG_FC_SMALL (linear layers at <= 64 rows: which kernel streams the weight = how the contraction is split),
 *       CPG_NO_STEM_FUSE (fused stem recomputes y instead of reading it back: same bits where the summation order is the same),
 *       and cpg_set_shared_chip_hint (above);
 *   no effect on any result bit (scheduling / mapping of the same work):
 *       CPG_WINO_PERSIST, CPG_WINO_GRIDS, CPG_WW_XCD, CPG_WG3_SHARE, CPG_NO_DEAD_SKIP (skips work whose inputs are exactly zero).
 * A run is bit-reproducible for a fixed set of switch values and a fixed hint; the defaults are what every committed number used.
 * The table is global state, but not per-call mutable state: no entry point writes it, and a concurrent cpg_set_option only ever
 * changes which of the above equivalent plans a LATER call picks (the workspace caveat of the hint applies to CPG_WW_UNITS,
 * CPG_C3W_BPC, CPG_PWW_BPC, CPG_PW_TILE, CPG_WINO_TAIL and CPG_FC_SMALL as well: a switch that sizes a workspace must not change
 * between the *_workspace_bytes query and the call it sized -- the Winograd tail falls back to one launch on a workspace that is too
 * small, the <= 64-row linear kernels return CPG_E_WORKSPACE). */
#define CPG_OPT_UNSET INT32_MIN
int cpg_set_option(const char *name, int32_t value);
int cpg_get_option(const char *name, int32_t *value);
/* human-readable text for the last non-zero status returned on THIS thread */
const char *cpg_last_error(void);

/* ---- K1: models/layers.py:11-23 + :103 / :190 (Binarizer + elementwise product) ----
 * w_eff[i] = w[i] * (pm[i] > thr ? 1 : pm[i] <= thr ? 0 : pm[i]); w == NULL gives the bare Binarizer
 * output.  Exported for callers that need W_eff itself (tests, checkpoint export); the conv/linear
 * kernels fuse this step. */
int cpg_binarize_mask_weight(const float *w, const float *pm, float thr, float *w_eff,
                             int64_t n, void *stream);

/* ---- K2/K3/K4: F.conv2d of models/layers.py:108-109 and its autograd ----
 * fwd   : y  = conv2d(x, W_eff, bias)                       bias may be NULL
 * dgrad : gx = conv2d_input_grad(gy, W_eff)
 * wgrad : gW_eff = conv2d_weight_grad(x, gy); then, as autograd of `bin(pm) * W`
 *         (models/layers.py:103): gw = gW_eff * bin(pm), gpm = gW_eff * W (gpm NULL when pm
 *         is NULL).  gb (may be NULL) = sum of gy over N,H,W.
 *         gw / gpm / gb are OVERWRITTEN (caller accumulates if it needs to). */
size_t cpg_conv2d_workspace_bytes(const cpg_conv_desc *d);
int cpg_conv2d_fwd(const cpg_conv_desc *d, const float *x, const float *w, const float *pm,
                   float thr, const float *bias, float *y, void *ws, size_t ws_bytes, void *stream);
int cpg_conv2d_dgrad(const cpg_conv_desc *d, const float *gy, const float *w, const float *pm,
                     float thr, float *gx, void *ws, size_t ws_bytes, void *stream);
/* Input gradient with the gradient of the input's OTHER consumer added in the epilogue: gx = dgrad(gy) + addend.  A residual block's
 * input feeds conv1 and the identity branch (models/resnet.py:84-104; models/spherenet.py:121-131: x + relu(conv(relu(conv(x))))); autograd
 * would sum the two gradients in a separate 3-pass kernel.  Dense 1x1 layers, and (round 5) the 3x3 s1 p1 layers whose input gradient
 * runs the two-wave Winograd kernel (>= 64 channels on both sides, or a 7 x 7 map): cpg_conv2d_dgrad_add_supported; addend is shaped
 * like gx and may not alias it. */
int32_t cpg_conv2d_dgrad_add_supported(const cpg_conv_desc *desc);
int cpg_conv2d_dgrad_add(const cpg_conv_desc *desc, const float *gy, const float *w, const float *piggymask, float threshold,
                         const float *addend, float *gx, void *workspace, size_t workspace_bytes, void *stream);
int cpg_conv2d_wgrad(const cpg_conv_desc *d, const float *x, const float *gy, const float *w,
                     const float *pm, float thr, float *gw, float *gpm, float *gb,
                     void *ws, size_t ws_bytes, void *stream);

/* Caller-owned packed weight operands (ABI version 3; SURVEY.md section 8b "an explicit, re-creatable cache (packed keep-bits / W_eff)").
 * The pointwise kernels and the one- / two-wave Winograd kernels stream W_eff = W * bin(pm) from a packed copy (K-major, resp. the
 * transformed filter in per-lane order) that every cpg_conv2d_fwd / _dgrad call otherwise produces into its own workspace first -- one
 * small launch per call, 109 per ResNet-50 step.  With these three calls the caller produces the operands ONCE per weight state -- the
 * forward's and the input gradient's of a layer in ONE launch -- and hands them to the calls that stream them:
 *   cpg_conv2d_pack_bytes(desc, pass): size of the operand of pass 0 (cpg_conv2d_fwd), 1 (cpg_conv2d_dgrad / _dgrad_add) or
 *       2 (cpg_conv2d_fwd_bnstats); 0 = a call of this shape streams no packed operand (it needs no cache).
 *   cpg_conv2d_pack(desc, w, pm, thr, pass_a, packed_a, bytes_a, pass_b, packed_b, bytes_b, stream): packs for one or (packed_b
 *       != NULL) two passes in one launch; bytes_* must equal cpg_conv2d_pack_bytes.  The buffers are the caller's; they are valid
 *       for as long as w, pm, thr, the shape in `desc` and the library options do not change (key them on the tensors' versions).
 *   cpg_conv2d_use_packed(packed, bytes): arms THE CALLING THREAD -- the next cpg_conv2d_fwd / _fwd_bnstats / _dgrad / _dgrad_add /
 *       _dgrad_bnbwd call on this thread streams `packed` instead of packing w / pm (which it still takes: a launch of another kernel
 *       family ignores the operand and packs for itself).  One-shot: that call disarms the thread whether it used the operand or
 *       not; a size that does not match what the launch streams is CPG_E_INVALID.  NULL disarms.
 * Results are bit-equal to the self-packing calls.  Reference: the `W_eff` the reference materialises in every forward
 * (models/layers.py:99-105) and autograd keeps for the backward. */
size_t cpg_conv2d_pack_bytes(const cpg_conv_desc *desc, int32_t pass);
int cpg_conv2d_pack(const cpg_conv_desc *desc, const float *w, const float *piggymask, float threshold, int32_t pass_a, void *packed_a,
                    size_t bytes_a, int32_t pass_b, void *packed_b, size_t bytes_b, void *stream);
int cpg_conv2d_use_packed(const void *packed, size_t bytes);

/* ---- K5: F.linear of models/layers.py:194 and its autograd ----
 * x [batch][in], w [out][in], y [batch][out]; same masking / gradient rules as conv. */
size_t cpg_linear_workspace_bytes(int32_t batch, int32_t in_features, int32_t out_features);
int cpg_linear_fwd(const float *x, const float *w, const float *pm, float thr, const float *bias,
                   float *y, int32_t batch, int32_t in_features, int32_t out_features,
                   void *ws, size_t ws_bytes, void *stream);
int cpg_linear_dgrad(const float *gy, const float *w, const float *pm, float thr, float *gx,
                     int32_t batch, int32_t in_features, int32_t out_features,
                     void *ws, size_t ws_bytes, void *stream);
int cpg_linear_wgrad(const float *x, const float *gy, const float *w, const float *pm, float thr,
                     float *gw, float *gpm, float *gb, int32_t batch, int32_t in_features,
                     int32_t out_features, void *ws, size_t ws_bytes, void *stream);

/* ---- K4e: utils/prune.py:195-211 (do_weight_decay_and_make_grads_zero), one layer ----
 * gw = (owner == cur) ? gw + wd * w : 0.   gpm (may be NULL): FINETUNE -> 0 where owner == 0 or
 * owner >= cur; PRUNE -> 0 everywhere.  One pass, in place. */
int cpg_route_grads(float *gw, const float *w, const uint8_t *owner, int32_t cur, float wd,
                    float *gpm, int32_t mode, int64_t n, void *stream);

/* ---- K6: utils/prune.py:30-53 (_pruning_mask), one layer, fully on device ----
 * candidates = owner in {cur, 0}; k = round-half-even(ratio * n_cand) in fp64 (python round());
 * cutoff = k-th smallest |w| over the candidates (radix select on the fp32 bit pattern);
 * owner[(|w| <= cutoff) & (owner == cur)] = 0.  k < 1 or k > n_cand -> status CPG_E_KRANGE in
 * *result and owner untouched.  `result` is a device pointer; nothing is synchronised. */
size_t cpg_rank_prune_workspace_bytes(void);
int cpg_rank_prune(const float *w, uint8_t *owner, int32_t cur, double ratio, int64_t n,
                   cpg_prune_result *result, void *ws, size_t ws_bytes, void *stream);

/* ---- K7: utils/prune.py:111-193 (the four mask statistics), one layer ----
 * hist[0..255] += #(owner == id); hist[256] += #(0 < owner < inference_idx && pm > 0.005f)
 * (only when pm != NULL).  `hist` = 257 uint64 in device memory, ACCUMULATED so one buffer can
 * collect all layers; the caller zeroes it. */
int cpg_mask_hist(const uint8_t *owner, const float *pm, int32_t inference_idx, int64_t n,
                  uint64_t *hist, void *stream);

/* ---- K8: utils/prune.py:213-243 ---- */
/* apply_mask (:223-231): w[owner == 0 || owner > inference_idx] = 0 */
int cpg_apply_mask(float *w, const uint8_t *owner, int32_t inference_idx, int64_t n, void *stream);
/* make_pruned_zero (:213-221): w[owner == 0] = 0 */
int cpg_zero_pruned(float *w, const uint8_t *owner, int64_t n, void *stream);
/* make_finetuning_mask (:233-243): owner[owner == 0] = new_idx */
int cpg_claim_free(uint8_t *owner, int32_t new_idx, int64_t n, void *stream);

/* ---- data-parallel payload (reference: nn.DataParallel reduces every gradient, CPG_cifar100_main_normal.py:199) ----
 * Gradient routing (K4e) zeroes every slot the current task does not own right after the all-reduce, so only the surviving
 * slots need to cross xGMI: select 0 = owner == cur (weight gradients), select 1 = 0 < owner < cur (piggymask gradients,
 * finetune mode).  cpg_owned_block_counts writes the number of selected slots of every 1024-element block
 * (cpg_owned_num_blocks(n) int32 values); the caller turns them into EXCLUSIVE prefix sums (int64, device) -- cacheable
 * until the owner mask mutates -- and cpg_pack_owned gathers g.flatten()[selected] (natural order) into `packed`,
 * cpg_unpack_owned scatters it back; unselected slots of g are left untouched.  Owner masks are replicated, so packed
 * buffers are element-aligned across ranks. */
int64_t cpg_owned_num_blocks(int64_t n);
int cpg_owned_block_counts(const uint8_t *owner, int32_t cur, int32_t select, int64_t n, int32_t *counts, void *stream);
int cpg_pack_owned(const float *g, const uint8_t *owner, int32_t cur, int32_t select, int64_t n, const int64_t *block_offsets,
                   float *packed, void *stream);
int cpg_unpack_owned(const float *packed, const uint8_t *owner, int32_t cur, int32_t select, int64_t n,
                     const int64_t *block_offsets, float *g, void *stream);

/* ---- fused masked SGD step, one layer ----
 * do_weight_decay_and_make_grads_zero (utils/prune.py:203-205) + torch.optim.SGD(lr, momentum, nesterov,
 * dampening 0, weight_decay 0) (CPG_cifar100_main_normal.py:339-340) in one pass over w / gw / momentum / owner:
 * g = owner == cur ? gw + wd*w : 0;  buf = first_step ? g : momentum*buf + g;  w -= lr * (nesterov ? g + momentum*buf : buf).
 * gw is overwritten with the routed gradient g (the state the reference leaves in .grad). */
int cpg_sgd_route_step(float *w, float *gw, float *momentum_buf, const uint8_t *owner, int32_t cur, float wd,
                       float lr, float momentum, int32_t nesterov, int32_t first_step, int64_t n, void *stream);

/* Fused piggymask step for task >= 2: the piggymask part of do_weight_decay_and_make_grads_zero (utils/prune.py:206-210:
 * finetune -> gradient zeroed where owner == 0 or owner >= cur, prune -> all zero) followed by torch.optim.Adam's update
 * (CPG_cifar100_main_normal.py:342-346; amsgrad = False, weight_decay = 0) in one pass; `step` is the 1-based Adam step
 * count, exp_avg / exp_avg_sq its state (zeros before step 1).  The routed gradient is written back to gpm.
 * The hyper-parameters are DOUBLES, as torch holds them (python floats): torch forms 1 - beta, lr / bias_correction1 and
 * sqrt(bias_correction2) in double and only then rounds to fp32 -- 1 - 0.999 is 0.001 there, but 0.00099998713 when formed from
 * the fp32 value 0.999f (ABI version 2; version 1 took floats and was 1.3e-5 off in exp_avg_sq). */
int cpg_adam_route_step(float *pm, float *gpm, float *exp_avg, float *exp_avg_sq, const uint8_t *owner, int32_t cur,
                        int32_t mode, double lr, double beta1, double beta2, double eps, int32_t step, int64_t n, void *stream);

/* The two fused optimizer passes over MANY layers in one launch (ABI version 3): ResNet-50 has 53 masked layers of 4 k - 2.4 M weights
 * each, and 53 launches of a few microseconds cost the queue more than the bytes they move.  `items_host` is a HOST array (the pointers
 * inside are device pointers, as everywhere else); the library passes them to the kernel by value, cpg_multi_tensor_max() layers per
 * launch, so nothing is copied to the device and the array may be freed or reused when the call returns.  All items share the scalar
 * arguments -- callers group parameters by (lr, momentum, nesterov, first_step) / (lr, betas, eps, step) as torch's own foreach path
 * does.  Element for element the arithmetic is cpg_sgd_route_step's / cpg_adam_route_step's: results are bit-equal to per-layer calls.
 * Replaces the same reference lines: utils/prune.py:195-211 + CPG_cifar100_main_normal.py:339-346. */
typedef struct cpg_sgd_item {
    float *w, *gw, *momentum_buf;
    const uint8_t *owner;
    int64_t n;
} cpg_sgd_item;
typedef struct cpg_adam_item {
    float *pm, *gpm, *exp_avg, *exp_avg_sq;
    const uint8_t *owner;
    int64_t n;
} cpg_adam_item;
int32_t cpg_multi_tensor_max(void);
int cpg_sgd_route_step_multi(const cpg_sgd_item *items_host, int32_t n_items, int32_t cur, float wd, float lr, float momentum,
                             int32_t nesterov, int32_t first_step, void *stream);
int cpg_adam_route_step_multi(const cpg_adam_item *items_host, int32_t n_items, int32_t cur, int32_t mode, double lr, double beta1,
                              double beta2, double eps, int32_t step, void *stream);

/* ---- nn.BatchNorm2d -> nn.ReLU(inplace) after each masked conv ----
 * (models/vgg.py:137-141: `layers += [conv2d, nn.BatchNorm2d(c), nn.ReLU(inplace=True)]`).
 * x, y, gy, gx: NCHW fp32 with HW = H*W; gamma/beta/mean/invstd/running_*: C floats.
 * fwd_train: batch statistics (biased variance for normalisation; running_var gets the unbiased one,
 *   running = (1 - momentum) * running + momentum * batch, as torch.nn.BatchNorm2d), writes mean and
 *   invstd = 1/sqrt(var + eps) for the backward, then y = [max(0,] (x-mean)*invstd*gamma + beta [)].
 *   running_mean/var may both be NULL.  Reductions are two-stage and deterministic.
 * fwd_eval : same affine map with caller-provided statistics.
 * bwd      : gradient of fwd_train (train != 0) or fwd_eval (train == 0); the ReLU mask is recomputed
 *   from x, so neither y nor a mask tensor is read.  dgamma / dbeta are overwritten. */
size_t cpg_bn_workspace_bytes(int32_t N, int32_t C, int32_t HW);
int cpg_bn_relu_fwd_train(const float *x, const float *gamma, const float *beta, float eps, float momentum,
                          float *running_mean, float *running_var, float *mean, float *invstd, float *y,
                          int32_t N, int32_t C, int32_t HW, int32_t relu, void *ws, size_t ws_bytes, void *stream);
int cpg_bn_relu_fwd_eval(const float *x, const float *gamma, const float *beta, const float *mean,
                         const float *invstd, float *y, int32_t N, int32_t C, int32_t HW, int32_t relu,
                         void *stream);
int cpg_bn_relu_bwd(const float *x, const float *gy, const float *gamma, const float *beta, const float *mean,
                    const float *invstd, float *gx, float *dgamma, float *dbeta, int32_t N, int32_t C,
                    int32_t HW, int32_t relu, int32_t train, void *ws, size_t ws_bytes, void *stream);

/* Which arithmetic a launch of this shape uses (pass: 0 forward, 1 input gradient, 2 weight gradient, 3 the inference entry
 * cpg_conv2d_fwd_bn_eval).
 * cpg_conv2d_fwd, cpg_conv2d_fwd_bnstats (0) and cpg_conv2d_dgrad (1) run 3x3 / stride 1 / pad 1 layers on even-sized maps with >= 16 channels on both sides (a multiple
 * of 4 on the contracted side) by Winograd F(2x2, 3x3): 16 instead of 36 multiplies per 2x2 output tile and channel pair, the
 * same sums in a different association -- 1-4e-6 of the output scale from fp64 where the direct kernels are at 0.5-1e-6; the
 * reference's own F.conv2d (models/layers.py:106-109) takes whichever algorithm cuDNN / MIOpen picks, Winograd included.
 * Returns 1 for such a launch, 0 for the direct / generic kernels.  Setting CPG_NO_WINO in the environment (read per call)
 * forces 0 everywhere.  cpg_conv2d_wgrad (2) uses the adjoint transform for maps that are 14 or a multiple of 28 pixels wide with channel counts that are
 * multiples of 32 (CPG_NO_WINO_WGRAD forces the direct kernels for this pass only).  cpg_conv2d_fwd_bn_eval (3) uses it on the
 * same shapes as the forward pass, with or without a conv bias and the skip statistics (the Winograd kernel's epilogue applies
 * the same expression and honours the same liveness words); the bf16 entry points never do. */
int32_t cpg_conv2d_winograd(const cpg_conv_desc *desc, int32_t pass);

/* Conv forward fused with the statistics pass of the BatchNorm2d that follows it in every CPG topology
 * (models/vgg.py:137-141 `conv2d, BatchNorm2d, ReLU`): the kernel that produces y also writes, per output channel and
 * pixel tile, {sum y, sum y^2} -> stats[K][tiles][2] (fp32), tiles = cpg_conv2d_bnstats_tiles(desc) (0: this shape has
 * no fused path -- use cpg_conv2d_fwd + cpg_bn_relu_fwd_train).  cpg_bn_stats_finalize merges them (fp64, fixed order)
 * into mean / invstd and the running statistics; the normalisation itself is then cpg_bn_relu_fwd_eval or
 * cpg_bn_relu_pool_fwd(train = 0) with those statistics, and backward is unchanged (train = 1). */
int32_t cpg_conv2d_bnstats_tiles(const cpg_conv_desc *desc);
int cpg_conv2d_fwd_bnstats(const cpg_conv_desc *desc, const float *x, const float *w, const float *piggymask, float threshold,
                           const float *bias, float *y, float *stats, size_t stats_bytes, void *workspace,
                           size_t workspace_bytes, void *stream);
int cpg_bn_stats_finalize(const float *stats, int32_t tiles, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                          float *running_mean, float *running_var, float *mean, float *invstd, void *stream);
/* ... and (ABI version 3) nn.BatchNorm2d's `num_batches_tracked += 1` in the same launch (torch.nn.modules.batchnorm: the counter every
 * training-mode forward bumps; one `add<long>` launch per layer in stock torch -- 53 per ResNet-50 step).  num_batches_tracked: one int64
 * in device memory, may be NULL. */
int cpg_bn_stats_finalize_count(const float *stats, int32_t tiles, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                float *running_mean, float *running_var, float *mean, float *invstd, int64_t *num_batches_tracked,
                                void *stream);

/* Inference (Manager.validate, utils/manager.py:103-121: apply_mask, then model.eval() forward): conv -> BatchNorm2d in
 * eval mode (-> ReLU) of models/vgg.py:137-141 as ONE kernel -- y = [max(0,] (conv(x, W_eff) + bias - running_mean) /
 * sqrt(running_var + eps) * gamma + beta [)] applied in the conv epilogue, so the raw conv output is never written and no
 * BatchNorm kernel runs.  cpg_conv2d_fwd_bn_eval_supported(desc) == 0: shape has no fused path (use cpg_conv2d_fwd +
 * cpg_bn_relu_fwd_eval).  No gradient counterpart: callers use it under torch.no_grad() only.
 * Dead channels are skipped: after apply_mask (utils/prune.py:223-231) every slot with owner 0 or > the inference task is
 * zero, so a network that was GROWN for later tasks (CPG_cifar100_main_normal.py:208-232) carries whole output channels and
 * trailing input channels that are zero for an earlier task.  The weight-pack pass records per-channel liveness (wave
 * ballots), a block whose output channels are all dead skips its MFMA loop (conv output exactly 0 -> it writes BatchNorm(bias))
 * and every block stops after the last live input channel: serving an old task from the resident wide model costs what
 * the reference's cropped model (CPG_cifar100_main_normal.py:233-249) costs.  skip_stats (device, may be NULL) receives
 * {1 + last live input channel, number of output tiles skipped}. */
int32_t cpg_conv2d_fwd_bn_eval_supported(const cpg_conv_desc *desc);
int cpg_conv2d_fwd_bn_eval(const cpg_conv_desc *desc, const float *x, const float *w, const float *piggymask, float threshold,
                           const float *bias, const float *gamma, const float *beta, const float *running_mean,
                           const float *running_var, float eps, int32_t relu, float *y, int32_t *skip_stats, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ---- OPT-IN bf16 MFMA path (north_star: "MFMA fp32/bf16 GEMM") of the 3x3 / stride 1 / pad 1 convolution ----
 * Same contraction as cpg_conv2d_fwd / cpg_conv2d_dgrad with the operands (activations, effective weights W * bin(pm))
 * rounded to bf16 on their way into LDS, exact products, fp32 accumulation (v_mfma_f32_32x32x16_bf16) -- the arithmetic of
 * an autocast(bf16) convolution.  All tensors in HBM stay fp32 NCHW.  NOT the default: north_star's 1e-4 parity bar is an
 * fp32 bar; this path has its own tolerance (about 1e-2 of the output scale) and its own roofline (2.5 PFLOP/s).
 * cpg_conv2d_wgrad_bf16: gW_eff from bf16-rounded x and gy, then the same autograd epilogue as cpg_conv2d_wgrad; no bias
 * gradient -- layers with a bias and the 3 -> 64 stem (< 16 channels) use cpg_conv2d_wgrad (fp32).
 * The *_bf16x3 entry points run the same kernels with every operand split into two bf16 terms, v = hi + lo (hi = bf16(v),
 * lo = bf16(v - hi)), and accumulate a_hi*b_hi + a_hi*b_lo + a_lo*b_hi in fp32: ~16 mantissa bits per product instead of 8
 * (measured 5e-6 of the output scale per layer against 2.5e-3 for plain bf16 and 2.5e-7 for fp32 MFMA) at 3/16 of the fp32
 * MFMA cost.  Also opt-in: it meets north_star's 1e-4 logit bar, but it is not the reference's fp32 arithmetic. */
int32_t cpg_conv2d_bf16_supported(const cpg_conv_desc *desc);
int32_t cpg_conv2d_wgrad_bf16_supported(const cpg_conv_desc *desc);
size_t cpg_conv2d_wgrad_bf16_workspace_bytes(const cpg_conv_desc *desc);
int cpg_conv2d_wgrad_bf16(const cpg_conv_desc *desc, const float *x, const float *gy, const float *w, const float *piggymask,
                          float threshold, float *gw, float *gpm, void *workspace, size_t workspace_bytes, void *stream);
int cpg_conv2d_fwd_bf16x3(const cpg_conv_desc *desc, const float *x, const float *w, const float *piggymask, float threshold,
                          const float *bias, float *y, void *workspace, size_t workspace_bytes, void *stream);
int cpg_conv2d_dgrad_bf16x3(const cpg_conv_desc *desc, const float *gy, const float *w, const float *piggymask, float threshold,
                            float *gx, void *workspace, size_t workspace_bytes, void *stream);
int cpg_conv2d_wgrad_bf16x3(const cpg_conv_desc *desc, const float *x, const float *gy, const float *w, const float *piggymask,
                            float threshold, float *gw, float *gpm, void *workspace, size_t workspace_bytes, void *stream);
size_t cpg_conv2d_bf16_workspace_bytes(const cpg_conv_desc *desc);
int cpg_conv2d_fwd_bf16(const cpg_conv_desc *desc, const float *x, const float *w, const float *piggymask, float threshold,
                        const float *bias, float *y, void *workspace, size_t workspace_bytes, void *stream);
int cpg_conv2d_dgrad_bf16(const cpg_conv_desc *desc, const float *gy, const float *w, const float *piggymask, float threshold,
                          float *gx, void *workspace, size_t workspace_bytes, void *stream);

/* The BatchNorm backward reduction riding in the NEXT layer's input-gradient kernel (the backward counterpart of
 * cpg_conv2d_fwd_bnstats): conv_{L+1}'s dgrad produces g = dL/da with a = relu(bn_L(ypre)); its epilogue reads ypre at the
 * same positions, stores gm = g * [bn_L(ypre) > 0] instead of g and writes {sum gm, sum gm * xhat} per (channel, pixel tile) ->
 * partials[C][tiles][2] (fp32), tiles = cpg_conv2d_dgrad_bnbwd_tiles(desc) (0: no fused path for this shape).
 * cpg_bn_bwd_from_partials merges them (fp64, fixed order) into dgamma / dbeta and runs the one remaining pass
 * dx = (gm - mean(gm) - xhat * mean(gm * xhat)) * invstd * gamma.  Saves the 2-read reduction pass of cpg_bn_relu_bwd. */
int32_t cpg_conv2d_dgrad_bnbwd_tiles(const cpg_conv_desc *desc);
int cpg_conv2d_dgrad_bnbwd(const cpg_conv_desc *desc, const float *gy, const float *w, const float *piggymask, float threshold,
                           const float *ypre, const float *gamma, const float *beta, const float *mean, const float *invstd,
                           float *gm, float *partials, size_t partial_bytes, void *workspace, size_t workspace_bytes, void *stream);
int cpg_bn_bwd_from_partials(const float *partials, int32_t tiles, const float *x, const float *gm, const float *gamma,
                             const float *beta, const float *mean, const float *invstd, float *gx, float *dgamma, float *dbeta,
                             int32_t N, int32_t C, int32_t HW, void *workspace, size_t workspace_bytes, void *stream);

/* The merge step of cpg_bn_bwd_from_partials alone: partials[C][tiles][2] = {sum gm, sum gm * xhat} -> dbeta, dgamma and
 * coef[2c] = mean(gm), coef[2c + 1] = mean(gm * xhat) over N * HW elements (fp64, fixed order). */
int cpg_bn_bwd_finalize_partials(const float *partials, int32_t tiles, int32_t N, int32_t C, int32_t HW, float *dgamma, float *dbeta,
                                 float *coef, void *stream);

/* The network stem fused with the training-mode BatchNorm2d -> ReLU behind it (VGG16 features.0-2, models/vgg.py:137-141 `conv2d,
 * nn.BatchNorm2d(v), nn.ReLU(inplace=True)`; SharableConv2d.forward, models/layers.py:98-109): a 3x3 s1 p1 conv from <= 3 channels to
 * 64 whose output y is NEVER written -- every pass recomputes it from the image (27 multiply-adds per output against 4 bytes of
 * HBM traffic per pass).  No conv bias.  Forward: cpg_stem_bn_stats (partial sums stats[64][tiles][2] of y) ->
 * cpg_bn_stats_finalize -> cpg_stem_bn_relu_fwd (z = relu(bn(y))).  Backward: cpg_stem_bn_relu_bwd_reduce (partials[64][tiles][2]
 * = {sum gm, sum gm * xhat}, gm = gz * [z > 0]) -> cpg_bn_bwd_finalize_partials -> cpg_stem_bn_relu_bwd_wgrad (the weight gradient; or
 * cpg_stem_bn_relu_bwd_apply for gy, the gradient w.r.t. the conv output, and cpg_conv2d_wgrad(x, gy)).  The image itself gets no gradient through this path. */
int32_t cpg_stem_bn_supported(const cpg_conv_desc *desc);
int32_t cpg_stem_bn_tiles(const cpg_conv_desc *desc);
int cpg_stem_bn_stats(const cpg_conv_desc *desc, const float *x, const float *w, const float *piggymask, float threshold,
                      const float *bias, float *stats, size_t stats_bytes, void *stream);
int cpg_stem_bn_relu_fwd(const cpg_conv_desc *desc, const float *x, const float *w, const float *piggymask, float threshold,
                         const float *bias, const float *gamma, const float *beta, const float *mean, const float *invstd, float *z,
                         void *stream);
int cpg_stem_bn_relu_bwd_reduce(const cpg_conv_desc *desc, const float *x, const float *w, const float *piggymask, float threshold,
                                const float *bias, const float *gamma, const float *beta, const float *mean, const float *invstd,
                                const float *gz, float *partials, size_t partials_bytes, void *stream);
int cpg_stem_bn_relu_bwd_apply(const cpg_conv_desc *desc, const float *x, const float *w, const float *piggymask, float threshold,
                               const float *bias, const float *gamma, const float *beta, const float *mean, const float *invstd,
                               const float *coef, const float *gz, float *gy, void *stream);
/* ... or, instead of cpg_stem_bn_relu_bwd_apply + cpg_conv2d_wgrad: gy contracted with the image patch inside the same pass (gy is
 * never written): gW = g * bin(pm), gPM = g * W as cpg_conv2d_wgrad.  workspace: cpg_stem_bn_wgrad_workspace(desc) bytes. */
size_t cpg_stem_bn_wgrad_workspace(const cpg_conv_desc *desc);
int cpg_stem_bn_relu_bwd_wgrad(const cpg_conv_desc *desc, const float *x, const float *w, const float *piggymask, float threshold,
                               const float *bias, const float *gamma, const float *beta, const float *mean, const float *invstd,
                               const float *coef, const float *gz, float *gw, float *gpm, void *workspace, size_t workspace_bytes,
                               void *stream);

/* y = relu(bn(x) + res): the tail of a residual block (models/resnet.py:69-74 `out = self.bn3(out); out += identity;
 * out = self.relu(out)`).  train != 0: batch statistics (mean / invstd out, running stats updated); train == 0: `mean`
 * / `invstd` are inputs.  relu_mask (may be NULL; cpg_bn_add_relu_mask_bytes() bytes, 0 = no mask for this shape: 4 must divide HW)
 * receives the ReLU mask of y, one byte per four consecutive elements (bit j: element 4 i + j > 0), for cpg_bn_add_relu_bwd. */
size_t cpg_bn_add_relu_mask_bytes(int32_t N, int32_t C, int32_t HW);
int cpg_bn_add_relu_fwd(const float *x, const float *res, const float *gamma, const float *beta, float eps,
                        float momentum, float *running_mean, float *running_var, float *mean, float *invstd, float *y,
                        int32_t N, int32_t C, int32_t HW, int32_t train, void *ws, size_t ws_bytes, void *stream, uint8_t *relu_mask);
/* Backward of cpg_bn_add_relu_fwd (out = relu(bn(x) + residual), the tail of a residual block, models/resnet.py:62-71,96-104 --
 * stock torch: threshold_backward, then the BatchNorm backward, 8 passes): gz = gy * [out > 0] (the residual branch's gradient; must
 * not alias gy) is written by the reduction pass that also needs it, gx is the BatchNorm input gradient.  7 passes -- 6 1/16 with
 * relu_mask (the forward's byte mask; `out` is then not read and may be NULL), 6 % of a ResNet-50 step's BatchNorm traffic. */
int cpg_bn_add_relu_bwd(const float *x, const float *out, const float *gy, const float *gamma, const float *beta,
                        const float *mean, const float *invstd, float *gx, float *gz, float *dgamma, float *dbeta, int32_t N,
                        int32_t C, int32_t HW, int32_t train, void *ws, size_t ws_bytes, void *stream, const uint8_t *relu_mask);

/* BatchNorm2d -> ReLU -> MaxPool2d(2, 2) fused (the 5 VGG blocks that end in 'M', models/vgg.py:131-141).
 * y_pooled / g_pooled: [N][C][H/2][W/2]; H and W even.  train != 0: batch statistics are computed (and
 * running stats updated) first; train == 0: `mean` / `invstd` are inputs.  The un-pooled activation is never
 * written: backward recomputes each window from x and routes the pooled gradient to its first maximum. */
int cpg_bn_relu_pool_fwd(const float *x, const float *gamma, const float *beta, float eps, float momentum,
                         float *running_mean, float *running_var, float *mean, float *invstd, float *y_pooled,
                         int32_t N, int32_t C, int32_t H, int32_t W, int32_t train, void *ws, size_t ws_bytes,
                         void *stream);
int cpg_bn_relu_pool_bwd(const float *x, const float *g_pooled, const float *gamma, const float *beta,
                         const float *mean, const float *invstd, float *gx, float *dgamma, float *dbeta,
                         int32_t N, int32_t C, int32_t H, int32_t W, int32_t train, void *ws, size_t ws_bytes,
                         void *stream);

/* BatchNorm2d -> ReLU -> MaxPool2d(3, stride 2, padding 1): the ResNet stem's tail (models/resnet.py:127-129,208-211 -- stock torch:
 * BatchNorm apply, max_pool2d with an int64 index tensor, max_pool2d backward, BatchNorm backward).  mean / invstd are given
 * (cpg_bn_stats_finalize after cpg_conv2d_fwd_bnstats in training, or the running statistics); the un-pooled activation is never
 * written; backward recomputes it plane by plane in LDS, routes the pooled gradient to each window's first maximum (torch's rule)
 * and applies the BatchNorm gradient (train != 0: batch statistics).  cpg_bn_relu_pool3_supported: rows of at most 512 pixels. */
int32_t cpg_bn_relu_pool3_supported(int32_t H, int32_t W);
int cpg_bn_relu_pool3_fwd(const float *x, const float *gamma, const float *beta, const float *mean, const float *invstd,
                          float *y_pooled, int32_t N, int32_t C, int32_t H, int32_t W, void *stream);
int cpg_bn_relu_pool3_bwd(const float *x, const float *g_pooled, const float *gamma, const float *beta, const float *mean,
                          const float *invstd, float *gx, float *dgamma, float *dbeta, int32_t N, int32_t C, int32_t H,
                          int32_t W, int32_t train, void *workspace, size_t workspace_bytes, void *stream);

/* Backward of nn.PReLU on an NCHW tensor (SphereNet-20's activation after every masked conv,
 * models/spherenet.py:126-166 `relu{stage}_{i} = nn.PReLU(channels)`):
 *   gx = x > 0 ? gy : slope[c] * gy,   gslope[c] = sum_{n,hw} (x > 0 ? 0 : gy * x)
 * in ONE pass over x and gy (deterministic two-stage slope reduction).  n_slopes = C (per-channel) or 1. */
size_t cpg_prelu_workspace_bytes(int32_t N, int32_t C, int32_t HW);
int cpg_prelu_bwd(const float *x, const float *gy, const float *slope, float *gx, float *gslope, int32_t N,
                  int32_t C, int32_t HW, int32_t n_slopes, void *ws, size_t ws_bytes, void *stream);
/* ... and, when x is the output of a conv WITH bias that only this PReLU consumes (every SphereNet conv, models/spherenet.py:203-217),
 * also gbias[c] = sum over (n, pixels) of gx -- the conv's bias gradient, so that cpg_conv2d_wgrad is called with gb = NULL and no
 * separate reduction pass over gx runs. */
int cpg_prelu_bwd_bias(const float *x, const float *gy, const float *slope, float *gx, float *gslope, float *gbias, int32_t N,
                       int32_t C, int32_t HW, int32_t n_slopes, void *workspace, size_t workspace_bytes, void *stream);
/* PReLU forward with the residual add of a SphereNet unit folded in (models/spherenet.py:219-247: x + relu(conv(.))):
 * y = residual + (x > 0 ? x : slope[c] x); residual may be NULL.  One pass instead of torch's prelu + add kernels. */
int cpg_prelu_fwd(const float *x, const float *residual, const float *slope, float *y, int32_t N, int32_t C, int32_t HW,
                  int32_t n_slopes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CPG_HIP_H */
