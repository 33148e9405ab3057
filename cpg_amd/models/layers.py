"""Masked operators of CPG on MI355X: drop-in counterparts of the reference's models/layers.py.

    Binarizer        models/layers.py:11-23
    SharableConv2d   models/layers.py:43-145
    SharableLinear   models/layers.py:147-218

Same constructor signatures, attributes (.weight/.bias/.piggymask/.info/...) and forward
semantics; the arithmetic is libcpg_hip.so's: the effective weight W * bin(piggymask) is formed
inside the conv / GEMM kernels' LDS staging pass (never materialised), the contraction runs on
fp32 MFMA, and backward returns gW = gW_eff * bin(pm), gPM = gW_eff * W exactly as autograd of
the reference's `mask_thresholded * self.weight` does (models/layers.py:103,190).

Ternarizer (models/layers.py:25-40) is dead code in the reference (pre-0.4 autograd API, never
selected by any script); asking for it raises NotImplementedError here.
"""
import ctypes
import os

import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair
from torch.nn.parameter import Parameter

from .. import _lib

DEFAULT_THRESHOLD = 5e-3      # models/layers.py:9

# Arithmetic of the masked 3x3 convolutions: 'fp32' (default; fp32 MFMA, the reference's precision and north_star's parity
# bar), or OPT-IN: 'bf16' (operands rounded to bf16 on their way into LDS, fp32 accumulation, on v_mfma_f32_32x32x16_bf16)
# and 'bf16x3' (each operand split into two bf16 terms, three MFMAs per product: ~16 mantissa bits, 5e-6 of the output scale
# per layer).  A layer's own `.math` attribute, when set, wins.
CONV_MATH = 'fp32'


def set_conv_math(math):
    """Select the arithmetic of every SharableConv2d that has no `.math` of its own: 'fp32' or 'bf16' (opt-in)."""
    global CONV_MATH
    if math not in ('fp32', 'bf16', 'bf16x3'):
        raise ValueError("conv math must be 'fp32', 'bf16' or 'bf16x3', got %r" % (math,))
    CONV_MATH = math


class Binarizer(torch.autograd.Function):
    """{0,1} hard threshold with straight-through gradient (models/layers.py:11-23)."""

    @staticmethod
    def forward(ctx, inputs, threshold):
        out = torch.empty_like(inputs, memory_format=torch.contiguous_format)
        src = inputs.contiguous()
        rc = _lib.lib().cpg_binarize_mask_weight(None, _lib.dptr(src, name='inputs'), float(threshold),
                                                 _lib.dptr(out), src.numel(), _lib.stream_ptr())
        _lib.check('cpg_binarize_mask_weight', rc)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out, None


def _conv_desc(x_shape, w_shape, stride, padding, dilation, groups):
    d = _lib.ConvDesc()
    d.N, d.C, d.H, d.W = [int(v) for v in x_shape]
    d.K, d.R, d.S = int(w_shape[0]), int(w_shape[2]), int(w_shape[3])
    d.stride_h, d.stride_w = stride
    d.pad_h, d.pad_w = padding
    d.dil_h, d.dil_w = dilation
    d.groups = groups
    return d


def _out_hw(d):
    oh = (d.H + 2 * d.pad_h - d.dil_h * (d.R - 1) - 1) // d.stride_h + 1
    ow = (d.W + 2 * d.pad_w - d.dil_w * (d.S - 1) - 1) // d.stride_w + 1
    return oh, ow


# CPG_PACK_CACHE=0: every conv call packs its own weight operand (the behaviour up to round 5; A/B switch)
PACK_CACHE = os.environ.get('CPG_PACK_CACHE', '1') not in ('0', '')
_PACK_BYTES = {}


def _pack_bytes(L, d, which):
    """cpg_conv2d_pack_bytes per (shape, pass), memoised: the answer depends on the shape and the library options only
    (cpg_amd._lib.set_option clears the table)."""
    key = (d.N, d.C, d.H, d.W, d.K, d.R, d.S, d.stride_h, d.stride_w, d.pad_h, d.pad_w, d.dil_h, d.dil_w, d.groups, which)
    v = _PACK_BYTES.get(key)
    if v is None:
        v = _PACK_BYTES[key] = int(L.cpg_conv2d_pack_bytes(ctypes.byref(d), which))
    return v


def _with_packed(L, pk, call):
    """Run `call()` -- ONE conv entry point -- with the calling thread armed to stream the packed operand `pk` (None: plain call).
    The library disarms itself inside that entry point; an exception raised on the way there (an argument that does not convert: a CPU
    or non-fp32 tensor) disarms here, so that no LATER call of this thread can pick up an operand that was not meant for it."""
    if pk is None:
        return call()
    _lib.check('cpg_conv2d_use_packed', L.cpg_conv2d_use_packed(_lib.dptr(pk), pk.numel() * 4))
    try:
        return call()
    except BaseException:
        L.cpg_conv2d_use_packed(None, 0)
        raise


class BiasGradSink(object):
    """Hand-off from the PReLU that is the ONLY consumer of a biased conv's output (SphereNet, models/spherenet.py:203-247) to that
    conv's backward: the PReLU's one-pass backward sums its input gradient per channel while it writes it -- that sum IS the conv's bias
    gradient --, so the conv asks its weight-gradient call for no bias gradient (no extra reduction pass over gy).  One sink per forward."""
    __slots__ = ('gb',)

    def __init__(self):
        self.gb = None


class _MaskedConv2dFn(torch.autograd.Function):
    """y = conv2d(x, W * bin(pm), b) and its gradients, all through the C ABI."""

    @staticmethod
    def forward(ctx, x, weight, pm, bias, thr, stride, padding, dilation, groups, bn_stats=False, math='fp32', bn_hint=None, bias_sink=None):
        # bn_hint (fused_bn.BnBwdHint or None): `x` is relu(bn(ypre)) of the layer below and nothing else consumes it -- the
        # input-gradient kernel may then do that BatchNorm's backward reduction in its epilogue (cpg_conv2d_dgrad_bnbwd)
        ctx.bn_hint = bn_hint
        ctx.set_materialize_grads(False)         # (no zero-filled "gradient" tensor for the statistics output on every backward)
        ctx.bias_sink = bias_sink if bias is not None else None     # BiasGradSink: the activation behind this conv delivers the bias gradient
        """bn_stats: also return the per-(channel, pixel tile) {sum, sum of squares} of y that the kernel accumulates
        in its epilogue (cpg_conv2d_fwd_bnstats) -- a second, non-differentiable output, or None when the shape has
        no fused-statistics kernel."""
        if x.dim() != 4 or x.shape[1] != weight.shape[1] * groups:
            raise RuntimeError('SharableConv2d: input %s does not match weight %s (groups=%d)'
                               % (tuple(x.shape), tuple(weight.shape), groups))
        x = x.contiguous()
        w = weight.contiguous()
        p = None if pm is None else pm.contiguous()
        d = _conv_desc(x.shape, w.shape, stride, padding, dilation, groups)
        oh, ow = _out_hw(d)
        if oh <= 0 or ow <= 0:
            raise RuntimeError('SharableConv2d: kernel larger than padded input')
        y = torch.empty((d.N, d.K, oh, ow), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ctx.empty = d.N == 0 or d.K == 0
        ctx.bf16 = ctx.x3 = False
        if ctx.empty:                   # an empty batch (or no output channels) is legal for F.conv2d: empty output, zero parameter gradients
            _lib.dptr(x, name='input'), _lib.dptr(w, name='weight')            # still no CPU / dtype fallback
            ctx.save_for_backward(x, w, p)
            ctx.desc, ctx.thr, ctx.has_bias = d, float(thr), bias is not None
            if not bn_stats:
                return y
            stats = torch.empty(0, dtype=torch.float32, device=x.device)
            ctx.mark_non_differentiable(stats)
            return y, stats
        ctx.bf16 = math in ('bf16', 'bf16x3') and bool(L.cpg_conv2d_bf16_supported(ctypes.byref(d)))
        ctx.x3 = math == 'bf16x3'
        if ctx.bf16:
            # opt-in bf16 MFMA forward (no fused BatchNorm statistics on this path: the BatchNorm runs its own pass)
            ws, nbytes = _lib.workspace(L.cpg_conv2d_bf16_workspace_bytes(ctypes.byref(d)), x.device)
            fwd = L.cpg_conv2d_fwd_bf16x3 if ctx.x3 else L.cpg_conv2d_fwd_bf16
            rc = fwd(ctypes.byref(d), _lib.dptr(x, name='input'), _lib.dptr(w, name='weight'),
                     _lib.dptr(p, name='piggymask'), float(thr), _lib.dptr(bias, name='bias'), _lib.dptr(y),
                     _lib.dptr(ws), nbytes, _lib.stream_ptr())
            _lib.check('cpg_conv2d_fwd_bf16', rc)
            ctx.save_for_backward(x, w, p)
            ctx.desc, ctx.thr, ctx.has_bias = d, float(thr), bias is not None
            if not bn_stats:
                return y
            stats = torch.empty(0, dtype=torch.float32, device=x.device)
            ctx.mark_non_differentiable(stats)
            return y, stats
        ws, nbytes = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), x.device)
        tiles = L.cpg_conv2d_bnstats_tiles(ctypes.byref(d)) if bn_stats else 0
        stats = None
        # Packed weight operands (cpg_conv2d_pack, ABI 3): when this layer's forward AND its input gradient stream one, both are produced
        # here in ONE launch; the forward takes its own now, the input gradient's rides in ctx until backward (the weights cannot change in
        # between: autograd's version check on the saved w / pm guards exactly that).  One launch instead of two per layer and step.
        ctx.packed_dgrad = pk_f = None
        if PACK_CACHE and ctx.needs_input_grad[0]:
            uses_bnbwd = bn_hint is not None and L.cpg_conv2d_dgrad_bnbwd_tiles(ctypes.byref(d)) > 0      # (that launch packs for the direct kernels)
            nb_f, nb_d = _pack_bytes(L, d, 2 if tiles > 0 else 0), (0 if uses_bnbwd else _pack_bytes(L, d, 1))
            if nb_f and nb_d:
                pk_f = torch.empty(nb_f // 4, dtype=torch.float32, device=x.device)
                pk_d = torch.empty(nb_d // 4, dtype=torch.float32, device=x.device)
                rc = L.cpg_conv2d_pack(ctypes.byref(d), _lib.dptr(w, name='weight'), _lib.dptr(p, name='piggymask'), float(thr),
                                       2 if tiles > 0 else 0, _lib.dptr(pk_f), nb_f, 1, _lib.dptr(pk_d), nb_d, _lib.stream_ptr())
                _lib.check('cpg_conv2d_pack', rc)
                ctx.packed_dgrad = pk_d
        if tiles > 0:
            stats = torch.empty((d.K, tiles, 2), dtype=torch.float32, device=x.device)
            rc = _with_packed(L, pk_f, lambda: L.cpg_conv2d_fwd_bnstats(
                ctypes.byref(d), _lib.dptr(x, name='input'), _lib.dptr(w, name='weight'), _lib.dptr(p, name='piggymask'), float(thr),
                _lib.dptr(bias, name='bias'), _lib.dptr(y), _lib.dptr(stats), stats.numel() * 4, _lib.dptr(ws), nbytes, _lib.stream_ptr()))
            _lib.check('cpg_conv2d_fwd_bnstats', rc)
        else:
            rc = _with_packed(L, pk_f, lambda: L.cpg_conv2d_fwd(
                ctypes.byref(d), _lib.dptr(x, name='input'), _lib.dptr(w, name='weight'), _lib.dptr(p, name='piggymask'), float(thr),
                _lib.dptr(bias, name='bias'), _lib.dptr(y), _lib.dptr(ws), nbytes, _lib.stream_ptr()))
            _lib.check('cpg_conv2d_fwd', rc)
        ctx.save_for_backward(x, w, p)
        ctx.desc, ctx.thr, ctx.has_bias = d, float(thr), bias is not None
        if not bn_stats:
            return y
        if stats is None:
            stats = torch.empty(0, dtype=torch.float32, device=x.device)      # "not available" marker
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, gy, _gstats=None, addend=None):
        """addend (only from _MaskedConv2dSkipFn): a tensor shaped like the input gradient that is added to it -- in the kernel's
        epilogue where the shape class has that (cpg_conv2d_dgrad_add), by one add otherwise."""
        if gy is None:                           # the conv's output is unused: only a skip gradient (if any) flows
            return (addend,) + (None,) * 12
        x, w, p = ctx.saved_tensors
        d, thr = ctx.desc, ctx.thr
        if ctx.empty:
            # (an empty OUTPUT with a non-empty input -- out_channels == 0 -- still passes a residual branch's gradient through)
            return (addend if addend is not None else torch.zeros_like(x), torch.zeros_like(w), None if p is None else torch.zeros_like(p),
                    torch.zeros(d.K, dtype=torch.float32, device=x.device) if ctx.has_bias else None, None, None, None, None, None, None,
                    None, None, None)
        gy = gy.contiguous()
        L = _lib.lib()
        s = _lib.stream_ptr()
        gx = gw = gpm = gb = None
        ws, nbytes = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), x.device)
        if ctx.needs_input_grad[0] and ctx.bf16:
            gx = torch.empty_like(x)
            ws16, nb16 = _lib.workspace(L.cpg_conv2d_bf16_workspace_bytes(ctypes.byref(d)), x.device)
            dgrad = L.cpg_conv2d_dgrad_bf16x3 if ctx.x3 else L.cpg_conv2d_dgrad_bf16
            rc = dgrad(ctypes.byref(d), _lib.dptr(gy, name='grad_output'), _lib.dptr(w), _lib.dptr(p), thr,
                       _lib.dptr(gx), _lib.dptr(ws16), nb16, s)
            _lib.check('cpg_conv2d_dgrad_bf16', rc)
        elif ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            hint, ctx.bn_hint = ctx.bn_hint, None
            tiles = L.cpg_conv2d_dgrad_bnbwd_tiles(ctypes.byref(d)) if (hint is not None and hint.usable(x)) else 0
            if tiles > 0:
                # gx becomes g * [bn(ypre) > 0] and the BatchNorm's two backward sums come out per (channel, pixel tile)
                partials = torch.empty((d.C, tiles, 2), dtype=torch.float32, device=x.device)
                rc = L.cpg_conv2d_dgrad_bnbwd(ctypes.byref(d), _lib.dptr(gy, name='grad_output'), _lib.dptr(w), _lib.dptr(p), thr,
                                              _lib.dptr(hint.ypre), _lib.dptr(hint.gamma), _lib.dptr(hint.beta), _lib.dptr(hint.mean),
                                              _lib.dptr(hint.invstd), _lib.dptr(gx), _lib.dptr(partials), partials.numel() * 4,
                                              _lib.dptr(ws), nbytes, s)
                _lib.check('cpg_conv2d_dgrad_bnbwd', rc)
                hint.partials, hint.tiles = partials, tiles
            elif addend is not None and L.cpg_conv2d_dgrad_add_supported(ctypes.byref(d)) and addend.shape == x.shape:
                addend = addend.contiguous()
                pk, ctx.packed_dgrad = getattr(ctx, 'packed_dgrad', None), None      # (the operand packed at forward time, one-shot)
                ad = addend
                rc = _with_packed(L, pk, lambda: L.cpg_conv2d_dgrad_add(
                    ctypes.byref(d), _lib.dptr(gy, name='grad_output'), _lib.dptr(w), _lib.dptr(p), thr, _lib.dptr(ad, name='skip gradient'),
                    _lib.dptr(gx), _lib.dptr(ws), nbytes, s))
                _lib.check('cpg_conv2d_dgrad_add', rc)
                addend = None
            else:
                pk, ctx.packed_dgrad = getattr(ctx, 'packed_dgrad', None), None
                rc = _with_packed(L, pk, lambda: L.cpg_conv2d_dgrad(
                    ctypes.byref(d), _lib.dptr(gy, name='grad_output'), _lib.dptr(w), _lib.dptr(p), thr, _lib.dptr(gx), _lib.dptr(ws), nbytes, s))
                _lib.check('cpg_conv2d_dgrad', rc)
        if addend is not None:                  # (no fused path for this launch)
            gx = addend.clone() if gx is None else gx.add_(addend)
        if ctx.needs_input_grad[1] or (p is not None and ctx.needs_input_grad[2]) or (ctx.has_bias and ctx.needs_input_grad[3]):
            gw = torch.empty_like(w)
            gpm = None if p is None else torch.empty_like(p)
            gb = torch.empty(d.K, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            sink, ctx.bias_sink = getattr(ctx, 'bias_sink', None), None
            given = sink.gb if sink is not None else None
            if given is not None:                # the PReLU behind this conv already summed gy per channel (cpg_prelu_bwd_bias)
                sink.gb, gb = None, None
            if ctx.bf16 and not ctx.has_bias and L.cpg_conv2d_wgrad_bf16_supported(ctypes.byref(d)):
                wsw, nbw = _lib.workspace(L.cpg_conv2d_wgrad_bf16_workspace_bytes(ctypes.byref(d)), x.device)
                wgrad = L.cpg_conv2d_wgrad_bf16x3 if ctx.x3 else L.cpg_conv2d_wgrad_bf16
                rc = wgrad(ctypes.byref(d), _lib.dptr(x), _lib.dptr(gy), _lib.dptr(w), _lib.dptr(p), thr,
                           _lib.dptr(gw), _lib.dptr(gpm), _lib.dptr(wsw), nbw, s)
                _lib.check('cpg_conv2d_wgrad_bf16', rc)
            else:
                rc = L.cpg_conv2d_wgrad(ctypes.byref(d), _lib.dptr(x), _lib.dptr(gy), _lib.dptr(w), _lib.dptr(p), thr,
                                        _lib.dptr(gw), _lib.dptr(gpm), _lib.dptr(gb), _lib.dptr(ws), nbytes, s)
                _lib.check('cpg_conv2d_wgrad', rc)
            if given is not None:
                gb = given
        return gx, gw, gpm, gb, None, None, None, None, None, None, None, None, None


class _MaskedConv2dSkipFn(torch.autograd.Function):
    """(y, stats, skip) = conv2d(x), its BatchNorm partial sums (or an empty tensor) and x itself, for a residual block whose input
    feeds this conv AND the identity branch (models/resnet.py:84-104).  Being the input's only consumer, the Function receives both
    gradients and folds their sum into the input-gradient kernel's epilogue instead of leaving it to a separate add kernel."""

    @staticmethod
    def forward(ctx, x, weight, pm, bias, thr, stride, padding, dilation, groups, math, bias_sink=None, want_stats=True):
        # (bias_sink / want_stats = False: SphereNet's biased conv -> PReLU pairs, fused_bn.conv_prelu_skip -- the PReLU's backward delivers the
        #  bias gradient, and no BatchNorm follows: the plain forward epilogue)
        if want_stats:
            y, stats = _MaskedConv2dFn.forward(ctx, x, weight, pm, bias, thr, stride, padding, dilation, groups, True, math, None, bias_sink)
        else:
            y = _MaskedConv2dFn.forward(ctx, x, weight, pm, bias, thr, stride, padding, dilation, groups, False, math, None, bias_sink)
            stats = torch.empty(0, dtype=torch.float32, device=x.device)
            ctx.mark_non_differentiable(stats)
        return y, stats, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, _gstats, gskip):
        r = _MaskedConv2dFn.backward(ctx, gy, None, addend=gskip)
        return r[:4] + (None,) * 8


class _MaskedLinearFn(torch.autograd.Function):
    """y = x @ (W * bin(pm))^T + b and its gradients through the C ABI."""

    @staticmethod
    def forward(ctx, x, weight, pm, bias, thr):
        if x.shape[-1] != weight.shape[1]:
            raise RuntimeError('SharableLinear: input %s does not match weight %s' % (tuple(x.shape), tuple(weight.shape)))
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w = weight.contiguous()
        p = None if pm is None else pm.contiguous()
        batch, fin, fout = x2.shape[0], w.shape[1], w.shape[0]
        # allocate with the caller-visible shape: returning a view of a Function output would forbid the
        # in-place ReLU that follows these layers in the reference topologies
        y = torch.empty((*lead, fout), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ctx.empty = batch == 0
        if ctx.empty:                   # F.linear accepts zero rows
            _lib.dptr(x2, name='input'), _lib.dptr(w, name='weight')
            ctx.save_for_backward(x2, w, p)
            ctx.thr, ctx.has_bias, ctx.lead = float(thr), bias is not None, lead
            return y
        ws, nbytes = _lib.workspace(L.cpg_linear_workspace_bytes(batch, fin, fout), x.device)
        rc = L.cpg_linear_fwd(_lib.dptr(x2, name='input'), _lib.dptr(w, name='weight'), _lib.dptr(p, name='piggymask'),
                              float(thr), _lib.dptr(bias, name='bias'), _lib.dptr(y), batch, fin, fout,
                              _lib.dptr(ws), nbytes, _lib.stream_ptr())
        _lib.check('cpg_linear_fwd', rc)
        ctx.save_for_backward(x2, w, p)
        ctx.thr, ctx.has_bias, ctx.lead = float(thr), bias is not None, lead
        # data parallel: a very large weight hands its gradient to the exchange in row blocks (cpg_amd.dist._ChunkedGradient)
        ctx.dp_chunk = getattr(weight, '_cpg_dp_chunk', None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, w, p = ctx.saved_tensors
        thr = ctx.thr
        batch, fin, fout = x2.shape[0], w.shape[1], w.shape[0]
        if ctx.empty:
            return (torch.zeros((*ctx.lead, fin), dtype=torch.float32, device=x2.device), torch.zeros_like(w),
                    None if p is None else torch.zeros_like(p),
                    torch.zeros(fout, dtype=torch.float32, device=x2.device) if ctx.has_bias else None, None)
        gy2 = gy.reshape(-1, fout).contiguous()
        L = _lib.lib()
        s = _lib.stream_ptr()
        ws, nbytes = _lib.workspace(L.cpg_linear_workspace_bytes(batch, fin, fout), x2.device)
        gx = gw = gpm = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((*ctx.lead, fin), dtype=torch.float32, device=x2.device)
            rc = L.cpg_linear_dgrad(_lib.dptr(gy2, name='grad_output'), _lib.dptr(w), _lib.dptr(p), thr, _lib.dptr(gx),
                                    batch, fin, fout, _lib.dptr(ws), nbytes, s)
            _lib.check('cpg_linear_dgrad', rc)
        if ctx.needs_input_grad[1] or (p is not None and ctx.needs_input_grad[2]) or (ctx.has_bias and ctx.needs_input_grad[3]):
            gpm = None if p is None else torch.empty_like(p)
            gb = torch.empty(fout, dtype=torch.float32, device=x2.device) if ctx.has_bias else None
            ch = getattr(ctx, 'dp_chunk', None)
            if ch is not None and ch.active() and fout % ch.nchunks == 0:
                # the same GEMM in blocks of output rows, each block's all-reduce started as soon as its kernels are launched
                gw = ch.buffer(w)
                rows = fout // ch.nchunks
                wsc, nbc = _lib.workspace(L.cpg_linear_workspace_bytes(batch, fin, rows), x2.device)
                for i in range(ch.nchunks):
                    r0, r1 = i * rows, (i + 1) * rows
                    gyc = gy2[:, r0:r1].contiguous()
                    rc = L.cpg_linear_wgrad(_lib.dptr(x2), _lib.dptr(gyc), _lib.dptr(w[r0:r1]), _lib.dptr(None if p is None else p[r0:r1]), thr,
                                            _lib.dptr(gw[r0:r1]), _lib.dptr(None if gpm is None else gpm[r0:r1]),
                                            _lib.dptr(None if gb is None else gb[r0:r1]), batch, fin, rows, _lib.dptr(wsc), nbc, s)
                    _lib.check('cpg_linear_wgrad', rc)
                    ch.ready(gw, r0, r1)
            else:
                gw = torch.empty_like(w)
                rc = L.cpg_linear_wgrad(_lib.dptr(x2), _lib.dptr(gy2), _lib.dptr(w), _lib.dptr(p), thr, _lib.dptr(gw),
                                        _lib.dptr(gpm), _lib.dptr(gb), batch, fin, fout, _lib.dptr(ws), nbytes, s)
                _lib.check('cpg_linear_wgrad', rc)
        return gx, gw, gpm, gb, None


class HeadLinear(nn.Linear):
    """A task head that is a plain nn.Linear in the reference (never masked, never pruned: models/spherenet.py:240-245, the 25 088 -> 512
    embedding in front of AngleLinear) on the same C-ABI GEMMs as the masked linear layers, without a mask: parameters, state_dict keys
    and initialisation are nn.Linear's.  In SphereNet-20's train step stock torch spent 0.38 ms (of 19.9) in three library GEMMs for this
    layer -- 27 TFLOP/s in its weight gradient, a 256-deep contraction."""

    def forward(self, input):
        return _MaskedLinearFn.apply(input, self.weight, None, self.bias, 0.0)


class _Sharable(nn.Module):
    """State shared by both masked layers: threshold bookkeeping and the late-bound piggymask.

    `piggymask` starts as None (task 1) and is later ASSIGNED an nn.Parameter shaped like the weight
    by the driver (CPG_cifar100_main_normal.py:263-270); nn.Module.__setattr__ then registers it,
    so it appears in named_parameters()/state_dict() and follows .cuda()/.to() like the reference's.
    """

    def _init_mask_state(self, mask_init, mask_scale, threshold_fn, threshold):
        self.mask_init, self.mask_scale = mask_init, mask_scale
        self.info = {'threshold_fn': threshold_fn,
                     'threshold': DEFAULT_THRESHOLD if threshold is None else threshold}
        if threshold_fn == 'binarizer':
            self.threshold_fn = Binarizer.apply
        elif threshold_fn == 'ternarizer':
            raise NotImplementedError('ternarizer is dead code in the reference (old autograd API); only '
                                      "'binarizer' is implemented")
        else:
            raise ValueError('unknown threshold_fn %r' % (threshold_fn,))
        self.piggymask = None


class SharableConv2d(_Sharable):
    """Conv2d whose effective weight is W * binarize(piggymask) (models/layers.py:43-109)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, mask_init='1s', mask_scale=1e-2, threshold_fn='binarizer', threshold=None):
        super().__init__()
        if in_channels % groups or out_channels % groups:
            raise ValueError('in_channels and out_channels must be divisible by groups')
        # groups > 1 (the reference forwards `groups` to F.conv2d, models/layers.py:108-109; no CPG configuration uses it -- the resnext*
        # factories and VGG(groups=...) exist in its model files): one launch of the groups == 1 kernels per group on a contiguous
        # channel slice, see _grouped() -- correct and on the HIP path, not tuned
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.transposed, self.output_padding = False, _pair(0)
        # uninitialised storage, no RNG consumed here (init happens in the model, models/vgg.py:59-70)
        self.weight = Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self._init_mask_state(mask_init, mask_scale, threshold_fn, threshold)

    def _grouped(self, input):
        """groups > 1: y = cat_g conv2d(x[:, g-th channel slice], W[g-th row block] * bin(pm[...]), b[...]) -- every group through the
        groups == 1 kernels (autograd slices / concatenates; the row blocks of weight, piggymask and bias are contiguous views)."""
        G = self.groups
        if input.dim() != 4 or input.shape[1] != self.weight.shape[1] * G:
            raise RuntimeError('SharableConv2d: input %s does not match weight %s (groups=%d)'
                               % (tuple(input.shape), tuple(self.weight.shape), G))
        ko = self.out_channels // G
        outs = []
        for gi, xg in enumerate(input.chunk(G, dim=1)):
            rows = slice(gi * ko, (gi + 1) * ko)
            outs.append(_MaskedConv2dFn.apply(xg.contiguous(), self.weight[rows], None if self.piggymask is None else self.piggymask[rows],
                                              None if self.bias is None else self.bias[rows], self.info['threshold'], self.stride,
                                              self.padding, self.dilation, 1, False, self._math(), None, None))
        return torch.cat(outs, dim=1)

    def forward(self, input, layer_info=None, name=None, bn_hint=None, bias_sink=None):
        if self.groups != 1:
            return self._grouped(input)
        return _MaskedConv2dFn.apply(input, self.weight, self.piggymask, self.bias, self.info['threshold'],
                                     self.stride, self.padding, self.dilation, self.groups, False, self._math(), bn_hint, bias_sink)

    def _math(self):
        return getattr(self, 'math', None) or CONV_MATH

    def forward_with_bn_stats(self, input, bn_hint=None):
        """(y, stats): forward plus the BatchNorm partial sums of y from the same kernel; stats is None when this shape
        has no fused-statistics kernel (groups > 1 among them: the per-group calls take no statistics epilogue, no BatchNorm-backward hint
        and no bias sink -- callers get stats None and run the separate statistics pass).  Used by cpg_amd.models.fused_bn.FusedSequential
        for conv -> BatchNorm2d runs."""
        if self.groups != 1:
            return self._grouped(input), None
        y, stats = _MaskedConv2dFn.apply(input, self.weight, self.piggymask, self.bias, self.info['threshold'],
                                         self.stride, self.padding, self.dilation, self.groups, True, self._math(), bn_hint)
        return y, (stats if stats.numel() else None)

    def forward_with_skip(self, input, bias_sink=None, want_stats=True):
        """(y, stats or None, skip): forward (+ BatchNorm partial sums where the shape has them) and the input handed back for the
        residual branch; the two gradients of the input are summed inside the input-gradient kernel (_MaskedConv2dSkipFn)."""
        if self.groups != 1:
            return self._grouped(input), None, input
        y, stats, skip = _MaskedConv2dSkipFn.apply(input, self.weight, self.piggymask, self.bias, self.info['threshold'],
                                                   self.stride, self.padding, self.dilation, self.groups, self._math(), bias_sink, want_stats)
        return y, (stats if stats.numel() else None), skip

    def forward_bn_eval(self, input, bn, relu=True, skip_stats=None):
        """relu(bn(conv(input))) with `bn` an eval-mode nn.BatchNorm2d, as ONE kernel (cpg_conv2d_fwd_bn_eval): the path of
        Manager.validate.  Inference only -- call it under torch.no_grad(); returns None when this shape has no fused
        kernel (the caller then runs the layers one by one).  skip_stats: optional int32[2] device tensor that receives
        {1 + last live input channel, output tiles skipped} (dead-channel skip, see include/cpg_hip.h)."""
        if (torch.is_grad_enabled() or input.dim() != 4 or input.shape[0] == 0 or input.shape[1] != self.weight.shape[1] * self.groups
                or self._math() != 'fp32' or self.groups != 1):
            return None
        x = input.contiguous()
        w = self.weight.contiguous()
        p = None if self.piggymask is None else self.piggymask.contiguous()
        d = _conv_desc(x.shape, w.shape, self.stride, self.padding, self.dilation, self.groups)
        L = _lib.lib()
        if not L.cpg_conv2d_fwd_bn_eval_supported(ctypes.byref(d)):
            return None
        oh, ow = _out_hw(d)
        y = torch.empty((d.N, d.K, oh, ow), dtype=torch.float32, device=x.device)
        ws, nbytes = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), x.device)
        rc = L.cpg_conv2d_fwd_bn_eval(ctypes.byref(d), _lib.dptr(x, name='input'), _lib.dptr(w, name='weight'), _lib.dptr(p, name='piggymask'),
                                      float(self.info['threshold']), _lib.dptr(self.bias, name='bias'), _lib.dptr(bn.weight, name='bn.weight'),
                                      _lib.dptr(bn.bias, name='bn.bias'), _lib.dptr(bn.running_mean, name='running_mean'),
                                      _lib.dptr(bn.running_var, name='running_var'), float(bn.eps), int(bool(relu)), _lib.dptr(y),
                                      None if skip_stats is None else ctypes.c_void_p(skip_stats.data_ptr()),
                                      _lib.dptr(ws), nbytes, _lib.stream_ptr())
        _lib.check('cpg_conv2d_fwd_bn_eval', rc)
        return y

    def extra_repr(self):
        s = '{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}'
        if any(self.padding):
            s += ', padding={padding}'
        if self.dilation != (1, 1):
            s += ', dilation={dilation}'
        if self.groups != 1:
            s += ', groups={groups}'
        if self.bias is None:
            s += ', bias=False'
        return s.format(**self.__dict__)


class SharableLinear(_Sharable):
    """Linear layer whose effective weight is W * binarize(piggymask) (models/layers.py:147-194)."""

    def __init__(self, in_features, out_features, bias=True, mask_init='1s', mask_scale=1e-2,
                 threshold_fn='binarizer', threshold=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self._init_mask_state(mask_init, mask_scale, threshold_fn, threshold)

    def forward(self, input):
        return _MaskedLinearFn.apply(input, self.weight, self.piggymask, self.bias, self.info['threshold'])

    def extra_repr(self):
        return 'in_features=%d, out_features=%d' % (self.in_features, self.out_features)
