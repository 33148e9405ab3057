"""Fused BatchNorm2d -> ReLU for the blocks that follow every masked conv of the VGG topologies
(reference: `layers += [conv2d, nn.BatchNorm2d(c), nn.ReLU(inplace=True)]`, models/vgg.py:137-141).

The modules stay what they are in the reference (an `nn.BatchNorm2d` with its parameters / buffers and an
`nn.ReLU`), so state_dict keys, optimizer parameter lists and `shared_layer_info` bookkeeping are
unchanged; only the arithmetic of the pair is replaced by libcpg_hip.so's three-pass forward / five-pass
backward (csrc/bn_kernels.hip).  Semantics are torch.nn.BatchNorm2d's: batch statistics in training mode
(biased variance for the normalisation, unbiased for running_var, momentum update, num_batches_tracked),
running statistics in eval mode.  SURVEY.md section 8(f) item 2.
"""
import ctypes
import threading

import torch
import torch.nn as nn

from .. import _lib


class BnBwdHint(object):
    """Link between a training-mode BatchNorm2d -> ReLU and the masked 3x3 conv that is the ONLY consumer of its output:
    the conv's input-gradient kernel does the BatchNorm's backward reduction in its epilogue (cpg_conv2d_dgrad_bnbwd) and
    leaves the partial sums here for _BnReluFn.backward (cpg_bn_bwd_from_partials).  One hint per forward pass."""
    __slots__ = ('ypre', 'gamma', 'beta', 'mean', 'invstd', 'partials', 'tiles', 'out_shape')

    def __init__(self):
        self.ypre = self.gamma = self.beta = self.mean = self.invstd = self.partials = self.out_shape = None
        self.tiles = 0

    def usable(self, x):
        return self.ypre is not None and tuple(x.shape) == self.out_shape and x.is_cuda


# The BatchNorm backward reduction riding in the next conv's input-gradient epilogue.  OFF by default: it removes the 2-read
# reduction kernels (-2.4 ms per VGG16 step) but the input-gradient kernels pay the HBM time of the extra reads in an epilogue
# that nothing hides (+0.28 ms per layer); interleaved in-process A/B (tools/step_ab.py): 184.6 ms with it, 183.8 ms without.
ENABLE_BWD_HINT = False


class _BnReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, training, relu, stats=None, hint=None):
        x = x.contiguous()
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        L = _lib.lib()
        y = torch.empty_like(x)
        s = _lib.stream_ptr()
        if training and stats is not None:
            # the producing conv already accumulated the partial sums: finalize, then the apply pass alone
            mean, invstd = _finalize_stats(stats, N, C, HW, eps, momentum, running_mean, running_var, x.device)
            rc = L.cpg_bn_relu_fwd_eval(_lib.dptr(x, name='input'), _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(mean),
                                        _lib.dptr(invstd), _lib.dptr(y), N, C, HW, int(relu), s)
            _lib.check('cpg_bn_relu_fwd_eval', rc)
        elif training:
            mean = torch.empty(C, dtype=torch.float32, device=x.device)
            invstd = torch.empty(C, dtype=torch.float32, device=x.device)
            ws, nb = _lib.workspace(L.cpg_bn_workspace_bytes(N, C, HW), x.device)
            rc = L.cpg_bn_relu_fwd_train(_lib.dptr(x, name='input'), _lib.dptr(gamma, name='bn.weight'), _lib.dptr(beta, name='bn.bias'),
                                         float(eps), float(momentum), _lib.dptr(running_mean, name='running_mean'),
                                         _lib.dptr(running_var, name='running_var'), _lib.dptr(mean), _lib.dptr(invstd),
                                         _lib.dptr(y), N, C, HW, int(relu), _lib.dptr(ws), nb, s)
            _lib.check('cpg_bn_relu_fwd_train', rc)
        else:
            mean = running_mean
            invstd = torch.rsqrt(running_var + eps)
            rc = L.cpg_bn_relu_fwd_eval(_lib.dptr(x, name='input'), _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(mean),
                                        _lib.dptr(invstd), _lib.dptr(y), N, C, HW, int(relu), s)
            _lib.check('cpg_bn_relu_fwd_eval', rc)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.cfg = (N, C, HW, bool(relu), bool(training))
        ctx.hint = None
        if hint is not None and relu and training:
            hint.ypre, hint.gamma, hint.beta, hint.mean, hint.invstd, hint.out_shape = x, gamma, beta, mean, invstd, tuple(y.shape)
            ctx.hint = hint
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        N, C, HW, relu, training = ctx.cfg
        gy = gy.contiguous()
        L = _lib.lib()
        gx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        ws, nb = _lib.workspace(L.cpg_bn_workspace_bytes(N, C, HW), x.device)
        hint, ctx.hint = ctx.hint, None                  # (the hint only lives for one backward pass: drop its tensors afterwards)
        if hint is not None:
            hint.ypre = hint.gamma = hint.beta = hint.mean = hint.invstd = hint.out_shape = None
        if hint is not None and hint.partials is not None:
            # the consumer conv's input-gradient kernel already masked gy by the ReLU and reduced it per (channel, tile)
            rc = L.cpg_bn_bwd_from_partials(_lib.dptr(hint.partials), hint.tiles, _lib.dptr(x), _lib.dptr(gy, name='grad_output'),
                                            _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(mean), _lib.dptr(invstd), _lib.dptr(gx),
                                            _lib.dptr(dgamma), _lib.dptr(dbeta), N, C, HW, _lib.dptr(ws), nb, _lib.stream_ptr())
            _lib.check('cpg_bn_bwd_from_partials', rc)
            hint.partials = None
            return gx, dgamma, dbeta, None, None, None, None, None, None, None, None
        rc = L.cpg_bn_relu_bwd(_lib.dptr(x), _lib.dptr(gy, name='grad_output'), _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(mean),
                               _lib.dptr(invstd), _lib.dptr(gx), _lib.dptr(dgamma), _lib.dptr(dbeta), N, C, HW, int(relu),
                               int(training), _lib.dptr(ws), nb, _lib.stream_ptr())
        _lib.check('cpg_bn_relu_bwd', rc)
        return gx, dgamma, dbeta, None, None, None, None, None, None, None, None


_tls = threading.local()


def _count_batch(bn):
    """nn.BatchNorm2d's `num_batches_tracked += 1` of a training-mode forward, handed to the kernel that finalises this layer's batch
    statistics (cpg_bn_stats_finalize_count bumps the counter in the same launch: stock torch spends one `add<long>` launch per layer on
    it, 53 per ResNet-50 step).  _settle_count() right behind the fused call does the plain add when no such kernel took it."""
    _tls.nbt = bn.num_batches_tracked


def _settle_count():
    nbt = getattr(_tls, 'nbt', None)
    if nbt is not None:
        _tls.nbt = None
        nbt.add_(1)


def _finalize_stats(stats, N, C, HW, eps, momentum, running_mean, running_var, device):
    """mean / invstd (+ running statistics update, + the pending num_batches_tracked bump) from a conv's [C][tiles][2] partial sums
    (cpg_bn_stats_finalize / cpg_bn_stats_finalize_count)."""
    L = _lib.lib()
    mean = torch.empty(C, dtype=torch.float32, device=device)
    invstd = torch.empty(C, dtype=torch.float32, device=device)
    nbt = getattr(_tls, 'nbt', None)
    if nbt is not None and running_mean is not None and nbt.is_cuda and nbt.dtype == torch.int64 and nbt.numel() == 1:
        _tls.nbt = None
        rc = L.cpg_bn_stats_finalize_count(_lib.dptr(stats, name='bn partial sums'), stats.shape[1], N, C, HW, float(eps), float(momentum),
                                           _lib.dptr(running_mean, name='running_mean'), _lib.dptr(running_var, name='running_var'),
                                           _lib.dptr(mean), _lib.dptr(invstd), _lib.dptr(nbt, torch.int64, 'num_batches_tracked'),
                                           _lib.stream_ptr())
        _lib.check('cpg_bn_stats_finalize_count', rc)
        return mean, invstd
    rc = L.cpg_bn_stats_finalize(_lib.dptr(stats, name='bn partial sums'), stats.shape[1], N, C, HW, float(eps), float(momentum),
                                 _lib.dptr(running_mean, name='running_mean'), _lib.dptr(running_var, name='running_var'),
                                 _lib.dptr(mean), _lib.dptr(invstd), _lib.stream_ptr())
    _lib.check('cpg_bn_stats_finalize', rc)
    return mean, invstd


class _BnAddReluFn(torch.autograd.Function):
    """relu(bn(x) + res) -- forward in the two passes of plain BN; backward = ReLU mask from the saved output, the plain
    BN backward kernels, and the masked gradient itself for the residual."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, eps, momentum, training, stats=None):
        x, res = x.contiguous(), res.contiguous()
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        L = _lib.lib()
        y = torch.empty_like(x)
        apply_only = not training
        if training and stats is not None:
            # the producing conv's epilogue already accumulated the partial sums (cpg_conv2d_fwd_bnstats): finalize + apply pass
            mean, invstd = _finalize_stats(stats, N, C, HW, eps, momentum, running_mean, running_var, x.device)
            ws, nb = None, 0
            apply_only = True
        elif training:
            mean = torch.empty(C, dtype=torch.float32, device=x.device)
            invstd = torch.empty(C, dtype=torch.float32, device=x.device)
            ws, nb = _lib.workspace(L.cpg_bn_workspace_bytes(N, C, HW), x.device)
        else:
            mean, invstd = running_mean, torch.rsqrt(running_var + eps)
            ws, nb = None, 0
        # the backward's ReLU mask as one byte per four outputs, written by the forward pass (the backward then does not re-read y)
        mask = None
        nmask = L.cpg_bn_add_relu_mask_bytes(N, C, HW) if (RELU_BYTE_MASK and any(ctx.needs_input_grad)) else 0
        if nmask and x.data_ptr() % 16 == 0 and res.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:
            mask = torch.empty(nmask, dtype=torch.uint8, device=x.device)
        rc = L.cpg_bn_add_relu_fwd(_lib.dptr(x, name='input'), _lib.dptr(res, name='residual'), _lib.dptr(gamma), _lib.dptr(beta),
                                   float(eps), float(momentum), _lib.dptr(None if apply_only else running_mean),
                                   _lib.dptr(None if apply_only else running_var), _lib.dptr(mean), _lib.dptr(invstd), _lib.dptr(y),
                                   N, C, HW, int(not apply_only), _lib.dptr(ws), nb, _lib.stream_ptr(), _lib.dptr(mask, torch.uint8))
        _lib.check('cpg_bn_add_relu_fwd', rc)
        if mask is not None:
            ctx.save_for_backward(x, mask, gamma, beta, mean, invstd)
        else:
            ctx.save_for_backward(x, y, gamma, beta, mean, invstd)
        ctx.has_mask = mask is not None
        ctx.cfg = (N, C, HW, bool(training))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, beta, mean, invstd = ctx.saved_tensors
        mask, y = (y, None) if ctx.has_mask else (None, y)
        N, C, HW, training = ctx.cfg
        gy = gy.contiguous()
        if mask is not None and gy.data_ptr() % 16:
            gy = gy.clone()
        L = _lib.lib()
        gx = torch.empty_like(x)
        gz = torch.empty_like(x)                       # gy * [y > 0]: the residual branch's gradient, written by the reduction pass
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        ws, nb = _lib.workspace(L.cpg_bn_workspace_bytes(N, C, HW), x.device)
        rc = L.cpg_bn_add_relu_bwd(_lib.dptr(x), _lib.dptr(y), _lib.dptr(gy, name='grad_output'), _lib.dptr(gamma), _lib.dptr(beta),
                                   _lib.dptr(mean), _lib.dptr(invstd), _lib.dptr(gx), _lib.dptr(gz), _lib.dptr(dgamma), _lib.dptr(dbeta),
                                   N, C, HW, int(training), _lib.dptr(ws), nb, _lib.stream_ptr(), _lib.dptr(mask, torch.uint8))
        _lib.check('cpg_bn_add_relu_bwd', rc)
        return gx, gz, dgamma, dbeta, None, None, None, None, None, None


class _BnReluPoolFn(torch.autograd.Function):
    """BatchNorm2d -> ReLU -> MaxPool2d(2, 2); only the pooled tensor is written."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, training, stats=None):
        x = x.contiguous()
        N, C, H, W = x.shape
        L = _lib.lib()
        y = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        compute_stats = training and stats is None
        if training and stats is not None:
            mean, invstd = _finalize_stats(stats, N, C, H * W, eps, momentum, running_mean, running_var, x.device)
        elif training:
            mean = torch.empty(C, dtype=torch.float32, device=x.device)
            invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        else:
            mean, invstd = running_mean, torch.rsqrt(running_var + eps)
        ws, nb = _lib.workspace(L.cpg_bn_workspace_bytes(N, C, H * W), x.device)
        rc = L.cpg_bn_relu_pool_fwd(_lib.dptr(x, name='input'), _lib.dptr(gamma, name='bn.weight'), _lib.dptr(beta, name='bn.bias'),
                                    float(eps), float(momentum), _lib.dptr(running_mean if compute_stats else None),
                                    _lib.dptr(running_var if compute_stats else None), _lib.dptr(mean), _lib.dptr(invstd),
                                    _lib.dptr(y), N, C, H, W, int(compute_stats), _lib.dptr(ws), nb, _lib.stream_ptr())
        _lib.check('cpg_bn_relu_pool_fwd', rc)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, gp):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        gp = gp.contiguous()
        L = _lib.lib()
        gx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        ws, nb = _lib.workspace(L.cpg_bn_workspace_bytes(N, C, H * W), x.device)
        rc = L.cpg_bn_relu_pool_bwd(_lib.dptr(x), _lib.dptr(gp, name='grad_output'), _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(mean),
                                    _lib.dptr(invstd), _lib.dptr(gx), _lib.dptr(dgamma), _lib.dptr(dbeta), N, C, H, W,
                                    int(ctx.training), _lib.dptr(ws), nb, _lib.stream_ptr())
        _lib.check('cpg_bn_relu_pool_bwd', rc)
        return gx, dgamma, dbeta, None, None, None, None, None, None


class _BnReluPool3Fn(torch.autograd.Function):
    """BatchNorm2d -> ReLU -> MaxPool2d(3, 2, 1) (the ResNet stem's tail); only the pooled tensor is written."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, training, stats):
        x = x.contiguous()
        N, C, H, W = x.shape
        L = _lib.lib()
        if training and stats is not None:
            mean, invstd = _finalize_stats(stats, N, C, H * W, eps, momentum, running_mean, running_var, x.device)
        elif training:
            mean = torch.empty(C, dtype=torch.float32, device=x.device)
            invstd = torch.empty(C, dtype=torch.float32, device=x.device)
            # (the producing conv has no fused-statistics kernel -- narrow test nets: torch's own reduction for the statistics)
            var, mu = torch.var_mean(x, dim=(0, 2, 3), unbiased=False)
            mean.copy_(mu)
            invstd.copy_(torch.rsqrt(var + eps))
            n = x.numel() // C
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(mu, alpha=momentum)
                running_var.mul_(1 - momentum).add_(var * (n / max(n - 1, 1)), alpha=momentum)
        else:
            mean, invstd = running_mean, torch.rsqrt(running_var + eps)
        y = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
        rc = L.cpg_bn_relu_pool3_fwd(_lib.dptr(x, name='input'), _lib.dptr(gamma, name='bn.weight'), _lib.dptr(beta, name='bn.bias'),
                                     _lib.dptr(mean), _lib.dptr(invstd), _lib.dptr(y), N, C, H, W, _lib.stream_ptr())
        _lib.check('cpg_bn_relu_pool3_fwd', rc)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, gp):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        gp = gp.contiguous()
        L = _lib.lib()
        gx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        ws, nb = _lib.workspace(L.cpg_bn_workspace_bytes(N, C, H * W), x.device)
        rc = L.cpg_bn_relu_pool3_bwd(_lib.dptr(x), _lib.dptr(gp, name='grad_output'), _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(mean),
                                     _lib.dptr(invstd), _lib.dptr(gx), _lib.dptr(dgamma), _lib.dptr(dbeta), N, C, H, W,
                                     int(ctx.training), _lib.dptr(ws), nb, _lib.stream_ptr())
        _lib.check('cpg_bn_relu_pool3_bwd', rc)
        return gx, dgamma, dbeta, None, None, None, None, None, None


def _is_pool3(m):
    def pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    return (isinstance(m, nn.MaxPool2d) and pair(m.kernel_size) == (3, 3) and pair(m.stride) == (2, 2) and pair(m.padding) == (1, 1)
            and pair(m.dilation) == (1, 1) and not m.ceil_mode and not m.return_indices)


def conv_bn_act_pool(conv, bn, act, pool, x):
    """pool(act(bn(conv(x)))): the ResNet stem (models/resnet.py:208-211).  conv -> BatchNorm statistics in the conv epilogue,
    BatchNorm -> ReLU -> MaxPool2d(3, 2, 1) as one forward kernel and two backward kernels when the plane fits LDS."""
    y, stats = _conv_with_stats(conv, bn, x)
    if (ENABLED and FusedSequential.fuse_pool and type(act) is nn.ReLU and _is_pool3(pool) and fusable(bn, y) and bn.track_running_stats
            and _lib.lib().cpg_bn_relu_pool3_supported(int(y.shape[2]), int(y.shape[3]))):
        training = bn.training
        if training and bn.num_batches_tracked is not None:
            _count_batch(bn)
        out = _BnReluPool3Fn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, training,
                                   stats if training else None)
        _settle_count()
        return out
    if stats is not None and type(act) is nn.ReLU and fusable(bn, y):
        return pool(bn_relu(y, bn, relu=True, stats=stats))
    return pool(bn_act(bn, act, y))


def _is_pool2(m):
    def pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    return (isinstance(m, nn.MaxPool2d) and pair(m.kernel_size) == (2, 2) and pair(m.stride) == (2, 2) and pair(m.padding) == (0, 0)
            and pair(m.dilation) == (1, 1) and not m.ceil_mode and not m.return_indices)


def bn_relu_pool(x, bn, stats=None):
    """max_pool2d(relu(bn(x)), 2, 2) with `bn` an nn.BatchNorm2d module; H and W must be even.  `stats`: partial sums
    of x from the conv that produced it (SharableConv2d.forward_with_bn_stats)."""
    training = bn.training or not bn.track_running_stats
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        _count_batch(bn)
    out = _BnReluPoolFn.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, training, stats if training else None)
    _settle_count()
    return out


def fusable(bn, x):
    """An affine, stat-tracking (or eval-mode) BatchNorm2d on a 4-D fp32 HIP tensor with the default
    exponential-average momentum -- everything the CPG topologies construct."""
    return (isinstance(bn, nn.BatchNorm2d) and bn.affine and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and bn.momentum is not None and (bn.track_running_stats or bn.training))


def bn_relu(x, bn, relu=True, stats=None, hint=None):
    """y = relu(bn(x)) with `bn` an nn.BatchNorm2d module (its buffers are updated as torch would).  hint: a BnBwdHint when the
    caller hands y to exactly one masked 3x3 conv (FusedSequential does)."""
    training = bn.training or not bn.track_running_stats
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        _count_batch(bn)
    out = _BnReluFn.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, training, relu,
                          stats if (training and rm is not None) else None, hint if training else None)
    _settle_count()
    return out


class _PReluFn(torch.autograd.Function):
    """nn.PReLU (+ an optional residual added to its output: SphereNet's `x + relu(conv(y))`) as one forward pass (cpg_prelu_fwd)
    and a one-pass backward (cpg_prelu_bwd); the residual's gradient is the incoming gradient itself."""

    @staticmethod
    def forward(ctx, x, weight, res=None, bias_sink=None):
        ctx.bias_sink = bias_sink        # layers.BiasGradSink of the biased conv that produced x (its only consumer is this PReLU)
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // max(N * C, 1)
        y = torch.empty_like(x)
        if x.numel():
            rc = _lib.lib().cpg_prelu_fwd(_lib.dptr(x, name='input'), _lib.dptr(res, name='residual'), _lib.dptr(weight, name='prelu.weight'),
                                          _lib.dptr(y), N, C, HW, weight.numel(), _lib.stream_ptr())
            _lib.check('cpg_prelu_fwd', rc)
        ctx.save_for_backward(x, weight)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        sink, ctx.bias_sink = ctx.bias_sink, None
        if not x.numel():
            return torch.zeros_like(x), torch.zeros_like(weight), (gy if ctx.has_res else None), None
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        L = _lib.lib()
        ws, nb = _lib.workspace(L.cpg_prelu_workspace_bytes(N, C, HW), x.device)
        gx = torch.empty_like(x)
        gw = torch.empty_like(weight)
        if sink is not None:
            gbias = torch.empty(C, dtype=torch.float32, device=x.device)
            _lib.check('cpg_prelu_bwd_bias', L.cpg_prelu_bwd_bias(_lib.dptr(x), _lib.dptr(gy), _lib.dptr(weight), _lib.dptr(gx), _lib.dptr(gw),
                                                                  _lib.dptr(gbias), N, C, HW, weight.numel(), _lib.dptr(ws), nb, _lib.stream_ptr()))
            sink.gb = gbias
        else:
            _lib.check('cpg_prelu_bwd', L.cpg_prelu_bwd(_lib.dptr(x), _lib.dptr(gy), _lib.dptr(weight), _lib.dptr(gx), _lib.dptr(gw),
                                                        N, C, HW, weight.numel(), _lib.dptr(ws), nb, _lib.stream_ptr()))
        return gx, gw, (gy if ctx.has_res else None), None


def conv_prelu(conv, mod, x, res=None):
    """mod(conv(x)) [+ res] for SphereNet's biased conv -> PReLU pairs (models/spherenet.py:203-247): the PReLU's backward pass also
    delivers the conv's bias gradient (the per-channel sum of the gradient it writes), so the conv's backward runs no bias reduction."""
    from .layers import BiasGradSink
    if (ENABLED and type(mod) is nn.PReLU and x.is_cuda and getattr(conv, 'bias', None) is not None and torch.is_grad_enabled()
            and mod.weight.numel() in (1, conv.out_channels) and conv._math() == 'fp32'
            and getattr(conv, 'groups', 1) == 1):           # (a grouped conv runs one call per group: each computes its own bias gradient)
        sink = BiasGradSink()
        y = conv(x, bias_sink=sink)
        if y.dtype == torch.float32 and y.dim() == 4 and y.is_contiguous() and (res is None or (res.shape == y.shape and res.is_contiguous())):
            return _PReluFn.apply(y, mod.weight, res, sink)
        return prelu(mod, y, res)
    return prelu(mod, conv(x), res)


def conv_prelu_skip(conv, mod, x):
    """(mod(conv(x)), x) for the FIRST conv of a SphereNet residual unit (models/spherenet.py:121-131: `x = x + relu(conv(relu(conv(x))))`):
    x feeds this conv and the unit's sum.  The returned x is routed through the conv's autograd node, which then receives BOTH gradients
    of x and adds the sum's in its input-gradient epilogue (cpg_conv2d_dgrad_add: the two-wave Winograd kernel's ADD instances) instead of
    leaving a separate add kernel to autograd.  Falls back to (conv_prelu(conv, mod, x), x) when the pair does not qualify."""
    from .layers import BiasGradSink
    if (ENABLED and type(mod) is nn.PReLU and x.is_cuda and torch.is_grad_enabled() and x.requires_grad and mod.weight.numel() in (1, conv.out_channels)
            and conv._math() == 'fp32' and getattr(conv, 'groups', 1) == 1 and x.dim() == 4 and x.is_contiguous()):
        sink = BiasGradSink() if getattr(conv, 'bias', None) is not None else None
        y, _, skip = conv.forward_with_skip(x, bias_sink=sink, want_stats=False)
        if y.dtype == torch.float32 and y.is_contiguous():
            return _PReluFn.apply(y, mod.weight, None, sink), skip
        return prelu(mod, y), skip
    return conv_prelu(conv, mod, x), x


def prelu(mod, x, res=None):
    """mod(x) [+ res] for an nn.PReLU module (models/spherenet.py); HIP kernels when the tensor qualifies."""
    if (ENABLED and type(mod) is nn.PReLU and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and mod.weight.numel() in (1, x.shape[1]) and (res is None or (res.shape == x.shape and res.is_contiguous()
                                                                            and res.dtype == torch.float32 and res.is_cuda))):
        return _PReluFn.apply(x, mod.weight, res)
    y = mod(x)
    return y if res is None else res + y


RELU_BYTE_MASK = True   # relu(bn(x) + res): the forward leaves the ReLU mask as one byte per four outputs for the backward
ENABLED = True      # module-wide switch (tests compare the fused against the stock evaluation)


def bn_act(bn, act, x):
    """act(bn(x)) for the residual topologies (models/resnet.py: `self.relu(self.bn1(...))`, `self.bn3(...)`): one fused
    kernel pair when `act` is a plain nn.ReLU (or None) and `bn` qualifies, the stock modules otherwise."""
    if ENABLED and (act is None or type(act) is nn.ReLU) and fusable(bn, x):
        return bn_relu(x, bn, relu=act is not None)
    y = bn(x)
    return y if act is None else act(y)


def bn_add_act(bn, act, x, res, stats=None):
    """act(bn(x) + res), the tail of a residual block; fused when `act` is a plain nn.ReLU and `bn` qualifies.  stats: partial sums
    of x from the conv that produced it (conv_bn_add_act)."""
    if ENABLED and type(act) is nn.ReLU and fusable(bn, x) and bn.track_running_stats and res.shape == x.shape:
        training = bn.training
        if training and bn.num_batches_tracked is not None:
            _count_batch(bn)
        out = _BnAddReluFn.apply(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, training,
                                 stats if training else None)
        _settle_count()
        return out
    out = bn(x)
    out = out + res
    return act(out)


def _conv_with_stats(conv, bn, x):
    """(conv(x), partial sums for `bn` or None): the conv's epilogue accumulates the BatchNorm statistics when the pair qualifies
    (a masked conv with a fused-statistics kernel feeding a training-mode, stat-tracking BatchNorm2d)."""
    if (ENABLED and FusedSequential.fuse_stats and hasattr(conv, 'forward_with_bn_stats') and isinstance(bn, nn.BatchNorm2d) and bn.training
            and bn.track_running_stats and bn.affine and bn.momentum is not None and x.is_cuda and torch.is_grad_enabled()):
        return conv.forward_with_bn_stats(x)
    return conv(x), None


def conv_bn_act(conv, bn, act, x):
    """act(bn(conv(x))) for the residual topologies (models/resnet.py:86-93): bn_act with the statistics pass folded into the conv."""
    y, stats = _conv_with_stats(conv, bn, x)
    if stats is not None and (act is None or type(act) is nn.ReLU) and fusable(bn, y):
        return bn_relu(y, bn, relu=act is not None, stats=stats)
    return bn_act(bn, act, y)


FUSE_STEM_WGRAD = True     # ... and its weight gradient inside the BatchNorm backward's apply pass (cpg_stem_bn_relu_bwd_wgrad)
FUSE_STEM = True           # conv(<= 3 -> 64 channels, 3x3 s1 p1) -> BatchNorm2d -> ReLU: the conv output is never written (cpg_stem_bn_*)


class _StemConvBnReluFn(torch.autograd.Function):
    """z = relu(bn(conv(x))) for the network stem (models/vgg.py:137-141) with the conv output recomputed in every pass that needs it
    instead of stored: forward = statistics pass + BatchNorm/ReLU pass over the image, backward = reduction pass + apply pass over gz
    (each recomputes conv(x)), then the stem's weight gradient from the resulting gy.  The image gets no gradient."""

    @staticmethod
    def forward(ctx, x, weight, pm, thr, gamma, beta, running_mean, running_var, eps, momentum):
        from .layers import _conv_desc
        x = x.contiguous()
        w = weight.contiguous()
        p = None if pm is None else pm.contiguous()
        d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
        L = _lib.lib()
        s = _lib.stream_ptr()
        N, K, H, W = x.shape[0], w.shape[0], x.shape[2], x.shape[3]
        tiles = L.cpg_stem_bn_tiles(ctypes.byref(d))
        stats = torch.empty((K, tiles, 2), dtype=torch.float32, device=x.device)
        rc = L.cpg_stem_bn_stats(ctypes.byref(d), _lib.dptr(x, name='input'), _lib.dptr(w, name='weight'), _lib.dptr(p, name='piggymask'),
                                 float(thr), None, _lib.dptr(stats), stats.numel() * 4, s)
        _lib.check('cpg_stem_bn_stats', rc)
        mean, invstd = _finalize_stats(stats, N, K, H * W, eps, momentum, running_mean, running_var, x.device)
        z = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
        rc = L.cpg_stem_bn_relu_fwd(ctypes.byref(d), _lib.dptr(x), _lib.dptr(w), _lib.dptr(p), float(thr), None, _lib.dptr(gamma, name='bn.weight'),
                                    _lib.dptr(beta, name='bn.bias'), _lib.dptr(mean), _lib.dptr(invstd), _lib.dptr(z), s)
        _lib.check('cpg_stem_bn_relu_fwd', rc)
        ctx.save_for_backward(x, w, p, gamma, beta, mean, invstd)
        ctx.desc, ctx.thr, ctx.tiles = d, float(thr), tiles
        return z

    @staticmethod
    def backward(ctx, gz):
        x, w, p, gamma, beta, mean, invstd = ctx.saved_tensors
        d, thr, tiles = ctx.desc, ctx.thr, ctx.tiles
        gz = gz.contiguous()
        L = _lib.lib()
        s = _lib.stream_ptr()
        N, K, H, W = x.shape[0], w.shape[0], x.shape[2], x.shape[3]
        partials = torch.empty((K, tiles, 2), dtype=torch.float32, device=x.device)
        args = (ctypes.byref(d), _lib.dptr(x), _lib.dptr(w), _lib.dptr(p), thr, None, _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(mean),
                _lib.dptr(invstd))
        rc = L.cpg_stem_bn_relu_bwd_reduce(*args, _lib.dptr(gz, name='grad_output'), _lib.dptr(partials), partials.numel() * 4, s)
        _lib.check('cpg_stem_bn_relu_bwd_reduce', rc)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        coef = torch.empty(2 * K, dtype=torch.float32, device=x.device)
        rc = L.cpg_bn_bwd_finalize_partials(_lib.dptr(partials), tiles, N, K, H * W, _lib.dptr(dgamma), _lib.dptr(dbeta), _lib.dptr(coef), s)
        _lib.check('cpg_bn_bwd_finalize_partials', rc)
        gw = gpm = None
        if ctx.needs_input_grad[1] or (p is not None and ctx.needs_input_grad[2]):
            gw = torch.empty_like(w)
            gpm = None if p is None else torch.empty_like(p)
            if FUSE_STEM_WGRAD:
                # gy is contracted with the image patch in the pass that forms it (never written)
                ws, nbytes = _lib.workspace(L.cpg_stem_bn_wgrad_workspace(ctypes.byref(d)), x.device)
                rc = L.cpg_stem_bn_relu_bwd_wgrad(*args, _lib.dptr(coef), _lib.dptr(gz), _lib.dptr(gw), _lib.dptr(gpm), _lib.dptr(ws), nbytes, s)
                _lib.check('cpg_stem_bn_relu_bwd_wgrad', rc)
            else:
                gy = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
                rc = L.cpg_stem_bn_relu_bwd_apply(*args, _lib.dptr(coef), _lib.dptr(gz), _lib.dptr(gy), s)
                _lib.check('cpg_stem_bn_relu_bwd_apply', rc)
                ws, nbytes = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), x.device)
                rc = L.cpg_conv2d_wgrad(ctypes.byref(d), _lib.dptr(x), _lib.dptr(gy), _lib.dptr(w), _lib.dptr(p), thr, _lib.dptr(gw), _lib.dptr(gpm),
                                        None, _lib.dptr(ws), nbytes, s)
                _lib.check('cpg_conv2d_wgrad', rc)
        return None, gw, gpm, None, dgamma, dbeta, None, None, None, None


def stem_conv_bn_relu(conv, bn, x):
    """relu(bn(conv(x))) through _StemConvBnReluFn when the triple qualifies (a bias-free masked 3x3 s1 p1 stem in fp32 feeding a
    training-mode, stat-tracking BatchNorm2d, an input that needs no gradient), else None."""
    from .layers import _conv_desc
    if not (ENABLED and FUSE_STEM and FusedSequential.fuse and FusedSequential.fuse_stats and hasattr(conv, 'forward_with_bn_stats')
            and isinstance(bn, nn.BatchNorm2d) and bn.training and bn.track_running_stats and bn.affine and bn.momentum is not None
            and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and torch.is_grad_enabled() and not x.requires_grad
            and conv.bias is None and conv._math() == 'fp32' and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and x.shape[0] > 0
            and x.shape[1] == conv.weight.shape[1] and bn.weight.dtype == torch.float32):
        return None
    d = _conv_desc(x.shape, conv.weight.shape, (1, 1), (1, 1), (1, 1), 1)
    if not _lib.lib().cpg_stem_bn_supported(ctypes.byref(d)):
        return None
    if bn.num_batches_tracked is not None:
        _count_batch(bn)
    out = _StemConvBnReluFn.apply(x, conv.weight, conv.piggymask, conv.info['threshold'], bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                  bn.eps, bn.momentum)
    _settle_count()
    return out


FUSE_SKIP_ADD = True       # residual blocks: the identity branch's gradient is added in conv1's input-gradient epilogue


def conv_bn_act_skip(conv, bn, act, x):
    """(act(bn(conv(x))), x for the block's second branch -- the identity or the downsample conv): conv_bn_act for the first conv of
    a residual block (models/resnet.py:84-104).  When the conv's input gradient can take an addend (dense 1x1 layers), x is routed
    through the conv's autograd node, which then receives both of x's gradients and sums them in its kernel's epilogue."""
    if (ENABLED and FUSE_SKIP_ADD and hasattr(conv, 'forward_with_skip') and x.is_cuda and torch.is_grad_enabled() and x.requires_grad
            and conv._math() == 'fp32' and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.in_channels % 16 == 0
            and conv.out_channels % 16 == 0):
        y, stats, skip = conv.forward_with_skip(x)
        use_stats = (stats is not None and FusedSequential.fuse_stats and isinstance(bn, nn.BatchNorm2d) and bn.training and bn.track_running_stats
                     and bn.affine and bn.momentum is not None)
        if use_stats and (act is None or type(act) is nn.ReLU) and fusable(bn, y):
            return bn_relu(y, bn, relu=act is not None, stats=stats), skip
        return bn_act(bn, act, y), skip
    return conv_bn_act(conv, bn, act, x), x


def conv_bn_add_act(conv, bn, act, x, res):
    """act(bn(conv(x)) + res): the block tail (models/resnet.py:94-104) with the statistics from the conv's epilogue."""
    y, stats = _conv_with_stats(conv, bn, x)
    return bn_add_act(bn, act, y, res, stats)


class FusedSequential(nn.Sequential):
    """nn.Sequential that runs BatchNorm2d -> ReLU (-> MaxPool2d(2, 2)) groups through the fused kernels.
    Module registration (names, parameters, buffers) is exactly nn.Sequential's."""

    fuse = True
    fuse_pool = True
    fuse_stats = True       # conv -> BatchNorm2d: the conv kernel's epilogue produces the batch-statistics partial sums
    fuse_eval = True        # inference: conv -> BatchNorm2d(eval) -> ReLU as one kernel (BatchNorm folded into the conv epilogue)
    skip_log = None         # diagnostics: set to a list to collect one int32[2] tensor per fused inference conv --
    #                         {1 + last live input channel, output tiles skipped by the dead-channel test}

    def forward(self, input):
        mods = list(self._modules.values())
        i, n = 0, len(mods)
        stats = None            # partial sums of `input`, when the module that produced it was asked for them
        hint = None             # BnBwdHint for the conv that consumes `input`, when `input` came out of a fused BatchNorm -> ReLU
        while i < n:
            m = mods[i]
            if (self.fuse and ENABLED and i + 1 < n and isinstance(m, nn.BatchNorm2d) and isinstance(mods[i + 1], nn.ReLU)
                    and fusable(m, input)):
                if (self.fuse_pool and i + 2 < n and _is_pool2(mods[i + 2]) and input.shape[2] % 2 == 0
                        and input.shape[3] % 2 == 0 and m.track_running_stats):
                    input = bn_relu_pool(input, m, stats)
                    i += 3
                    hint = None
                else:
                    # BatchNorm -> ReLU feeding a masked conv directly: let that conv's input-gradient kernel do this
                    # BatchNorm's backward reduction (only this Sequential knows that nothing else reads the activation)
                    nxt2 = mods[i + 2] if i + 2 < n else None
                    hint = BnBwdHint() if (ENABLE_BWD_HINT and m.training and torch.is_grad_enabled()
                                           and hasattr(nxt2, 'forward_with_bn_stats')) else None
                    input = bn_relu(input, m, relu=True, stats=stats, hint=hint)
                    i += 2
                stats = None
                continue
            if self.fuse and ENABLED and isinstance(m, nn.BatchNorm2d) and fusable(m, input):
                input = bn_relu(input, m, relu=False, stats=stats)       # a lone BatchNorm2d (ResNet shortcut: conv1x1 -> BN)
                i += 1
                stats = None
                continue
            nxt = mods[i + 1] if i + 1 < n else None
            if (self.fuse and self.fuse_eval and ENABLED and not torch.is_grad_enabled() and hasattr(m, 'forward_bn_eval')
                    and isinstance(nxt, nn.BatchNorm2d) and not nxt.training and nxt.track_running_stats and nxt.affine
                    and i + 2 < n and isinstance(mods[i + 2], nn.ReLU) and input.is_cuda
                    and not (self.fuse_pool and i + 3 < n and _is_pool2(mods[i + 3]))):
                # (a following MaxPool2d keeps the conv + fused BN/ReLU/pool pair: same HBM traffic, and the un-pooled
                # activation is never written either way)
                st = None
                if self.skip_log is not None:
                    st = torch.zeros(2, dtype=torch.int32, device=input.device)
                y = m.forward_bn_eval(input, nxt, relu=True, skip_stats=st)
                if y is not None:
                    if st is not None:
                        self.skip_log.append(st)
                    input, stats = y, None
                    i += 3
                    continue
            if (self.fuse and FUSE_STEM and i + 2 < n and isinstance(nxt, nn.BatchNorm2d) and type(mods[i + 2]) is nn.ReLU
                    and hasattr(m, 'forward_with_bn_stats') and getattr(m, 'in_channels', 99) <= 3
                    and not (self.fuse_pool and i + 3 < n and _is_pool2(mods[i + 3]))):
                z = stem_conv_bn_relu(m, nxt, input)         # the stem: conv -> BatchNorm2d -> ReLU without ever writing the conv output
                if z is not None:
                    input, stats, hint = z, None, None
                    i += 3
                    continue
            if (self.fuse and self.fuse_stats and ENABLED and hasattr(m, 'forward_with_bn_stats') and isinstance(nxt, nn.BatchNorm2d)
                    and nxt.training and nxt.track_running_stats and nxt.affine and nxt.momentum is not None and input.is_cuda):
                input, stats = m.forward_with_bn_stats(input, bn_hint=hint)
            elif hint is not None and hasattr(m, 'forward_with_bn_stats'):
                input, stats = m(input, bn_hint=hint), None
            else:
                input, stats = m(input), None
            hint = None
            i += 1
        return input
