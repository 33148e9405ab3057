"""Fused BatchNorm2d -> ReLU for the blocks that follow every masked conv of the VGG topologies
(reference: `layers += [conv2d, nn.BatchNorm2d(c), nn.ReLU(inplace=True)]`, models/vgg.py:137-141).

The modules stay what they are in the reference (an `nn.BatchNorm2d` with its parameters / buffers and an
`nn.ReLU`), so state_dict keys, optimizer parameter lists and `shared_layer_info` bookkeeping are
unchanged; only the arithmetic of the pair is replaced by libcpg_hip.so's three-pass forward / five-pass
backward (csrc/bn_kernels.hip).  Semantics are torch.nn.BatchNorm2d's: batch statistics in training mode
(biased variance for the normalisation, unbiased for running_var, momentum update, num_batches_tracked),
running statistics in eval mode.  SURVEY.md section 8(f) item 2.
"""
import torch
import torch.nn as nn

from .. import _lib


class _BnReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, training, relu):
        x = x.contiguous()
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        L = _lib.lib()
        y = torch.empty_like(x)
        s = _lib.stream_ptr()
        if training:
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
        rc = L.cpg_bn_relu_bwd(_lib.dptr(x), _lib.dptr(gy, name='grad_output'), _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(mean),
                               _lib.dptr(invstd), _lib.dptr(gx), _lib.dptr(dgamma), _lib.dptr(dbeta), N, C, HW, int(relu),
                               int(training), _lib.dptr(ws), nb, _lib.stream_ptr())
        _lib.check('cpg_bn_relu_bwd', rc)
        return gx, dgamma, dbeta, None, None, None, None, None, None


def fusable(bn, x):
    """An affine, stat-tracking (or eval-mode) BatchNorm2d on a 4-D fp32 HIP tensor with the default
    exponential-average momentum -- everything the CPG topologies construct."""
    return (isinstance(bn, nn.BatchNorm2d) and bn.affine and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and bn.momentum is not None and (bn.track_running_stats or bn.training))


def bn_relu(x, bn, relu=True):
    """y = relu(bn(x)) with `bn` an nn.BatchNorm2d module (its buffers are updated as torch would)."""
    training = bn.training or not bn.track_running_stats
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _BnReluFn.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, training, relu)


class FusedSequential(nn.Sequential):
    """nn.Sequential that runs [masked conv] -> BatchNorm2d -> ReLU triples through the fused kernels.
    Module registration (names, parameters, buffers) is exactly nn.Sequential's."""

    fuse = True

    def forward(self, input):
        mods = list(self._modules.values())
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            if (self.fuse and i + 1 < n and isinstance(m, nn.BatchNorm2d) and isinstance(mods[i + 1], nn.ReLU)
                    and fusable(m, input)):
                input = bn_relu(input, m, relu=True)
                i += 2
                continue
            input = m(input)
            i += 1
        return input
