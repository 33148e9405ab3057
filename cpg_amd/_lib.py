"""ctypes binding of libcpg_hip.so (include/cpg_hip.h) -- the only bridge between the Python
mirror of the reference's classes and the HIP kernels.

There is NO CPU fallback: if the shared library is missing or a tensor is not a contiguous
fp32/uint8 HIP tensor, the call raises.  (The CPU oracle under oracle/ is test infrastructure
and is never imported from here.)
"""
import ctypes
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CPG_HIP_LIB') or os.path.join(_HERE, 'lib', 'libcpg_hip.so')   # override: A/B kernel experiments

ABI_VERSION = 3         # include/cpg_hip.h: CPG_ABI_VERSION
CPG_OK = 0
CPG_E_KRANGE = 2
MODE_FINETUNE = 0
MODE_PRUNE = 1

_c_f32p = ctypes.c_void_p
_vp = ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ('N', 'C', 'H', 'W', 'K', 'R', 'S', 'stride_h', 'stride_w', 'pad_h', 'pad_w', 'dil_h', 'dil_w', 'groups')]


class PruneResult(ctypes.Structure):
    _fields_ = [('n_candidates', ctypes.c_int64), ('k', ctypes.c_int64), ('n_released', ctypes.c_int64),
                ('cutoff', ctypes.c_float), ('status', ctypes.c_int32)]


class SgdItem(ctypes.Structure):                  # cpg_sgd_item
    _fields_ = [('w', ctypes.c_void_p), ('gw', ctypes.c_void_p), ('momentum_buf', ctypes.c_void_p), ('owner', ctypes.c_void_p),
                ('n', ctypes.c_int64)]


class AdamItem(ctypes.Structure):                 # cpg_adam_item
    _fields_ = [('pm', ctypes.c_void_p), ('gpm', ctypes.c_void_p), ('exp_avg', ctypes.c_void_p), ('exp_avg_sq', ctypes.c_void_p),
                ('owner', ctypes.c_void_p), ('n', ctypes.c_int64)]


PRUNE_RESULT_BYTES = ctypes.sizeof(PruneResult)
assert PRUNE_RESULT_BYTES == 32

# name -> (restype, argtypes); mirrors include/cpg_hip.h one to one
_SIGNATURES = {
    'cpg_version': (ctypes.c_int, []),
    'cpg_set_shared_chip_hint': (ctypes.c_int, [ctypes.c_int32]),
    'cpg_get_shared_chip_hint': (ctypes.c_int32, []),
    'cpg_set_option': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int32]),
    'cpg_get_option': (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32)]),
    'cpg_last_error': (ctypes.c_char_p, []),
    'cpg_binarize_mask_weight': (ctypes.c_int, [_vp, _vp, ctypes.c_float, _vp, ctypes.c_int64, _vp]),
    'cpg_conv2d_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    'cpg_conv2d_fwd': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_dgrad': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_dgrad_add_supported': (ctypes.c_int32, [ctypes.POINTER(ConvDesc)]),
    'cpg_conv2d_dgrad_add': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_wgrad': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_pack_bytes': (ctypes.c_size_t, [ctypes.POINTER(ConvDesc), ctypes.c_int32]),
    'cpg_conv2d_pack': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, ctypes.c_float, ctypes.c_int32, _vp, ctypes.c_size_t, ctypes.c_int32, _vp,
                                       ctypes.c_size_t, _vp]),
    'cpg_conv2d_use_packed': (ctypes.c_int, [_vp, ctypes.c_size_t]),
    'cpg_linear_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'cpg_linear_fwd': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t, _vp]),
    'cpg_linear_dgrad': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t, _vp]),
    'cpg_linear_wgrad': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t, _vp]),
    'cpg_route_grads': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, ctypes.c_float, _vp, ctypes.c_int32, ctypes.c_int64, _vp]),
    'cpg_rank_prune_workspace_bytes': (ctypes.c_size_t, []),
    'cpg_rank_prune': (ctypes.c_int, [_vp, _vp, ctypes.c_int32, ctypes.c_double, ctypes.c_int64, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_mask_hist': (ctypes.c_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int64, _vp, _vp]),
    'cpg_apply_mask': (ctypes.c_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int64, _vp]),
    'cpg_zero_pruned': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp]),
    'cpg_claim_free': (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int64, _vp]),
    'cpg_owned_num_blocks': (ctypes.c_int64, [ctypes.c_int64]),
    'cpg_owned_block_counts': (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, _vp, _vp]),
    'cpg_pack_owned': (ctypes.c_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, _vp, _vp, _vp]),
    'cpg_unpack_owned': (ctypes.c_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, _vp, _vp, _vp]),
    'cpg_sgd_route_step': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, _vp]),
    'cpg_multi_tensor_max': (ctypes.c_int32, []),
    'cpg_sgd_route_step_multi': (ctypes.c_int, [ctypes.POINTER(SgdItem), ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_int32, ctypes.c_int32, _vp]),
    'cpg_adam_route_step_multi': (ctypes.c_int, [ctypes.POINTER(AdamItem), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_double,
                                                 ctypes.c_double, ctypes.c_double, ctypes.c_int32, _vp]),
    'cpg_adam_route_step': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_int64, _vp]),
    'cpg_bn_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'cpg_bn_relu_fwd_train': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t, _vp]),
    'cpg_bn_relu_fwd_eval': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int32, _vp]),
    'cpg_bn_relu_pool_fwd': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t, _vp]),
    'cpg_bn_relu_pool_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t, _vp]),
    'cpg_bn_relu_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_winograd': (ctypes.c_int32, [ctypes.POINTER(ConvDesc), ctypes.c_int32]),
    'cpg_conv2d_bnstats_tiles': (ctypes.c_int32, [ctypes.POINTER(ConvDesc)]),
    'cpg_conv2d_fwd_bnstats': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, ctypes.c_size_t, _vp,
                                              ctypes.c_size_t, _vp]),
    'cpg_conv2d_fwd_bn_eval_supported': (ctypes.c_int32, [ctypes.POINTER(ConvDesc)]),
    'cpg_conv2d_fwd_bn_eval': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, ctypes.c_float,
                                              ctypes.c_int32, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_bf16_supported': (ctypes.c_int32, [ctypes.POINTER(ConvDesc)]),
    'cpg_conv2d_bf16_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    'cpg_conv2d_fwd_bf16': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_wgrad_bf16_supported': (ctypes.c_int32, [ctypes.POINTER(ConvDesc)]),
    'cpg_conv2d_wgrad_bf16_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    'cpg_conv2d_wgrad_bf16': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_fwd_bf16x3': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_dgrad_bf16x3': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_wgrad_bf16x3': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_dgrad_bf16': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_conv2d_dgrad_bnbwd_tiles': (ctypes.c_int32, [ctypes.POINTER(ConvDesc)]),
    'cpg_conv2d_dgrad_bnbwd': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                              ctypes.c_size_t, _vp, ctypes.c_size_t, _vp]),
    'cpg_bn_bwd_from_partials': (ctypes.c_int, [_vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, _vp, ctypes.c_size_t, _vp]),
    'cpg_bn_bwd_finalize_partials': (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp]),
    'cpg_stem_bn_supported': (ctypes.c_int32, [ctypes.POINTER(ConvDesc)]),
    'cpg_stem_bn_tiles': (ctypes.c_int32, [ctypes.POINTER(ConvDesc)]),
    'cpg_stem_bn_stats': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_stem_bn_relu_fwd': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'cpg_stem_bn_relu_bwd_reduce': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                   ctypes.c_size_t, _vp]),
    'cpg_stem_bn_relu_bwd_apply': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                  _vp, _vp]),
    'cpg_stem_bn_wgrad_workspace': (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    'cpg_stem_bn_relu_bwd_wgrad': (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                  _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'cpg_bn_stats_finalize_count': (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                                   ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    'cpg_bn_stats_finalize': (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                             ctypes.c_float, _vp, _vp, _vp, _vp, _vp]),
    'cpg_bn_add_relu_mask_bytes': (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'cpg_bn_add_relu_fwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_size_t, _vp, _vp]),
    'cpg_bn_relu_pool3_supported': (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32]),
    'cpg_bn_relu_pool3_fwd': (ctypes.c_int, [_vp] * 6 + [ctypes.c_int32] * 4 + [_vp]),
    'cpg_bn_relu_pool3_bwd': (ctypes.c_int, [_vp] * 9 + [ctypes.c_int32] * 5 + [_vp, ctypes.c_size_t, _vp]),
    'cpg_bn_add_relu_bwd': (ctypes.c_int, [_vp] * 11 + [ctypes.c_int32] * 4 + [_vp, ctypes.c_size_t, _vp, _vp]),
    'cpg_prelu_fwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp]),
    'cpg_prelu_bwd_bias': (ctypes.c_int, [_vp] * 6 + [ctypes.c_int32] * 4 + [_vp, ctypes.c_size_t, _vp]),
    'cpg_prelu_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    'cpg_prelu_bwd': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp,
                                     ctypes.c_size_t, _vp]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class CpgHipError(RuntimeError):
    def __init__(self, fn, code, text):
        super().__init__('%s failed with status %d: %s' % (fn, code, text))
        self.code = code


def lib():
    """Load libcpg_hip.so once; raise (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'cpg_amd: %s is missing -- build it with `python -m cpg_amd.build` (hipcc, gfx950). '
                'There is no CPU fallback for the masked-layer / prune path.' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if handle.cpg_version() != ABI_VERSION:
            raise RuntimeError('cpg_amd: ABI version mismatch (library %d, binding %d)' % (handle.cpg_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(fn_name, code):
    if code != CPG_OK:
        text = lib().cpg_last_error()
        raise CpgHipError(fn_name, code, text.decode(errors='replace') if text else '')


OPT_UNSET = -(1 << 31)
_WINO_KERNEL = {'block': 0, 'wave': 1, 'pair': 2, '64': 3}


def get_option(name):
    """Current value of a library switch (include/cpg_hip.h: cpg_get_option); None when it was never given."""
    v = ctypes.c_int32(0)
    check('cpg_get_option', lib().cpg_get_option(name.encode(), ctypes.byref(v)))
    return None if v.value == OPT_UNSET else v.value


def set_option(name, value):
    """Set (value None: unset) a library switch through the C ABI.  The table is filled from the environment once, at load time; this is
    the only way to change it afterwards.  CPG_WINO_KERNEL also takes its environment spelling ('block' | 'wave' | 'pair' | '64')."""
    if isinstance(value, str):
        value = _WINO_KERNEL[value] if name == 'CPG_WINO_KERNEL' else int(value)
    check('cpg_set_option', lib().cpg_set_option(name.encode(), OPT_UNSET if value is None else int(value)))
    mod = sys.modules.get('cpg_amd.models.layers')
    if mod is not None:
        mod._PACK_BYTES.clear()             # which kernel family (hence which packed operand) a shape gets depends on the switches


class option(object):
    """`with option('CPG_NO_WINO', 1): ...` -- a library switch for the duration of a block (tests, A/B tools)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def stream_ptr():
    """hipStream_t of torch's current stream on the current device."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t, dtype=torch.float32, name='tensor'):
    """Device pointer of a contiguous HIP tensor of the expected dtype (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('cpg_amd: %s lives on %s; the masked-layer kernels only run on a HIP device '
                           '(no CPU fallback -- move the module with .cuda())' % (name, t.device))
    if t.dtype != dtype:
        raise TypeError('cpg_amd: %s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError('cpg_amd: %s must be contiguous' % name)
    return ctypes.c_void_p(t.data_ptr())


def workspace(nbytes, device):
    """Scratch buffer from torch's caching allocator (stream-ordered reuse, no hipMalloc in steady state)."""
    if nbytes == 0:
        return None, 0
    buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    return buf, buf.numel() * 4


def _selftest():
    h = lib()
    print('libcpg_hip.so ABI', h.cpg_version(), 'exports', len(EXPORTS))


if __name__ == '__main__':
    _selftest()
    sys.exit(0)
