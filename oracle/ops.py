"""CPU oracle for the CPG masked-layer / prune hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product (cpg_amd/) never does and fails loudly without its HIP library.

Every function restates one piece of ivclab/CPG (file:line cited per function) as a
pure function over arrays: numpy for the byte/integer/mask arithmetic, torch-CPU fp32
conv/matmul for the floating-point contractions (the reference itself delegates those
to torch).

Parity status: PINNED against outputs of the reference itself, run in the build
container under torch 2.10.0 CPU (tests/golden/*.npz, produced by
tests/golden/make_golden.py, checked by tests/test_oracle_golden.py).  Against the
reference authors' original torch-1.x/cuDNN environment parity is unpinned (the
reference ships no tests or golden vectors of its own).
"""
import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_THRESHOLD = 5e-3          # models/layers.py:9


class NotEnoughWeights(Exception):
    """utils/prune.py:38-42 -- kthvalue fails (k == 0 or k > n) -> reference sys.exit(2)."""
    exit_code = 2


# ---------------------------------------------------------------------------
# masked operators (models/layers.py)
# ---------------------------------------------------------------------------
def binarize(pm, threshold=DEFAULT_THRESHOLD):
    """models/layers.py:14-19.  fp32 compare; x <= thr -> 0, x > thr -> 1, NaN stays NaN."""
    pm = np.asarray(pm, dtype=np.float32)
    out = pm.copy()
    # torch compares an fp32 tensor against a python scalar in fp32 (the scalar is cast)
    t32 = np.float32(threshold)
    out[pm <= t32] = 0.0
    out[pm > t32] = 1.0
    return out


def effective_weight(w, pm=None, threshold=DEFAULT_THRESHOLD):
    """models/layers.py:99-105 / 185-192: W_eff = bin(pm) * W, or W when there is no piggymask."""
    w = np.asarray(w, dtype=np.float32)
    if pm is None:
        return w
    return binarize(pm, threshold) * w


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a


def conv2d_forward(x, w, pm=None, bias=None, stride=1, padding=0, dilation=1, groups=1,
                   threshold=DEFAULT_THRESHOLD):
    """models/layers.py:98-109 (SharableConv2d.forward)."""
    weff = _t(effective_weight(np.asarray(w), None if pm is None else np.asarray(pm), threshold))
    y = F.conv2d(_t(np.asarray(x, dtype=np.float32)), weff, None if bias is None else _t(np.asarray(bias)),
                 stride, padding, dilation, groups)
    return y.numpy()


def conv2d_backward(x, w, gy, pm=None, has_bias=False, stride=1, padding=0, dilation=1, groups=1,
                    threshold=DEFAULT_THRESHOLD):
    """Autograd of models/layers.py:98-109: gx, gW = gW_eff * bin(pm), gPM = gW_eff * W, gb."""
    x = _t(np.asarray(x, dtype=np.float32)).clone().requires_grad_(True)
    weff = _t(effective_weight(np.asarray(w), None if pm is None else np.asarray(pm), threshold)).clone().requires_grad_(True)
    y = F.conv2d(x, weff, None, stride, padding, dilation, groups)
    y.backward(_t(np.asarray(gy, dtype=np.float32)))
    gweff = weff.grad.numpy()
    out = {'gx': x.grad.numpy()}
    if pm is None:
        out['gw'] = gweff
    else:
        out['gw'] = gweff * binarize(pm, threshold)
        out['gpm'] = gweff * np.asarray(w, dtype=np.float32)     # straight-through (layers.py:21-23)
    if has_bias:
        out['gb'] = np.asarray(gy, dtype=np.float32).sum(axis=(0, 2, 3))
    return out


def linear_forward(x, w, pm=None, bias=None, threshold=DEFAULT_THRESHOLD):
    """models/layers.py:184-194 (SharableLinear.forward)."""
    weff = _t(effective_weight(np.asarray(w), None if pm is None else np.asarray(pm), threshold))
    return F.linear(_t(np.asarray(x, dtype=np.float32)), weff, None if bias is None else _t(np.asarray(bias))).numpy()


def linear_backward(x, w, gy, pm=None, threshold=DEFAULT_THRESHOLD):
    x = _t(np.asarray(x, dtype=np.float32))
    gy = _t(np.asarray(gy, dtype=np.float32))
    weff = _t(effective_weight(np.asarray(w), None if pm is None else np.asarray(pm), threshold))
    gweff = (gy.t() @ x).numpy()
    out = {'gx': (gy @ weff).numpy(), 'gb': gy.sum(0).numpy()}
    if pm is None:
        out['gw'] = gweff
    else:
        out['gw'] = gweff * binarize(pm, threshold)
        out['gpm'] = gweff * np.asarray(w, dtype=np.float32)
    return out


# ---------------------------------------------------------------------------
# gradient routing (utils/prune.py:195-211)
# ---------------------------------------------------------------------------
def route_grads(gw, w, owner, cur, weight_decay, gpm=None, mode='finetune'):
    """do_weight_decay_and_make_grads_zero for ONE layer; returns (gw', gpm').

    gw += wd * W over all elements first (prune.py:203), then gw[owner != cur] = 0 (:204-205).
    Piggymask grad: finetune -> zero where owner == 0 or owner >= cur (:207-208); prune -> all
    zero (:209-210)."""
    gw = np.asarray(gw, dtype=np.float32).copy()
    w = np.asarray(w, dtype=np.float32)
    owner = np.asarray(owner)
    gw = gw + np.float32(weight_decay) * w        # torch add_(alpha, tensor): alpha cast to fp32
    gw[owner != cur] = 0.0
    if gpm is not None:
        gpm = np.asarray(gpm, dtype=np.float32).copy()
        if mode == 'finetune':
            gpm[(owner == 0) | (owner >= cur)] = 0.0
        elif mode == 'prune':
            gpm[...] = 0.0
    return gw, gpm


# ---------------------------------------------------------------------------
# schedule + rank prune (utils/prune.py:30-92)
# ---------------------------------------------------------------------------
def adjust_sparsity(step, begin, end, initial, target, exponent=3):
    """utils/prune.py:55-66; python float (fp64) arithmetic, exactly as written there."""
    p = min(1.0, max(0.0, ((step - begin) / (end - begin))))
    return target + (initial - target) * pow(1 - p, exponent)


def time_to_update(step, begin, end, last_prune_step, frequency):
    """utils/prune.py:68-76."""
    return (begin <= step <= end) and (last_prune_step + frequency <= step)


def cutoff_rank(ratio, n_candidates):
    """utils/prune.py:37 -- python round(): round-half-even on the fp64 product."""
    return round(ratio * n_candidates)


def rank_prune(w, owner, cur, ratio):
    """utils/prune.py:30-53 (_pruning_mask) for one layer.

    candidates = owner in {cur, 0}; k = round(ratio * n); cutoff = k-th smallest |w| among the
    candidates (1-indexed); owner[(|W| <= cutoff) & (owner == cur)] = 0.  Returns
    (new_owner, k, cutoff).  Raises NotEnoughWeights when kthvalue would (k < 1 or k > n)."""
    w = np.asarray(w, dtype=np.float32)
    owner = np.asarray(owner, dtype=np.uint8)
    cand = np.abs(w[(owner == cur) | (owner == 0)])
    n = cand.size
    k = cutoff_rank(ratio, n)
    if k < 1 or k > n:
        raise NotEnoughWeights('k=%d of n=%d' % (k, n))
    cutoff = np.partition(cand, k - 1)[k - 1]
    remove = (np.abs(w) <= cutoff) & (owner == cur)
    out = owner.copy()
    out[remove] = 0
    return out, k, cutoff


# ---------------------------------------------------------------------------
# statistics (utils/prune.py:111-193) -- over lists of per-layer arrays
# ---------------------------------------------------------------------------
def owner_histogram(owner):
    return np.bincount(np.asarray(owner, dtype=np.uint8).ravel(), minlength=256).astype(np.int64)


def sparsity(owners, inference_idx):
    """calculate_sparsity (:111-136): #0 / #(0 or idx)."""
    tot = sum(int(((o == inference_idx) | (o == 0)).sum()) for o in owners)
    zero = sum(int((o == 0).sum()) for o in owners)
    return float(zero) / float(tot) if tot != 0 else 0.0


def curr_task_ratio(owners, inference_idx, width_mult):
    """calculate_curr_task_ratio (:138-157)."""
    tot = sum(o.size for o in owners)
    mine = sum(int((o == inference_idx).sum()) for o in owners)
    return float(mine) / tot * (width_mult ** 2)


def zero_ratio(owners, width_mult):
    """calculate_zero_ratio (:159-178)."""
    tot = sum(o.size for o in owners)
    zero = sum(int((o == 0).sum()) for o in owners)
    return float(zero) / tot * (width_mult ** 2)


def shared_part_ratio(owners, piggymasks, inference_idx):
    """calculate_shared_part_ratio (:180-193); note the literal 0.005 (not the layer threshold)."""
    tot = 0
    shared = 0
    for o, pm in zip(owners, piggymasks):
        older = (o > 0) & (o < inference_idx)
        tot += int(older.sum())
        shared += int((older & (np.asarray(pm, dtype=np.float32) > np.float32(0.005))).sum())
    return float(shared) / float(tot) if tot != 0 else 0.0


# ---------------------------------------------------------------------------
# mask application (utils/prune.py:213-243)
# ---------------------------------------------------------------------------
def zero_pruned(w, owner):
    """make_pruned_zero (:213-221)."""
    w = np.asarray(w, dtype=np.float32).copy()
    w[np.asarray(owner) == 0] = 0.0
    return w


def apply_mask(w, owner, inference_idx):
    """apply_mask (:223-231): zero free slots and slots of later tasks."""
    w = np.asarray(w, dtype=np.float32).copy()
    owner = np.asarray(owner)
    w[owner == 0] = 0.0
    w[owner > inference_idx] = 0.0
    return w


def claim_free(owner, new_idx):
    """make_finetuning_mask (:233-243) for one layer: owner[owner == 0] = new_idx."""
    owner = np.asarray(owner, dtype=np.uint8).copy()
    owner[owner == 0] = new_idx
    return owner
