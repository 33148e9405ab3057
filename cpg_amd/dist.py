"""Data parallelism for the CPG hot path: one process per GPU, gradients all-reduced with RCCL over xGMI.

Replaces the reference's single-process `nn.DataParallel(model)` (CPG_cifar100_main_normal.py:199-200),
which re-broadcasts every parameter and piggymask (~537 MB for VGG16@224) to all replicas each step and
reduces gradients onto GPU 0.  Here every rank keeps resident weights / owner masks / optimizer state and
only gradients move: each parameter's gradient is all-reduced (mean) as soon as autograd has produced it,
on RCCL's own stream, overlapping the rest of backward; large tensors go out individually (no staging
copy), small ones (BN, biases, heads) are coalesced into one flat message.

Payload: gradient routing (utils/prune.py:195-211) zeroes, right after the exchange, every weight-gradient slot the current
task does not own and every piggymask-gradient slot outside the older tasks' weights.  With a pruner attached
(`set_gradient_filter`, done by Manager) those slots are not sent at all: the surviving slots of a layer are gathered into a
dense buffer by cpg_pack_owned (ballot / popcount positions, natural order), that buffer is all-reduced and scattered back
(from task 2 on a task owns only what earlier tasks released, so most of the 537 MB stays home).

Semantics kept from the reference:
  * loss is the mean over the GLOBAL batch  <=> mean over ranks of equal-size shard means;
  * BatchNorm uses per-replica batch statistics; running stats are rank 0's (sync_buffers());
  * gradient routing / optimizer step / rank prune / statistics are deterministic functions of
    replicated state and run redundantly on every rank -- no further collective.
The wrapper keeps the `.module` attribute and the `module.` prefix in named_modules(), so owner-mask
dictionaries keyed like the reference's (`module.features.0`, ...) work unchanged.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


class _ChunkedGradient(object):
    """Hand-off between a masked linear layer's weight-gradient kernels and the gradient exchange for ONE very large weight
    (VGG16 features.45: 4096 x 25088 = 411 MB): SharableLinear's backward computes the gradient in `nchunks` blocks of output rows
    (independent GEMMs, each writes its rows of the gradient) and calls ready() after launching each block, which starts that
    block's all-reduce -- a single 411 MB message could only start after the whole weight-gradient kernel and could not overlap its
    own producer; with 4 blocks, 3/4 of the exchange runs under the producer's remaining kernels.  Bits are unchanged: every output
    row is computed by the same instruction sequence whichever block it is launched in.

    The gradient is a FRESH tensor per backward pass that autograd adopts as `p.grad` (no clone): nothing here keeps a reference to
    the tensor itself -- the row blocks handed to RCCL alias its storage through Tensor.set_ (a plain slice would be a view whose
    `_base` pins the tensor, and AccumulateGrad clones a gradient somebody else still holds: round 3 paid a 411 MB clone plus a
    411 MB copy-back per step for that).  The collectives then reduce p.grad's own rows in place."""

    def __init__(self, owner, param, nchunks):
        self.owner, self.param, self.nchunks = owner, param, int(nchunks)
        self.pending = []           # [(work, rows alias, first row)] of the current backward pass
        self.base_ptr = None        # data_ptr of the gradient tensor the pending rows alias

    def active(self):
        """Chunk this step?  Not when the surviving slots are exchanged as a packed buffer (task >= 2: the payload is small), and not
        when a gradient is already there (accumulation over several backward passes: autograd ADDS into p.grad, so the rows on the
        wire would not be p.grad's -- the whole-tensor path reduces the accumulated gradient correctly)."""
        o = self.owner
        if self.pending:
            # A weight used TWICE in one graph: its second backward call arrives while the first call's row blocks are on the wire and before
            # AccumulateGrad has set p.grad.  Autograd will SUM the two gradients, so the first call's rows are joined NOW (they then hold the
            # global mean, identical on every rank) and this call takes the whole-tensor path: the parameter's gradient hook all-reduces
            # mean(g1) + g2_local, whose mean over ranks is mean(g1) + mean(g2).
            self.join()
            return False
        return o._active and self.param.grad is None and (o._filter is None or o._plan(self.param) is None)

    def join(self):
        o = self.owner
        for work, rows, _ in self.pending:
            work.wait()
            if not o._avg:
                rows.mul_(1.0 / o._world)
        self.pending = []

    def buffer(self, like):
        if self.pending:
            raise RuntimeError('cpg_amd.dist: a chunked gradient exchange is still in flight for this weight (buffer() without active())')
        gw = torch.empty_like(like, memory_format=torch.contiguous_format)
        self.base_ptr = gw.data_ptr()
        return gw

    def ready(self, gw, r0, r1):
        o = self.owner
        op = dist.ReduceOp.AVG if o._avg else dist.ReduceOp.SUM
        cols = gw.shape[1]
        rows = torch.empty(0, dtype=gw.dtype, device=gw.device).set_(gw.untyped_storage(), gw.storage_offset() + r0 * cols,
                                                                    (r1 - r0, cols), (cols, 1))      # alias, no `_base`
        self.pending.append((dist.all_reduce(rows, op=op, group=o.process_group, async_op=True), rows, r0))
        o.bucket_log.append(('chunk', rows.numel() * 4))


class DataParallel(nn.Module):
    def __init__(self, module, process_group=None, large_numel=1 << 20, broadcast_init=True, chunk_numel=1 << 26, nchunks=4):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.large_numel = int(large_numel)
        self.chunk_numel, self.nchunks = int(chunk_numel), int(nchunks)       # weights at least this large go out in `nchunks` row blocks
        self.bucket_log = []             # (kind, bytes) of every message of the current step, in launch order (bench.py reports it)
        self.shared_chip_bytes = 256 << 20     # gradient volume from which RCCL's kernels are a noticeable share of the backward
        self.shared_chip_hint = 0
        self.last_bucket_log = []
        # CPG_DP_FORCE=1: run the hooks / collectives even at world size 1 (functional test of the RCCL path on one GPU)
        self._active = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(process_group) > 1 or os.environ.get('CPG_DP_FORCE') == '1')
        self._world = dist.get_world_size(process_group) if self._active else 1
        if self._active and self._world > 1 and any(p.is_cuda for p in module.parameters()):
            # the gradient all-reduce runs on RCCL's stream beside the backward kernels and holds some CUs while a message is on the
            # wire.  When that is a noticeable share of the backward (VGG16: 537 MB of gradients, ~3 ms of an 8-GPU step) tell the
            # library (a process-wide hint of its C ABI -- the planners that read it run on autograd's engine thread, not on this
            # one): the Winograd weight gradient then runs two rounds of half-length units per wave slot, so that a launch that finds
            # CUs taken grows by half instead of doubling (conv3x3_wino_wgrad.hip).  Networks whose whole exchange lasts a few
            # hundred microseconds (ResNet-50 94 MB, SphereNet-20 56 MB) keep the idle-chip plans: beside RCCL's real kernels the
            # extra rounds cost more than the exchange lasts (profiles/r04_ab_shared_chip_plans.txt).  CPG_DP_SHARED_CHIP=0 / 1
            # overrides.
            from . import _lib
            grad_bytes = sum(p.numel() * p.element_size() for p in module.parameters() if p.requires_grad)
            want = os.environ.get('CPG_DP_SHARED_CHIP')
            self.shared_chip_hint = int(want) if want in ('0', '1') else int(grad_bytes >= self.shared_chip_bytes)
            _lib.lib().cpg_set_shared_chip_hint(self.shared_chip_hint)
        self._handles = []
        self._small = []
        self._chunked = []               # [(param, _ChunkedGradient)] whose row blocks are in flight
        # marks the Parameters this wrapper has hooked: an attribute on the Parameter itself, which dies with it.  (A set of
        # id()s does not work: piggymasks are re-created between phases and CPython reuses the ids of the freed ones, so new
        # Parameters looked "already hooked" and their gradients were silently left un-reduced.)
        self._token = object()
        self.sync_events = None          # list of (start, end) HIP events around finish_gradient_sync() when timing is on
        self._filter = None              # SparsePruner whose owner masks say which gradient slots survive routing
        self.compact_below = 0.5         # compact a layer's gradient when at most this share of its slots survives
        self.last_payload = {'dense_elems': 0, 'sent_elems': 0}     # per step: what a dense exchange would send / what was sent
        self._step_payload = {'dense_elems': 0, 'sent_elems': 0}
        # RCCL averages inside the collective; gloo (CPU tests) has no AVG, there the sum is scaled afterwards
        self._avg = self._active and dist.get_backend(process_group) == 'nccl'
        if self._active and broadcast_init:
            self.sync_parameters()
        self._install_hooks()

    # -- wiring -------------------------------------------------------------------------------
    def _install_hooks(self):
        """(Re)attach the post-accumulate hooks; call again after assigning new Parameters
        (e.g. piggymasks created after wrapping, CPG_cifar100_main_normal.py:263-270)."""
        if not self._active:
            return
        for p in self.module.parameters():
            if p.requires_grad and getattr(p, '_cpg_dp_token', None) is not self._token:
                p.register_post_accumulate_grad_hook(self._on_grad)
                p._cpg_dp_token = self._token
        # which owner mask governs a parameter: (mask key, select) with select 0 = weight, 1 = piggymask
        for name, m in self.named_modules():
            if hasattr(m, 'piggymask') and hasattr(m, 'weight'):
                m.weight._cpg_mask_key = (name, 0)
                if m.piggymask is not None:
                    m.piggymask._cpg_mask_key = (name, 1)
                # a very large linear weight: its backward hands the gradient over in row blocks (see _ChunkedGradient)
                if (m.weight.dim() == 2 and m.weight.numel() >= self.chunk_numel and m.weight.shape[0] % self.nchunks == 0
                        and getattr(m.weight, '_cpg_dp_chunk', None) is None):
                    m.weight._cpg_dp_chunk = _ChunkedGradient(self, m.weight, self.nchunks)

    refresh_hooks = _install_hooks

    def set_gradient_filter(self, pruner):
        """Attach (or, with None, detach) the SparsePruner whose owner masks decide which gradient slots are exchanged.
        Only safe when gradient routing runs after finish_gradient_sync() -- Manager.train does both."""
        self._filter = pruner
        for p in self.module.parameters():          # a new pruner restarts its mutation counter: never reuse the old one's plans
            if hasattr(p, '_cpg_pack_plan'):
                del p._cpg_pack_plan

    def _plan(self, p):
        """None: exchange the whole gradient.  Otherwise (owner, cur, select, block offsets, total): exchange only the
        `total` slots that survive routing; total == 0: nothing of this gradient survives."""
        pr = self._filter
        key = getattr(p, '_cpg_mask_key', None)
        if pr is None or key is None or not p.is_cuda or p.dtype != torch.float32:
            return None
        name, select = key
        owner = pr.masks.get(name)
        mode = getattr(pr.args, 'mode', None)
        if owner is None or owner.numel() != p.numel() or mode not in ('finetune', 'prune'):
            return None
        cur = int(pr.current_dataset_idx)
        if select == 1 and mode == 'prune':
            return (owner, cur, select, None, 0)                  # routing zeroes every piggymask gradient in prune mode
        ckey = (pr._epoch, id(owner), owner._version, pr._mutations, cur, select)
        cached = getattr(p, '_cpg_pack_plan', None)
        if cached is not None and cached[0] == ckey:
            return cached[1]
        from . import _lib
        import ctypes
        L = _lib.lib()
        owner = pr._owner(name, p.data)
        nblk = int(L.cpg_owned_num_blocks(p.numel()))
        counts = torch.empty(nblk, dtype=torch.int32, device=p.device)
        rc = L.cpg_owned_block_counts(_lib.dptr(owner, torch.uint8, 'mask'), cur, select, p.numel(), ctypes.c_void_p(counts.data_ptr()),
                                      _lib.stream_ptr())
        _lib.check('cpg_owned_block_counts', rc)
        ends = torch.cumsum(counts, 0, dtype=torch.int64)
        total = int(ends[-1].item())                              # one read-back per mask mutation, then cached
        plan = None if total > self.compact_below * p.numel() else (owner, cur, select, (ends - counts).contiguous(), total)
        p._cpg_pack_plan = (ckey, plan)
        return plan

    def _on_grad(self, p):
        if p.grad is None:
            return
        ch = getattr(p, '_cpg_dp_chunk', None)
        if ch is not None and ch.pending:
            # already on the wire, block by block, since the weight-gradient kernels were launched; finish_gradient_sync() joins
            self._step_payload['dense_elems'] += p.numel()
            self._step_payload['sent_elems'] += p.numel()
            self._chunked.append((p, ch))
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        plan = self._plan(p) if self._filter is not None else None
        self._step_payload['dense_elems'] += p.numel()
        if plan is not None:
            owner, cur, select, offsets, total = plan
            if total == 0:
                return                                            # routing will zero all of it: nothing to exchange
            from . import _lib
            import ctypes
            g = p.grad
            if not g.is_contiguous():
                p.grad = g = g.contiguous()
            buf = torch.empty(total, dtype=torch.float32, device=g.device)
            L = _lib.lib()
            rc = L.cpg_pack_owned(_lib.dptr(g, name='grad'), _lib.dptr(owner, torch.uint8, 'mask'), cur, select, g.numel(),
                                  ctypes.c_void_p(offsets.data_ptr()), _lib.dptr(buf), _lib.stream_ptr())
            _lib.check('cpg_pack_owned', rc)
            self._step_payload['sent_elems'] += total
            self.bucket_log.append(('packed', total * 4))
            work = dist.all_reduce(buf, op=op, group=self.process_group, async_op=True)
            self._handles.append((work, buf, (g, owner, cur, select, offsets)))
            return
        self._step_payload['sent_elems'] += p.numel()
        if p.numel() >= self.large_numel:
            g = p.grad
            if not g.is_contiguous():
                p.grad = g = g.contiguous()
            self.bucket_log.append(('tensor', g.numel() * 4))
            self._handles.append((dist.all_reduce(g, op=op, group=self.process_group, async_op=True), g, None))
        else:
            self._small.append(p)

    # -- public API ---------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def finish_gradient_sync(self):
        """Block the current stream until every gradient holds the global-batch mean.  Call once after
        backward() and before gradient routing / optimizer.step() (Manager.train does)."""
        if not self._active:
            return
        ev = None
        if self.sync_events is not None and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        inv = 1.0 / self._world
        if self._small:
            grads = [p.grad for p in self._small]
            flat = _flatten_dense_tensors(grads)
            self.bucket_log.append(('coalesced', flat.numel() * 4))
            if self._avg:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.process_group)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.process_group)
                flat.mul_(inv)
            torch._foreach_copy_(grads, list(_unflatten_dense_tensors(flat, grads)))   # (one multi-tensor kernel, not one copy per tensor)
            self._small = []
        for work, g, packed in self._handles:
            work.wait()
            if not self._avg:
                g.mul_(inv)
            if packed is not None:                                # scatter the reduced survivors back into the gradient
                from . import _lib
                import ctypes
                grad, owner, cur, select, offsets = packed
                rc = _lib.lib().cpg_unpack_owned(_lib.dptr(g), _lib.dptr(owner, torch.uint8, 'mask'), cur, select, grad.numel(),
                                                 ctypes.c_void_p(offsets.data_ptr()), _lib.dptr(grad, name='grad'), _lib.stream_ptr())
                _lib.check('cpg_unpack_owned', rc)
        self._handles = []
        for p, ch in self._chunked:
            adopted = p.grad.data_ptr() == ch.base_ptr           # autograd took the gradient tensor itself: the rows ARE p.grad's
            for work, rows, r0 in ch.pending:
                work.wait()
                if not self._avg:
                    rows.mul_(inv)
                if not adopted:                                   # (it cloned: put the reduced rows where the optimizer reads)
                    p.grad[r0:r0 + rows.shape[0]].copy_(rows)
            ch.pending = []
        self._chunked = []
        # Row blocks of a backward pass that kept NO gradient for their weight (the hook above never saw the parameter, so it is not in
        # _chunked): wait for them here and drop them -- left pending they would only be drained by the weight's next active() call,
        # which silently puts that step on the whole-tensor path.
        for q in self.module.parameters():
            ch = getattr(q, '_cpg_dp_chunk', None)
            if ch is not None and ch.pending:
                for work, _, _ in ch.pending:
                    work.wait()
                ch.pending = []
        self.last_bucket_log, self.bucket_log = self.bucket_log, []
        self.last_payload, self._step_payload = self._step_payload, {'dense_elems': 0, 'sent_elems': 0}
        if ev is not None:
            ev[1].record()
            self.sync_events.append(ev)

    def sync_parameters(self, src=0):
        """Make every rank start from rank `src`'s parameters and buffers."""
        if not self._active:
            return
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src=src, group=self.process_group)

    def sync_buffers(self, src=0):
        """BatchNorm running statistics: keep rank 0's, as the reference's replica 0 does."""
        if not self._active:
            return
        by_dtype = {}
        for b in self.module.buffers():
            by_dtype.setdefault(b.dtype, []).append(b.data)
        for bufs in by_dtype.values():          # one flat broadcast per dtype (running_mean / running_var; num_batches_tracked)
            flat = _flatten_dense_tensors(bufs)
            dist.broadcast(flat, src=src, group=self.process_group)
            torch._foreach_copy_(bufs, list(_unflatten_dense_tensors(flat, bufs)))


def seed_per_rank(base=1, process_group=None):
    """Re-seed torch's generators with a rank-dependent seed AFTER the (identically seeded) model construction, so that
    Dropout masks differ between ranks the way nn.DataParallel's replicas draw different masks.
    Returns the seed used."""
    rank = dist.get_rank(process_group) if dist.is_available() and dist.is_initialized() else 0
    seed = int(base) + 7919 * rank
    torch.manual_seed(seed)
    return seed


def shard_batch(data, target, rank=None, world=None):
    """Even split of a global batch over ranks (the scatter of nn.DataParallel, along dim 0)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return data, target
    n = data.size(0)
    if n % world:
        raise ValueError('global batch %d is not divisible by world size %d' % (n, world))
    per = n // world
    return data[rank * per:(rank + 1) * per], target[rank * per:(rank + 1) * per]
