"""Flat, bucketed gradient all-reduce for the data-parallel step.

The reference wraps ``G.mapping``, ``G.synthesis``, ``G.const_encoding``, ``G.style_encoding`` and ``D`` in five
``DistributedDataParallel`` instances with ``find_unused_parameters=True`` (training_loop_wo_flow_fullbody.py:316-324)
only so that ``misc.ddp_sync`` can gate their all-reduces one by one (misc.py:172-179).  On MI355X the exchange is a
few large RCCL ring all-reduces over xGMI (per-link bound, so few and large beats many and small), and the host thread
that issues the step's ~3000 launches has no time to spare for five reducers' autograd-graph walks.  ``FlatGradReducer``
is the replacement: ONE reducer per optimised module (G, D):

* the gradients of all parameters live as views of a few flat fp32 buckets (default 64 MiB, filled in reverse
  registration order = roughly the order backward produces them);
* ``begin()`` zeroes the buckets and points every ``param.grad`` at its view (autograd then accumulates in place);
* ``arm(expected)`` is called before the last accumulation round of a phase (the reference's ``sync=True`` round);
  a post-accumulate hook counts the gradients that arrive and launches ``all_reduce(bucket, async_op=True)`` the
  moment a bucket is complete, so the exchange overlaps the rest of backward;
* ``finish()`` launches what is left (buckets holding parameters that took no gradient this phase), waits, and sets
  ``grad = None`` on parameters no backward pass reached - which is what the reference's optimiser sees for them
  (Adam skips them; a zero gradient would still advance its moments).

Ranks execute the same phases on the same graph structure, so the set of parameters reached is the same on every rank
and no "used" bitmap is exchanged.  Sum -> mean is folded in as a pre-scale by 1/world (identical on gloo and RCCL).
"""

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ('flat', 'params', 'pending', 'work', 'touched', 'launched')

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.pending, self.work, self.touched, self.launched = 0, None, False, False


class FlatGradReducer:
    def __init__(self, module, world_size, bucket_mb=64, process_group=None, force_collective=False):
        self.world = int(world_size)
        self.group = process_group
        # force_collective: issue the asynchronous all-reduce even at world size 1 (a one-rank sum is the identity), so a
        # single-GPU box can put RCCL's stream, the pre-scale and work.wait() under test
        self.collective = self.world > 1 or bool(force_collective)
        self.collectives_issued = 0
        params = [p for p in module.parameters()]
        assert params, 'FlatGradReducer: module has no parameters'
        assert all(p.dtype == torch.float32 for p in params), 'FlatGradReducer: fp32 master parameters expected'
        device = params[0].device
        cap = max(int(bucket_mb * (1 << 20)) // 4, 1)
        self.buckets, self._where, self._view = [], {}, {}
        group, size = [], 0
        for p in reversed(params):
            if group and size + p.numel() > cap:
                self._close(group, size, device)
                group, size = [], 0
            group.append(p)
            size += p.numel()
        self._close(group, size, device)
        self._armed = False
        self._expected = 1
        self._touched = set()
        flags = [p.requires_grad for p in params]       # hooks can only be attached while a tensor requires grad
        for p in params:
            p.requires_grad_(True)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        for p, f in zip(params, flags):
            p.requires_grad_(f)
        self.launched_early = 0        # buckets whose all-reduce started inside backward (diagnostic)

    def _close(self, group, size, device):
        flat = torch.zeros([size], dtype=torch.float32, device=device)
        bucket = _Bucket(flat, [])
        off = 0
        for p in group:
            view = flat[off:off + p.numel()].view(p.shape)
            bucket.params.append((p, view))
            self._where[p] = bucket
            self._view[p] = view
            off += p.numel()
        self.buckets.append(bucket)

    # -- phase protocol -------------------------------------------------------------------------------------------

    def begin(self):
        """Start of a phase: zero the buckets and make every gradient a view of its bucket."""
        self._armed = False
        self._touched.clear()
        for b in self.buckets:
            b.flat.zero_()
            b.work, b.touched, b.launched = None, False, False
            for p, view in b.params:
                p.grad = view

    def arm(self, expected_backward_passes=1):
        """Before the last accumulation round: gradients arriving from now on complete their bucket.
        ``expected_backward_passes`` = how many ``backward()`` calls will reach this module in that round
        (None/0: no overlap, everything is exchanged in ``finish()``)."""
        self._expected = int(expected_backward_passes or 0)
        self._armed = self._expected > 0
        for b in self.buckets:
            b.pending = self._expected * sum(1 for p, _ in b.params if p.requires_grad)

    def _on_grad(self, p):
        self._touched.add(p)
        b = self._where[p]
        if b.launched and self.collective:
            # the bucket left (pre-scaled by 1/world) when its expected count was reached: a gradient arriving now would be
            # added during or after the exchange and the replicas would diverge silently.  The count comes from
            # StyleGAN2Loss.backward_passes - a loss variant that calls backward() once more must say so there.
            raise RuntimeError('FlatGradReducer: a gradient arrived after its bucket was sent: arm(expected_backward_passes) '
                               'was told fewer backward passes than reach this module in the last accumulation round')
        b.touched = True
        view = self._view[p]
        if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
            # autograd replaced the view (out-of-place accumulation, e.g. under create_graph): fold it back
            view.copy_(p.grad)
            p.grad = view
        if self._armed:
            b.pending -= 1
            if b.pending == 0:
                self._launch(b)
                self.launched_early += 1

    def _launch(self, b):
        if b.launched:
            return
        b.launched = True
        if self.collective:
            if self.world > 1:
                b.flat.mul_(1.0 / self.world)
            b.work = dist.all_reduce(b.flat, group=self.group, async_op=True)
            self.collectives_issued += 1

    def finish(self):
        """End of a phase: exchange the buckets not yet sent, wait for all of them, and drop the gradients of
        parameters that no backward pass reached."""
        self._armed = False
        for b in self.buckets:
            if b.touched:
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
            b.launched = False              # the phase is over: a backward pass outside the begin / arm / finish protocol is not "late"
            for p, _ in b.params:
                if p not in self._touched:
                    p.grad = None

    def flat_gradients(self):
        """The buckets that hold at least one live gradient (for the gradient clean-up before the optimiser step)."""
        return [b.flat for b in self.buckets if b.touched]

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def broadcast_module_states(modules, src=0, process_group=None):
    """Rank ``src``'s parameters and buffers to every rank (what the reference gets from the DistributedDataParallel
    constructors, training_loop_wo_flow_fullbody.py:316-324, including the transient wrapper around ``G_ema``): one
    flat broadcast per dtype."""
    by_dtype = {}
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            by_dtype.setdefault(t.dtype, []).append(t.detach())
    for dtype, tensors in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src, group=process_group)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view(t.shape))
            off += t.numel()
