"""Data-parallel gradient synchronisation (replaces apex/torch DistributedDataParallel at
yolo/procedures/initialize.py:48 and torchvision_models/detection/train.py:160).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).  The engine keeps all
gradients in ONE flat fp32 buffer laid out in forward-layer order, and backward produces them in
reverse, so buckets are contiguous tail slices of that buffer: as soon as the layers of a bucket have
issued their weight-gradient kernels, an asynchronous all-reduce of the slice is enqueued (RCCL runs
on its own stream and waits on an event of the compute stream), overlapping communication with the
rest of backward.  Bucket size is tuned for xGMI (point-to-point links, ring per-link bound):
few large buckets (default 64 MiB; the 248 MB YOLOv3 gradient makes 4).
Loss normalisation stays per rank (yolo_forw.py:158-160) and gradients are averaged across ranks,
exactly what DDP does in the reference.
"""
import torch
import torch.distributed as dist


def plan_buckets(marks, total, bucket_elems):
    """marks: [(position, lowest_completed_offset)] in backward order (offsets decreasing).
    -> [(position, lo, hi)]: after `position` calls of the backward list, all-reduce flat[lo:hi]."""
    out, hi = [], total
    for pos, lo in marks:
        if hi - lo >= bucket_elems:
            out.append((pos, lo, hi))
            hi = lo
    if hi > 0:
        last_pos = marks[-1][0] if marks else 0
        if out and out[-1][0] == last_pos:
            pos, lo, h2 = out.pop()
            out.append((pos, 0, h2))
        else:
            out.append((last_pos, 0, hi))
    return out


class GradSync:
    """algo: "all_reduce" (default; RCCL picks ring / tree per message size) or "rs_ag" (explicit reduce-scatter + all-gather of every
    bucket: the two halves of a ring all-reduce as separate collectives, selectable so that the first multi-GPU run can compare them on
    the xGMI mesh).  Environment overrides for A/B runs without code changes: MI355DET_GRADSYNC=all_reduce|rs_ag, MI355DET_BUCKET_MB=<MiB>."""

    def __init__(self, flat_grad, process_group=None, bucket_mb=64, algo="all_reduce"):
        import os
        algo = os.environ.get("MI355DET_GRADSYNC", algo)
        bucket_mb = float(os.environ.get("MI355DET_BUCKET_MB", bucket_mb))
        if algo not in ("all_reduce", "rs_ag"):
            raise ValueError(f"GradSync: unknown algo {algo!r}")
        self.flat = flat_grad
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.handles = []
        backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self.use_avg = backend == "nccl"
        self.ordered = backend == "nccl"          # RCCL runs a group's collectives in issue order on its own stream; gloo's worker threads do not
        self.algo = algo
        self._shards = {}

    def reduce_slice(self, lo, hi, stream=None):
        """stream: the side stream of the plan whose backward is running (bound into the hook by install(): every plan has its own,
        and the collective must be ordered behind THAT plan's weight-gradient kernels)."""
        if self.world == 1:
            return
        if stream is not None:
            # weight gradients are produced on the plan's side stream (which is ordered behind the BN-parameter gradients
            # of the same layers on the main stream): launch the collective from there
            with torch.cuda.stream(stream):
                self._reduce(lo, hi)
        else:
            self._reduce(lo, hi)

    def _reduce(self, lo, hi):
        if self.algo == "rs_ag":
            return self._reduce_rs_ag(lo, hi)
        t = self.flat[lo:hi]
        if self.use_avg:
            self.handles.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:
            self.handles.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), t))

    def _reduce_rs_ag(self, lo, hi):
        """Bucket [lo, hi) as reduce-scatter into a per-rank shard + all-gather back in place; the (< world) elements that do not divide
        evenly ride on a tiny all-reduce."""
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        n = hi - lo
        main = n - n % self.world
        if main:
            t = self.flat[lo:lo + main]
            shard = self._shards.get((lo, hi))
            if shard is None:
                shard = self._shards[(lo, hi)] = torch.empty(main // self.world, dtype=t.dtype, device=t.device)
            h = dist.reduce_scatter_tensor(shard, t, op=op, group=self.group, async_op=True)
            if not self.ordered:
                h.wait()
            else:
                self.handles.append((h, None))      # RCCL keeps issue order; the handle is still waited on in wait()
            self.handles.append((dist.all_gather_into_tensor(t, shard, group=self.group, async_op=True), None if self.use_avg else t))
        if n - main:
            t = self.flat[lo + main:hi]
            self.handles.append((dist.all_reduce(t, op=op, group=self.group, async_op=True), None if self.use_avg else t))

    def wait(self):
        for h, t in self.handles:
            h.wait()
            if t is not None:
                t.div_(self.world)
        self.handles = []

    def attach(self, engine):
        """Register with an engine: every training plan the engine builds from now on (one per input size - multi-scale training,
        train_one_epoch.py:64-69 - and again after an LRU eviction) gets the bucket hooks, and so do the plans it already holds."""
        syncs = engine.__dict__.setdefault("grad_syncs", [])
        if self not in syncs:
            syncs.append(self)
        for plan in engine.plans.values():
            if plan.training:
                self.install(plan)
        return self

    def install(self, plan):
        """Weave bucket all-reduces into a training plan's backward call list (idempotent per plan)."""
        from .yolo.nets.engine import comm_hook
        if getattr(plan, "_gradsync", None) is self:
            return
        buckets = plan_buckets(plan.bwd_marks, self.flat.numel(), self.bucket_elems)
        base = plan.bwd_base if hasattr(plan, "bwd_base") else plan.bwd
        stream = getattr(plan, "side_stream", None)
        calls, prev = [], 0
        for pos, lo, hi in buckets:
            calls += base[prev:pos]
            calls.append((comm_hook, (self.reduce_slice, lo, hi, stream)))
            prev = pos
        calls += base[prev:]
        plan.bwd_base = base
        plan.bwd = calls
        plan._gradsync = self
        self.buckets = buckets


class ParamGradSync:
    """Gradient averaging for ordinary torch parameters that live OUTSIDE an engine's flat buffer: the TwoMLPHead / FastRCNNPredictor of
    `tvision.frcnn.FasterRCNN` (`model.head_parameters()`; torch DDP covers them in the reference, detection/train.py:160).  Their
    gradients are produced by autograd before the engine's backward starts, so ONE flattened all-reduce right after `loss.backward()` runs
    under the whole backbone backward:

        sync = GradSync(model.engine.flat_g).attach(model.engine)        # backbone + RPN, every plan
        hsync = ParamGradSync(model.head_parameters())                   # box head
        losses = model(images, targets); hsync.reduce(); sync.wait(); hsync.wait()
    """

    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.use_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.flat = None
        self.handle = None

    def reduce(self):
        if self.world == 1 or not self.params:
            return
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        self.flat = torch.cat([g.reshape(-1) for g in grads])
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        self.handle = dist.all_reduce(self.flat, op=op, group=self.group, async_op=True)

    def wait(self):
        if self.handle is None:
            return
        self.handle.wait()
        self.handle = None
        if not self.use_avg:
            self.flat.div_(self.world)
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n


def step_stream(device):
    """High-priority stream for the training step's dependency chain.  Make it current (`torch.cuda.set_stream` or
    `with torch.cuda.stream(...)`) BEFORE the engine builds its plan: plans bind their launches to the stream that is current
    at build time, and their side stream (weight gradients, gradient all-reduce) stays at the default priority, so the chain
    is never queued behind side-stream work."""
    return torch.cuda.Stream(device=device, priority=-1)

