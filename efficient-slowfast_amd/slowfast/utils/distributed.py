"""Clip-batch data parallelism for the MI355X path (one process per GPU, RCCL over xGMI).

The reference wraps the model in DistributedDataParallel (models/build.py:39-43): bucketed all-reduces
issued from autograd hooks.  The native design is ONE all-reduce per step: every
parameter's .grad is a view into one flat fp32 buffer (135.9 MB for SlowFastDualAttention R50), so after
backward a single ncclAllReduce(sum) over the fully connected xGMI mesh plus one scale by 1/world replaces
DDP's ~6 buckets and needs no gradient copies.  `build_model` still offers the DDP wrap for drop-in use."""
from slowfast._overlay import chain_module as _chain_module

_chain_module(globals())  # the reference's namesake (when importable) supplies every name not defined below
import os

import torch
import torch.distributed as dist


_LOCAL_PROCESS_GROUP = None  # per-node group (reference utils/distributed.py:12, 258-273)


def init_distributed_training(cfg):
    """Create one process group per machine and keep this machine's (utils/distributed.py:258-273)."""
    global _LOCAL_PROCESS_GROUP
    if cfg.NUM_GPUS == 1:
        return
    per = cfg.NUM_GPUS
    for i in range(dist.get_world_size() // per):
        pg = dist.new_group(list(range(i * per, (i + 1) * per)))
        if i == cfg.SHARD_ID:
            _LOCAL_PROCESS_GROUP = pg


def get_local_size():
    """Processes per machine (utils/distributed.py:276-286)."""
    if not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size(group=_LOCAL_PROCESS_GROUP)


def get_local_rank():
    """Rank within the per-machine group (utils/distributed.py:289-299)."""
    if not dist.is_available() or not dist.is_initialized():
        return 0
    assert _LOCAL_PROCESS_GROUP is not None
    return dist.get_rank(group=_LOCAL_PROCESS_GROUP)


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_master_proc(num_gpus=8):
    """True on the first process of every machine (utils/distributed.py:94-101); True when not distributed."""
    return get_rank() % num_gpus == 0


def synchronize():
    """Barrier over the default group; nothing to do for one process (utils/distributed.py:126-139)."""
    if get_world_size() > 1:
        dist.barrier()


def all_reduce(tensors, average=True):
    """Sum (or mean) each tensor of the list over all processes, in place; returns the list
    (utils/distributed.py:37-53).  The timed train step does not use this — gradients go through FlatGradients —
    it serves the reference's logging path (`train_net.py:129-138`)."""
    world = get_world_size()
    if world > 1:
        for t in tensors:
            dist.all_reduce(t, async_op=False)
        if average:
            for t in tensors:
                t.mul_(1.0 / world)
    return tensors


def all_gather(tensors):
    """Concatenate every process's copy of each tensor along dim 0 (utils/distributed.py:15-34)."""
    world = get_world_size()
    out = []
    for t in tensors:
        parts = [torch.ones_like(t) for _ in range(world)]
        if world > 1:
            dist.all_gather(parts, t, async_op=False)
        else:
            parts = [t]
        out.append(torch.cat(parts, dim=0))
    return out


def shard_sizes(global_batch, world):
    """Clips per rank (loader.py:67: TRAIN.BATCH_SIZE / NUM_GPUS, must divide)."""
    if global_batch % world != 0:
        raise ValueError("global batch %d is not divisible by %d ranks" % (global_batch, world))
    return [global_batch // world] * world


class FlatGradients(object):
    """Makes every parameter's .grad a view of one flat buffer and all-reduces it: in ONE collective after the
    backward, or (overlap_with_backward) in a few contiguous chunks that start as soon as the stages they cover have
    finished their backward — the tail of the buffer (res5 + head, then res4) holds 85 % of the bytes and is final a
    fifth of the way into the backward pass."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=ref.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._cuts = {}      # top-level child name -> flat offset of its first parameter
        self._hi = n         # [self._hi, n) is already being reduced this step
        self._pending = []   # async work handles of this step's chunks
        self._tape = None    # the backward pass whose milestones launched this step's chunks
        self._model = None   # weakref of the model overlap_with_backward was registered for
        self._comm = None
        self.chunks_last_step = 0

    def zero(self):
        """Start a step: zero the buffer.  Chunks a previous, unfinished step left in flight (a step that skipped
        all_reduce_mean after an error / NaN) are waited for first, and the chunk state starts over."""
        for w in self._pending:
            w.wait()
        self._pending = []
        self._hi = self.flat.numel()
        self._tape = None
        self.flat.zero_()

    def rebind(self):
        """Optimizers / zero_grad(set_to_none=True) may drop the views: re-attach them."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    # ---- chunked all-reduce overlapped with the backward pass
    def _active(self):
        force = os.environ.get("SF_FORCE_ALLREDUCE") == "1"  # exercise the RCCL path on a 1-GPU box
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or force)

    def overlap_with_backward(self, model, boundaries=("s5", "s4")):
        """Cut the flat buffer in front of the first parameter of each named top-level child of `model` and register
        the engine's backward milestone hook: when the backward has passed child X (its gradients and those of every
        later child are complete in the buffer — needs engine.set_grad_sink(True), gradients accumulate in place), the
        range [offset(X), previous cut) is all-reduced on a dedicated stream while the earlier stages' backward goes
        on.  all_reduce_mean() then reduces what is left and waits for all of it.  Chunks are contiguous ranges of
        the same buffer, so the result equals the single collective's element for element."""
        import weakref
        from slowfast.models import engine
        off, where = 0, {}
        for p in self.params:
            where[id(p)] = off
            off += p.numel()
        children = list(model.named_children())
        self._cuts = {}
        for name in boundaries:
            idx = next((i for i, (n, _) in enumerate(children) if n == name), None)
            if idx is None:
                continue
            tail = [p for _, c in children[idx:] for p in c.parameters() if p.requires_grad]
            if not tail:
                continue
            cut = min(where.get(id(p), -1) for p in tail)
            # the range [cut, end) must hold exactly the parameters of this child and of every later child: with a
            # re-ordered parameter list (e.g. BN / non-BN groups, as the reference's optimizer builder makes them:
            # models/optimizer.py:25-40) a range would be reduced before all of its gradients are complete
            behind = {id(p) for p, o in zip(self.params, self._offsets()) if o >= cut}
            if cut < 0 or behind != {id(p) for p in tail}:
                raise ValueError(
                    "FlatGradients.overlap_with_backward: the parameters behind the first parameter of %r are not "
                    "exactly those of %r and the children after it — build FlatGradients from model.parameters() in "
                    "the model's own order, or reduce in one collective (no overlap)" % (name, name))
            self._cuts[name] = cut
        self._model = weakref.ref(model)
        engine.set_milestone_hook(self._on_milestone, model)
        return self

    def _offsets(self):
        off = 0
        for p in self.params:
            yield off
            off += p.numel()

    def _on_milestone(self, name, tape):
        from slowfast.models import engine
        if self._model is None or getattr(tape, "model", None) is not self._model():
            return  # the backward of another model
        lo = self._cuts.get(name)
        if lo is None or not self._active() or not engine._GRAD_SINK:
            return
        if self._tape is not None and tape is not self._tape and self._hi < self.flat.numel():
            # a SECOND backward since zero() (gradient accumulation) while ranges of the first are already being summed
            # over the ranks: its gradients would land on top of reduced values and the ranks would diverge silently
            raise RuntimeError(
                "FlatGradients: a second backward pass ran before all_reduce_mean() while chunks of the first were "
                "already being all-reduced; for gradient accumulation build it without overlap_with_backward (one "
                "collective after the last backward) or call zero() / all_reduce_mean() between the passes")
        if lo >= self._hi:
            return
        self._tape = tape
        self._launch(lo, self._hi, tape)
        self._hi = lo

    def _launch(self, lo, hi, tape=None):
        chunk = self.flat[lo:hi]
        if chunk.is_cuda:
            dev = chunk.device
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=dev)
            self._comm.wait_stream(torch.cuda.current_stream(dev))
            for wg in (tape.joins if tape is not None else ()):  # weight gradients in flight on companion streams
                self._comm.wait_stream(wg)
            with torch.cuda.stream(self._comm):
                work = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append(work)

    def all_reduce_mean(self):
        """grad <- mean over ranks: ONE collective on the flat buffer, or the remainder of the chunked schedule."""
        if self._active():
            if self._hi > 0:
                self._launch(0, self._hi)
            for w in self._pending:
                w.wait()
            self.chunks_last_step = len(self._pending)
            self._pending = []
            self._hi = self.flat.numel()
            self._tape = None
            self.flat.div_(dist.get_world_size(self.group))
        return self.flat


def max_over_ranks(value, device):
    """Scalar MAX across ranks (bench.py's step time)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
