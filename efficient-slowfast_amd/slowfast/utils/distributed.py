"""Clip-batch data parallelism for the MI355X path (one process per GPU, RCCL over xGMI).

The reference wraps the model in DistributedDataParallel (models/build.py:39-43): bucketed all-reduces
issued from autograd hooks.  The native design is ONE all-reduce per step: every
parameter's .grad is a view into one flat fp32 buffer (135.9 MB for SlowFastDualAttention R50), so after
backward a single ncclAllReduce(sum) over the fully connected xGMI mesh plus one scale by 1/world replaces
DDP's ~6 buckets and needs no gradient copies.  `build_model` still offers the DDP wrap for drop-in use."""
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


def shard_sizes(global_batch, world):
    """Clips per rank (loader.py:67: TRAIN.BATCH_SIZE / NUM_GPUS, must divide)."""
    if global_batch % world != 0:
        raise ValueError("global batch %d is not divisible by %d ranks" % (global_batch, world))
    return [global_batch // world] * world


class FlatGradients(object):
    """Makes every parameter's .grad a view of one flat buffer and all-reduces it in a single collective."""

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

    def zero(self):
        self.flat.zero_()

    def rebind(self):
        """Optimizers / zero_grad(set_to_none=True) may drop the views: re-attach them."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def all_reduce_mean(self):
        """grad <- mean over ranks: ONE collective on the flat buffer."""
        force = os.environ.get("SF_FORCE_ALLREDUCE") == "1"  # exercise the RCCL path on a 1-GPU box
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or force):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(dist.get_world_size(self.group))
        return self.flat


def max_over_ranks(value, device):
    """Scalar MAX across ranks (bench.py's step time)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
