"""Clip-batch data parallelism for the MI355X path (one process per GPU, RCCL over xGMI).

The reference wraps the model in DistributedDataParallel (models/build.py:39-43): bucketed all-reduces
issued from autograd hooks.  The native design is ONE all-reduce per step: every
parameter's .grad is a view into one flat fp32 buffer (135.9 MB for SlowFastDualAttention R50), so after
backward a single ncclAllReduce(sum) over the fully connected xGMI mesh plus one scale by 1/world replaces
DDP's ~6 buckets and needs no gradient copies.  `build_model` still offers the DDP wrap for drop-in use."""
import os

import torch
import torch.distributed as dist


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
