"""BatchNorm selection and the custom batch-size BN layers (reference batchnorm_helper.py).

The modules below only hold parameters/buffers under the reference's names; the arithmetic is the HIP path in
`engine.bn_train_apply` / `engine.bn_affine`:
  * SubBatchNorm3d: sample n belongs to split n % num_splits (that is what view(n//S, c*S, t, h, w) means for a
    contiguous NCTHW batch); the statistics kernel reduces per (split, channel) in one pass, the normalise and
    backward kernels index their per-channel vectors by split.
  * NaiveSyncBatchNorm3d: local (mean, E[x^2]) are all-gathered over the per-node process group and averaged
    over the sync group, exactly GroupGather's forward; its backward is the same gather-and-sum applied to
    (sum g, sum g*xhat) between the two BN-backward launches."""
from functools import partial

import torch
import torch.nn as nn

import slowfast.utils.distributed as du


def get_norm(cfg):
    """cfg.BN.NORM_TYPE -> the constructor every model passes its `num_features=..., eps=..., momentum=...` to
    (reference batchnorm_helper.py:15-34; same three names, same NotImplementedError text for anything else)."""
    kind = cfg.BN.NORM_TYPE
    makers = {
        "batchnorm": lambda: nn.BatchNorm3d,
        "sub_batchnorm": lambda: partial(SubBatchNorm3d, num_splits=cfg.BN.NUM_SPLITS),
        "sync_batchnorm": lambda: partial(NaiveSyncBatchNorm3d, num_sync_devices=cfg.BN.NUM_SYNC_DEVICES),
    }
    if kind not in makers:
        raise NotImplementedError("Norm type {} is not supported".format(kind))
    return makers[kind]()


class SubBatchNorm3d(nn.Module):
    """BN whose batch statistics cover 1/num_splits of the batch each (multigrid training).  Checkpoint layout
    (reference batchnorm_helper.py:37-109): one shared `weight` / `bias`; `split_bn` (affine-free, num_splits * C
    features) carries the per-split running statistics that training updates; `bn` (affine-free, C features) the
    statistics evaluation uses, filled from `split_bn` by aggregate_stats()."""

    def __init__(self, num_splits, **args):
        super(SubBatchNorm3d, self).__init__()
        self.num_splits = num_splits
        feats = args["num_features"]
        self.affine = bool(args.get("affine", True))
        if self.affine:  # the two inner layers stay affine-free: the (one) affine pair lives on this module
            self.weight = torch.nn.Parameter(torch.ones(feats))
            self.bias = torch.nn.Parameter(torch.zeros(feats))
        inner = dict(args, affine=False)
        self.bn = nn.BatchNorm3d(**inner)
        self.split_bn = nn.BatchNorm3d(**dict(inner, num_features=feats * num_splits))

    def _get_aggregated_mean_std(self, means, stds, n):
        """Statistics of the union of n equally sized splits from the per-split ones (law of total variance):
        mean = average of the split means; variance = average split variance + variance of the split means.
        `stds` holds variances (the reference's naming, batchnorm_helper.py:75-88)."""
        per_split_mean = means.detach().reshape(n, -1)
        per_split_var = stds.detach().reshape(n, -1)
        mean = per_split_mean.sum(dim=0) / n
        between = (per_split_mean - mean).pow(2).sum(dim=0) / n
        return mean, per_split_var.sum(dim=0) / n + between

    def aggregate_stats(self):
        """Fill `bn`'s running statistics from the per-split ones (the training loop calls this before evaluating,
        utils/misc.py:257-273)."""
        if not self.split_bn.track_running_stats:
            return
        mean, var = self._get_aggregated_mean_std(self.split_bn.running_mean, self.split_bn.running_var,
                                                  self.num_splits)
        self.bn.running_mean.data = mean
        self.bn.running_var.data = var

    def forward(self, x):
        from slowfast.models import engine
        return engine.norm_forward(self, x)


class NaiveSyncBatchNorm3d(nn.BatchNorm3d):
    """BatchNorm3d whose training statistics are averaged over `num_sync_devices` ranks of the local process
    group (batchnorm_helper.py:174-218).  Running variance is the BIASED batch variance, as in the reference."""

    def __init__(self, num_sync_devices, **args):
        self.num_sync_devices = num_sync_devices
        if self.num_sync_devices > 0:
            assert du.get_local_size() % self.num_sync_devices == 0, (du.get_local_size(), self.num_sync_devices)
            self.num_groups = du.get_local_size() // self.num_sync_devices
        else:
            self.num_sync_devices = du.get_local_size()
            self.num_groups = 1
        super(NaiveSyncBatchNorm3d, self).__init__(**args)
        # its statistics are collectives: engine.run_model keeps a model that contains such a layer (and runs on more
        # than one local rank) on ONE stream, so that every rank issues its collectives in program order — decided per
        # model, not process-wide
        self._sf_collective = True

    def forward(self, input):
        from slowfast.models import engine
        return engine.norm_forward(self, input)


def group_gather_sum(vec, num_sync_devices, num_groups):
    """GroupGather.forward (batchnorm_helper.py:112-141): all-gather over the local process group, keep this
    rank's sync group, sum.  Used for the statistics and — its autograd backward being the same operation — for
    the gradient sums."""
    import torch.distributed as dist
    group = du._LOCAL_PROCESS_GROUP
    dev = vec.device
    if vec.is_cuda and dist.get_backend(group) == "gloo":
        vec = vec.cpu()  # gloo has no device all_gather; RCCL ("nccl") gathers in place on the GPU
    parts = [torch.zeros_like(vec) for _ in range(du.get_local_size())]
    dist.all_gather(parts, vec, async_op=False, group=group)
    stacked = torch.stack(parts, dim=0).to(dev)
    if num_groups > 1:
        g = du.get_local_rank() // num_sync_devices
        stacked = stacked[g * num_sync_devices:(g + 1) * num_sync_devices]
    return stacked.sum(0)
