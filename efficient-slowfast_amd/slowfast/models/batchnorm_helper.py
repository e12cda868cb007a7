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
    """batchnorm_helper.py:15-34."""
    if cfg.BN.NORM_TYPE == "batchnorm":
        return nn.BatchNorm3d
    elif cfg.BN.NORM_TYPE == "sub_batchnorm":
        return partial(SubBatchNorm3d, num_splits=cfg.BN.NUM_SPLITS)
    elif cfg.BN.NORM_TYPE == "sync_batchnorm":
        return partial(NaiveSyncBatchNorm3d, num_sync_devices=cfg.BN.NUM_SYNC_DEVICES)
    else:
        raise NotImplementedError("Norm type {} is not supported".format(cfg.BN.NORM_TYPE))


class SubBatchNorm3d(nn.Module):
    """BN with statistics over 1/num_splits of the batch each (multigrid training); one shared affine; `bn`
    holds the aggregated eval statistics, `split_bn` the per-split running statistics
    (batchnorm_helper.py:37-109)."""

    def __init__(self, num_splits, **args):
        super(SubBatchNorm3d, self).__init__()
        self.num_splits = num_splits
        num_features = args["num_features"]
        if args.get("affine", True):  # keep only one set of weight and bias
            self.affine = True
            args["affine"] = False
            self.weight = torch.nn.Parameter(torch.ones(num_features))
            self.bias = torch.nn.Parameter(torch.zeros(num_features))
        else:
            self.affine = False
        self.bn = nn.BatchNorm3d(**args)
        args["num_features"] = num_features * num_splits
        self.split_bn = nn.BatchNorm3d(**args)

    def _get_aggregated_mean_std(self, means, stds, n):
        mean = means.view(n, -1).sum(0) / n
        std = stds.view(n, -1).sum(0) / n + ((means.view(n, -1) - mean) ** 2).view(n, -1).sum(0) / n
        return mean.detach(), std.detach()

    def aggregate_stats(self):
        """Fold the per-split running statistics into `bn` (call before eval)."""
        if self.split_bn.track_running_stats:
            self.bn.running_mean.data, self.bn.running_var.data = self._get_aggregated_mean_std(
                self.split_bn.running_mean, self.split_bn.running_var, self.num_splits)

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
