"""Precise-BN pass: replace the exponential running statistics by the plain average of per-batch statistics
over `num_iters` training-mode forwards with frozen weights.  The reference calls fvcore's
`update_bn_stats(model, loader, num_iters)` from `calculate_and_update_precise_bn` (tools/train_net.py:277-296,
vendored fvcore/nn/precise_bn.py); fvcore is not part of this image, so the two entry points live here under the
same names.  The forwards run the HIP path: momentum 1.0 makes `sf_bn_train_stats` write the batch statistics
straight into running_mean / running_var, which are then averaged on the device."""
import itertools

import torch
from torch import nn

BN_MODULE_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)


def get_bn_modules(model):
    """All BN layers that are in training mode (SubBatchNorm3d contributes its inner `bn` and `split_bn`)."""
    return [m for m in model.modules() if m.training and isinstance(m, BN_MODULE_TYPES)]


@torch.no_grad()
def update_bn_stats(model, data_loader, num_iters=200):
    """model: in training mode for the layers that need precise statistics; data_loader: iterable of model inputs."""
    layers = get_bn_modules(model)
    if not layers:
        return
    saved = [bn.momentum for bn in layers]
    for bn in layers:
        bn.momentum = 1.0
    mean = [torch.zeros_like(bn.running_mean) for bn in layers]
    var = [torch.zeros_like(bn.running_var) for bn in layers]
    ind = -1
    for ind, inputs in enumerate(itertools.islice(data_loader, num_iters)):
        model(inputs)
        for i, bn in enumerate(layers):
            mean[i] += (bn.running_mean - mean[i]) / (ind + 1)
            var[i] += (bn.running_var - var[i]) / (ind + 1)
    if ind != num_iters - 1:
        raise AssertionError("precise-BN pass: the loader ran dry after %d of the %d batches asked for" % (ind + 1, num_iters))
    for i, bn in enumerate(layers):
        bn.running_mean = mean[i]
        bn.running_var = var[i]
        bn.momentum = saved[i]
