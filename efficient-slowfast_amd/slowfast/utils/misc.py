"""The BN-related helpers of the reference's utils/misc.py that sit on the training path (:241-274)."""
from slowfast._overlay import chain_module as _chain_module

_chain_module(globals())  # the reference's namesake (when importable) supplies every name not defined below
import torch.nn as nn

from slowfast.models.batchnorm_helper import SubBatchNorm3d


def frozen_bn_stats(model):
    """Set all BN layers to eval mode (utils/misc.py:246-254)."""
    for m in model.modules():
        if isinstance(m, nn.BatchNorm3d):
            m.eval()


def aggregate_sub_bn_stats(module):
    """Recursively aggregate every SubBatchNorm3d's split statistics; returns how many were found
    (utils/misc.py:257-273)."""
    count = 0
    for child in module.children():
        if isinstance(child, SubBatchNorm3d):
            child.aggregate_stats()
            count += 1
        else:
            count += aggregate_sub_bn_stats(child)
    return count
