"""Normalisation selection (reference batchnorm_helper.py:15-34).  Only plain BatchNorm3d is on the HIP
path; Sub/Sync-BN (multigrid long cycle) are SURVEY §8(f) rank 1."""
import torch.nn as nn


def get_norm(cfg):
    if cfg.BN.NORM_TYPE == "batchnorm":
        return nn.BatchNorm3d
    if cfg.BN.NORM_TYPE in ("sub_batchnorm", "sync_batchnorm"):
        raise NotImplementedError(
            "BN.NORM_TYPE={} is not implemented on the MI355X path yet (only 'batchnorm')".format(cfg.BN.NORM_TYPE))
    raise NotImplementedError("Norm type {} is not supported".format(cfg.BN.NORM_TYPE))
