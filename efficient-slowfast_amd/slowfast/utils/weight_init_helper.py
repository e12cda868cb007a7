"""ResNet-style initialisation with the reference's semantics and RNG consumption order
(utils/weight_init_helper.py:10-43 + fvcore c2_msra_fill, fvcore/nn/weight_init.py:21-32): walking
model.modules() in registration order, every nn.Conv3d gets kaiming_normal_(fan_out, relu) and a zero
bias, BatchNorm3d weight is 0 for a block's final BN when zero_init_final_bn else 1, Linear ~ N(0, std)."""
import torch.nn as nn


def c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def init_weights(model, fc_init_std=0.01, zero_init_final_bn=True):
    for m in model.modules():
        if isinstance(m, nn.Conv3d):
            c2_msra_fill(m)
        elif isinstance(m, nn.BatchNorm3d):
            final = bool(getattr(m, "transform_final_bn", False)) and zero_init_final_bn
            if m.weight is not None:
                m.weight.data.fill_(0.0 if final else 1.0)
            if m.bias is not None:
                m.bias.data.zero_()
        if isinstance(m, nn.Linear):
            m.weight.data.normal_(mean=0.0, std=fc_init_std)
            m.bias.data.zero_()
