"""ResNet stages (reference resnet_helper.py).  Each bottleneck is three fused kernel launches
(conv+BN+ReLU, conv+BN+ReLU, conv+BN+residual+ReLU); a projection shortcut adds one more."""
import torch.nn as nn

import sfhip
from . import engine
from .nonlocal_helper import Nonlocal


def get_trans_func(name):
    trans_funcs = {"bottleneck_transform": BottleneckTransform, "basic_transform": BasicTransform}
    assert name in trans_funcs.keys(), "Transformation function '{}' not supported".format(name)
    return trans_funcs[name]


class BasicTransform(nn.Module):
    """[kT,3,3] conv+BN+ReLU -> [1,3,3] conv+BN (resnet_helper.py:25-107; R18/R34 YAMLs)."""

    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, dim_inner=None, num_groups=1, stride_1x1=None,
                 inplace_relu=True, eps=1e-5, bn_mmt=0.1, dilation=1, norm_module=nn.BatchNorm3d):
        super(BasicTransform, self).__init__()
        self.temp_kernel_size = temp_kernel_size
        self.a = nn.Conv3d(dim_in, dim_out, kernel_size=[temp_kernel_size, 3, 3], stride=[1, stride, stride],
                           padding=[int(temp_kernel_size // 2), 1, 1], bias=False)
        self.a_bn = norm_module(num_features=dim_out, eps=eps, momentum=bn_mmt)
        self.a_relu = nn.ReLU(inplace=inplace_relu)
        self.b = nn.Conv3d(dim_out, dim_out, kernel_size=[1, 3, 3], stride=[1, 1, 1], padding=[0, 1, 1], bias=False)
        self.b_bn = norm_module(num_features=dim_out, eps=eps, momentum=bn_mmt)
        self.b_bn.transform_final_bn = True

    def forward(self, x, res=None, relu=False, out=None, reserve=(0, 0)):
        x = engine.conv_bn_act(x, self.a, self.a_bn, relu=True)
        return engine.conv_bn_act(x, self.b, self.b_bn, relu=relu, res=res, out=out, out_reserve=reserve)


class BottleneckTransform(nn.Module):
    """[kT,1,1] -> [1,3,3] (stride, groups, dilation) -> [1,1,1], BN after each (resnet_helper.py:110-240)."""

    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, dim_inner, num_groups, stride_1x1=False,
                 inplace_relu=True, eps=1e-5, bn_mmt=0.1, dilation=1, norm_module=nn.BatchNorm3d):
        super(BottleneckTransform, self).__init__()
        self.temp_kernel_size = temp_kernel_size
        self._stride_1x1 = stride_1x1
        (str1x1, str3x3) = (stride, 1) if stride_1x1 else (1, stride)
        self.a = nn.Conv3d(dim_in, dim_inner, kernel_size=[temp_kernel_size, 1, 1], stride=[1, str1x1, str1x1],
                           padding=[int(temp_kernel_size // 2), 0, 0], bias=False)
        self.a_bn = norm_module(num_features=dim_inner, eps=eps, momentum=bn_mmt)
        self.a_relu = nn.ReLU(inplace=inplace_relu)
        self.b = nn.Conv3d(dim_inner, dim_inner, [1, 3, 3], stride=[1, str3x3, str3x3],
                           padding=[0, dilation, dilation], groups=num_groups, bias=False,
                           dilation=[1, dilation, dilation])
        self.b_bn = norm_module(num_features=dim_inner, eps=eps, momentum=bn_mmt)
        self.b_relu = nn.ReLU(inplace=inplace_relu)
        self.c = nn.Conv3d(dim_inner, dim_out, kernel_size=[1, 1, 1], stride=[1, 1, 1], padding=[0, 0, 0],
                           bias=False)
        self.c_bn = norm_module(num_features=dim_out, eps=eps, momentum=bn_mmt)
        self.c_bn.transform_final_bn = True

    def forward(self, x, res=None, relu=False, out=None, reserve=(0, 0)):
        x = engine.conv_bn_act(x, self.a, self.a_bn, relu=True)
        x = engine.conv_bn_act(x, self.b, self.b_bn, relu=True)
        return engine.conv_bn_act(x, self.c, self.c_bn, relu=relu, res=res, out=out, out_reserve=reserve)


class ResBlock(nn.Module):
    """relu(shortcut(x) + branch2(x)); shortcut = 1x1x1 conv s[1,stride,stride] + BN iff dims/stride change
    (resnet_helper.py:243-358).  The add and the ReLU run in branch2's last conv epilogue."""

    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, trans_func, dim_inner, num_groups=1,
                 stride_1x1=False, inplace_relu=True, eps=1e-5, bn_mmt=0.1, dilation=1, norm_module=nn.BatchNorm3d):
        super(ResBlock, self).__init__()
        self._inplace_relu, self._eps, self._bn_mmt = inplace_relu, eps, bn_mmt
        if (dim_in != dim_out) or (stride != 1):
            self.branch1 = nn.Conv3d(dim_in, dim_out, kernel_size=1, stride=[1, stride, stride], padding=0,
                                     bias=False, dilation=1)
            self.branch1_bn = norm_module(num_features=dim_out, eps=self._eps, momentum=self._bn_mmt)
        self.branch2 = trans_func(dim_in, dim_out, temp_kernel_size, stride, dim_inner, num_groups,
                                  stride_1x1=stride_1x1, inplace_relu=inplace_relu, dilation=dilation,
                                  norm_module=norm_module)
        self.relu = nn.ReLU(self._inplace_relu)

    def forward(self, x, reserve=(0, 0)):
        plain = not isinstance(x, engine.Act)
        if plain:
            x = engine.enter([x])[0]
        sc = engine.conv_bn_act(x, self.branch1, self.branch1_bn) if hasattr(self, "branch1") else x
        y = self.branch2(x, res=sc, relu=True, reserve=reserve)
        return engine.leave([y])[0] if plain else y


class ResStage(nn.Module):
    """p pathways x num_blocks ResBlocks, children pathway{p}_res{i} (resnet_helper.py:361-561)."""

    def __init__(self, dim_in, dim_out, stride, temp_kernel_sizes, num_blocks, dim_inner, num_groups,
                 num_block_temp_kernel, nonlocal_inds, nonlocal_group, nonlocal_pool, dilation,
                 instantiation="softmax", trans_func_name="bottleneck_transform", stride_1x1=False,
                 inplace_relu=True, norm_module=nn.BatchNorm3d):
        super(ResStage, self).__init__()
        assert all((num_block_temp_kernel[i] <= num_blocks[i] for i in range(len(temp_kernel_sizes))))
        self.num_blocks = num_blocks
        self.nonlocal_group = nonlocal_group
        self.temp_kernel_sizes = [
            (temp_kernel_sizes[i] * num_blocks[i])[: num_block_temp_kernel[i]]
            + [1] * (num_blocks[i] - num_block_temp_kernel[i])
            for i in range(len(temp_kernel_sizes))
        ]
        assert len({len(dim_in), len(dim_out), len(temp_kernel_sizes), len(stride), len(num_blocks),
                    len(dim_inner), len(num_groups), len(num_block_temp_kernel), len(nonlocal_inds),
                    len(nonlocal_group)}) == 1
        self.num_pathways = len(self.num_blocks)
        for pathway in range(self.num_pathways):
            for i in range(self.num_blocks[pathway]):
                res_block = ResBlock(
                    dim_in[pathway] if i == 0 else dim_out[pathway], dim_out[pathway],
                    self.temp_kernel_sizes[pathway][i], stride[pathway] if i == 0 else 1,
                    get_trans_func(trans_func_name), dim_inner[pathway], num_groups[pathway],
                    stride_1x1=stride_1x1, inplace_relu=inplace_relu, dilation=dilation[pathway],
                    norm_module=norm_module)
                self.add_module("pathway{}_res{}".format(pathway, i), res_block)
                if i in nonlocal_inds[pathway]:
                    nln = Nonlocal(dim_out[pathway], dim_out[pathway] // 2, nonlocal_pool[pathway],
                                   instantiation=instantiation, norm_module=norm_module)
                    self.add_module("pathway{}_nonlocal{}".format(pathway, i), nln)

    def forward(self, inputs, reserve=None):
        xs = engine.enter(inputs)

        def pathway_fn(pathway):
            def run():
                x = xs[pathway]
                n = self.num_blocks[pathway]
                for i in range(n):
                    m = getattr(self, "pathway{}_res{}".format(pathway, i))
                    nln = getattr(self, "pathway{}_nonlocal{}".format(pathway, i), None)
                    room = reserve[pathway] if (reserve and i == n - 1) else (0, 0)
                    x = m(x, room if nln is None else (0, 0))
                    if nln is not None:
                        grp = self.nonlocal_group[pathway]
                        if grp > 1:  # fold groups of T/grp frames into the batch: a free view in NDHWC
                            assert x.coff == 0 and x.T % grp == 0
                            x = sfhip.Act(x.buf.view(x.N * grp, x.T // grp, x.H, x.W, x.cs), 0, x.C)
                        x = nln.run(x, room)
                        if grp > 1:
                            x = sfhip.Act(x.buf.view(x.N // grp, x.T * grp, x.H, x.W, x.cs), x.coff, x.C)
                return x
            return run

        with engine.internal():  # the two pathways are independent inside a stage: Fast on the side stream
            output = engine.run_paths([pathway_fn(p) for p in range(self.num_pathways)], xs[0].buf.device)
        return engine.leave(output)
