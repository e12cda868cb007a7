"""SlowFast (reference video_model_builder.py:93-416): two-pathway ResNet with one-way Fast->Slow lateral
connections.  Graph wiring and parameter names follow the reference; the arithmetic is libsfhip."""
import torch.nn as nn

import sfhip
from slowfast.utils import weight_init_helper as init_helper
from . import engine, head_helper, resnet_helper, stem_helper
from .batchnorm_helper import get_norm
from .build import MODEL_REGISTRY

# Number of blocks per stage for each depth (video_model_builder.py:16-17)
_MODEL_STAGE_DEPTH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 18: (2, 2, 2, 2), 34: (3, 4, 6, 3)}

# [stage][pathway] temporal kernel basis; the fork's slowfast entry (custom_video_model_builder.py:155-163,
# identical to video_model_builder.py:62-68)
_TEMPORAL_KERNEL_BASIS = {
    "c2d": [[[1]], [[1]], [[1]], [[1]], [[1]]],
    "c2d_nopool": [[[1]], [[1]], [[1]], [[1]], [[1]]],
    "i3d": [[[5]], [[3]], [[3, 1]], [[3, 1]], [[1, 3]]],
    "i3d_nopool": [[[5]], [[3]], [[3, 1]], [[3, 1]], [[1, 3]]],
    "slow": [[[1]], [[1]], [[1]], [[3]], [[3]]],
    "slowfast": [[[1], [5]], [[1], [3]], [[1], [3]], [[3], [3]], [[3], [3]]],
}
_POOL1 = {
    "c2d": [[2, 1, 1]],
    "c2d_nopool": [[1, 1, 1]],
    "i3d": [[2, 1, 1]],
    "i3d_nopool": [[1, 1, 1]],
    "slow": [[1, 1, 1]],
    "slowfast": [[1, 1, 1], [1, 1, 1]],
}


class FuseFastToSlow(nn.Module):
    """conv_f2s [K,1,1]/s[alpha,1,1] (C_f -> ratio*C_f) + BN + ReLU, concatenated onto the Slow pathway
    (video_model_builder.py:93-150).  One kernel launch, written into the slow tensor's reserved slice."""

    def __init__(self, dim_in, fusion_conv_channel_ratio, fusion_kernel, alpha, eps=1e-5, bn_mmt=0.1,
                 inplace_relu=True, norm_module=nn.BatchNorm3d):
        super(FuseFastToSlow, self).__init__()
        self.conv_f2s = nn.Conv3d(dim_in, dim_in * fusion_conv_channel_ratio, kernel_size=[fusion_kernel, 1, 1],
                                  stride=[alpha, 1, 1], padding=[fusion_kernel // 2, 0, 0], bias=False)
        self.bn = norm_module(num_features=dim_in * fusion_conv_channel_ratio, eps=eps, momentum=bn_mmt)
        self.relu = nn.ReLU(inplace_relu)

    def reserve(self, dims):
        """(before, after) channel room each pathway's producer should leave: slow gets the fuse after it."""
        return [(0, self.conv_f2s.out_channels), (0, 0)]

    def forward(self, x, defer_join=False):  # single direction: nothing to defer
        x_s, x_f = engine.enter(x)
        cf = self.conv_f2s.out_channels
        if x_s.coff == 0 and x_s.cs == x_s.C + cf:
            wide = sfhip.Act(x_s.buf)
        else:  # producer left no room (stand-alone call): allocate the concat buffer and copy the slow half
            if engine.tape() is not None:
                raise NotImplementedError("taped lateral fusion needs the producer-reserved concat slice")
            wide = sfhip.new_act(x_s, x_s.N, x_s.T, x_s.H, x_s.W, x_s.C + cf)
            sfhip.copy_channels(x_s, wide.slice(0, x_s.C))
        engine.conv_bn_act(x_f, self.conv_f2s, self.bn, relu=True, out=wide.slice(x_s.C, cf))
        return engine.leave([wide, x_f])


class _TwoPathwayResNet(nn.Module):
    """Shared wiring of SlowFast and SlowFastDualAttention: s1, s1_fuse, s2, s2_fuse, pathway{0,1}_pool,
    s3, s3_fuse, s4, s4_fuse, s5, head (child order is part of the Grad-CAM contract)."""

    def _make_fuse(self, cfg, dim_in):
        raise NotImplementedError

    def _construct_network(self, cfg, fuse_widths):
        assert cfg.MODEL.ARCH in _POOL1.keys()
        pool_size = _POOL1[cfg.MODEL.ARCH]
        assert len({len(pool_size), self.num_pathways}) == 1
        assert cfg.RESNET.DEPTH in _MODEL_STAGE_DEPTH.keys()
        (d2, d3, d4, d5) = _MODEL_STAGE_DEPTH[cfg.RESNET.DEPTH]
        num_groups = cfg.RESNET.NUM_GROUPS
        width_per_group = cfg.RESNET.WIDTH_PER_GROUP
        dim_inner = num_groups * width_per_group
        beta_inv = cfg.SLOWFAST.BETA_INV
        temp_kernel = _TEMPORAL_KERNEL_BASIS[cfg.MODEL.ARCH]

        self.s1 = stem_helper.VideoModelStem(
            dim_in=cfg.DATA.INPUT_CHANNEL_NUM,
            dim_out=[width_per_group, width_per_group // beta_inv],
            kernel=[temp_kernel[0][0] + [7, 7], temp_kernel[0][1] + [7, 7]],
            stride=[[1, 2, 2]] * 2,
            padding=[[temp_kernel[0][0][0] // 2, 3, 3], [temp_kernel[0][1][0] // 2, 3, 3]],
            norm_module=self.norm_module,
        )
        self.s1_fuse = self._make_fuse(cfg, [width_per_group, width_per_group // beta_inv])

        depths = (d2, d3, d4, d5)
        for i in range(4):  # res2..res5 = s2..s5
            mult_out = 2 ** (i + 2)
            c_s_in = width_per_group if i == 0 else width_per_group * 2 ** (i + 1)
            c_f_in = c_s_in // beta_inv
            extra_s, extra_f = fuse_widths(c_s_in, c_f_in)
            stage = resnet_helper.ResStage(
                dim_in=[c_s_in + extra_s, c_f_in + extra_f],
                dim_out=[width_per_group * mult_out, width_per_group * mult_out // beta_inv],
                dim_inner=[dim_inner * 2 ** i, dim_inner * 2 ** i // beta_inv],
                temp_kernel_sizes=temp_kernel[i + 1],
                stride=cfg.RESNET.SPATIAL_STRIDES[i],
                num_blocks=[depths[i]] * 2,
                num_groups=[num_groups] * 2,
                num_block_temp_kernel=cfg.RESNET.NUM_BLOCK_TEMP_KERNEL[i],
                nonlocal_inds=cfg.NONLOCAL.LOCATION[i],
                nonlocal_group=cfg.NONLOCAL.GROUP[i],
                nonlocal_pool=cfg.NONLOCAL.POOL[i],
                instantiation=cfg.NONLOCAL.INSTANTIATION,
                trans_func_name=cfg.RESNET.TRANS_FUNC,
                dilation=cfg.RESNET.SPATIAL_DILATIONS[i],
                norm_module=self.norm_module,
            )
            setattr(self, "s{}".format(i + 2), stage)
            if i < 3:
                setattr(self, "s{}_fuse".format(i + 2), self._make_fuse(
                    cfg, [width_per_group * mult_out, width_per_group * mult_out // beta_inv]))
            if i == 0:
                for pathway in range(self.num_pathways):
                    pool = nn.MaxPool3d(kernel_size=pool_size[pathway], stride=pool_size[pathway], padding=[0, 0, 0])
                    self.add_module("pathway{}_pool".format(pathway), pool)

        if cfg.DETECTION.ENABLE:
            raise NotImplementedError("DETECTION.ENABLE (ResNetRoIHead / AVA) is out of scope of the HIP path")
        self.head = head_helper.ResNetBasicHead(
            dim_in=[width_per_group * 32, width_per_group * 32 // beta_inv],
            num_classes=cfg.MODEL.NUM_CLASSES,
            pool_size=[None, None] if cfg.MULTIGRID.SHORT_CYCLE else [
                [cfg.DATA.NUM_FRAMES // cfg.SLOWFAST.ALPHA // pool_size[0][0],
                 cfg.DATA.CROP_SIZE // 32 // pool_size[0][1], cfg.DATA.CROP_SIZE // 32 // pool_size[0][2]],
                [cfg.DATA.NUM_FRAMES // pool_size[1][0],
                 cfg.DATA.CROP_SIZE // 32 // pool_size[1][1], cfg.DATA.CROP_SIZE // 32 // pool_size[1][2]],
            ],
            dropout_rate=cfg.MODEL.DROPOUT_RATE,
            act_func=cfg.MODEL.HEAD_ACT,
        )

    def forward(self, x, bboxes=None):
        return engine.run_model(self, x)

    def _forward_impl(self, x):
        x = list(x)
        with engine.internal():
            x = self.s1(x, reserve=self.s1_fuse.reserve(None))
            x = self._fuse(self.s1_fuse, x)
            x = self.s2(x, reserve=self.s2_fuse.reserve(None))
            x = self._fuse(self.s2_fuse, x)
            for pathway in range(self.num_pathways):
                pool = getattr(self, "pathway{}_pool".format(pathway))
                ks = pool.kernel_size if isinstance(pool.kernel_size, (list, tuple)) else [pool.kernel_size] * 3
                if list(ks) != [1, 1, 1]:  # _POOL1["slowfast"] is the identity (elided)
                    x[pathway] = engine.maxpool(x[pathway], tuple(ks), tuple(ks))
            engine.milestone("s3")  # backward: the gradients of s3 .. head are complete here (chunked all-reduce)
            x = self.s3(x, reserve=self.s3_fuse.reserve(None))
            x = self._fuse(self.s3_fuse, x)
            engine.milestone("s4")
            x = self.s4(x, reserve=self.s4_fuse.reserve(None))
            x = self._fuse(self.s4_fuse, x)
            engine.milestone("s5")
            x = self.s5(x)
            x = self.head(x)
        return x

    def _fuse(self, fuse, x):
        """A lateral fusion that is followed by a ResStage: CMDA may leave its attention running on the side stream
        (the stage's own two-stream region orders everything again); the identity pools in between touch nothing."""
        pools_are_identity = all(
            list(getattr(self, "pathway{}_pool".format(p)).kernel_size) == [1, 1, 1] for p in range(self.num_pathways))
        return fuse(x, defer_join=pools_are_identity and engine.DEFER_JOIN)


@MODEL_REGISTRY.register()
class SlowFast(_TwoPathwayResNet):
    """SlowFast networks for video recognition (Feichtenhofer et al.), reference
    video_model_builder.py:154-416: FuseFastToSlow laterals, slow widths C + ratio*C/beta_inv."""

    def __init__(self, cfg):
        super(SlowFast, self).__init__()
        self.norm_module = get_norm(cfg)
        self.enable_detection = cfg.DETECTION.ENABLE
        self.num_pathways = 2
        # slow input width = C + C // out_dim_ratio, out_dim_ratio = beta_inv // ratio (:198-200, :224-233)
        out_dim_ratio = cfg.SLOWFAST.BETA_INV // cfg.SLOWFAST.FUSION_CONV_CHANNEL_RATIO
        self._construct_network(cfg, lambda c_s, c_f: (c_s // out_dim_ratio, 0))
        init_helper.init_weights(self, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN)

    def _make_fuse(self, cfg, dim_in):
        return FuseFastToSlow(dim_in[1], cfg.SLOWFAST.FUSION_CONV_CHANNEL_RATIO, cfg.SLOWFAST.FUSION_KERNEL_SZ,
                              cfg.SLOWFAST.ALPHA, norm_module=self.norm_module)


@MODEL_REGISTRY.register()
class ResNet(nn.Module):
    """Single-pathway ResNet backbone without lateral connections — C2D, I3D, Slow (reference
    video_model_builder.py:420-616); children s1, s2, pathway0_pool, s3, s4, s5, head."""

    def __init__(self, cfg):
        super(ResNet, self).__init__()
        self.norm_module = get_norm(cfg)
        self.enable_detection = cfg.DETECTION.ENABLE
        self.num_pathways = 1
        self._construct_network(cfg)
        init_helper.init_weights(self, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN)

    def _construct_network(self, cfg):
        assert cfg.MODEL.ARCH in _POOL1.keys()
        pool_size = _POOL1[cfg.MODEL.ARCH]
        assert len({len(pool_size), self.num_pathways}) == 1
        assert cfg.RESNET.DEPTH in _MODEL_STAGE_DEPTH.keys()
        depths = _MODEL_STAGE_DEPTH[cfg.RESNET.DEPTH]
        num_groups = cfg.RESNET.NUM_GROUPS
        width_per_group = cfg.RESNET.WIDTH_PER_GROUP
        dim_inner = num_groups * width_per_group
        temp_kernel = _TEMPORAL_KERNEL_BASIS[cfg.MODEL.ARCH]
        self.s1 = stem_helper.VideoModelStem(
            dim_in=cfg.DATA.INPUT_CHANNEL_NUM, dim_out=[width_per_group], kernel=[temp_kernel[0][0] + [7, 7]],
            stride=[[1, 2, 2]], padding=[[temp_kernel[0][0][0] // 2, 3, 3]], norm_module=self.norm_module)
        for i in range(4):  # res2..res5 = s2..s5
            stage = resnet_helper.ResStage(
                dim_in=[width_per_group * (1 if i == 0 else 2 ** (i + 1))], dim_out=[width_per_group * 2 ** (i + 2)],
                dim_inner=[dim_inner * 2 ** i], temp_kernel_sizes=temp_kernel[i + 1],
                stride=cfg.RESNET.SPATIAL_STRIDES[i], num_blocks=[depths[i]], num_groups=[num_groups],
                num_block_temp_kernel=cfg.RESNET.NUM_BLOCK_TEMP_KERNEL[i], nonlocal_inds=cfg.NONLOCAL.LOCATION[i],
                nonlocal_group=cfg.NONLOCAL.GROUP[i], nonlocal_pool=cfg.NONLOCAL.POOL[i],
                instantiation=cfg.NONLOCAL.INSTANTIATION, trans_func_name=cfg.RESNET.TRANS_FUNC,
                stride_1x1=cfg.RESNET.STRIDE_1X1, inplace_relu=cfg.RESNET.INPLACE_RELU,
                dilation=cfg.RESNET.SPATIAL_DILATIONS[i], norm_module=self.norm_module)
            setattr(self, "s{}".format(i + 2), stage)
            if i == 0:
                for pathway in range(self.num_pathways):
                    pool = nn.MaxPool3d(kernel_size=pool_size[pathway], stride=pool_size[pathway], padding=[0, 0, 0])
                    self.add_module("pathway{}_pool".format(pathway), pool)
        if cfg.DETECTION.ENABLE:
            raise NotImplementedError("DETECTION.ENABLE (ResNetRoIHead / AVA) is out of scope of the HIP path")
        self.head = head_helper.ResNetBasicHead(
            dim_in=[width_per_group * 32], num_classes=cfg.MODEL.NUM_CLASSES,
            pool_size=[None, None] if cfg.MULTIGRID.SHORT_CYCLE else [
                [cfg.DATA.NUM_FRAMES // pool_size[0][0], cfg.DATA.CROP_SIZE // 32 // pool_size[0][1],
                 cfg.DATA.CROP_SIZE // 32 // pool_size[0][2]]],
            dropout_rate=cfg.MODEL.DROPOUT_RATE, act_func=cfg.MODEL.HEAD_ACT)

    def forward(self, x, bboxes=None):
        return engine.run_model(self, x)

    def _forward_impl(self, x):
        x = list(x)
        with engine.internal():
            x = self.s1(x)
            x = self.s2(x)
            ks = self.pathway0_pool.kernel_size
            ks = tuple(ks) if isinstance(ks, (list, tuple)) else (ks,) * 3
            if ks != (1, 1, 1):
                x[0] = engine.maxpool(x[0], ks, ks)
            engine.milestone("s3")
            x = self.s3(x)
            engine.milestone("s4")
            x = self.s4(x)
            engine.milestone("s5")
            x = self.s5(x)
            x = self.head(x)
        return x
