"""The fork's models (reference custom_video_model_builder.py): CMDA lateral fusion (FuseFastAndSlow),
SlowFastDualAttention, SlowFastShuffleNetV2, SlowFastGhostNet — same names, cfg keys, state_dict layout
and child order; arithmetic on libsfhip."""
import torch.nn as nn

import sfhip
from slowfast.utils import weight_init_helper as init_helper
from . import engine, head_helper, stem_helper
from .batchnorm_helper import get_norm
from .build import MODEL_REGISTRY
from .ghostnet_helper import GhostNet_Stage, _make_divisible
from .mobilenetv2_helper import MobileNetV2_Stage
from .shufflenet_helper import ShuffleNet_Stage
from .shufflenetv2_helper import ShuffleNetV2_Stage
from .video_model_builder import _TwoPathwayResNet
from .wdf_attention_helper import ECA, SpatialAttention


class FuseFastAndSlow(nn.Module):
    """CMDA — cross-modality dual attention, bidirectional lateral fusion
    (custom_video_model_builder.py:42-148):
      Fast->Slow: MaxPool3d((alpha,1,1)) -> ECA -> bn_f2s -> ReLU -> cat([x_s, .])
      Slow->Fast: Conv3d 1x1x1 (C_s -> C_s/beta_inv) -> SpatialAttention -> bn_s2f -> ReLU ->
                  nearest upsample x alpha in T -> cat([., x_f])   (slow-derived channels FIRST)
    On the HIP path that is 2 bandwidth passes for the F->S edge and 2 GEMMs + the flash kernel for S->F;
    both results land in channel slices their producers reserved, so no concat is executed."""

    def __init__(self, dim_in, alpha, beta_inv, eps=1e-5, bn_mmt=0.1, inplace_relu=True,
                 norm_module=nn.BatchNorm3d, reduction=1):
        super(FuseFastAndSlow, self).__init__()
        self.alpha = alpha
        self.downsample_t_of_fast = nn.MaxPool3d(kernel_size=(alpha, 1, 1), stride=(alpha, 1, 1))
        self.attention_channel_f2s = ECA(dim_in[1])
        print('fusion layer dim input: ', dim_in)  # the reference prints this at construction (:92)
        self.bn_f2s = norm_module(num_features=dim_in[1], eps=eps, momentum=bn_mmt)
        self.relu_f2s = nn.ReLU(inplace_relu)
        self.downsample_c_of_slow = nn.Conv3d(dim_in[0], dim_in[0] // beta_inv, kernel_size=[1, 1, 1],
                                              stride=[1, 1, 1], bias=False)
        self.attention_spatial_s2f = SpatialAttention(int(dim_in[0] // beta_inv), reduction=reduction)
        self.bn_s2f = norm_module(num_features=int(dim_in[0] // beta_inv), eps=eps, momentum=bn_mmt)
        self.relu_s2f = nn.ReLU(inplace_relu)
        self.upsample_s2f = nn.Upsample(scale_factor=(alpha, 1, 1), mode='nearest')
        self._c_f2s = dim_in[1]
        self._c_s2f = int(dim_in[0] // beta_inv)

    def reserve(self, dims):
        """(before, after) channel room per pathway: slow = [x_s | from_fast], fast = [from_slow | x_f]."""
        return [(0, self._c_f2s), (self._c_s2f, 0)]

    def forward(self, x, defer_join=False):
        x_s, x_f = engine.enter(x)
        a = self.alpha
        # ---- destination buffers (in place when the producers reserved room)
        if x_s.coff == 0 and x_s.cs == x_s.C + self._c_f2s:
            s_wide = sfhip.Act(x_s.buf)
        else:
            if engine.tape() is not None:
                raise NotImplementedError("taped CMDA needs the producer-reserved concat slices")
            s_wide = sfhip.new_act(x_s, x_s.N, x_s.T, x_s.H, x_s.W, x_s.C + self._c_f2s)
            sfhip.copy_channels(x_s, s_wide.slice(0, x_s.C))
        if x_f.coff == self._c_s2f and x_f.cs == x_f.C + self._c_s2f:
            f_wide = sfhip.Act(x_f.buf)
        else:
            f_wide = sfhip.new_act(x_f, x_f.N, x_f.T, x_f.H, x_f.W, x_f.C + self._c_s2f)
            sfhip.copy_channels(x_f, f_wide.slice(self._c_s2f, x_f.C))
        def fast_to_slow():
            if self.bn_f2s.training:  # batch statistics sit between the gate and the ReLU: one extra pass
                z = self.attention_channel_f2s.run(x_f, alpha=a)
                engine.bn_train_apply(self.bn_f2s, z, relu=True, out=s_wide.slice(x_s.C, self._c_f2s))
            else:
                sc, bi = engine.bn_affine(self.bn_f2s)
                self.attention_channel_f2s.run(x_f, alpha=a, scale=sc, bias=bi, relu=True,
                                               out=s_wide.slice(x_s.C, self._c_f2s))

        def slow_to_fast():
            y = engine.conv_bn_act(x_s, self.downsample_c_of_slow)
            if self.bn_s2f.training:
                z = self.attention_spatial_s2f.run(y)
                engine.bn_train_apply(self.bn_s2f, z, relu=True, rep=a, out=f_wide.slice(0, self._c_s2f))
            else:
                sc, bi = engine.bn_affine(self.bn_s2f)
                self.attention_spatial_s2f.run(y, scale=sc, bias=bi, relu=True, alpha=a,
                                               out=f_wide.slice(0, self._c_s2f))

        # the two directions touch disjoint channel slices: the MFMA-bound attention runs beside the HBM-bound
        # pool / gate / BN chain of the other direction
        # (Fast->Slow on the caller's stream, the attention on the side stream: with defer_join the Slow pathway's
        # next stage starts as soon as its own input is complete)
        # a forward hook on this module (feature extraction, Grad-CAM, the parity tests) reads the outputs on the
        # caller's stream as soon as forward returns: with hooks attached the join is not deferred
        import torch.nn.modules.module as _tm
        hooked = bool(self._forward_hooks) or bool(_tm._global_forward_hooks)
        engine.run_paths([fast_to_slow, slow_to_fast], x_s.buf.device,
                         defer_join=defer_join and engine.is_internal() and not hooked, fuse=True)
        return engine.leave([s_wide, f_wide])


@MODEL_REGISTRY.register()
class SlowFastDualAttention(_TwoPathwayResNet):
    """Efficient Dual Attention SlowFast Networks for Video Action Recognition (Wei et al.), reference
    custom_video_model_builder.py:172-445: CMDA laterals; both pathways widen by C_s/beta_inv."""

    def __init__(self, cfg):
        super(SlowFastDualAttention, self).__init__()
        self.norm_module = get_norm(cfg)
        self.enable_detection = cfg.DETECTION.ENABLE
        self.num_pathways = 2
        out_dim_ratio = cfg.SLOWFAST.BETA_INV  # "WDF-FIX" (:215)
        self._construct_network(cfg, lambda c_s, c_f: (c_s // out_dim_ratio, c_s // out_dim_ratio))
        init_helper.init_weights(self, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN)

    def _make_fuse(self, cfg, dim_in):
        return FuseFastAndSlow(dim_in=dim_in, alpha=cfg.SLOWFAST.ALPHA, beta_inv=cfg.SLOWFAST.BETA_INV,
                               norm_module=self.norm_module, reduction=1)


class _EfficientTwoPathway(nn.Module):
    """forward = registered children in order: [stems], (stage, fuse)*, [last stage], head."""

    def forward(self, x, bboxes=None):
        return engine.run_model(self, x)

    def _forward_impl(self, x):
        x = list(x)
        with engine.internal():
            names = [n for n, _ in self.named_children()]
            for i, n in enumerate(names):
                m = getattr(self, n)
                if not n.endswith("_fuse"):
                    engine.milestone(n)
                if n == "head":
                    x = m(x)
                elif n.endswith("_fuse"):
                    nxt = names[i + 1] if i + 1 < len(names) else "head"
                    x = m(x, defer_join=(nxt != "head") and engine.DEFER_JOIN)  # followed by a stage: the attention overlaps it
                else:
                    nxt = getattr(self, names[i + 1]) if i + 1 < len(names) else None
                    x = m(x, reserve=nxt.reserve(None) if isinstance(nxt, FuseFastAndSlow) else None)
        return x


@MODEL_REGISTRY.register()
class SlowFastShuffleNetV2(_EfficientTwoPathway):
    """Two-pathway ShuffleNetV2 + CMDA (reference custom_video_model_builder.py:449-617); children
    s1, s1_fuse, s2, s2_fuse, s3, s3_fuse, s4, s4_fuse, head."""

    def __init__(self, cfg):
        super(SlowFastShuffleNetV2, self).__init__()
        self.norm_module = get_norm(cfg)
        self.enable_detection = cfg.DETECTION.ENABLE
        self.num_pathways = 2
        width_mult = cfg.SLOWFAST.WIDTH_MULTI
        table = {0.25: [-1, 24, 32, 64, 128, 1024], 0.5: [-1, 24, 48, 96, 192, 1024],
                 1.0: [-1, 24, 116, 240, 464, 1024], 1.5: [-1, 24, 176, 352, 704, 1024],
                 2.0: [-1, 24, 224, 496, 976, 2048]}
        if width_mult not in table:
            raise ValueError("{} groups is not supported for 1x1 Grouped Convolutions".format(width_mult))
        self.stage_out_channels = table[width_mult]
        self.fast_stage_out_channels = [c // cfg.SLOWFAST.BETA_INV for c in self.stage_out_channels]
        self._construct_network(cfg)
        init_helper.init_weights(self, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN)

    def _construct_network(self, cfg):
        so, fo, bi = self.stage_out_channels, self.fast_stage_out_channels, cfg.SLOWFAST.BETA_INV
        self.s1 = stem_helper.ShuffleNetV2_Model_Stem(
            input_channels=[so[1], so[1] // bi], sample_size=cfg.DATA.CROP_SIZE,
            width_mult=[cfg.SLOWFAST.WIDTH_MULTI, cfg.SLOWFAST.WIDTH_MULTI / bi], img_dim=len(cfg.DATA.MEAN))
        for i in range(1, 5):
            fuse = FuseFastAndSlow(dim_in=[so[i], fo[i]], alpha=cfg.SLOWFAST.ALPHA, beta_inv=bi,
                                   norm_module=self.norm_module)
            setattr(self, "s{}_fuse".format(i), fuse)
            if i < 4:
                stage = ShuffleNetV2_Stage(input_channel=[so[i] + fo[i], fo[i] + so[i] // bi], idxstage=i - 1,
                                           slow_stage_out_channels=so, fast_stage_out_channels=fo)
                setattr(self, "s{}".format(i + 1), stage)
        if cfg.DETECTION.ENABLE:
            raise NotImplementedError("DETECTION.ENABLE is out of scope of the HIP path")
        self.head = head_helper.ShuffleNetV2BasicHead(
            input_channel=[so[4] + fo[4], fo[4] + so[4] // bi], last_channel=[so[-1], fo[-1]],
            num_classes=cfg.MODEL.NUM_CLASSES, dropout_rate=cfg.MODEL.DROPOUT_RATE, act_func=cfg.MODEL.HEAD_ACT)


@MODEL_REGISTRY.register()
class SlowFastGhostNet(_EfficientTwoPathway):
    """Two-pathway GhostNet + CMDA (reference custom_video_model_builder.py:793-1026); children
    s0, s1, s1_fuse, s2, s2_fuse, s3, s3_fuse, s4, s4_fuse, s5, head."""

    _STAGES = [
        [[3, 16, 16, 0, 1]],
        [[3, 48, 24, 0, 2], [3, 72, 24, 0, 1]],
        [[5, 72, 40, 0.25, 2], [5, 120, 40, 0.25, 1]],
        [[3, 240, 80, 0, 2], [3, 200, 80, 0, 1], [3, 184, 80, 0, 1], [3, 184, 80, 0, 1],
         [3, 480, 112, 0.25, 1], [3, 672, 112, 0.25, 1]],
        [[5, 672, 160, 0.25, 2], [5, 960, 160, 0, 1], [5, 960, 160, 0.25, 1], [5, 960, 160, 0, 1],
         [5, 960, 160, 0.25, 1]],
    ]

    def __init__(self, cfg):
        super(SlowFastGhostNet, self).__init__()
        self.norm_module = get_norm(cfg)
        self.enable_detection = cfg.DETECTION.ENABLE
        self.num_pathways = 2
        self.num_blocks = [4, 8, 4]
        wm, bi = cfg.SLOWFAST.WIDTH_MULTI, cfg.SLOWFAST.BETA_INV
        self.fast_cfgs, self.slow_cfgs = [], []
        for st in self._STAGES:  # note the float floor-division on the fast widths (:852-861)
            self.fast_cfgs.append([[c[0], _make_divisible(c[1] * wm // bi, 4), _make_divisible(c[2] * wm // bi, 4),
                                    c[3], c[4]] for c in st])
            self.slow_cfgs.append([[c[0], _make_divisible(c[1] * wm, 4), _make_divisible(c[2] * wm, 4), c[3], c[4]]
                                   for c in st])
        print(self.slow_cfgs)
        print(self.fast_cfgs)
        self._construct_network(cfg)
        init_helper.init_weights(self, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN)

    def _construct_network(self, cfg):
        wm, bi = cfg.SLOWFAST.WIDTH_MULTI, cfg.SLOWFAST.BETA_INV
        sc, fc = self.slow_cfgs, self.fast_cfgs
        widths = [_make_divisible(16 * wm, 4), _make_divisible(16 * wm // bi, 4)]
        out_ch = [int(1280 * wm), int(1280 * wm // bi)]
        self.s0 = stem_helper.GhostNet_Model_Stem(input_channels=widths, sample_size=cfg.DATA.CROP_SIZE,
                                                  img_dim=len(cfg.DATA.MEAN))
        self.s1 = GhostNet_Stage(input_channel=widths, slow_cfg=sc[0], fast_cfg=fc[0])
        for i in range(4):  # s{i+1}_fuse then s{i+2}
            fuse = FuseFastAndSlow(dim_in=[sc[i][-1][2], fc[i][-1][2]], alpha=cfg.SLOWFAST.ALPHA, beta_inv=bi,
                                   norm_module=self.norm_module)
            setattr(self, "s{}_fuse".format(i + 1), fuse)
            if i < 3:  # reference indexes cfg[i][0][2] for the slow/fast base widths (:903-906)
                inp = [sc[i][0][2] + fc[i][-1][2], fc[i][0][2] + sc[i][-1][2] // bi]
            else:
                inp = [sc[3][-1][2] + fc[3][-1][2], fc[3][-1][2] + sc[3][-1][2] // bi]
            setattr(self, "s{}".format(i + 2), GhostNet_Stage(input_channel=inp, slow_cfg=sc[i + 1],
                                                             fast_cfg=fc[i + 1]))
        self.head = head_helper.GhostNetBasicHead(
            input_channel=[sc[4][-1][2], fc[4][-1][2]], mid_channel=[sc[4][-1][1], fc[4][-1][1]],
            output_channel=out_ch, num_classes=cfg.MODEL.NUM_CLASSES, dropout_rate=cfg.MODEL.DROPOUT_RATE,
            act_func=cfg.MODEL.HEAD_ACT)


_MOBILE_NET_V2_CONFIGS = {  # t, c, n, s (custom_video_model_builder.py:1029-1054)
    "slow_interverted_residual_setting": [
        [1, 16, 1, (1, 1, 1)], [6, 24, 2, (1, 2, 2)], [6, 32, 3, (1, 2, 2)], [6, 64, 4, (1, 2, 2)],
        [6, 96, 3, (1, 1, 1)], [6, 160, 3, (1, 2, 2)], [6, 320, 1, (1, 1, 1)]],
    "fast_interverted_residual_setting": [
        [1, 16, 1, (1, 1, 1)], [6, 24, 2, (1, 2, 2)], [6, 32, 3, (1, 2, 2)], [6, 64, 4, (1, 2, 2)],
        [6, 96, 3, (1, 1, 1)], [6, 160, 3, (1, 2, 2)], [6, 320, 1, (1, 1, 1)]],
}


@MODEL_REGISTRY.register()
class SlowFastMoibleNetV2(_EfficientTwoPathway):
    """Two-pathway MobileNetV2 + CMDA (reference custom_video_model_builder.py:1057-1285, class name spelled
    as the reference spells it); children s1, s2, s3_fuse, s4, s4_fuse, s5, s5_fuse, s6, s7, s7_fuse, s8, head."""

    def __init__(self, cfg):
        super(SlowFastMoibleNetV2, self).__init__()
        self.norm_module = get_norm(cfg)
        self.enable_detection = cfg.DETECTION.ENABLE
        self.num_pathways = 2
        self._construct_network(cfg)
        init_helper.init_weights(self, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN)

    def _construct_network(self, cfg):
        wm, bi = cfg.SLOWFAST.WIDTH_MULTI, cfg.SLOWFAST.BETA_INV
        width_per_group = 32
        self.last_channel = int(1280 * wm) if wm > 1.0 else 1280
        slow = _MOBILE_NET_V2_CONFIGS["slow_interverted_residual_setting"]
        fast = _MOBILE_NET_V2_CONFIGS["fast_interverted_residual_setting"]
        self.s1 = stem_helper.MobilenetV2_Model_Stem(
            input_channels=[width_per_group, width_per_group], sample_size=cfg.DATA.CROP_SIZE,
            width_mult=[wm, wm / bi], img_dim=len(cfg.DATA.MEAN))

        def stage(inp, a, b):
            return MobileNetV2_Stage(input_channel=inp, slow_residual_setting=slow[a:b],
                                     fast_residual_setting=fast[a:b], width_mult=wm, beta_inv=bi)

        def fuse(row):
            return FuseFastAndSlow(dim_in=[int(slow[row][1] * wm), int(slow[row][1] * wm) // bi],
                                   alpha=cfg.SLOWFAST.ALPHA, beta_inv=bi, norm_module=self.norm_module)

        def fused_in(row):  # channel counts after the CMDA that follows settings row `row`
            c = slow[row][1] * wm
            return [int(c + c // bi), int(c // bi + c // bi)]

        self.s2 = stage([int(width_per_group * wm), int(width_per_group * wm // bi)], 0, 2)
        self.s3_fuse = fuse(1)
        self.s4 = stage(fused_in(1), 2, 3)
        self.s4_fuse = fuse(2)
        self.s5 = stage(fused_in(2), 3, 4)
        self.s5_fuse = fuse(3)
        self.s6 = stage(fused_in(3), 4, 5)
        self.s7 = stage([int(slow[4][1] * wm), int(slow[4][1] * wm // bi)], 5, 6)
        self.s7_fuse = fuse(5)
        self.s8 = stage(fused_in(5), 6, 7)
        if cfg.DETECTION.ENABLE:
            raise NotImplementedError("DETECTION.ENABLE is out of scope of the HIP path")
        self.head = head_helper.MobileNetV2BasicHead(
            input_channel=[int(slow[6][1] * wm), int(slow[6][1] * wm // bi)],
            last_channel=[self.last_channel, self.last_channel // bi], num_classes=cfg.MODEL.NUM_CLASSES,
            dropout_rate=cfg.MODEL.DROPOUT_RATE, act_func=cfg.MODEL.HEAD_ACT)


@MODEL_REGISTRY.register()
class SlowFastShuffleNet(_EfficientTwoPathway):
    """Two-pathway ShuffleNet v1 + CMDA (reference custom_video_model_builder.py:620-789); children s1, s1_fuse,
    s2, s2_fuse, s3, s3_fuse, s4, s4_fuse, head."""

    _OUT_PLANES = {1: [24, 144, 288, 567], 2: [24, 200, 400, 800], 3: [24, 240, 480, 960],
                   4: [24, 272, 544, 1088], 8: [24, 384, 768, 1536]}

    def __init__(self, cfg):
        super(SlowFastShuffleNet, self).__init__()
        self.norm_module = get_norm(cfg)
        self.enable_detection = cfg.DETECTION.ENABLE
        self.num_pathways = 2
        width_mult = cfg.SLOWFAST.WIDTH_MULTI
        groups = cfg.SLOWFAST.GROUPS
        self.num_blocks = [4, 8, 4]
        self.groups = groups
        if groups not in self._OUT_PLANES:
            raise ValueError("{} groups is not supported for 1x1 Grouped Convolutions".format(groups))
        out_planes = self._OUT_PLANES[groups]
        self.stage_out_channels = [int(i * width_mult) for i in out_planes]
        self.fast_stage_out_channels = [c // cfg.SLOWFAST.BETA_INV for c in self.stage_out_channels]
        self.in_planes = out_planes[0]
        self._construct_network(cfg)
        init_helper.init_weights(self, cfg.MODEL.FC_INIT_STD, cfg.RESNET.ZERO_INIT_FINAL_BN)

    def _construct_network(self, cfg):
        so, fo, bi = self.stage_out_channels, self.fast_stage_out_channels, cfg.SLOWFAST.BETA_INV
        self.s1 = stem_helper.ShuffleNet_Model_Stem(input_channels=[so[0], fo[0]], sample_size=cfg.DATA.CROP_SIZE,
                                                    img_dim=len(cfg.DATA.MEAN))
        for i in range(4):
            fuse = FuseFastAndSlow(dim_in=[so[i], fo[i]], alpha=cfg.SLOWFAST.ALPHA, beta_inv=bi,
                                   norm_module=self.norm_module)
            setattr(self, "s{}_fuse".format(i + 1), fuse)
            if i < 3:
                stage = ShuffleNet_Stage(input_channel=[so[i] + fo[i], fo[i] + so[i] // bi],
                                         slow_stage_out_channels=so[i + 1], fast_stage_out_channels=fo[i + 1],
                                         num_block=self.num_blocks[i], group=cfg.SLOWFAST.GROUPS)
                setattr(self, "s{}".format(i + 2), stage)
        if cfg.DETECTION.ENABLE:
            raise NotImplementedError("DETECTION.ENABLE is out of scope of the HIP path")
        self.head = head_helper.ShuffleNetBasicHead(
            input_channel=[so[3] + fo[3], fo[3] + so[3] // bi], num_classes=cfg.MODEL.NUM_CLASSES,
            dropout_rate=cfg.MODEL.DROPOUT_RATE, act_func=cfg.MODEL.HEAD_ACT)
