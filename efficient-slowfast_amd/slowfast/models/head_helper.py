"""Classification heads (reference head_helper.py).  Pooling kernels write into one concat buffer, the
projection is the MFMA GEMM, the eval-mode softmax+mean is one small kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import sfhip
from . import engine


def _act_code(name):
    return {"softmax": sfhip.ACT_SOFTMAX, "sigmoid": sfhip.ACT_SIGMOID, "relu": sfhip.ACT_RELU}[name]


def _project(pooled, linear, dropout, training):
    """[N,T,H,W,C] -> logits Act [N,T,H,W,K] through nn.Linear's parameters (1x1x1 GEMM)."""
    t = engine.tape()
    if dropout is not None and training and dropout.p > 0.0:
        # nn.Dropout on the pooled [N, C] features through the same ATen kernel nn.Dropout dispatches to, so the mask
        # and the generator's consumption equal the reference's under a seed (head_helper.py:207-208)
        keep = 1.0 - dropout.p
        src = pooled
        out, keep_mask = torch.native_dropout(src.buf, dropout.p, True)
        pooled = sfhip.Act(out)
        if t is not None:
            dropped = pooled
            t.record(lambda: t.grad_of(src).buf.add_(t.grad_of(dropped).buf * keep_mask.to(torch.float32) / keep))
    wp = engine._cached(linear, "_sf_wp", engine._key(linear.weight),
                        lambda: sfhip.pack_conv_weight(linear.weight.reshape(linear.out_features, -1, 1, 1, 1)))
    logits = sfhip.conv(pooled, wp, (1, 1, 1), bias=linear.bias)
    if t is not None:
        engine._record_conv(pooled, linear.weight, linear.bias, wp.shape, lambda: t.grad_of(logits), (1, 1, 1),
                            (1, 1, 1), (0, 0, 0), (1, 1, 1),
                            unpack=lambda dwp: dwp[:, 0, :linear.in_features].contiguous())
        t.out_act = logits
    return logits


LOGITS_TAP = None  # tests set this to a callable(logits_tensor [N,T,H,W,K]) to observe the pre-activation logits


def _finish(logits, training, act_name):
    if LOGITS_TAP is not None:
        LOGITS_TAP(logits.buf)
    if not training:
        return sfhip.head_act_mean(logits, _act_code(act_name))
    # train: the reference returns x.view(N, -1) of the raw projection (head_helper.py:222)
    return logits.buf.reshape(logits.N, -1)


class ResNetBasicHead(nn.Module):
    """AvgPool3d(pool_size, stride 1) | AdaptiveAvgPool3d(1) per pathway -> cat -> Dropout -> Linear ->
    eval: Softmax(dim=4) + mean over T,H,W (head_helper.py:133-223)."""

    def __init__(self, dim_in, num_classes, pool_size, dropout_rate=0.0, act_func="softmax"):
        super(ResNetBasicHead, self).__init__()
        assert len({len(pool_size), len(dim_in)}) == 1, "pathway dimensions are not consistent."
        self.num_pathways = len(pool_size)
        for pathway in range(self.num_pathways):
            if pool_size[pathway] is None:
                avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1))
            else:
                avg_pool = nn.AvgPool3d(pool_size[pathway], stride=1)
            self.add_module("pathway{}_avgpool".format(pathway), avg_pool)
        if dropout_rate > 0.0:
            self.dropout = nn.Dropout(dropout_rate)
        self.projection = nn.Linear(sum(dim_in), num_classes, bias=True)
        if act_func == "softmax":
            self.act = nn.Softmax(dim=4)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        else:
            raise NotImplementedError("{} is not supported as an activation function.".format(act_func))
        self._act_name = act_func

    def forward(self, inputs):
        assert len(inputs) == self.num_pathways, "Input tensor does not contain {} pathway".format(self.num_pathways)
        xs = engine.enter(inputs)
        total = sum(x.C for x in xs)
        cat, off = None, 0
        for pathway, x in enumerate(xs):
            m = getattr(self, "pathway{}_avgpool".format(pathway))
            k = (x.T, x.H, x.W) if isinstance(m, nn.AdaptiveAvgPool3d) else tuple(
                m.kernel_size if isinstance(m.kernel_size, (tuple, list)) else [m.kernel_size] * 3)
            to, ho, wo = x.T - k[0] + 1, x.H - k[1] + 1, x.W - k[2] + 1
            if cat is None:
                cat = sfhip.new_act(x, x.N, to, ho, wo, total)
            assert (cat.T, cat.H, cat.W) == (to, ho, wo), "pathway pooled sizes differ"
            if (to, ho, wo) == (1, 1, 1):  # window == extent: a global mean (parallel tree reduction)
                _global_mean(x, cat.slice(off, x.C))
            else:
                if engine.tape() is not None:
                    raise NotImplementedError("training through a fully-convolutional head (pooled extent > 1)")
                sfhip.pool(x, k, (1, 1, 1), avg=True, out=cat.slice(off, x.C))
            off += x.C
        logits = _project(cat, self.projection, getattr(self, "dropout", None), self.training)
        return _finish(logits, self.training, self._act_name)


class _ConvBnAct(nn.Module):
    """conv -> bn1 -> act1 (ghostnet_helper.py:55-68 / head_helper.py:612-627)."""

    def __init__(self, in_chs, out_chs, kernel_size, stride=1):
        super(_ConvBnAct, self).__init__()
        self.conv = nn.Conv3d(in_chs, out_chs, kernel_size, stride, kernel_size // 2, bias=False)
        self.bn1 = nn.BatchNorm3d(out_chs)
        self.act1 = nn.ReLU(inplace=True)

    def forward(self, x):
        return engine.conv_bn_act(x, self.conv, self.bn1, relu=True)


def _global_mean(x, out=None):
    """[N,T,H,W,C] -> [N,1,1,1,C] mean (F.avg_pool3d(x, x.size()[-3:])); taped."""
    return engine.global_mean(x, out)


class MobileNetV2BasicHead(nn.Module):
    """per pathway 1x1x1 conv+BN+ReLU6 -> global avg-pool -> cat -> Dropout+Linear -> eval softmax+mean
    (head_helper.py:436-486)."""

    def __init__(self, input_channel, last_channel, num_classes, dropout_rate, act_func="softmax"):
        super(MobileNetV2BasicHead, self).__init__()
        self.num_pathways = len(input_channel)
        for pathway in range(self.num_pathways):
            seq = nn.Sequential(nn.Conv3d(input_channel[pathway], last_channel[pathway], 1, 1, 0, bias=False),
                                nn.BatchNorm3d(last_channel[pathway]), nn.ReLU6(inplace=True))
            self.add_module("pathway{}_conv1x1x1".format(pathway), seq)
        if act_func == "softmax":
            self.act = nn.Softmax(dim=4)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        self._act_name = act_func
        self.classifier = nn.Sequential(nn.Dropout(dropout_rate), nn.Linear(sum(last_channel), num_classes, bias=True))

    def forward(self, inputs):
        xs = engine.enter(inputs)
        outs = []
        for pathway, x in enumerate(xs):
            seq = getattr(self, "pathway{}_conv1x1x1".format(pathway))
            outs.append(engine.conv_bn_act(x, seq[0], seq[1], relu=6))
        cat = sfhip.new_act(outs[0], outs[0].N, 1, 1, 1, sum(o.C for o in outs))
        off = 0
        for o in outs:
            _global_mean(o, cat.slice(off, o.C))
            off += o.C
        logits = _project(cat, self.classifier[1], self.classifier[0], self.training)
        return _finish(logits, self.training, self._act_name)


class ShuffleNetBasicHead(nn.Module):
    """global avg-pool per pathway -> cat -> Dropout+Linear -> eval softmax+mean (head_helper.py:562-609)."""

    def __init__(self, input_channel, num_classes, dropout_rate, act_func="softmax"):
        super(ShuffleNetBasicHead, self).__init__()
        self.num_pathways = len(input_channel)
        if act_func == "softmax":
            self.act = nn.Softmax(dim=4)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        self._act_name = act_func
        self.classifier = nn.Sequential(nn.Dropout(dropout_rate), nn.Linear(sum(input_channel), num_classes, bias=True))

    def forward(self, inputs):
        xs = engine.enter(inputs)
        cat = sfhip.new_act(xs[0], xs[0].N, 1, 1, 1, sum(x.C for x in xs))
        off = 0
        for x in xs:
            _global_mean(x, cat.slice(off, x.C))
            off += x.C
        logits = _project(cat, self.classifier[1], self.classifier[0], self.training)
        return _finish(logits, self.training, self._act_name)


class ShuffleNetV2BasicHead(nn.Module):
    """per pathway 1x1x1 conv+BN+ReLU -> global avg-pool -> cat -> Dropout+Linear -> eval softmax+mean
    (head_helper.py:499-557)."""

    def __init__(self, input_channel, last_channel, num_classes, dropout_rate, act_func="softmax"):
        super(ShuffleNetV2BasicHead, self).__init__()
        self.num_pathways = len(input_channel)
        for pathway in range(self.num_pathways):
            inner = nn.Sequential(nn.Conv3d(input_channel[pathway], last_channel[pathway], 1, 1, 0, bias=False),
                                  nn.BatchNorm3d(last_channel[pathway]), nn.ReLU(inplace=True))
            self.add_module("pathway{}_conv1x1x1".format(pathway), nn.Sequential(inner))
        if act_func == "softmax":
            self.act = nn.Softmax(dim=4)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        self._act_name = act_func
        self.classifier = nn.Sequential(nn.Dropout(dropout_rate), nn.Linear(sum(last_channel), num_classes, bias=True))

    def forward(self, inputs):
        xs = engine.enter(inputs)
        outs = []
        for pathway, x in enumerate(xs):
            seq = getattr(self, "pathway{}_conv1x1x1".format(pathway))[0]
            outs.append(engine.conv_bn_act(x, seq[0], seq[1], relu=True))
        cat = sfhip.new_act(outs[0], outs[0].N, 1, 1, 1, sum(o.C for o in outs))
        off = 0
        for o in outs:
            _global_mean(o, cat.slice(off, o.C))
            off += o.C
        logits = _project(cat, self.classifier[1], self.classifier[0], self.training)
        return _finish(logits, self.training, self._act_name)


class GhostNetBasicHead(nn.Module):
    """ConvBnAct -> global avg-pool -> conv_head(+bias) -> ReLU per pathway -> cat -> Dropout+Linear ->
    eval: self.act + mean.  Bug-compatible: self.act (softmax) is overwritten by nn.ReLU
    (head_helper.py:640-643 vs :653), so eval returns mean(relu(logits)), not probabilities."""

    def __init__(self, input_channel, mid_channel, output_channel, num_classes, dropout_rate, act_func="softmax"):
        super(GhostNetBasicHead, self).__init__()
        self.num_pathways = len(input_channel)
        self.input_channel, self.mid_channel = input_channel, mid_channel
        self.stage5_conv_slow = _ConvBnAct(input_channel[0], mid_channel[0], 1)
        self.stage5_conv_fast = _ConvBnAct(input_channel[1], mid_channel[1], 1)
        self.conv_head_slow = nn.Conv3d(mid_channel[0], output_channel[0], 1, 1, 0, bias=True)
        self.conv_head_fast = nn.Conv3d(mid_channel[1], output_channel[1], 1, 1, 0, bias=True)
        self.act = nn.ReLU(inplace=True)
        self.classifier = nn.Sequential(nn.Dropout(dropout_rate), nn.Linear(sum(output_channel), num_classes, bias=True))

    def forward(self, inputs):
        xs = engine.enter(inputs)
        total = self.conv_head_slow.out_channels + self.conv_head_fast.out_channels
        cat = sfhip.new_act(xs[0], xs[0].N, 1, 1, 1, total)
        off = 0
        for x, stage5, conv_head in ((xs[0], self.stage5_conv_slow, self.conv_head_slow),
                                     (xs[1], self.stage5_conv_fast, self.conv_head_fast)):
            y = _global_mean(stage5(x))
            # conv_head (+bias) + ReLU on the pooled [N, C] features: a 1x1x1 conv over N rows on the library's GEMM
            # kernels, written straight into its slice of the concatenated feature vector
            engine.conv_bn_act(y, conv_head, relu=True, out=cat.slice(off, conv_head.out_channels))
            off += conv_head.out_channels
        logits = _project(cat, self.classifier[1], self.classifier[0], self.training)
        return _finish(logits, self.training, "relu")
