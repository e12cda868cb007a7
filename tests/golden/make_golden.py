#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REFERENCE itself (CPU, fp32).

Runs only in the build container (needs /root/reference); the GPU box and the test-suite only
read the resulting ``tests/golden/*.npz``.  Re-run: ``python tests/golden/make_golden.py``.

Every fixture holds *data only*: the cfg overrides and oracle hyper-parameters (JSON), the seeds
the parameters / clips are re-generated from (``paramgen.py``), the reference state_dict's key
and shape list, and the reference's outputs (strided samples of every top-level child's output,
pre-activation logits, eval output, train-mode logits, CE loss and sampled gradients).
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refimport import import_reference, ref_yaml  # noqa: E402
from paramgen import fill_state_dict, make_clip, make_upstream, sample_activation, upstream_seed  # noqa: E402

PARAM_SEED = 7
CLIP_SEED = 3

DUAL_YAML = "SLOWFAST_DUAL_8x8_R50_stepwise_multigrid.yaml"
COMMON = ["NUM_GPUS", 0, "MULTIGRID.SHORT_CYCLE", False, "MULTIGRID.LONG_CYCLE", False]


def small(s, t):
    return ["DATA.NUM_FRAMES", t, "DATA.CROP_SIZE", s, "DATA.TRAIN_CROP_SIZE", s, "DATA.TEST_CROP_SIZE", s]


CASES = [
    # BASELINE.json configs[0] exactly: SlowFastShuffleNetV2 w0.25, 4x16 (slow 4 / fast 32, alpha 8), 32^2, B=2
    dict(name="shufflenetv2_cfg1", yaml="SLOWFAST_SHUFFLENETV2_8x8_R50_stepwise_multigrid.yaml",
         model="SlowFastShuffleNetV2", batch=2, t=32, alpha=8, size=32,
         over=["SLOWFAST.WIDTH_MULTI", 0.25, "SLOWFAST.ALPHA", 8] + small(32, 32)),
    # configs[1] architecture (SlowFast 8x8 R50, FuseFastToSlow K=7) at S=64, T=16
    dict(name="slowfast_r50_s64", yaml=DUAL_YAML, model="SlowFast", batch=1, t=16, alpha=4, size=64,
         over=small(64, 16)),
    # configs[2]/[3] architecture (SlowFastDualAttention 8x8 R50, CMDA) at S=64, T=16, B=2
    dict(name="dual_r50_s64", yaml=DUAL_YAML, model="SlowFastDualAttention", batch=2, t=16, alpha=4, size=64,
         over=small(64, 16)),
    # configs[4] architecture (SlowFastGhostNet w2.0 + CMDA) at S=64, T=16 (224^2 has no runnable reference)
    dict(name="ghostnet_w2_s64", yaml="SLOWFAST_GHOSTNET_8x8_R50_stepwise_multigrid.yaml",
         model="SlowFastGhostNet", batch=1, t=16, alpha=4, size=64,
         over=["SLOWFAST.WIDTH_MULTI", 2.0, "SLOWFAST.ALPHA", 4] + small(64, 16)),
    # ... and at the LARGEST size the reference itself can run (SURVEY §8c / BASELINE.md §2: dense attention over
    # N = 8 x 56 x 56 = 25 088 positions at s1_fuse, 2 x 2.5 GB per call): S = 112, T = 32 (the 32x2 sampling of cfg #5)
    dict(name="ghostnet_w2_s112", yaml="SLOWFAST_GHOSTNET_8x8_R50_stepwise_multigrid.yaml",
         model="SlowFastGhostNet", batch=1, t=32, alpha=4, size=112,
         over=["SLOWFAST.WIDTH_MULTI", 2.0, "SLOWFAST.ALPHA", 4] + small(112, 32)),
    # SlowFastMoibleNetV2 (sic) w1.0 + CMDA at S=64, T=16
    dict(name="mobilenetv2_w1_s64", yaml="SLOWFAST_MOBILENETV2_8x8_R50_stepwise_multigrid.yaml",
         model="SlowFastMoibleNetV2", batch=2, t=16, alpha=4, size=64,
         over=["SLOWFAST.WIDTH_MULTI", 1.0, "SLOWFAST.ALPHA", 4] + small(64, 16)),
    # SubBatchNorm3d (multigrid long cycle), NUM_SPLITS 2 over a batch of 4
    dict(name="dual_r50_subbn_s64", yaml="SLOWFAST_DUAL_8x8_R50_stepwise_multigrid.yaml",
         model="SlowFastDualAttention", batch=4, t=16, alpha=4, size=64,
         over=["BN.NORM_TYPE", "sub_batchnorm", "BN.NUM_SPLITS", 2] + small(64, 16)),
    # SURVEY §8(b) registry surface: single-pathway ResNet, I3D (temporal kernels 5/3, MaxPool3d([2,1,1]) after res2)
    dict(name="i3d_r50_s64", yaml="I3D_8x8_R50.yaml", model="ResNet", batch=2, t=8, alpha=1, size=64,
         over=small(64, 8), single=True),
    # ... and Slow at DEPTH 18 = (2,2,2,2) bottleneck blocks (TIRED_SLOW_*_R18 YAMLs).  RESNET.TRANS_FUNC
    # basic_transform cannot be built by the reference: ResBlock passes `dilation=` to BasicTransform, which does
    # not take it (resnet_helper.py:338-349 vs :31-42) -> TypeError, so there is nothing to pin for it.
    dict(name="slow_r18_s64", yaml="SLOW_8x8_R50.yaml", model="ResNet", batch=2, t=8, alpha=1, size=64,
         over=["RESNET.DEPTH", 18, "RESNET.NUM_BLOCK_TEMP_KERNEL", [[2], [2], [2], [2]]] + small(64, 8), single=True),
    # Nonlocal blocks.  SLOWFAST_NLN (dot_product, slow pathway res3 [1,3] / res4 [1,3,5],
    # keys max-pooled (1,2,2)) and C2D_NLN (softmax instantiation, single pathway, frame groups folded into the batch)
    dict(name="slowfast_nln_s64", yaml="SLOWFAST_NLN_8x8_R50.yaml", model="SlowFast", batch=2, t=16, alpha=4, size=64,
         over=small(64, 16)),
    dict(name="c2d_nln_s64", yaml="C2D_NLN_8x8_R50.yaml", model="ResNet", batch=2, t=8, alpha=1, size=64,
         over=["NONLOCAL.GROUP", [[1], [2], [1], [1]]] + small(64, 8), single=True),
    # SlowFastShuffleNet (v1, GROUPS 1 as its YAML) + CMDA at S=64, T=16
    dict(name="shufflenet_g1_s64", yaml="SLOWFAST_SHUFFLENET_8x8_R50_stepwise_multigrid.yaml",
         model="SlowFastShuffleNet", batch=2, t=16, alpha=4, size=64,
         over=["SLOWFAST.ALPHA", 4] + small(64, 16)),
    # ... and the reference's PUBLISHED ShuffleNet-v1 configuration: width 2.0, 3 groups (README.md:260, 53.84 top-1;
    # wdf_all_run_scripts/run_shufflenet_w2_g3.sh:12) — grouped 1x1 convs + channel_shuffle(., 3)
    dict(name="shufflenet_w2_g3_s64", yaml="SLOWFAST_SHUFFLENET_8x8_R50_stepwise_multigrid.yaml",
         model="SlowFastShuffleNet", batch=2, t=16, alpha=4, size=64,
         over=["SLOWFAST.ALPHA", 4, "SLOWFAST.WIDTH_MULTI", 2.0, "SLOWFAST.GROUPS", 3] + small(64, 16),
         grad_keys=["s1.pathway0_stem.0.weight", "s2.pathway0_channel_480.features.0.shortcut.0.weight",
                    "s2.pathway0_channel_480.features.0.conv1.weight", "s3.pathway1_channel_120.features.2.conv2.weight",
                    "s3.pathway1_channel_120.features.1.conv3.weight", "s3_fuse.bn_s2f.weight",
                    "s4.pathway0_channel_1920.features.1.conv3.weight", "head.classifier.1.weight"]),
]

GRAD_KEYS = {
    "SlowFast": ["s1.pathway0_stem.conv.weight", "s1_fuse.conv_f2s.weight", "s3.pathway1_res1.branch2.b.weight",
                 "s4.pathway0_res0.branch2.a.weight", "s5.pathway0_res2.branch2.c_bn.weight",
                 "head.projection.weight"],
    "ResNet": ["s1.pathway0_stem.conv.weight", "s2.pathway0_res1.branch2.b.weight", "s3.pathway0_res0.branch1.weight",
               "s4.pathway0_res1.branch2.a.weight", "s5.pathway0_res1.branch2.b_bn.weight", "head.projection.weight"],
    "SlowFastDualAttention": ["s1.pathway1_stem.conv.weight", "s2_fuse.attention_spatial_s2f.query_conv.weight",
                              "s2_fuse.attention_spatial_s2f.gamma", "s3_fuse.attention_channel_f2s.conv.weight",
                              "s4.pathway0_res0.branch2.a.weight", "s3_fuse.bn_s2f.weight",
                              "head.projection.weight"],
    "SlowFastShuffleNetV2": ["s1.pathway0_stem.0.weight", "s2_fuse.attention_spatial_s2f.value_conv.weight",
                             "s3.pathway0_channel_64.features.1.banch2.3.weight", "head.classifier.1.weight"],
    "SlowFastMoibleNetV2": ["s1.pathway0_stem.features.0.weight", "s4.pathway0_channel_32.features.1.conv.3.weight",
                            "s5_fuse.attention_spatial_s2f.gamma", "s7.pathway1_channel_160.features.0.conv.0.weight",
                            "head.classifier.1.weight"],
    "SlowFastShuffleNet": ["s1.pathway0_stem.0.weight", "s2.pathway0_channel_144.features.0.shortcut.0.weight",
                           "s3.pathway1_channel_36.features.2.conv2.weight", "s3_fuse.bn_s2f.weight",
                           "s4.pathway0_channel_567.features.1.conv3.weight", "head.classifier.1.weight"],
    "SlowFastGhostNet": ["s0.pathway0_stem.0.weight", "s3.pathway0_channel_80.features.0.se.conv_reduce.weight",
                         "s2_fuse.attention_spatial_s2f.gamma", "head.classifier.1.weight"],
}


def plain_cfg(node):
    """The fully resolved cfg (YAML + overrides) as plain nested dicts, so tests can rebuild it without
    the reference's YAML files."""
    return {k: plain_cfg(v) if isinstance(v, dict) else (list(v) if isinstance(v, tuple) else v)
            for k, v in node.items()}


def hparams_from_cfg(cfg):
    return dict(
        alpha=cfg.SLOWFAST.ALPHA, beta_inv=cfg.SLOWFAST.BETA_INV, depth=cfg.RESNET.DEPTH,
        width_per_group=cfg.RESNET.WIDTH_PER_GROUP, num_groups=cfg.RESNET.NUM_GROUPS,
        fusion_conv_channel_ratio=cfg.SLOWFAST.FUSION_CONV_CHANNEL_RATIO,
        fusion_kernel=cfg.SLOWFAST.FUSION_KERNEL_SZ,
        spatial_strides=[s[0] for s in cfg.RESNET.SPATIAL_STRIDES],
        spatial_dilations=[s[0] for s in cfg.RESNET.SPATIAL_DILATIONS],
        num_block_temp_kernel=[list(x) for x in cfg.RESNET.NUM_BLOCK_TEMP_KERNEL],
        num_frames=cfg.DATA.NUM_FRAMES, crop_size=cfg.DATA.CROP_SIZE, num_classes=cfg.MODEL.NUM_CLASSES,
        short_cycle=bool(cfg.MULTIGRID.SHORT_CYCLE), head_act=cfg.MODEL.HEAD_ACT,
        width_multi=cfg.SLOWFAST.WIDTH_MULTI, eps=1e-5, groups=cfg.SLOWFAST.get("GROUPS", 1), arch=cfg.MODEL.ARCH,
        nonlocal_group=[list(g) for g in cfg.NONLOCAL.GROUP], nonlocal_pool=[[list(q) for q in st] for st in cfg.NONLOCAL.POOL],
        nonlocal_instantiation=cfg.NONLOCAL.INSTANTIATION,
    )


def run_case(case, get_cfg, build_model):
    t0 = time.time()
    cfg = get_cfg()
    cfg.merge_from_file(ref_yaml(case["yaml"]))
    over = COMMON + ["MODEL.MODEL_NAME", case["model"]] + case["over"]
    cfg.merge_from_list(over)
    torch.manual_seed(0)
    model = build_model(cfg)
    sd = model.state_dict()
    fill_state_dict(sd, PARAM_SEED)
    slow, fast = make_clip(CLIP_SEED, case["batch"], case["t"], case["alpha"], case["size"])
    single = bool(case.get("single"))

    def clips(grad=False):
        arrs = [fast] if single else [slow, fast]
        return [torch.from_numpy(a.copy()).requires_grad_(grad) for a in arrs]

    out = {
        "meta": json.dumps(dict(name=case["name"], model=case["model"], yaml=case["yaml"],
                                overrides=[str(o) if not isinstance(o, (int, float, bool)) else o for o in over],
                                cfg_dump=plain_cfg(cfg), hparams=hparams_from_cfg(cfg), param_seed=PARAM_SEED, clip_seed=CLIP_SEED,
                                batch=case["batch"], t=case["t"], alpha=case["alpha"], size=case["size"],
                                single=single, torch=torch.__version__)),
        "sd_keys": np.array(list(sd.keys())),
        "sd_shapes": np.array([json.dumps(list(v.shape)) for v in sd.values()]),
        "children": np.array([n for n, _ in model.named_children()]),
    }

    acts = {}

    def hook(name):
        def f(m, i, o):
            acts[name] = [x.detach().clone() for x in o] if isinstance(o, (list, tuple)) else o.detach().clone()
        return f

    handles = [m.register_forward_hook(hook(n)) for n, m in model.named_children()]
    head_fc = model.head.projection if hasattr(model.head, "projection") else model.head.classifier[1]
    handles.append(head_fc.register_forward_hook(hook("logits")))

    # ---- eval forward (what perform_test does at test_net.py:92)
    model.eval()
    with torch.no_grad():
        probs = model(clips())
    for k, v in acts.items():
        if k == "head":
            continue
        vs = v if isinstance(v, list) else [v]
        for i, a in enumerate(vs):
            s, amax, mean = sample_activation(a.numpy())
            tag = "eval/%s/%d" % (k, i) if isinstance(v, list) else "eval/%s" % k
            out[tag] = s
            out[tag + "/stats"] = np.array([amax, mean], np.float64)
            out[tag + "/shape"] = np.array(a.shape)
    out["eval/logits_full"] = acts["logits"].numpy().reshape(case["batch"], -1)
    out["eval/out"] = probs.numpy()

    # ---- train forward + backward (train_net.py:78, :84-96), dropout off so it is deterministic
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    acts.clear()
    child_in = {}

    def pre_hook(name):
        def f(m, args):
            a = args[0]
            child_in[name] = [t.detach().clone() for t in a] if isinstance(a, (list, tuple)) else a.detach().clone()
        return f

    handles += [m.register_forward_pre_hook(pre_hook(n)) for n, m in model.named_children()]
    xs = clips(grad=True)
    xin = list(xs)
    logits = model(xin)
    labels = torch.from_numpy(np.random.RandomState(11).randint(0, cfg.MODEL.NUM_CLASSES, case["batch"]))
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    out["train/logits"] = logits.detach().numpy()
    out["train/labels"] = labels.numpy()
    out["train/loss"] = np.array([loss.item()], np.float64)
    for k in ("s2", "s3_fuse", "s5"):
        if k in acts:
            for i, a in enumerate(acts[k]):
                s, amax, mean = sample_activation(a.numpy())
                out["train/%s/%d" % (k, i)] = s
    params = dict(model.named_parameters())
    for k in case.get("grad_keys") or GRAD_KEYS[case["model"]]:
        g = params[k].grad
        s, amax, mean = sample_activation(g.numpy(), 4096)
        out["grad/" + k] = s
        out["grad/" + k + "/stats"] = np.array([amax, float(g.norm())], np.float64)
    # running statistics after ONE training forward (momentum update; SubBatchNorm3d: per-split buffers)
    after = model.state_dict()
    bufs = [k for k in after if k.endswith("running_mean") or k.endswith("running_var")]
    for k in bufs[:4] + bufs[len(bufs) // 2:len(bufs) // 2 + 4] + bufs[-4:]:
        out["train_buffers/" + k] = after[k].numpy().copy()
    for i, nm in enumerate(("fast",) if single else ("slow", "fast")):
        s, amax, mean = sample_activation(xs[i].grad.numpy(), 4096)
        out["grad_input/" + nm] = s
        out["grad_input/%s/stats" % nm] = np.array([amax, float(xs[i].grad.norm())], np.float64)
    for h in handles:
        h.remove()
    stage_gradients(model, child_in, out)
    path = os.path.join(HERE, case["name"] + ".npz")
    np.savez_compressed(path, **out)
    print("%-20s %5.1fs  %6.1f KB  loss=%.5f  max-prob=%.4f" % (
        case["name"], time.time() - t0, os.path.getsize(path) / 1024, loss.item(), float(probs.max())))


def stage_gradients(model, child_in, out):
    """Stage-wise gradient vectors: every top-level child is run on ITS OWN train-mode input of
    the full forward above with a SEEDED upstream gradient G (paramgen.make_upstream), L = sum_j <G_j, out_j>, and
    the reference's dL/d(input) and dL/d(parameter) of that child alone are recorded (strided samples + norms).  The
    end-to-end gradients of these ReLU / max-pool networks amplify fp32 noise to 1-5 % (fp32 vs fp64 of the
    reference itself); one child at a time does not, so the HIP tape's wiring can be held to a tight tolerance."""
    names = []
    for k, (name, child) in enumerate(model.named_children()):
        if name not in child_in or isinstance(child, torch.nn.MaxPool3d):
            continue
        a = child_in[name]
        ins = [t.clone().requires_grad_(True) for t in a] if isinstance(a, list) else a.clone().requires_grad_(True)
        model.zero_grad(set_to_none=True)
        o = child(list(ins) if isinstance(ins, list) else ins)  # a fresh list: several children overwrite x[pathway]
        outs = list(o) if isinstance(o, (list, tuple)) else [o]
        loss = 0.0
        for j, t in enumerate(outs):
            loss = loss + (t * torch.from_numpy(make_upstream(upstream_seed(k, j), t.shape))).sum()
        loss.backward()
        names.append(name)
        out["stage/%s/index" % name] = np.array([k, len(outs)])
        for j, t in enumerate(outs):
            out["stage/%s/out_shape/%d" % (name, j)] = np.array(t.shape)
        for i, t in enumerate(ins if isinstance(ins, list) else [ins]):
            g = t.grad if t.grad is not None else torch.zeros_like(t)
            s_, amax, _ = sample_activation(g.numpy(), 4096)
            out["stage/%s/gin/%d" % (name, i)] = s_
            out["stage/%s/gin/%d/stats" % (name, i)] = np.array([amax, float(g.norm())], np.float64)
        for pn, p in child.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            s_, amax, _ = sample_activation(g.numpy(), 512)
            out["stage/%s/p/%s" % (name, pn)] = s_
            out["stage/%s/p/%s/stats" % (name, pn)] = np.array([amax, float(g.norm())], np.float64)
    out["stage_children"] = np.array(names)


def op_vectors(get_cfg, build_model):
    """Per-op vectors at production channel widths from the reference's own modules."""
    from slowfast.models.wdf_attention_helper import ECA, SpatialAttention
    from slowfast.models.custom_video_model_builder import FuseFastAndSlow
    from slowfast.models.resnet_helper import ResBlock, BottleneckTransform, BasicTransform
    from slowfast.models.video_model_builder import FuseFastToSlow
    from slowfast.models.stem_helper import ResNetBasicStem

    out = {}
    specs = {}

    def run(name, mod, shapes, seed):
        mod.eval()
        fill_state_dict(mod.state_dict(), seed)
        rs = np.random.RandomState(seed + 1000)
        xs = [rs.standard_normal(s).astype(np.float32) for s in shapes]
        with torch.no_grad():
            y = mod(*[torch.from_numpy(x) for x in xs]) if len(xs) == 1 else mod([torch.from_numpy(x) for x in xs])
        ys = y if isinstance(y, (list, tuple)) else [y]
        for i, a in enumerate(ys):
            out["%s/out%d" % (name, i)] = a.numpy()
        specs[name] = dict(shapes=[list(s) for s in shapes], seed=seed,
                           keys=list(mod.state_dict().keys()),
                           key_shapes=[list(v.shape) for v in mod.state_dict().values()])

    # SpatialAttention at the four R50 head dims (+ the tiny ShuffleNet/Ghost ones), N = 2*28*28 / 4*14*14
    for c, thw in ((8, (2, 28, 28)), (32, (2, 28, 28)), (64, (4, 14, 14)), (128, (2, 14, 14)), (3, (4, 8, 8)),
                   (28, (2, 14, 14))):
        run("attn_c%d" % c, SpatialAttention(c, reduction=1), [(2, c) + thw], 100 + c)
    run("eca_c32", ECA(32), [(2, 32, 4, 14, 14)], 140)
    # production-width convs: 1x3x3 64->64 @56^2, temporal 3x1x1 1152->512 @14^2 (reduced T), stride-2 block
    run("bottleneck_s3", BottleneckTransform(288, 512, 1, 2, 128, 1), [(1, 288, 2, 28, 28)], 150)
    run("resblock_s5_fast", ResBlock(256, 256, 3, 1, BottleneckTransform, 64), [(1, 256, 8, 7, 7)], 151)
    run("resblock_s4_slow", ResBlock(1152, 1024, 3, 2, BottleneckTransform, 256), [(1, 1152, 2, 14, 14)], 152)
    # BasicTransform (resnet_helper.py:25-107) instantiated on its own: ResBlock cannot build it (it passes dilation=),
    # but the class itself is constructible — Tx3x3 -> BN -> ReLU -> 1x3x3 -> BN, stride 2 and stride 1
    run("basic_transform_s2", BasicTransform(64, 128, 3, 2), [(2, 64, 4, 16, 16)], 157)
    run("basic_transform_s1", BasicTransform(32, 32, 1, 1), [(2, 32, 4, 14, 14)], 158)
    run("stem_slow", ResNetBasicStem(3, 64, [1, 7, 7], [1, 2, 2], [0, 3, 3]), [(1, 3, 2, 64, 64)], 153)
    run("stem_fast", ResNetBasicStem(3, 8, [5, 7, 7], [1, 2, 2], [2, 3, 3]), [(1, 3, 8, 64, 64)], 154)
    run("f2s_k7", FuseFastToSlow(32, 2, 7, 4), [(1, 256, 2, 14, 14), (1, 32, 8, 14, 14)], 155)
    run("cmda_256_32", FuseFastAndSlow([256, 32], 4, 8), [(1, 256, 2, 14, 14), (1, 32, 8, 14, 14)], 156)
    out["specs"] = json.dumps(specs)
    path = os.path.join(HERE, "op_vectors.npz")
    np.savez_compressed(path, **out)
    print("op_vectors           %6.1f KB" % (os.path.getsize(path) / 1024))


def init_digests(get_cfg, build_model):
    """SHA-256 of every tensor of the reference's FRESH state_dict under torch.manual_seed(0) (no paramgen fill): pins
    init_weights AND the order in which module construction consumes the RNG (a15 'parameter identity
    under a seed')."""
    import hashlib
    out = {}
    for case in CASES:
        if case["name"] not in ("shufflenetv2_cfg1", "slowfast_r50_s64", "dual_r50_s64", "ghostnet_w2_s64",
                                "mobilenetv2_w1_s64", "shufflenet_g1_s64", "i3d_r50_s64", "shufflenet_w2_g3_s64"):
            continue
        cfg = get_cfg()
        cfg.merge_from_file(ref_yaml(case["yaml"]))
        cfg.merge_from_list(COMMON + ["MODEL.MODEL_NAME", case["model"]] + case["over"])
        torch.manual_seed(0)
        sd = build_model(cfg).state_dict()
        out[case["name"]] = {k: hashlib.sha256(v.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
                             for k, v in sd.items()}
        # what the generator state looks like afterwards: the same number of draws must have been consumed
        out[case["name"]]["__next_rand__"] = "%.9f" % float(torch.rand(1))
    path = os.path.join(HERE, "init_digests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("init_digests         %6.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    torch.set_num_threads(8)
    get_cfg, build_model = import_reference()
    which = sys.argv[1:]
    for c in CASES:
        if not which or c["name"] in which:
            run_case(c, get_cfg, build_model)
    if not which or "ops" in which:
        op_vectors(get_cfg, build_model)
    if not which or "init" in which:
        init_digests(get_cfg, build_model)
