#!/usr/bin/env python3
"""What is a kernel family worth to the STEP (not to the sum of kernel times)?  The eager training step with one family
of launches skipped (outputs left as allocated: garbage, but every other launch keeps its shape and its place on its
stream), timed like bench.py.  The drop against the full step is the upper bound of what a perfect (zero-time) version of
that family could return under the two-stream schedule.  usage: tools/whatif_skip.py [workload]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402
import bench  # noqa: E402
from slowfast.models import engine  # noqa: E402
from slowfast.utils.distributed import FlatGradients  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "dual"
dev = torch.device("cuda:0")
cfg, model, batch, desc = bench.build(workload, dev)
clips = bench.synthetic_clips(cfg, batch, dev, 100)
labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), device=dev)
model.train()
flat = FlatGradients(model.parameters())
engine.set_grad_sink(True)
opt = torch.optim.SGD(model.parameters(), lr=0.0, momentum=0.0, weight_decay=0.0)  # lr 0: garbage gradients change nothing
side = torch.cuda.Stream()
SKIP = set()
orig = {n: getattr(sfhip, n) for n in ("conv", "conv_dgrad", "conv_wgrad", "bn_bwd", "affine", "attention",
                                       "attention_bwd", "bn_train_stats_merge")}
L = sfhip.lib()


def small(cin, cout):
    return min(cin, cout) <= 32


def conv(x, wp, kernel, *a, **k):
    cin = k.get("cin") or x.C
    cout = wp.shape[0]
    if ("conv_small" in SKIP and small(cin, cout)) or ("conv_fwd_big" in SKIP and not small(cin, cout)):
        saved = L.sf_conv_fwd, L.sf_conv_fwd_ws, L.sf_conv_fwd_stats, L.sf_conv_fwd_bx, L.sf_bx_split
        try:
            L.sf_conv_fwd = L.sf_conv_fwd_ws = L.sf_conv_fwd_bx = L.sf_bx_split = lambda *aa: 0
            # statistics launches still need a parts count: fall back to "no statistics from the epilogue"
            k2 = dict(k)
            st = k2.pop("stats", False)
            y = orig["conv"](x, wp, kernel, *a, **k2)
            return (y, None) if st else y
        finally:
            L.sf_conv_fwd, L.sf_conv_fwd_ws, L.sf_conv_fwd_stats, L.sf_conv_fwd_bx, L.sf_bx_split = saved
    return orig["conv"](x, wp, kernel, *a, **k)


def conv_dgrad(dz, wtp, x_like, kernel, *a, **k):
    if ("conv_small" in SKIP and small(dz.C, x_like.C)) or ("conv_dgrad_big" in SKIP and not small(dz.C, x_like.C)):
        out = k.get("out")
        return out if out is not None else sfhip.new_act(dz, x_like.N, x_like.T, x_like.H, x_like.W, wtp.shape[0])
    return orig["conv_dgrad"](dz, wtp, x_like, kernel, *a, **k)


def conv_wgrad(x, dz, cout, kernel, *a, **k):
    cin = k.get("cin") or x.C
    if ("conv_small" in SKIP and small(cin, cout)) or ("wgrad_big" in SKIP and not small(cin, cout)):
        if k.get("finish_into") is not None:
            return None
        cp = k.get("cin_pad") or (cin + 15) // 16 * 16
        return torch.empty((cout, kernel[0] * kernel[1] * kernel[2], cp), device=x.buf.device)
    return orig["conv_wgrad"](x, dz, cout, kernel, *a, **k)


def bn_bwd(*a, **k):
    if "bn_bwd" in SKIP:
        return None
    return orig["bn_bwd"](*a, **k)


def attention_bwd(q, k, v, dz, o, lse, gamma, dq, dk, dv):
    if "attn_bwd" in SKIP:
        return torch.zeros((o.shape[0], o.shape[1]), device=o.device)
    return orig["attention_bwd"](q, k, v, dz, o, lse, gamma, dq, dk, dv)


sfhip.conv, sfhip.conv_dgrad, sfhip.conv_wgrad, sfhip.bn_bwd = conv, conv_dgrad, conv_wgrad, bn_bwd
sfhip.attention_bwd = attention_bwd


def step():
    flat.zero()
    out = model([clips[0], clips[1]])
    loss = torch.nn.functional.cross_entropy(out, labels)
    loss.backward()
    flat.all_reduce_mean()
    opt.step()
    flat.rebind()


def measure(tag, skip, n=20):
    SKIP.clear()
    SKIP.update(skip)
    with torch.cuda.stream(side):
        for _ in range(6):
            step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(n):
            step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print("%-44s %7.2f ms per step" % (tag, ms), flush=True)
    return ms


with torch.cuda.stream(side):
    for _ in range(20):
        step()
torch.cuda.synchronize()
gc.collect()
gc.freeze()
# every configuration REPS times, interleaved (one after the other inside a repetition): a family's worth is the median
# of its per-repetition drops against that repetition's full step; the spread says what the number can carry
REPS = int(os.environ.get("WHATIF_REPS", "3"))
CONFIGS = (("full step", []),
           ("without convs with Cin or Cout <= 32 (all 3 kinds)", ["conv_small"]),
           ("without the other forward convs", ["conv_fwd_big"]),
           ("without the other data gradients", ["conv_dgrad_big"]),
           ("without the other weight gradients", ["wgrad_big"]),
           ("without the BN backward kernels", ["bn_bwd"]),
           ("without the attention backward kernels", ["attn_bwd"]))
results = {tag: [] for tag, _ in CONFIGS}
for rep in range(REPS):
    print("repetition %d" % rep, flush=True)
    for tag, skip in CONFIGS:
        results[tag].append(measure(tag, skip))
full = results["full step"]
print("\nfull step: median %.2f ms (min %.2f, max %.2f over %d repetitions)" % (
    sorted(full)[len(full) // 2], min(full), max(full), REPS))
for tag, _ in CONFIGS[1:]:
    drops = sorted(f - v for f, v in zip(full, results[tag]))
    print("%-52s worth %.2f ms of step (min %.2f, max %.2f)" % (tag, drops[len(drops) // 2], drops[0], drops[-1]))
