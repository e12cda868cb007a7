#!/usr/bin/env python3
"""Which lines of this repo issue torch's own small device ops (fills, zeros, clones, in-place arithmetic, copies) in a
training step — each is a launch on the step's path and matters for the launch-bound configs (cfg #1, cfg #5 at 2
clips).  Counts calls by the nearest caller frame inside the repo over ONE eager step after warm-up.
usage: python tools/count_torch_ops.py [workload]"""
import collections
import os
import sys
import traceback

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402

counts = collections.Counter()
ON = [False]


def caller():
    out = []
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/torch/" not in fr.filename and "count_torch_ops" not in fr.filename:
            out.append("%s:%d" % (os.path.basename(fr.filename), fr.lineno))
            if len(out) == 2:
                break
    return (" <- ".join(out) or "?",)


def wrap(obj, name, pred=None):
    orig = getattr(obj, name)

    def f(*a, **k):
        if ON[0] and (pred is None or pred(*a, **k)):
            counts[(name,) + caller()] += 1
        return orig(*a, **k)

    setattr(obj, name, f)


cuda0 = lambda t, *a, **k: isinstance(t, torch.Tensor) and t.is_cuda  # noqa: E731
for n in ("copy_", "clone", "zero_", "fill_", "add_", "mul_", "sub_", "div_", "addcmul_", "sqrt", "rsqrt", "sum", "mean",
          "add", "mul", "sub", "div", "double", "float", "to", "__add__", "__mul__", "__sub__", "__truediv__",
          "__iadd__", "__imul__", "__radd__", "__rmul__", "__rsub__", "__neg__", "reciprocal", "index_select"):
    if hasattr(torch.Tensor, n):
        wrap(torch.Tensor, n, cuda0)
wrap(torch.Tensor, "contiguous", lambda t, *a, **k: t.is_cuda and not t.is_contiguous())
for n in ("zeros", "zeros_like", "ones", "full", "cat", "stack", "where", "sqrt", "rsqrt", "tensor"):
    wrap(torch, n)
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "shufflenetv2"
dev = torch.device("cuda:0")
cfg, model, batch, desc = bench.build(workload, dev)
clips = bench.synthetic_clips(cfg, batch, dev, 100)
labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), device=dev)
step, flat, opt = bench.make_train_step(model, clips, labels)
for _ in range(4):
    step()
torch.cuda.synchronize()
ON[0] = True
step()
ON[0] = False
torch.cuda.synchronize()
print("torch-level device ops in one step: %d" % sum(counts.values()))
for k, v in counts.most_common(60):
    print("%5d  %-14s %s" % (v, k[0], k[1]))
