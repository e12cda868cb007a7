#!/usr/bin/env python3
"""Which host lines issue the device-to-device copies (and other ATen ops) of a training step?  One eager step under
torch.profiler with Python stacks; ATen ops grouped by their innermost repo frame.  usage: tools/find_copies.py [workload]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
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
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)


def step():
    flat.zero()
    out = model([clips[0], clips[1]])
    loss = torch.nn.functional.cross_entropy(out, labels)
    loss.backward()
    flat.all_reduce_mean()
    opt.step()
    flat.rebind()


for _ in range(4):
    step()
torch.cuda.synchronize()
import json  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
path = os.path.join(ROOT, "gpurun_out", "find_copies_trace.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
prof.export_chrome_trace(path)
ev = json.load(open(path))["traceEvents"]
os.remove(path)
want = os.environ.get("FIND", "Memcpy,Memset").split(",")
rt = [e for e in ev if e.get("cat") == "cuda_runtime" and any(w.lower() in e["name"].lower() for w in want)]
py = [e for e in ev if e.get("cat") == "python_function" and "dur" in e]
print("runtime calls matching %s: %d" % (want, len(rt)))
groups = collections.Counter()
for r in rt:
    best = None
    for f in py:
        if f["tid"] == r["tid"] and f["ts"] <= r["ts"] and f["ts"] + f["dur"] >= r["ts"] + r.get("dur", 0):
            nm = f["name"]
            if ("efficient-slowfast_amd" in nm or "bench.py" in nm or "tools/" in nm) and (best is None or f["ts"] >= best["ts"]):
                best = f
    groups[(r["name"], best["name"].split("efficient-slowfast_amd/")[-1] if best else "?")] += 1
for (name, frame), n in sorted(groups.items(), key=lambda kv: -kv[1])[:60]:
    print("%5d  %-22s %s" % (n, name, frame))
