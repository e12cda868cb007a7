#!/usr/bin/env python3
"""Who issues the device memcpys of one training step?  torch profiler (CPU + CUDA activities): every runtime memcpy /
memset call of a step with the Python frame that made it.  usage: tools/prof_memcpy.py [workload]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
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
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)


def step():
    flat.zero()
    loss = torch.nn.functional.cross_entropy(model([clips[0], clips[1]]), labels)
    loss.backward()
    flat.all_reduce_mean()
    opt.step()
    flat.rebind()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
names = collections.Counter()
by = collections.Counter()
for ev in prof.events():
    n = ev.name
    if "emcpy" in n or "emset" in n:
        names[n] += 1
        frames = [f for f in (ev.stack or []) if "efficient-slowfast_amd" in f or "bench.py" in f or "tools/" in f]
        by[(n[:40], frames[0] if frames else "<no python frame>")] += 1
for n, c in names.most_common(12):
    print("%5d  %s" % (c, n))
print()
for (n, where), c in by.most_common(25):
    print("%5d  %-40s %s" % (c, n, where))
