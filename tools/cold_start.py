#!/usr/bin/env python3
"""Per-step wall time of the first 100 training steps of a process (run it as the FIRST GPU process on a fresh box):
how long does the cold-start penalty last?  Groups of 4 steps, synchronised per group.
usage: python tools/cold_start.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from slowfast.models import engine  # noqa: E402
from slowfast.utils.distributed import FlatGradients  # noqa: E402

dev = torch.device("cuda", 0)
cfg, model, batch, desc = bench.build("dual", dev)
clips = bench.synthetic_clips(cfg, batch, dev, 100)
labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), device=dev)
model.train()
flat = FlatGradients(model.parameters())
engine.set_grad_sink(True)
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
side = torch.cuda.Stream(priority=-1)
import gc  # noqa: E402
if os.environ.get("COLD_GC") == "off":
    gc.collect()
    gc.disable()
elif os.environ.get("COLD_GC") == "freeze":
    gc.collect()
    gc.freeze()
t_start = time.perf_counter()
out = []
with torch.cuda.stream(side):
    for g in range(25):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            flat.zero()
            loss = torch.nn.functional.cross_entropy(model([clips[0], clips[1]]), labels)
            loss.backward()
            flat.all_reduce_mean()
            opt.step()
            flat.rebind()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 4 * 1e3)
print("ms per step, groups of 4 steps:", " ".join("%.1f" % v for v in out))
print("seconds since the first step: %.1f" % (time.perf_counter() - t_start))
gc.enable()
n_obj = len(gc.get_objects())
t0 = time.perf_counter()
gc.collect()
print("tracked objects %d (frozen %d), one full collection %.1f ms; thresholds %s" % (
    n_obj, gc.get_freeze_count(), (time.perf_counter() - t0) * 1e3, gc.get_threshold()))
