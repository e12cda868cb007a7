#!/usr/bin/env python3
"""Host-side issue time of one training step (how long Python needs to ENQUEUE it) vs its GPU time, and a cProfile of
the enqueue.  usage: tools/prof_host.py [workload]"""
import cProfile
import os
import pstats
import sys
import time

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
for _ in range(3):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.1f ms, until GPU idle %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
