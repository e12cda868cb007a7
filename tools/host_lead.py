#!/usr/bin/env python3
"""How far ahead of the GPU is the Python thread during the eager training step?  Every K tape ops of the backward
(and at a few points of the forward) the host notes its clock and records an event on the issuing stream; afterwards
lead = (time the GPU reached the event) - (time the host issued it).  A lead near zero means the GPU had drained its
queue and was waiting for the next launch there (host-bound); tens of milliseconds mean the host is far ahead.
usage: tools/host_lead.py [workload]  (un-profiled: rocprofv3 slows the host)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
from slowfast.models import engine  # noqa: E402
from slowfast.utils.distributed import FlatGradients  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "dual"
K = int(os.environ.get("LEAD_EVERY", "25"))
dev = torch.device("cuda:0")
cfg, model, batch, desc = bench.build(workload, dev)
clips = bench.synthetic_clips(cfg, batch, dev, 100)
labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), device=dev)
model.train()
flat = FlatGradients(model.parameters())
engine.set_grad_sink(True)
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
side = torch.cuda.Stream()
marks = []  # (label, host time, event)
REC = [False]


def mark(label):
    if REC[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        marks.append((label, time.perf_counter(), e))


def backward(self):
    n = len(self.ops)
    for i, (fn, s) in enumerate(reversed(self.ops)):
        if s is None:
            fn()
            if i % K == 0:
                mark("bwd op %4d/%d" % (i, n))
        else:
            with torch.cuda.stream(s):
                fn()
    for wg in self.joins:
        engine._sync_streams(wg, torch.cuda.current_stream(wg.device))
    self.joins = set()
    self.ops = []
    self.gbuf = {}
    mark("bwd end")


engine.Tape.backward = backward


def step():
    flat.zero()
    mark("step start")
    out = model([clips[0], clips[1]])
    mark("fwd issued")
    loss = torch.nn.functional.cross_entropy(out, labels)
    loss.backward()
    flat.all_reduce_mean()
    opt.step()
    flat.rebind()
    mark("step end")


import gc
with torch.cuda.stream(side):
    for _ in range(10):
        step()
torch.cuda.synchronize()
gc.collect()
gc.freeze()
with torch.cuda.stream(side):
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    t0 = time.perf_counter()  # GPU idle here: e0's GPU time ~ t0
    REC[0] = True
    for _ in range(3):
        step()
    REC[0] = False
torch.cuda.synchronize()
print("%-18s %10s %10s %9s" % ("point", "host ms", "gpu ms", "lead ms"))
for label, th, e in marks:
    tg = e0.elapsed_time(e)
    print("%-18s %10.2f %10.2f %9.2f" % (label, (th - t0) * 1e3, tg, tg - (th - t0) * 1e3))
