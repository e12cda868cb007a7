#!/usr/bin/env python3
"""Where a training step's wall time goes, seen from the caller's stream: HIP events after zero / forward / loss /
backward / optimizer, and around every top-level child's forward (two-stream schedule on, as in bench.py).
usage: tools/prof_phases.py [workload]"""
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
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
marks = []


def ev(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((tag, e))


def pre(mod, a, n):
    ev("fwd>" + n)
    t = engine.tape()
    if t is not None:  # replayed in reverse: this marker fires when the child's backward has been issued
        t.record(lambda: ev("bwd<" + n))


def post(mod, a, o, n):
    ev("fwd<" + n)
    t = engine.tape()
    if t is not None:
        t.record(lambda: ev("bwd>" + n))


for n, m in model.named_children():
    m.register_forward_pre_hook(lambda mod, a, n=n: pre(mod, a, n))
    m.register_forward_hook(lambda mod, a, o, n=n: post(mod, a, o, n))


def step():
    ev("start")
    flat.zero()
    ev("zeroed")
    logits = model([clips[0], clips[1]])
    ev("forward")
    loss = torch.nn.functional.cross_entropy(logits, labels)
    ev("loss")
    loss.backward()
    ev("backward")
    flat.all_reduce_mean()
    opt.step()
    flat.rebind()
    ev("optimizer")


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    acc = {}
    for it in range(5):
        marks.clear()
        step()
        torch.cuda.synchronize()
        for (t0, e0), (t1, e1) in zip(marks[:-1], marks[1:]):
            acc[(t0, t1)] = acc.get((t0, t1), 0.0) + e0.elapsed_time(e1) / 5
        acc[("start", "END")] = acc.get(("start", "END"), 0.0) + marks[0][1].elapsed_time(marks[-1][1]) / 5
for (a, b), v in acc.items():
    print("%-14s -> %-14s %7.2f ms" % (a, b, v))
