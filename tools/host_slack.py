#!/usr/bin/env python3
"""Is the eager training step ever waiting for Python?  Steady-state step time with an artificial host stall of d ms
at the start of every step (d = 0, 5, 10, 20): if the GPU queue holds more than d ms of work the step time does not
move; where it starts to move is the host's lead over the GPU.  usage: tools/host_slack.py [workload]"""
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
dev = torch.device("cuda:0")
cfg, model, batch, desc = bench.build(workload, dev)
clips = bench.synthetic_clips(cfg, batch, dev, 100)
labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), device=dev)
model.train()
flat = FlatGradients(model.parameters())
engine.set_grad_sink(True)
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
side = torch.cuda.Stream()


def step(mid_delay=0.0):
    flat.zero()
    out = model([clips[0], clips[1]])
    if mid_delay:
        time.sleep(mid_delay)
    loss = torch.nn.functional.cross_entropy(out, labels)
    loss.backward()
    flat.all_reduce_mean()
    opt.step()
    flat.rebind()


with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.synchronize()
for where in ("start", "before backward"):
    for d in (0.0, 5.0, 10.0, 20.0, 40.0):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            for _ in range(8):
                if where == "start":
                    time.sleep(d * 1e-3)
                    step()
                else:
                    step(d * 1e-3)
        torch.cuda.synchronize()
        print("host stall %4.0f ms at %-16s -> %.2f ms per step" % (d, where, (time.perf_counter() - t0) / 8 * 1e3))
