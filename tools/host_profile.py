#!/usr/bin/env python3
"""Where does the launch thread spend its time in a training step?  cProfile over 6 steps after warm-up (both the
forward in the main thread and the taped backward, which autograd runs in its own thread, are profiled).
usage: python tools/host_profile.py [n_lines]"""
import cProfile
import os
import pstats
import sys
import threading
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


def step():
    flat.zero()
    loss = torch.nn.functional.cross_entropy(model([clips[0], clips[1]]), labels)
    loss.backward()
    flat.all_reduce_mean()
    opt.step()
    flat.rebind()


for _ in range(10):
    step()
torch.cuda.synchronize()
# host time of one step without waiting for the GPU: enqueue only
t0 = time.perf_counter()
for _ in range(6):
    step()
t_enq = (time.perf_counter() - t0) / 6
torch.cuda.synchronize()
print("host enqueue time per step (GPU running behind): %.1f ms" % (t_enq * 1e3))
prof = cProfile.Profile()
threading.setprofile(lambda *a: None)  # autograd's worker thread is not covered by cProfile: report the main thread
prof.enable()
for _ in range(6):
    step()
prof.disable()
torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
