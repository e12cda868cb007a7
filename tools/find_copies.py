#!/usr/bin/env python3
"""Where do the device-to-device memcpys of a training step come from?  torch.profiler with stacks over ONE eager step
after warm-up: every "Memcpy DtoD" / copy-kernel event by its CPU op and nearest caller inside the repo.
usage: python tools/find_copies.py [workload]"""
import collections
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "dual"
dev = torch.device("cuda:0")
cfg, model, batch, desc = bench.build(workload, dev)
clips = bench.synthetic_clips(cfg, batch, dev, 100)
labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), device=dev)
step, flat, opt = bench.make_train_step(model, clips, labels)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
by_op = collections.Counter()
for e in ev:
    name = e.name
    if "emcpy" in name or "copyBuffer" in name or "copy_" in name:
        stack = [s for s in (e.stack or []) if "/torch/" not in s and "find_copies" not in s][:2]
        by_op[(name[:40], str(e.device_type).split(".")[-1], tuple(stack), str(getattr(e, "input_shapes", ""))[:60])] += 1
for k, v in by_op.most_common(40):
    print(v, k)
print("--- runtime calls")
rt = collections.Counter(e.name for e in ev if e.name.startswith("hip"))
for k, v in rt.most_common(12):
    print(v, k)
