import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch
from torch.profiler import ProfilerActivity, profile
import bench
workload = sys.argv[1] if len(sys.argv) > 1 else "shufflenetv2"
dev = torch.device("cuda:0")
cfg, model, batch, desc = bench.build(workload, dev)
clips = bench.synthetic_clips(cfg, batch, dev, 100)
labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), device=dev)
step, flat, opt = bench.make_train_step(model, clips, labels)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA and ("copyBuffer" in e.name or "Memcpy" in e.name or "memcpy" in e.name):
        cnt[("GPU", e.name[:50])] += 1
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and ("hipMemcpy" in e.name or "Memcpy" in e.name):
        st = [s for s in (e.stack or []) if "/torch/" not in s][:2]
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        gp = e.cpu_parent.cpu_parent.name if (e.cpu_parent is not None and e.cpu_parent.cpu_parent is not None) else "-"
        cnt[("CPU", e.name, par, gp, tuple(st))] += 1
for k, v in cnt.most_common(25):
    print(v, k)
