#!/usr/bin/env python3
"""Per-shape timing of every depthwise-conv launch (forward, backward = weight + data gradient, data gradient alone) of
one training step of a bench workload: HIP events around each binding call on ONE stream, read after the step.
Algorithmic bytes: forward = input + output once; backward = x + dz read, dx read-modify-write.
usage: tools/prof_dwconvs.py [workload] [batch]"""
import os
import sys

os.environ["SF_OVERLAP_PATHS"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "ghostnet"
dev = torch.device("cuda:0")
cfg, model, batch, desc = bench.build(workload, dev)
batch = int(sys.argv[2]) if len(sys.argv) > 2 else batch
clips = bench.synthetic_clips(cfg, batch, dev, 100)
labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), device=dev)
model.train()


def step():
    model.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(model([clips[0], clips[1]]), labels)
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
pending = []


def timed(kind, fn, describe):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        pending.append(((kind,) + describe(*a, **k), e0, e1))
        return r
    return w


def d_fwd(x, wp, kernel, stride=(1, 1, 1), padding=(0, 0, 0), scale=None, bias=None, relu=False, res=None, out=None,
          cout=None, out_cmul=1):
    To = sfhip._out_dim(x.T, kernel[0], stride[0], padding[0], 1)
    Ho = sfhip._out_dim(x.H, kernel[1], stride[1], padding[1], 1)
    Wo = sfhip._out_dim(x.W, kernel[2], stride[2], padding[2], 1)
    c = x.C
    by = 4.0 * (x.rows * c + x.N * To * Ho * Wo * (cout or c) * (2 if res is not None else 1))
    return (x.N, x.T, x.H, x.W, c, tuple(kernel), tuple(stride), by)


def d_bwd(x, dz, wp, kernel, stride, padding, dx=None):
    by = 4.0 * (x.rows * x.C + dz.rows * x.C * (2 if dx is not None else 1) + (2 * x.rows * x.C if dx is not None else 0))
    return (x.N, x.T, x.H, x.W, x.C, tuple(kernel), tuple(stride), by)


def d_dgrad(x, dz, wp, kernel, stride, padding, dx):
    by = 4.0 * (dz.rows * x.C + 2 * x.rows * x.C)
    return (x.N, x.T, x.H, x.W, x.C, tuple(kernel), tuple(stride), by)


sfhip.dwconv = timed("fwd", sfhip.dwconv, d_fwd)
sfhip.dwconv_bwd = timed("bwd(w+d)", sfhip.dwconv_bwd, d_bwd)
sfhip.dwconv_dgrad = timed("dgrad", sfhip.dwconv_dgrad, d_dgrad)
step()
torch.cuda.synchronize()
rec = {}
for key, e0, e1 in pending:
    v = rec.setdefault(key, [0, 0.0])
    v[0] += 1
    v[1] += e0.elapsed_time(e1)
rows = sorted(((ms, k, n) for k, (n, ms) in rec.items()), reverse=True)
print("%-9s %3s %-16s %5s %-7s %-7s %8s %8s %7s" % ("kind", "n", "N x T x H x W", "C", "kernel", "stride", "ms", "us/call", "GB/s"))
for ms, (kind, n_, t, h, w, c, k, s, by), n in rows:
    print("%-9s %3d %-16s %5d %-7s %-7s %8.3f %8.1f %7.0f" % (
        kind, n, "%dx%dx%dx%d" % (n_, t, h, w), c, "x".join(map(str, k)), "x".join(map(str, s)), ms, ms / n * 1e3,
        by * n / (ms * 1e-3) / 1e9))
for kind in ("fwd", "bwd(w+d)", "dgrad"):
    sel = [r for r in rows if r[1][0] == kind]
    tot = sum(r[0] for r in sel)
    by = sum(r[1][8] * r[2] for r in sel)
    print("%-9s total %.2f ms, %.0f GB/s of algorithmic bytes (%.3f of 8 TB/s)" % (kind, tot, by / max(tot, 1e-9) / 1e6, by / max(tot, 1e-9) / 1e6 / 8000))
