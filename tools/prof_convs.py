#!/usr/bin/env python3
"""Per-shape timing of every dense conv launch (forward, dgrad, wgrad) of one training step of a bench workload:
HIP events around each C-ABI call (synchronised per call: a profiling aid, not a benchmark).
usage: tools/prof_convs.py [workload] [batch]"""
import os
import sys

os.environ["SF_OVERLAP_PATHS"] = "0"  # one stream: an event pair around a call then times that call, not a join wait
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "dual"
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
rec = {}
orig = {n: getattr(sfhip, n) for n in ("conv", "conv_dgrad", "conv_wgrad")}


pending = []


def timed(kind, fn, describe):
    def w(*a, **k):
        # events only, no host synchronisation per call (a sync would add the host's launch latency of the NEXT call to
        # every small kernel): the pairs are read after the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        pending.append(((kind,) + describe(*a, **k), e0, e1))
        return r
    return w


def _odim(i, k, s, p, d):
    return (i + 2 * p - d * (k - 1) - 1) // s + 1


def d_conv(x, wp, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), **k):
    cin = k.get("cin") or x.C
    othw = k.get("out_thw") or tuple(_odim(i, kk, s, p, d) for i, kk, s, p, d in
                                     zip((x.T, x.H, x.W), kernel, stride, padding, dilation))
    # rows = OUTPUT positions (the algorithmic MAC count of a strided conv), not input positions
    return (x.N * othw[0] * othw[1] * othw[2], cin, wp.shape[0], tuple(kernel), tuple(stride), (x.T, x.H, x.W), x.N)


def d_dgrad(dz, wtp, x_like, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), **k):
    # rows = positions of dL/dz = the forward conv's OUTPUT positions: the useful MACs of the data gradient
    return (dz.N * dz.T * dz.H * dz.W, dz.C, x_like.C, tuple(kernel), tuple(stride),
            (x_like.T, x_like.H, x_like.W), dz.N)


def d_wgrad(x, dz, cout, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), cin=None, cin_pad=None, **k):
    return (dz.N * dz.T * dz.H * dz.W, cin or x.C, cout, tuple(kernel), tuple(stride), (x.T, x.H, x.W), dz.N)


sfhip.conv = timed("fwd", orig["conv"], d_conv)
sfhip.conv_dgrad = timed("dgrad", orig["conv_dgrad"], d_dgrad)
sfhip.conv_wgrad = timed("wgrad", orig["conv_wgrad"], d_wgrad)
step()
torch.cuda.synchronize()
for key, e0, e1 in pending:
    v = rec.setdefault(key, [0, 0.0])
    v[0] += 1
    v[1] += e0.elapsed_time(e1)
PEAK_TF, PEAK_TB = 157.3, 8.0  # dense f32 MFMA peak, HBM3E spec peak (MI355X_MICROARCH.md)
rows = []
for (kind, m, cin, cout, k, s, thw, nb), (n, ms) in rec.items():
    taps = k[0] * k[1] * k[2]
    flops = 2.0 * m * cin * cout * taps * n  # m = forward-output positions for every kind
    # algorithmic bytes of one launch: the big-side tensor (thw positions: the conv's INPUT for fwd / wgrad, the data
    # gradient's OUTPUT for dgrad) + the m-position tensor + the weights, each moved once
    big = nb * thw[0] * thw[1] * thw[2]
    if kind == "dgrad":  # reads dz [m, cin-of-this-gemm], writes dx [big, cout-of-this-gemm]
        by = 4.0 * (m * cin + big * cout + cin * cout * taps)
    else:                # fwd: reads x [big, cin], writes z [m, cout]; wgrad: reads x and dz, writes dW
        by = 4.0 * (big * cin + m * cout + cin * cout * taps)
    t = ms * 1e-3 / n
    tmin = max(flops / n / (PEAK_TF * 1e12), by / (PEAK_TB * 1e12))
    rows.append((ms, kind, n, m, cin, cout, k, s, thw, flops / (ms * 1e-3) / 1e12, by / t / 1e12, tmin / t))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("%-6s %3s %9s %5s %5s %-9s %-9s %-12s %8s %7s %6s %7s" % ("kind", "n", "rows", "cin", "cout", "kernel", "stride",
                                                              "THW", "ms", "TF/s", "TB/s", "tmin/t"))
cap = int(os.environ.get("PROF_ROWS", "0")) or len(rows)  # default: EVERY row
for ms, kind, n, m, cin, cout, k, s, thw, tf, tb, fr in rows[:cap]:
    print("%-6s %3d %9d %5d %5d %-9s %-9s %-12s %8.3f %7.1f %6.2f %7.2f" % (
        kind, n, m, cin, cout, "x".join(map(str, k)), "x".join(map(str, s)), "x".join(map(str, thw)), ms, tf, tb, fr))
for kind in ("fwd", "dgrad", "wgrad"):
    sel = [r for r in rows if r[1] == kind]
    print("%-6s total %.2f ms, %.1f TF/s aggregate" % (kind, sum(r[0] for r in sel),
                                                        sum(r[9] * r[0] for r in sel) / max(sum(r[0] for r in sel), 1e-9)))
small = [r for r in rows if min(r[4], r[5]) <= 32]
print("rows with Cin or Cout <= 32: %d shapes, %.2f ms" % (len(small), sum(r[0] for r in small)))
print("rows below 0.5 of their own roofline (tmin/t): %.2f ms" % sum(r[0] for r in rows if r[11] < 0.5))
print("all convs: %.2f ms" % tot)
