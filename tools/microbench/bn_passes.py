#!/usr/bin/env python3
"""Achieved HBM bandwidth of the batch-norm passes (stats, affine, backward reduce + apply) on the activation shapes
of the R50 SlowFast step.  usage: tools/microbench/bn_passes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import sfhip  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(8, 8, 56, 56, 256), (8, 8, 56, 56, 64), (8, 8, 28, 28, 512), (8, 8, 14, 14, 1024), (8, 32, 56, 56, 32),
          (8, 32, 56, 56, 8), (8, 32, 28, 28, 64)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


print("%-22s %10s %10s %10s %10s   (GB/s of algorithmic bytes; us)" % ("shape", "stats", "affine+res", "bwd_reduce", "bwd_apply"))
for shp in SHAPES:
    n, t, h, w, c = shp
    z = sfhip.Act(torch.randn(*shp, device=dev))
    res = sfhip.Act(torch.randn(*shp, device=dev))
    dy = sfhip.Act(torch.randn(*shp, device=dev))
    out = sfhip.Act(torch.empty(*shp, device=dev))
    dz = sfhip.Act(torch.empty(*shp, device=dev))
    g, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    nbytes = z.buf.numel() * 4
    mean, invstd, scale, shift = sfhip.bn_train_stats(z, g, b, 1e-5, 0.1, rm, rv)
    t_stats = timed(lambda: sfhip.bn_train_stats(z, g, b, 1e-5, 0.1, rm, rv))
    t_aff = timed(lambda: sfhip.affine(z, scale, shift, res=res, relu=True, out=out))
    lib = sfhip.lib()
    ws = torch.empty((lib.sf_bn_bwd_ws_floats(c),), device=dev)
    dbeta, dgamma = torch.empty(c, device=dev), torch.empty(c, device=dev)
    head = (dy.ptr(), dy.cs, dy.coff, out.ptr(), out.cs, out.coff, z.ptr(), z.cs, z.coff, n, t, h, w, c)
    tail = (1, 1, sfhip._ptr(mean), sfhip._ptr(invstd))
    st = sfhip._stream

    def red():
        sfhip._check(lib.sf_bn_bwd_reduce(*head, *tail, sfhip._ptr(dbeta), sfhip._ptr(dgamma), sfhip._ptr(ws), st()), "r")

    def app():
        sfhip._check(lib.sf_bn_bwd_apply(*head, *tail, sfhip._ptr(g), sfhip._ptr(dbeta), sfhip._ptr(dgamma), dz.ptr(),
                                         dz.cs, dz.coff, res.ptr(), res.cs, res.coff, st()), "a")
    t_red, t_app = timed(red), timed(app)
    row = [(t_stats, 1), (t_aff, 3), (t_red, 3), (t_app, 5)]  # tensors touched per pass
    print("%-22s " % (shp,) + " ".join("%5.0f/%-4.0f" % (k * nbytes / tt / 1e9, tt * 1e6) for tt, k in row))
