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


print("%-22s %10s %10s %10s %10s %10s %10s   (GB/s of algorithmic bytes; us; m = byte-mask form, as in the step)" % (
    "shape", "stats", "affine+res", "bwd_reduce", "bwd_apply", "reduce(m)", "apply(m)"))
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
    # the form the training step runs: the ReLU decision as one byte per 4 channels (sf_affine_fwd_mask), relu = 3
    mk = {}
    sfhip.affine(z, scale, shift, res=res, relu=True, out=out, mask=mk)
    t_redm = t_appm = float("nan")
    if "bytes" in mk:
        mb = mk["bytes"]
        headm = (dy.ptr(), dy.cs, dy.coff, sfhip._ptr(mb), c // 4, 0, z.ptr(), z.cs, z.coff, n, t, h, w, c)
        tailm = (1, 3, sfhip._ptr(mean), sfhip._ptr(invstd))

        def redm():
            sfhip._check(lib.sf_bn_bwd_reduce(*headm, *tailm, sfhip._ptr(dbeta), sfhip._ptr(dgamma), sfhip._ptr(ws), st()), "r")

        def appm():
            sfhip._check(lib.sf_bn_bwd_apply(*headm, *tailm, sfhip._ptr(g), sfhip._ptr(dbeta), sfhip._ptr(dgamma),
                                             dz.ptr(), dz.cs, dz.coff, res.ptr(), res.cs, res.coff, st()), "a")
        t_redm, t_appm = timed(redm), timed(appm)
    row = [(t_stats, 1), (t_aff, 3), (t_red, 3), (t_app, 5), (t_redm, 2.0625), (t_appm, 4.0625)]  # tensors touched per pass
    print("%-22s " % (shp,) + " ".join("%5.0f/%-4.0f" % (k * nbytes / tt / 1e9, tt * 1e6) for tt, k in row))
