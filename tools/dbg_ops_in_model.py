"""Debug: re-check every bn_bwd and 1x1 conv_wgrad call of a full training step against torch fp64 on the same
buffers."""
import sys, io, contextlib
ROOT = '/root/repo'
for p in (ROOT, ROOT + '/efficient-slowfast_amd', ROOT + '/tests', ROOT + '/tests/golden'):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import sfhip
from _util import load_case, case_inputs, seeded_state_dict
from slowfast.config.defaults import get_cfg
from slowfast.models import build_model, engine

name = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2_w1_s64'
z, meta = load_case(name)
cfg = get_cfg(); cfg.merge_from_other_cfg(meta['cfg_dump']); cfg.NUM_GPUS = 1
with contextlib.redirect_stdout(io.StringIO()):
    model = build_model(cfg)
model.load_state_dict(seeded_state_dict(z['sd_keys'], z['sd_shapes'], meta['param_seed']))
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
model.train()
orig_bwd, orig_wgrad = sfhip.bn_bwd, sfhip.conv_wgrad


def v(a):
    return a.buf[..., a.coff:a.coff + a.C].double()


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


cnt = [0]


def bwd(dy, y, zz, mean, invstd, gamma, relu, rep=1, dres=None, dz_out=None, dgamma_out=None):
    cnt[0] += 1
    ref = None
    if rep == 1:
        g = v(dy).clone()
        if relu:
            yy = v(y)
            g = g * ((yy > 0) & ((yy < 6) if relu == 6 else (yy == yy))).double()
        zf = v(zz)
        C = zz.C
        xhat = (zf - mean.double()[:C]) * invstd.double()[:C]
        M = zf.numel() // C
        gm = g.reshape(-1, C)
        xm = xhat.reshape(-1, C)
        db, dg = gm.sum(0), (gm * xm).sum(0)
        ref = (gamma.detach().double()[:C] * invstd.double()[:C]) * (gm - db / M - xm * dg / M)
        gnorm = float(gm.norm())
    r = orig_bwd(dy, y, zz, mean, invstd, gamma, relu, rep=rep, dres=dres, dz_out=dz_out, dgamma_out=dgamma_out)
    if ref is not None:
        out = v(r[0]).reshape(-1, zz.C)
        print("bn_bwd #%d shape %s relu %s dz %.3e dgamma %.3e dbeta %.3e  |dz|/|g| %.3e" % (
            cnt[0], tuple(zz.buf.shape), relu, rel(out, ref), rel(r[1].double(), dg), rel(r[2].double(), db),
            float(ref.norm()) / max(gnorm, 1e-30)))
    return r


def wgrad(x, dz, cout, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), cin=None, cin_pad=None):
    r = orig_wgrad(x, dz, cout, kernel, stride, padding, dilation, cin, cin_pad)
    if tuple(kernel) == (1, 1, 1) and tuple(stride) == (1, 1, 1):
        c = x.C if cin is None else cin
        ref = v(dz).reshape(-1, cout).t() @ x.buf[..., x.coff:x.coff + c].double().reshape(-1, c)
        print("  wgrad %dx%d rows %d  err %.3e" % (cout, c, x.buf.numel() // x.cs, rel(r[:, 0, :c].double(), ref)))
    return r


sfhip.bn_bwd, sfhip.conv_wgrad = bwd, wgrad
xs = case_inputs(meta)
logits = model([x.cuda() for x in xs])
labels = torch.from_numpy(z['train/labels'])
F.cross_entropy(logits, labels.cuda()).backward()
torch.cuda.synchronize()
