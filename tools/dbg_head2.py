"""Debug: real s8 outputs + real head params -> head alone, taped backward vs torch fp64 / fp32 autograd."""
import sys, io, contextlib
ROOT = '/root/repo'
for p in (ROOT, ROOT + '/efficient-slowfast_amd', ROOT + '/tests', ROOT + '/tests/golden'):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from _util import load_case, case_inputs, seeded_state_dict
from slowfast.config.defaults import get_cfg
from slowfast.models import build_model, engine

z, meta = load_case('mobilenetv2_w1_s64')
cfg = get_cfg(); cfg.merge_from_other_cfg(meta['cfg_dump']); cfg.NUM_GPUS = 1
with contextlib.redirect_stdout(io.StringIO()):
    model = build_model(cfg)
model.load_state_dict(seeded_state_dict(z['sd_keys'], z['sd_shapes'], meta['param_seed']))
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
model.train()
x = [t.cuda() for t in case_inputs(meta)]
with torch.no_grad():
    for n, m in model.named_children():
        if n != "head":
            x = m(x)
xs = [t.clone() for t in x]
print([tuple(t.shape) for t in xs])
head = model.head


class M(object):
    training = True

    def _forward_impl(self, x):
        with engine.internal():
            return head(list(x))


params = list(head.parameters())
out = engine.TapedForward.apply(M(), xs[0], xs[1], *params)
lab = torch.from_numpy(z['train/labels']).cuda()
F.cross_entropy(out, lab).backward()
mine = {k: p.grad.double().cpu() for k, p in head.named_parameters()}
res = {}
for dt in (torch.float64, torch.float32):
    sd = {k: v.detach().to(dt).cpu().requires_grad_(True) for k, v in head.named_parameters()}
    pooled = []
    for pw in range(2):
        zz = F.conv3d(xs[pw].to(dt).cpu(), sd["pathway%d_conv1x1x1.0.weight" % pw])
        zz = F.batch_norm(zz, None, None, sd["pathway%d_conv1x1x1.1.weight" % pw], sd["pathway%d_conv1x1x1.1.bias" % pw], True, 0.0, 1e-5)
        pooled.append(F.relu6(zz).mean((2, 3, 4)))
    lo = F.linear(torch.cat(pooled, 1), sd["classifier.1.weight"], sd["classifier.1.bias"])
    F.cross_entropy(lo, lab.cpu()).backward()
    res[dt] = {k: v.grad.double() for k, v in sd.items()}
for k in mine:
    r = res[torch.float64][k]
    r32 = res[torch.float32][k]
    print("%-32s mine-vs-64 %.3e   cpu32-vs-64 %.3e" % (k, float((mine[k] - r).norm() / r.norm().clamp_min(1e-30)),
                                                      float((r32 - r).norm() / r.norm().clamp_min(1e-30))))
