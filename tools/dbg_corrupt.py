"""Debug: detect forward buffers modified between forward and their own backward."""
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
saved = {}
order = []
orig_stats, orig_bwd = sfhip.bn_train_stats, sfhip.bn_bwd
orig_affine = sfhip.affine


def stats(zz, *a, **k):
    r = orig_stats(zz, *a, **k)
    saved[("z", zz.buf.data_ptr())] = (zz.buf.clone(), len(order))
    order.append(tuple(zz.buf.shape))
    return r


def affine(x, *a, **k):
    y = orig_affine(x, *a, **k)
    saved[("y", y.buf.data_ptr())] = (y, y.buf[..., y.coff:y.coff + y.C].clone(), len(order))
    return y


def bwd(dy, y, zz, *a, **k):
    c, idx = saved[("z", zz.buf.data_ptr())]
    d = float((c - zz.buf).abs().max())
    if d != 0.0:
        print("z CORRUPTED before its backward: layer#%d shape %s maxdiff %.3e nbad %d" % (idx, tuple(c.shape), d, int((c != zz.buf).sum())))
    if y is not None and ("y", y.buf.data_ptr()) in saved:
        ya, yc, idx = saved[("y", y.buf.data_ptr())]
        d = float((yc - y.buf[..., ya.coff:ya.coff + ya.C]).abs().max()) if (ya.coff, ya.C) == (y.coff, y.C) else 0.0
        if d != 0.0:
            print("y CORRUPTED before its backward: layer#%d shape %s maxdiff %.3e" % (idx, tuple(yc.shape), d))
    return orig_bwd(dy, y, zz, *a, **k)


sfhip.bn_train_stats, sfhip.bn_bwd, sfhip.affine = stats, bwd, affine
xs = case_inputs(meta)
logits = model([x.cuda() for x in xs])
labels = torch.from_numpy(z['train/labels'])
F.cross_entropy(logits, labels.cuda()).backward()
torch.cuda.synchronize()
print("done; layers", len(order))
