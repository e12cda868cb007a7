import sys, os, io, contextlib
ROOT='/root/repo'
for p in (ROOT, ROOT+'/efficient-slowfast_amd', ROOT+'/tests', ROOT+'/tests/golden'): sys.path.insert(0,p)
import numpy as np, torch
from _util import load_case, case_inputs, seeded_state_dict, rel_err
from oracle import slowfast_oracle as oracle
from slowfast.config.defaults import get_cfg
from slowfast.models import build_model
name = sys.argv[1] if len(sys.argv)>1 else 'slowfast_r50_s64'
z, meta = load_case(name)
cfg = get_cfg(); cfg.merge_from_other_cfg(meta['cfg_dump']); cfg.NUM_GPUS=1
with contextlib.redirect_stdout(io.StringIO()): model = build_model(cfg)
sd = seeded_state_dict(z['sd_keys'], z['sd_shapes'], meta['param_seed'])
model.load_state_dict(sd)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout): m.p = 0.0
model.train()
xs = case_inputs(meta)
logits = model([x.cuda() for x in xs])
labels = torch.from_numpy(z['train/labels'])
loss = torch.nn.functional.cross_entropy(logits, labels.cuda()); loss.backward(); torch.cuda.synchronize()
# oracle autograd
torch.set_num_threads(32)
sdr = {k: (v.clone().requires_grad_(True) if v.dtype==torch.float32 and 'running' not in k else v) for k,v in sd.items()}
acts = oracle.FORWARDS[meta['model']](sdr, [x.clone() for x in xs], meta['hparams'], training=True)
lo = torch.nn.functional.cross_entropy(acts['out'], labels); lo.backward()
print('loss', loss.item(), lo.item())
params = dict(model.named_parameters())
worst=[]
for k,p in params.items():
    g = p.grad.cpu(); r = sdr[k].grad
    if r is None: print('no ref grad', k); continue
    e = float((g-r).norm()/r.norm().clamp_min(1e-30))
    worst.append((e,k, float(g.norm()), float(r.norm())))
worst.sort(reverse=True)
for e,k,a,b in worst[:25]: print('%.3e %-60s %.5e %.5e'%(e,k,a,b))
print('median', worst[len(worst)//2][0])
import collections
agg = collections.OrderedDict()
for k, p in params.items():
    top = ".".join(k.split(".")[:2])
    a = agg.setdefault(top, [0.0, 0.0])
    a[0] += float(p.grad.norm()) ** 2
    a[1] += float(sdr[k].grad.norm()) ** 2 if sdr[k].grad is not None else 0.0
for k, (a, b) in agg.items():
    print("%-40s mine %.4e  ref %.4e" % (k, a ** 0.5, b ** 0.5))
if len(sys.argv) > 2 and sys.argv[2] == "all":
    byk = {k: (e, a, b) for e, k, a, b in worst}
    for k in params:
        if k in byk and not k.endswith(".bias"):
            print("%.3e %-62s %.5e %.5e" % ((byk[k][0], k) + byk[k][1:]))
if len(sys.argv) > 3:
    keys = sys.argv[3].split(",")
    os.makedirs(ROOT + "/gpurun_out", exist_ok=True)
    np.savez(ROOT + "/gpurun_out/dbg_grads.npz", **{k: params[k].grad.cpu().numpy() for k in keys})
