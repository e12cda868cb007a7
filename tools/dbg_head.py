"""Debug: MobileNetV2BasicHead alone, taped backward vs torch fp64 autograd of the same sub-graph."""
import sys
ROOT = '/root/repo'
for p in (ROOT, ROOT + '/efficient-slowfast_amd'):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from slowfast.models import engine, head_helper

torch.manual_seed(0)
cin, cl = [320, 40], [1280, 160]
head = head_helper.MobileNetV2BasicHead(cin, cl, 400, 0.0).cuda().train()
for p in head.parameters():
    if p.dim() == 1:
        p.data.uniform_(0.5, 1.5)
xs = [torch.randn(2, 320, 4, 2, 2).cuda(), torch.randn(2, 40, 16, 2, 2).cuda()]


class M(object):
    training = True

    def _forward_impl(self, x):
        with engine.internal():
            return head(list(x))


params = list(head.parameters())
out = engine.TapedForward.apply(M(), xs[0], xs[1], *params)
lab = torch.tensor([3, 7]).cuda()
F.cross_entropy(out, lab).backward()
mine = {k: p.grad.double().cpu() for k, p in head.named_parameters()}

sd = {k: v.detach().double().cpu().requires_grad_(True) for k, v in head.named_parameters()}
pooled = []
for pw in range(2):
    z = F.conv3d(xs[pw].double().cpu(), sd["pathway%d_conv1x1x1.0.weight" % pw])
    z = F.batch_norm(z, None, None, sd["pathway%d_conv1x1x1.1.weight" % pw], sd["pathway%d_conv1x1x1.1.bias" % pw], True, 0.0, 1e-5)
    pooled.append(F.relu6(z).mean((2, 3, 4)))
lo = F.linear(torch.cat(pooled, 1), sd["classifier.1.weight"], sd["classifier.1.bias"])
F.cross_entropy(lo, lab.cpu()).backward()
print("logits", float((out.double().cpu() - lo).abs().max()))
for k in mine:
    r = sd[k].grad
    print("%-32s %.3e  %.4e %.4e" % (k, float((mine[k] - r).norm() / r.norm().clamp_min(1e-30)), float(mine[k].norm()), float(r.norm())))
