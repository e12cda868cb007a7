import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch, sfhip
dev = torch.device("cuda:0")
for B, thw, c, scale in ((1, (8, 56, 56), 8, 0.6), (1, (8, 56, 56), 8, 3.0), (2, (8, 56, 56), 8, 3.0), (1, (8, 28, 28), 64, 1.0), (1, (8, 56, 56), 32, 1.0)):
    t, h, w = thw
    n = t * h * w
    g = torch.Generator(device="cpu").manual_seed(c)
    qkv = (torch.randn(B, t, h, w, 3 * c, generator=g) * scale).to(dev)
    x = torch.randn(B, t, h, w, c, generator=g).to(dev)
    gamma = torch.tensor([0.7], device=dev)
    a = sfhip.Act(qkv)
    z = sfhip.attention(a.slice(0, c), a.slice(c, c), a.slice(2 * c, c), sfhip.Act(x), gamma)
    q, k, v = [qkv[..., i * c:(i + 1) * c].reshape(B, n, c).double() for i in range(3)]
    ref = torch.empty(B, n, c, dtype=torch.float64, device=dev)
    for b in range(B):
        for i0 in range(0, n, 4096):
            s = q[b, i0:i0 + 4096] @ k[b].T
            ref[b, i0:i0 + 4096] = torch.softmax(s, -1) @ v[b]
    ref = 0.7 * ref + x.reshape(B, n, c).double()
    got = z.buf[..., z.coff:z.coff + c].reshape(B, n, c).double()
    print("B=%d N=%d d=%d scale=%.1f: max rel err %.3e" % (B, n, c, scale, float((got - ref).abs().max() / ref.abs().max())))
# poison the caching allocator: every later torch.empty hands out NaN-filled memory, so a kernel that reads a workspace
# element nobody wrote shows up
junk = [torch.full((1 << 28,), float("nan"), device=dev) for _ in range(8)]
del junk
# eval-mode epilogue: BN affine + ReLU + x alpha nearest upsample along T into a channel slice of a wider tensor
for B, thw, c, alpha in ((1, (8, 56, 56), 8, 4), (3, (8, 56, 56), 8, 4), (1, (8, 28, 28), 64, 4)):
    t, h, w = thw
    n = t * h * w
    g = torch.Generator(device="cpu").manual_seed(c + 1)
    qkv = (torch.randn(B, t, h, w, 3 * c, generator=g) * 0.8).to(dev)
    x = torch.randn(B, t, h, w, c, generator=g).to(dev)
    gamma = torch.tensor([0.7], device=dev)
    sc = (torch.rand(c, generator=g) + 0.5).to(dev)
    bi = torch.randn(c, generator=g).to(dev)
    wide = sfhip.Act(torch.zeros(B, t * alpha, h, w, c + 8, device=dev))
    a = sfhip.Act(qkv)
    sfhip.attention(a.slice(0, c), a.slice(c, c), a.slice(2 * c, c), sfhip.Act(x), gamma, scale=sc, bias=bi, relu=True,
                    alpha=alpha, out=wide.slice(0, c))
    q, k, v = [qkv[..., i * c:(i + 1) * c].reshape(B, n, c).double() for i in range(3)]
    ref = torch.empty(B, n, c, dtype=torch.float64, device=dev)
    for b in range(B):
        for i0 in range(0, n, 4096):
            s = q[b, i0:i0 + 4096] @ k[b].T
            ref[b, i0:i0 + 4096] = torch.softmax(s, -1) @ v[b]
    ref = torch.relu((0.7 * ref + x.reshape(B, n, c).double()) * sc.double() + bi.double()).reshape(B, t, h, w, c)
    ref = ref.repeat_interleave(alpha, dim=1)
    got = wide.buf[..., :c].double()
    print("eval epilogue B=%d N=%d d=%d alpha=%d: max rel err %.3e" % (B, n, c, alpha, float((got - ref).abs().max() / ref.abs().max())))
