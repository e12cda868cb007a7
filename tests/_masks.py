"""Activation-mask injection for gradient parity (test infrastructure).

A ReLU network's gradient is discontinuous in its activations: a pre-activation within fp32 rounding of zero lands on
different sides of the kink in two correct implementations, and ONE such flip moves that tensor's gradient by
1/sqrt(numel) and everything upstream with it.  Summed over ~100 ReLU layers that is a 1-3 % relative-L2 floor that does
not shrink with tensor size (flips per tensor grow with numel, the share of one flip falls with it) — measured: HIP vs
oracle at 224^2, forward equal to 5e-6, gradients 2.9e-2 median.  To compare the backward ARITHMETIC the oracle is
handed the HIP forward's masks (y = x * mask), so both sides differentiate the same piecewise-linear function.

capture():  context that records, in issue order, the mask of every ReLU / ReLU6 output the HIP engine produces in a
            training forward (engine.bn_train_apply and the bare conv+ReLU of engine.conv_bn_act), and the arg-max
            decisions of every max-pool (engine.maxpool, and CMDA's temporal max inside ECA.run): the window winner by
            ATen's own first-maximum rule on the HIP input — the rule the HIP backward implements.
inject(m):  context that sets oracle.ACT_HOOK / oracle.POOL_HOOK to replay them, matched by NCTHW shape in FIFO order
            (within one shape both sides visit the layers in the same order: blocks of a pathway are sequential, and
            the pathways / fusion directions never share a shape)."""
import collections
import contextlib

import torch


class Masks(object):
    def __init__(self):
        self.by_shape = collections.defaultdict(collections.deque)
        self.pools = collections.defaultdict(collections.deque)   # (shape, kernel, stride, padding) -> flat indices
        self.pool_count = 0
        self.pool_used = 0
        self.count = 0
        self.used = 0
        self.dropped = 0
        self.missed = []

    def add(self, y_ncthw, relu):
        m = y_ncthw > 0
        hi = None
        if relu in (6, "relu6"):  # ReLU6: pass-through where 0 < y < 6, the constant 6 where it saturates
            hi = (y_ncthw >= 6).cpu()
            m &= y_ncthw < 6
        self.by_shape[tuple(m.shape)].append((m.cpu(), hi))
        self.count += 1

    def fork(self):
        """A fresh replay cursor over the same masks (the oracle starts every child evaluation from the first layer)."""
        m = Masks()
        for k, q in self.by_shape.items():
            m.by_shape[k] = collections.deque(q)
        for k, q in self.pools.items():
            m.pools[k] = collections.deque(q)
        m.count, m.pool_count = self.count, self.pool_count
        return m

    def add_pool(self, x_ncthw, kernel, stride, padding):
        import torch.nn.functional as F
        x = x_ncthw.detach().float().cpu()
        _, idx = F.max_pool3d(x, kernel, stride, padding, return_indices=True)
        self.pools[(tuple(x.shape), tuple(kernel), tuple(stride), tuple(padding))].append(idx)
        self.pool_count += 1

    def pool_hook(self, x, kernel, stride, padding):
        """max_pool3d(x) with the window winners of the HIP forward: y = x[argmax_HIP].  A captured index set must
        reproduce the oracle's own pooled values up to rounding (it differs only where two window elements tie within
        it); one that does not belongs to another layer and is dropped."""
        import torch.nn.functional as F
        q = self.pools.get((tuple(x.shape), tuple(kernel), tuple(stride), tuple(padding)))
        own = F.max_pool3d(x.detach(), kernel, stride, padding)
        while q:
            idx = q.popleft()
            if idx.shape == own.shape:
                y = x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
                if float((y.detach() - own).abs().max()) <= 1e-4 * float(own.abs().max().clamp_min(1e-30)):
                    self.pool_used += 1
                    return y
            self.dropped += 1
        self.missed.append(("pool", tuple(x.shape)))
        return None

    def hook(self, kind, x):
        """x * (the next captured mask of this shape).  A captured mask must agree with the oracle's own on all but a
        few elements (they differ only where a pre-activation sits within rounding of the kink): one that does not is
        a mask of another layer (a capture the oracle never asks for in this form, e.g. the two halves of ShuffleNet's
        relu(cat[a, b])) and is left in place; with no agreeing one the oracle keeps its own activation (recorded in
        `missed`)."""
        q = self.by_shape.get(tuple(x.shape))
        own = x.detach() > 0
        if kind == "relu6":
            own &= x.detach() < 6
        # the first captured mask of this shape that agrees: normally the head of the queue; the two sides may visit
        # same-shaped layers of DIFFERENT pathways in another order (cfg #1: the Slow and the Fast pathway of
        # SlowFastShuffleNetV2 at 32^2 meet in shapes like [2, 16, 4, 1, 1]), so a non-matching head is skipped, not
        # dropped
        for i, (m, hi) in enumerate(q or ()):
            if float((m == own).float().mean()) > 0.99:
                del q[i]
                self.used += 1
                # torch.where, not x * mask: the product's backward is WRONG on this torch build (2.10 CPU) when the
                # incoming gradient is the expanded (stride-0) gradient of a mean over size-1 dims — found on
                # ShuffleNetV2's head, [2, 1024, 4, 1, 1]: relu(x) and where(x > 0, x, 0) agree to the bit, x * (x > 0)
                # is off by 150 %
                y = torch.where(m, x, torch.zeros_like(x))
                if hi is not None:
                    y = torch.where(hi, torch.full_like(x, 6.0), y)
                return y
        self.missed.append((kind, tuple(x.shape)))
        return None


@contextlib.contextmanager
def capture():
    import sfhip
    from slowfast.models import engine
    masks = Masks()
    orig_bn, orig_conv = engine.bn_train_apply, sfhip.conv

    def bn_train_apply(bn, z, res=None, relu=False, rep=1, out=None, out_reserve=(0, 0), keep=None, out_cmul=1,
                       conv_stats=None):
        y = orig_bn(bn, z, res=res, relu=relu, rep=rep, out=out, out_reserve=out_reserve, keep=keep,
                    out_cmul=out_cmul, conv_stats=conv_stats)
        if relu and out_cmul == 1:
            full = sfhip.to_ncthw(y)
            masks.add(full[:, :, ::rep] if rep > 1 else full, relu)
        elif relu:  # channel-shuffled store (ShuffleNetV2): channel c of the result sits at out.coff + c * out_cmul
            c = z.C if keep is None else keep
            v = y.buf[..., y.coff:y.coff + (c - 1) * out_cmul + 1:out_cmul].permute(0, 4, 1, 2, 3)
            masks.add(v[:, :, ::rep] if rep > 1 else v, relu)
        return y

    def conv(x, wp, kernel, *a, **k):  # bare conv + ReLU without a BN (engine.conv_bn_act's taped path)
        y = orig_conv(x, wp, kernel, *a, **k)
        if k.get("relu") and k.get("scale") is None and engine.tape() is not None:
            masks.add(sfhip.to_ncthw(y), k["relu"])
        return y

    from slowfast.models import wdf_attention_helper as wah
    orig_pool, orig_eca = engine.maxpool, wah.ECA.run

    def maxpool(x, kernel, stride, padding=(0, 0, 0), out_reserve=(0, 0)):
        masks.add_pool(sfhip.to_ncthw(x), tuple(kernel), tuple(stride), tuple(padding))
        return orig_pool(x, kernel, stride, padding, out_reserve=out_reserve)

    def eca_run(self, x, alpha=1, **k):
        if alpha > 1:  # CMDA's MaxPool3d((alpha, 1, 1)) is fused into the ECA kernels
            masks.add_pool(sfhip.to_ncthw(x), (alpha, 1, 1), (alpha, 1, 1), (0, 0, 0))
        return orig_eca(self, x, alpha=alpha, **k)

    engine.bn_train_apply, sfhip.conv, engine.maxpool, wah.ECA.run = bn_train_apply, conv, maxpool, eca_run
    try:
        yield masks
    finally:
        engine.bn_train_apply, sfhip.conv, engine.maxpool, wah.ECA.run = orig_bn, orig_conv, orig_pool, orig_eca


@contextlib.contextmanager
def inject(masks):
    from oracle import slowfast_oracle as oracle
    prev, oracle.ACT_HOOK = oracle.ACT_HOOK, masks.hook
    prev_pool, oracle.POOL_HOOK = oracle.POOL_HOOK, masks.pool_hook
    try:
        yield masks
    finally:
        oracle.ACT_HOOK = prev
        oracle.POOL_HOOK = prev_pool
