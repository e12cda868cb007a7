"""Child-level ("stage-wise") gradients through the oracle: d<G, child(in)>/d(in) and /d(parameters) of ONE top-level
child, evaluated either on the oracle's own train-mode input of that child or on tensors handed in (the HIP path's
activations).  G is the seeded upstream gradient of tests/golden/paramgen.py::make_upstream — the same one the
reference side used for the 'stage/*' vectors of the model fixtures (make_golden.py::stage_gradients)."""
import torch

from oracle import slowfast_oracle as oracle
from paramgen import make_upstream, upstream_seed


def boundaries(meta, sd, clips):
    """Names of the child boundaries the oracle records for this model (one plain train-mode forward)."""
    with torch.no_grad():
        acts = oracle.FORWARDS[meta["model"]](dict(sd), [x.clone() for x in clips], meta["hparams"], training=True)
    return acts


def predecessor(children, child, acts):
    i = children.index(child)
    for name in reversed(children[:i]):
        if name in acts:
            return name
    return None


def oracle_child_grads(meta, sd, clips, children, child, k, acts0, inputs=None, masks=None):
    """(outs, input grads, {param name: grad}) of top-level child number k (`child`).  inputs: tensors to evaluate the
    child on instead of the oracle's own activations at its input boundary.  masks: a tests/_masks.py::Masks cursor —
    the HIP forward's ReLU masks, replayed by the oracle's activations (same piecewise-linear function on both sides)."""
    prev = predecessor(children, child, acts0)
    sdr = {n: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in n else v)
           for n, v in sd.items()}
    xs = [x.clone() for x in clips]
    if prev is None:
        leaves = [x.requires_grad_(True) for x in xs] if inputs is None else \
            [t.clone().requires_grad_(True) for t in inputs]
        xs = leaves
    else:
        src = acts0[prev] if inputs is None else inputs
        leaves = [t.detach().clone().requires_grad_(True) for t in src]
        sdr["__override__"] = {prev: leaves}
    if child != "head":
        sdr["__stop__"] = child
    prev_hook, prev_pool = oracle.ACT_HOOK, oracle.POOL_HOOK
    if masks is not None:
        oracle.ACT_HOOK, oracle.POOL_HOOK = masks.hook, masks.pool_hook
    try:
        acts = oracle.FORWARDS[meta["model"]](sdr, xs, meta["hparams"], training=True)
    except oracle.StopForward as e:
        acts = e.args[0]
    finally:
        oracle.ACT_HOOK, oracle.POOL_HOOK = prev_hook, prev_pool
    outs = [acts["out"]] if child == "head" else list(acts[child])
    loss = 0.0
    for j, t in enumerate(outs):
        loss = loss + (t * torch.from_numpy(make_upstream(upstream_seed(k, j), t.shape))).sum()
    loss.backward()
    gin = [l.grad if l.grad is not None else torch.zeros_like(l) for l in leaves]
    pg = {n[len(child) + 1:]: v.grad for n, v in sdr.items()
          if n.startswith(child + ".") and isinstance(v, torch.Tensor) and v.requires_grad}
    return outs, gin, pg
