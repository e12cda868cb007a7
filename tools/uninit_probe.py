#!/usr/bin/env python3
"""Does any kernel of a training step READ memory that nothing wrote in that step?  torch's deterministic mode fills
every torch.empty() with NaN (torch.utils.deterministic.fill_uninitialized_memory): with it on, a read of an
uninitialised activation / workspace / packed buffer turns the logits or some gradient into NaN.  In a process that runs
alone such a read is invisible — fresh device memory reads as zeros and the caching allocator hands a tensor the block
that held the same tensor one step earlier — but not when two processes churn one GPU's memory (the round-6 open issue:
tests/test_multigpu_gpu.py, one pass in ~140 differing from its forward on).
usage: python tools/uninit_probe.py [fixture ...]      (tests/golden fixtures: dual_r50_s64, slowfast_r50_s64, ...)"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "efficient-slowfast_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402

torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
from _ddp_worker import build  # noqa: E402
from _util import case_inputs, load_case  # noqa: E402

for name in (sys.argv[1:] or ["dual_r50_s64"]):
    z, meta = load_case(name)
    labels = torch.from_numpy(z["train/labels"]).cuda()
    xs = [x.cuda() for x in case_inputs(meta)]
    model = build(meta, z, 1)
    for it in range(2):  # cold pass (per-weight packs, no gradient arena), warm pass (batched re-pack, arena)
        model.zero_grad(set_to_none=True)
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.0)  # new parameter version: every packed copy is re-made
        out = model([x.clone() for x in xs])
        loss = torch.nn.functional.cross_entropy(out, labels)
        loss.backward()
        torch.cuda.synchronize()
        bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        print("%s pass %d: logits finite %s, loss %.6f, parameters with a non-finite gradient: %d %s" % (
            name, it, bool(torch.isfinite(out).all()), float(loss), len(bad), bad[:8]))
    model.eval()
    with torch.no_grad():
        pe = model([x.clone() for x in xs])
    print("%s eval: probabilities finite %s" % (name, bool(torch.isfinite(pe).all())))
