#!/usr/bin/env python3
"""How chaotic is a child's gradient in the S = 64 fixtures?  The oracle's own child-level backward (CPU) evaluated
on its own child input vs the same input perturbed by relative noise eps.  Justifies the tolerances of
tests/test_stage_grads_gpu.py.   usage: python tests/tools/stage_grad_sensitivity.py [fixture] [child]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "golden"), os.path.dirname(os.path.dirname(HERE))]
import torch  # noqa: E402
from _stage import boundaries, oracle_child_grads, predecessor  # noqa: E402
from _util import case_inputs, load_case, seeded_state_dict  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "slow_r18_s64"
child = sys.argv[2] if len(sys.argv) > 2 else "s2"
torch.set_num_threads(8)
z, meta = load_case(name)
sd = seeded_state_dict(z["sd_keys"], z["sd_shapes"], meta["param_seed"])
clips = case_inputs(meta)
children = [str(c) for c in z["children"]]
acts0 = boundaries(meta, sd, clips)
k = int(z["stage/%s/index" % child][0])
_, g0, p0 = oracle_child_grads(meta, sd, clips, children, child, k, acts0)
prev = predecessor(children, child, acts0)
for eps in (1e-7, 1e-6, 1e-5):
    torch.manual_seed(0)
    inp = [t * (1 + eps * torch.randn_like(t)) for t in acts0[prev]]
    _, g1, p1 = oracle_child_grads(meta, sd, clips, children, child, k, acts0, inputs=inp)
    e = max(float((a - b).norm() / b.norm()) for a, b in zip(g1, g0))
    ep = max(float((p1[n] - p0[n]).norm() / p0[n].norm()) for n in p0 if p0[n] is not None and float(p0[n].norm()) > 1e-3)
    print("%s %s: input noise %.0e -> dL/d(input) moves %.2e, parameters up to %.2e (relative L2)" % (name, child, eps, e, ep))
