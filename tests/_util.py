"""Shared helpers for the tests: fixture loading and seeded parameter reconstruction."""
import json
import os

import numpy as np
import torch

from paramgen import fill_array, make_clip, sample_activation  # noqa: F401  (tests/golden on sys.path)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_CASES = ["shufflenetv2_cfg1", "slowfast_r50_s64", "dual_r50_s64", "ghostnet_w2_s64", "ghostnet_w2_s112", "mobilenetv2_w1_s64", "shufflenet_g1_s64", "shufflenet_w2_g3_s64", "dual_r50_subbn_s64", "i3d_r50_s64", "slow_r18_s64", "slowfast_nln_s64", "c2d_nln_s64"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def seeded_state_dict(keys, shapes, seed, dtype=torch.float32):
    """state_dict rebuilt from (key, shape) lists with paramgen's deterministic fill;
    untouched entries (num_batches_tracked) are zero."""
    sd = {}
    for k, s in zip(keys, shapes):
        s = tuple(json.loads(s)) if isinstance(s, (str, np.str_)) else tuple(s)
        a = fill_array(str(k), s, seed)
        if a is None:
            sd[str(k)] = torch.zeros(s, dtype=torch.long)
        else:
            sd[str(k)] = torch.from_numpy(a).to(dtype)
    return sd


def case_inputs(meta, dtype=torch.float32):
    slow, fast = make_clip(meta["clip_seed"], meta["batch"], meta["t"], meta["alpha"], meta["size"])
    if meta.get("single"):  # single-pathway ResNet (c2d / i3d / slow): the clip itself
        return [torch.from_numpy(fast).to(dtype)]
    return [torch.from_numpy(slow).to(dtype), torch.from_numpy(fast).to(dtype)]


def rel_err(a, b):
    """max-norm relative error: max|a-b| / max|b| (the parity metric, tolerance 1e-3 per north_star)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
