#!/usr/bin/env python3
"""Golden vectors for the input step either side of the hot path, produced by the
REFERENCE's own functions: slowfast/datasets/utils.py tensor_normalize (:298-315), spatial_sampling (:151-203),
pack_pathway_output (:73-112) and the transform.py helpers they call.

Build container only.  `slowfast.datasets` cannot be imported as a package here (its __init__ pulls in cv2 / av /
torchvision), so utils.py and transform.py are loaded by path under a synthetic package with inert stand-ins for
`cv2` and `torchvision.transforms` (neither is touched by the three functions above).  Writes
tests/golden/input_step.npz: the uint8 clips, the numpy RNG seed each case starts from, and the reference's
[slow, fast] float tensors."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
DS = "/root/reference/SlowFast/slowfast/datasets"


def load_reference_input_functions():
    import _refimport  # noqa: F401  (fvcore's own stubs: portalocker, yacs, simplejson)
    _refimport.import_reference()
    for name in ("cv2", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    pkg = types.ModuleType("sfds")
    pkg.__path__ = [DS]
    sys.modules["sfds"] = pkg
    mods = {}
    for name in ("transform", "utils"):
        spec = importlib.util.spec_from_file_location("sfds." + name, os.path.join(DS, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["sfds." + name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["utils"], mods["transform"]


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def cfg_like(alpha, reverse):
    return _Cfg(DATA=_Cfg(REVERSE_INPUT_CHANNEL=reverse),
                MODEL=_Cfg(ARCH="slowfast", SINGLE_PATHWAY_ARCH=["c2d", "i3d", "slow"], MULTI_PATHWAY_ARCH=["slowfast"]),
                SLOWFAST=_Cfg(ALPHA=alpha))


CASES = [
    # name, (T,H,W), spatial_idx, min_scale, max_scale, crop, flip enabled, inverse-uniform, alpha, reverse, seed
    dict(name="train_jitter", thw=(8, 36, 48), spatial_idx=-1, min_scale=20, max_scale=28, crop=16, flip=True,
         inv=False, alpha=4, reverse=False, seed=5),
    dict(name="train_jitter_tall_inv", thw=(16, 50, 34), spatial_idx=-1, min_scale=24, max_scale=40, crop=20,
         flip=True, inv=True, alpha=4, reverse=True, seed=2),
    dict(name="train_jitter_flip", thw=(8, 40, 40), spatial_idx=-1, min_scale=22, max_scale=30, crop=16, flip=True,
         inv=False, alpha=8, reverse=False, seed=0),
    dict(name="test_left", thw=(8, 30, 44), spatial_idx=0, min_scale=24, max_scale=24, crop=24, flip=False,
         inv=False, alpha=4, reverse=False, seed=1),
    dict(name="test_center", thw=(8, 30, 44), spatial_idx=1, min_scale=24, max_scale=24, crop=24, flip=False,
         inv=False, alpha=4, reverse=False, seed=1),
    dict(name="test_bottom_tall", thw=(32, 47, 29), spatial_idx=2, min_scale=24, max_scale=24, crop=24, flip=False,
         inv=False, alpha=4, reverse=False, seed=1),
    dict(name="noresize_upscale", thw=(8, 16, 20), spatial_idx=1, min_scale=32, max_scale=32, crop=32, flip=False,
         inv=False, alpha=2, reverse=False, seed=1),
]
MEAN, STD = [0.45, 0.40, 0.50], [0.225, 0.25, 0.2]


if __name__ == "__main__":
    utils, transform = load_reference_input_functions()
    out = {"cases": json.dumps(CASES), "mean": np.array(MEAN, np.float32), "std": np.array(STD, np.float32)}
    for c in CASES:
        t, h, w = c["thw"]
        clip = np.random.RandomState(100 + c["seed"]).randint(0, 256, (t, h, w, 3)).astype(np.uint8)
        np.random.seed(c["seed"])  # the reference draws (scale, y, x, flip) from numpy's global RNG
        frames = utils.tensor_normalize(torch.from_numpy(clip), MEAN, STD)
        frames = frames.permute(3, 0, 1, 2)
        frames = utils.spatial_sampling(frames, spatial_idx=c["spatial_idx"], min_scale=c["min_scale"],
                                        max_scale=c["max_scale"], crop_size=c["crop"],
                                        random_horizontal_flip=c["flip"], inverse_uniform_sampling=c["inv"])
        slow, fast = utils.pack_pathway_output(cfg_like(c["alpha"], c["reverse"]), frames)
        out[c["name"] + "/clip"] = clip
        out[c["name"] + "/slow"] = slow.contiguous().numpy()
        out[c["name"] + "/fast"] = fast.contiguous().numpy()
        print("%-24s slow %s fast %s" % (c["name"], tuple(slow.shape), tuple(fast.shape)))
    # pack_pathway_output known answers (T=32, alpha=4 -> [0,4,8,13,17,22,26,31])
    idx = {}
    for t, a in ((32, 4), (16, 4), (32, 8), (8, 4), (64, 4), (8, 2), (4, 4)):
        fr = torch.arange(t, dtype=torch.float32).view(1, t, 1, 1)
        idx["%d/%d" % (t, a)] = utils.pack_pathway_output(cfg_like(a, False), fr)[0].flatten().long().tolist()
    out["slow_indices"] = json.dumps(idx)
    path = os.path.join(HERE, "input_step.npz")
    np.savez_compressed(path, **out)
    print("input_step %.1f KB" % (os.path.getsize(path) / 1024), idx["32/4"])
