#!/usr/bin/env python3
"""Golden vectors for checkpoint compatibility from the REFERENCE's own functions:
utils/c2_model_loading.py get_name_convert_func, utils/checkpoint.py sub_to_normal_bn / normal_to_sub_bn /
c2_normal_to_sub_bn / inflate_weight / is_checkpoint_epoch / get_path_to_checkpoint, and one checkpoint file
written by its save_checkpoint.  Build container only; writes tests/golden/checkpoint_vectors.npz (+ .pyth)."""
import json
import os
import sys
import tempfile
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

C2_NAMES = [
    "conv1_w", "res_conv1_bn_s", "res_conv1_bn_b", "res_conv1_bn_rm", "res_conv1_bn_riv", "res_conv1_w",
    "res2_0_branch2a_w", "res2_0_branch2a_bn_s", "res2_0_branch2a_bn_b", "res2_0_branch2b_bn_rm",
    "res2_0_branch2c_bn_riv", "res2_0_branch1_w", "res2_0_branch1_bn_s", "res2_0_branch1_bn_rm",
    "res4_22_branch2b_w", "res5_2_branch2c_bn_b", "t_conv1_w", "t_res_conv1_bn_s", "t_res_conv1_bn_riv",
    "t_res_conv1_w", "t_res3_1_branch2b_w", "t_res3_1_branch2b_bn_riv", "t_res2_0_branch1_w",
    "t_res2_0_branch1_bn_b", "t_pool1_subsample_w", "t_pool1_subsample_bn_s", "t_pool1_subsample_bn_rm",
    "t_res2_2_branch2c_bn_subsample_w", "t_res2_2_branch2c_bn_subsample_bn_b",
    "t_res3_3_branch2c_bn_subsample_bn_riv", "t_res4_5_branch2c_bn_subsample_w", "nonlocal_conv3_1_theta_w",
    "nonlocal_conv3_1_theta_b", "nonlocal_conv3_1_phi_w", "nonlocal_conv3_3_g_b", "nonlocal_conv4_5_out_w",
    "nonlocal_conv4_1_bn_s", "nonlocal_conv4_1_bn_b", "nonlocal_conv4_1_bn_rm", "nonlocal_conv4_1_bn_riv",
    "pred_w", "pred_b", "lr", "model_iter", "conv1_w_momentum", "res2_0_branch2a_w_momentum",
]


class _Cfg(dict):
    __getattr__ = dict.__getitem__

    def dump(self):
        return json.dumps(self)


def main():
    import _refimport
    _refimport.import_reference()
    import slowfast.utils.checkpoint as cu
    from slowfast.utils.c2_model_loading import get_name_convert_func
    from slowfast.models.batchnorm_helper import SubBatchNorm3d
    from slowfast.models.stem_helper import ResNetBasicStem
    from functools import partial
    out = {}
    conv = get_name_convert_func()
    out["c2_names"] = json.dumps({k: conv(k) for k in C2_NAMES})

    # --- Sub-BN <-> BN state-dict conversion on a real module pair
    torch.manual_seed(1)
    sub = torch.nn.Sequential(OrderedDict(stem=ResNetBasicStem(3, 8, [1, 7, 7], [1, 2, 2], [0, 3, 3],
                                                              norm_module=partial(SubBatchNorm3d, num_splits=2))))
    for v in sub.state_dict().values():
        if v.dtype == torch.float32:
            v.copy_(torch.randn(v.shape))
    sd_sub = sub.state_dict()
    normal = cu.sub_to_normal_bn(sd_sub)
    back = cu.normal_to_sub_bn(OrderedDict((k, v.clone()) for k, v in normal.items()), sub.state_dict())
    for tag, d in (("sub", sd_sub), ("normal", normal), ("back", back)):
        out["subbn/%s/keys" % tag] = np.array(list(d.keys()))
        for k, v in d.items():
            out["subbn/%s/%s" % (tag, k)] = v.numpy().copy()
    out["c2_sub_keys"] = json.dumps({k: cu.c2_normal_to_sub_bn(k, sub.state_dict()) for k in
                                     ("stem.bn.running_mean", "stem.bn.running_var", "stem.conv.weight",
                                      "stem.bn.weight", "other.bn.running_mean")})

    # --- 2D -> 3D inflation
    g = torch.Generator().manual_seed(2)
    sd2 = OrderedDict(a=torch.randn(4, 3, 3, 3, generator=g), b=torch.randn(4, generator=g),
                      c=torch.randn(6, 4, 1, 1, generator=g), d=torch.randn(5, 5, generator=g))
    sd3 = OrderedDict(a=torch.zeros(4, 3, 5, 3, 3), b=torch.zeros(4), c=torch.zeros(6, 4, 3, 1, 1),
                      d=torch.zeros(7, 5))
    inf = cu.inflate_weight(sd2, sd3)
    for tag, d in (("in2d", sd2), ("in3d", sd3), ("out", inf)):
        out["inflate/%s/keys" % tag] = np.array(list(d.keys()))
        for k, v in d.items():
            out["inflate/%s/%s" % (tag, k)] = v.numpy().copy()

    # --- schedule helpers
    sched = []
    for max_epoch, period, epochs in ((10, 3, range(10)), (7, 1, range(7)), (12, 5, range(12))):
        cfg = _Cfg(SOLVER=_Cfg(MAX_EPOCH=max_epoch), TRAIN=_Cfg(CHECKPOINT_PERIOD=period),
                   MULTIGRID=_Cfg(EVAL_FREQ=3))
        sched.append(dict(max_epoch=max_epoch, period=period, plain=[bool(cu.is_checkpoint_epoch(cfg, e)) for e in epochs],
                          multigrid=[bool(cu.is_checkpoint_epoch(cfg, e, [[0, 0, 4], [1, 1, 9], [2, 2, max_epoch]]))
                                     for e in epochs]))
    out["schedule"] = json.dumps(sched)
    out["paths"] = json.dumps({"ckpt": cu.get_path_to_checkpoint("/job", 7), "dir": cu.get_checkpoint_dir("/job")})

    # --- one checkpoint written by the reference (format pin: keys, sub-BN normalisation, file name)
    tmp = tempfile.mkdtemp()
    opt = torch.optim.SGD(sub.parameters(), lr=0.1, momentum=0.9)
    sub(torch.randn(2, 3, 2, 16, 16)).sum().backward()
    opt.step()
    cfg = _Cfg(NUM_GPUS=1, NUM_SHARDS=1, TAG="golden")
    path = cu.save_checkpoint(tmp, sub, opt, 4, cfg)
    out["saved/name"] = np.array(os.path.basename(path))
    blob = open(path, "rb").read()
    out["saved/bytes"] = np.frombuffer(blob, dtype=np.uint8)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    out["saved/top_keys"] = np.array(sorted(ck.keys()))
    out["saved/model_keys"] = np.array(list(ck["model_state"].keys()))
    np.savez_compressed(os.path.join(HERE, "checkpoint_vectors.npz"), **out)
    print("checkpoint_vectors %.1f KB" % (os.path.getsize(os.path.join(HERE, "checkpoint_vectors.npz")) / 1024),
          os.path.basename(path), sorted(ck.keys()))


if __name__ == "__main__":
    main()
