"""Checkpoint compatibility: the product's utils/checkpoint.py + c2_model_loading.py against
vectors produced by the reference's own functions (tests/golden/make_golden_ckpt.py)."""
import io
import json
import os
import pickle
from collections import OrderedDict

import numpy as np
import torch

from _util import GOLDEN


def _z():
    return np.load(os.path.join(GOLDEN, "checkpoint_vectors.npz"))


def _sd(z, prefix):
    return OrderedDict((str(k), torch.from_numpy(z["%s/%s" % (prefix, k)])) for k in z[prefix + "/keys"])


def test_caffe2_name_conversion_matches_reference():
    from slowfast.utils.c2_model_loading import get_name_convert_func
    conv = get_name_convert_func()
    want = json.loads(str(_z()["c2_names"]))
    assert len(want) >= 40
    for k, v in want.items():
        assert conv(k) == v, (k, conv(k), v)


def test_sub_bn_state_dict_conversions_match_reference():
    from slowfast.utils import checkpoint as cu
    z = _z()
    sub, normal, back = _sd(z, "subbn/sub"), _sd(z, "subbn/normal"), _sd(z, "subbn/back")
    got = cu.sub_to_normal_bn(sub)
    assert list(got.keys()) == list(normal.keys())
    assert all(torch.equal(got[k], normal[k]) for k in normal)
    got_back = cu.normal_to_sub_bn(OrderedDict((k, v.clone()) for k, v in normal.items()), sub)
    assert list(got_back.keys()) == list(back.keys())
    assert all(torch.equal(got_back[k], back[k]) for k in back)
    for k, v in json.loads(str(z["c2_sub_keys"])).items():
        assert cu.c2_normal_to_sub_bn(k, sub) == v, k


def test_inflate_and_schedule_helpers_match_reference():
    from slowfast.config.cfgnode import CfgNode
    from slowfast.utils import checkpoint as cu
    z = _z()
    got = cu.inflate_weight(_sd(z, "inflate/in2d"), _sd(z, "inflate/in3d"))
    want = _sd(z, "inflate/out")
    assert list(got.keys()) == list(want.keys()) and all(torch.equal(got[k], want[k]) for k in want)
    for case in json.loads(str(z["schedule"])):
        cfg = CfgNode({"SOLVER": {"MAX_EPOCH": case["max_epoch"]}, "TRAIN": {"CHECKPOINT_PERIOD": case["period"]},
                       "MULTIGRID": {"EVAL_FREQ": 3}})
        n = len(case["plain"])
        assert [cu.is_checkpoint_epoch(cfg, e) for e in range(n)] == case["plain"]
        sched = [[0, 0, 4], [1, 1, 9], [2, 2, case["max_epoch"]]]
        assert [cu.is_checkpoint_epoch(cfg, e, sched) for e in range(n)] == case["multigrid"]
    paths = json.loads(str(z["paths"]))
    assert cu.get_path_to_checkpoint("/job", 7) == paths["ckpt"] and cu.get_checkpoint_dir("/job") == paths["dir"]


def test_reference_written_checkpoint_loads_and_roundtrips(tmp_path):
    """A .pyth file written by the reference's save_checkpoint (Sub-BN stem, SGD state) loads into the product's
    Sub-BN module; the product's own save is read back identically and has the reference's file name / keys."""
    from functools import partial
    from slowfast.config.cfgnode import CfgNode
    from slowfast.models.batchnorm_helper import SubBatchNorm3d
    from slowfast.models.stem_helper import ResNetBasicStem
    from slowfast.utils import checkpoint as cu
    z = _z()
    job = str(tmp_path)
    os.makedirs(cu.get_checkpoint_dir(job))
    name = str(z["saved/name"])
    with open(os.path.join(cu.get_checkpoint_dir(job), name), "wb") as f:
        f.write(z["saved/bytes"].tobytes())
    assert cu.has_checkpoint(job) and cu.get_last_checkpoint(job).endswith(name)
    model = torch.nn.Sequential(OrderedDict(stem=ResNetBasicStem(3, 8, [1, 7, 7], [1, 2, 2], [0, 3, 3],
                                                                norm_module=partial(SubBatchNorm3d, num_splits=2))))
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    epoch = cu.load_checkpoint(cu.get_last_checkpoint(job), model, data_parallel=False, optimizer=opt)
    assert epoch == 4
    ck = torch.load(io.BytesIO(z["saved/bytes"].tobytes()), map_location="cpu", weights_only=False)
    sd = model.state_dict()
    assert torch.equal(sd["stem.conv.weight"], ck["model_state"]["stem.conv.weight"])
    assert torch.equal(sd["stem.bn.bn.running_mean"], ck["model_state"]["stem.bn.running_mean"])
    assert torch.equal(sd["stem.bn.split_bn.running_var"], torch.cat([ck["model_state"]["stem.bn.running_var"]] * 2))
    assert len(opt.state_dict()["state"]) == len(ck["optimizer_state"]["state"]) > 0
    cfg = CfgNode({"NUM_GPUS": 1, "NUM_SHARDS": 1, "TRAIN": {"AUTO_RESUME": True, "CHECKPOINT_FILE_PATH": ""},
                   "OUTPUT_DIR": job})
    path = cu.save_checkpoint(job, model, opt, 6, cfg)
    assert os.path.basename(path) == "checkpoint_epoch_00007.pyth"
    mine = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(mine.keys()) == [str(k) for k in z["saved/top_keys"]]
    assert list(mine["model_state"].keys()) == [str(k) for k in z["saved/model_keys"]]
    assert cu.load_train_checkpoint(cfg, model, opt) == 7  # auto-resume picks the newest file
    # Caffe2 pickle: names converted, Sub-BN running stats tiled, unknown blobs ignored
    blobs = {"conv1_w": sd["stem.conv.weight"].numpy() * 2.0, "res_conv1_bn_rm": np.arange(8, dtype=np.float32),
             "res_conv1_bn_s": np.ones(8, np.float32) * 3.0, "lr": np.zeros(1, np.float32),
             "pred_w": np.zeros((4, 4), np.float32)}
    c2 = os.path.join(job, "c2.pkl")
    with open(c2, "wb") as f:
        pickle.dump({"blobs": blobs}, f)
    want_w = torch.from_numpy(blobs["conv1_w"].copy())
    holder = torch.nn.Module()
    holder.s1 = torch.nn.Module()
    holder.s1.pathway0_stem = model.stem
    assert cu.load_checkpoint(c2, holder, data_parallel=False, convert_from_caffe2=True) == -1
    sd2 = holder.state_dict()
    assert torch.equal(sd2["s1.pathway0_stem.conv.weight"], want_w)
    assert torch.equal(sd2["s1.pathway0_stem.bn.split_bn.running_mean"], torch.arange(8.0).repeat(2))
    assert torch.equal(sd2["s1.pathway0_stem.bn.weight"], torch.full((8,), 3.0))
