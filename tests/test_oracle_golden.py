"""The oracle (oracle/slowfast_oracle.py) against every golden vector generated from the reference."""
import json

import numpy as np
import pytest
import torch

from _util import MODEL_CASES, case_inputs, load_case, rel_err, sample_activation, seeded_state_dict
from oracle import slowfast_oracle as oracle

TOL = 1e-4  # same ATen CPU kernels on both sides; only fp32 re-association noise (mean vs avg-pool etc.) is allowed


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_forward_matches_reference(name):
    z, meta = load_case(name)
    sd = seeded_state_dict(z["sd_keys"], z["sd_shapes"], meta["param_seed"])
    acts = oracle.forward(meta["model"], sd, case_inputs(meta), meta["hparams"], training=False)
    checked = 0
    for child in z["children"]:
        child = str(child)
        if child not in acts:
            continue
        for i, a in enumerate(acts[child]):
            tag = "eval/%s/%d" % (child, i)
            assert tuple(a.shape) == tuple(z[tag + "/shape"]), tag
            s, amax, mean = sample_activation(a.numpy())
            assert rel_err(s, z[tag]) < TOL, tag
            checked += 1
    assert checked >= 16
    assert rel_err(acts["logits"].reshape(meta["batch"], -1).numpy(), z["eval/logits_full"]) < TOL
    assert rel_err(acts["out"].numpy(), z["eval/out"]) < TOL


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_train_mode_logits(name):
    z, meta = load_case(name)
    sd = seeded_state_dict(z["sd_keys"], z["sd_shapes"], meta["param_seed"])
    acts = oracle.forward(meta["model"], sd, case_inputs(meta), meta["hparams"], training=True)
    assert rel_err(acts["out"].numpy(), z["train/logits"]) < 1e-4
    loss = torch.nn.functional.cross_entropy(acts["out"], torch.from_numpy(z["train/labels"]))
    assert abs(loss.item() - float(z["train/loss"][0])) < 1e-4


def test_op_vectors():
    import os
    from _util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "op_vectors.npz"))
    specs = json.loads(str(z["specs"]))
    hp = oracle.default_hparams(alpha=4, beta_inv=8, fusion_kernel=7)

    def build(name):
        sp = specs[name]
        sd = seeded_state_dict(sp["keys"], sp["key_shapes"], sp["seed"])
        rs = np.random.RandomState(sp["seed"] + 1000)
        xs = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)) for s in sp["shapes"]]
        return {"m." + k: v for k, v in sd.items()}, xs

    with torch.no_grad():
        for name in specs:
            sd, xs = build(name)
            if name.startswith("attn_"):
                ys = [oracle.spatial_attention(sd, "m", xs[0])]
            elif name.startswith("eca_"):
                ys = [oracle.eca(sd, "m", xs[0])]
            elif name == "bottleneck_s3":
                ys = [oracle.bottleneck(sd, "m", xs[0], 1, 2, 1, 1, False)]
            elif name == "resblock_s5_fast":
                ys = [oracle.res_block(sd, "m", xs[0], 3, 1, 1, 1, False)]
            elif name == "resblock_s4_slow":
                ys = [oracle.res_block(sd, "m", xs[0], 3, 2, 1, 1, False)]
            elif name == "stem_slow":
                ys = [oracle.resnet_basic_stem(sd, "m", xs[0], 1, False)]
            elif name == "stem_fast":
                ys = [oracle.resnet_basic_stem(sd, "m", xs[0], 5, False)]
            elif name == "f2s_k7":
                ys = oracle.fuse_fast_to_slow(sd, "m", xs, hp, False)
            elif name == "cmda_256_32":
                ys = oracle.fuse_fast_and_slow(sd, "m", xs, hp, False)
            else:
                raise AssertionError("unhandled op vector " + name)
            for i, y in enumerate(ys):
                assert rel_err(y.numpy(), z["%s/out%d" % (name, i)]) < TOL, name


def test_pack_pathway_known_answer():
    # datasets/utils.py:93-104 for T=32, alpha=4: NOT a plain stride-4
    assert oracle.pack_pathway_indices(32, 4) == [0, 4, 8, 13, 17, 22, 26, 31]
    assert oracle.pack_pathway_indices(32, 8) == [0, 10, 20, 31]
