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
            if tag + "/shape" not in z.files:  # a child called on a bare tensor (pathway0_pool): recorded without index
                tag = "eval/%s" % child
            assert tuple(a.shape) == tuple(z[tag + "/shape"]), tag
            s, amax, mean = sample_activation(a.numpy())
            assert rel_err(s, z[tag]) < TOL, tag
            checked += 1
    assert checked >= (5 if meta.get("single") else 16)
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
            elif name == "basic_transform_s2":
                ys = [oracle.basic_transform(sd, "m", xs[0], 3, 2, False)]
            elif name == "basic_transform_s1":
                ys = [oracle.basic_transform(sd, "m", xs[0], 1, 1, False)]
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


def test_sync_bn_two_rank_reference_equals_full_batch_oracle():
    """NaiveSyncBatchNorm3d over 2 ranks x 2 clips (reference run with gloo, make_golden_sync.py) is, by
    batchnorm_helper.py:186-218, plain batch-statistics BN over the union batch: the oracle on the 4-clip batch
    must reproduce both ranks' logits, and its gradient of sum_r loss_r the SUM of the ranks' local gradients
    (GroupGather.backward sums the statistic gradients over ranks)."""
    z, meta = load_case("dual_r50_syncbn_s64")
    sd = seeded_state_dict(z["sd_keys"], z["sd_shapes"], meta["param_seed"])
    sdr = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v)
           for k, v in sd.items()}
    acts = oracle.FORWARDS[meta["model"]](sdr, case_inputs(meta), meta["hparams"], training=True)
    logits, labels = acts["out"], torch.from_numpy(z["labels"])
    per = meta["batch"] // meta["world"]
    total = 0.0
    for r in range(meta["world"]):
        sl = slice(r * per, (r + 1) * per)
        assert rel_err(logits[sl].detach().numpy(), z["r%d/logits" % r]) < TOL
        loss = torch.nn.functional.cross_entropy(logits[sl], labels[sl])
        assert abs(loss.item() - float(z["r%d/loss" % r][0])) < 1e-4
        total = total + loss
    total.backward()
    keys = [k[len("r0/grad/"):] for k in z.files if k.startswith("r0/grad/") and not k.endswith("/stats")]
    assert len(keys) >= 6
    for k in keys:
        ref = sum(z["r%d/grad/%s" % (r, k)].astype(np.float64) for r in range(meta["world"]))
        s, _, _ = sample_activation(sdr[k].grad.numpy(), 4096)
        e = float(np.linalg.norm(s - ref) / max(np.linalg.norm(ref), 1e-30))
        assert e < 5e-2, (k, e)  # ReLU / max-pool tie flips between the two runs (see test_models_gpu gradient note)
