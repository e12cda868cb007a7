"""Input step: oracle vs the reference's golden vectors (CPU), host-side sampling logic vs the
oracle (CPU), and the HIP prologue kernel vs both (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN
from oracle import input_oracle


def _golden():
    z = np.load(os.path.join(GOLDEN, "input_step.npz"))
    return z, json.loads(str(z["cases"])), [float(v) for v in z["mean"]], [float(v) for v in z["std"]]


def test_input_oracle_matches_reference_golden():
    z, cases, mean, std = _golden()
    flips = 0
    for c in cases:
        np.random.seed(c["seed"])
        (slow, fast), p = input_oracle.input_step(z[c["name"] + "/clip"], mean, std, c["spatial_idx"], c["min_scale"],
                                                  c["max_scale"], c["crop"], c["flip"], c["inv"], c["alpha"],
                                                  c["reverse"])
        assert torch.equal(fast, torch.from_numpy(z[c["name"] + "/fast"])), c["name"]
        assert torch.equal(slow, torch.from_numpy(z[c["name"] + "/slow"])), c["name"]
        flips += int(p["flip"])
    assert 0 < flips < len(cases)  # the fixture exercises both flip outcomes
    for key, want in json.loads(str(z["slow_indices"])).items():
        t, a = (int(v) for v in key.split("/"))
        assert input_oracle.slow_indices(t, a).tolist() == want


def test_host_sampling_matches_oracle_and_consumes_rng_identically():
    """slowfast.datasets.utils.sample_spatial_params draws (scale, y, x, flip) from numpy's global RNG exactly as
    the reference's spatial_sampling does, so a seeded loader reproduces the reference's augmentation."""
    from slowfast.datasets import utils as ds
    z, cases, mean, std = _golden()
    for c in cases:
        t, h, w = c["thw"]
        np.random.seed(c["seed"])
        _, want = input_oracle.input_step(z[c["name"] + "/clip"], mean, std, c["spatial_idx"], c["min_scale"],
                                          c["max_scale"], c["crop"], c["flip"], c["inv"], c["alpha"], c["reverse"])
        after_oracle = np.random.uniform()
        np.random.seed(c["seed"])
        got = ds.sample_spatial_params(h, w, spatial_idx=c["spatial_idx"], min_scale=c["min_scale"],
                                       max_scale=c["max_scale"], crop_size=c["crop"],
                                       random_horizontal_flip=c["flip"], inverse_uniform_sampling=c["inv"])
        assert np.random.uniform() == after_oracle, c["name"]
        assert (got.new_h, got.new_w, got.y, got.x, got.flip) == (want["new_h"], want["new_w"], want["y"], want["x"],
                                                                  want["flip"]), c["name"]
    for key, want in json.loads(str(z["slow_indices"])).items():
        t, a = (int(v) for v in key.split("/"))
        assert ds.slow_frame_indices(t, a).tolist() == want
    fr = torch.arange(2 * 8, dtype=torch.float32).view(2, 8, 1, 1)
    slow, fast = ds.pack_pathway_output(ds.PathwayCfg(alpha=4, reverse_input_channel=False), fr)
    assert fast is fr and slow[:, :, 0, 0].tolist() == [[0.0, 7.0], [8.0, 15.0]]


@pytest.mark.gpu
def test_gpu_prologue_matches_reference_golden():
    """uint8 THWC clip on the GPU -> normalise + scale + crop + flip + slow/fast packing in ONE kernel per pathway,
    written straight into the stems' border-padded NDHWC4 layout; compared with the reference's float tensors."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sfhip
    from slowfast.datasets import utils as ds
    z, cases, mean, std = _golden()
    worst = 0.0
    for c in cases:
        t, h, w = c["thw"]
        clip = torch.from_numpy(z[c["name"] + "/clip"]).cuda()
        np.random.seed(c["seed"])
        p = ds.sample_spatial_params(h, w, spatial_idx=c["spatial_idx"], min_scale=c["min_scale"],
                                     max_scale=c["max_scale"], crop_size=c["crop"],
                                     random_horizontal_flip=c["flip"], inverse_uniform_sampling=c["inv"])
        for pad, wp in (((3, 3), None), ((1, 1), c["crop"] + 4)):
            slow, fast = ds.gpu_input_step([clip], [p], mean, std, alpha=c["alpha"],
                                           reverse_input_channel=c["reverse"], pad=pad, wp=wp)
            torch.cuda.synchronize()
            for name, packed in (("slow", slow), ("fast", fast)):
                ref = torch.from_numpy(z[c["name"] + "/" + name])
                got = packed.to_ncthw().cpu()[0]
                e = float((got - ref).abs().max())
                worst = max(worst, e)
                assert e < 2e-6, (c["name"], name, e)
                # borders and the pad channel are zero (the stem reads them as conv padding)
                buf = packed.buf.cpu()
                assert float(buf[..., 3].abs().max()) == 0.0
                assert float(buf[:, :, :pad[0]].abs().max()) == 0.0 and float(buf[:, :, :, :pad[1]].abs().max()) == 0.0
                assert float(buf[:, :, pad[0] + c["crop"]:].abs().max()) == 0.0
                assert float(buf[:, :, :, pad[1] + c["crop"]:].abs().max()) == 0.0
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "ops_report.txt"), "a") as f:
        f.write("%-60s %.3e\n" % ("input prologue vs reference (max abs)", worst))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["dual_r50_s64", "ghostnet_w2_s64", "mobilenetv2_w1_s64"])
def test_models_accept_packed_clips(case):
    """model([slow, fast]) with the prologue's PackedClips == model on the equivalent NCTHW float tensors
    (eval probabilities and train-mode logits), i.e. the layout pass is skipped, not changed."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import contextlib
    import io
    from _util import load_case, seeded_state_dict
    from slowfast.config.defaults import get_cfg
    from slowfast.datasets import utils as ds
    from slowfast.models import build_model
    z, meta = load_case(case)
    cfg = get_cfg()
    cfg.merge_from_other_cfg(meta["cfg_dump"])
    cfg.NUM_GPUS = 1
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_model(cfg)
    model.load_state_dict(seeded_state_dict(z["sd_keys"], z["sd_shapes"], meta["param_seed"]))
    crop, t = meta["size"], meta["t"]
    rs = np.random.RandomState(4)
    clips = [torch.from_numpy(rs.randint(0, 256, (t, crop + 9, crop + 17, 3)).astype(np.uint8)).cuda()
             for _ in range(2)]
    np.random.seed(9)
    params = [ds.sample_spatial_params(crop + 9, crop + 17, -1, crop, crop + 12, crop) for _ in clips]
    ph, pw, wp = ds.stem_input_geometry(model, crop)
    packed = ds.gpu_input_step(clips, params, cfg.DATA.MEAN, cfg.DATA.STD, cfg.SLOWFAST.ALPHA, pad=(ph, pw), wp=wp)
    dense = [p.to_ncthw() for p in packed]
    assert dense[0].shape == (2, 3, t // cfg.SLOWFAST.ALPHA, crop, crop) and dense[1].shape == (2, 3, t, crop, crop)
    model.eval()
    with torch.no_grad():
        a = model(list(packed))
        b = model([d.clone() for d in dense])
    assert torch.equal(a, b)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    la = model(list(packed))
    la.sum().backward()
    ga = model.head.parameters().__next__().grad.clone()
    model.zero_grad()
    lb = model([d.clone() for d in dense])
    lb.sum().backward()
    assert torch.allclose(la, lb, rtol=1e-5, atol=1e-6)
    assert torch.allclose(ga, model.head.parameters().__next__().grad, rtol=1e-4, atol=1e-6)
    wrong = ds.gpu_input_step(clips, params, cfg.DATA.MEAN, cfg.DATA.STD, cfg.SLOWFAST.ALPHA, pad=(ph + 1, pw), wp=wp)
    with pytest.raises(ValueError):
        model(list(wrong))


@pytest.mark.parametrize("method", ["sum", "max"])
def test_test_meter_ensemble_matches_loop_restatement(method):
    from slowfast.utils.meters import TestMeter
    rs = np.random.RandomState(3)
    nv, nc, ncls = 7, 6, 20
    ids = rs.permutation(nv * nc)
    labels_v = rs.randint(0, ncls, nv)
    preds = rs.rand(nv * nc, ncls).astype(np.float32)
    labels = labels_v[ids // nc]
    meter = TestMeter(nv, nc, ncls, overall_iters=6, ensemble_method=method)
    for s in range(0, nv * nc, 7):
        meter.update_stats(torch.from_numpy(preds[s:s + 7]), torch.from_numpy(labels[s:s + 7]),
                           torch.from_numpy(ids[s:s + 7]))
    vp, vl, cnt, top = input_oracle.test_meter_ensemble(preds, labels, ids, nv, nc, method)
    assert np.allclose(meter.video_preds.numpy(), vp, atol=1e-6) and np.array_equal(meter.video_labels.numpy(), vl)
    assert np.array_equal(meter.clip_count.numpy(), cnt)
    stats = meter.finalize_metrics()
    assert stats["complete"] and stats["top1_acc"] == "%.2f" % top[1] and stats["top5_acc"] == "%.2f" % top[5]
