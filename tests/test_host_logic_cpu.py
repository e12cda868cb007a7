"""Host-side drop-in surface on the CPU (no kernels run): registry / factory behaviour (models/build.py:9-44), every
model's state_dict keys, shapes and top-level child order against the lists recorded from the reference, config
handling, and that a CPU forward refuses to run (no fallback)."""
import contextlib
import io
import json
import os

import pytest
import torch

from _util import MODEL_CASES, load_case


def _build_cpu(meta):
    from slowfast.config.defaults import get_cfg
    from slowfast.models import build_model
    cfg = get_cfg()
    cfg.merge_from_other_cfg(meta["cfg_dump"])
    cfg.NUM_GPUS = 0
    with contextlib.redirect_stdout(io.StringIO()) as out:
        model = build_model(cfg)
    return cfg, model, out.getvalue()


@pytest.mark.parametrize("name", MODEL_CASES)
def test_state_dict_and_children_equal_the_references(name):
    z, meta = load_case(name)
    cfg, model, printed = _build_cpu(meta)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["sd_keys"]]
    assert [list(v.shape) for v in sd.values()] == [json.loads(str(s)) for s in z["sd_shapes"]]
    assert [n for n, _ in model.named_children()] == [str(c) for c in z["children"]]
    assert all(v.dtype in (torch.float32, torch.int64) for v in sd.values())
    if "attention_spatial_s2f.gamma" in " ".join(sd.keys()):
        assert "fusion layer dim input:" in printed  # FuseFastAndSlow prints at construction, like the reference


def test_registry_and_factory_contract():
    from slowfast.config.defaults import get_cfg
    from slowfast.models import build_model
    from slowfast.models.build import MODEL_REGISTRY, Registry
    for name in ("SlowFast", "SlowFastDualAttention", "SlowFastShuffleNetV2", "SlowFastGhostNet", "SlowFastShuffleNet",
                 "SlowFastMoibleNetV2", "ResNet"):
        assert name in MODEL_REGISTRY and MODEL_REGISTRY.get(name).__name__ == name
    with pytest.raises(KeyError):
        MODEL_REGISTRY.get("SlowFastMobileNetV2")  # the reference's spelling is the registered one
    reg = Registry("T")

    @reg.register()
    class A(object):
        pass

    with pytest.raises(AssertionError):
        reg.register(A)
    cfg = get_cfg()
    cfg.NUM_GPUS = torch.cuda.device_count() + 1
    with pytest.raises(AssertionError):
        build_model(cfg)


def test_config_yaml_and_overrides(tmp_path):
    from slowfast.config.defaults import get_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for y in sorted(os.listdir(os.path.join(root, "configs"))):
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(root, "configs", y))
        assert cfg.MODEL.MODEL_NAME and cfg.DATA.NUM_FRAMES % cfg.SLOWFAST.ALPHA == 0
    cfg = get_cfg()
    cfg.merge_from_list(["SLOWFAST.ALPHA", 8, "SOLVER.WEIGHT_DECAY", "1e-4", "BN.NORM_TYPE", "sub_batchnorm"])
    assert cfg.SLOWFAST.ALPHA == 8 and abs(cfg.SOLVER.WEIGHT_DECAY - 1e-4) < 1e-12
    with pytest.raises(KeyError):
        cfg.merge_from_list(["SLOWFAST.NOT_A_KEY", 1])
    bad = tmp_path / "bad.yaml"
    bad.write_text("MODEL:\n  NOT_A_KEY: 3\n")
    with pytest.raises(KeyError):
        get_cfg().merge_from_file(str(bad))


def test_forward_on_cpu_refuses_to_run():
    """NUM_GPUS 0 builds the module tree (checkpoint conversion, state_dict work) but there is no CPU arithmetic."""
    import sfhip
    z, meta = load_case("shufflenetv2_cfg1")
    cfg, model, _ = _build_cpu(meta)
    model.eval()
    x = [torch.zeros(1, 3, 4, 32, 32), torch.zeros(1, 3, 32, 32, 32)]
    with pytest.raises(sfhip.SfhipError):
        model(x)


def test_init_weights_semantics():
    """init_weights (utils/weight_init_helper.py:10-43): final BN of every bottleneck zeroed iff ZERO_INIT_FINAL_BN,
    SpatialAttention.gamma = 0, conv biases 0, Linear ~ N(0, FC_INIT_STD)."""
    z, meta = load_case("dual_r50_s64")
    cfg, model, _ = _build_cpu(meta)
    sd = model.state_dict()
    zero_final = bool(cfg.RESNET.ZERO_INIT_FINAL_BN)
    cbn = [k for k in sd if k.endswith("branch2.c_bn.weight")]
    assert len(cbn) == 32 and all(float(sd[k].abs().max()) == (0.0 if zero_final else 1.0) for k in cbn)
    assert all(float(sd[k].abs().max()) == 1.0 for k in sd if k.endswith("a_bn.weight"))
    assert all(float(sd[k]) == 0.0 for k in sd if k.endswith(".gamma"))
    assert float(sd["head.projection.bias"].abs().max()) == 0.0
    assert abs(float(sd["head.projection.weight"].std()) - cfg.MODEL.FC_INIT_STD) < 0.2 * cfg.MODEL.FC_INIT_STD


@pytest.mark.parametrize("name", ["shufflenetv2_cfg1", "slowfast_r50_s64", "dual_r50_s64", "ghostnet_w2_s64",
                                  "mobilenetv2_w1_s64", "shufflenet_g1_s64", "i3d_r50_s64",
                                  "shufflenet_w2_g3_s64"])
def test_fresh_init_is_bit_identical_to_the_reference(name):
    """build_model(cfg) under torch.manual_seed(0) yields the reference's parameters BIT FOR BIT (tests/golden/
    init_digests.json: SHA-256 per tensor of the reference's fresh state_dict, make_golden.py::init_digests): same
    module construction order, same default initialisers, same init_weights walk (utils/weight_init_helper.py:10-43),
    and the generator is left in the same state."""
    import hashlib
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "init_digests.json")) as f:
        ref = json.load(f)[name]
    z, meta = load_case(name)
    torch.manual_seed(0)
    cfg, model, _ = _build_cpu(meta)
    nxt = "%.9f" % float(torch.rand(1))
    sd = model.state_dict()
    assert list(sd.keys()) == [k for k in z["sd_keys"]]
    bad = [k for k, v in sd.items()
           if hashlib.sha256(v.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16] != ref[k]]
    assert not bad, bad[:8]
    assert nxt == ref["__next_rand__"]


def test_cache_keys_follow_fused_optimizer_steps():
    """torch's fused optimizers (torch._fused_sgd_) update parameters without moving their version counters; the
    engine's caches (packed conv weights, folded BN, ...) must still see the update: every optimizer step of the
    process bumps an epoch that is part of every cache key (engine._key) — round 5: eager training steps behind
    torch.optim.SGD(fused=True) ran on frozen packed weights until this was keyed in."""
    import torch
    from slowfast.models import engine
    p = torch.nn.Parameter(torch.randn(16))
    p.grad = torch.randn(16)
    for it, kwargs in enumerate(({"fused": True}, {"foreach": True}, {})):
        slot = "_t_slot%d" % it
        try:
            opt = torch.optim.SGD([p], lr=0.1, momentum=0.9, **kwargs)
        except (TypeError, RuntimeError):
            continue
        made = []
        engine._cached_t(p, slot, engine._key(p), lambda: made.append(1) or p.detach().clone())
        engine._cached_t(p, slot, engine._key(p), lambda: made.append(1) or p.detach().clone())
        assert len(made) == 1                      # unchanged parameter: the cached copy is reused
        before = p.detach().clone()
        opt.step()
        assert not torch.equal(before, p.detach())
        got = engine._cached_t(p, slot, engine._key(p), lambda: made.append(1) or p.detach().clone())
        assert len(made) == 2 and torch.equal(got, p.detach()), kwargs   # re-made from the updated values
    k = engine._key(p)
    engine.parameters_changed()
    assert engine._key(p) != k


def _fake_sysfs(root, gpus, cpu_nodes):
    """a sysfs tree with the KFD topology (CPU nodes first, like a real host), PCI numa_node files and node cpulists"""
    import os
    nodes = os.path.join(root, "class/kfd/kfd/topology/nodes")
    k = 0
    for _ in cpu_nodes:
        os.makedirs(os.path.join(nodes, str(k)))
        with open(os.path.join(nodes, str(k), "properties"), "w") as f:
            f.write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
        k += 1
    for dom, bus, dev, numa in gpus:
        os.makedirs(os.path.join(nodes, str(k)))
        with open(os.path.join(nodes, str(k), "properties"), "w") as f:
            f.write("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain %d\n" % ((bus << 8) | (dev << 3), dom))
        pci = os.path.join(root, "bus/pci/devices", "%04x:%02x:%02x.0" % (dom, bus, dev))
        os.makedirs(pci)
        with open(os.path.join(pci, "numa_node"), "w") as f:
            f.write("%d\n" % numa)
        k += 1
    for node, cpulist in cpu_nodes.items():
        d = os.path.join(root, "devices/system/node/node%d" % node)
        os.makedirs(d)
        with open(os.path.join(d, "cpulist"), "w") as f:
            f.write(cpulist + "\n")


def test_rank_cpu_sets_follow_the_pci_address_not_the_render_minor(tmp_path, monkeypatch):
    """bench.py --gpus N pins rank r next to HIP device r: KFD topology order -> PCI address -> numa_node, composed with
    the *_VISIBLE_DEVICES lists; no hint (even split) where sysfs or the lists cannot be followed."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    ncpu = 8
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(ncpu)))
    for var in ("ROCR_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    sysfs = str(tmp_path / "sys")
    # four GPUs: two on node 1, two on node 0 — in an order no render-minor rule would give
    _fake_sysfs(sysfs, [(0, 0x85, 0, 1), (0, 0x05, 0, 0), (0, 0xc5, 0, 1), (1, 0x45, 0, 0)], {0: "0-3", 1: "4-7"})
    assert bench.gpu_numa_nodes(4, sysfs) == [1, 0, 1, 0]
    sets = bench.rank_cpu_sets(4, sysfs)
    assert [n for n, _ in sets] == [1, 0, 1, 0]
    assert [c for _, c in sets] == [[4, 5], [0, 1], [6, 7], [2, 3]]
    # visibility lists compose: ROCR filters, HIP indexes what is left
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1,2,3")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert bench.gpu_numa_nodes(2, sysfs) == [0, 0]          # physical 3, 1
    assert [c for _, c in bench.rank_cpu_sets(2, sysfs)] == [[0, 1], [2, 3]]
    # a UUID list cannot be followed: no hint, even split of the allowed CPUs
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-abcdef,GPU-123456")
    assert bench.gpu_numa_nodes(2, sysfs) == [None, None]
    assert bench.rank_cpu_sets(2, sysfs) == [(-1, [0, 1, 2, 3]), (-1, [4, 5, 6, 7])]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    # no KFD topology at all (this container): even split
    assert bench.rank_cpu_sets(2, str(tmp_path / "nothing")) == [(-1, [0, 1, 2, 3]), (-1, [4, 5, 6, 7])]
    assert bench._fmt_cpulist([0, 1, 2, 3, 8, 9, 12]) == "0-3,8-9,12"


def test_bench_parity_gate_contract():
    """bench.py exits non-zero (after printing its line) exactly when the seeded state's forward or typical gradient
    leaves north_star's 1e-3; worst single gradients and the trained-state fields are reported, not gated."""
    import math
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    ok = {"fwd_max_rel_err": 2.2e-6, "fwd_logits_max_rel_err": 4.5e-6, "bwd_median_rel_err": 2.6e-4,
          "bwd_max_rel_err": 1.04e-3, "bwd_median_rel_err_after_steps": 1.2e-3, "bwd_median_rel_err_vs_fp64": 2e-4}
    assert bench.parity_gate(ok) == []
    assert bench.parity_gate({}) == []                      # eval-only / --no-cpu-baseline lines carry no parity fields
    for k in ("fwd_max_rel_err", "fwd_logits_max_rel_err", "bwd_median_rel_err"):
        bad = dict(ok, **{k: 1.5e-3})
        assert len(bench.parity_gate(bad)) == 1 and k in bench.parity_gate(bad)[0]
        assert len(bench.parity_gate(dict(ok, **{k: math.nan}))) == 1


def test_clock_sampler_reads_the_hwmon_files(tmp_path):
    """bench.py's `clocks` block: sclk / package power of rank 0's GPU sampled from its hwmon files while the timed steps
    run (the rooflines are quoted at the peak clock; the part holds ~2.2 of 2.4 GHz under the step).  Fake hwmon here;
    no hwmon -> no block, never an error."""
    import time
    import bench
    hw = tmp_path / "class" / "drm" / "card0" / "device" / "hwmon" / "hwmon4"
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text("2200000000\n")
    (hw / "power1_input").write_text("1200000000\n")
    (hw / "power1_cap").write_text("1400000000\n")
    # the PCI-address route fails without a GPU: the only GPU hwmon of the box is taken
    assert bench.gpu_hwmon_dir(0, sysfs=str(tmp_path)) == str(hw)
    s = bench.ClockSampler(str(hw), period=0.005)
    s.start()
    time.sleep(0.08)
    s.stop()
    out = s.summary()
    assert out["sclk_mhz_mean"] == 2200.0 and out["power_w_mean"] == 1200.0 and out["power_cap_w"] == 1400.0
    assert out["samples"] >= 3 and out["peak_sclk_mhz"] == bench.PEAK_SCLK_MHZ
    none = bench.ClockSampler(bench.gpu_hwmon_dir(0, sysfs=str(tmp_path / "nothing")))
    none.start()
    none.stop()
    assert none.summary() is None
