"""End-to-end GPU parity: the HIP models (built through the drop-in build_model(cfg) surface, parameters
loaded through load_state_dict with the reference's key names) against
  (1) the golden vectors produced by the REFERENCE itself (tests/golden/*.npz), and
  (2) the oracle run on the same seeded inputs,
at every top-level child boundary, the pre-activation logits and the eval output.

Tolerance: 1e-3 max-norm relative (BASELINE.json north_star: "within 1e-3 rel fp32")."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from _util import MODEL_CASES, case_inputs, load_case, rel_err, sample_activation, seeded_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _build(meta, z):
    import sfhip  # noqa: F401
    from slowfast.config.defaults import get_cfg
    from slowfast.models import build_model
    cfg = get_cfg()
    cfg.merge_from_other_cfg(meta["cfg_dump"])
    cfg.NUM_GPUS = 1
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_model(cfg)
    sd = seeded_state_dict(z["sd_keys"], z["sd_shapes"], meta["param_seed"])
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.eval(), sd


def _report(line):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "models_report.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("name", MODEL_CASES)
def test_eval_forward_matches_reference_golden(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sfhip
    from slowfast.models import head_helper
    z, meta = load_case(name)
    model, sd = _build(meta, z)
    acts, tap = {}, {}

    def hook(child):
        def f(m, i, o):
            if isinstance(o, (list, tuple)):
                acts[child] = [sfhip.to_ncthw(a).cpu().numpy() if isinstance(a, sfhip.Act) else a.cpu().numpy()
                               for a in o]
        return f

    for n, m in model.named_children():
        m.register_forward_hook(hook(n))
    head_helper.LOGITS_TAP = lambda t: tap.__setitem__("logits", t.detach().cpu().numpy())
    try:
        with torch.no_grad():
            out = model([x.cuda() for x in case_inputs(meta)])
        torch.cuda.synchronize()
    finally:
        head_helper.LOGITS_TAP = None
    worst = 0.0
    checked = 0
    for child in z["children"]:
        child = str(child)
        if child not in acts:
            continue
        for i, a in enumerate(acts[child]):
            tag = "eval/%s/%d" % (child, i)
            assert tuple(a.shape) == tuple(z[tag + "/shape"]), tag
            s, amax, mean = sample_activation(a)
            e = rel_err(s, z[tag])
            _report("%-22s %-14s p%d  %.3e" % (name, child, i, e))
            worst = max(worst, e)
            assert e < TOL, (tag, e)
            checked += 1
    assert checked >= (5 if meta.get("single") else 16)
    e_log = rel_err(tap["logits"].reshape(meta["batch"], -1), z["eval/logits_full"])
    e_out = rel_err(out.cpu().numpy(), z["eval/out"])
    _report("%-22s logits %.3e  out %.3e  worst-stage %.3e" % (name, e_log, e_out, worst))
    assert e_log < TOL and e_out < TOL


@pytest.mark.parametrize("name", ["dual_r50_s64", "shufflenetv2_cfg1"])
def test_eval_forward_matches_oracle(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import slowfast_oracle as oracle
    z, meta = load_case(name)
    model, sd = _build(meta, z)
    xs = case_inputs(meta)
    with torch.no_grad():
        out = model([x.cuda() for x in xs])
    torch.cuda.synchronize()
    ref = oracle.forward(meta["model"], sd, xs, meta["hparams"])["out"]
    assert rel_err(out.cpu().numpy(), ref.numpy()) < TOL


def test_children_are_independently_callable():
    """Grad-CAM contract (wdf_visualization/gradcam_video.py:92-105): walking model._modules and calling
    each child on the running [slow, fast] NCTHW list reproduces model(x)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    z, meta = load_case("dual_r50_s64")
    model, sd = _build(meta, z)
    xs = [x.cuda() for x in case_inputs(meta)]
    with torch.no_grad():
        ref = model([x.clone() for x in xs])
        x = [t.clone() for t in xs]
        for name, child in model._modules.items():
            if "pool" in name:
                continue  # MaxPool3d k=s=1 children are kept for ordering only
            x = child(x)
            if name != "head":
                assert all(isinstance(t, torch.Tensor) and t.dim() == 5 for t in x), name
    torch.cuda.synchronize()
    assert rel_err(x.cpu().numpy(), ref.cpu().numpy()) < 1e-5


def test_state_dict_roundtrip_and_cache_invalidation():
    """Packed-weight / folded-BN caches follow in-place parameter updates (optimizer steps, load_state_dict)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    z, meta = load_case("slowfast_r50_s64")
    model, sd = _build(meta, z)
    xs = [x.cuda() for x in case_inputs(meta)]
    with torch.no_grad():
        a = model([x.clone() for x in xs]).cpu()
        model.s3.pathway0_res1.branch2.b.weight.mul_(1.5)
        model.s2_fuse.bn.running_var.mul_(2.0)
        b = model([x.clone() for x in xs]).cpu()
        model.load_state_dict(sd)
        c = model([x.clone() for x in xs]).cpu()
    assert (a - b).abs().max() > 1e-6
    assert torch.equal(a, c)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_train_mode_forward_matches_reference_golden(name):
    """model.train() forward (batch-statistics BatchNorm, raw logits — train_net.py:78) vs the reference's
    train-mode logits and stage samples; dropout disabled on both sides."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sfhip
    z, meta = load_case(name)
    model, sd = _build(meta, z)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    acts = {}

    def hook(child):
        def f(m, i, o):
            if isinstance(o, (list, tuple)):
                acts[child] = [sfhip.to_ncthw(a).cpu().numpy() if isinstance(a, sfhip.Act) else a.cpu().numpy()
                               for a in o]
        return f

    for n, m in model.named_children():
        m.register_forward_hook(hook(n))
    keys = set(model.state_dict().keys())
    # SubBatchNorm3d updates split_bn.* in training; its aggregated `bn.*` only changes in aggregate_stats()
    rm_before = {k: v.clone() for k, v in model.state_dict().items() if k.endswith("running_var") and not (
        k.endswith(".bn.running_var") and k.replace(".bn.running_var", ".split_bn.running_var") in keys)}
    with torch.no_grad():
        out = model([x.cuda() for x in case_inputs(meta)])
    torch.cuda.synchronize()
    for k in ("s2", "s3_fuse", "s5"):
        if k in acts and ("train/%s/0" % k) in z:
            for i, a in enumerate(acts[k]):
                s, _, _ = sample_activation(a)
                e = rel_err(s, z["train/%s/%d" % (k, i)])
                _report("%-22s train %-10s p%d %.3e" % (name, k, i, e))
                assert e < TOL, (k, i, e)
    e = rel_err(out.cpu().numpy(), z["train/logits"])
    _report("%-22s train logits %.3e" % (name, e))
    assert e < TOL
    # running statistics were updated in place (momentum 0.1) and the eval caches follow them
    changed = sum(int(not torch.equal(v, model.state_dict()[k])) for k, v in rm_before.items())
    assert changed == len(rm_before)
    assert max(int(v) for k, v in model.state_dict().items() if k.endswith("num_batches_tracked")) == 1
    after = model.state_dict()
    nbuf = 0
    for tag in z.files:  # running statistics after one training forward vs the reference's
        if tag.startswith("train_buffers/"):
            e = rel_err(after[tag[len("train_buffers/"):]].cpu().numpy(), z[tag])
            assert e < 1e-4, (tag, e)  # 0.1 x batch statistics at the end of a train-mode chain
            nbuf += 1
    _report("%-22s train running-stat buffers checked: %d" % (name, nbuf))
    assert nbuf >= 8, "fixture %s carries no train_buffers/*: regenerate it with tests/golden/make_golden.py" % name


@pytest.mark.parametrize("name", MODEL_CASES)
def test_train_step_gradients_match_reference_golden(name):
    """train_net.py:78-96: logits = model(x); loss = CE(logits, labels); loss.backward() — loss, sampled
    parameter gradients and their norms against the reference's autograd (golden 'grad/*')."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    z, meta = load_case(name)
    model, sd = _build(meta, z)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    logits = model([x.cuda() for x in case_inputs(meta)])
    labels = torch.from_numpy(z["train/labels"]).cuda()
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(z["train/loss"][0])) < 1e-3
    params = dict(model.named_parameters())
    keys = [k[5:] for k in z.files if k.startswith("grad/") and not k.endswith("/stats")]
    assert len(keys) >= 4
    # Tolerance: gradients of this ReLU / max-pool network are discontinuous in the activations — one
    # activation whose sign differs by rounding flips a whole column of contributions.  The REFERENCE's own
    # fp32-vs-fp64 gradients differ by 1.6 % (median) to 5 % (worst) in relative L2 on this fixture (measured
    # with the oracle), so the end-to-end check is calibrated to that floor; each backward kernel is held to
    # 2e-4 against autograd in tests/test_backward_ops_gpu.py.
    worst = 0.0
    for k in keys:
        g = params[k].grad
        assert g is not None, k
        s, amax, _ = sample_activation(g.cpu().numpy(), 4096)
        ref = z["grad/" + k].astype(np.float64)
        e = float(np.linalg.norm(s.astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30))
        norm, rnorm = float(g.norm()), float(z["grad/" + k + "/stats"][1])
        _report("%-22s grad %-50s L2rel %.3e  |g| %.4e vs %.4e" % (name, k, e, norm, rnorm))
        worst = max(worst, e)
        # a scalar parameter's L2rel IS the relative error of one number: a single ReLU6/ReLU mask flip at an fp32
        # tie moves it by 3-14 % (measured: one element of head.pathway0 in mobilenetv2_w1_s64 moves gamma's
        # gradient by 3-7 %; re-associating the Fast stem's K sum moves s2_fuse...s2f.gamma of the Sub-BN fixture
        # by 11.6 % with every kernel exact to 2e-4 on the same buffers) — same bound as tests/test_syncbn_gpu.py
        assert e < (0.3 if g.numel() < 16 else 8e-2), (k, e)
        if g.numel() >= 16:
            assert abs(norm - rnorm) < 5e-2 * rnorm + 1e-9, (k, norm, rnorm)
    missing = [k for k, p in params.items() if p.grad is None]
    assert not missing, missing[:5]


@pytest.mark.parametrize("name", ["dual_r50_s64", "slowfast_r50_s64", "shufflenetv2_cfg1", "ghostnet_w2_s64",
                                  "i3d_r50_s64"])
def test_input_gradients_match_reference_golden(name):
    """dL/d(clip) of the training step (the fixtures' 'grad_input/*') through the stems' data gradient:
    clips passed with requires_grad=True receive .grad like any autograd input.  End-to-end tolerance as for the
    parameter gradients (the chaotic 1-5 % floor of these fixtures); the stems' own data gradient is held tightly by
    tests/test_stage_grads_gpu.py (child s1 / s0)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    z, meta = load_case(name)
    model, sd = _build(meta, z)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    xs = [x.cuda().requires_grad_(True) for x in case_inputs(meta)]
    logits = model(xs)
    labels = torch.from_numpy(z["train/labels"]).cuda()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    torch.cuda.synchronize()
    for x, nm in zip(xs, ("fast",) if meta.get("single") else ("slow", "fast")):
        assert x.grad is not None and tuple(x.grad.shape) == tuple(x.shape), nm
        g = x.grad.cpu().numpy()
        s, _, _ = sample_activation(g, 4096)
        ref = z["grad_input/" + nm].astype(np.float64)
        e = float(np.linalg.norm(s.astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30))
        norm, rnorm = float(np.linalg.norm(g.astype(np.float64))), float(z["grad_input/%s/stats" % nm][1])
        _report("%-22s grad_input/%s L2rel %.3e  |g| %.4e vs %.4e" % (name, nm, e, norm, rnorm))
        assert e < 8e-2 and abs(norm - rnorm) < 5e-2 * rnorm, (nm, e, norm, rnorm)


def test_head_dropout_draws_like_nn_dropout():
    """The head's training dropout goes through the ATen kernel nn.Dropout dispatches to (head_helper.py:207-208 in
    the reference): same mask and same generator consumption under a seed, so a seeded training run stays comparable
    with the reference's."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sfhip
    from slowfast.models import head_helper
    x = torch.randn(8, 1, 1, 1, 2304, device="cuda")
    lin = torch.nn.Linear(2304, 400).cuda()
    drop = torch.nn.Dropout(0.5)
    torch.manual_seed(123)
    ref = torch.nn.functional.linear(drop(x), lin.weight, lin.bias)
    after_ref = torch.rand(1, device="cuda")
    torch.manual_seed(123)
    with torch.no_grad():
        got = head_helper._project(sfhip.Act(x.clone()), lin, drop, True).buf
    after_got = torch.rand(1, device="cuda")
    torch.cuda.synchronize()
    assert rel_err(got.cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5
    assert torch.equal(after_ref, after_got)


def test_precise_bn_pass_matches_oracle_batch_statistics():
    """calculate_and_update_precise_bn (train_net.py:277-296 -> fvcore update_bn_stats): after the pass every BN's
    running statistics are the plain average over the iterations of its per-batch (mean, unbiased var); the
    oracle's train-mode forward records those per batch."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import slowfast_oracle as oracle
    from slowfast.utils.precise_bn import get_bn_modules, update_bn_stats
    z, meta = load_case("dual_r50_s64")
    model, sd = _build(meta, z)
    model.train()
    base = case_inputs(meta)
    batches = [[x * s for x in base] for s in (1.0, 0.7, 1.3)]
    assert len(get_bn_modules(model)) == sum(isinstance(m, torch.nn.BatchNorm3d) for m in model.modules())
    mom = model.s1.pathway0_stem.bn.momentum
    update_bn_stats(model, ([x.cuda() for x in b] for b in batches), num_iters=3)
    torch.cuda.synchronize()
    assert model.s1.pathway0_stem.bn.momentum == mom
    sums = {}
    for b in batches:
        rec = {}
        sdr = dict(sd)
        sdr["__bn_batch_stats__"] = rec
        with torch.no_grad():
            oracle.FORWARDS[meta["model"]](sdr, [x.clone() for x in b], meta["hparams"], training=True)
        for k, (m, v) in rec.items():
            a = sums.setdefault(k, [torch.zeros_like(m), torch.zeros_like(v)])
            a[0] += m / 3.0
            a[1] += v / 3.0
    after = model.state_dict()
    assert len(sums) >= 100
    worst = 0.0
    for k, (m, v) in sums.items():
        worst = max(worst, rel_err(after[k + ".running_mean"].cpu().numpy(), m.numpy()),
                    rel_err(after[k + ".running_var"].cpu().numpy(), v.numpy()))
    _report("precise-bn: %d layers, worst running-stat error %.3e" % (len(sums), worst))
    assert worst < 1e-3
    with pytest.raises(AssertionError):
        update_bn_stats(model, iter([[x.cuda() for x in base]]), num_iters=2)  # loader shorter than num_iters


@pytest.mark.parametrize("name", ["dual_r50_s64", "ghostnet_w2_s64", "shufflenetv2_cfg1", "dual_r50_subbn_s64"])
def test_grad_sink_equals_autograd_path(name):
    """engine.set_grad_sink(True) + FlatGradients (the bench/training fast path: backward kernels accumulate into
    the flat .grad buffer directly) produces the same gradients as handing them to autograd, and accumulates over
    two backward passes exactly like autograd does."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slowfast.models import engine
    from slowfast.utils.distributed import FlatGradients
    z, meta = load_case(name)
    model, sd = _build(meta, z)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    xs = [x.cuda() for x in case_inputs(meta)]
    labels = torch.from_numpy(z["train/labels"]).cuda()

    def run(times):
        model.load_state_dict(sd)  # identical running statistics before every pass
        for _ in range(times):
            torch.nn.functional.cross_entropy(model([x.clone() for x in xs]), labels).backward()
            model.load_state_dict(sd)
        torch.cuda.synchronize()

    run(1)
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    model.zero_grad(set_to_none=True)
    flat = FlatGradients(model.parameters())
    engine.set_grad_sink(True)
    try:
        flat.zero()
        run(1)
        one = flat.flat.clone()
        run(1)  # second backward without zeroing: gradients accumulate
        two = flat.flat.clone()
    finally:
        engine.set_grad_sink(False)
    assert float((one - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float((two - 2 * ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    off = 0
    for p in flat.params:  # .grad tensors are still the views of the flat buffer
        assert p.grad.data_ptr() == flat.flat.data_ptr() + 4 * off
        off += p.numel()


@pytest.mark.parametrize("name", ["dual_r50_s64", "ghostnet_w2_s64"])
def test_two_stream_overlap_is_deterministic_and_equal_to_serial(name):
    """engine.run_paths issues the Fast pathway / one CMDA direction on a side stream.  Every kernel is
    deterministic, so any missing fork/join ordering would show up as run-to-run or overlap-vs-serial differences:
    five training steps with the overlap must be bit-identical to each other and to the serial schedule."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slowfast.models import engine
    z, meta = load_case(name)
    model, sd = _build(meta, z)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    xs = [x.cuda() for x in case_inputs(meta)]
    labels = torch.from_numpy(z["train/labels"]).cuda()

    def step():
        model.load_state_dict(sd)
        model.zero_grad(set_to_none=True)
        out = model([x.clone() for x in xs])
        torch.nn.functional.cross_entropy(out, labels).backward()
        torch.cuda.synchronize()
        return torch.cat([out.detach().reshape(-1)] + [p.grad.reshape(-1) for p in model.parameters()]).clone()

    saved, saved_fuse = engine.OVERLAP_PATHS, engine.FUSE_STREAM
    try:
        engine.OVERLAP_PATHS = False
        serial = step()
        engine.OVERLAP_PATHS = True
        engine.FUSE_STREAM = False
        runs = [step() for _ in range(5)]
        engine.FUSE_STREAM = True   # SF_FUSE_STREAM=1: the fusions' attention direction on a stream of its own, the
        runs += [step() for _ in range(5)]  # Fast pathway's backward released early (run_paths(fuse=True))
    finally:
        engine.OVERLAP_PATHS, engine.FUSE_STREAM = saved, saved_fuse
    for r in runs:
        assert torch.equal(r, serial)
    model.eval()
    with torch.no_grad():
        engine.OVERLAP_PATHS = False
        a = model([x.clone() for x in xs])
        engine.OVERLAP_PATHS = saved
        b = model([x.clone() for x in xs])
    assert torch.equal(a, b)


def test_packed_weight_caches_follow_optimizer_steps():
    """Three SGD steps on the HIP path == three SGD steps with every cache dropped before each forward/backward:
    packed (forward, data-gradient, strided-class) weight copies must be rebuilt after each update even when the new
    copy lands at the old one's address."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    z, meta = load_case("slowfast_r50_s64")
    xs = [x.cuda() for x in case_inputs(meta)]
    labels = torch.from_numpy(z["train/labels"]).cuda()

    def train(drop_caches):
        model, sd = _build(meta, z)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        for _ in range(3):
            if drop_caches:
                for p in model.parameters():
                    p.__dict__.pop("_sf_cache", None)
                for m in model.modules():
                    for k in [k for k in m.__dict__ if k.startswith("_sf_")]:
                        m.__dict__.pop(k)
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(model([x.clone() for x in xs]), labels).backward()
            opt.step()
        torch.cuda.synchronize()
        return torch.cat([p.detach().reshape(-1) for p in model.parameters()])

    assert torch.equal(train(False), train(True))
