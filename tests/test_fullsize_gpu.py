"""BASELINE.json's full sizes (8x8 R50 at 224^2: T = 32, alpha = 4; the s2_fuse attention at N = 25 088, d = 32), where
the CPU oracle is too slow to be the checker: exact chunked fp64 attention on the GPU (test infrastructure, torch) and
size-independent properties of the whole model — probabilities sum to one, clips are independent in eval mode, eager ==
hipGraph replay, the training step is deterministic and equal to the single-stream schedule."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _report(line):
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "fullsize_report.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _chunked_attention_fp64(q, k, v, dz, gamma, chunk=3136):
    """z = gamma * softmax(q k^T) v + x and its gradients, exact in fp64, queries processed in chunks
    (wdf_attention_helper.py:41-54; one [chunk x N] score block at a time)."""
    n = q.shape[0]
    o = torch.empty_like(q)
    dq = torch.empty_like(q)
    dk = torch.zeros_like(k)
    dv = torch.zeros_like(v)
    dgamma = torch.zeros((), dtype=q.dtype, device=q.device)
    for s in range(0, n, chunk):
        qs, ds = q[s:s + chunk], dz[s:s + chunk] * gamma
        p = torch.softmax(qs @ k.t(), dim=-1)
        os_ = p @ v
        o[s:s + chunk] = os_
        dgamma += (dz[s:s + chunk] * os_).sum()
        dp = ds @ v.t()
        dsc = p * (dp - (dp * p).sum(-1, keepdim=True))
        dq[s:s + chunk] = dsc @ k
        dk += dsc.t() @ qs
        dv += p.t() @ ds
    return o, dq, dk, dv, dgamma


def test_fast_stem_ring_kernels_at_full_size():
    """The Fast pathway's stem at its real size (T = 32, 224^2 -> 8 x 32 x 112 x 112, 3.2 M positions per 8 clips;
    2 clips here): conv_stem_fwd_kernel and conv_wgrad_stem_kernel (LDS rings, persistent one-workgroup-per-CU grid,
    t range cut in parts) against an fp64 GPU restatement by exact identities — the forward as a dense fp64 conv3d of
    the same clip, the weight gradient through linearity: <dW, V> = <dy, conv(x, V)> for random directions V."""
    import sfhip
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    n, t, h, w, cout, kt = 2, 32, 224, 224, 8, 5
    x = torch.randn(n, 3, t, h, w, generator=g)
    wt = torch.randn(cout, 3, kt, 7, 7, generator=g) / np.sqrt(147 * kt)
    xa = sfhip.from_ncthw(x.to(dev), cpad=4, ph=3, pw=3, wp=230)
    view = sfhip.Act(xa.buf.view(n, t, 230, 115, 8))
    w4 = torch.zeros(cout, 4, kt, 7, 7)
    w4[:, :3] = wt
    wp = torch.zeros(cout, kt * 7, 32)
    wp[:, :, :28] = w4.permute(0, 2, 3, 4, 1).reshape(cout, kt * 7, 28)
    z = sfhip.conv(view, wp.to(dev).contiguous(), (kt, 7, 1), (1, 2, 1), (kt // 2, 0, 0), cin=28, out_thw=(t, 112, 112))
    xd, wd = x.to(dev).double(), wt.to(dev).double()
    ref = torch.nn.functional.conv3d(xd, wd, None, (1, 2, 2), (kt // 2, 3, 3))          # fp64, NCTHW
    got = z.buf.permute(0, 4, 1, 2, 3).double()
    e_f = float((got - ref).abs().max() / ref.abs().max())
    dy = torch.randn(n, t, 112, 112, cout, generator=g).to(dev)
    dwp = sfhip.conv_wgrad(view, sfhip.Act(dy), cout, (kt, 7, 1), (1, 2, 1), (kt // 2, 0, 0), cin=28, cin_pad=32)
    again = sfhip.conv_wgrad(view, sfhip.Act(dy), cout, (kt, 7, 1), (1, 2, 1), (kt // 2, 0, 0), cin=28, cin_pad=32)
    torch.cuda.synchronize()
    dw = dwp[:, :, :28].reshape(cout, kt, 7, 7, 4)[..., :3].permute(0, 4, 1, 2, 3).double()
    errs = []
    dyd = dy.permute(0, 4, 1, 2, 3).double()
    for _ in range(3):
        v = torch.randn(wt.shape, generator=g).to(dev).double()
        lhs = float((dw * v).sum())
        rhs = float((dyd * torch.nn.functional.conv3d(xd, v, None, (1, 2, 2), (kt // 2, 3, 3))).sum())
        errs.append(abs(lhs - rhs) / max(abs(rhs), 1e-30))
    _report("fast stem at full size: forward max rel err %.2e, wgrad <dW,V> rel err %.2e" % (e_f, max(errs)))
    assert e_f < 1e-5, e_f
    assert max(errs) < 1e-4, errs
    assert torch.equal(dwp, again), "bit-reproducible"


@pytest.mark.parametrize("c,thw,B", [(32, (8, 56, 56), 1), (32, (8, 56, 56), 3), (8, (8, 56, 56), 1), (8, (8, 56, 56), 3),
                                     (4, (8, 112, 112), 1),
                                     # s3_fuse / s4_fuse of cfg #3 (custom_video_model_builder.py:317-358): d = 64 on the
                                     # bf16-piece kernels (attn_*_bx2), d = 128 on the f32-input two-kernel backward
                                     (64, (8, 28, 28), 1), (64, (8, 28, 28), 3), (128, (8, 14, 14), 1),
                                     (128, (8, 14, 14), 3), (128, (8, 14, 14), 8)],
                         ids=["d32_n25088", "d32_n25088_b3_nw8", "d8_n25088", "d8_n25088_b3", "d4_n100352_cfg5",
                              "d64_n6272", "d64_n6272_b3", "d128_n1568", "d128_n1568_b3", "d128_n1568_b8"])
def test_attention_at_production_size_against_exact_fp64(c, thw, B):
    """B = 3 at N = 25 088 is the smallest batch at which the d = 32 backward takes the 8-wavefront / 256-key form
    (attn_bwd_bx_kernel<0, 8>) that bench.py times at 8 clips; the sweep-part counts are batch-keyed as well."""
    import sfhip
    dev = _dev()
    t, h, w = thw
    n = t * h * w
    L = sfhip.lib()
    if c == 32 and L.sf_attn_products_per_fp32(32) == 6:
        assert L.sf_attn_bwd_variant(B, n, c) == (38 if B >= 3 else 34)
        assert L.sf_attn_bwd_variant(8, n, c) == 38  # what the benchmark launches
    g = torch.Generator(device="cpu").manual_seed(c)
    q = (torch.randn(B, n, c, generator=g) * 0.6).to(dev)
    k = (torch.randn(B, n, c, generator=g) * 0.6).to(dev)
    v = torch.randn(B, n, c, generator=g).to(dev)
    x = torch.randn(B, n, c, generator=g).to(dev)
    dz = torch.randn(B, n, c, generator=g).to(dev)
    gamma = torch.tensor([0.7], device=dev)
    refs = [_chunked_attention_fp64(q[b].double(), k[b].double(), v[b].double(), dz[b].double(), 0.7) for b in range(B)]
    o_r, dq_r, dk_r, dv_r = [torch.stack([r[i] for r in refs]) for i in range(4)]
    dg_r = sum(r[4] for r in refs)
    qkv = sfhip.Act(torch.cat([q, k, v], -1).view(B, t, h, w, 3 * c).contiguous())
    save = {}
    out = sfhip.attention(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c), sfhip.Act(x.view(B, t, h, w, c)), gamma,
                          save=save)
    grads = {}
    for mode, fused in (("fused", True), ("split", False)):
        saved = sfhip.FUSED_ATTN_BWD
        sfhip.FUSED_ATTN_BWD = fused
        try:
            d = sfhip.Act(torch.zeros(B, t, h, w, 3 * c, device=dev))
            dvec = sfhip.attention_bwd(qkv.slice(0, c), qkv.slice(c, c), qkv.slice(2 * c, c),
                                       sfhip.Act(dz.view(B, t, h, w, c)), save["o"], save["lse"], gamma,
                                       d.slice(0, c), d.slice(c, c), d.slice(2 * c, c))
        finally:
            sfhip.FUSED_ATTN_BWD = saved
        grads[mode] = (d.buf.view(B, n, 3 * c).double(), dvec.double().sum())
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())

    e_out = rel(out.buf.view(B, n, c).double(), 0.7 * o_r + x.double())
    assert e_out < 1e-5, e_out
    for mode, (d, dgam) in grads.items():
        errs = [rel(d[..., :c], dq_r), rel(d[..., c:2 * c], dk_r), rel(d[..., 2 * c:], dv_r), float(abs(dgam - dg_r) / abs(dg_r))]
        _report("attention B=%d N=%d d=%d %s backward: out %.2e  dq %.2e dk %.2e dv %.2e dgamma %.2e" % ((B, n, c, mode, e_out) + tuple(errs)))
        assert max(errs) < 2e-5, (mode, errs)
    # the single-sweep and the two-kernel backward are different summation orders of the same 5 products
    assert rel(grads["fused"][0], grads["split"][0]) < 1e-5


def _model(workload):
    import sys
    sys.path.insert(0, ROOT)
    import bench
    with contextlib.redirect_stdout(io.StringIO()):
        cfg, model, batch, desc = bench.build(workload, _dev())
    return bench, cfg, model


@pytest.mark.parametrize("workload", ["dual", "slowfast", "ghostnet"])
def test_model_properties_at_baseline_size(workload):
    """cfg #3 / #2 / #5 of BASELINE.json at 224^2, T = 32 (3 clips to keep the test short; cfg #5 = SlowFastGhostNet
    w2.0 + CMDA, whose s1_fuse attention runs at N = 100 352, d = 4: custom_video_model_builder.py:872-1005)."""
    from slowfast.models import engine
    dev = _dev()
    bench, cfg, model = _model(workload)
    clips = bench.synthetic_clips(cfg, 3, dev, 5)
    model.eval()
    with torch.no_grad():
        p_all = model([clips[0].clone(), clips[1].clone()])
        p_one = model([clips[0][1:2].clone(), clips[1][1:2].clone()])
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            model([clips[0].clone(), clips[1].clone()])
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        static = [clips[0].clone(), clips[1].clone()]
        with torch.cuda.graph(graph, stream=side):
            p_graph = model([static[0], static[1]])
        graph.replay()
        torch.cuda.synchronize()
    assert tuple(p_all.shape) == (3, cfg.MODEL.NUM_CLASSES) and bool(torch.isfinite(p_all).all())
    if workload != "ghostnet":  # GhostNet's head ends in ReLU, not softmax (head_helper.py:640-653, bug-compatible)
        assert float((p_all.sum(1) - 1.0).abs().max()) < 1e-5      # softmax-mean probabilities
    # clips do not interact in eval mode (a different batch size may pick other conv tilings: fp32 re-association only)
    assert float((p_all[1:2] - p_one).abs().max()) < 1e-5 * max(1.0, float(p_one.abs().max()))
    assert torch.equal(p_graph, p_all)                              # hipGraph replay == eager launches
    # training step: deterministic, and the two-stream schedule equals the serial one bit for bit
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    labels = torch.tensor([3, 11, 7], device=dev)

    def step():
        model.load_state_dict(sd)
        model.zero_grad(set_to_none=True)
        out = model([clips[0].clone(), clips[1].clone()])
        torch.nn.functional.cross_entropy(out, labels).backward()
        torch.cuda.synchronize()
        return torch.cat([out.detach().reshape(-1)] + [p.grad.reshape(-1) for p in model.parameters()]).clone()

    saved = engine.OVERLAP_PATHS
    try:
        a = step()
        b = step()
        engine.OVERLAP_PATHS = False
        c = step()
    finally:
        engine.OVERLAP_PATHS = saved
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b) and torch.equal(a, c)
    _report("%s 224^2 T=32 B=3: eval row-sum err %.1e, batch independence %.1e, graph == eager, train step deterministic "
            "and == serial schedule (|grad| %.3e)" % (workload, float((p_all.sum(1) - 1.0).abs().max()),
                                                   float((p_all[1:2] - p_one).abs().max()), float(a.norm())))


@pytest.mark.parametrize("workload,clips", [("dual", 1), ("slowfast", 1), ("dual", 8), ("slowfast", 8)],
                         ids=["dual", "slowfast", "dual_b8", "slowfast_b8"])
def test_fullsize_eval_forward_matches_oracle(workload, clips):
    """cfg #3 / #2 of BASELINE.json at their real size (224^2, T = 32, alpha = 4), ONE clip and the benchmark's EIGHT:
    every top-level child's output, the pre-activation logits and the output probabilities of the HIP eval forward
    against the oracle's on the same seeded parameters and clips (north_star: within 1e-3 rel fp32, max-norm).  Clips do
    not interact in eval mode, so the 8-clip HIP forward — the conv tiles, sweep parts and attention variants the
    benchmark's batch selects — is compared row by row with eight one-clip oracle forwards (the oracle needs ~6 GB and
    a few seconds per clip for the dense N = 25 088 attention)."""
    import sfhip
    from oracle import slowfast_oracle as oracle
    from slowfast.models import head_helper
    dev = _dev()
    bench, cfg, model = _model(workload)
    xs = bench.synthetic_clips(cfg, clips, "cpu", 1)
    model.eval()
    acts, tap = {}, {}

    def hook(child):
        def f(m, i, o):
            if isinstance(o, (list, tuple)):
                acts[child] = [sfhip.to_ncthw(a).cpu() if isinstance(a, sfhip.Act) else a.cpu() for a in o]
        return f

    handles = [m.register_forward_hook(hook(n)) for n, m in model.named_children()]
    head_helper.LOGITS_TAP = lambda t: tap.__setitem__("logits", t.detach().cpu())
    try:
        with torch.no_grad():
            out = model([x.to(dev) for x in xs]).cpu()
        torch.cuda.synchronize()
    finally:
        head_helper.LOGITS_TAP = None
        for h in handles:
            h.remove()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max())

    worst, checked, e_log, e_out = 0.0, 0, 0.0, 0.0
    for b in range(clips):
        ref = oracle.forward(cfg.MODEL.MODEL_NAME, sd, [x[b:b + 1] for x in xs], bench.oracle_hparams(cfg))
        for child, outs in acts.items():
            if child not in ref:
                continue
            for i, a in enumerate(outs):
                e = rel(a[b:b + 1], ref[child][i])
                worst = max(worst, e)
                checked += 1
                assert e < 1e-3, (child, i, b, e)
        e_log = max(e_log, rel(tap["logits"][b:b + 1].reshape(1, -1), ref["logits"].reshape(1, -1)))
        e_out = max(e_out, rel(out[b:b + 1], ref["out"]))
    _report("%s 224^2 T=32 B=%d vs oracle: %d child outputs, worst %.2e; logits %.2e; output %.2e" % (
        workload, clips, checked, worst, e_log, e_out))
    assert checked >= 16 * clips and e_log < 1e-3 and e_out < 1e-3


# relative-L2 bounds of the full-size training step against the oracle (fp32 both sides), with the HIP forward's ReLU
# masks AND max-pool arg-max decisions injected into the oracle (tests/_masks.py: both sides differentiate the same
# piecewise-linear function).  Without injection the comparison measures flips: 2.9e-2 median at this size with the
# forward equal to 5e-6; with the ReLU masks alone 2.8e-4 .. 1.1e-3 median, 1.3e-3 on the input gradients and 1.5e-2 on a
# scalar gamma (stem / CMDA pool winners still decided independently).  With both, measured on MI355X: cfg #3 median
# 2.5e-4, p90 3.0e-4, worst 3.2e-3 (s3_fuse gamma: one scalar summed over every position), inputs 2.7e-4; cfg #2 median
# 9.7e-5, worst 1.8e-4, inputs 9.7e-5.  The bounds leave a factor of 3-4 on that.
TRAIN_TOL = {"loss": 1e-4, "logits": 1e-3, "grad_median": 1e-3, "grad_worst": 1e-2, "grad_input": 1e-3}


def _host_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:  # noqa: BLE001
        return 0.0


# (workload, clips): one clip of every BASELINE architecture; THREE clips of cfg #3 — batch-statistics BN over more than
# one clip and the batch-keyed kernel choices (attention backward variant <0, 8>, sweep parts, conv tiles, weight-gradient
# splits) that a one-clip step never takes.  Host memory: the oracle's dense N = 25 088 attention keeps ~20 GB per clip
# alive for autograd (cfg #5: N = 100 352 at s1_fuse, ~200 GB).
# ("dual", 8) is the benchmark's own batch (tools/train_net.py:78-96 at TRAIN.BATCH_SIZE / NUM_GPUS = 8): ~240 GB of
# host memory for the oracle's eight dense attention graphs — skipped below that.
@pytest.mark.parametrize("workload,clips", [("dual", 1), ("slowfast", 1), ("dual", 3), ("ghostnet", 1), ("dual", 8)],
                         ids=["dual", "slowfast", "dual_b3", "ghostnet", "dual_b8"])
def test_fullsize_train_step_matches_oracle(workload, clips):
    """cfg #3 / #2 / #5 at their real size (224^2, T = 32), dropout off: the HIP training step (train-mode forward
    with batch-statistics BN, cross-entropy, backward through every kernel) against the oracle's autograd on the same
    seeded parameters and clips — loss, train-mode logits, EVERY parameter's gradient (relative L2) and dL/d(clip) of
    both pathways.  The caller being matched is tools/train_net.py:78-96 of the reference."""
    import _masks
    import _zero_grads
    from oracle import slowfast_oracle as oracle
    need = {"dual": 30, "slowfast": 12, "ghostnet": 320}[workload] * clips + (20 if clips >= 8 else 0)
    if _host_gb() < need:
        pytest.skip("the oracle needs ~%d GB of host memory for this case" % need)
    dev = _dev()
    bench, cfg, model = _model(workload)
    xs = bench.synthetic_clips(cfg, clips, "cpu", 1)
    label = torch.tensor([5, 17, 301, 0, 399, 123, 250, 77][:clips])
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if workload == "dual" and clips >= 3:  # the variant bench.py times at 8 clips
        import sfhip
        if sfhip.lib().sf_attn_products_per_fp32(32) == 6:
            assert sfhip.lib().sf_attn_bwd_variant(clips, 25088, 32) == sfhip.lib().sf_attn_bwd_variant(8, 25088, 32) == 38
    model.train()
    model.zero_grad(set_to_none=True)
    gx = [x.to(dev).requires_grad_(True) for x in xs]
    with _masks.capture() as masks:
        logits = model(gx)
    loss = torch.nn.functional.cross_entropy(logits, label.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    got = {k: v.grad.detach().cpu() for k, v in model.named_parameters()}
    got_in = [x.grad.detach().cpu() for x in gx]
    # ---- oracle: autograd through the CPU restatement (dense attention: ~20 GB of host memory per clip with the graph kept)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    sdr = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v)
           for k, v in sd.items()}
    rx = [x.clone().requires_grad_(True) for x in xs]
    with _masks.inject(masks):
        acts = oracle.FORWARDS[cfg.MODEL.MODEL_NAME](sdr, rx, bench.oracle_hparams(cfg), training=True)
    assert not masks.missed and masks.used == masks.count and masks.pool_used == masks.pool_count, (
        masks.missed[:4], masks.used, masks.count, masks.pool_used, masks.pool_count)
    rloss = torch.nn.functional.cross_entropy(acts["out"], label)
    rloss.backward()

    def l2(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))

    e_logits = float((logits.detach().cpu() - acts["out"].detach()).abs().max() / acts["out"].detach().abs().max())
    e_loss = abs(float(loss.detach()) - float(rloss.detach()))
    ref = {k: v.grad for k, v in sdr.items() if getattr(v, "grad", None) is not None}
    missing = [k for k in ref if k not in got]
    # gradients that are ZERO in exact arithmetic (tests/_zero_grads.py lists the classes and why): rounding noise on
    # both sides, bounded absolutely against the model's largest parameter gradient
    noise, gmax = _zero_grads.split(ref)
    _zero_grads.check_noise(got, ref, noise, gmax)
    # the class list takes exactly the parameters it is meant to: a pattern that starts to swallow real gradients, or a
    # renamed module dropping out of it, changes the count
    # (pinned at ONE clip, the shape bench.py's parity leg runs: with more clips a few of the listed parameters' oracle
    # gradients rise above the 1e-5 noise line and are then compared relatively like every other one)
    if clips == 1:
        assert len(noise) == _zero_grads.EXPECTED[workload], (len(noise), sorted(noise)[:8])
    assert len(noise) <= _zero_grads.EXPECTED[workload], (len(noise), sorted(noise)[:8])
    errs = sorted((l2(got[k], g), k) for k, g in ref.items() if k not in noise and float(g.norm()) > 0)
    e_in = [l2(a, b.grad) for a, b in zip(got_in, rx)]
    med, worst = errs[len(errs) // 2][0], errs[-1]
    _report("%s 224^2 T=32 B=%d TRAIN STEP vs oracle (%d ReLU masks, %d max-pool arg-max sets injected): loss |d| %.2e (%.5f), logits %.2e, %d "
            "parameter gradients rel-L2 median %.2e p90 %.2e worst %.2e (%s), %d analytically-zero gradients bounded at %.0e of the largest, "
            "input gradients slow %.2e fast %.2e" % (
                workload, clips, masks.count, masks.pool_count, e_loss, float(rloss.detach()), e_logits, len(errs), med,
                errs[len(errs) * 9 // 10][0], worst[0], worst[1], len(noise), _zero_grads.ABS_BOUND, e_in[0], e_in[1]))
    for e, k in errs[-6:]:
        _report("    %-70s %.2e" % (k, e))
    assert not missing, missing
    assert len(errs) >= 150
    assert e_loss < TRAIN_TOL["loss"] * max(1.0, abs(float(rloss.detach()))) and e_logits < TRAIN_TOL["logits"]
    assert med < TRAIN_TOL["grad_median"] and worst[0] < TRAIN_TOL["grad_worst"], (med, worst)
    assert max(e_in) < TRAIN_TOL["grad_input"], e_in


def test_cfg1_every_gradient_within_the_reference_own_fp32_noise():
    """BASELINE cfg #1 exactly (SlowFastShuffleNetV2 w0.25, 4x16, 32^2, 2 clips), EVERY parameter gradient of the
    training step against the oracle in fp64 — and, beside it, the oracle in fp32 against the same fp64 run (all three
    on the HIP forward's ReLU masks / max-pool winners).  At 32^2 the last stages are 1 x 1 frames, batch statistics
    are taken over 8 values per channel and a few 2-element BN gradients are differences of nearly equal sums: round
    4's bench line showed `s2.pathway1_channel_4.features.3.banch2.4.weight` at 1.2e-2 with a median of 1.7e-4.  The
    reference's own fp32 arithmetic is off by the same amount on the same parameter (tools/oracle_conditioning.py,
    profiles/r05_oracle_conditioning_shufflenetv2.txt: 1.04e-2, median 1.67e-4), so the bound per parameter is
    max(1e-3, 4 x the reference's own fp32-vs-fp64 error).  The caller matched: tools/train_net.py:78-96."""
    import _masks
    import _zero_grads
    from oracle import slowfast_oracle as oracle
    dev = _dev()
    bench, cfg, model = _model("shufflenetv2")
    clips = 2
    xs = bench.synthetic_clips(cfg, clips, "cpu", 1)
    label = torch.arange(clips, dtype=torch.long) % cfg.MODEL.NUM_CLASSES
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.train()
    model.zero_grad(set_to_none=True)
    with _masks.capture() as masks:
        logits = model([x.to(dev) for x in xs])
    torch.nn.functional.cross_entropy(logits, label.to(dev)).backward()
    torch.cuda.synchronize()
    got = {k: v.grad.detach().cpu() for k, v in model.named_parameters()}
    hp = bench.oracle_hparams(cfg)

    def oracle_grads(dtype):
        sdr = {k: (v.to(dtype).clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k
                   else (v.to(dtype) if v.dtype == torch.float32 else v)) for k, v in sd.items()}
        with _masks.inject(masks.fork()) as mk:
            acts = oracle.FORWARDS[cfg.MODEL.MODEL_NAME](sdr, [x.to(dtype).clone() for x in xs], hp, training=True)
        assert not mk.missed, mk.missed[:4]
        torch.nn.functional.cross_entropy(acts["out"], label).backward()
        return {k: v.grad.detach().double() for k, v in sdr.items() if getattr(v, "grad", None) is not None}

    g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
    noise_class, gmax = _zero_grads.split(g64)
    _zero_grads.check_noise(got, {k: v.float() for k, v in g64.items()}, noise_class, gmax)
    rows = []
    for k, g in g64.items():
        if k in noise_class or float(g.norm()) == 0:
            continue
        e_hip = float((got[k].double() - g).norm() / g.norm())
        e_ref = float((g32[k] - g).norm() / g.norm())
        rows.append((e_hip, e_ref, k))
    rows.sort(reverse=True)
    med = sorted(r[0] for r in rows)[len(rows) // 2]
    med_ref = sorted(r[1] for r in rows)[len(rows) // 2]
    _report("cfg #1 TRAIN STEP, %d parameter gradients vs oracle fp64: HIP median %.2e worst %.2e (%s); the oracle's own "
            "fp32 run vs fp64: median %.2e, on that parameter %.2e" % (len(rows), med, rows[0][0], rows[0][2], med_ref,
                                                                      rows[0][1]))
    assert len(rows) >= 300
    assert med < 1e-3
    for e_hip, e_ref, k in rows:
        assert e_hip <= max(1e-3, 4.0 * e_ref), (k, e_hip, e_ref)
