"""hipGraph replay of the TRAINING step == its eager launches, bit for bit.

bench.py times whichever of the two launch forms is faster warm; cfg #1 (SlowFastShuffleNetV2: ~1500 launches of a
0.007 GMAC model) and cfg #5 at 2 clips always take the replay.  The step is bench.make_train_step's closure — the
reference loop tools/train_net.py:78-96: zero grads, train-mode forward (batch-statistics BN with running-stat
updates, dropout), cross-entropy, backward, (all-reduce), SGD with momentum and weight decay.  Every kernel is
deterministic and nn.Dropout draws from torch's graph-safe Philox state, so K replays from a snapshot must leave the
same parameters, BN buffers, momentum buffers, flat gradient and loss as K eager steps from the same snapshot.
"""
import contextlib
import io
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
K = 3


def _state(model, opt):
    """every tensor a training step changes in place: parameters, BN buffers, SGD momentum buffers"""
    named = [("p/" + k, v) for k, v in model.named_parameters()] + [("b/" + k, v) for k, v in model.named_buffers()]
    for i, p in enumerate(model.parameters()):
        mb = opt.state.get(p, {}).get("momentum_buffer")
        if mb is not None:
            named.append(("m/%d" % i, mb))
    return named


@pytest.mark.parametrize("workload,clips", [("dual", 2), ("shufflenetv2", 2), ("ghostnet", 2)])
def test_graph_replayed_train_steps_equal_eager_steps(workload, clips):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, ROOT)
    import bench
    from slowfast.models import engine
    try:
        _run(bench, workload, clips)
    finally:
        engine.set_grad_sink(False)   # make_train_step switched the in-kernel gradient sink on (process-global)


def _run(bench, workload, clips):
    dev = torch.device("cuda", 0)
    with contextlib.redirect_stdout(io.StringIO()):
        cfg, model, _, _ = bench.build(workload, dev)
    xs = bench.synthetic_clips(cfg, clips, dev, 11)
    labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (clips,), device=dev,
                           generator=torch.Generator(device=dev).manual_seed(3))
    # lr large enough that three steps move every weight by many ulps (a stale packed-weight copy would show)
    step, flat, opt = bench.make_train_step(model, xs, labels, overlap_allreduce=True, lr=0.02)
    assert any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in model.modules()), "dropout is part of the step"
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):      # momentum buffers, packed-weight caches, allocator pools
            step()
    torch.cuda.synchronize()
    state = _state(model, opt)
    assert sum(1 for k, _ in state if k.startswith("m/")) == len(list(model.parameters()))
    snap = [v.detach().clone() for _, v in state]

    def restore():
        with torch.no_grad():
            for (_, v), s in zip(state, snap):
                v.copy_(s)
        torch.cuda.manual_seed(4242)
        torch.cuda.synchronize()

    def result(losses):
        torch.cuda.synchronize()
        return ([v.detach().clone() for _, v in state], flat.flat.detach().clone(),
                torch.stack([x.detach().reshape(()) for x in losses]).clone())

    restore()
    losses = []
    with torch.cuda.stream(side):
        for _ in range(K):
            losses.append(step().detach().clone())
    eager = result(losses)

    restore()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        static_loss = step()
    restore()   # the capture itself must not have run anything, but the generator and the caches are re-armed anyway
    losses = []
    with torch.cuda.stream(side):
        for _ in range(K):
            g.replay()
            losses.append(static_loss.detach().clone())
    replay = result(losses)

    assert bool(torch.isfinite(eager[2]).all()) and bool(torch.isfinite(eager[1]).all())
    assert torch.equal(eager[2], replay[2]), ("loss", eager[2].tolist(), replay[2].tolist())
    assert torch.equal(eager[1], replay[1]), "flat gradient of the last step"
    changed = 0
    for (name, _), a, b, s in zip(state, eager[0], replay[0], snap):
        assert torch.equal(a, b), name
        changed += int(not torch.equal(a, s))
    # the steps really trained: (nearly) every tensor moved away from the snapshot
    assert changed > 0.9 * len(state), (changed, len(state))
    # and the replay is not a constant: the K losses differ from step to step
    assert len(set(eager[2].tolist())) == K


def test_eval_after_replays_sees_the_trained_weights():
    """eager eval -> K replays of a captured training step (engine.replay) -> eager eval must equal the
    eval after K eager steps from the same snapshot; with a bare graph.replay() the second eval would run on the packed
    weights / folded BN cached by the first one."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, ROOT)
    import bench
    from slowfast.models import engine
    try:
        dev = torch.device("cuda", 0)
        with contextlib.redirect_stdout(io.StringIO()):
            cfg, model, _, _ = bench.build("shufflenetv2", dev)
        xs = bench.synthetic_clips(cfg, 2, dev, 11)
        labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (2,), device=dev,
                               generator=torch.Generator(device=dev).manual_seed(3))
        step, flat, opt = bench.make_train_step(model, xs, labels, overlap_allreduce=False, lr=0.002)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.synchronize()
        state = _state(model, opt)
        snap = [v.detach().clone() for _, v in state]

        def restore():
            with torch.no_grad():
                for (_, v), s_ in zip(state, snap):
                    v.copy_(s_)
            torch.cuda.manual_seed(77)
            torch.cuda.synchronize()

        def evaluate():
            model.eval()
            with torch.no_grad(), torch.cuda.stream(side):
                out = model([xs[0], xs[1]]).detach().clone()
            torch.cuda.synchronize()
            model.train()
            return out

        restore()
        before = evaluate()
        with torch.cuda.stream(side):
            for _ in range(K):
                step()
        want = evaluate()
        assert not torch.equal(before, want), "the steps must move the eval output"

        # one eager step between the eval forward and the capture: the eval pass re-packed every conv weight one by one
        # into fresh tensors, and the batched re-pack's device-side pointer table is rebuilt (a host-to-device copy,
        # not capturable) the first time it sees them
        with torch.cuda.stream(side):
            step()
        restore()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step()
        restore()
        assert torch.equal(evaluate(), before)          # fills the eager caches from the snapshot weights
        with torch.cuda.stream(side):
            for _ in range(K):
                engine.replay(g)
        got = evaluate()
        assert torch.equal(got, want), float((got - want).abs().max())
    finally:
        engine.set_grad_sink(False)
