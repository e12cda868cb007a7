"""world_size-2 gloo tests (CPU) of the data-parallel path: flat-buffer gradient all-reduce == the mean of the
per-rank gradients == single-process gradient of the concatenated batch; clip sharding; max-over-ranks."""
import os
import sys
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv3d(3, 4, 1), torch.nn.BatchNorm3d(4), torch.nn.ReLU(),
                               torch.nn.AdaptiveAvgPool3d(1), torch.nn.Flatten(), torch.nn.Linear(4, 5))


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "efficient-slowfast_amd"))
    from slowfast.utils.distributed import FlatGradients, max_over_ranks, shard_sizes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _make_model()
        flat = FlatGradients(model.parameters())
        g = torch.Generator().manual_seed(5)
        x = torch.randn(4, 3, 2, 4, 4, generator=g)
        y = torch.randint(0, 5, (4,), generator=g)
        per = shard_sizes(4, world)[rank]
        xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
        flat.zero()
        torch.nn.functional.cross_entropy(model(xs), ys).backward()
        local = flat.flat.clone()
        flat.all_reduce_mean()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(flat.flat, sum(gathered) / world, atol=1e-7)
        # every parameter's .grad is still a view of the flat buffer (no copies)
        off = 0
        for p in flat.params:
            assert p.grad.data_ptr() == flat.flat.data_ptr() + 4 * off
            off += p.numel()
        assert max_over_ranks(float(rank + 1), "cpu") == float(world)
        if rank == 0:
            torch.save(flat.flat.clone(), out)
    finally:
        dist.destroy_process_group()


def test_flat_gradient_allreduce_world2(tmp_path):
    out = str(tmp_path / "g.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    # per-rank BN statistics (the reference default, BN.NORM_TYPE batchnorm) => compare with the mean of
    # two independent half-batch gradients computed in this process
    model = _make_model()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 2, 4, 4, generator=g)
    y = torch.randint(0, 5, (4,), generator=g)
    grads = []
    for r in range(2):
        model.zero_grad()
        torch.nn.functional.cross_entropy(model(x[2 * r:2 * r + 2]), y[2 * r:2 * r + 2]).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
        # keep the running stats identical to the workers' (each saw ONE half batch)
        model = _make_model()
    assert torch.allclose(got, (grads[0] + grads[1]) / 2, atol=1e-6)


def test_shard_sizes():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "efficient-slowfast_amd"))
    from slowfast.utils.distributed import shard_sizes
    assert shard_sizes(64, 8) == [8] * 8
    with pytest.raises(ValueError):
        shard_sizes(10, 4)


def _gather_worker(rank, world, port):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "efficient-slowfast_amd"))
    import slowfast.utils.distributed as du
    from slowfast.models.batchnorm_helper import NaiveSyncBatchNorm3d, get_norm, group_gather_sum
    from slowfast.config.defaults import get_cfg
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        assert du.get_local_size() == world  # default group until init_distributed_training narrows it
        cfg = get_cfg()
        cfg.NUM_GPUS, cfg.SHARD_ID = world, 0
        du.init_distributed_training(cfg)
        assert du.get_local_size() == world and du.get_local_rank() == rank
        vec = torch.arange(6, dtype=torch.float64) + 10.0 * rank
        tot = group_gather_sum(vec, world, 1)              # one sync group of both ranks
        assert torch.equal(tot, 2 * torch.arange(6, dtype=torch.float64) + 10.0)
        own = group_gather_sum(vec, 1, world)              # NUM_SYNC_DEVICES 1 -> every rank its own group
        assert torch.equal(own, vec)
        cfg.BN.NORM_TYPE, cfg.BN.NUM_SYNC_DEVICES = "sync_batchnorm", 1
        bn = get_norm(cfg)(num_features=4)
        assert isinstance(bn, NaiveSyncBatchNorm3d) and (bn.num_sync_devices, bn.num_groups) == (1, world)
        cfg.BN.NUM_SYNC_DEVICES = 0                        # <= 0 means "all local ranks"
        bn = get_norm(cfg)(num_features=4)
        assert (bn.num_sync_devices, bn.num_groups) == (world, 1)
    finally:
        dist.destroy_process_group()


def test_sync_bn_group_gather_world2():
    """GroupGather semantics (batchnorm_helper.py:112-171) and the per-node process group on 2 gloo ranks."""
    mp.spawn(_gather_worker, args=(2, _free_port()), nprocs=2, join=True)


class _Staged(torch.nn.Module):
    """Top-level children named like the models' (s3, s4, s5, head): what FlatGradients cuts the flat buffer at."""

    def __init__(self):
        super(_Staged, self).__init__()
        torch.manual_seed(1)
        self.s3 = torch.nn.Linear(6, 7)
        self.s4 = torch.nn.Linear(7, 9)
        self.s4_fuse = torch.nn.Linear(9, 9)
        self.s5 = torch.nn.Linear(9, 8)
        self.head = torch.nn.Linear(8, 5)

    def forward(self, x):
        return self.head(torch.relu(self.s5(self.s4_fuse(torch.relu(self.s4(torch.relu(self.s3(x))))))))


class _FakeTape(object):
    def __init__(self, model):
        self.model, self.joins = model, ()


def test_overlap_refuses_a_reordered_parameter_list():
    """Chunk ranges are only complete when the buffer keeps the model's own parameter order (the reference's optimizer
    builder groups BN and non-BN parameters, models/optimizer.py:25-40: such a list must not be cut)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "efficient-slowfast_amd"))
    from slowfast.utils.distributed import FlatGradients
    model = _Staged()
    weights = [p for p in model.parameters() if p.dim() > 1]
    biases = [p for p in model.parameters() if p.dim() == 1]
    flat = FlatGradients(weights + biases)
    with pytest.raises(ValueError, match="model's own order"):
        flat.overlap_with_backward(model, boundaries=("s5", "s4"))
    FlatGradients(model.parameters()).overlap_with_backward(model, boundaries=("s5", "s4"))


def _chunk_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "efficient-slowfast_amd"))
    from slowfast.models import engine
    from slowfast.utils.distributed import FlatGradients
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        results = []
        for chunked in (False, True):
            model = _Staged()
            flat = FlatGradients(model.parameters())
            if chunked:
                flat.overlap_with_backward(model, boundaries=("s5", "s4"))
                engine.set_grad_sink(True)
            g = torch.Generator().manual_seed(11 + rank)
            x, y = torch.randn(3, 6, generator=g), torch.randint(0, 5, (3,), generator=g)
            flat.zero()
            torch.nn.functional.cross_entropy(model(x), y).backward()
            if chunked:  # what the tape's milestones do during the HIP backward, in backward order
                assert flat._cuts["s4"] < flat._cuts["s5"] < flat.flat.numel()
                hook = engine._MILESTONE_HOOKS[model]
                tape = _FakeTape(model)
                hook("s5", _FakeTape(_Staged()))  # the backward of ANOTHER model: not this buffer's business
                assert not flat._pending
                hook("s5", tape)
                hook("s4", tape)
                hook("s3", tape)  # not a boundary: nothing happens
                assert len(flat._pending) == 2 and flat._hi == flat._cuts["s4"]
                with pytest.raises(RuntimeError, match="second backward"):
                    hook("s5", _FakeTape(model))  # gradient accumulation on top of ranges already being reduced
            flat.all_reduce_mean()
            assert flat.chunks_last_step == (3 if chunked else 1)
            assert flat._hi == flat.flat.numel() and not flat._pending
            results.append(flat.flat.clone())
            if chunked:  # a step that never reaches all_reduce_mean(): zero() waits for its chunks and starts over
                flat.zero()
                hook("s5", _FakeTape(model))
                assert len(flat._pending) == 1
                flat.zero()
                assert not flat._pending and flat._hi == flat.flat.numel() and float(flat.flat.abs().sum()) == 0.0
            engine.set_grad_sink(False)
            engine.set_milestone_hook(None, model)
        assert torch.equal(results[0], results[1]), "chunked all-reduce differs from the single collective"
        if rank == 0:
            torch.save(results[1], out)
    finally:
        dist.destroy_process_group()


def test_chunked_allreduce_equals_single_collective_world2(tmp_path):
    """The overlapped schedule (chunks [s5+head | s4+s4_fuse | rest]) is bit-identical to ONE all-reduce of the flat
    buffer, and both equal the mean of the two ranks' gradients."""
    out = str(tmp_path / "g.pt")
    mp.spawn(_chunk_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    grads = []
    for r in range(2):
        model = _Staged()
        g = torch.Generator().manual_seed(11 + r)
        x, y = torch.randn(3, 6, generator=g), torch.randint(0, 5, (3,), generator=g)
        torch.nn.functional.cross_entropy(model(x), y).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
    assert torch.allclose(got, (grads[0] + grads[1]) / 2, atol=1e-7)


def _run_bench(args, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=300)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent starts one child per rank, rank 0's JSON line is the
    command's output, and every rank took part in the collective (the launcher contract, checked without a GPU)."""
    import json
    r = _run_bench(["--gpus", "2", "--spawn-selftest"])
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"selftest": True, "n_gpus": 2, "n_ranks_seen": 2}


def test_bench_ranks_pin_themselves_under_torch_distributed_run():
    """The driver's N > 1 form: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` starts the
    ranks itself and pins nothing — every rank then takes the CPU set bench.py's own launcher would have handed its
    LOCAL_RANK (rank_cpu_sets: sysfs only, before any GPU call); the launcher-started ranks do the same."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "SF_RANK_CPUS", "SF_RANK_NUMA"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--spawn-selftest"], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"selftest": True, "n_gpus": 2, "n_ranks_seen": 2}
    assert "2 of 2 ranks pinned themselves" in r.stderr, r.stderr
    r = _run_bench(["--gpus", "2", "--spawn-selftest"])
    assert "2 of 2 ranks pinned themselves" in r.stderr, r.stderr
    r = _run_bench(["--gpus", "2", "--spawn-selftest"], env={"SF_BENCH_PIN": "0"})
    assert "0 of 2 ranks pinned themselves" in r.stderr, r.stderr


def test_bench_launcher_reports_a_failed_rank():
    r = _run_bench(["--gpus", "2", "--spawn-selftest"], env={"SF_SELFTEST_FAIL_RANK": "1"})
    assert r.returncode == 3


def _settle_worker(rank, world, port, out):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        # rank 0's own times agree at once (it would stop after 8 steps), rank 1's only from the fourth group on
        series = {0: [10.0, 10.0, 10.0, 10.0, 10.0, 10.0], 1: [30.0, 20.0, 15.0, 15.0, 15.0, 15.0]}[rank]
        calls = []

        def measure(n):
            calls.append(n)
            dist.barrier()   # the measured steps hold collectives: a rank that ran more groups would hang here
            return series[len(calls) - 1], 1.0

        (ms, issue), steps = bench.settle(measure, torch.device("cpu"))
        torch.save((ms, issue, steps, len(calls)), "%s.%d" % (out, rank))
    finally:
        dist.destroy_process_group()


def test_steady_state_warmup_runs_the_same_steps_on_every_rank(tmp_path):
    """bench.settle decides on the slowest rank's times (one MAX all-reduce per group): both ranks stop after the same
    group although rank 0's own times would have stopped two groups earlier — a rank-local decision would leave the
    ranks with different numbers of all-reduce-carrying steps (SlowFast/tools/train_net.py:78-96 under DDP)."""
    out = str(tmp_path / "settle")
    mp.spawn(_settle_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0 == r1
    assert r0[2] == 16 and r0[3] == 4 and r0[0] == 15.0
