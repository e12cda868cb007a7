"""One rank of the 2-process NaiveSyncBatchNorm3d parity run (launched by test_syncbn_gpu.py).

Both ranks share the box's single GPU and talk over gloo/127.0.0.1 (RCCL refuses two ranks on one device); the
collective code path is the product's `group_gather_sum`.  Writes <out>/rank<r>.json with the errors against
tests/golden/dual_r50_syncbn_s64.npz (the reference run as two ranks)."""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "efficient-slowfast_amd"), HERE, os.path.join(HERE, "golden")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank,
                            world_size=world)
    from _util import case_inputs, load_case, rel_err, sample_activation, seeded_state_dict
    import slowfast.utils.distributed as du
    from slowfast.config.defaults import get_cfg
    from slowfast.models import build_model
    du._LOCAL_PROCESS_GROUP = dist.new_group(list(range(world)))  # init_distributed_training for one machine
    z, meta = load_case("dual_r50_syncbn_s64")
    cfg = get_cfg()
    cfg.merge_from_other_cfg(meta["cfg_dump"])
    cfg.NUM_GPUS = 1
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_model(cfg)
    model.load_state_dict(seeded_state_dict(z["sd_keys"], z["sd_shapes"], meta["param_seed"]))
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    per = meta["batch"] // world
    sl = slice(rank * per, (rank + 1) * per)
    xs = [x[sl].cuda() for x in case_inputs(meta)]
    labels = torch.from_numpy(z["labels"])[sl].cuda()
    logits = model(xs)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    rep = {"rank": rank, "logits": rel_err(logits.detach().cpu().numpy(), z["r%d/logits" % rank]),
           "loss": abs(loss.item() - float(z["r%d/loss" % rank][0])), "grads": {}, "scalar_grads": {}, "buffers": {}}
    params = dict(model.named_parameters())
    pre = "r%d/grad/" % rank
    for tag in z.files:
        if tag.startswith(pre) and not tag.endswith("/stats"):
            k = tag[len(pre):]
            s, _, _ = sample_activation(params[k].grad.cpu().numpy(), 4096)
            ref = z[tag].astype(np.float64)
            # scalar parameters (SpatialAttention.gamma): one ReLU mask flip at an fp32 tie moves them by several %
            rep["scalar_grads" if params[k].numel() < 16 else "grads"][k] = float(
                np.linalg.norm(s - ref) / max(np.linalg.norm(ref), 1e-30))
    after = model.state_dict()
    pre = "r%d/buffers/" % rank
    for tag in z.files:
        if tag.startswith(pre):
            rep["buffers"][tag[len(pre):]] = rel_err(after[tag[len(pre):]].cpu().numpy(), z[tag])
    import sfhip
    rep["lib"] = sfhip.lib_path()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(rep, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
