#!/usr/bin/env python3
"""Golden vectors for NaiveSyncBatchNorm3d: the REFERENCE run as TWO ranks (CPU, gloo, 127.0.0.1) with
BN.NORM_TYPE sync_batchnorm / NUM_SYNC_DEVICES 2, each rank holding half of a batch of 4.

Build container only (needs /root/reference).  Writes tests/golden/dual_r50_syncbn_s64.npz with, per rank:
train-mode logits, CE loss, sampled LOCAL parameter gradients (before any DDP averaging), and a few BN running
statistics after the one training forward.  Data only; parameters/clips regenerate from the seeds in `meta`."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

NAME = "dual_r50_syncbn_s64"
BATCH, T, ALPHA, SIZE, WORLD = 4, 16, 4, 64, 2


def worker(rank, port, ret):
    from _refimport import import_reference, ref_yaml
    from paramgen import fill_state_dict, make_clip, sample_activation
    import make_golden as mg
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    torch.set_num_threads(4)
    get_cfg, build_model = import_reference()
    import slowfast.utils.distributed as du
    du._LOCAL_PROCESS_GROUP = dist.new_group(list(range(WORLD)))
    cfg = get_cfg()
    cfg.merge_from_file(ref_yaml(mg.DUAL_YAML))
    over = mg.COMMON + ["MODEL.MODEL_NAME", "SlowFastDualAttention", "BN.NORM_TYPE", "sync_batchnorm",
                        "BN.NUM_SYNC_DEVICES", WORLD] + mg.small(SIZE, T)
    cfg.merge_from_list(over)
    torch.manual_seed(0)
    model = build_model(cfg)
    sd = model.state_dict()
    fill_state_dict(sd, mg.PARAM_SEED)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    slow, fast = make_clip(mg.CLIP_SEED, BATCH, T, ALPHA, SIZE)
    per = BATCH // WORLD
    sl = slice(rank * per, (rank + 1) * per)
    labels = torch.from_numpy(np.random.RandomState(11).randint(0, cfg.MODEL.NUM_CLASSES, BATCH))
    logits = model([torch.from_numpy(slow[sl].copy()), torch.from_numpy(fast[sl].copy())])
    loss = torch.nn.functional.cross_entropy(logits, labels[sl])
    loss.backward()
    out = {"r%d/logits" % rank: logits.detach().numpy(), "r%d/loss" % rank: np.array([loss.item()])}
    params = dict(model.named_parameters())
    for k in mg.GRAD_KEYS["SlowFastDualAttention"] + ["s1.pathway0_stem.bn.weight", "s5.pathway1_res2.branch2.c_bn.bias"]:
        g = params[k].grad
        s, amax, mean = sample_activation(g.numpy(), 4096)
        out["r%d/grad/%s" % (rank, k)] = s
        out["r%d/grad/%s/stats" % (rank, k)] = np.array([amax, float(g.norm())], np.float64)
    after = model.state_dict()
    bufs = [k for k in after if k.endswith("running_mean") or k.endswith("running_var")]
    for k in bufs[:4] + bufs[len(bufs) // 2:len(bufs) // 2 + 4] + bufs[-4:]:
        out["r%d/buffers/%s" % (rank, k)] = after[k].numpy().copy()
    if rank == 0:
        out["meta"] = json.dumps(dict(
            name=NAME, model="SlowFastDualAttention", yaml=mg.DUAL_YAML,
            overrides=[str(o) if not isinstance(o, (int, float, bool)) else o for o in over],
            cfg_dump=mg.plain_cfg(cfg), hparams=mg.hparams_from_cfg(cfg), param_seed=mg.PARAM_SEED,
            clip_seed=mg.CLIP_SEED, batch=BATCH, t=T, alpha=ALPHA, size=SIZE, world=WORLD, torch=torch.__version__))
        out["sd_keys"] = np.array(list(sd.keys()))
        out["sd_shapes"] = np.array([json.dumps(list(v.shape)) for v in sd.values()])
        out["labels"] = labels.numpy()
    np.savez(os.path.join(ret, "part%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile
    tmp = tempfile.mkdtemp()
    mp.spawn(worker, args=(29611, tmp), nprocs=WORLD, join=True)
    merged = {}
    for r in range(WORLD):
        z = np.load(os.path.join(tmp, "part%d.npz" % r))
        merged.update({k: z[k] for k in z.files})
    path = os.path.join(HERE, NAME + ".npz")
    np.savez_compressed(path, **merged)
    print("%s %.1f KB  losses %s" % (NAME, os.path.getsize(path) / 1024,
                                      [float(merged["r%d/loss" % r][0]) for r in range(WORLD)]))
