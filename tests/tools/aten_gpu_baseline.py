#!/usr/bin/env python3
"""The ORACLE's functional graph run on the GPU through stock ATen (MIOpen / rocBLAS kernels, dense N x N attention)
— a measured stand-in for "the reference's nn.Modules on ROCm" on the same MI355X.  Test tooling (it executes the
oracle), kept out of bench.py.  usage: tests/tools/aten_gpu_baseline.py [workload] [train|eval] [batch]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
from bench import synthetic_clips  # noqa: E402


def aten_gpu_baseline(cfg, model, batch, device, train, steps):
    """The oracle's functional graph run ON THE GPU through stock ATen (MIOpen / rocBLAS kernels, dense N x N
    attention) — i.e. what the reference's nn.Modules execute on ROCm.  Informational (--aten-gpu-baseline):
    it is a measured stand-in for 'the reference on MI355X', never part of `value`."""
    from oracle import slowfast_oracle as oracle
    hp = oracle.default_hparams(
        alpha=cfg.SLOWFAST.ALPHA, beta_inv=cfg.SLOWFAST.BETA_INV, depth=cfg.RESNET.DEPTH,
        width_per_group=cfg.RESNET.WIDTH_PER_GROUP, num_groups=cfg.RESNET.NUM_GROUPS,
        fusion_kernel=cfg.SLOWFAST.FUSION_KERNEL_SZ,
        spatial_strides=[s[0] for s in cfg.RESNET.SPATIAL_STRIDES],
        spatial_dilations=[s[0] for s in cfg.RESNET.SPATIAL_DILATIONS],
        num_block_temp_kernel=[list(x) for x in cfg.RESNET.NUM_BLOCK_TEMP_KERNEL],
        num_frames=cfg.DATA.NUM_FRAMES, crop_size=cfg.DATA.CROP_SIZE, num_classes=cfg.MODEL.NUM_CLASSES,
        short_cycle=bool(cfg.MULTIGRID.SHORT_CYCLE), head_act=cfg.MODEL.HEAD_ACT,
        width_multi=cfg.SLOWFAST.WIDTH_MULTI)
    name = cfg.MODEL.MODEL_NAME
    xs = synthetic_clips(cfg, batch, device, 1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if train:
        sd = {k: (v.requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v)
              for k, v in sd.items()}
        labels = torch.zeros(batch, dtype=torch.long, device=device)

    def step():
        if train:
            acts = oracle.FORWARDS[name](sd, [x.clone() for x in xs], hp, training=True)
            torch.nn.functional.cross_entropy(acts["out"], labels).backward()
            for v in sd.values():
                v.grad = None
        else:
            with torch.no_grad():
                oracle.FORWARDS[name](sd, [x.clone() for x in xs], hp, training=False)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(batch / dt, 3), "unit": "clips/s", "ms_per_step": round(dt * 1e3, 2), "batch": batch,
            "what": "oracle graph on stock ATen ROCm kernels (MIOpen conv, rocBLAS bmm + softmax: dense N x N "
                    "attention), %s, same GPU" % ("train-mode forward + CE + autograd backward" if train
                                                  else "eval forward"),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}




if __name__ == "__main__":
    workload = sys.argv[1] if len(sys.argv) > 1 else "dual"
    train = (sys.argv[2] if len(sys.argv) > 2 else "eval") == "train"
    dev = torch.device("cuda:0")
    cfg, model, batch, desc = bench.build(workload, dev)
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else (1 if train else batch)
    print(json.dumps(aten_gpu_baseline(cfg, model, batch, dev, train, 3)))
