"""How well-conditioned are a workload's parameter gradients?  The oracle (CPU restatement of the reference) run in
fp32 against ITSELF in fp64 on the same clips, parameters and labels, the fp64 run differentiating the fp32 run's
piecewise-linear function (its ReLU masks / max-pool winners are injected, as the HIP-vs-oracle comparisons do).
What the two runs disagree by is the reference's own fp32 rounding noise on that parameter — the floor any fp32
implementation of the step sits on.  CPU only (runs in the build container).

    python tools/oracle_conditioning.py shufflenetv2 [clips]      -> profiles/r06_oracle_conditioning_<workload>.txt
"""
import collections
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "efficient-slowfast_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import _masks  # noqa: E402
import _zero_grads  # noqa: E402
import bench  # noqa: E402
from oracle import slowfast_oracle as oracle  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "shufflenetv2"
    clips = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    from paramgen import fill_state_dict
    from slowfast.config.defaults import get_cfg
    from slowfast.models import build_model
    yaml_name, _, desc = bench.WORKLOADS[workload]
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", yaml_name))
    cfg.NUM_GPUS = 0
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_model(cfg)   # CPU module tree: only its state_dict (names, shapes) is used
    fill_state_dict(model.state_dict(), bench.PARAM_SEED)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    xs = bench.synthetic_clips(cfg, clips, "cpu", 1)
    label = torch.arange(clips, dtype=torch.long) % cfg.MODEL.NUM_CLASSES
    hp = bench.oracle_hparams(cfg)
    name = cfg.MODEL.MODEL_NAME

    def run(dtype, hooks):
        sdr = {k: (v.to(dtype).clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k
                   else (v.to(dtype) if v.dtype == torch.float32 else v)) for k, v in sd.items()}
        oracle.ACT_HOOK, oracle.POOL_HOOK = hooks
        try:
            acts = oracle.FORWARDS[name](sdr, [x.to(dtype).clone() for x in xs], hp, training=True)
        finally:
            oracle.ACT_HOOK, oracle.POOL_HOOK = None, None
        loss = torch.nn.functional.cross_entropy(acts["out"], label)
        loss.backward()
        return float(loss), {k: v.grad.detach() for k, v in sdr.items() if getattr(v, "grad", None) is not None}

    # fp32 run, recording its own masks / pool winners
    rec = _masks.Masks()

    def act_rec(kind, x):
        rec.add(x.detach(), 6 if kind == "relu6" else 1)
        return None

    def pool_rec(x, kernel, stride, padding):
        rec.add_pool(x.detach(), kernel, stride, padding)
        return None

    l32, g32 = run(torch.float32, (act_rec, pool_rec))
    l64, g64 = run(torch.float64, (rec.hook, rec.pool_hook))
    noise, gmax = _zero_grads.split(g64)
    errs = sorted(((float((g32[k].double() - g).norm() / g.norm()), k) for k, g in g64.items()
                   if k not in noise and float(g.norm()) > 0), reverse=True)
    out = ["oracle fp32 vs oracle fp64 (fp32's ReLU masks / max-pool winners injected: %d / %d used, %d missed) — %s, %d clips"
           % (rec.used, rec.pool_used, len(rec.missed), desc, clips),
           "loss %.6f / %.6f; %d parameter gradients; relative L2 median %.2e, p90 %.2e, worst %.2e; %d analytically-zero "
           "gradients left out" % (l32, l64, len(errs), errs[len(errs) // 2][0], errs[len(errs) // 10][0], errs[0][0], len(noise))]
    for e, k in errs[:12]:
        out.append("    %-72s %.2e   |g| %.3e  n=%d" % (k, e, float(g64[k].norm()), g64[k].numel()))
    attn = [(e, k) for e, k in errs if "attention_spatial" in k]
    if attn:
        out.append("SpatialAttention parameters (query / key / value convs, gamma) of every fusion:")
        for e, k in sorted(attn, key=lambda t: t[1]):
            out.append("    %-72s %.2e   |g| %.3e  n=%d" % (k, e, float(g64[k].norm()), g64[k].numel()))
    txt = "\n".join(out)
    print(txt)
    with open(os.path.join(ROOT, "profiles", "r06_oracle_conditioning_%s.txt" % workload), "w") as f:
        f.write(txt + "\n")


if __name__ == "__main__":
    main()
