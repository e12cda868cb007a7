"""Per-parameter gradient parity of one HIP training step against the oracle (mask / pool-winner injection as in
tests/test_fullsize_gpu.py), every parameter listed with its error and norms — the tool that finds the parameter
classes whose gradient is zero in exact arithmetic (both sides hold rounding noise there).

    python tools/parity_params.py <workload> [batch] [crop]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    workload = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import _masks
    from oracle import slowfast_oracle as oracle
    dev = torch.device("cuda:0")
    cfg, model, _, desc = bench.build(workload, dev)
    if len(sys.argv) > 3:  # smaller crops than the YAML's (the models' heads pool globally on this path)
        cfg.DATA.CROP_SIZE = cfg.DATA.TRAIN_CROP_SIZE = cfg.DATA.TEST_CROP_SIZE = int(sys.argv[3])
    xs = bench.synthetic_clips(cfg, B, "cpu", 1)
    label = torch.arange(B) % cfg.MODEL.NUM_CLASSES
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.train()
    model.zero_grad(set_to_none=True)
    with _masks.capture() as masks:
        logits = model([x.to(dev) for x in xs])
    loss = torch.nn.functional.cross_entropy(logits, label.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    got = {k: v.grad.detach().cpu() for k, v in model.named_parameters() if v.grad is not None}
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    sdr = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v)
           for k, v in sd.items()}
    with _masks.inject(masks):
        acts = oracle.FORWARDS[cfg.MODEL.MODEL_NAME](sdr, [x.clone() for x in xs], bench.oracle_hparams(cfg), training=True)
    rloss = torch.nn.functional.cross_entropy(acts["out"], label)
    rloss.backward()
    print("%s B=%d: loss %.6f / %.6f, logits rel %.2e; masks %d/%d pools %d/%d missed %d %s" % (
        workload, B, float(loss), float(rloss),
        float((logits.detach().cpu() - acts["out"].detach()).abs().max() / acts["out"].detach().abs().max()),
        masks.used, masks.count, masks.pool_used, masks.pool_count, len(masks.missed), masks.missed[:6]))
    rows = []
    gmax = max(float(v.grad.norm()) for v in sdr.values() if getattr(v, "grad", None) is not None)
    for k, v in sdr.items():
        g = getattr(v, "grad", None)
        if g is None or k not in got:
            continue
        rows.append((float((got[k] - g).norm() / g.norm().clamp_min(1e-30)), k, float(g.norm()), float(got[k].norm()),
                     g.numel()))
    rows.sort(reverse=True)
    print("largest gradient norm %.3e; %d parameters; median rel-L2 %.2e" % (gmax, len(rows), rows[len(rows) // 2][0]))
    for e, k, rn, gn, n in rows[:int(os.environ.get("TOP", "60"))]:
        print("%-72s %.2e  |ref| %.2e |hip| %.2e  n=%d" % (k, e, rn, gn, n))


if __name__ == "__main__":
    main()
