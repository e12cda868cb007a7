"""Diagnostic: where does a hipGraph-replayed training step first differ from the eager one?
    python tests/tools/dbg_graph_step.py [workload] [clips] [dropout: 1|0]"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

workload = sys.argv[1] if len(sys.argv) > 1 else "shufflenetv2"
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dropout = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(io.StringIO()):
    cfg, model, _, _ = bench.build(workload, dev)
xs = bench.synthetic_clips(cfg, clips, dev, 11)
labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (clips,), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
step, flat, opt = bench.make_train_step(model, xs, labels, overlap_allreduce=True, lr=0.02)
if os.environ.get("DBG_LR0") == "1":
    for gparam in opt.param_groups:
        gparam["lr"] = 0.0
if not dropout:
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.synchronize()
named = [("p/" + k, v) for k, v in model.named_parameters()] + [("b/" + k, v) for k, v in model.named_buffers()]
for i, p in enumerate(model.parameters()):
    named.append(("m/%d" % i, opt.state[p]["momentum_buffer"]))
snap = [v.detach().clone() for _, v in named]


def restore():
    with torch.no_grad():
        for (_, v), s in zip(named, snap):
            v.copy_(s)
    torch.cuda.manual_seed(4242)
    torch.cuda.synchronize()


def grab():
    torch.cuda.synchronize()
    return [v.detach().clone() for _, v in named] + [flat.flat.detach().clone()]


restore()
eager = []
with torch.cuda.stream(side):
    for _ in range(3):
        l = step()
        torch.cuda.synchronize()
        eager.append((float(l), grab(), torch.cuda.get_rng_state()[-16:].tolist()))
restore()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    sl = step()
restore()
for i in range(3):
    with torch.cuda.stream(side):
        g.replay()
    torch.cuda.synchronize()
    got = grab()
    names = [n for n, _ in named] + ["flat_grad"]
    bad = [n for n, a, b in zip(names, eager[i][1], got) if not torch.equal(a, b)]
    print("step %d: loss eager %.6f replay %.6f; rng tail eager %s replay %s; %d of %d tensors differ: %s" % (
        i, eager[i][0], float(sl), eager[i][2], torch.cuda.get_rng_state()[-16:].tolist(), len(bad), len(names), bad[:12]))
