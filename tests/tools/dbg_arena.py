import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import numpy as np, torch
import test_models_gpu as T
from slowfast.models import engine
name = "shufflenetv2_cfg1"
z, meta = T.load_case(name)
model, sd = T._build(meta, z)
print("arena hint before:", model.__dict__.get("_sf_arena_floats"))
model.train()
orig = engine.Tape._open_arena
def dbg(self):
    orig(self)
    print("open_arena: model", type(self.model).__name__, "want", self.model.__dict__.get("_sf_arena_floats"), "arena", None if self.arena is None else self.arena.numel())
engine.Tape._open_arena = dbg
for it in range(2):
    model.zero_grad(set_to_none=True)
    logits = model([x.cuda() for x in T.case_inputs(meta)])
    labels = torch.from_numpy(z["train/labels"]).cuda()
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    g = dict(model.named_parameters())["s1.pathway0_stem.0.weight"].grad
    print(it, "loss", loss.item(), "stem grad norm", float(g.norm()), "hint", model.__dict__.get("_sf_arena_floats"))
