"""RESNET.TRANS_FUNC basic_transform (reference resnet_helper.py:25-107).  The reference's ResBlock cannot build it
(it passes dilation=, :338-349), but the class instantiates on its own: eval-mode parity against the vector the
reference class produced (tests/golden/op_vectors.npz, make_golden.py::op_vectors), and the training forward /
backward of the HIP module against autograd through the oracle restatement (oracle.basic_transform)."""
import json
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, rel_err, seeded_state_dict

pytestmark = pytest.mark.gpu
ARGS = {"basic_transform_s2": (64, 128, 3, 2), "basic_transform_s1": (32, 32, 1, 1)}


@pytest.mark.parametrize("name", sorted(ARGS))
def test_basic_transform_matches_reference_and_oracle_autograd(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sfhip
    from oracle import slowfast_oracle as oracle
    from slowfast.models import engine
    from slowfast.models.resnet_helper import BasicTransform
    z = np.load(os.path.join(GOLDEN, "op_vectors.npz"))
    sp = json.loads(str(z["specs"]))[name]
    sd = seeded_state_dict(sp["keys"], sp["key_shapes"], sp["seed"])
    m = BasicTransform(*ARGS[name]).cuda()
    assert list(m.state_dict().keys()) == sp["keys"]
    m.load_state_dict(sd)
    x = torch.from_numpy(np.random.RandomState(sp["seed"] + 1000).standard_normal(sp["shapes"][0]).astype(np.float32))
    # ---- eval: BN folded into the conv epilogues, against the reference class's own output
    m.eval()
    with torch.no_grad():
        y = sfhip.to_ncthw(m(sfhip.from_ncthw(x.cuda()))).cpu().numpy()
    assert rel_err(y, z[name + "/out0"]) < 1e-3
    # ---- training forward + backward against autograd through the oracle
    kt, stride = ARGS[name][2], ARGS[name][3]
    sdr = {"m." + k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v)
           for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = oracle.basic_transform(sdr, "m", xr, kt, stride, True)
    G = torch.from_numpy(np.random.RandomState(5).standard_normal(tuple(yr.shape)).astype(np.float32))
    (yr * G).sum().backward()
    m.train()
    t = engine.Tape()
    with torch.no_grad(), engine.taping(t):
        a = sfhip.from_ncthw(x.cuda())
        out = m(a)
        yt = sfhip.to_ncthw(out).cpu()
        g = t.grad_of(out)
        g.buf[..., g.coff:g.coff + g.C].permute(0, 4, 1, 2, 3).copy_(G.cuda())
        t.backward_ops = list(t.ops)
        for fn, side in reversed(t.ops):
            fn()
        gx = sfhip.to_ncthw(sfhip.Act(t.gbuf[a.buf.data_ptr()].view(a.buf.shape))).cpu()
    torch.cuda.synchronize()
    assert rel_err(yt.numpy(), yr.detach().numpy()) < 1e-4
    assert rel_err(gx.numpy(), xr.grad.numpy()) < 2e-4
    for pn, p in m.named_parameters():
        got = t.pgrads[p].reshape(p.shape).cpu().numpy()
        assert rel_err(got, sdr["m." + pn].grad.numpy()) < 2e-4, pn
