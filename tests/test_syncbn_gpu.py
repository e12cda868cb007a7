"""NaiveSyncBatchNorm3d on the HIP path: two processes, each with half of the batch, against
the reference run as two ranks (tests/golden/dual_r50_syncbn_s64.npz, make_golden_sync.py)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_sync_batchnorm_two_ranks_match_reference():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import socket
    out = tempfile.mkdtemp()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))  # a free port per run: a lingering worker of an earlier run cannot block the rendezvous
    port = str(sock.getsockname()[1])
    sock.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_syncbn_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    rep_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
    for r in range(2):
        with open(os.path.join(out, "rank%d.json" % r)) as f:
            rep = json.load(f)
        try:
            os.makedirs(rep_dir, exist_ok=True)
            with open(os.path.join(rep_dir, "models_report.txt"), "a") as f:
                f.write("syncbn rank%d logits %.3e loss %.3e worst-grad %.3e worst-buffer %.3e\n" % (
                    r, rep["logits"], rep["loss"], max(rep["grads"].values()), max(rep["buffers"].values())))
        except OSError:
            pass
        assert rep["logits"] < 1e-3 and rep["loss"] < 1e-3, rep
        assert len(rep["grads"]) >= 6 and max(rep["grads"].values()) < 8e-2, rep["grads"]
        assert all(v < 0.3 for v in rep["scalar_grads"].values()), rep["scalar_grads"]
        assert len(rep["buffers"]) >= 8 and max(rep["buffers"].values()) < 1e-4, rep["buffers"]
