#!/usr/bin/env python3
"""Who makes device-to-device copies in a training step?  Counts Tensor.copy_ / clone / contiguous (when it copies) /
torch.cat / index_select calls by caller line over 3 bench steps after warm-up (each is a launch on the step's path).
usage: python tools/count_copies.py"""
import collections
import os
import sys
import traceback

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "efficient-slowfast_amd")]
import torch  # noqa: E402

counts = collections.Counter()
ON = [False]


def wrap(obj, name, tag, pred=None):
    orig = getattr(obj, name)

    def f(*a, **k):
        if ON[0] and (pred is None or pred(*a, **k)):
            fr = traceback.extract_stack(limit=3)[0]
            counts[(tag, os.path.basename(fr.filename), fr.lineno)] += 1
        return orig(*a, **k)

    setattr(obj, name, f)


wrap(torch.Tensor, "copy_", "copy_")
wrap(torch.Tensor, "clone", "clone")
wrap(torch.Tensor, "contiguous", "contiguous(copying)", lambda t, *a, **k: not t.is_contiguous())
wrap(torch, "cat", "cat")
wrap(torch, "index_select", "index_select")
wrap(torch.Tensor, "float", "float()", lambda t, *a, **k: t.dtype != torch.float32)
wrap(torch.Tensor, "double", "double()")
import bench  # noqa: E402

orig_sync = torch.cuda.synchronize
nsync = [0]


def sync(*a, **k):  # bench brackets its timed region with synchronize: count only inside it
    nsync[0] += 1
    ON[0] = nsync[0] >= 2
    return orig_sync(*a, **k)


torch.cuda.synchronize = sync
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-extras"]
bench.main()
for k, v in counts.most_common(30):
    print(v, k)
