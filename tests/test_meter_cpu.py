"""TestMeter (multi-view test-time ensemble) against the REFERENCE's own utils/meters.py::TestMeter run in the build
container (tests/golden/make_golden_meter.py -> test_meter.npz): per-video sum / max ensembles, labels, clip counts
and top-1 accuracy; top-5 against the oracle restatement (the reference's utils/metrics.py:40 raises for k > 1 on
torch >= 1.8 — recorded in the fixture's meta)."""
import json
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN
from oracle import input_oracle


@pytest.mark.parametrize("case", ["k100_like", "tiny"])
@pytest.mark.parametrize("method", ["sum", "max"])
def test_test_meter_matches_reference(case, method):
    from slowfast.utils.meters import TestMeter
    z = np.load(os.path.join(GOLDEN, "test_meter.npz"))
    meta = json.loads(str(z["meta"]))[case]
    preds, labels, ids = z[case + "/preds"], z[case + "/labels"], z[case + "/clip_ids"]
    nv, nc, ncls, bs = meta["num_videos"], meta["num_clips"], meta["num_cls"], meta["batch"]
    m = TestMeter(nv, nc, ncls, (len(ids) + bs - 1) // bs, ensemble_method=method)
    for s in range(0, len(ids), bs):
        m.update_stats(torch.from_numpy(preds[s:s + bs]), torch.from_numpy(labels[s:s + bs]),
                       torch.from_numpy(ids[s:s + bs]))
    ref_vp = z["%s/%s/video_preds" % (case, method)]
    # 'sum' adds the clips of a video in loader order on both sides; index_add_ may re-associate within a batch
    assert np.abs(m.video_preds.numpy() - ref_vp).max() <= 1e-5 * np.abs(ref_vp).max()
    assert np.array_equal(m.video_labels.numpy(), z["%s/%s/video_labels" % (case, method)])
    assert np.array_equal(m.clip_count.numpy(), z["%s/%s/clip_count" % (case, method)])
    stats = m.finalize_metrics(ks=(1, 5))
    assert stats["top1_acc"] == meta[method]["top1_acc"]
    assert stats["complete"] is True
    # the oracle restatement agrees with the reference too (it is the GPU test's comparator)
    vp, vl, cnt, top = input_oracle.test_meter_ensemble(preds, labels, ids, nv, nc, method)
    assert np.abs(vp - ref_vp).max() <= 1e-5 * np.abs(ref_vp).max()
    assert "%.2f" % top[1] == meta[method]["top1_acc"]
    assert "%.2f" % top[5] == stats["top5_acc"]
    assert 0.0 < float(stats["top1_acc"]) < 100.0 or case == "tiny"
