#!/usr/bin/env python3
"""Golden vectors for the test-time multi-view ensemble: the REFERENCE's own
slowfast/utils/meters.py::TestMeter (:216-373) and utils/metrics.py::topks_correct (:9-42), imported in the build
container.  meters.py pulls in AVA tooling at import time (datasets/__init__ -> cv2, ava_eval_helper -> the
ava_evaluation package); none of it is touched by TestMeter, so `slowfast.datasets` is registered as a path-only
package (its utils.py loads, its __init__ does not run) and cv2 / torchvision / ava_eval_helper are inert stand-ins.
Writes tests/golden/test_meter.npz: seeded clip predictions / labels / clip ids fed in shuffled batches, and the
reference's per-video ensemble, clip counts and top-k accuracies for the 'sum' and 'max' methods."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True


def load_reference_meters():
    import _refimport
    _refimport.import_reference()
    for name in ("cv2", "torchvision", "torchvision.transforms", "slowfast.utils.ava_eval_helper"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["slowfast.utils.ava_eval_helper"].__dict__.update(
        evaluate_ava=None, read_csv=None, read_exclusions=None, read_labelmap=None)
    pkg = types.ModuleType("slowfast.datasets")
    pkg.__path__ = [_refimport.REF_ROOT + "/SlowFast/slowfast/datasets"]
    sys.modules["slowfast.datasets"] = pkg
    import slowfast.utils.meters as meters
    return meters


def main():
    meters = load_reference_meters()
    import slowfast.utils.logging as logging
    out, meta = {}, {}
    for case, (nv, nc, ncls, bs, seed) in {"k100_like": (37, 10, 100, 8, 1), "tiny": (5, 3, 7, 4, 2)}.items():
        rs = np.random.RandomState(seed)
        n = nv * nc
        labels_v = rs.randint(0, ncls, nv)
        preds = rs.standard_normal((n, ncls)).astype(np.float32)
        # make the right class likely enough that top-1 / top-5 are neither 0 nor 100
        preds[np.arange(n), labels_v[np.arange(n) // nc]] += 0.45
        ids = rs.permutation(n)  # loader order: shuffled clips, ragged last batch
        out[case + "/preds"], out[case + "/clip_ids"] = preds[ids], ids.astype(np.int64)
        out[case + "/labels"] = labels_v[ids // nc].astype(np.int64)
        meta[case] = dict(num_videos=nv, num_clips=nc, num_cls=ncls, batch=bs)
        for method in ("sum", "max"):
            m = meters.TestMeter(nv, nc, ncls, (n + bs - 1) // bs, ensemble_method=method)
            for s in range(0, n, bs):
                m.update_stats(torch.from_numpy(out[case + "/preds"][s:s + bs]),
                               torch.from_numpy(out[case + "/labels"][s:s + bs]),
                               torch.from_numpy(out[case + "/clip_ids"][s:s + bs]))
            logged = []
            orig = logging.log_json_stats
            logging.log_json_stats = lambda stats: logged.append(dict(stats))
            try:
                try:
                    m.finalize_metrics(ks=(1, 5))
                except RuntimeError as e:
                    # torch >= 1.8: utils/metrics.py:40 calls .view(-1) on a transposed slice, which raises for k > 1
                    # ("view size is not compatible ..."); the reference's own top-1 still runs
                    meta[case].setdefault("reference_top5_raises", str(e)[:60])
                    m.finalize_metrics(ks=(1,))
            finally:
                logging.log_json_stats = orig
            out["%s/%s/video_preds" % (case, method)] = m.video_preds.numpy()
            out["%s/%s/video_labels" % (case, method)] = m.video_labels.numpy()
            out["%s/%s/clip_count" % (case, method)] = m.clip_count.numpy()
            meta[case][method] = logged[-1]
    out["meta"] = json.dumps(meta)
    path = os.path.join(HERE, "test_meter.npz")
    np.savez_compressed(path, **out)
    print("test_meter %.1f KB" % (os.path.getsize(path) / 1024), json.dumps(meta)[:300])


if __name__ == "__main__":
    main()
