"""Head-side test-time ensemble (reference utils/meters.py:214-373 TestMeter, utils/metrics.py:9-42).

The reference copies every batch of predictions to the host and adds them to per-video rows one clip at a time in a
Python loop.  Here the accumulators live on the model's device and one `index_add_` / `scatter_reduce_` per batch
performs the ensemble, so the test loop never synchronises with the GPU until `finalize_metrics`."""
from slowfast._overlay import chain_module as _chain_module

_chain_module(globals())  # the reference's namesake (when importable) supplies every name not defined below
import datetime
import json
import logging
import time

import torch

_LOG = logging.getLogger(__name__)


def topks_correct(preds, labels, ks):
    """Number of samples whose label is within the top-k predictions, for each k (utils/metrics.py:9-42)."""
    assert preds.size(0) == labels.size(0), "Batch dim of predictions and labels must match"
    _, inds = torch.topk(preds, max(ks), dim=1, largest=True, sorted=True)
    correct = inds.t().eq(labels.view(1, -1).expand(max(ks), -1))
    return [correct[:k, :].reshape(-1).float().sum() for k in ks]


class TestMeter(object):
    """Sum / max ensemble of `num_clips` clip predictions per video and the final top-k accuracies."""

    __test__ = False  # not a pytest class

    def __init__(self, num_videos, num_clips, num_cls, overall_iters, multi_label=False, ensemble_method="sum",
                 device="cpu"):
        if ensemble_method not in ("sum", "max"):
            raise NotImplementedError("Ensemble Method {} is not supported".format(ensemble_method))
        self.num_clips = num_clips
        self.overall_iters = overall_iters
        self._tic, self._lap = time.perf_counter(), 0.0
        self.multi_label = multi_label
        self.ensemble_method = ensemble_method
        self.video_preds = torch.zeros((num_videos, num_cls), device=device)
        self.video_labels = (torch.zeros((num_videos, num_cls), device=device) if multi_label
                             else torch.zeros((num_videos,), dtype=torch.long, device=device))
        self.clip_count = torch.zeros((num_videos,), dtype=torch.long, device=device)
        self.reset()

    def reset(self):
        self.clip_count.zero_()
        self.video_preds.zero_()
        if self.multi_label:
            self.video_preds -= 1e10
        self.video_labels.zero_()

    def update_stats(self, preds, labels, clip_ids):
        """preds [N, C], labels [N] (or [N, C] multi-label), clip_ids [N]: video = clip_id // num_clips."""
        dev = self.video_preds.device
        preds, labels = preds.detach().to(dev), labels.to(dev)
        vid = torch.div(clip_ids.to(dev).long(), self.num_clips, rounding_mode="floor")
        self.video_labels[vid] = labels.to(self.video_labels.dtype)
        if self.ensemble_method == "sum":
            self.video_preds.index_add_(0, vid, preds.to(self.video_preds.dtype))
        else:
            self.video_preds.scatter_reduce_(0, vid.view(-1, 1).expand_as(preds), preds.to(self.video_preds.dtype),
                                             reduce="amax", include_self=True)
        self.clip_count.index_add_(0, vid, torch.ones_like(vid))

    def iter_tic(self):
        """Start timing one test iteration (utils/meters.py:337-338)."""
        self._tic = time.perf_counter()

    def iter_toc(self):
        """Stop timing; the lap feeds `log_iter_stats` (utils/meters.py:340-341)."""
        self._lap = time.perf_counter() - self._tic

    def log_iter_stats(self, cur_iter):
        """One `json_stats:` line per iteration with the remaining-time estimate (utils/meters.py:319-335)."""
        eta = str(datetime.timedelta(seconds=int(self._lap * (self.overall_iters - cur_iter))))
        stats = {"split": "test_iter", "cur_iter": "{}".format(cur_iter + 1), "eta": eta, "time_diff": self._lap}
        _LOG.info("json_stats: {}".format(json.dumps(stats, sort_keys=True)))
        return stats

    def finalize_metrics(self, ks=(1, 5)):
        """Returns {"split": "test_final", "top{k}_acc": "xx.xx", ...} (multi-label mAP is not on this path)."""
        if self.multi_label:
            raise NotImplementedError("multi-label mAP (AVA/Charades) is outside the hot-path scope")
        stats = {"split": "test_final", "complete": bool((self.clip_count == self.num_clips).all())}
        correct = topks_correct(self.video_preds, self.video_labels, ks)
        for k, c in zip(ks, correct):
            stats["top{}_acc".format(k)] = "{:.{prec}f}".format(float(c) / self.video_preds.size(0) * 100.0, prec=2)
        return stats
