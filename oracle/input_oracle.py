"""ORACLE — test infrastructure, NOT product code.

CPU restatement of the reference's per-clip input step, in the order
datasets/kinetics.py:230-248 applies it:  tensor_normalize (datasets/utils.py:298-315) -> permute THWC->CTHW ->
spatial_sampling (datasets/utils.py:151-203: short-side scale jitter, crop, flip — transform.py:283-337, 359-393,
395-423, 425-468) -> pack_pathway_output (datasets/utils.py:73-112).  Random choices come from numpy's global RNG
in the reference's order (scale; y offset; x offset; flip), so seeding it reproduces the reference's draw.

Pinned by tests/golden/input_step.npz (generated from the reference's own functions by make_golden_input.py).
Only tests/ may import this module."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def tensor_normalize(tensor, mean, std):
    if tensor.dtype == torch.uint8:
        tensor = tensor.float() / 255.0
    return (tensor - torch.tensor(mean)) / torch.tensor(std)


def short_side_scale(images, min_size, max_size, inverse_uniform_sampling=False):
    """images [C,T,H,W]; returns (scaled, (new_h, new_w))."""
    if inverse_uniform_sampling:
        size = int(round(1.0 / np.random.uniform(1.0 / max_size, 1.0 / min_size)))
    else:
        size = int(round(np.random.uniform(min_size, max_size)))
    h, w = images.shape[2], images.shape[3]
    if (w <= h and w == size) or (h <= w and h == size):
        return images, (h, w)
    nh = nw = size
    if w < h:
        nh = int(math.floor((float(h) / w) * size))
    else:
        nw = int(math.floor((float(w) / h) * size))
    return F.interpolate(images, size=(nh, nw), mode="bilinear", align_corners=False), (nh, nw)


def spatial_sampling(frames, spatial_idx, min_scale, max_scale, crop_size, random_horizontal_flip=True,
                     inverse_uniform_sampling=False):
    """frames [C,T,H,W] float; returns (frames, params dict)."""
    assert spatial_idx in (-1, 0, 1, 2)
    p = {}
    if spatial_idx == -1:
        frames, (nh, nw) = short_side_scale(frames, min_scale, max_scale, inverse_uniform_sampling)
        y = x = 0
        if not (nh == crop_size and nw == crop_size):
            if nh > crop_size:
                y = int(np.random.randint(0, nh - crop_size))
            if nw > crop_size:
                x = int(np.random.randint(0, nw - crop_size))
        frames = frames[:, :, y:y + crop_size, x:x + crop_size]
        flip = False
        if random_horizontal_flip:
            flip = bool(np.random.uniform() < 0.5)
            if flip:
                frames = frames.flip(-1)
    else:
        assert len({min_scale, max_scale, crop_size}) == 1
        frames, (nh, nw) = short_side_scale(frames, min_scale, max_scale)
        y = int(math.ceil((nh - crop_size) / 2))
        x = int(math.ceil((nw - crop_size) / 2))
        if nh > nw:
            y = 0 if spatial_idx == 0 else (nh - crop_size if spatial_idx == 2 else y)
        else:
            x = 0 if spatial_idx == 0 else (nw - crop_size if spatial_idx == 2 else x)
        frames = frames[:, :, y:y + crop_size, x:x + crop_size]
        flip = False
    p.update(new_h=nh, new_w=nw, y=y, x=x, flip=flip)
    return frames, p


def slow_indices(t, alpha):
    return torch.linspace(0, t - 1, t // alpha).long()


def pack_pathway_output(frames, alpha, reverse_input_channel=False):
    if reverse_input_channel:
        frames = frames[[2, 1, 0]]
    return [torch.index_select(frames, 1, slow_indices(frames.shape[1], alpha)), frames]


def input_step(clip_u8, mean, std, spatial_idx, min_scale, max_scale, crop_size, flip, inverse_uniform, alpha,
               reverse):
    frames = tensor_normalize(torch.from_numpy(clip_u8), mean, std).permute(3, 0, 1, 2)
    frames, params = spatial_sampling(frames, spatial_idx, min_scale, max_scale, crop_size, flip, inverse_uniform)
    return pack_pathway_output(frames, alpha, reverse), params


def test_meter_ensemble(preds, labels, clip_ids, num_videos, num_clips, method="sum"):
    """TestMeter.update_stats' per-clip loop (utils/meters.py:277-312) + finalize top-k (:351-366,
    utils/metrics.py:9-42) restated with numpy.  Pinned by tests/test_meter_cpu.py against the reference's own TestMeter
    run in the build container behind import stand-ins (tests/golden/make_golden_meter.py -> test_meter.npz):
    ensembles, labels, counts and top-1; the reference's own top-5 raises on torch >= 1.8 (metrics.py:40 .view(-1))."""
    vp = np.zeros((num_videos, preds.shape[1]), np.float32)
    vl = np.zeros((num_videos,), np.int64)
    cnt = np.zeros((num_videos,), np.int64)
    for i in range(preds.shape[0]):
        v = int(clip_ids[i]) // num_clips
        vl[v] = labels[i]
        vp[v] = vp[v] + preds[i] if method == "sum" else np.maximum(vp[v], preds[i])
        cnt[v] += 1
    order = np.argsort(-vp, axis=1, kind="stable")
    top = {k: float((order[:, :k] == vl[:, None]).any(1).sum()) / num_videos * 100.0 for k in (1, 5)}
    return vp, vl, cnt, top
