"""The per-clip input step of the reference's Kinetics loader (datasets/kinetics.py:230-248) for the MI355X path.

Host side (this file) keeps what must stay on the host: the random draws of `spatial_sampling`
(datasets/utils.py:151-203 -> transform.py) taken from numpy's global RNG in the reference's order, and the slow
pathway's frame indices (`pack_pathway_output`, datasets/utils.py:73-112).  The arithmetic — uint8 -> float
normalise (`tensor_normalize`, :298-315), bilinear short-side scale, crop, flip, frame selection — runs as one HIP
kernel per pathway (`sf_clip_prologue`) that writes the stems' input layout directly, so the decoded clip crosses
PCIe as uint8 once instead of as two float NCTHW tensors."""
from slowfast._overlay import chain_module as _chain_module

_chain_module(globals())  # the reference's namesake (when importable) supplies every name not defined below
import collections
import math

import numpy as np
import torch

import sfhip

SpatialParams = collections.namedtuple("SpatialParams", "new_h new_w y x flip crop")
PathwayCfg = collections.namedtuple("PathwayCfg", "alpha reverse_input_channel")


def _short_side(h, w, min_size, max_size, inverse_uniform_sampling):
    """transform.py:305-327 (random_short_side_scale_jitter): the scaled (height, width)."""
    if inverse_uniform_sampling:
        size = int(round(1.0 / np.random.uniform(1.0 / max_size, 1.0 / min_size)))
    else:
        size = int(round(np.random.uniform(min_size, max_size)))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(math.floor((float(h) / w) * size)), size
    return size, int(math.floor((float(w) / h) * size))


def sample_spatial_params(height, width, spatial_idx=-1, min_scale=256, max_scale=320, crop_size=224,
                          random_horizontal_flip=True, inverse_uniform_sampling=False):
    """The decisions of `spatial_sampling` for one clip of height x width frames, without touching pixels.
    spatial_idx -1: random scale / crop / flip (training); 0, 1, 2: left|top, center, right|bottom uniform crop."""
    assert spatial_idx in [-1, 0, 1, 2]
    if spatial_idx == -1:
        nh, nw = _short_side(height, width, min_scale, max_scale, inverse_uniform_sampling)
        y = x = 0
        if not (nh == crop_size and nw == crop_size):  # transform.py:374-382 (random_crop)
            if nh > crop_size:
                y = int(np.random.randint(0, nh - crop_size))
            if nw > crop_size:
                x = int(np.random.randint(0, nw - crop_size))
        flip = bool(np.random.uniform() < 0.5) if random_horizontal_flip else False  # transform.py:417
        return SpatialParams(nh, nw, y, x, flip, crop_size)
    assert len({min_scale, max_scale, crop_size}) == 1
    nh, nw = _short_side(height, width, min_scale, max_scale, False)
    y = int(math.ceil((nh - crop_size) / 2))  # transform.py:446-458 (uniform_crop)
    x = int(math.ceil((nw - crop_size) / 2))
    if nh > nw:
        y = 0 if spatial_idx == 0 else (nh - crop_size if spatial_idx == 2 else y)
    else:
        x = 0 if spatial_idx == 0 else (nw - crop_size if spatial_idx == 2 else x)
    return SpatialParams(nh, nw, y, x, False, crop_size)


def slow_frame_indices(num_frames, alpha):
    """torch.linspace(0, T-1, T//alpha).long() (datasets/utils.py:96-104): T=32, alpha=4 -> 0,4,8,13,17,22,26,31."""
    return torch.linspace(0, num_frames - 1, num_frames // alpha).long()


def pack_pathway_output(cfg, frames):
    """[slow, fast] from C x T x H x W frames (any device).  `cfg`: a CfgNode (DATA.REVERSE_INPUT_CHANNEL,
    SLOWFAST.ALPHA, MODEL.ARCH ...) or a PathwayCfg."""
    if isinstance(cfg, PathwayCfg):
        alpha, reverse, single = cfg.alpha, cfg.reverse_input_channel, False
    else:
        alpha, reverse = cfg.SLOWFAST.ALPHA, cfg.DATA.REVERSE_INPUT_CHANNEL
        single = cfg.MODEL.ARCH in cfg.MODEL.SINGLE_PATHWAY_ARCH
        if not single and cfg.MODEL.ARCH not in cfg.MODEL.MULTI_PATHWAY_ARCH:
            raise NotImplementedError("Model arch {} is not in {}".format(
                cfg.MODEL.ARCH, cfg.MODEL.SINGLE_PATHWAY_ARCH + cfg.MODEL.MULTI_PATHWAY_ARCH))
    if reverse:
        frames = frames[[2, 1, 0], :, :, :]
    if single:
        return [frames]
    idx = slow_frame_indices(frames.shape[1], alpha).to(frames.device)
    return [torch.index_select(frames, 1, idx), frames]


def tensor_normalize(tensor, mean, std):
    """(uint8 -> float / 255) - mean, / std on the tensor's device (datasets/utils.py:298-315)."""
    if tensor.dtype == torch.uint8:
        tensor = tensor.float() / 255.0
    mean = torch.tensor(mean, device=tensor.device) if isinstance(mean, (list, tuple)) else mean
    std = torch.tensor(std, device=tensor.device) if isinstance(std, (list, tuple)) else std
    return (tensor - mean) / std


def gpu_input_step(clips, params, mean, std, alpha, reverse_input_channel=False, pad=(0, 0), wp=None):
    """The whole input step for a batch on the GPU.

    clips:  list of B decoded clips, uint8 [T, H_i, W_i, 3] device tensors (sizes may differ between clips)
    params: list of B SpatialParams (same crop for all)
    pad/wp: the stem's (ph, pw) and row pitch, from `engine.stem_geometry(first_conv, crop, crop)`
    Returns [slow, fast] as sfhip.PackedClip — pass them to the model in place of the NCTHW tensors."""
    B = len(clips)
    assert B == len(params) and B > 0
    crop = params[0].crop
    T = clips[0].shape[0]
    ph, pw = pad
    wp = crop + 2 * pw if wp is None else wp
    dev = clips[0].device
    idx = slow_frame_indices(T, alpha).to(device=dev, dtype=torch.int32)
    slow = torch.empty((B, idx.numel(), crop + 2 * ph, wp, 4), dtype=torch.float32, device=dev)
    fast = torch.empty((B, T, crop + 2 * ph, wp, 4), dtype=torch.float32, device=dev)
    for b, (clip, p) in enumerate(zip(clips, params)):
        assert clip.shape[0] == T and p.crop == crop
        for dst, fi in ((slow[b], idx), (fast[b], None)):
            sfhip.clip_prologue(clip, dst, (p.new_h, p.new_w), (p.y, p.x), crop, p.flip, mean, std, frame_idx=fi,
                                reverse=reverse_input_channel, ph=ph, pw=pw)
    return [sfhip.PackedClip(slow, 3, crop, crop, ph, pw), sfhip.PackedClip(fast, 3, crop, crop, ph, pw)]


def stem_input_geometry(model, crop):
    """(ph, pw, Wp) shared by the model's pathway stems (their first convolutions have the same H/W kernel,
    stride and padding) for crop x crop frames."""
    from slowfast.models import engine
    geos = set()
    stem = next(model.children())  # s1 (s0 for GhostNet): one `pathway{p}_stem` per pathway
    for m in stem.modules():
        if isinstance(m, torch.nn.Conv3d) and m.in_channels <= 4 and m.groups == 1:
            geos.add(engine.stem_geometry(m, crop, crop))
    if len(geos) != 1:
        raise ValueError("pathway stems disagree on the input geometry: %s" % sorted(geos))
    return geos.pop()
