"""Deterministic, de-degenerated parameter + input generator shared by the golden-vector
generator (reference side) and the tests (oracle / HIP side).

Why not the models' own fresh init: SURVEY.md §8(c) — ZERO_INIT_FINAL_BN zeroes every
bottleneck's last BN (each ResBlock collapses to relu(shortcut)), SpatialAttention.gamma is
0 (attention contributes exactly 0) and BN running stats are (0, 1).  A kernel that skipped
the bottleneck or the attention would still "pass".  Here every tensor of the state_dict is
filled from a numpy RandomState keyed on (seed, crc32(key)) so both sides build bit-identical
parameters without shipping a 136 MB state_dict, and the fill keeps activations O(1).
"""
import zlib

import numpy as np


def _rs(seed, key):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def fill_array(key, shape, seed):
    """float32 array for state_dict entry `key` of `shape` (or None to leave untouched)."""
    rs = _rs(seed, key)
    shape = tuple(int(s) for s in shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return None
    if leaf == "running_mean":
        return rs.uniform(-0.2, 0.2, shape).astype(np.float32)
    if leaf == "running_var":
        return rs.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "gamma":  # SpatialAttention residual gate (wdf_attention_helper.py:30)
        return np.full(shape, 0.5, np.float32)
    if len(shape) == 1:
        if leaf == "weight" and "_nonlocal" in key:  # Nonlocal's final BN: keep x + BN(...) from growing block by block
            return rs.uniform(0.1, 0.3, shape).astype(np.float32)
        if leaf == "weight":  # BN affine weight
            return rs.uniform(0.5, 1.0, shape).astype(np.float32)
        return rs.uniform(-0.1, 0.1, shape).astype(np.float32)  # any bias
    fan_in = int(np.prod(shape[1:]))
    if "attention_channel_f2s.conv" in key:  # ECA Conv1d(1,1,3) weight
        return rs.uniform(-0.8, 0.8, shape).astype(np.float32)
    gain = 2.0
    if "query_conv" in key or "key_conv" in key:
        gain = 0.5  # keep the un-scaled logits O(1): softmax must not be one-hot
    if "conv_theta" in key or "conv_phi" in key:
        gain = 0.02  # Nonlocal: theta^T phi sums d = 256 / 512 products of post-ReLU (non-zero-mean) features
    if "_nonlocal" in key and ("conv_g" in key or "conv_out" in key):
        gain = 0.5
    if leaf == "weight" and len(shape) == 2:  # Linear
        gain = 1.0
    return (rs.standard_normal(shape) * np.sqrt(gain / fan_in)).astype(np.float32)


def fill_state_dict(sd, seed):
    """In-place fill of a torch state_dict-like mapping {key: tensor}."""
    import torch

    for key, t in sd.items():
        a = fill_array(key, t.shape, seed)
        if a is not None:
            with torch.no_grad():
                t.copy_(torch.from_numpy(a).to(t.dtype))
    return sd


def make_clip(seed, batch, t_fast, alpha, size, channels=3):
    """[slow, fast] NCTHW float32 numpy clips; slow = pack_pathway_output's index_select
    (datasets/utils.py:93-104): frames linspace(0, T-1, T//alpha).long()."""
    rs = np.random.RandomState(seed)
    fast = rs.standard_normal((batch, channels, t_fast, size, size)).astype(np.float32)
    idx = np.linspace(0, t_fast - 1, t_fast // alpha).astype(np.int64)  # trunc == .long()
    slow = np.ascontiguousarray(fast[:, :, idx])
    return slow, fast


def sample_activation(a, n=2048):
    """Strided digest of a big activation: (flat strided sample, absmax, mean).  The stride is made ODD: every
    tensor dimension here is a power of two times 7 or 3, and an even stride would only ever visit w = 0 and a few h
    (a [2,64,8,16,16] tensor sampled at stride 64 sees one column) — an odd stride walks through all of them."""
    flat = np.asarray(a, dtype=np.float32).reshape(-1)
    step = max(1, flat.size // n) | 1
    return flat[::step][:n].copy(), float(np.abs(flat).max()), float(flat.mean())


def make_upstream(seed, shape):
    """Seeded upstream gradient dL/d(output) of a top-level child (stage-wise gradient fixtures): both the
    reference side (make_golden.py) and the HIP side regenerate it instead of shipping activation-sized tensors."""
    return np.random.RandomState(int(seed) & 0x7FFFFFFF).standard_normal(tuple(int(s) for s in shape)).astype(np.float32)


def upstream_seed(child_index, out_index):
    return 90001 + 131 * int(child_index) + int(out_index)
