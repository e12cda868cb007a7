"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol include/sfhip.h
declares (no compute calls without a GPU), and the product path refuses to run without a GPU."""
import os
import re

import pytest
import torch


def _header_symbols(root):
    txt = open(os.path.join(root, "include", "sfhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(repo_root):
    import sfhip
    if not os.path.exists(sfhip.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    L = sfhip.lib()
    syms = _header_symbols(repo_root)
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(L, s), "libsfhip.so does not export %s" % s
    assert sorted(sfhip.EXPORTS) == syms
    assert L.sf_abi_version() == 1
    assert L.sf_build_arch() == b"gfx950"


def test_descriptor_structs_match_header(repo_root):
    """ctypes mirrors of sf_conv_desc / sf_pool_desc list the header's fields in the header's order."""
    import sfhip
    txt = open(os.path.join(repo_root, "include", "sfhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for name, cls in (("sf_conv_desc", sfhip.ConvDesc), ("sf_pool_desc", sfhip.PoolDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), txt, re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl.startswith("int "):
                fields += [f.strip() for f in decl[4:].split(",")]
        assert fields == [f[0] for f in cls._fields_], name


def test_no_cpu_fallback():
    import sfhip
    with pytest.raises(sfhip.SfhipError):
        sfhip.from_ncthw(torch.zeros(1, 3, 2, 4, 4))
    a = sfhip.Act(torch.zeros(1, 2, 4, 4, 8))
    with pytest.raises(sfhip.SfhipError):
        sfhip.conv(a, torch.zeros(8, 1, 16), (1, 1, 1))


def test_sub_bn_aggregate_matches_oracle():
    """SubBatchNorm3d.aggregate_stats / utils.misc.aggregate_sub_bn_stats (host logic) vs the oracle's restatement
    of batchnorm_helper.py:66-95."""
    import torch
    from oracle import slowfast_oracle as oracle
    from slowfast.models.batchnorm_helper import SubBatchNorm3d
    from slowfast.utils.misc import aggregate_sub_bn_stats
    torch.manual_seed(3)
    holder = torch.nn.Sequential(SubBatchNorm3d(4, num_features=6, eps=1e-5, momentum=0.1), torch.nn.ReLU())
    sb = holder[0]
    sb.split_bn.running_mean.copy_(torch.randn(24))
    sb.split_bn.running_var.copy_(torch.rand(24) + 0.5)
    assert aggregate_sub_bn_stats(holder) == 1
    mean, var = oracle.sub_bn_aggregate(sb.split_bn.running_mean, sb.split_bn.running_var, 4)
    assert torch.allclose(sb.bn.running_mean, mean) and torch.allclose(sb.bn.running_var, var)
    assert list(sb.state_dict().keys()) == ["weight", "bias", "bn.running_mean", "bn.running_var",
                                            "bn.num_batches_tracked", "split_bn.running_mean",
                                            "split_bn.running_var", "split_bn.num_batches_tracked"]


def _conv_desc(n, t, h, w, cin, cout, k, stride=(1, 1, 1), cin_pad=None, in_cs=None, pad=None):
    import sfhip
    pad = tuple(kk // 2 for kk in k) if pad is None else pad
    od = [(i + 2 * p - kk) // s + 1 for i, p, kk, s in zip((t, h, w), pad, k, stride)]
    cin_pad = (cin + 15) // 16 * 16 if cin_pad is None else cin_pad
    return sfhip.ConvDesc(n, t, h, w, cin, in_cs or cin, 0, od[0], od[1], od[2], cout, cout, 0, 1, k[0], k[1], k[2],
                          stride[0], stride[1], stride[2], pad[0], pad[1], pad[2], 1, 1, 1, cin_pad, 0, 0, 0, 0)


def test_launch_planning_queries_are_host_only_and_fill_the_chip():
    """The workspace / split queries of the C ABI are pure host logic (callable without a GPU): the weight-gradient
    position splits land near the 768-workgroup target on whole rounds of 256, split-K workspaces appear exactly for
    the short-M / long-K layers, the Fast stem takes the one-workgroup-per-CU ring kernel, and the attention
    workspaces hold their planes / parts."""
    import ctypes
    import sfhip
    L = sfhip.lib()
    # res2 3x3 64->64 at 8x8x56x56.  Per-wavefront kernel (conv_wgrad_wave.hip): 9 (tap, 64-channel) column blocks,
    # two per wavefront -> 5 tiles, S chosen so that 5*S workgroups fill whole rounds of 256; LDS-tiled fallback
    # (sf_conv_tune(10, 0)): 9 tiles of 64x64, 9*S workgroups
    d = _conv_desc(8, 8, 56, 56, 64, 64, (1, 3, 3))
    S = L.sf_conv_wgrad_splits(ctypes.byref(d))
    wg = 5 * S
    assert 256 <= wg <= 1024 and wg / (-(-wg // 256) * 256) >= 0.94, (S, wg)
    L.sf_conv_tune(10, 0)
    try:
        S = L.sf_conv_wgrad_splits(ctypes.byref(d))
    finally:
        L.sf_conv_tune(10, 1)
    wg = 9 * S
    assert 512 <= wg <= 1024 and wg / (-(-wg // 256) * 256) >= 0.94, (S, wg)
    assert L.sf_conv_fwd_ws_floats(ctypes.byref(d)) == 0                       # 1568 tiles: no split-K
    # res4 3x1x1 1024->256 at M = 12544 (K = 3072) on the bf16-piece kernel (conv_bx.hip): 49 tiles of 256 x 256 share
    # their K steps between S workgroups that together fill the chip.  Default mode (SF_CONV_BX_AF32=1): the activation
    # operand travels as fp32 rows, so the workspace = the weight planes (unless handed in) + S partial tiles — no
    # activation planes whether or not the caller has them.  sf_conv_tune(7, 0): the per-wavefront kernel splits K inside
    # the workgroup (no workspace); the LDS-tiled fallback (sf_conv_tune(0, 0)): 196 tiles, 192 K steps -> split-K with a
    # [S][M][Cout] workspace.  res3's 1x3x3 128 -> 128 (196 tiles of 256 x 128 = 0.77 of a round): on conv_wave until
    # round 5 (97 TFLOP/s there against 87 here), on conv_bx since its loader state left scratch (112 in the step)
    d4 = _conv_desc(8, 8, 14, 14, 1024, 256, (3, 1, 1))
    w_planes = -(-(3 * (256 + 1) * 3072 // 2) // 4) * 4
    n_bx = L.sf_conv_bx_ws_floats(ctypes.byref(d4), 0, 1)
    S = (n_bx - 4) // (12544 * 256)
    assert n_bx == 4 + S * 12544 * 256 and 200 <= 49 * S <= 256, (n_bx, S)
    assert L.sf_conv_bx_ws_floats(ctypes.byref(d4), 1, 1) == n_bx          # fp32 rows: no activation planes either way
    assert L.sf_conv_bx_ws_floats(ctypes.byref(d4), 0, 0) == n_bx + w_planes
    assert L.sf_conv_fwd_ws_floats(ctypes.byref(d4)) == n_bx + w_planes
    assert L.sf_conv_bx_ws_floats(ctypes.byref(_conv_desc(8, 8, 28, 28, 128, 128, (1, 3, 3))), 1, 1) > 0
    assert L.sf_bx_planes_elems(12544, 1024) == 3 * 12545 * 1024
    assert L.sf_conv_bx_ws_floats(ctypes.byref(d), 0, 0) == 0                  # res2 3x3 64 -> 64: not a bx shape
    assert L.sf_conv_tune(7, 0) == 0
    try:
        assert L.sf_conv_bx_ws_floats(ctypes.byref(d4), 0, 1) == 0
        assert L.sf_conv_fwd_ws_floats(ctypes.byref(d4)) == 0
        assert L.sf_conv_tune(0, 0) == 0 and L.sf_conv_tune(99, 0) != 0
        try:
            n = L.sf_conv_fwd_ws_floats(ctypes.byref(d4))
        finally:
            L.sf_conv_tune(0, 1)
    finally:
        L.sf_conv_tune(7, 1)
    assert n > 0 and n % (12544 * 256) == 0 and 2 <= n // (12544 * 256) <= 12
    # pointwise layers (conv_pw_bx_kernel): res4's 256 -> 1024 at M = 12 544 goes to the bf16 pipe with ONLY the weight
    # planes as workspace (the rows are split after the LDS read) and leaves its BN statistics itself, one record row
    # per 256-position tile; 64-wide outputs (res2's 256 -> 64) and few tiles under a long reduction (the data gradient
    # 1024 -> 256 at 12 544 rows: 98 tiles x 64 steps, no split-K) stay on the f32 kernels; sf_conv_tune(21, 0) = off
    dp = _conv_desc(8, 8, 14, 14, 256, 1024, (1, 1, 1))
    wpl = -(-(3 * (1024 + 1) * 256 // 2) // 4) * 4
    assert L.sf_conv_pw_ws_floats(ctypes.byref(dp), 1) == 4 and L.sf_conv_pw_ws_floats(ctypes.byref(dp), 0) == 4 + wpl
    assert L.sf_conv_fwd_ws_floats(ctypes.byref(dp)) == 4 + wpl
    assert L.sf_conv_pw_stats_floats(ctypes.byref(dp)) == 49 * 4 * 1024 and L.sf_conv_stats_ws_floats(ctypes.byref(dp)) == 0
    # round 6: 64-wide outputs over >= 128 input channels take the 256 x 64 tile (conv_pw_bx_kernel<64, 2>); short
    # reductions into 64 channels stay on the f32 kernels
    assert L.sf_conv_pw_ws_floats(ctypes.byref(_conv_desc(8, 8, 56, 56, 256, 64, (1, 1, 1))), 1) == 4
    assert L.sf_conv_pw_ws_floats(ctypes.byref(_conv_desc(8, 8, 56, 56, 64, 64, (1, 1, 1))), 1) == 0
    dt = _conv_desc(8, 8, 14, 14, 1024, 256, (1, 1, 1))
    dt.transposed = 1
    assert L.sf_conv_pw_ws_floats(ctypes.byref(dt), 1) == 0
    assert L.sf_conv_pw_ws_floats(ctypes.byref(_conv_desc(8, 8, 56, 56, 64, 256, (1, 1, 1))), 1) == 4
    assert L.sf_conv_tune(21, 0) == 0
    try:
        assert L.sf_conv_pw_ws_floats(ctypes.byref(dp), 1) == 0 and L.sf_conv_stats_ws_floats(ctypes.byref(dp)) > 0
    finally:
        L.sf_conv_tune(21, 1)
    # Fast pathway res2 1x3x3 8 -> 8 at 8x32x56x56 (M = 802 816): the rows kernel (conv_wgrad_rows.hip) — ONE channel block,
    # ~640 workgroups of >= 2 stages of 128 positions; without it (sf_conv_tune(20, 0)) 9 per-tap tiles x 85 splits
    df = _conv_desc(8, 32, 56, 56, 8, 8, (1, 3, 3))
    S = L.sf_conv_wgrad_splits(ctypes.byref(df))
    assert 512 <= S <= 640 and -(-802816 // S) >= 256, S
    assert L.sf_conv_tune(20, 0) == 0
    try:
        assert L.sf_conv_wgrad_splits(ctypes.byref(df)) == 85
    finally:
        L.sf_conv_tune(20, 1)
    # Fast stem in the stem-trick layout (5x7x1 over pixels of 8 floats, 28 packed channels): persistent ring kernel
    ds = _conv_desc(8, 32, 230, 115, 28, 8, (5, 7, 1), stride=(1, 2, 1), cin_pad=32, in_cs=8, pad=(2, 0, 0))
    ds.Wo = 112
    assert (ds.To, ds.Ho) == (32, 112)
    assert L.sf_conv_wgrad_splits(ctypes.byref(ds)) == 256
    # attention workspaces: dQ planes (one per 128 / 64 keys) + room for 8 dK / dV parts; forward: 8 key parts
    B, N = 8, 25088
    # d = 32 adds the bf16 piece planes of the split-operand kernels (attn_bx.h): 3 pieces x 2 B per element of
    # [B, N rounded up to 64, 32], two tensors per direction (Q and gamma dz backward; K and V^T forward)
    planes = B * 3 * N * 32  # N % 64 == 0 here; in floats: 2 tensors x 2 B = 4 B per piece element
    assert L.sf_attn_bwd_fused_ws_floats(B, N, 32) == B * (N // 128 + 16) * N * 32 + planes
    assert L.sf_attn_bwd_fused_ws_floats(B, N, 8) == B * (N // 64 + 16) * N * 8 + B * N * 32  # + packed planes (d <= 8)
    assert L.sf_attn_bwd_fused_ws_floats(B, N, 128) == 0                       # d = 128 keeps the two-kernel form
    assert L.sf_attn_bwd_fused_ws_floats(B, N, 64) == B * (N // 128 + 16) * N * 64 + 2 * planes  # two channel blocks
    assert L.sf_attn_fwd_ws_floats(B, N, 32) == B * 8 * N * 34 + planes
    assert L.sf_attn_fwd_ws_floats(B, N, 64) == B * 8 * N * 66 + 2 * planes
    assert L.sf_attn_fwd_ws_floats(B, N, 8) == B * 8 * N * 10 + B * N * 32
    assert L.sf_attn_products_per_fp32(32) in (0, 6) and L.sf_attn_products_per_fp32(64) in (0, 6)
    assert L.sf_attn_products_per_fp32(128) == 0 and L.sf_attn_products_per_fp32(16) == 0
    assert L.sf_attn_products_per_fp32(8) in (0, 6) and L.sf_attn_products_per_fp32(4) == 0
    assert L.sf_attn_fwd_ws_floats(0, N, 32) == 0 and L.sf_attn_fwd_ws_floats(B, N, 129) == 0


def test_error_codes_match_header(repo_root):
    """The binding's SF_E* constants are the header's (SF_ENOTTAKEN, round 6: "not this entry point's shape" — the only
    code besides SF_EALIGN on which the binding falls back to a general entry point)."""
    import sfhip
    txt = open(os.path.join(repo_root, "include", "sfhip.h")).read()
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define (SF_E[A-Z]+) \((-\d+)\)", txt)}
    assert codes == {"SF_EINVAL": sfhip.SF_EINVAL, "SF_EALIGN": sfhip.SF_EALIGN, "SF_ELAUNCH": sfhip.SF_ELAUNCH,
                     "SF_ENOTTAKEN": sfhip.SF_ENOTTAKEN}
    assert set(sfhip._ERR) == set(codes.values())
