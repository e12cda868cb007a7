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
