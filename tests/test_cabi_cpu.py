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
