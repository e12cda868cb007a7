"""Caffe2 -> PyTorch parameter-name conversion for the published PySlowFast weights (reference
utils/c2_model_loading.py:7-112).  The mapping is the Caffe2 blob naming contract itself, held here as an ordered
rewrite table; the slow ("res*") and fast ("t_res*") pathway rules are generated from one template."""
import re


def _pathway_rules(prefix, pathway):
    p = "pathway%d" % pathway
    return [
        (r"^%sres([0-9]+)_([0-9]+)_branch([0-9]+)([a-z])_(.*)" % prefix, r"s\1.%s_res\2.branch\3.\4_\5" % p),
        (r"^%sres_conv1_bn_(.*)" % prefix, r"s1.%s_stem.bn.\1" % p),
        (r"^%sconv1_(.*)" % prefix, r"s1.%s_stem.conv.\1" % p),
        (r"^%sres([0-9]+)_([0-9]+)_branch([0-9]+)_(.*)" % prefix, r"s\1.%s_res\2.branch\3_\4" % p),
        (r"^%sres_conv1_(.*)" % prefix, r"s1.%s_stem.conv.\1" % p),
    ]


_RULES = (
    # non-local blocks: nonlocal_conv<stage>_<block>_<part>
    [(r"^nonlocal_conv([0-9]+)_([0-9]+)_(.*)", r"s\1.pathway0_nonlocal\2_\3")]
    + [(r"^(.*)_nonlocal([0-9]+)_(%s)(.*)" % part, r"\1_nonlocal\2.conv_\3\4") for part in ("theta", "g", "phi", "out")]
    + [(r"^(.*)_nonlocal([0-9]+)_(bn)_(.*)", r"\1_nonlocal\2.\3.\4")]
    # Fast -> Slow lateral connections
    + [(r"^t_pool1_subsample_bn_(.*)", r"s1_fuse.bn.\1"),
       (r"^t_pool1_subsample_(.*)", r"s1_fuse.conv_f2s.\1"),
       (r"^t_res([0-9]+)_([0-9]+)_branch2c_bn_subsample_bn_(.*)", r"s\1_fuse.bn.\3"),
       (r"^t_res([0-9]+)_([0-9]+)_branch2c_bn_subsample_(.*)", r"s\1_fuse.conv_f2s.\3")]
    + _pathway_rules("", 0) + _pathway_rules("t_", 1)
    # head and parameter suffixes
    + [(r"pred_(.*)", r"head.projection.\1"),
       (r"(.*)bn.b\Z", r"\1bn.bias"), (r"(.*)bn.s\Z", r"\1bn.weight"),
       (r"(.*)bn.rm\Z", r"\1bn.running_mean"), (r"(.*)bn.riv\Z", r"\1bn.running_var"),
       (r"(.*)[\._]b\Z", r"\1.bias"), (r"(.*)[\._]w\Z", r"\1.weight")]
)
_COMPILED = [(re.compile(src), dst) for src, dst in _RULES]


def get_name_convert_func():
    """Returns f(caffe2_blob_name) -> state_dict key (every rule applied in order)."""

    def convert_caffe2_name_to_pytorch(name):
        for pattern, repl in _COMPILED:
            name = pattern.sub(repl, name)
        return name

    return convert_caffe2_name_to_pytorch
