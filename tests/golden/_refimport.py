"""Import the reference's ``slowfast.models`` in THIS container (never on the GPU box).

Test tooling only: used by ``make_golden.py`` to generate the committed fixtures.
The reference (``/root/reference``) needs five third-party modules that are not
installed here (fvcore's deps ``yacs``/``portalocker``, ``simplejson``,
``detectron2.layers.ROIAlign`` imported by head_helper.py:8, ``mmcv.cnn`` imported by
wdf_attention_helper.py:279).  None of them is on the hot path, so they are replaced
by inert stand-ins in ``sys.modules``.  Nothing is written
into /root/reference (bytecode writing is switched off).
"""
import copy
import json
import sys
import types

import yaml
import torch.nn as nn

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _CfgNode(dict):
    """Minimal yacs.config.CfgNode stand-in (attribute dict + merge helpers)."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = _CfgNode(v) if isinstance(v, dict) and not isinstance(v, _CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_other_cfg(self, o):
        for k, v in o.items():
            if isinstance(v, dict) and k in self and isinstance(self[k], dict):
                self[k].merge_from_other_cfg(_CfgNode(v))
            else:
                self[k] = _CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, f, allow_unsafe=False):
        with open(f) as fh:
            self.merge_from_other_cfg(_CfgNode(yaml.safe_load(fh)))

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            d = self
            ks = k.split(".")
            for s in ks[:-1]:
                d = d[s]
            d[ks[-1]] = yaml.safe_load(v) if isinstance(v, str) else v

    def freeze(self):
        pass

    def defrost(self):
        pass


class _ROIAlign(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


_DONE = False


def import_reference():
    """Returns (get_cfg, build_model) of the reference."""
    global _DONE
    sys.dont_write_bytecode = True
    if not _DONE:
        _mod("simplejson", dumps=lambda o, **k: json.dumps(o, default=str))
        _mod("portalocker", Lock=object, lock=lambda *a, **k: None,
             unlock=lambda *a, **k: None, LOCK_EX=1)
        _mod("yacs").config = _mod("yacs.config", CfgNode=_CfgNode)
        _mod("detectron2").layers = _mod("detectron2.layers", ROIAlign=_ROIAlign)
        _mod("mmcv").cnn = _mod(
            "mmcv.cnn",
            constant_init=lambda m, val=0: nn.init.constant_(m.weight, val),
            kaiming_init=lambda m, **k: nn.init.kaiming_normal_(m.weight),
        )
        sys.path[:0] = [REF_ROOT + "/SlowFast", REF_ROOT + "/config_slowfast/fvcore"]
        _DONE = True
    from slowfast.config.defaults import get_cfg
    from slowfast.models import build_model
    return get_cfg, build_model


def ref_yaml(name):
    return REF_ROOT + "/SlowFast/configs/Kinetics/" + name
