"""A small yacs-compatible CfgNode (the reference uses fvcore's CfgNode over yacs, config/defaults.py:4,
utils/parser.py:67-81).  Behaviour kept: attribute access, clone(), merge_from_file(yaml),
merge_from_list([k, v, ...]) with dotted keys, KeyError on keys that are not in the defaults, type
compatibility checks (int->float, list<->tuple allowed), freeze()/defrost()."""
import ast
import copy

import yaml


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init or {}).items():
            dict.__setitem__(self, k, CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v)

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode.IMMUTABLE]:
            raise AttributeError("Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode.IMMUTABLE] = self.__dict__[CfgNode.IMMUTABLE]
        return out

    # ---- merging
    @staticmethod
    def _coerce(new, old, full_key):
        if isinstance(new, str) and not isinstance(old, str):
            try:  # yacs decodes strings with literal_eval ("1e-4" is a str to PyYAML)
                new = ast.literal_eval(new)
            except (ValueError, SyntaxError):
                pass
        if old is None or new is None or type(new) == type(old):
            return new
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        raise ValueError("Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(
            type(old), type(new), old, new, full_key))

    def _merge_dict(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: {}".format(full))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("Expected a mapping for config key: {}".format(full))
                self[k]._merge_dict(v, path + [k])
            else:
                dict.__setitem__(self, k, self._coerce(copy.deepcopy(v), self[k], full))

    def merge_from_other_cfg(self, other):
        self._merge_dict(other, [])

    def merge_from_file(self, cfg_filename, allow_unsafe=False):
        with open(cfg_filename, "r") as f:
            loaded = yaml.safe_load(f) or {}
        self._merge_dict(loaded, [])

    def merge_from_list(self, cfg_list):
        if len(cfg_list) % 2 != 0:
            raise AssertionError("Override list has odd length: {}; it must be a list of pairs".format(cfg_list))
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError("Non-existent key: {}".format(full_key))
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("Non-existent key: {}".format(full_key))
            if isinstance(v, str):
                try:
                    v = yaml.safe_load(v)
                except yaml.YAMLError:
                    pass
            dict.__setitem__(node, parts[-1], self._coerce(v, node[parts[-1]], full_key))

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self))
