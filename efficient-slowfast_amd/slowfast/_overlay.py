"""How this `slowfast` package sits IN FRONT of the reference's own `slowfast` tree.

The reference's entry points (`SlowFast/tools/run_net.py:5-6`, `tools/train_net.py:10-21`, `tools/test_net.py`) import
about twenty `slowfast.*` modules; this repo carries only the ones on the hot path.  Two rules make the
rest keep resolving to the reference when both trees are importable:

* `chain_package(globals())` in every `__init__.py` here: the package's `__path__` becomes [this repo's directory, the
  same-named directories of every other `slowfast` tree that is importable], so `slowfast.utils.lr_policy`,
  `slowfast.models.optimizer`, `slowfast.datasets.loader` ... are found in the reference, while any module this repo
  DOES carry is found here first.  The other tree's `__init__.py` (dataset registration in
  `slowfast/datasets/__init__.py`, `setup_environment()` in `slowfast/__init__.py`) is run inside this package's
  namespace, after this repo's own names exist, and never replaces a name this repo defines.
* `chain_module(globals())` as the first statement of a module whose reference namesake has a LARGER surface than the
  hot path needs (`utils/misc.py` — `launch_job`, `log_model_info`; `utils/distributed.py` — `all_reduce`,
  `is_master_proc`; `utils/meters.py` — `TrainMeter`, `ValMeter`; `datasets/utils.py`): the namesake's source is
  executed in this module's namespace first, then the definitions below it override the names they share.

With no other `slowfast` tree importable both calls do nothing and the package stands alone (tests, bench.py)."""
import importlib.util
import os
import sys
import warnings

_HERE = os.path.dirname(os.path.abspath(__file__))


def _same(a, b):
    try:
        return os.path.samefile(a, b)
    except OSError:
        return os.path.abspath(a) == os.path.abspath(b)


def _parent_search_path(pkg_name):
    """Directories that may hold `pkg_name`'s directory: the parent package's `__path__`, or `sys.path`."""
    parent, _, leaf = pkg_name.rpartition(".")
    if parent:
        roots = list(getattr(sys.modules.get(parent), "__path__", []))
    else:
        roots = [p or os.getcwd() for p in sys.path]
    return roots, leaf


def _is_ours(path):
    path = os.path.abspath(path)
    return path == _HERE or path.startswith(_HERE + os.sep)


def other_dirs(pkg_name, own_path):
    """Same-named package directories of OTHER trees, in search order, without duplicates."""
    roots, leaf = _parent_search_path(pkg_name)
    found = []
    for root in roots:
        cand = os.path.join(root, leaf)
        if not os.path.isfile(os.path.join(cand, "__init__.py")):
            continue
        if _is_ours(cand) or any(_same(cand, p) for p in list(own_path) + found):
            continue
        found.append(cand)
    return found


def _exec_under(path, namespace, keep):
    """Run the file at `path` in `namespace`; names listed in `keep` are restored afterwards.  A namesake that cannot
    be imported (one of ITS dependencies is missing — cv2, fvcore ...) leaves the namespace as it was and warns: the
    hot path needs only this repo's names, and the reference's own entry points would fail on the same import."""
    before = dict(namespace)
    with open(path, "rb") as f:
        code = compile(f.read(), path, "exec")
    try:
        exec(code, namespace)
    except ImportError as e:
        namespace.clear()
        namespace.update(before)
        warnings.warn("slowfast overlay: %s not chained (%s: %s); only this repo's names are available in %s"
                      % (path, type(e).__name__, e, namespace.get("__name__")), RuntimeWarning, stacklevel=3)
        return False
    namespace.update({k: before[k] for k in keep if k in before})
    return True


def chain_package(namespace, run_init=True):
    """Call from a package `__init__.py` AFTER its own imports: `chain_package(globals())`."""
    name, own = namespace["__name__"], namespace["__path__"]
    others = other_dirs(name, own)
    for d in others:
        own.append(d)
    if run_init and others:
        mine = [k for k in namespace if not (k.startswith("__") and k.endswith("__"))]
        _exec_under(os.path.join(others[0], "__init__.py"), namespace,
                    keep=mine + ["__name__", "__path__", "__file__", "__doc__", "__package__", "__spec__",
                                 "__loader__"])
    return others


def chain_module(namespace):
    """Call as the FIRST statement of a module: `chain_module(globals())`; returns the namesake's path or None.

    The namesake is loaded as its own module object (`<package>._chained_<leaf>`, so its relative imports and its
    functions' globals stay its own) and its non-dunder names are copied into `namespace`; whatever the calling
    module defines afterwards replaces them for every importer of the public name."""
    name = namespace["__name__"]
    parent, _, leaf = name.rpartition(".")
    for root in getattr(sys.modules.get(parent), "__path__", []):
        cand = os.path.join(root, leaf + ".py")
        if not os.path.isfile(cand) or _is_ours(cand):
            continue
        alias = "%s._chained_%s" % (parent, leaf)
        spec = importlib.util.spec_from_file_location(alias, cand)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[alias] = mod
        try:
            spec.loader.exec_module(mod)
        except ImportError as e:
            del sys.modules[alias]
            warnings.warn("slowfast overlay: %s not chained (%s: %s); only this repo's names are available in %s"
                          % (cand, type(e).__name__, e, name), RuntimeWarning, stacklevel=2)
            return None
        for k, v in vars(mod).items():
            if not (k.startswith("__") and k.endswith("__")):
                namespace[k] = v
        namespace["__chained_from__"] = cand
        return cand
    return None
