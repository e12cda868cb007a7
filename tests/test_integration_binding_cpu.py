"""INTEGRATION.md §1 / §2 as a maintainer would type them, run in a subprocess.

A SYNTHETIC stand-in for the reference's tree is written into tmp_path (nothing is copied from the reference): a
`slowfast/` package holding the modules its entry points import that this repo does NOT carry — `utils/lr_policy.py`,
`utils/logging.py`, `utils/env.py`, `models/optimizer.py`, `models/losses.py`, `datasets/{build,loader,kinetics}.py` —
and namesakes of modules this repo DOES carry, each with a larger surface (`utils/misc.py: launch_job`,
`utils/distributed.py: all_gather_unaligned`, `utils/meters.py: TrainMeter`) or with bodies that must NOT win
(`models/build.py`, `config/defaults.py`).  `tools/run.py` then does what `SlowFast/tools/run_net.py:5-6` +
`tools/train_net.py:10-21, 373-378` do: the imports, `load_config`-style cfg, `build_model(cfg)`, the
`"bn" in name` split of `models/optimizer.py:30-44`, `launch_job`.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "efficient-slowfast_amd")

TREE = {
    "slowfast/__init__.py": """
        from slowfast.utils.env import setup_environment
        setup_environment()
        """,
    "slowfast/utils/__init__.py": "",
    "slowfast/utils/env.py": """
        _ENV_SETUP_DONE = []
        def setup_environment():
            _ENV_SETUP_DONE.append(1)
        """,
    "slowfast/utils/lr_policy.py": """
        def get_lr_at_epoch(cfg, cur_epoch):
            return cfg.SOLVER.BASE_LR * 0.5
        """,
    "slowfast/utils/logging.py": """
        import logging
        def get_logger(name):
            return logging.getLogger(name)
        def log_json_stats(stats):
            pass
        """,
    # namesakes of modules this repo carries: larger surface, plus one body each that must be overridden
    "slowfast/utils/misc.py": """
        import slowfast.utils.logging as logging
        from slowfast.datasets.utils import pack_pathway_output
        from slowfast.models.batchnorm_helper import SubBatchNorm3d
        logger = logging.get_logger(__name__)
        def launch_job(cfg, init_method, func, daemon=False):
            return func(cfg=cfg)
        def frozen_bn_stats(model):
            raise RuntimeError("stand-in body: the MI355X package's frozen_bn_stats must win")
        def is_eval_epoch(cfg, cur_epoch, multigrid_schedule):
            return (cur_epoch + 1) % cfg.TRAIN.EVAL_PERIOD == 0
        """,
    "slowfast/utils/distributed.py": """
        import logging
        _LOCAL_PROCESS_GROUP = None
        def all_gather_unaligned(data, group=None):
            return [data]
        def get_world_size():
            raise RuntimeError("stand-in body: must be overridden")
        """,
    "slowfast/utils/meters.py": """
        import slowfast.utils.logging as logging
        class TrainMeter(object):
            def __init__(self, epoch_iters, cfg):
                self.epoch_iters = epoch_iters
            def log(self):
                logging.log_json_stats({"a": 1})
                return "logged"
        class TestMeter(object):
            STAND_IN = True
        """,
    "slowfast/models/__init__.py": """
        from .build import MODEL_REGISTRY, build_model  # noqa
        from .custom_video_model_builder import *  # noqa
        from .video_model_builder import ResNet, SlowFast  # noqa
        """,
    "slowfast/models/build.py": """
        raise RuntimeError("stand-in models/build.py imported: the MI355X package is not in front")
        """,
    "slowfast/models/video_model_builder.py": """
        raise RuntimeError("stand-in video_model_builder imported")
        """,
    "slowfast/models/optimizer.py": """
        import torch
        import slowfast.utils.lr_policy as lr_policy
        def construct_optimizer(model, cfg):
            bn, rest = [], []
            for name, p in model.named_parameters():
                (bn if "bn" in name else rest).append(p)
            assert len(list(model.parameters())) == len(bn) + len(rest)
            return torch.optim.SGD([{"params": bn, "weight_decay": cfg.BN.WEIGHT_DECAY},
                                    {"params": rest, "weight_decay": cfg.SOLVER.WEIGHT_DECAY}],
                                   lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)
        def get_epoch_lr(cur_epoch, cfg):
            return lr_policy.get_lr_at_epoch(cfg, cur_epoch)
        """,
    "slowfast/models/losses.py": """
        import torch.nn as nn
        _LOSSES = {"cross_entropy": nn.CrossEntropyLoss}
        def get_loss_func(loss_name):
            return _LOSSES[loss_name]
        """,
    "slowfast/config/__init__.py": "",
    "slowfast/config/defaults.py": """
        raise RuntimeError("stand-in config/defaults.py imported")
        """,
    "slowfast/datasets/__init__.py": """
        from .build import DATASET_REGISTRY, build_dataset  # noqa
        from .kinetics import Kinetics  # noqa
        """,
    "slowfast/datasets/build.py": """
        DATASET_REGISTRY = {}
        def build_dataset(name, cfg, split):
            return DATASET_REGISTRY[name.capitalize()](cfg, split)
        """,
    "slowfast/datasets/kinetics.py": """
        from . import utils as utils
        from .build import DATASET_REGISTRY
        class Kinetics(object):
            def __init__(self, cfg, mode):
                self.cfg, self.mode = cfg, mode
            def pack(self, frames):
                return utils.pack_pathway_output(self.cfg, frames)
        DATASET_REGISTRY["Kinetics"] = Kinetics
        """,
    "slowfast/datasets/loader.py": """
        from .build import build_dataset
        def construct_loader(cfg, split):
            return build_dataset(cfg.TRAIN.DATASET, cfg, split)
        """,
    "slowfast/datasets/utils.py": """
        def pack_pathway_output(cfg, frames):
            raise RuntimeError("stand-in body: must be overridden")
        def get_sequence(center_idx, half_len, sample_rate, num_frames):
            return list(range(center_idx - half_len, center_idx + half_len, sample_rate))
        """,
    "tools/run.py": """
        import json, os, sys
        from slowfast.utils.misc import launch_job
        import slowfast.models.losses as losses
        import slowfast.models.optimizer as optim
        import slowfast.utils.checkpoint as cu
        import slowfast.utils.distributed as du
        import slowfast.utils.logging as logging
        import slowfast.utils.misc as misc
        from slowfast.datasets import loader
        from slowfast.models import build_model
        from slowfast.utils.meters import TrainMeter, TestMeter
        from slowfast.config.defaults import get_cfg
        import slowfast, slowfast.models, slowfast.utils.env as env
        import torch

        def train(cfg):
            model = build_model(cfg)
            optimizer = optim.construct_optimizer(model, cfg)
            misc.frozen_bn_stats(model)
            ds = loader.construct_loader(cfg, "train")
            frames = torch.zeros(3, cfg.DATA.NUM_FRAMES, 4, 4)
            slow, fast = ds.pack(frames)
            return {
                "model_file": sys.modules[type(model).__module__].__file__,
                "models_file": slowfast.models.__file__,
                "build_module_file": sys.modules[build_model.__module__].__file__,
                "optim_file": optim.__file__,
                "losses_file": losses.__file__,
                "loader_file": loader.__file__,
                "cfg_file": sys.modules[get_cfg.__module__].__file__,
                "bn_group": len(optimizer.param_groups[0]["params"]),
                "rest_group": len(optimizer.param_groups[1]["params"]),
                "n_params": len(list(model.parameters())),
                "bn_names_ok": all(("bn" in n) == (id(p) in {id(q) for q in optimizer.param_groups[0]["params"]})
                                   for n, p in model.named_parameters()),
                "lr": optim.get_epoch_lr(0, cfg),
                "loss": losses.get_loss_func(cfg.MODEL.LOSS_FUNC).__name__,
                "env_setup": len(env._ENV_SETUP_DONE),
                "is_master": du.is_master_proc(),
                "world": du.get_world_size(),
                "unaligned": du.all_gather_unaligned(7),
                "train_meter": TrainMeter(3, cfg).log(),
                "test_meter_is_standin": hasattr(TestMeter, "STAND_IN"),
                "test_meter_has_tic": hasattr(TestMeter, "iter_tic"),
                "is_eval_epoch": misc.is_eval_epoch(cfg, 0, None),
                "slow_T": int(slow.shape[1]), "fast_T": int(fast.shape[1]),
                "registered": sorted(slowfast.models.MODEL_REGISTRY._obj_map),
                "last_ckpt_fn": cu.get_last_checkpoint.__module__,
            }

        cfg = get_cfg()
        cfg.merge_from_file(os.environ["SF_TEST_YAML"])          # utils/parser.py:67-81 load_config
        cfg.merge_from_list(["NUM_GPUS", 0, "DATA.NUM_FRAMES", 16, "SLOWFAST.ALPHA", 4])
        print("RESULT " + json.dumps(launch_job(cfg=cfg, init_method=None, func=train)))
        """,
}


def _write_tree(root, patch_init=False):
    for rel, body in TREE.items():
        path = os.path.join(str(root), rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        body = textwrap.dedent(body)
        if patch_init and rel == "slowfast/__init__.py":
            # the two-line patch to the reference's own slowfast/__init__.py
            body = textwrap.dedent("""
                import os
                if os.environ.get("SLOWFAST_AMD_ROOT"):
                    __path__.insert(0, os.path.join(os.environ["SLOWFAST_AMD_ROOT"], "slowfast"))
                """) + body
        with open(path, "w") as f:
            f.write(body)


YAMLS = {"SlowFastDualAttention": "SLOWFAST_DUAL_8x8_R50.yaml", "SlowFastShuffleNetV2": "SLOWFAST_SHUFFLENETV2_4x16.yaml"}


def _run(tmp_path, env_extra, model="SlowFastDualAttention"):
    env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "SLOWFAST_AMD_ROOT")}
    env.update(env_extra)
    env.update({"SF_TEST_YAML": os.path.join(ROOT, "configs", YAMLS[model]), "PYTHONDONTWRITEBYTECODE": "1", "PYTHONWARNINGS": "error::RuntimeWarning"})
    p = subprocess.run([sys.executable, os.path.join(str(tmp_path), "tools", "run.py")], cwd=str(tmp_path), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def _check(r, tmp_path):
    ref = os.path.realpath(str(tmp_path))
    for k in ("model_file", "models_file", "build_module_file", "cfg_file"):
        assert os.path.realpath(r[k]).startswith(os.path.realpath(PKG)), (k, r[k])
    for k in ("optim_file", "losses_file", "loader_file"):
        assert os.path.realpath(r[k]).startswith(ref), (k, r[k])
    assert r["bn_group"] + r["rest_group"] == r["n_params"] and r["bn_group"] > 0 and r["bn_names_ok"]
    assert r["lr"] == pytest.approx(0.05) and r["loss"] == "CrossEntropyLoss"
    assert r["env_setup"] == 1, "the other tree's slowfast/__init__.py must run exactly once"
    assert r["is_master"] is True and r["world"] == 1 and r["unaligned"] == [7]
    assert r["train_meter"] == "logged"
    assert r["test_meter_is_standin"] is False and r["test_meter_has_tic"] is True
    assert r["is_eval_epoch"] is True
    assert (r["slow_T"], r["fast_T"]) == (4, 16)
    assert r["registered"] == sorted(["ResNet", "SlowFast", "SlowFastDualAttention", "SlowFastGhostNet",
                                      "SlowFastMoibleNetV2", "SlowFastShuffleNet", "SlowFastShuffleNetV2"])
    assert r["last_ckpt_fn"] == "slowfast.utils.checkpoint"


@pytest.mark.parametrize("model", ["SlowFastDualAttention", "SlowFastShuffleNetV2"])
def test_package_swap_by_pythonpath(tmp_path, model):
    """PYTHONPATH=<this repo>/efficient-slowfast_amd:<reference>/SlowFast."""
    _write_tree(tmp_path)
    r = _run(tmp_path, {"PYTHONPATH": PKG + os.pathsep + str(tmp_path)}, model=model)
    _check(r, tmp_path)


def test_patch_to_the_reference_package_init(tmp_path):
    """only the reference's tree on PYTHONPATH + SLOWFAST_AMD_ROOT + the patch to its __init__."""
    _write_tree(tmp_path, patch_init=True)
    r = _run(tmp_path, {"PYTHONPATH": str(tmp_path) + os.pathsep + PKG, "SLOWFAST_AMD_ROOT": PKG})
    _check(r, tmp_path)


def test_standalone_package_has_no_chain(tmp_path):
    """With no other slowfast tree importable the package stands alone and carries the du.* wrappers itself."""
    code = ("import slowfast, slowfast.models, slowfast.utils.distributed as du, slowfast.utils.misc as m;"
            "assert len(slowfast.__path__) == 1 and len(slowfast.models.__path__) == 1;"
            "assert du.is_master_proc() and du.get_world_size() == 1 and not hasattr(m, '__chained_from__');"
            "import torch; t = torch.ones(3); assert du.all_reduce([t])[0] is t and du.all_gather([t])[0].shape == (3,)")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = PKG
    subprocess.run([sys.executable, "-c", code], check=True, env=env, cwd=str(tmp_path), timeout=300)


@pytest.mark.skipif(not os.path.isdir("/root/reference/SlowFast/slowfast"), reason="reference tree not present")
def test_against_the_real_reference_tree(tmp_path):
    """The exact imports VERDICT r05 found broken, against the real tree (this container only; read-only, no
    bytecode written).  The reference's utils/misc.py needs fvcore/psutil/cv2, absent here, so that chain is
    allowed to warn; lr_policy / optimizer / losses / distributed need nothing that is missing."""
    code = textwrap.dedent("""
        import warnings; warnings.simplefilter("ignore", RuntimeWarning)
        import slowfast.utils.lr_policy as lp, slowfast.models.optimizer as optim, slowfast.models.losses as losses
        import slowfast.utils.distributed as du, slowfast.models as M
        from slowfast.config.defaults import get_cfg
        assert lp.__file__.startswith("/root/reference") and optim.__file__.startswith("/root/reference")
        assert not M.__file__.startswith("/root/reference") and du.__chained_from__.startswith("/root/reference")
        cfg = get_cfg()
        cfg.merge_from_file("%s")
        cfg.merge_from_list(["NUM_GPUS", 0, "DATA.NUM_FRAMES", 16, "SLOWFAST.ALPHA", 4,
                             "SOLVER.LR_POLICY", "cosine", "SOLVER.MAX_EPOCH", 10])
        model = M.build_model(cfg)
        opt = optim.construct_optimizer(model, cfg)
        n_bn = sum(1 for n, _ in model.named_parameters() if "bn" in n)
        assert len(opt.param_groups[0]["params"]) == n_bn > 0
        lr = optim.get_epoch_lr(1, cfg); optim.set_lr(opt, lr); assert 0 < lr < cfg.SOLVER.BASE_LR
        assert "CrossEntropyLoss" in repr(losses.get_loss_func("cross_entropy"))
        assert du.is_master_proc() and du.all_gather_unaligned is not None
        print("OK")
        """) % os.path.join(ROOT, "configs", YAMLS["SlowFastDualAttention"])
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env.update({"PYTHONPATH": PKG + os.pathsep + "/root/reference/SlowFast", "PYTHONDONTWRITEBYTECODE": "1"})
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-3000:]


def test_namesake_with_a_missing_dependency_warns_and_keeps_this_repos_names(tmp_path):
    """A reference namesake whose own third-party import is missing (cv2, fvcore ... in a bare environment) is skipped
    with a RuntimeWarning; the names this repo defines are there, the ones only the namesake defines are not."""
    _write_tree(tmp_path)
    with open(os.path.join(str(tmp_path), "slowfast", "utils", "misc.py"), "w") as f:
        f.write("import a_module_that_is_not_installed_anywhere\ndef launch_job(cfg, init_method, func, daemon=False):\n    return 1\n")
    code = textwrap.dedent("""
        import warnings
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            import slowfast.utils.misc as misc
        msgs = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
        assert any("not chained" in m and "a_module_that_is_not_installed_anywhere" in m for m in msgs), msgs
        assert hasattr(misc, "frozen_bn_stats") and hasattr(misc, "aggregate_sub_bn_stats")
        assert not hasattr(misc, "launch_job") and getattr(misc, "__chained_from__", None) is None
        import slowfast.utils.lr_policy as lp        # the rest of the other tree is still there
        assert lp.get_lr_at_epoch is not None
        print("OK")
        """)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env.update({"PYTHONPATH": PKG + os.pathsep + str(tmp_path), "PYTHONDONTWRITEBYTECODE": "1"})
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd="/", stdout=subprocess.PIPE,  # (not tmp_path: -c puts the cwd first)
                       stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-3000:]
