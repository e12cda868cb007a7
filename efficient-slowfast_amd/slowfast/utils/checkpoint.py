"""Checkpoint I/O compatible with the reference (utils/checkpoint.py:18-473): the same `.pyth` files
(`torch.save({"epoch", "model_state", "optimizer_state", "cfg"})` under <job>/checkpoints/checkpoint_epoch_%05d.pyth),
Sub-BN <-> BN state-dict normalisation, 2D -> 3D weight inflation and Caffe2 pickle loading, so weights trained
with the reference (or published for PySlowFast) load into the HIP-path models and vice versa.  Plain files only
(the reference's PathManager is fvcore, which this image does not have)."""
import copy
import os
import pickle
from collections import OrderedDict

import numpy as np
import torch

import slowfast.utils.distributed as du
from slowfast.utils.c2_model_loading import get_name_convert_func


def _is_master(num_gpus=8):
    return du.get_rank() % max(num_gpus, 1) == 0


def get_checkpoint_dir(path_to_job):
    return os.path.join(path_to_job, "checkpoints")


def make_checkpoint_dir(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    if _is_master() and not os.path.exists(d):
        try:
            os.makedirs(d)
        except OSError:
            pass
    return d


def get_path_to_checkpoint(path_to_job, epoch):
    return os.path.join(get_checkpoint_dir(path_to_job), "checkpoint_epoch_{:05d}.pyth".format(epoch))


def _checkpoint_files(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    return d, [f for f in (os.listdir(d) if os.path.exists(d) else []) if "checkpoint" in f]


def get_last_checkpoint(path_to_job):
    d, names = _checkpoint_files(path_to_job)
    assert len(names), "No checkpoints found in '{}'.".format(d)
    return os.path.join(d, sorted(names)[-1])


def has_checkpoint(path_to_job):
    return len(_checkpoint_files(path_to_job)[1]) > 0


def is_checkpoint_epoch(cfg, cur_epoch, multigrid_schedule=None):
    """Save on the last epoch, on the multigrid long-cycle evaluation grid, or every TRAIN.CHECKPOINT_PERIOD."""
    if cur_epoch + 1 == cfg.SOLVER.MAX_EPOCH:
        return True
    if multigrid_schedule is not None:
        prev_epoch = 0
        for s in multigrid_schedule:
            if cur_epoch < s[-1]:
                period = max((s[-1] - prev_epoch) // cfg.MULTIGRID.EVAL_FREQ + 1, 1)
                return (s[-1] - 1 - cur_epoch) % period == 0
            prev_epoch = s[-1]
    return (cur_epoch + 1) % cfg.TRAIN.CHECKPOINT_PERIOD == 0


def save_checkpoint(path_to_job, model, optimizer, epoch, cfg):
    """Writes checkpoint_epoch_{epoch+1:05d}.pyth (master process only); Sub-BN layers are stored as plain BN."""
    if not _is_master(cfg.NUM_GPUS * cfg.NUM_SHARDS):
        return None
    os.makedirs(get_checkpoint_dir(path_to_job), exist_ok=True)
    sd = model.module.state_dict() if cfg.NUM_GPUS > 1 else model.state_dict()
    checkpoint = {"epoch": epoch, "model_state": sub_to_normal_bn(sd), "optimizer_state": optimizer.state_dict(),
                  "cfg": cfg.dump()}
    path = get_path_to_checkpoint(path_to_job, epoch + 1)
    with open(path, "wb") as f:
        torch.save(checkpoint, f)
    return path


def inflate_weight(state_dict_2d, state_dict_3d):
    """I3D inflation: a [Co,Ci,kH,kW] weight becomes [Co,Ci,kT,kH,kW] = repeat over kT / kT; equal shapes are
    copied; anything else keeps the 3D model's own tensor."""
    out = OrderedDict()
    for k, v2d in state_dict_2d.items():
        assert k in state_dict_3d.keys()
        v3d = state_dict_3d[k]
        if v2d.dim() == 4 and v3d.dim() == 5:
            assert v2d.shape[-2:] == v3d.shape[-2:] and v2d.shape[:2] == v3d.shape[:2]
            v3d = v2d.unsqueeze(2).repeat(1, 1, v3d.shape[2], 1, 1) / v3d.shape[2]
        elif v2d.shape == v3d.shape:
            v3d = v2d
        out[k] = v3d.clone()
    return out


def _tile_1d(src_len, dst_len):
    """How many times a 1-D checkpoint tensor must be repeated to fill a (Sub-BN split) model tensor, or 0."""
    return dst_len // src_len if (dst_len > src_len and dst_len % src_len == 0) else 0


def load_checkpoint(path_to_checkpoint, model, data_parallel=True, optimizer=None, inflation=False,
                    convert_from_caffe2=False):
    """Loads a `.pyth` (or Caffe2 pickle) checkpoint into `model` (strict=False); returns its epoch (-1 if none)."""
    assert os.path.exists(path_to_checkpoint), "Checkpoint '{}' not found".format(path_to_checkpoint)
    ms = model.module if data_parallel else model
    if convert_from_caffe2:
        with open(path_to_checkpoint, "rb") as f:
            blobs = pickle.load(f, encoding="latin1")["blobs"]
        target = ms.state_dict()
        convert = get_name_convert_func()
        state_dict = OrderedDict()
        for key in blobs.keys():
            name = c2_normal_to_sub_bn(convert(key), target)
            if name not in target:
                continue  # momentum / lr / model_iter blobs and layers this model does not have
            blob, want = blobs[key], tuple(target[name].shape)
            if len(want) == 1 and blob.ndim == 1:
                rep = _tile_1d(blob.shape[0], want[0])
                if rep:
                    blob = np.concatenate([blob] * rep)
            if tuple(blob.shape) == want:
                state_dict[name] = torch.tensor(blob).clone()
        ms.load_state_dict(state_dict, strict=False)
        return -1
    with open(path_to_checkpoint, "rb") as f:
        checkpoint = torch.load(f, map_location="cpu", weights_only=False)
    model_sd = ms.state_dict()
    checkpoint["model_state"] = normal_to_sub_bn(checkpoint["model_state"], model_sd)
    if inflation:
        ms.load_state_dict(inflate_weight(checkpoint["model_state"], model_sd), strict=False)
    else:
        ms.load_state_dict(checkpoint["model_state"], strict=False)
        if optimizer:
            optimizer.load_state_dict(checkpoint["optimizer_state"])
    return checkpoint["epoch"] if "epoch" in checkpoint.keys() else -1


def sub_to_normal_bn(sd):
    """Sub-BN layers are saved as plain BN: `X.bn.bn.running_*` -> `X.bn.running_*`, the split copy's
    `num_batches_tracked` kept, `bn.bn.*` / `.split_bn.*` dropped, [C,1,1,1] affine tensors flattened."""
    new_sd = copy.deepcopy(sd)
    renames = (("bn.bn.running_mean", "bn.running_mean"), ("bn.bn.running_var", "bn.running_var"),
               ("bn.split_bn.num_batches_tracked", "bn.num_batches_tracked"))
    for key in sd:
        for before, after in renames:
            if key.endswith(before):
                new_sd[key.split(before)[0] + after] = new_sd.pop(key)
        for marker in ("bn.bn.", ".split_bn."):
            if marker in key and key in new_sd:
                del new_sd[key]
    for key in new_sd:
        if (key.endswith("bn.weight") or key.endswith("bn.bias")) and new_sd[key].dim() == 4:
            assert all(d == 1 for d in new_sd[key].size()[1:])
            new_sd[key] = new_sd[key][:, 0, 0, 0]
    return new_sd


def c2_normal_to_sub_bn(key, model_keys):
    """A converted Caffe2 running-stat name, redirected to the Sub-BN split copy when the model has one (returns
    None for a running stat the model lacks, like the reference)."""
    if "bn.running_" in key:
        if key in model_keys:
            return key
        new_key = key.replace("bn.running_", "bn.split_bn.running_")
        if new_key in model_keys:
            return new_key
        return None
    return key


def normal_to_sub_bn(checkpoint_sd, model_sd):
    """Plain-BN checkpoint -> a model with Sub-BN layers: running stats feed both `split_bn` (tiled NUM_SPLITS
    times) and `bn`."""
    for key in model_sd:
        if key not in checkpoint_sd and "bn.split_bn." in key:
            checkpoint_sd[key] = checkpoint_sd.pop(key.replace("bn.split_bn.", "bn."))
            checkpoint_sd[key.replace("bn.split_bn.", "bn.bn.")] = checkpoint_sd[key]
    for key in model_sd:
        if key in checkpoint_sd and model_sd[key].dim() == 1 and checkpoint_sd[key].dim() == 1:
            rep = _tile_1d(checkpoint_sd[key].shape[0], model_sd[key].shape[0])
            if rep:
                checkpoint_sd[key] = torch.cat([checkpoint_sd[key]] * rep)
    return checkpoint_sd


def load_test_checkpoint(cfg, model):
    """TEST.CHECKPOINT_FILE_PATH, else the newest checkpoint in OUTPUT_DIR, else TRAIN.CHECKPOINT_FILE_PATH."""
    c2 = cfg.TRAIN.CHECKPOINT_TYPE == "caffe2"
    if cfg.TEST.CHECKPOINT_FILE_PATH != "":
        load_checkpoint(cfg.TEST.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, None, inflation=False,
                        convert_from_caffe2=c2)
    elif has_checkpoint(cfg.OUTPUT_DIR):
        load_checkpoint(get_last_checkpoint(cfg.OUTPUT_DIR), model, cfg.NUM_GPUS > 1)
    elif cfg.TRAIN.CHECKPOINT_FILE_PATH != "":
        load_checkpoint(cfg.TRAIN.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, None, inflation=False,
                        convert_from_caffe2=c2)


def load_train_checkpoint(cfg, model, optimizer):
    """Auto-resume from OUTPUT_DIR, else TRAIN.CHECKPOINT_FILE_PATH (optionally inflated / Caffe2); returns the
    epoch to start from."""
    if cfg.TRAIN.AUTO_RESUME and has_checkpoint(cfg.OUTPUT_DIR):
        return load_checkpoint(get_last_checkpoint(cfg.OUTPUT_DIR), model, cfg.NUM_GPUS > 1, optimizer) + 1
    if cfg.TRAIN.CHECKPOINT_FILE_PATH != "":
        return load_checkpoint(cfg.TRAIN.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, optimizer,
                               inflation=cfg.TRAIN.CHECKPOINT_INFLATE,
                               convert_from_caffe2=cfg.TRAIN.CHECKPOINT_TYPE == "caffe2") + 1
    return 0
