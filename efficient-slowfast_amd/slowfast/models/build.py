"""Model registry + factory: the drop-in boundary (reference models/build.py:9-44, fvcore Registry
fvcore/common/registry.py:40-75)."""
import torch


class Registry(object):
    """name -> class map; register() works as decorator or call; duplicate names assert, unknown names KeyError."""

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, "An object named '{}' was already registered in '{}' registry!".format(
            name, self._name)
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return ret

    def __contains__(self, name):
        return name in self._obj_map


MODEL_REGISTRY = Registry("MODEL")
MODEL_REGISTRY.__doc__ = "Registry for video models: obj(cfg) -> torch.nn.Module."


def build_model(cfg):
    """cfg.MODEL.MODEL_NAME -> module; moved to the current GPU iff NUM_GPUS >= 1, wrapped in
    DistributedDataParallel iff NUM_GPUS > 1 (one process per GPU, RCCL) — reference build.py:18-44."""
    assert cfg.NUM_GPUS <= torch.cuda.device_count(), "Cannot use more GPU devices than available"
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    if cfg.NUM_GPUS >= 1:
        cur_device = torch.cuda.current_device()
        model = model.cuda(device=cur_device)
    if cfg.NUM_GPUS > 1:
        model._sf_ddp_wrapped = True  # engine.set_grad_sink refuses models whose gradients must pass DDP's hooks
        model = torch.nn.parallel.DistributedDataParallel(
            module=model, device_ids=[cur_device], output_device=cur_device)
    return model
