"""MI355X-native drop-in for the hot path of weidafeng/Efficient-SlowFast's `slowfast` package:
`slowfast.config.defaults.get_cfg`, `slowfast.models.{MODEL_REGISTRY, build_model}` with the reference's
model names, YAML keys, state_dict layout and child-module order; forward runs on libsfhip (HIP, gfx950)."""

from ._overlay import chain_package as _chain_package

_chain_package(globals())  # the reference's own slowfast/ tree, when importable, serves every module not carried here
