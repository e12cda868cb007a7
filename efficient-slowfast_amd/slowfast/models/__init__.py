from .build import MODEL_REGISTRY, build_model  # noqa: F401
from .video_model_builder import SlowFast  # noqa: F401
from .custom_video_model_builder import (  # noqa: F401
    SlowFastDualAttention, SlowFastGhostNet, SlowFastShuffleNetV2)
from slowfast._overlay import chain_package as _chain_package

_chain_package(globals())  # modules this repo does not carry resolve to the reference's slowfast/models/
