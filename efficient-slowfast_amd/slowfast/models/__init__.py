from .build import MODEL_REGISTRY, build_model  # noqa: F401
from .video_model_builder import SlowFast  # noqa: F401
from .custom_video_model_builder import (  # noqa: F401
    SlowFastDualAttention, SlowFastGhostNet, SlowFastShuffleNetV2)
