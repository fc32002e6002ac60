"""``wan`` package surface of the reference (seaweed_apt/wan/__init__.py:1-3)."""
from . import configs, modules  # noqa: F401
from .image2video import WanI2V  # noqa: F401
from .text2video import WanT2V  # noqa: F401
