"""seaweed_apt/wan/modules/__init__.py:1-5 — hot-path modules only (T5/CLIP
encoders are out of scope)."""
from .attention import flash_attention
from .model import WanModel
from .vae import WanVAE

__all__ = ["WanVAE", "WanModel", "flash_attention"]
