"""seaweed_apt/wan/modules/__init__.py:1-5: the DiT, the VAE, the attention operator and the two prompt-side
encoders (umT5 text encoder, CLIP vision tower)."""
from .attention import flash_attention
from .clip import CLIPModel
from .model import WanModel
from .t5 import T5Encoder, T5EncoderModel
from .vae import WanVAE

__all__ = ["WanVAE", "WanModel", "T5Encoder", "T5EncoderModel", "CLIPModel", "flash_attention"]
