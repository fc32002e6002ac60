"""Wan 3D causal VAE (seaweed_apt/wan/modules/vae.py) — gfx950 build in progress."""
import torch

__all__ = ["WanVAE"]


class WanVAE:
    def __init__(self, z_dim=16, vae_pth=None, dtype=torch.float, device="cuda"):
        raise NotImplementedError("WanVAE on gfx950: not built yet in this commit")
