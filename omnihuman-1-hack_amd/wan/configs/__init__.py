"""Model / pipeline constants of the reference (seaweed_apt/wan/configs/*.py),
restated as plain attribute dicts (easydict is not a dependency here)."""
import torch


class _Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _shared():
    # shared_config.py:7-19
    return _Cfg(t5_model="umt5_xxl", t5_dtype=torch.bfloat16, text_len=512, param_dtype=torch.bfloat16,
                num_train_timesteps=1000, sample_fps=16,
                sample_neg_prompt="色调艳丽，过曝，静态，细节模糊不清，字幕，风格，作品，画作，画面，静止，整体发灰，最差质量，"
                                  "低质量，JPEG压缩残留，丑陋的，残缺的，多余的手指，画得不好的手部，画得不好的脸部，畸形的，"
                                  "毁容的，形态畸形的肢体，手指融合，静止不动的画面，杂乱的背景，三条腿，背景人很多，倒着走")


# wan_t2v_1_3B.py:9-29
t2v_1_3B = _shared()
t2v_1_3B.update(__name__="Config: Wan T2V 1.3B", t5_checkpoint="models_t5_umt5-xxl-enc-bf16.pth",
                t5_tokenizer="google/umt5-xxl", vae_checkpoint="Wan2.1_VAE.pth", vae_stride=(4, 8, 8),
                patch_size=(1, 2, 2), dim=1536, ffn_dim=8960, freq_dim=256, num_heads=12, num_layers=30,
                window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6)

# wan_t2v_14B.py
t2v_14B = _shared()
t2v_14B.update(__name__="Config: Wan T2V 14B", t5_checkpoint="models_t5_umt5-xxl-enc-bf16.pth",
               t5_tokenizer="google/umt5-xxl", vae_checkpoint="Wan2.1_VAE.pth", vae_stride=(4, 8, 8),
               patch_size=(1, 2, 2), dim=5120, ffn_dim=13824, freq_dim=256, num_heads=40, num_layers=40,
               window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6)

# wan_i2v_14B.py:9-35
i2v_14B = _shared()
i2v_14B.update(__name__="Config: Wan I2V 14B", t5_checkpoint="models_t5_umt5-xxl-enc-bf16.pth",
               t5_tokenizer="google/umt5-xxl", clip_model="clip_xlm_roberta_vit_h_14", clip_dtype=torch.float16,
               clip_checkpoint="models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth",
               clip_tokenizer="xlm-roberta-large", vae_checkpoint="Wan2.1_VAE.pth", vae_stride=(4, 8, 8),
               patch_size=(1, 2, 2), dim=5120, ffn_dim=13824, freq_dim=256, num_heads=40, num_layers=40,
               window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6)

WAN_CONFIGS = {"t2v-14B": t2v_14B, "t2v-1.3B": t2v_1_3B, "i2v-14B": i2v_14B}
SIZE_CONFIGS = {"720*1280": (720, 1280), "1280*720": (1280, 720), "480*832": (480, 832), "832*480": (832, 480)}
MAX_AREA_CONFIGS = {"720*1280": 720 * 1280, "1280*720": 1280 * 720, "480*832": 480 * 832, "832*480": 832 * 480}
SUPPORTED_SIZES = {"t2v-14B": ("720*1280", "1280*720", "480*832", "832*480"), "t2v-1.3B": ("480*832", "832*480"),
                   "i2v-14B": ("720*1280", "1280*720", "480*832", "832*480")}


def dit_kwargs(cfg, model_type="t2v", in_dim=16):
    """WanModel constructor arguments for a config above."""
    return dict(model_type=model_type, patch_size=cfg.patch_size, text_len=cfg.text_len, in_dim=in_dim, dim=cfg.dim,
                ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim, text_dim=4096, out_dim=16, num_heads=cfg.num_heads,
                num_layers=cfg.num_layers, window_size=cfg.window_size, qk_norm=cfg.qk_norm,
                cross_attn_norm=cfg.cross_attn_norm, eps=cfg.eps)
