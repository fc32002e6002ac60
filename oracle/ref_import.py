"""Shimmed import of the REAL reference modules (TEST INFRASTRUCTURE).

Only usable where ``/root/reference`` exists (the build container); the GPU
box never has it, so nothing run there may call :func:`load_reference`.
Used by oracle/make_golden.py and by CPU tests that skip when the reference is
absent.  The reference is read in place — nothing is copied.

Shims:
 1. stub ``diffusers.configuration_utils`` / ``diffusers.models.modeling_utils``
    (not installed; model.py:7-8 imports them at top level);
 2. pre-register empty ``wan`` / ``wan.modules`` packages so their
    ``__init__`` (torchvision, ftfy, ...) is bypassed;
 3. ``flash_attention`` (needs CUDA + flash_attn, attention.py:54,112) is
    rebound to a masked fp32 softmax attention — mathematically what the
    varlen kernel computes.
"""
import importlib
import os
import sys
import tempfile
import types

import torch

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "seaweed_apt", "wan", "modules"))


def _masked_sdpa(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None,
                 q_scale=None, causal=False, window_size=(-1, -1), deterministic=False,
                 dtype=torch.bfloat16, version=None):
    assert q_lens is None and not causal and q_scale is None
    B, Lq, N, D = q.shape
    Lk = k.shape[1]
    mask = None
    if k_lens is not None:
        mask = (torch.arange(Lk)[None, :] < k_lens.view(B, 1))[:, None, None, :]
    out = torch.nn.functional.scaled_dot_product_attention(
        q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2),
        attn_mask=mask, scale=softmax_scale)
    return out.transpose(1, 2).contiguous().type(q.dtype)


_CACHE = {}


def load_reference():
    """Returns (model_module, vae_module) = the reference's wan.modules.model / .vae."""
    if "mods" in _CACHE:
        return _CACHE["mods"]
    if not reference_available():
        raise RuntimeError("reference tree not present")
    sw = os.path.join(REFERENCE_ROOT, "seaweed_apt")
    # the reference's logger writes project.log into the cwd: import from a temp dir
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="omh_ref_")
    os.chdir(tmp)
    try:
        if sw not in sys.path:
            sys.path.insert(0, sw)
        dcfg = types.ModuleType("diffusers.configuration_utils")

        class ConfigMixin:  # noqa: D401 - stub
            pass

        def register_to_config(fn):
            return fn

        dcfg.ConfigMixin, dcfg.register_to_config = ConfigMixin, register_to_config
        dmod = types.ModuleType("diffusers.models.modeling_utils")
        dmod.ModelMixin = torch.nn.Module
        for name, mod in (("diffusers", types.ModuleType("diffusers")),
                          ("diffusers.models", types.ModuleType("diffusers.models")),
                          ("diffusers.configuration_utils", dcfg),
                          ("diffusers.models.modeling_utils", dmod)):
            sys.modules.setdefault(name, mod)
        wan = types.ModuleType("wan")
        wan.__path__ = [os.path.join(sw, "wan")]
        wmods = types.ModuleType("wan.modules")
        wmods.__path__ = [os.path.join(sw, "wan", "modules")]
        sys.modules.setdefault("wan", wan)
        sys.modules.setdefault("wan.modules", wmods)
        model = importlib.import_module("wan.modules.model")
        vae = importlib.import_module("wan.modules.vae")
        model.flash_attention = _masked_sdpa
    finally:
        os.chdir(cwd)
    _CACHE["mods"] = (model, vae)
    return model, vae


def build_reference_dit(cfg, state_dict):
    """Instantiate the reference WanModel with ``cfg`` (an oracle DiTConfig)
    and load ``state_dict`` into it."""
    model_mod, _ = load_reference()
    m = model_mod.WanModel(
        model_type=cfg.model_type, patch_size=cfg.patch_size, text_len=cfg.text_len,
        in_dim=cfg.in_dim, dim=cfg.dim, ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim,
        text_dim=cfg.text_dim, out_dim=cfg.out_dim, num_heads=cfg.num_heads,
        num_layers=cfg.num_layers, qk_norm=cfg.qk_norm, cross_attn_norm=cfg.cross_attn_norm,
        eps=cfg.eps, use_checkpoint=False)
    missing, unexpected = m.load_state_dict(state_dict, strict=True)
    torch.cuda.empty_cache = lambda: None  # model.py:503 calls it on every forward
    return m.eval().requires_grad_(False)


def build_reference_vae(state_dict, **cfg):
    _, vae_mod = load_reference()
    base = dict(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                temperal_downsample=[False, True, True], dropout=0.0)
    base.update(cfg)
    m = vae_mod.WanVAE_(**base)
    m.load_state_dict(state_dict, strict=True)
    return m.eval().requires_grad_(False)


def load_reference_unipc():
    """The reference FlowUniPCMultistepScheduler, importable with a small stub of
    the diffusers base classes it inherits from."""
    if "unipc" in _CACHE:
        return _CACHE["unipc"]
    load_reference()
    import functools
    import inspect

    class _Cfg(dict):
        __getattr__ = dict.__getitem__

    def register_to_config(init):
        @functools.wraps(init)
        def wrapped(self, *a, **k):
            sig = inspect.signature(init)
            bound = sig.bind(self, *a, **k)
            bound.apply_defaults()
            self.config = _Cfg({n: v for n, v in bound.arguments.items() if n != "self"})
            init(self, *a, **k)
        return wrapped

    class ConfigMixin:
        def register_to_config(self, **kw):
            self.config.update(kw)

    dcfg = sys.modules["diffusers.configuration_utils"]
    dcfg.ConfigMixin, dcfg.register_to_config = ConfigMixin, register_to_config
    su = types.ModuleType("diffusers.schedulers.scheduling_utils")

    class SchedulerMixin:
        pass

    class SchedulerOutput:
        def __init__(self, prev_sample):
            self.prev_sample = prev_sample

    su.SchedulerMixin, su.SchedulerOutput, su.KarrasDiffusionSchedulers = SchedulerMixin, SchedulerOutput, []
    du = types.ModuleType("diffusers.utils")
    du.deprecate = lambda *a, **k: None
    du.is_scipy_available = lambda: False
    sys.modules["diffusers.schedulers"] = types.ModuleType("diffusers.schedulers")
    sys.modules["diffusers.schedulers.scheduling_utils"] = su
    sys.modules["diffusers.utils"] = du
    wutils = types.ModuleType("wan.utils")
    wutils.__path__ = [os.path.join(REFERENCE_ROOT, "seaweed_apt", "wan", "utils")]
    sys.modules.setdefault("wan.utils", wutils)
    tu = types.ModuleType("diffusers.utils.torch_utils")
    tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(
        shape, generator=generator, device=device, dtype=dtype)
    sys.modules["diffusers.utils.torch_utils"] = tu
    mod = importlib.import_module("wan.utils.fm_solvers_unipc")
    _CACHE["unipc"] = mod.FlowUniPCMultistepScheduler
    return _CACHE["unipc"]


def load_reference_dpmpp():
    """(FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps) of the reference
    (seaweed_apt/wan/utils/fm_solvers.py), through the same diffusers base-class stub."""
    if "dpmpp" not in _CACHE:
        load_reference_unipc()
        mod = importlib.import_module("wan.utils.fm_solvers")
        _CACHE["dpmpp"] = (mod.FlowDPMSolverMultistepScheduler, mod.get_sampling_sigmas, mod.retrieve_timesteps)
    return _CACHE["dpmpp"]


def load_reference_omnihuman():
    """The reference's ``Omnihuman/omnihuman_wan_t2v.py`` module (for ``OmniConditionsModule`` and the unbound
    methods of ``OmniHumanWanT2V``), imported in place.  Its top-level imports that are unavailable here are
    stubbed: ``omegaconf`` (DictConfig = dict), ``wan.WanT2V`` / ``wan.modules.t5`` / ``wan.configs`` (T5, tokenizers
    and easydict are outside this path and not installed).  Nothing the golden vectors exercise runs through a stub."""
    if "omni" in _CACHE:
        return _CACHE["omni"]
    load_reference_dpmpp()
    import importlib.util
    wan = sys.modules["wan"]
    if not hasattr(wan, "WanT2V"):
        wan.WanT2V = type("WanT2V", (), {})
    t5 = types.ModuleType("wan.modules.t5")
    t5.T5EncoderModel = type("T5EncoderModel", (), {})
    sys.modules.setdefault("wan.modules.t5", t5)
    cfgs = types.ModuleType("wan.configs")
    cfgs.t2v_14B = {}
    sys.modules.setdefault("wan.configs", cfgs)
    oc = types.ModuleType("omegaconf")
    oc.DictConfig, oc.OmegaConf = dict, type("OmegaConf", (), {})
    sys.modules.setdefault("omegaconf", oc)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="omh_ref_")
    os.chdir(tmp)
    try:
        spec = importlib.util.spec_from_file_location(
            "ref_omnihuman_wan_t2v", os.path.join(REFERENCE_ROOT, "Omnihuman", "omnihuman_wan_t2v.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        os.chdir(cwd)
    _CACHE["omni"] = mod
    return mod


def load_reference_encoders():
    """(t5_module, clip_module) = the reference's ``wan.modules.t5`` / ``wan.modules.clip``, imported in place.
    Shims: ``ftfy`` (text cleaning of the tokenizer wrapper — tokenisation is outside this path) and
    ``torchvision.transforms`` (only ``Normalize`` is ever called, clip.py:537) are stubbed; ``t5.py:487`` evaluates
    ``torch.cuda.current_device()`` in a default argument at class-definition time, so that call is answered with 0
    during the import; ``flash_attention`` in clip.py is rebound to the masked SDPA used for model.py."""
    if "enc" in _CACHE:
        return _CACHE["enc"]
    load_reference()
    if not hasattr(sys.modules.get("wan.modules.t5"), "T5Encoder"):
        sys.modules.pop("wan.modules.t5", None)                       # load_reference_omnihuman registers a stub
    f = types.ModuleType("ftfy")
    f.fix_text = lambda s: s
    sys.modules.setdefault("ftfy", f)
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")

    class _Compose:
        def __init__(self, ts):
            self.transforms = list(ts)

    class _Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)

        def __call__(self, x):
            return (x - self.mean.to(x)) / self.std.to(x)

    tvt.Compose, tvt.Normalize = _Compose, _Normalize
    tvt.Resize = tvt.ToTensor = lambda *a, **k: None
    tvt.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
    tv.transforms = tvt
    import importlib.machinery
    for m_ in (f, tv, tvt):                                          # transformers probes find_spec() on its imports
        m_.__spec__ = importlib.machinery.ModuleSpec(m_.__name__, None)
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="omh_ref_"))
    cd = torch.cuda.current_device
    torch.cuda.current_device = lambda: 0
    try:
        t5 = importlib.import_module("wan.modules.t5")
        clip = importlib.import_module("wan.modules.clip")
    finally:
        torch.cuda.current_device = cd
        os.chdir(cwd)
    clip.flash_attention = _masked_sdpa
    _CACHE["enc"] = (t5, clip)
    return t5, clip
