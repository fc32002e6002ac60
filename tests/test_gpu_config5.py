"""BASELINE config 5 at its real size on one MI355X: the Wan2.1-I2V-14B backbone (wan_i2v_14B.py:26-35: d = 5120,
40 heads x 128, ffn 13 824, 40 layers, 36 input channels, 257 CLIP + 512 text tokens; 16.4 G random-init parameters,
~108 GB of HBM with the packed bf16 copies) on an 81-frame 480x832 clip (latent [16,21,60,104] + 20 conditioning
channels, S = 32 760), as image2video.py:237-337 drives it: VAE encode of the conditioning clip, the
conditional / unconditional forward pair + fused sampler update, VAE decode of all 81 frames.

The fp32 CPU oracle needs ~hours per forward at this size, so the checks are the size-independent ones of
tests/test_gpu_full_size.py (the kernels' arithmetic at this WIDTH is pinned against the oracle by
test_gpu_dit.py::test_wan_14b_width_one_layer, and at this LENGTH by test_gpu_full_size.py): the forward is finite,
repeats bit for bit, a batch of two equals two batches of one bit for bit, depends on its inputs, and the sampler
update on the pair equals the oracle's CFG + UniPC arithmetic on the same two predictions; the VAE round trip of
the 81 frames is finite, clamped, causal-consistent with a shorter prefix, and repeats bit for bit."""
import importlib

import pytest
import torch

from conftest import rel_rms

pytestmark = pytest.mark.gpu
PKG = "omnihuman-1-hack_amd"
F, LT, LH, LW = 81, 21, 60, 104
S = LT * (LH // 2) * (LW // 2)


@pytest.fixture(scope="module")
def i2v14b():
    dev = torch.device("cuda", 0)
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    torch.manual_seed(5)
    with torch.device(dev):
        m = model_mod.WanModel(**cfgs.dit_kwargs(cfgs.i2v_14B, model_type="i2v", in_dim=36))
        torch.nn.init.xavier_uniform_(m.head.head.weight)       # zero-init in the reference: output would be the bias
    m = m.eval().requires_grad_(False)
    assert m.num_layers == 40 and m.dim == 5120 and len(m.blocks) == 40
    yield m
    del m
    torch.cuda.empty_cache()


def test_i2v_14b_full_depth_forward_pair_and_sampler_step(i2v14b):
    m = i2v14b
    dev = torch.device("cuda", 0)
    i2v = importlib.import_module(PKG + ".wan.image2video")
    sched_mod = importlib.import_module(PKG + ".wan.utils.fm_solvers_unipc")
    assert sum(p.numel() for p in m.parameters()) > 16.0e9
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(16, LT, LH, LW, device=dev, generator=g)
    y = torch.cat([i2v.first_frame_mask(F, LH, LW, device=dev), torch.randn(16, LT, LH, LW, device=dev, generator=g)])
    ctx, ctx_null = [torch.randn(120, 4096, device=dev, generator=g)], [torch.randn(40, 4096, device=dev, generator=g)]
    clip_fea = torch.randn(1, 257, 1280, device=dev, generator=g)
    st_c, st_u = m.encode_context(ctx, clip_fea=clip_fea), m.encode_context(ctx_null, clip_fea=clip_fea)
    t = torch.tensor([937.0], device=dev)
    c = m([x], t, st_c, S, y=[y])[0]
    u = m([x], t, st_u, S, y=[y])[0]
    assert c.shape == (16, LT, LH, LW) and c.dtype == torch.float32
    assert bool(torch.isfinite(c).all()) and bool(torch.isfinite(u).all())
    assert 0.05 < float(c.std()) < 50.0                            # a real signal, not a saturated or dead one
    assert rel_rms(c, u) > 1e-3                                    # the text branch reaches the output
    # repeatable bit for bit; the cached context state equals passing the raw context (model.py:531-537)
    assert torch.equal(m([x], t, st_c, S, y=[y])[0], c)
    assert torch.equal(m([x], t, ctx, S, clip_fea=clip_fea, y=[y])[0], c)
    # a batch of two = two batches of one, bit for bit (cond and uncond as ONE forward, different context lengths)
    x2 = torch.randn(16, LT, LH, LW, device=dev, generator=g)
    both = m([x, x2], torch.cat([t, t]), ctx + ctx_null, S, clip_fea=torch.cat([clip_fea, clip_fea]), y=[y, y])
    assert torch.equal(both[0], c)
    assert torch.equal(both[1], m([x2], t, st_u, S, y=[y])[0])
    # the latent matters (not a constant function of the conditioning)
    assert rel_rms(both[1], u) > 1e-2
    # fused CFG + UniPC update on the pair (image2video.py:316-327) against the oracle scheduler's arithmetic
    from oracle import sampler_oracle as SO
    sch = sched_mod.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    sch.set_timesteps(40, device=dev, shift=3.0)
    ref = SO.UniPCOracle(40, 3.0)
    xn = sch.step_cfg(c, u, 5.0, x)
    v = (u + 5.0 * (c - u)).cpu()
    want = ref.step(v, x.cpu())
    assert float((xn.cpu() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))


def test_vae_81_frames_encode_decode_480x832():
    """vae.py:516-568 on the whole 81-frame clip (image2video.py:237-246 encode, :333 decode)."""
    dev = torch.device("cuda", 0)
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    clip = (torch.rand(3, F, 480, 832, device=dev, generator=g) * 2 - 1) * 0.5
    z = vae.encode([clip])[0]
    assert z.shape == (16, LT, LH, LW) and z.dtype == torch.float32 and bool(torch.isfinite(z).all())
    assert torch.equal(vae.encode([clip])[0], z)                                   # not re-entrant, but repeatable
    # causal, and the kernel choice depends on the layer only (not on the frame count of a call): a prefix of the clip
    # encodes to the same latent frames bit for bit
    zp = vae.encode([clip[:, :41]])[0]
    assert zp.shape == (16, 11, LH, LW) and torch.equal(zp, z[:, :11])
    video = vae.decode([z])[0]
    assert video.shape == (3, F, 480, 832) and bool(torch.isfinite(video).all())
    assert float(video.min()) >= -1.0 and float(video.max()) <= 1.0                # vae.py:566 clamp
    assert torch.equal(vae.decode([z])[0], video)
    vp = vae.decode([z[:, :6]])[0]                                                 # ... and a prefix of the latent decodes alike
    assert vp.shape == (3, 21, 480, 832) and torch.equal(vp, video[:, :21])
