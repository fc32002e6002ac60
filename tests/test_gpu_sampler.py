"""The sampler side of the hot path on the GPU: both schedulers' fused CFG+step kernel against the
reference trajectories in tests/golden, and WanT2V.generate (tiny DiT, real VAE) against the oracle."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import rel_rms

pytestmark = pytest.mark.gpu
PKG = "omnihuman-1-hack_amd"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _schedulers():
    u = importlib.import_module(PKG + ".wan.utils")
    return u


@pytest.mark.parametrize("solver", ["unipc", "dpm++"])
def test_scheduler_trajectory_matches_reference(solver):
    """6 steps, shift 3.0, the velocities the golden generator fed the REFERENCE scheduler class."""
    from oracle import detgen
    u = _schedulers()
    tag = "unipc" if solver == "unipc" else "dpmpp"
    g = np.load(os.path.join(GOLD, f"{tag}_6steps.npz"))
    if solver == "unipc":
        s = u.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(6, device="cuda", shift=3.0)
        ts = s.timesteps
    else:
        s = u.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        ts, _ = u.retrieve_timesteps(s, device="cuda", sigmas=u.get_sampling_sigmas(6, 3.0))
    assert np.array_equal(s.sigmas.numpy(), g["sigmas"]) and np.array_equal(ts.cpu().numpy(), g["timesteps"])
    x = torch.from_numpy(detgen.normalish(f"golden/{tag}/x", (1, 16, 2, 6, 8))).cuda()
    for k, t in enumerate(ts):
        v = torch.from_numpy(detgen.normalish(f"golden/{tag}/v{k}", (1, 16, 2, 6, 8))).cuda()
        x = s.step(v, t, x, return_dict=False)[0]                 # reference call form (index found from t)
        assert x.dtype == torch.float32 and x.shape == v.shape
        assert np.abs(x.cpu().numpy() - g["traj"][k]).max() < 2e-5, (solver, k)


@pytest.mark.parametrize("solver", ["unipc", "dpm++"])
def test_step_cfg_equals_step_on_guided_velocity(solver):
    u = _schedulers()
    torch.manual_seed(3)

    def make():
        if solver == "unipc":
            s = u.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
            s.set_timesteps(5, device="cuda", shift=5.0)
        else:
            s = u.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
            u.retrieve_timesteps(s, device="cuda", sigmas=u.get_sampling_sigmas(5, 5.0))
        s.set_begin_index(0)
        return s
    a, b = make(), make()
    xa = xb = torch.randn(16, 3, 6, 8, device="cuda")
    for t in a.timesteps:
        c, un = torch.randn_like(xa), torch.randn_like(xa)
        xa = a.step_cfg(c, un, 5.0, xa)
        xb = b.step(un + 5.0 * (c - un), t, xb, return_dict=False)[0]
        assert float((xa - xb).abs().max()) < 1e-5 * float(xb.abs().max())


@pytest.mark.parametrize("solver", ["unipc", "dpm++"])
def test_wan_t2v_generate_tiny_matches_oracle(solver):
    """WanT2V.generate end to end (noise from the seeded generator, cached contexts, 4 CFG steps, fused
    scheduler kernel) against the oracle DiT + oracle sampler on the same noise (text2video.py:112-269)."""
    from oracle import detgen, sampler_oracle as SO, wan_dit_oracle as O
    wan = importlib.import_module(PKG + ".wan")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    t2v = importlib.import_module(PKG + ".wan.text2video")
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    kw = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=32, freq_dim=64)
    ocfg = O.DiTConfig(**kw)
    sd = O.synth_state_dict(ocfg, "t2vgen")
    model = wan.modules.model.WanModel(**kw)
    model.load_state_dict(sd)
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda", dim=16)
    pipe = t2v.WanT2V(cfgs.t2v_1_3B, checkpoint_dir="", model=model, vae=vae)
    ctx = [torch.from_numpy(detgen.normalish("t2vgen/c", (9, 64)))]
    ctx0 = [torch.from_numpy(detgen.normalish("t2vgen/n", (21, 64)))]
    lat = pipe.generate("", size=(64, 48), frame_num=5, shift=3.0, sample_solver=solver, sampling_steps=4,
                        guide_scale=4.0, seed=11, context=ctx, context_null=ctx0, return_latent=True)
    assert lat.shape == (16, 2, 6, 8) and lat.dtype == torch.float32
    noise = torch.randn(16, 2, 6, 8, dtype=torch.float32, device="cuda",
                        generator=torch.Generator(device="cuda").manual_seed(11)).cpu()

    def vel(x, t):
        tt = torch.stack([t]).float()
        return (O.dit_forward(sd, ocfg, [x], tt, ctx, 24)[0], O.dit_forward(sd, ocfg, [x], tt, ctx0, 24)[0])
    ref = SO.sample_loop(vel, noise, 4, 3.0, 4.0, solver=solver)
    assert rel_rms(lat, ref) < 2.5e-2          # 8 bf16 forwards chained through the sampler (TOL_TINY per forward)
    # cond + uncond as one forward on a batch of two (the default) = two separate forwards, bit for bit
    lat2 = pipe.generate("", size=(64, 48), frame_num=5, shift=3.0, sample_solver=solver, sampling_steps=4,
                         guide_scale=4.0, seed=11, context=ctx, context_null=ctx0, return_latent=True, batched_cfg=False)
    assert torch.equal(lat, lat2)
    vid = pipe.generate("", size=(64, 48), frame_num=5, shift=3.0, sample_solver=solver, sampling_steps=2,
                        guide_scale=4.0, seed=11, context=ctx, context_null=ctx0)
    assert vid.shape == (3, 5, 48, 64) and bool(torch.isfinite(vid).all()) and float(vid.abs().max()) <= 1.0
    with pytest.raises(NotImplementedError):
        pipe.generate("", sample_solver="euler", context=ctx, context_null=ctx0)


def test_wan_i2v_generate_tiny_matches_oracle():
    """WanI2V.generate end to end (image2video.py:129-347): VAE-encoded conditioning clip + first-frame mask as
    ``y``, CLIP tokens through img_emb and the image-token attention, cached contexts, 3 CFG steps; against the
    oracle VAE encoder + oracle i2v DiT + oracle sampler fed the same noise."""
    from oracle import detgen, make_golden, sampler_oracle as SO, wan_dit_oracle as O, wan_vae_oracle as V
    wan = importlib.import_module(PKG + ".wan")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    ocfg = O.DiTConfig(model_type="i2v", in_dim=36, num_layers=2, **make_golden.TINY)
    sd = O.synth_state_dict(ocfg, "i2vgen")
    model = wan.modules.model.WanModel(model_type="i2v", in_dim=36, num_layers=2, **make_golden.TINY)
    model.load_state_dict(sd)
    vcfg = V.VAEConfig(dim=16)
    vsd = V.synth_state_dict(vcfg, "i2vgen/vae")
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda", dim=16)
    vae.model.load_state_dict(vsd)
    clip_fea = torch.from_numpy(detgen.normalish("i2vgen/clip", (1, 257, 1280)))

    class _Clip:                                    # stands in for CLIP ViT-H (out of scope): records its input
        def visual(self, videos):
            self.seen = [tuple(v.shape) for v in videos]
            return clip_fea.cuda()
    clip = _Clip()
    pipe = wan.WanI2V(cfgs.i2v_14B, checkpoint_dir="", model=model, vae=vae, clip=clip)
    img = torch.from_numpy(detgen.uniform("i2vgen/img", (3, 40, 60), 0.0, 1.0))
    ctx = [torch.from_numpy(detgen.normalish("i2vgen/c", (9, 64)))]
    ctx0 = [torch.from_numpy(detgen.normalish("i2vgen/n", (21, 64)))]
    kw = dict(max_area=48 * 64, frame_num=5, shift=3.0, sampling_steps=3, guide_scale=4.0, seed=5, context=ctx,
              context_null=ctx0)
    lat = pipe.generate("", img, return_latent=True, **kw)
    assert clip.seen == [(3, 1, 40, 60)]
    assert torch.equal(lat, pipe.generate("", img, return_latent=True, batched_cfg=False, **kw))
    # ---- the same thing on the oracle
    aspect = 40 / 60
    lat_h = round(np.sqrt(48 * 64 * aspect) // 8 // 2 * 2)
    lat_w = round(np.sqrt(48 * 64 / aspect) // 8 // 2 * 2)
    assert lat.shape == (16, 2, lat_h, lat_w)
    h, w = lat_h * 8, lat_w * 8
    first = torch.nn.functional.interpolate((img[None] - 0.5) / 0.5, size=(h, w), mode="bicubic").transpose(0, 1)
    y_lat = V.vae_encode(vsd, vcfg, torch.concat([first, torch.zeros(3, 4, h, w)], dim=1))
    i2v = importlib.import_module(PKG + ".wan.image2video")
    y = torch.concat([i2v.first_frame_mask(5, lat_h, lat_w), y_lat])
    assert y.shape == (20, 2, lat_h, lat_w) and float(y[:4, 0].min()) == 1.0 and float(y[:4, 1:].max()) == 0.0
    noise = torch.randn(16, 2, lat_h, lat_w, dtype=torch.float32, device="cuda",
                        generator=torch.Generator(device="cuda").manual_seed(5)).cpu()
    seq_len = 2 * lat_h * lat_w // 4

    def vel(x, t):
        tt = torch.stack([t]).float()
        return (O.dit_forward(sd, ocfg, [x], tt, ctx, seq_len, clip_fea=clip_fea, y=[y])[0],
                O.dit_forward(sd, ocfg, [x], tt, ctx0, seq_len, clip_fea=clip_fea, y=[y])[0])
    ref = SO.sample_loop(vel, noise, 3, 3.0, 4.0)
    assert rel_rms(lat, ref) < 3e-2                 # bf16 VAE encoder feeding 6 chained bf16 forwards
    vid = pipe.generate("", img, sample_solver="dpm++", **kw)
    assert vid.shape == (3, 5, h, w) and bool(torch.isfinite(vid).all()) and float(vid.abs().max()) <= 1.0


def _tiny_t2v(rank=0):
    from oracle import detgen, wan_dit_oracle as O
    wan = importlib.import_module(PKG + ".wan")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    t2v = importlib.import_module(PKG + ".wan.text2video")
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    kw = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=32, freq_dim=64)
    model = wan.modules.model.WanModel(**kw)
    model.load_state_dict(O.synth_state_dict(O.DiTConfig(**kw), "t2vgen"))
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda", dim=16)
    pipe = t2v.WanT2V(cfgs.t2v_1_3B, checkpoint_dir="", model=model, vae=vae, rank=rank)
    args = dict(size=(64, 48), frame_num=5, shift=3.0, sampling_steps=4, guide_scale=4.0, return_latent=True,
                context=[torch.from_numpy(detgen.normalish("t2vgen/c", (9, 64)))],
                context_null=[torch.from_numpy(detgen.normalish("t2vgen/n", (21, 64)))])
    return pipe, args


class _ClipStub:
    """Stands in for CLIP ViT-H (out of scope)."""

    def __init__(self, fea):
        self.fea = fea

    def visual(self, videos):
        return self.fea.cuda()


def _tiny_i2v(rank=0):
    from oracle import detgen, make_golden, wan_dit_oracle as O, wan_vae_oracle as V
    wan = importlib.import_module(PKG + ".wan")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    ocfg = O.DiTConfig(model_type="i2v", in_dim=36, num_layers=2, **make_golden.TINY)
    model = wan.modules.model.WanModel(model_type="i2v", in_dim=36, num_layers=2, **make_golden.TINY)
    model.load_state_dict(O.synth_state_dict(ocfg, "i2vgen"))
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda", dim=16)
    vae.model.load_state_dict(V.synth_state_dict(V.VAEConfig(dim=16), "i2vgen/vae"))
    clip = _ClipStub(torch.from_numpy(detgen.normalish("i2vgen/clip", (1, 257, 1280))))
    pipe = wan.WanI2V(cfgs.i2v_14B, checkpoint_dir="", model=model, vae=vae, clip=clip, rank=rank)
    img = torch.from_numpy(detgen.uniform("i2vgen/img", (3, 40, 60), 0.0, 1.0))
    args = dict(max_area=48 * 64, frame_num=5, shift=3.0, sampling_steps=3, guide_scale=4.0, return_latent=True,
                context=[torch.from_numpy(detgen.normalish("i2vgen/c", (9, 64)))],
                context_null=[torch.from_numpy(detgen.normalish("i2vgen/n", (21, 64)))])
    return pipe, img, args


def _cfg_split_worker(rank, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        par = importlib.import_module(PKG + ".parallel")
        pipe, args = _tiny_t2v(rank)
        split = par.CFGPairSplit()
        out = {}
        for solver in ("unipc", "dpm++"):
            got = pipe.generate("", seed=11, sample_solver=solver, cfg_split=split, **args)
            assert (got is None) == (rank != 0)
            if got is not None:
                out[solver] = got.cpu().numpy()
        ipipe, img, iargs = _tiny_i2v(rank)
        got = ipipe.generate("", img, seed=-1, cfg_split=split, **iargs)       # random seed: rank 0's is shared
        got2 = ipipe.generate("", img, seed=5, cfg_split=split, **iargs)
        assert (got is None) == (rank != 0)
        if got2 is not None:
            out["i2v"] = got2.cpu().numpy()
            assert got.shape == got2.shape and bool(torch.isfinite(got).all())
        q.put((rank, "ok", out))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()[-1500:], None))
    finally:
        dist.destroy_process_group()


def test_cfg_pair_split_over_two_processes_equals_single_gpu():
    """one clip's conditional / unconditional forwards on two ranks, one all-gather per step
    (parallel.CFGPairSplit), = the single-GPU generate() bit for bit — WanT2V with both solvers and WanI2V.  The two processes share cuda:0 over gloo
    (RCCL refuses two ranks on one device; the collective is staged through the host there)."""
    import socket
    import torch.multiprocessing as mp
    pipe, args = _tiny_t2v()
    ref = {s: pipe.generate("", seed=11, sample_solver=s, batched_cfg=False, **args).cpu().numpy()
           for s in ("unipc", "dpm++")}
    ipipe, img, iargs = _tiny_i2v()
    ref["i2v"] = ipipe.generate("", img, seed=5, batched_cfg=False, **iargs).cpu().numpy()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfg_split_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == ["ok", "ok"], res
    for s in ("unipc", "dpm++", "i2v"):
        assert np.array_equal(res[0][2][s], ref[s]), s
