"""Generates the golden vectors under tests/golden/ by running the REAL
reference (imported in place from /root/reference with the shims of
oracle/ref_import.py) on deterministic detgen inputs.  TEST INFRASTRUCTURE.

    python -m oracle.make_golden            # from the repo root, build container only

Only inputs that detgen cannot regenerate and the reference's OUTPUTS are
stored (small .npz files); weights and inputs are regenerated from their names
by the tests.  The reference's source never leaves /root/reference.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import detgen, ref_import, sampler_oracle as SO, wan_dit_oracle as O, wan_vae_oracle as V  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

TINY = dict(dim=256, ffn_dim=512, num_heads=2, text_dim=64, text_len=32, freq_dim=64)


def tiny_case(model_type, layers):
    """Inputs of the tiny DiT cases (shared with tests/test_oracle_golden.py)."""
    cfg = O.DiTConfig(model_type=model_type, in_dim=16 if model_type == "t2v" else 36, num_layers=layers, **TINY)
    tag = f"golden/{model_type}{layers}"
    xs = [torch.from_numpy(detgen.normalish(f"{tag}/x0", (16, 2, 6, 8))),
          torch.from_numpy(detgen.normalish(f"{tag}/x1", (16, 1, 4, 6)))]
    ctx = [torch.from_numpy(detgen.normalish(f"{tag}/c0", (32, 64))),
           torch.from_numpy(detgen.normalish(f"{tag}/c1", (11, 64)))]
    ys = clip = None
    if model_type == "i2v":
        ys = [torch.from_numpy(detgen.normalish(f"{tag}/y0", (20, 2, 6, 8))),
              torch.from_numpy(detgen.normalish(f"{tag}/y1", (20, 1, 4, 6)))]
        clip = torch.from_numpy(detgen.normalish(f"{tag}/clip", (2, 257, 1280)))
    return cfg, tag, xs, ctx, torch.tensor([999., 500.]), 30, ys, clip


def golden_dpmpp():
    os.makedirs(OUT, exist_ok=True)
    # ---- DPM-Solver++ scheduler (sample_solver='dpm++', text2video.py:212-221): 6 steps, shift 3.0
    Dpm, get_sigmas, retrieve = ref_import.load_reference_dpmpp()
    r = Dpm(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    tsteps, _ = retrieve(r, device="cpu", sigmas=get_sigmas(6, 3.0))
    x = torch.from_numpy(detgen.normalish("golden/dpmpp/x", (1, 16, 2, 6, 8)))
    traj = []
    for k, tstep in enumerate(tsteps):
        v = torch.from_numpy(detgen.normalish(f"golden/dpmpp/v{k}", (1, 16, 2, 6, 8)))
        x = r.step(v, tstep, x, return_dict=False)[0]
        traj.append(x.numpy())
    np.savez_compressed(os.path.join(OUT, "dpmpp_6steps.npz"), traj=np.stack(traj), sigmas=r.sigmas.numpy(),
                        timesteps=r.timesteps.numpy())


OMNI_TINY = dict(model_dim=256, num_frames=5, audio_dim=32, pose_keypoints=6)


def omni_state_dict(tag="golden/omni", pose_prefix="pose_guider.", widths=(64, 128), **kw):
    """detgen weights of an ``OmniConditionsModule`` (names/shapes as the reference's, omnihuman_wan_t2v.py:28-51);
    ``pose_prefix='pose_processor.', widths=(128, 256)`` gives the ``OmniHumanWanT2V`` variant (:149-157)."""
    c = dict(OMNI_TINY)
    c.update(kw)
    d, a, k, t = c["model_dim"], c["audio_dim"], c["pose_keypoints"], c["num_frames"]
    w0, w1 = widths
    pp = pose_prefix
    shapes = {"temporal_embed": (1, t, d), "audio_processor.0.weight": (d, a), "audio_processor.0.bias": (d,),
              "audio_processor.2.weight": (d, d), "audio_processor.2.bias": (d,),
              pp + "0.weight": (w0, k, 3, 3, 3), pp + "0.bias": (w0,),
              pp + "2.weight": (w1, w0, 3, 3, 3), pp + "2.bias": (w1,),
              pp + "4.weight": (d // 4, w1, 3, 3, 3), pp + "4.bias": (d // 4,),
              "pose_fc.weight": (d, (d // 4) * 256), "pose_fc.bias": (d,),
              "condition_projector.weight": (d, d), "condition_projector.bias": (d,)}
    sd = {}
    for name, shp in shapes.items():
        fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
        scale = 0.1 if name.endswith("bias") else (1.0 / d ** 0.5 if name == "temporal_embed" else (3.0 / fan_in) ** 0.5)
        sd[name] = torch.from_numpy(detgen.uniform(f"{tag}/{name}", shp, -1.0, 1.0) * np.float32(scale))
    return sd


def omni_inputs(tag="golden/omni", **kw):
    c = dict(OMNI_TINY)
    c.update(kw)
    audio = torch.from_numpy(detgen.normalish(f"{tag}/audio", (2, c["num_frames"], c["audio_dim"])))
    pose = torch.from_numpy(detgen.uniform(f"{tag}/pose", (2, c["pose_keypoints"], c["num_frames"], 64, 64), 0.0, 1.0))
    return audio, pose


def golden_omnihuman():
    """OmniHuman adapters through the reference's own OmniConditionsModule (omnihuman_wan_t2v.py:13-92):
    process_audio as written, and the pose Conv3d stack on [B, K, T, H, W] (process_pose itself raises a
    shape error in the reference for T != C' — recorded in the fixture)."""
    os.makedirs(OUT, exist_ok=True)
    mod = ref_import.load_reference_omnihuman()
    m = mod.OmniConditionsModule(**OMNI_TINY)
    m.load_state_dict(omni_state_dict(), strict=True)
    audio, pose = omni_inputs()
    with torch.no_grad():
        a = m.process_audio(audio)
        pf = m.pose_guider(pose)
        try:
            m.process_pose(pose)
            pose_err = ""
        except RuntimeError as e:
            pose_err = str(e)[:120]
    # the default-schedule DPM-Solver++ the OmniHuman loop uses (scheduler built with shift=1.0, set_timesteps(n))
    Dpm, _, _ = ref_import.load_reference_dpmpp()
    r = Dpm(num_train_timesteps=1000, solver_order=2, prediction_type="flow_prediction", shift=1.0)
    r.set_timesteps(5, device="cpu")
    x = torch.from_numpy(detgen.normalish("golden/omni/x", (1, 16, 2, 6, 8)))
    traj = []
    for k, tstep in enumerate(r.timesteps):
        v = torch.from_numpy(detgen.normalish(f"golden/omni/v{k}", (1, 16, 2, 6, 8)))
        x = r.step(v, tstep, x, return_dict=False)[0]
        traj.append(x.numpy())
    np.savez_compressed(os.path.join(OUT, "omnihuman_adapters.npz"), audio_tokens=a.numpy(), pose_features=pf.numpy(),
                        process_pose_error=np.array(pose_err), dpm_traj=np.stack(traj), dpm_sigmas=r.sigmas.numpy(),
                        dpm_timesteps=r.timesteps.numpy())
    print("omnihuman: audio", tuple(a.shape), "pose feat", tuple(pf.shape), "reference process_pose:", pose_err or "ran")


def golden_train_i2v():
    """Gradients of the REAL reference i2v backbone (tiny, L=2): loss = mse(out[0], v_teacher) as the trainer forms
    it, with CLIP tokens and the conditioning channels y — pins the image branch of the cross-attention
    (k_img / v_img / norm_k_img) and img_emb for the training backward."""
    os.makedirs(OUT, exist_ok=True)
    cfg, tag, xs, ctx, tt, seq_len, ys, clip = tiny_case("i2v", 2)
    sd = O.synth_state_dict(cfg, tag)
    ref = ref_import.build_reference_dit(cfg, sd)
    ref.requires_grad_(True)
    with torch.enable_grad():
        v_teacher = torch.from_numpy(detgen.normalish(f"{tag}/vt", (16, 2, 6, 8)))
        out = ref(xs, torch.tensor([1000.0, 1000.0]), ctx, seq_len, clip_fea=clip, y=ys)
        loss = torch.nn.functional.mse_loss(out[0], v_teacher)
        loss.backward()
    grads = {"loss": np.float32(loss.item())}
    params = dict(ref.named_parameters())
    for name in ("blocks.0.cross_attn.k_img.weight", "blocks.1.cross_attn.v_img.bias", "blocks.0.cross_attn.norm_k_img.weight",
                 "blocks.1.cross_attn.q.weight", "blocks.0.cross_attn.k.weight", "img_emb.proj.0.weight", "img_emb.proj.0.bias",
                 "img_emb.proj.1.weight", "img_emb.proj.3.bias", "img_emb.proj.4.weight", "patch_embedding.weight",
                 "text_embedding.2.weight", "blocks.0.self_attn.v.weight"):
        g = params[name].grad.numpy()
        grads[name] = g if g.size <= 100000 else g[:16]            # large matrices: their first 16 rows
    np.savez_compressed(os.path.join(OUT, "dit_train_i2v_L2.npz"), **grads)
    print("i2v train loss", loss.item(), {k: float(np.abs(v).mean()) for k, v in grads.items() if k != "loss"})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(False)
    model_mod, vae_mod = ref_import.load_reference()

    # ---- per-op fixtures
    ops = {}
    t = torch.tensor([0., 1., 999., 1000.])
    ops["sinusoid"] = model_mod.sinusoidal_embedding_1d(256, t).float().numpy()
    d = 128
    freqs = torch.cat([model_mod.rope_params(1024, d - 4 * (d // 6)), model_mod.rope_params(1024, 2 * (d // 6)),
                       model_mod.rope_params(1024, 2 * (d // 6))], dim=1)
    xq = torch.from_numpy(detgen.normalish("golden/rope/x", (2, 30, 2, 128)))
    ops["rope"] = model_mod.rope_apply(xq, torch.tensor([[2, 3, 4], [1, 5, 6]]), freqs).numpy()
    rn = model_mod.WanRMSNorm(256, eps=1e-6)
    rn.weight.data = torch.from_numpy(1.0 + detgen.uniform("golden/rms/w", (256,), -0.2, 0.2))
    xr = torch.from_numpy(detgen.normalish("golden/rms/x", (3, 7, 256)))
    ops["rmsnorm"] = rn(xr).numpy()
    ln = model_mod.WanLayerNorm(256, eps=1e-6)
    ops["layernorm"] = ln(xr * 3 + 0.5).numpy()
    np.savez_compressed(os.path.join(OUT, "dit_ops.npz"), **ops)

    # ---- tiny DiT models: t2v L=2, L=13 (crosses the block_idx>10 branch), i2v L=2
    for mt, layers in (("t2v", 2), ("t2v", 13), ("i2v", 2)):
        cfg, tag, xs, ctx, tt, seq_len, ys, clip = tiny_case(mt, layers)
        sd = O.synth_state_dict(cfg, tag)
        ref = ref_import.build_reference_dit(cfg, sd)
        out = ref(xs, tt, ctx, seq_len, clip_fea=clip, y=ys)
        np.savez_compressed(os.path.join(OUT, f"dit_{mt}_L{layers}.npz"), out0=out[0].numpy(), out1=out[1].numpy())
        print("dit", mt, layers, float(out[0].abs().mean()))

    # ---- trainer step (distilled_trainer.py:241-316 semantics) on the tiny t2v L=13 model:
    #      loss = mse(v_student[0], v_teacher) (sample 0 only, broadcast) and selected gradients
    cfg, tag, xs, ctx, tt, seq_len, _, _ = tiny_case("t2v", 13)
    sd = O.synth_state_dict(cfg, tag)
    ref = ref_import.build_reference_dit(cfg, sd)
    ref.requires_grad_(True)
    with torch.enable_grad():
        noise = torch.stack([xs[0], torch.from_numpy(detgen.normalish(f"{tag}/x0b", (16, 2, 6, 8)))])
        v_teacher = torch.from_numpy(detgen.normalish(f"{tag}/vt", (2, 16, 2, 6, 8)))
        cl = [ctx[0], torch.from_numpy(detgen.normalish(f"{tag}/c0b", (32, 64)))]
        out = ref(noise, torch.ones(2) * 1000.0, cl, 24)
        loss = torch.nn.functional.mse_loss(out[0], v_teacher)
        loss.backward()
    grads = {"loss": np.float32(loss.item())}
    for name in ("blocks.0.self_attn.q.weight", "blocks.0.self_attn.norm_k.weight", "blocks.10.ffn.0.weight",
                 "blocks.12.cross_attn.v.bias", "blocks.5.modulation", "patch_embedding.weight",
                 "head.head.weight", "time_projection.1.bias", "text_embedding.0.weight", "blocks.3.norm3.weight"):
        g = dict(ref.named_parameters())[name].grad
        grads[name] = g.numpy()
    grads["ffn_grad_none_from"] = np.int32(min(i for i in range(13) if ref.blocks[i].ffn[0].weight.grad is None))
    np.savez_compressed(os.path.join(OUT, "dit_train_t2v_L13.npz"), **grads)
    print("train loss", loss.item(), "first block without FFN grad:", grads["ffn_grad_none_from"])

    # ---- VAE: whole-model decode / encode at dim=16 and dim=96, plus a causal-conv-with-cache case
    mean, std = torch.tensor(V.LATENT_MEAN), torch.tensor(V.LATENT_STD)
    scale = [mean, 1.0 / std]
    z = torch.from_numpy(detgen.normalish("golden/vae/z", (16, 3, 8, 8)))
    vid = torch.from_numpy(detgen.uniform("golden/vae/vid", (3, 9, 32, 32)))
    for dim in (16, 96):
        cfgv = V.VAEConfig(dim=dim)
        sdv = V.synth_state_dict(cfgv, f"golden/vae{dim}")
        ref = ref_import.build_reference_vae(sdv, dim=dim)
        dec = ref.decode(z.unsqueeze(0), scale).float().clamp_(-1, 1).squeeze(0)
        enc = ref.encode(vid.unsqueeze(0), scale).float().squeeze(0)
        np.savez_compressed(os.path.join(OUT, f"vae_dim{dim}.npz"), decode=dec.numpy().astype(np.float16)
                            if dim == 96 else dec.numpy(), encode=enc.numpy())
        print("vae", dim, float(dec.abs().mean()), float(enc.abs().mean()))

    # ---- UniPC scheduler: trajectory of 6 steps, shift 3.0
    Ref = ref_import.load_reference_unipc()
    r = Ref(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    r.set_timesteps(6, device="cpu", shift=3.0)
    x = torch.from_numpy(detgen.normalish("golden/unipc/x", (1, 16, 2, 6, 8)))
    traj = []
    for k, tstep in enumerate(r.timesteps):
        v = torch.from_numpy(detgen.normalish(f"golden/unipc/v{k}", (1, 16, 2, 6, 8)))
        x = r.step(v, tstep, x, return_dict=False)[0]
        traj.append(x.numpy())
    np.savez_compressed(os.path.join(OUT, "unipc_6steps.npz"), traj=np.stack(traj), sigmas=r.sigmas.numpy(),
                        timesteps=r.timesteps.numpy())

    golden_dpmpp()

    # ---- full-size Wan2.1-T2V-1.3B forward (config 1): checksums + 64 probe elements
    if os.environ.get("OMH_GOLDEN_FULL", "1") == "1":
        cfg = O.DiTConfig.wan_t2v_1_3b()
        sd = O.synth_state_dict(cfg, "wan1.3b")
        ref = ref_import.build_reference_dit(cfg, sd)
        noise = torch.from_numpy(detgen.normalish("c1/noise", (16, 1, 60, 104)))
        cneg = torch.from_numpy(detgen.normalish("c1/neg", (37, 4096)))
        out = ref([noise], torch.tensor([999.]), [cneg], 1560)[0]
        idx = (np.arange(64) * 1559 + 7) % out.numel()
        np.savez_compressed(os.path.join(OUT, "dit_wan1_3b_c1.npz"), probe_idx=idx,
                            probe=out.flatten()[idx].numpy(), mean=np.float64(out.double().mean()),
                            abs_mean=np.float64(out.double().abs().mean()),
                            coarse=out[:, 0, ::6, ::8].numpy())
        print("1.3B", float(out.abs().mean()))


def encoder_cases():
    """Shared by the generator and the tests: tiny umT5 / ViT configurations and their inputs."""
    from oracle import encoders_oracle as E
    tc, vc = E.T5Config.tiny(), E.ViTConfig.tiny()
    ids = torch.from_numpy((detgen.uniform("golden/enc/ids", (2, 24), 0.0, 1.0) * tc.vocab).astype(np.int64)).clamp_(0, tc.vocab - 1)
    mask = torch.ones(2, 24, dtype=torch.long)
    mask[1, 15:] = 0
    img = torch.from_numpy(detgen.normalish("golden/enc/img", (2, 3, vc.image_size, vc.image_size)))
    return tc, vc, ids, mask, img


def golden_encoders():
    """umT5 encoder (with the repository's cut-down block) and the CLIP vision tower of the REAL reference at a small
    width -> tests/golden/encoders_t5_clip.npz."""
    from oracle import encoders_oracle as E
    t5, clip = ref_import.load_reference_encoders()
    tc, vc, ids, mask, img = encoder_cases()
    enc = t5.T5Encoder(vocab=tc.vocab, dim=tc.dim, dim_attn=tc.dim_attn, dim_ffn=tc.dim_ffn, num_heads=tc.num_heads,
                       num_layers=tc.num_layers, num_buckets=tc.num_buckets, shared_pos=tc.shared_pos, dropout=0.1).eval()
    enc.load_state_dict(E.t5_state_dict(tc, "golden/t5"), strict=True)
    t5_out = enc(ids, mask).float()
    vit = clip.VisionTransformer(image_size=vc.image_size, patch_size=vc.patch_size, dim=vc.dim, mlp_ratio=vc.mlp_ratio,
                                 out_dim=32, num_heads=vc.num_heads, num_layers=vc.num_layers, pool_type="token",
                                 pre_norm=True, post_norm=False, activation="gelu", norm_eps=vc.norm_eps).eval()
    vit.load_state_dict(E.vit_state_dict(vc, "golden/vit"), strict=True)
    vit_out = vit(img, use_31_block=True).float()
    bk = t5.T5RelativeEmbedding(32, 4, bidirectional=True)._relative_position_bucket(
        torch.arange(40)[None, :] - torch.arange(40)[:, None])
    np.savez_compressed(os.path.join(OUT, "encoders_t5_clip.npz"), t5=t5_out.numpy(), vit=vit_out.numpy(),
                        buckets=bk.numpy().astype(np.int16))
    print("encoders", float(t5_out.abs().mean()), float(vit_out.abs().mean()))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "encoders":
    torch.set_grad_enabled(False)
    golden_encoders()
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "train_i2v":
    golden_train_i2v()
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "omnihuman":
    torch.set_grad_enabled(False)
    golden_omnihuman()
    sys.exit(0)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dpmpp":        # regenerate just that fixture
        torch.set_grad_enabled(False)
        golden_dpmpp()
    else:
        main()
