"""OmniHuman conditioning path (BASELINE config 4; Omnihuman/omnihuman_wan_t2v.py) on the HIP kernels against
oracle/omnihuman_oracle.py and the vectors produced by the reference's own OmniConditionsModule."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import rel_rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp32 dense kernels: accumulation order only.  bf16 conv stack (3 layers, bf16 activations): 3 roundings.
TOL_F32, TOL_CONV = 2e-5, 1.5e-2


@pytest.fixture(scope="module")
def omni(omh):
    return importlib.import_module("omnihuman-1-hack_amd.omnihuman_wan_t2v")


def test_relu_kernel(ops):
    x = torch.randn(4099 * 8, device="cuda").bfloat16()
    x[5] = -0.0
    want = torch.relu(x.float()).bfloat16()
    got = ops.relu_bf16_(x.clone())
    assert torch.equal(got, want) and not torch.signbit(got.float()).any()


def test_adapters_match_reference_vectors_and_oracle(omni):
    from oracle import make_golden, omnihuman_oracle as OH
    g = np.load(os.path.join(GOLD, "omnihuman_adapters.npz"))
    sd = make_golden.omni_state_dict()
    audio, pose = make_golden.omni_inputs()
    m = omni.OmniConditionsModule(**make_golden.OMNI_TINY)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    a = m.process_audio(audio.cuda())
    assert rel_rms(a, torch.from_numpy(g["audio_tokens"])) < TOL_F32          # the reference's own output
    feat = m.pose_features(pose.cuda())
    assert tuple(feat.shape) == (2, 64, 5, 16, 16)
    assert rel_rms(feat, torch.from_numpy(g["pose_features"])) < TOL_CONV      # the reference's own Conv3d stack
    tok = m.process_pose(pose.cuda())
    want = OH.process_pose(sd, pose, prefix="pose_guider.")
    assert tuple(tok.shape) == (2, 5, 256) and rel_rms(tok, want) < 2 * TOL_CONV
    ct = m.condition_tokens(a, tok)
    assert rel_rms(ct, OH.condition_tokens(sd, OH.process_audio(sd, audio), want)) < 2 * TOL_CONV
    # exact-input check of the projector alone
    ct2 = m.condition_tokens(OH.process_audio(sd, audio).cuda(), want.cuda())
    assert rel_rms(ct2, OH.condition_tokens(sd, OH.process_audio(sd, audio), want)) < TOL_F32
    out = m(audio=audio.cuda(), pose=pose.cuda())
    assert set(out) == {"audio", "pose", "temporal"} and tuple(out["temporal"].shape) == (2, 5, 256)


def test_dit_forward_with_condition_tokens(wan_model_mod):
    """WanModel.forward(extra_conditions=...) — the call of omnihuman_wan_t2v.py:408-414 — against the oracle:
    tokens prepended to the text context, masked text padding behind them still excluded."""
    from oracle import detgen, make_golden, wan_dit_oracle as O
    cfg, tag, xs, ctx, tt, seq_len, _, _ = make_golden.tiny_case("t2v", 2)
    sd = O.synth_state_dict(cfg, tag)
    extra = torch.from_numpy(detgen.normalish("omni/extra", (2, 13, 256)))
    want = O.dit_forward(sd, cfg, xs, tt, ctx, seq_len, extra_tokens=extra)
    base = O.dit_forward(sd, cfg, xs, tt, ctx, seq_len)
    m = wan_model_mod.WanModel(num_layers=2, **make_golden.TINY)
    m.load_state_dict(sd)
    m = m.cuda().eval().requires_grad_(False)
    x, c = [u.cuda() for u in xs], [u.cuda() for u in ctx]
    got = m(x, tt.cuda(), c, seq_len, extra_conditions={"tokens": extra.cuda()})
    st = m.encode_context(c, extra_conditions=extra.cuda())
    got2 = m(x, tt.cuda(), st, seq_len)
    for o, o2, w, b in zip(got, got2, want, base):
        assert rel_rms(o, w) < 8e-3 and torch.equal(o, o2)
        assert rel_rms(w, b) > 5e-2                     # the tokens matter: this is not the plain forward
    with pytest.raises(ValueError):
        m(x, tt.cuda(), c, seq_len, extra_conditions=extra[:1].cuda())


class _StubVAE:
    """process_reference / the final decode only move data here; the VAE itself is covered by test_gpu_vae.py."""

    class model:
        z_dim = 16

    def __init__(self, ref):
        self.ref = ref

    def encode(self, xs):
        return [self.ref.to(xs[0].device)]

    def decode(self, zs):
        return [z * 1.0 for z in zs]


class _StubT2V:
    def __init__(self, model, vae):
        self.model, self.vae, self.text_encoder = model, vae, None


def test_omnihuman_sampling_loop_matches_oracle(omni, wan_model_mod):
    """The whole conditioning path: adapters -> condition tokens -> reference-latent concat -> uncond / cond DiT
    forwards -> annealed CFG -> DPM-Solver++ (default schedule), 3 steps, against oracle/omnihuman_oracle.sample."""
    from oracle import detgen, make_golden, omnihuman_oracle as OH, wan_dit_oracle as O
    cfg = O.DiTConfig(model_type="t2v", in_dim=16, num_layers=2, **make_golden.TINY)
    dsd = O.synth_state_dict(cfg, "omni/dit")
    dsd["head.head.weight"] = torch.from_numpy(detgen.uniform("omni/dit/headw", tuple(dsd["head.head.weight"].shape),
                                                              -0.08, 0.08))
    kw = dict(model_dim=256, num_frames=5, audio_dim=32, pose_keypoints=6)
    osd = make_golden.omni_state_dict("omni/sd", pose_prefix="pose_processor.", widths=(128, 256), **kw)
    audio, pose = make_golden.omni_inputs("omni/in", **kw)
    audio, pose = audio[:1], pose[:1]
    noise = torch.from_numpy(detgen.normalish("omni/noise", (16, 2, 4, 6)))
    ref = torch.from_numpy(detgen.normalish("omni/ref", (16, 1, 4, 6)))
    ctx = torch.from_numpy(detgen.normalish("omni/ctx", (20, 64)))
    ctx_null = torch.from_numpy(detgen.normalish("omni/ctxn", (7, 64)))
    want = OH.sample(dsd, cfg, osd, noise, ctx, ctx_null, reference_latent=ref, audio=audio, pose=pose,
                     num_inference_steps=3, cfg_scale=7.5)
    plain = OH.sample(dsd, cfg, osd, noise, ctx, ctx_null, reference_latent=ref, num_inference_steps=3, cfg_scale=7.5)
    dit = wan_model_mod.WanModel(num_layers=2, **make_golden.TINY)
    dit.load_state_dict(dsd)
    dit = dit.cuda().eval().requires_grad_(False)
    m = omni.OmniHumanWanT2V(dict(num_frames=5, num_keypoints=6, model_dim=256, audio_dim=32), device_id=0,
                             wan_t2v=_StubT2V(dit, _StubVAE(ref)))
    missing, unexpected = m.load_state_dict(osd, strict=False)
    assert not unexpected and all(k.startswith("wan_t2v") for k in missing), (missing, unexpected)
    got = m(audio=audio, pose=pose, reference_image=torch.zeros(3, 1, 32, 48), num_inference_steps=3, cfg_scale=7.5,
            text_context=ctx, text_context_null=ctx_null, noise=noise, return_latent=True)
    assert tuple(got.shape) == (16, 2, 4, 6)
    assert rel_rms(got, want) < 2e-2
    assert rel_rms(want, plain) > 2e-2                  # audio / pose conditioning changes the sample
    video = m(reference_image=torch.zeros(3, 1, 32, 48), num_inference_steps=2, text_context=ctx,
              text_context_null=ctx_null, noise=noise)
    assert tuple(video.shape) == (16, 2, 4, 6) and torch.isfinite(video).all()


def test_relu_backward_kernel(ops):
    g = torch.Generator(device="cuda").manual_seed(4)
    y = torch.relu(torch.randn(4099 * 8, device="cuda", generator=g)).bfloat16()
    y[7] = -0.0
    dy = torch.randn(4099 * 8, device="cuda", generator=g).bfloat16()
    got = ops.relu_bwd_bf16(dy, y)
    assert torch.equal(got, torch.where(y.float() > 0, dy, torch.zeros_like(dy)))


class _GateSpy:
    """Records the on/off decisions of the product's three pose Conv3d + ReLU layers (bf16 forward), per sample, so the
    fp32 oracle can differentiate the SAME piecewise-linear function (test_pose_conv_gradients_with_the_products_relu_
    gates shows that the 0.1 % of decisions that differ are all of the 5-9 % gradient gap)."""

    def __init__(self, omni, monkeypatch):
        self.rec, orig = [], omni._conv3d_relu_fwd

        def spy(x_cl, conv, stride_hw):
            y = orig(x_cl, conv, stride_hw)
            self.rec.append((conv.out_channels, y.detach()))
            return y
        monkeypatch.setattr(omni, "_conv3d_relu_fwd", spy)

    def gates(self, B):
        assert len(self.rec) >= 3 * B
        rec = self.rec[-3 * B:]
        out = []
        for layer in range(3):
            out.append(torch.stack([(rec[3 * b + layer][1][..., :rec[3 * b + layer][0]].float() > 0).permute(3, 0, 1, 2).cpu()
                                    for b in range(B)]))
        return out


def test_adapter_gradients_match_autograd(omni, monkeypatch):
    """Backward of every adapter layer (omh_dense_f32_bwd, the Conv3d dgrad / wgrad on omh_conv_cl_bf16 and
    omh_gemm_bf16_tn, pose_fc on the GEMMs) against autograd through the oracle's fp32 formulas
    (oracle/omnihuman_oracle.py, whose forward is pinned to the reference's OmniConditionsModule)."""
    from oracle import detgen, make_golden, omnihuman_oracle as OH
    sd = make_golden.omni_state_dict()
    audio, pose = make_golden.omni_inputs()
    m = omni.OmniConditionsModule(**make_golden.OMNI_TINY)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    wa = torch.from_numpy(detgen.normalish("omni/grad/wa", (2, 4, 512)))
    wp = torch.from_numpy(detgen.normalish("omni/grad/wp", (2, 5, 256)))
    wt = torch.from_numpy(detgen.normalish("omni/grad/wt", (2, 13, 256)))
    # product side (its ReLU decisions are recorded for the oracle)
    spy = _GateSpy(omni, monkeypatch)
    ag = m.process_audio(audio.cuda())
    pg = m.process_pose(pose.cuda())
    tg = m.condition_tokens(ag, pg)
    lg = (ag * wa.cuda()).sum() + (pg * wp.cuda()).sum() + (tg * wt.cuda()).sum()
    lg.backward()
    # oracle side: the same piecewise-linear pose stack
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    a = OH.process_audio(osd, audio)
    p = OH.process_pose(osd, pose, prefix="pose_guider.", gates=spy.gates(pose.shape[0]))
    tok = OH.condition_tokens(osd, a, p)
    lo = (a * wa).sum() + (p * wp).sum() + (tok * wt).sum()
    lo.backward()
    assert abs(lg.item() - lo.item()) < 2e-2 * abs(lo.item()) + 1e-2
    bad = []
    for name, prm in m.named_parameters():
        og = osd[name].grad
        assert prm.grad is not None and og is not None, name
        err = rel_rms(prm.grad, og)
        # fp32 layers: accumulation order only.  Pose Conv3d stack, differentiated at the product's own ReLU decisions
        # (with the oracle's own they sit 5-9 % apart: 0.1 % of the gates flip in the bf16 forward, see the test above):
        # bf16 operands, as any matrix gradient of this build (measured 5e-3).  The layers downstream of the ReLUs
        # (pose_fc, projector, temporal embedding) see the forward's bf16 rounding only.
        tol = 1e-3 if name.startswith("audio_processor") else (2e-2 if name.startswith("pose_guider") else 4e-2)
        if err > tol:
            bad.append((name, err))
    assert not bad, bad


def test_pose_conv_gradients_with_the_products_relu_gates(omni, monkeypatch):
    """The pose Conv3d gradients sit 5-9 % (relative RMS) from the fp32 autograd oracle (bound 1.5e-1 above).  Claimed
    cause: the bf16 forward flips ~1 % of the ReLU gates at pre-activations ~ 0, where the gradient is
    discontinuous.  Proof: hand the oracle the PRODUCT's gate decisions (y = conv * gate instead of ReLU) — the
    residual must then collapse to the bound of a bf16-operand matrix gradient (2e-2, as for the DiT's weights)."""
    from oracle import detgen, make_golden, omnihuman_oracle as OH
    sd = make_golden.omni_state_dict()
    audio, pose = make_golden.omni_inputs()
    m = omni.OmniConditionsModule(**make_golden.OMNI_TINY)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    wp = torch.from_numpy(detgen.normalish("omni/grad/wp", (2, 5, 256)))
    rec = []
    orig = omni._conv3d_relu_fwd

    def spy(x_cl, conv, stride_hw):
        y = orig(x_cl, conv, stride_hw)
        rec.append((conv.out_channels, y.detach()))
        return y
    monkeypatch.setattr(omni, "_conv3d_relu_fwd", spy)
    pg = m.process_pose(pose.cuda())
    (pg * wp.cuda()).sum().backward()
    B = pose.shape[0]
    assert len(rec) == 3 * B                                   # three layers per sample, samples in order
    gates, flips = [], []
    for layer in range(3):
        per = []
        for b in range(B):
            co, y = rec[3 * b + layer]                         # bf16 [T, h, w, Cout_p], channels last
            per.append((y[..., :co].float() > 0).permute(3, 0, 1, 2).cpu())      # [C', T, h, w]
        gates.append(torch.stack(per))
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    with torch.no_grad():                                      # how many decisions differ from the fp32 forward's own
        x = pose.float()
        for layer, (key, st) in enumerate((("0", 1), ("2", 2), ("4", 2))):
            x = torch.nn.functional.conv3d(x, sd["pose_guider." + key + ".weight"], sd["pose_guider." + key + ".bias"],
                                           stride=(1, st, st), padding=1)
            flips.append(float(((x > 0) != gates[layer]).float().mean()))
            x = torch.relu(x)
    p_o = OH.process_pose(osd, pose, prefix="pose_guider.", gates=gates)
    (p_o * wp).sum().backward()
    worst_gated = 0.0
    for name, prm in m.named_parameters():
        if name.startswith("pose_guider"):
            worst_gated = max(worst_gated, rel_rms(prm.grad, osd[name].grad))
    # the same comparison with the oracle's own ReLUs, for the record
    osd2 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (OH.process_pose(osd2, pose, prefix="pose_guider.") * wp).sum().backward()
    worst_relu = max(rel_rms(prm.grad, osd2[name].grad) for name, prm in m.named_parameters() if name.startswith("pose_guider"))
    print(f"[measured] pose Conv3d gradients vs the fp32 oracle: {worst_relu:.3e} with its own ReLUs, {worst_gated:.3e} with "
          f"the product's gates; gate decisions that differ per layer: {[f'{f:.2%}' for f in flips]}")
    assert worst_gated < 2e-2, worst_gated
    assert worst_relu > 2 * worst_gated                        # ... and the gates ARE what the 5-9 % consisted of


def test_omnihuman_training_step_matches_autograd_oracle(omni, wan_model_mod, monkeypatch):
    """OmniHumanWanT2V.training_step (omnihuman_wan_t2v.py:453-488) with audio + pose conditioning: loss and the
    gradients of every adapter parameter and of the DiT against autograd through the oracle (fp32 DiT with the
    condition tokens prepended to the context + fp32 adapters)."""
    from oracle import detgen, make_golden, omnihuman_oracle as OH, wan_dit_oracle as O
    cfg = O.DiTConfig(model_type="t2v", in_dim=16, num_layers=2, **make_golden.TINY)
    dsd = O.synth_state_dict(cfg, "omni/dit")
    dsd["head.head.weight"] = torch.from_numpy(detgen.uniform("omni/dit/headw", tuple(dsd["head.head.weight"].shape),
                                                              -0.08, 0.08))
    kw = dict(model_dim=256, num_frames=5, audio_dim=32, pose_keypoints=6)
    osd = make_golden.omni_state_dict("omni/sd", pose_prefix="pose_processor.", widths=(128, 256), **kw)
    audio, pose = make_golden.omni_inputs("omni/in", **kw)
    frames = torch.from_numpy(detgen.normalish("omni/tr/frames", (2, 16, 2, 4, 6)))
    noise = torch.from_numpy(detgen.normalish("omni/tr/noise", (2, 16, 2, 4, 6)))
    ctx = torch.from_numpy(detgen.normalish("omni/ctx", (20, 64)))
    t = torch.tensor([0.3, 0.7])
    # ---- product (the pose stack's ReLU decisions are recorded for the oracle)
    spy = _GateSpy(omni, monkeypatch)
    dit = wan_model_mod.WanModel(num_layers=2, **make_golden.TINY)
    dit.load_state_dict(dsd)
    dit = dit.cuda().train()
    m = omni.OmniHumanWanT2V(dict(num_frames=5, num_keypoints=6, model_dim=256, audio_dim=32), device_id=0,
                             wan_t2v=_StubT2V(dit, _StubVAE(None)))
    m.load_state_dict(osd, strict=False)
    loss = m.training_step(frames, {"text": ctx, "audio_features": audio, "pose_heatmaps": pose}, t, noise=noise)
    loss.backward()
    # ---- oracle
    d_o = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    o_o = {k: v.clone().requires_grad_(True) for k, v in osd.items()}
    tok = OH.condition_tokens(o_o, OH.process_audio(o_o, audio), OH.process_pose(o_o, pose, gates=spy.gates(pose.shape[0])))
    tok.retain_grad()
    tt = t.view(-1, 1, 1, 1, 1)
    noisy = (1 - tt) * frames + tt * noise
    pred = torch.stack(O.dit_forward_autograd(d_o, cfg, list(noisy), t, [ctx, ctx], 24, reference_ffn_freeze=True,
                                              extra_tokens=tok))
    lo = torch.mean((pred - frames) ** 2 * (1 - tt))
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-2 * lo.item()
    bad, worst = [], 0.0
    for name, prm in m.named_parameters():
        if name.startswith("wan_t2v"):
            continue
        og = o_o[name].grad
        assert prm.grad is not None, name
        err = rel_rms(prm.grad, og)
        worst = max(worst, err)
        if err > 2e-2:                  # through two DiT blocks of bf16 attention + the adapters (measured 8.6e-3)
            bad.append((name, err))
    for name in ("blocks.0.cross_attn.k.weight", "blocks.1.cross_attn.v.weight", "blocks.0.self_attn.q.weight",
                 "patch_embedding.weight"):
        err = rel_rms(dict(dit.named_parameters())[name].grad, d_o[name].grad)
        if err > 6e-2:
            bad.append((name, err))
    print(f"[measured] OmniHuman training step, adapter gradients vs the autograd oracle (product's ReLU gates): worst {worst:.3e}")
    assert not bad, bad
    # ready tokens that require grad receive theirs
    tok_g = m.condition_tokens(m.process_audio(audio.cuda()), m.process_pose(pose.cuda())).detach().requires_grad_(True)
    for prm in m.parameters():
        prm.grad = None
    m.training_step(frames, {"text": ctx, "tokens": tok_g}, t, noise=noise).backward()
    assert tok_g.grad is not None and rel_rms(tok_g.grad, tok.grad) < 6e-2
