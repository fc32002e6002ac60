"""CPU: the oracle (oracle/*.py) against the golden vectors produced by the REAL
reference (oracle/make_golden.py), and against the live reference when
/root/reference is present (build container only)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_rms

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _g(name):
    return np.load(os.path.join(GOLD, name))


def test_dit_ops_match_reference_vectors():
    from oracle import detgen, wan_dit_oracle as O
    g = _g("dit_ops.npz")
    t = torch.tensor([0., 1., 999., 1000.])
    assert np.abs(O.sinusoidal_embedding_1d(256, t).float().numpy() - g["sinusoid"]).max() < 1e-6
    xq = torch.from_numpy(detgen.normalish("golden/rope/x", (2, 30, 2, 128)))
    rope = O.rope_apply(xq, [(2, 3, 4), (1, 5, 6)], O.rope_table(128))
    assert np.abs(rope.numpy() - g["rope"]).max() < 1e-6
    w = torch.from_numpy(1.0 + detgen.uniform("golden/rms/w", (256,), -0.2, 0.2))
    xr = torch.from_numpy(detgen.normalish("golden/rms/x", (3, 7, 256)))
    assert np.abs(O.rms_norm(xr, w, 1e-6).numpy() - g["rmsnorm"]).max() < 1e-5
    assert np.abs(O.layer_norm(xr * 3 + 0.5, 1e-6).numpy() - g["layernorm"]).max() < 1e-5


@pytest.mark.parametrize("mt,layers", [("t2v", 2), ("t2v", 13), ("i2v", 2)])
def test_dit_forward_matches_reference_vectors(mt, layers):
    from oracle import make_golden, wan_dit_oracle as O
    cfg, tag, xs, ctx, tt, seq_len, ys, clip = make_golden.tiny_case(mt, layers)
    sd = O.synth_state_dict(cfg, tag)
    out = O.dit_forward(sd, cfg, xs, tt, ctx, seq_len, clip_fea=clip, y=ys)
    g = _g(f"dit_{mt}_L{layers}.npz")
    for o, k in zip(out, ("out0", "out1")):
        assert o.shape == g[k].shape
        assert np.abs(o.numpy() - g[k]).max() < 2e-5          # fp32 vs fp32, summation order only


def test_vae_matches_reference_vectors():
    from oracle import detgen, wan_vae_oracle as V
    z = torch.from_numpy(detgen.normalish("golden/vae/z", (16, 3, 8, 8)))
    vid = torch.from_numpy(detgen.uniform("golden/vae/vid", (3, 9, 32, 32)))
    for dim, tol in ((16, 2e-5), (96, 1e-3)):               # dim 96 decode is stored as fp16
        cfg = V.VAEConfig(dim=dim)
        sd = V.synth_state_dict(cfg, f"golden/vae{dim}")
        g = _g(f"vae_dim{dim}.npz")
        dec = V.vae_decode(sd, cfg, z)
        assert dec.shape == (3, 9, 64, 64)
        assert np.abs(dec.numpy() - g["decode"].astype(np.float32)).max() < tol
        enc = V.vae_encode(sd, cfg, vid)
        assert np.abs(enc.numpy() - g["encode"]).max() < 2e-5


def test_unipc_matches_reference_vectors():
    from oracle import detgen, sampler_oracle as SO
    g = _g("unipc_6steps.npz")
    o = SO.UniPCOracle(6, 3.0)
    assert np.array_equal(o.sigmas.numpy(), g["sigmas"]) and np.array_equal(o.timesteps.numpy(), g["timesteps"])
    x = torch.from_numpy(detgen.normalish("golden/unipc/x", (1, 16, 2, 6, 8)))
    for k in range(6):
        v = torch.from_numpy(detgen.normalish(f"golden/unipc/v{k}", (1, 16, 2, 6, 8)))
        x = o.step(v, x)
        assert np.abs(x.numpy() - g["traj"][k]).max() < 1e-6
    # closed forms: x0 = x - sigma v ; last step lands exactly on the x0 prediction
    o1 = SO.UniPCOracle(1, 5.0)
    x = torch.from_numpy(detgen.normalish("golden/unipc/x", (1, 16, 2, 6, 8)))
    v = torch.from_numpy(detgen.normalish("golden/unipc/v0", (1, 16, 2, 6, 8)))
    assert torch.allclose(o1.step(v, x), x - o1.sigmas[0] * v, atol=1e-6)


def test_dpmpp_matches_reference_vectors():
    """sample_solver='dpm++' (fm_solvers.py): the oracle reproduces the reference trajectory bit for bit."""
    from oracle import detgen, sampler_oracle as SO
    g = _g("dpmpp_6steps.npz")
    o = SO.DPMSolverOracle(6, 3.0)
    assert np.array_equal(o.sigmas.numpy(), g["sigmas"]) and np.array_equal(o.timesteps.numpy(), g["timesteps"])
    assert o.sigmas[0] == 1.0 and o.sigmas[-1] == 0.0          # lambda = -/+inf at both ends: must stay finite
    x = torch.from_numpy(detgen.normalish("golden/dpmpp/x", (1, 16, 2, 6, 8)))
    for k in range(6):
        v = torch.from_numpy(detgen.normalish(f"golden/dpmpp/v{k}", (1, 16, 2, 6, 8)))
        x = o.step(v, x)
        assert np.array_equal(x.numpy(), g["traj"][k])
    # closed forms: one step lands on x0 = x - v (sigma 1 -> 0); the last step of any run returns the x0 prediction
    o1 = SO.DPMSolverOracle(1, 5.0)
    x = torch.from_numpy(detgen.normalish("golden/dpmpp/x", (1, 16, 2, 6, 8)))
    v = torch.from_numpy(detgen.normalish("golden/dpmpp/v0", (1, 16, 2, 6, 8)))
    assert torch.allclose(o1.step(v, x), x - v, atol=1e-6)


def test_oracle_1_3b_forward_matches_reference_probes():
    """BASELINE config 1 at its real size: the oracle's Wan2.1-T2V-1.3B forward (30 layers, d = 1536, S = 1560) on
    c1/noise + c1/neg, t = 999, against 64 probe elements, the mean / abs-mean and a coarse grid of the REAL
    reference's output on the same inputs (oracle/make_golden.py, `dit_wan1_3b_c1.npz`; ~10 s on 8 cores)."""
    from oracle import detgen, wan_dit_oracle as O
    g = np.load(os.path.join(GOLD, "dit_wan1_3b_c1.npz"))
    cfg = O.DiTConfig.wan_t2v_1_3b()
    sd = O.synth_state_dict(cfg, "wan1.3b")
    noise = torch.from_numpy(detgen.normalish("c1/noise", (16, 1, 60, 104)))
    cneg = torch.from_numpy(detgen.normalish("c1/neg", (37, 4096)))
    with torch.no_grad():
        out = O.dit_forward(sd, cfg, [noise], torch.tensor([999.]), [cneg], 1560)[0]
    assert np.abs(out.flatten()[torch.from_numpy(g["probe_idx"])].numpy() - g["probe"]).max() < 2e-4
    assert np.abs(out[:, 0, ::6, ::8].numpy() - g["coarse"]).max() < 2e-4
    assert abs(float(out.double().mean()) - float(g["mean"])) < 1e-5
    assert abs(float(out.double().abs().mean()) - float(g["abs_mean"])) < 1e-5


@pytest.mark.skipif(not os.path.isdir("/root/reference/seaweed_apt"), reason="reference tree not present")
def test_oracle_against_live_reference():
    from oracle import detgen, ref_import, wan_dit_oracle as O, wan_vae_oracle as V
    cfg = O.DiTConfig(dim=128, ffn_dim=256, num_heads=1, num_layers=3, text_dim=32, text_len=16, freq_dim=32)
    sd = O.synth_state_dict(cfg, "live")
    ref = ref_import.build_reference_dit(cfg, sd)
    xs = [torch.from_numpy(detgen.normalish("live/x", (16, 1, 4, 4)))]
    ctx = [torch.from_numpy(detgen.normalish("live/c", (5, 32)))]
    with torch.no_grad():
        r = ref(xs, torch.tensor([7.]), ctx, 6)[0]
    assert rel_rms(O.dit_forward(sd, cfg, xs, torch.tensor([7.]), ctx, 6)[0], r) < 1e-5
    cfgv = V.VAEConfig(dim=16)
    sdv = V.synth_state_dict(cfgv, "live/vae")
    refv = ref_import.build_reference_vae(sdv, dim=16)
    z = torch.from_numpy(detgen.normalish("live/z", (16, 2, 4, 4)))
    scale = [torch.tensor(V.LATENT_MEAN), 1.0 / torch.tensor(V.LATENT_STD)]
    assert rel_rms(V.vae_decode(sdv, cfgv, z), refv.decode(z[None], scale).clamp(-1, 1)[0]) < 1e-5


def test_detgen_is_stable():
    """The generator must give the same bits everywhere (weights are never shipped)."""
    from oracle import detgen
    a = detgen.uniform("stability", (5,))
    assert a.dtype == np.float32
    assert [float(v) for v in a] == [float(v) for v in detgen.uniform("stability", (5,))]
    assert abs(float(detgen.normalish("stability", (100000,)).std()) - 1.0) < 0.02
    assert float(np.abs(a).max()) < 1.0


def test_omnihuman_adapters_match_reference_vectors():
    """OmniHuman conditioning adapters (Omnihuman/omnihuman_wan_t2v.py:13-92): process_audio and the pose Conv3d
    stack of the oracle against outputs of the reference's own OmniConditionsModule; the default-schedule
    DPM-Solver++ its sampling loop uses (scheduler shift 1.0, set_timesteps(n)) bit for bit."""
    from oracle import detgen, make_golden, omnihuman_oracle as OH, sampler_oracle as SO
    g = _g("omnihuman_adapters.npz")
    sd = make_golden.omni_state_dict()
    audio, pose = make_golden.omni_inputs()
    assert rel_rms(OH.process_audio(sd, audio), torch.from_numpy(g["audio_tokens"])) < 1e-6
    feat = OH.pose_conv_stack(sd, pose, prefix="pose_guider.")
    assert tuple(feat.shape) == tuple(g["pose_features"].shape) == (2, 64, 5, 16, 16)
    assert rel_rms(feat, torch.from_numpy(g["pose_features"])) < 1e-5
    # the reference's own process_pose cannot run (axis labels swapped before pose_fc): the fixture records its error
    assert "shapes cannot be multiplied" in str(g["process_pose_error"])
    tok = OH.process_pose(sd, pose, prefix="pose_guider.")
    assert tuple(tok.shape) == (2, 5, 256)
    ct = OH.condition_tokens(sd, OH.process_audio(sd, audio), tok)
    assert tuple(ct.shape) == (2, 2 * 4 + 5, 256)
    o = SO.DPMSolverOracle(5, 1.0, default_schedule=True)
    assert np.array_equal(o.sigmas.numpy(), g["dpm_sigmas"]) and np.array_equal(o.timesteps.numpy(), g["dpm_timesteps"])
    x = torch.from_numpy(detgen.normalish("golden/omni/x", (1, 16, 2, 6, 8)))
    for k in range(5):
        v = torch.from_numpy(detgen.normalish(f"golden/omni/v{k}", (1, 16, 2, 6, 8)))
        x = o.step(v, x)
        assert np.array_equal(x.numpy(), g["dpm_traj"][k])
    assert OH.annealed_cfg(0, 50, 7.5) == 7.5 and abs(OH.annealed_cfg(25, 50, 7.5) - 4.25) < 1e-12


def test_i2v_training_gradients_match_reference_vectors():
    """The autograd oracle on the i2v backbone (CLIP tokens through img_emb, image-token attention branch, 36 input
    channels) against gradients produced by the real reference (oracle/make_golden.py train_i2v)."""
    from oracle import detgen, make_golden, wan_dit_oracle as O
    g = _g("dit_train_i2v_L2.npz")
    cfg, tag, xs, ctx, tt, seq_len, ys, clip = make_golden.tiny_case("i2v", 2)
    sd = O.synth_state_dict(cfg, tag)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.dit_forward_autograd(osd, cfg, xs, torch.tensor([1000.0, 1000.0]), ctx, seq_len, clip_fea=clip, y=ys)
    vt = torch.from_numpy(detgen.normalish(f"{tag}/vt", (16, 2, 6, 8)))
    loss = torch.nn.functional.mse_loss(out[0], vt)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    for name in g.files:
        if name == "loss":
            continue
        a = osd[name].grad
        a = a if a.numel() <= 100000 else a[:16]
        assert rel_rms(a, torch.from_numpy(g[name])) < 1e-5, name


def test_encoder_oracles_match_reference_vectors():
    """umT5 encoder (with the repository's cut-down block, t5.py:166-176) and the CLIP vision tower
    (use_31_block, clip.py:275-301) of the oracle against outputs of the REAL reference modules at a small width
    (oracle/make_golden.py encoders), and the relative-position buckets against the reference's own function."""
    from oracle import encoders_oracle as E, make_golden
    g = _g("encoders_t5_clip.npz")
    tc, vc, ids, mask, img = make_golden.encoder_cases()
    with torch.no_grad():
        t5 = E.t5_encode(E.t5_state_dict(tc, "golden/t5"), tc, ids, mask)
        vit = E.vit_forward(E.vit_state_dict(vc, "golden/vit"), vc, img)
    assert np.abs(t5.numpy() - g["t5"]).max() < 2e-5
    assert np.abs(vit.numpy() - g["vit"]).max() < 5e-5
    assert np.array_equal(E.t5_relative_buckets(40, 40, 32).numpy(), g["buckets"].astype(np.int64))
    # the upstream block (attention residual on the un-normalised stream + gated feed-forward) is a different function
    with torch.no_grad():
        up = E.t5_encode(E.t5_state_dict(tc, "golden/t5"), tc, ids, mask, reference_block_quirk=False)
    assert rel_rms(up, t5) > 0.05
    # padded positions do not influence the valid ones
    with torch.no_grad():
        short = E.t5_encode(E.t5_state_dict(tc, "golden/t5"), tc, ids[1:, :15], mask[1:, :15])
    assert float((short[0] - t5[1, :15]).abs().max()) < 1e-5
