"""End-to-end parity of the HIP WanModel against the CPU oracle
(oracle/wan_dit_oracle.py, itself pinned to the reference by tests/golden)."""
import pytest
import torch

from conftest import rel_rms, set_option

pytestmark = pytest.mark.gpu

# Stated tolerance of the bf16 path vs the fp32 oracle (north_star: "within a
# stated bf16 tolerance"): every GEMM/attention operand is rounded to bf16
# (2^-9 relative), accumulation and the residual stream stay fp32.  Relative
# RMS error of the final velocity grows ~sqrt(layers):
# (measured with tests/probes/err_report.py: 3.5e-3 for the tiny models, 5.0e-3 for Wan2.1-1.3B; bounds = 2x)
TOL_TINY = 8.0e-3     # 2..13 layers, d=256
TOL_FULL = 1.2e-2     # 30 layers, d=1536
# guided velocity v = u + 7.5 (c - u) of config 1: the errors of the two forwards enter with weights 7.5 and 6.5
# while |v| stays O(|u|) when c ~ u, so its relative error is ~ sqrt(7.5^2 + 6.5^2) = 9.9 x the forward's.
# Measured on MI355X (round 2): forward 4.99e-3, guided velocity 9.5e-3 (c and u share most of their rounding
# noise, so the amplification is 1.9 x rather than the worst-case 9.9 x); bound = 2 x measured.
TOL_CFG = 2.0e-2


def _inputs(cfg, grids, ctx_lens, tag):
    from oracle import detgen
    xs = [torch.from_numpy(detgen.normalish(f"{tag}/x{i}", (cfg.in_dim, g[0], g[1] * 2, g[2] * 2)))
          for i, g in enumerate(grids)]
    ctx = [torch.from_numpy(detgen.normalish(f"{tag}/c{i}", (n, cfg.text_dim))) for i, n in enumerate(ctx_lens)]
    return xs, ctx


@pytest.mark.parametrize("layers", [2, 13])
def test_tiny_model_matches_oracle(wan_model_mod, layers):
    from oracle import wan_dit_oracle as O
    cfg = O.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=layers, text_dim=64, text_len=32, freq_dim=64)
    sd = O.synth_state_dict(cfg, f"tiny{layers}")
    grids, seq_len = [(2, 3, 4), (1, 2, 3)], 30
    xs, ctx = _inputs(cfg, grids, [32, 11], f"tiny{layers}")
    t = torch.tensor([999., 500.])
    ref = O.dit_forward(sd, cfg, xs, t, ctx, seq_len)
    m = wan_model_mod.WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=layers, text_dim=64, text_len=32,
                               freq_dim=64)
    m.load_state_dict(sd)
    m = m.cuda().eval().requires_grad_(False)
    out = m([u.cuda() for u in xs], t.cuda(), [c.cuda() for c in ctx], seq_len)
    assert len(out) == 2
    for o, r in zip(out, ref):
        assert o.dtype == torch.float32 and o.shape == r.shape
        assert rel_rms(o, r) < TOL_TINY
    # batched-tensor input form used by distilled_trainer.py:273
    xb = torch.stack([xs[0], xs[0]]).cuda()
    out2 = m(xb, t.cuda(), [ctx[0].cuda(), ctx[0].cuda()], seq_len)
    assert out2[0].shape == ref[0].shape


def test_context_state_reuse_is_bit_identical(wan_model_mod):
    """encode_context() (text embedding + per-block cross-attention K/V computed once per sample)
    must not change a single bit of the forward, for t2v and for i2v (image-token K/V cached too)."""
    from oracle import wan_dit_oracle as O, make_golden
    cfg = O.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=3, text_dim=64, text_len=32, freq_dim=64)
    m = wan_model_mod.WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=3, text_dim=64, text_len=32, freq_dim=64)
    m.load_state_dict(O.synth_state_dict(cfg, "ctxstate"))
    m = m.cuda().eval().requires_grad_(False)
    xs, ctx = _inputs(cfg, [(2, 3, 4), (1, 2, 3)], [32, 11], "ctxstate")
    xs, ctx = [u.cuda() for u in xs], [c.cuda() for c in ctx]
    st = m.encode_context(ctx)
    for tval in (999., 250.):                      # first use fills the K/V cache, second use reads it
        t = torch.tensor([tval, tval / 2]).cuda()
        plain = m(xs, t, ctx, 30)
        cached = m(xs, t, st, 30)
        assert all(torch.equal(a, b) for a, b in zip(plain, cached))
    assert len(st.kv) == 3
    with torch.no_grad():
        m.blocks[1].cross_attn.k.weight.mul_(1.5)  # stale cache must be refused, not silently reused
    with pytest.raises(ValueError):
        m(xs, t, st, 30)
    # i2v
    cfg, tag, xs, ctx, tt, seq_len, ys, clip = make_golden.tiny_case("i2v", 2)
    mi = wan_model_mod.WanModel(model_type="i2v", in_dim=36, num_layers=2, **make_golden.TINY)
    mi.load_state_dict(O.synth_state_dict(cfg, tag))
    mi = mi.cuda().eval().requires_grad_(False)
    xs, ctx, ys, clip = [u.cuda() for u in xs], [c.cuda() for c in ctx], [u.cuda() for u in ys], clip.cuda()
    sti = mi.encode_context(ctx, clip_fea=clip)
    plain = mi(xs, tt.cuda(), ctx, seq_len, clip_fea=clip, y=ys)
    for _ in range(2):
        cached = mi(xs, tt.cuda(), sti, seq_len, y=ys)
        assert all(torch.equal(a, b) for a, b in zip(plain, cached))
    assert len(sti.kv) == 4                        # (k, k_img) x 2 blocks


def test_block_hooks_and_errors(wan_model_mod):
    from oracle import wan_dit_oracle as O
    cfg = O.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=32, freq_dim=64)
    m = wan_model_mod.WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=32, freq_dim=64)
    m.load_state_dict(O.synth_state_dict(cfg, "hooks"))
    m = m.cuda().eval().requires_grad_(False)
    seen = []
    h = m.blocks[1].register_forward_hook(lambda mod, inp, out: seen.append(tuple(out.shape)))
    xs, ctx = _inputs(cfg, [(1, 2, 3)], [7], "hooks")
    m([xs[0].cuda()], torch.tensor([3.]).cuda(), [ctx[0].cuda()], 8)
    h.remove()
    assert seen == [(1, 8, 256)]
    with pytest.raises(AssertionError):
        m([xs[0].cuda()], torch.tensor([3.]).cuda(), [ctx[0].cuda()], 5)   # tokens > seq_len (model.py:521)


def test_wan_1_3b_single_frame_cfg_pair():
    """BASELINE config 1: one [16,1,60,104] latent, S=1560, teacher CFG pair (generate.py:227-229)."""
    import importlib
    from oracle import wan_dit_oracle as O, detgen
    mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.model")
    cfg = O.DiTConfig.wan_t2v_1_3b()
    sd = O.synth_state_dict(cfg, "wan1.3b")
    noise = torch.from_numpy(detgen.normalish("c1/noise", (16, 1, 60, 104)))
    cpos = torch.from_numpy(detgen.normalish("c1/ctx", (512, 4096)))
    cneg = torch.from_numpy(detgen.normalish("c1/neg", (37, 4096)))
    t = torch.tensor([999.])
    ref = O.cfg_velocity(sd, cfg, noise, t, cpos, cneg, 1560, 7.5)
    m = mod.WanModel(**{k: getattr(cfg, k) for k in ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim",
                                                      "freq_dim", "text_dim", "out_dim", "num_heads", "num_layers",
                                                      "qk_norm", "cross_attn_norm", "eps")})
    m.load_state_dict(sd)
    m = m.cuda().eval().requires_grad_(False)
    c = m([noise.cuda()], t.cuda(), [cpos.cuda()], 1560)[0]
    u = m([noise.cuda()], t.cuda(), [cneg.cuda()], 1560)[0]
    v = (u + 7.5 * (c - u)).cpu()
    assert v.shape == (16, 1, 60, 104)
    # CFG amplifies the difference of two bf16 forwards by 7.5: compare the
    # individual forwards at TOL_FULL and the guided velocity at a looser bound
    ref_u = O.dit_forward(sd, cfg, [noise], t, [cneg], 1560)[0]
    assert rel_rms(u, ref_u) < TOL_FULL
    # ... and the unconditional forward straight against the REAL reference's output on these inputs (64 probe
    # elements, a coarse grid, the mean: tests/golden/dit_wan1_3b_c1.npz written by oracle/make_golden.py from the
    # imported reference WanModel) — no oracle in between
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_wan1_3b_c1.npz"))
    uc = u.cpu()
    probe = uc.flatten()[torch.from_numpy(g["probe_idx"])]
    assert rel_rms(probe, torch.from_numpy(g["probe"])) < TOL_FULL
    assert rel_rms(uc[:, 0, ::6, ::8], torch.from_numpy(g["coarse"])) < TOL_FULL
    assert abs(float(uc.double().mean()) - float(g["mean"])) < 2e-3 * float(g["abs_mean"])
    assert abs(float(uc.double().abs().mean()) - float(g["abs_mean"])) < 2e-3 * float(g["abs_mean"])
    e_v = rel_rms(v, ref)
    print(f"[measured] 1.3B forward rel-RMS {rel_rms(u, ref_u):.3e}, guided velocity rel-RMS {e_v:.3e}")
    assert e_v < TOL_CFG
    # the same pair as one batch-2 forward (trainer.teacher_cfg_velocity).  Since ABI v9 the FFN-down contraction runs in
    # 4 slices at 1 560 rows and in 2 at 3 120 (ops.gemm_raw(split_k=True)): another fp32 summation order, bf16 roundings
    # flip and travel through 30 layers, CFG multiplies the difference by 7.5 — the batched pair is as far from the two
    # calls as either is from the oracle; with the slices switched off it is the two calls bit for bit.
    trainer = importlib.import_module("omnihuman-1-hack_amd.trainer")
    vt = trainer.teacher_cfg_velocity(m, noise, t, cpos, cneg, 7.5)
    e_pair = rel_rms(vt, torch.add(u, c - u, alpha=7.5))
    print(f"[measured] batched teacher pair vs two calls (split K: 2 vs 4 slices): {e_pair:.3e}")
    assert e_pair < TOL_CFG and rel_rms(vt.cpu(), ref) < TOL_CFG
    set_option("OMH_GEMM_SPLITK", "0")
    try:
        c0 = m([noise.cuda()], t.cuda(), [cpos.cuda()], 1560)[0]
        u0 = m([noise.cuda()], t.cuda(), [cneg.cuda()], 1560)[0]
        assert torch.equal(trainer.teacher_cfg_velocity(m, noise, t, cpos, cneg, 7.5), torch.add(u0, c0 - u0, alpha=7.5))
    finally:
        set_option("OMH_GEMM_SPLITK", None)


def test_tiny_i2v_model_matches_oracle_and_reference_vectors(wan_model_mod):
    """i2v: 36 input channels (latent + mask/first-frame `y`), CLIP tokens through img_emb, the extra
    image-token attention (model.py:189-230, 362-374, 511-512, 534-537)."""
    import os
    import numpy as np
    from oracle import make_golden
    cfg, tag, xs, ctx, tt, seq_len, ys, clip = make_golden.tiny_case("i2v", 2)
    from oracle import wan_dit_oracle as O
    sd = O.synth_state_dict(cfg, tag)
    m = wan_model_mod.WanModel(model_type="i2v", in_dim=36, num_layers=2, **make_golden.TINY)
    m.load_state_dict(sd)
    m = m.cuda().eval().requires_grad_(False)
    out = m([u.cuda() for u in xs], tt.cuda(), [c.cuda() for c in ctx], seq_len, clip_fea=clip.cuda(),
            y=[u.cuda() for u in ys])
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_i2v_L2.npz"))
    for o, k in zip(out, ("out0", "out1")):
        assert rel_rms(o, torch.from_numpy(g[k])) < TOL_TINY      # straight against the reference's own output


@pytest.mark.parametrize("model_type", ["t2v", "i2v"])
def test_wan_14b_width_one_layer(wan_model_mod, model_type):
    """The 14B geometry (d=5120, 40 heads x 128, ffn 13824; wan_t2v_14B.py / wan_i2v_14B.py:26-35) on one
    layer (0.4 G synthetic parameters): every kernel at its widest row (LN/RMSNorm 20 vectors per lane, 40-head attention, K=13824 GEMM)."""
    from oracle import wan_dit_oracle as O, detgen
    i2v = model_type == "i2v"
    kw = dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=1, text_dim=4096, text_len=512, freq_dim=256)
    cfg = O.DiTConfig(model_type=model_type, in_dim=36 if i2v else 16, **kw)
    sd = O.synth_state_dict(cfg, f"wan14b/{model_type}")
    grids, seq_len = [(2, 6, 10), (1, 4, 6)], 128
    xs = [torch.from_numpy(detgen.normalish(f"w14/x{i}", (16, g[0], g[1] * 2, g[2] * 2))) for i, g in enumerate(grids)]
    ctx = [torch.from_numpy(detgen.normalish(f"w14/c{i}", (n, 4096))) for i, n in enumerate((77, 512))]
    ys = clip = None
    if i2v:
        ys = [torch.from_numpy(detgen.normalish(f"w14/y{i}", (20, g[0], g[1] * 2, g[2] * 2))) for i, g in enumerate(grids)]
        clip = torch.from_numpy(detgen.normalish("w14/clip", (2, 257, 1280)))
    t = torch.tensor([937., 250.])
    ref = O.dit_forward(sd, cfg, xs, t, ctx, seq_len, clip_fea=clip, y=ys)
    m = wan_model_mod.WanModel(model_type=model_type, in_dim=36 if i2v else 16, **kw)
    m.load_state_dict(sd)
    m = m.cuda().eval().requires_grad_(False)
    out = m([u.cuda() for u in xs], t.cuda(), [c.cuda() for c in ctx], seq_len,
            clip_fea=clip.cuda() if i2v else None, y=[u.cuda() for u in ys] if i2v else None)
    for o, r in zip(out, ref):
        assert o.shape == r.shape and rel_rms(o, r) < TOL_TINY


def test_batch_invariant_flag_pins_the_summation_order(wan_model_mod):
    """the inference block asks for split K on its gated-residual GEMMs, so at one / two [16,1,60,104]
    clips the FFN-down contraction (K = 8 960 over 56 / 104 tiles of 256 x 192) is summed in 4 / 2 slices and a
    sample's last bits depend on its batch.  ``model.batch_invariant = True`` switches that off: a clip alone and the
    same clip inside a batch of two then agree BIT FOR BIT; with the default the two agree to fp32 summation noise
    (and are NOT identical — which is what pins that the split is really taken by default)."""
    torch.manual_seed(7)
    m = wan_model_mod.WanModel(dim=1536, ffn_dim=8960, num_heads=12, num_layers=2, text_dim=4096, text_len=512,
                               freq_dim=256)
    with torch.no_grad():
        torch.nn.init.xavier_uniform_(m.head.head.weight)
    m = m.cuda().eval().requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(3)
    x0 = torch.randn(16, 1, 60, 104, device="cuda", generator=g)
    x1 = torch.randn(16, 1, 60, 104, device="cuda", generator=g)
    c0 = torch.randn(77, 4096, device="cuda", generator=g)
    c1 = torch.randn(31, 4096, device="cuda", generator=g)
    t1, t2 = torch.tensor([600.0], device="cuda"), torch.tensor([600.0, 600.0], device="cuda")
    assert m.batch_invariant is False
    alone = m([x0], t1, [c0], 1560)[0]
    both = m([x0, x1], t2, [c0, c1], 1560)[0]
    assert rel_rms(both, alone) < 4e-3            # another fp32 summation order flips bf16 roundings downstream: bf16 noise
    assert not torch.equal(both, alone), "the default takes the k slices: 4 for one clip, 2 for two"
    m.batch_invariant = True
    alone_i = m([x0], t1, [c0], 1560)[0]
    both_i = m([x0, x1], t2, [c0, c1], 1560)[0]
    assert torch.equal(both_i, alone_i)
    assert rel_rms(alone_i, alone) < 4e-3


def test_fused_qkv_projection_in_the_forward_changes_no_bit(wan_model_mod):
    """One clip of a long sequence takes q | k | v as ONE product (OMH_EPI_BF16_SPLIT_T, V^T written transposed by the
    stream); GEMM_QKV = 0 routes the same call through the two separate products: the forward must not move by a bit."""
    torch.manual_seed(9)
    m = wan_model_mod.WanModel(dim=1536, ffn_dim=8960, num_heads=12, num_layers=2, text_dim=4096, text_len=512,
                               freq_dim=256)
    with torch.no_grad():
        torch.nn.init.xavier_uniform_(m.head.head.weight)
    m = m.cuda().eval().requires_grad_(False)
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(16, 6, 60, 104, device="cuda", generator=g)          # S = 9 360 >= the fused path's threshold
    c = torch.randn(50, 4096, device="cuda", generator=g)
    t = torch.tensor([300.0], device="cuda")
    fused = m([x], t, [c], 9360)[0]
    set_option("GEMM_QKV", "0")
    plain = m([x], t, [c], 9360)[0]
    assert torch.isfinite(fused).all() and torch.equal(fused, plain)


def test_cfg_pair_shares_block_0_self_attention_and_changes_no_bit(wan_model_mod):
    """WanModel.forward_cfg_pair (round 6): the conditional and the unconditional forward of a CFG step on the same
    latents and timestep (text2video.py:238-241, generate.py:205-229) share everything that does not see the context —
    the embeddings and block 0's self-attention sub-layer — and must equal two forward() calls BIT FOR BIT: t2v with raw
    contexts and with ContextStates, i2v with y / clip_fea, a two-clip batch with unequal grids, a sequence past the
    long-sequence threshold (fused q | k | v + the attention stream); the shared sub-layer is launched once, block
    forward hooks fire once per branch with the block's full output."""
    from oracle import wan_dit_oracle as O, make_golden
    ops = __import__("importlib").import_module(wan_model_mod.__name__.rsplit(".", 3)[0] + ".ops")
    cfg = O.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=3, text_dim=64, text_len=32, freq_dim=64)
    m = wan_model_mod.WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=3, text_dim=64, text_len=32, freq_dim=64)
    m.load_state_dict(O.synth_state_dict(cfg, "ctxstate"))
    m = m.cuda().eval().requires_grad_(False)
    xs, ctx = _inputs(cfg, [(2, 3, 4), (1, 2, 3)], [32, 11], "pair")
    xs = [u.cuda() for u in xs]
    ctx = [c.cuda() for c in ctx]
    null = [c[:5].clone() * 0.5 for c in ctx]
    t = torch.tensor([999., 250.]).cuda()
    want_c, want_u = m(xs, t, ctx, 30), m(xs, t, null, 30)
    calls = {"self": 0}
    raw = ops.flash_attn_raw

    def counting(*a, **k):
        calls["self"] += int(a[7] == a[8] == 30)                 # Lq == Lk == seq_len: a self-attention launch
        return raw(*a, **k)
    taps = []
    hook = m.blocks[0].register_forward_hook(lambda mod, i, o: taps.append(o.clone()))
    try:
        ops.flash_attn_raw = counting
        got_c, got_u = m.forward_cfg_pair(xs, t, ctx, null, 30)
    finally:
        ops.flash_attn_raw = raw
        hook.remove()
    assert calls["self"] == 2 * 3 - 1                            # block 0's once, blocks 1, 2 once per branch
    assert len(taps) == 2 and not torch.equal(taps[0], taps[1])  # one hook call per branch, with that branch's output
    for a, b in zip(got_c + got_u, want_c + want_u):
        assert torch.equal(a, b)
    st_c, st_u = m.encode_context(ctx), m.encode_context(null)
    got_c, got_u = m.forward_cfg_pair(xs, t, st_c, st_u, 30)
    for a, b in zip(got_c + got_u, want_c + want_u):
        assert torch.equal(a, b)
    # i2v: y and clip_fea ride along
    cfg_i, tag, xi, ci, tt, seq_i, ys, clip = make_golden.tiny_case("i2v", 2)
    mi = wan_model_mod.WanModel(model_type="i2v", in_dim=36, num_layers=2, **make_golden.TINY)
    mi.load_state_dict(O.synth_state_dict(cfg_i, tag))
    mi = mi.cuda().eval().requires_grad_(False)
    xi, ci, ys, clip = [u.cuda() for u in xi], [c.cuda() for c in ci], [u.cuda() for u in ys], clip.cuda()
    c0 = [c[:3] * 0.25 for c in ci]
    a_c, a_u = mi.encode_context(ci, clip_fea=clip), mi.encode_context(c0, clip_fea=clip)
    want = mi(xi, tt.cuda(), a_c, seq_i, y=ys), mi(xi, tt.cuda(), a_u, seq_i, y=ys)
    got = mi.forward_cfg_pair(xi, tt.cuda(), a_c, a_u, seq_i, y=ys)
    assert all(torch.equal(p, q) for p, q in zip(got[0] + got[1], want[0] + want[1]))
    g = torch.Generator().manual_seed(5)
    # past the long-sequence threshold: S = 2 x 64 x 68 = 8 704 >= 8 192 (fused q | k | v, the attention stream where it applies)
    cfg_l = O.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=32, freq_dim=64)
    ml = wan_model_mod.WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=32, freq_dim=64)
    ml.load_state_dict(O.synth_state_dict(cfg_l, "pairlong"))
    ml = ml.cuda().eval().requires_grad_(False)
    xl = [torch.randn(16, 2, 128, 136, generator=g).cuda()]
    tl = torch.tensor([321.]).cuda()
    w_c, w_u = ml(xl, tl, ctx[:1], 8704)[0], ml(xl, tl, null[:1], 8704)[0]
    g_c, g_u = ml.forward_cfg_pair(xl, tl, ctx[:1], null[:1], 8704)
    assert torch.equal(g_c[0], w_c) and torch.equal(g_u[0], w_u) and not torch.equal(w_c, w_u)
    # an inference path: refused when autograd would be expected to follow it
    ml.requires_grad_(True)
    with pytest.raises(RuntimeError):
        ml.forward_cfg_pair(xl, tl, ctx[:1], null[:1], 8704)
    with torch.no_grad():
        g2 = ml.forward_cfg_pair(xl, tl, ctx[:1], null[:1], 8704)
    assert torch.equal(g2[0][0], w_c)
