"""Training step (distilled_trainer.py:241-316 semantics) on the HIP path:
loss and gradients against the reference's golden gradients and the autograd oracle."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import rel_rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# bf16 forward + bf16 backward operands, fp32 accumulation: per-tensor relative RMS error of a
# parameter gradient after 13 blocks of back-propagation.
# measured on MI355X (round 2, printed by the tests): matrices <= 1.01e-2 against the reference's own gradients and
# against the autograd oracle (t2v 13 layers, both freeze settings, i2v); 1-D parameters (bias / gain gradients: long
# sums with heavy cancellation) <= 4.5e-2.  Bounds = 2 x measured.
TOL_GRAD = 2e-2
TOL_GRAD_1D = 9e-2


def _setup(wan_model_mod, freeze=True):
    from oracle import make_golden, wan_dit_oracle as O, detgen
    cfg, tag, xs, ctx, tt, seq_len, _, _ = make_golden.tiny_case("t2v", 13)
    sd = O.synth_state_dict(cfg, tag)
    noise = torch.stack([xs[0], torch.from_numpy(detgen.normalish(f"{tag}/x0b", (16, 2, 6, 8)))])
    vt = torch.from_numpy(detgen.normalish(f"{tag}/vt", (2, 16, 2, 6, 8)))
    cl = [ctx[0], torch.from_numpy(detgen.normalish(f"{tag}/c0b", (32, 64)))]
    m = wan_model_mod.WanModel(num_layers=13, **make_golden.TINY)
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.reference_ffn_freeze = freeze
    return cfg, sd, m, noise, vt, cl


def test_training_step_matches_reference_gradients(wan_model_mod):
    cfg, sd, m, noise, vt, cl = _setup(wan_model_mod)
    g = np.load(os.path.join(GOLD, "dit_train_t2v_L13.npz"))
    # distilled_trainer.py:265-289: t = 1000, batched noise tensor, loss on sample 0 broadcast against the batch
    out = m(noise.cuda(), t=torch.ones(2, device="cuda") * 1000.0, context=[c.cuda() for c in cl], seq_len=24)
    assert out[0].requires_grad and out[0].dtype == torch.float32
    loss = torch.nn.functional.mse_loss(out[0], vt.cuda())
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * float(g["loss"])
    loss.backward()
    params = dict(m.named_parameters())
    worst = {1: 0.0, 2: 0.0}
    for name in g.files:
        if name in params:
            got = params[name].grad
            assert got is not None, name
            e = rel_rms(got, torch.from_numpy(g[name]))
            worst[min(got.dim(), 2)] = max(worst[min(got.dim(), 2)], e)
            assert e < (TOL_GRAD if got.dim() > 1 else TOL_GRAD_1D), (name, e)
    print(f"[measured] reference golden gradients (t2v, 13 layers): worst matrix {worst[2]:.3e}, worst 1-D {worst[1]:.3e}")
    # the reference's block_idx > 10 quirk: those FFN weights get no gradient at all (model.py:317-324)
    first = int(g["ffn_grad_none_from"])
    for i in range(13):
        gw = m.blocks[i].ffn[0].weight.grad
        assert (gw is None) == (i >= first)


@pytest.mark.parametrize("freeze", [True, False])
def test_all_gradients_match_autograd_oracle(wan_model_mod, freeze):
    from oracle import wan_dit_oracle as O
    cfg, sd, m, noise, vt, cl = _setup(wan_model_mod, freeze)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oo = O.dit_forward_autograd(osd, cfg, list(noise), torch.ones(2) * 1000.0, cl, 24, reference_ffn_freeze=freeze)
    # use both samples so every path carries gradient
    lo = sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(oo, vt))
    lo.backward()
    out = m(list(noise.cuda()), t=torch.ones(2, device="cuda") * 1000.0, context=[c.cuda() for c in cl], seq_len=24)
    lg = sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, vt.cuda()))
    lg.backward()
    assert abs(lg.item() - lo.item()) < 2e-2 * lo.item()
    bad = []
    worst = {1: 0.0, 2: 0.0}
    norms = sorted(float(v.grad.norm()) for v in osd.values() if v.grad is not None and float(v.grad.abs().max()) > 0)
    floor = 1e-2 * norms[len(norms) // 2]      # near-null gradients (e.g. the cross-attention K bias, to which the
    for name, p in m.named_parameters():       # softmax is invariant) are compared on the absolute scale instead
        og = osd[name].grad
        if og is None or float(og.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        err = float((p.grad.double().cpu() - og.double()).norm() / max(float(og.double().norm()), floor))
        worst[min(og.dim(), 2)] = max(worst[min(og.dim(), 2)], err)
        # matrices: TOL_GRAD; 1-D parameters (bias / gain gradients are long sums with heavy cancellation, so the
        # same bf16 operand noise is a larger fraction of the result): TOL_GRAD_1D
        if err > (TOL_GRAD if og.dim() > 1 else TOL_GRAD_1D):
            bad.append((name, err))
    print(f"[measured] all gradients vs autograd oracle (freeze={freeze}): worst matrix {worst[2]:.3e}, worst 1-D {worst[1]:.3e}")
    assert not bad, bad[:10]


def test_backward_kernels(ops):
    """Unit checks of the backward kernels against autograd on the same formulas."""
    torch.manual_seed(3)
    x = torch.randn(70, 200, device="cuda").to(torch.bfloat16)
    assert torch.equal(ops.transpose_bf16(x)[:, :70], x.t())
    acc = torch.ones(200, device="cuda")
    assert rel_rms(ops.colsum_accum(x, acc), 1 + x.float().sum(0)) < 1e-5
    # column sums on the shapes of the step (bias gradients), strided rows, odd widths, one row, bf16 and fp32
    for R, C, ld, dt in [(6240, 1536, 1536, torch.bfloat16), (333, 8960, 8960, torch.bfloat16), (50, 204, 208, torch.float32),
                         (97, 201, 201, torch.bfloat16), (1, 3072, 3072, torch.float32), (4001, 1536, 3072, torch.bfloat16)]:
        full = torch.randn(R, ld, device="cuda").to(dt)
        xs = full[:, :C]
        acc = torch.full((C,), 2.0, device="cuda")
        assert rel_rms(ops.colsum_accum(xs, acc), 2 + xs.float().sum(0)) < 2e-5, (R, C, ld, dt)
    # the same sums batched into one launch per 16 matrices (the bias gradients of a block backward)
    mats = [torch.randn(R, C, device="cuda").to(dt) for R, C, dt in
            [(6240, 1536, torch.bfloat16), (6240, 8960, torch.bfloat16), (2048, 3072, torch.float32), (1, 8, torch.bfloat16),
             (97, 201, torch.bfloat16)] + [(33 + i, 40 + 3 * i, torch.bfloat16) for i in range(14)]]
    outs = [torch.full((m.shape[1],), 3.0, device="cuda") for m in mats]
    ops.colsum_accum_multi(list(zip(mats, outs)))                 # 19 entries: two launches
    for m, o in zip(mats, outs):
        assert rel_rms(o, 3 + m.float().sum(0)) < 2e-5, tuple(m.shape)
    # GELU
    xp = torch.randn(64, 64, device="cuda").to(torch.bfloat16)
    dy = torch.randn(64, 64, device="cuda").to(torch.bfloat16)
    xf = xp.float().requires_grad_(True)
    torch.nn.functional.gelu(xf, approximate="tanh").backward(dy.float())
    assert rel_rms(ops.gelu_tanh_bwd(dy, xp).float(), xf.grad) < 5e-3
    assert rel_rms(ops.gelu_tanh(xp).float(), torch.nn.functional.gelu(xp.float(), approximate="tanh")) < 5e-3
    # LayerNorm + modulate backward
    B, S, d = 2, 9, 256
    xx = (torch.randn(B * S, d, device="cuda") * 2 + 0.3).requires_grad_(True)
    mod = torch.randn(6, d, device="cuda", requires_grad=True)
    e0 = torch.randn(B, 6, d, device="cuda", requires_grad=True)
    gy = torch.randn(B * S, d, device="cuda")
    e = (mod[None] + e0)
    y = torch.nn.functional.layer_norm(xx, (d,), eps=1e-6).view(B, S, d) * (1 + e[:, 1:2]) + e[:, 0:1]
    y.backward(gy.view(B, S, d))
    dx = torch.zeros(B * S, d, device="cuda")
    d_eb = torch.zeros(B, 6, d, device="cuda")
    ops.layernorm_modulate_bwd_raw(ops.ptr(xx.detach()), ops.ptr(gy), ops.ptr(dx), B * S, d, 1e-6, 1.0,
                                   ops.ptr(mod.detach(), d), ops.ptr(e0.detach(), d), 6 * d, ops.ptr(d_eb, d),
                                   ops.ptr(d_eb, 0), 6 * d, S)
    assert rel_rms(dx, xx.grad) < 1e-4
    assert rel_rms(d_eb[:, :2], e0.grad[:, :2]) < 1e-4
    # RMSNorm + RoPE backward
    from oracle import wan_dit_oracle as O
    N, D = 2, 128
    dd = N * D
    grids = [(1, 3, 3), (1, 2, 4)]
    xq = torch.randn(B * S, dd, device="cuda")
    w = (torch.rand(dd, device="cuda") + 0.5)
    ang = O.rope_table(D)
    gq = torch.randn(B * S, dd, device="cuda")
    xc = xq.cpu().requires_grad_(True)
    wc = w.cpu().requires_grad_(True)
    yq = O.rope_apply(O.rms_norm(xc, wc, 1e-6).view(B, S, N, D), grids, ang)
    yq.backward(gq.cpu().view(B, S, N, D))
    dxq = torch.empty(B * S, dd, dtype=torch.bfloat16, device="cuda")
    dw = torch.zeros(dd, device="cuda")
    cos, sin = torch.cos(ang).float().cuda(), torch.sin(ang).float().cuda()
    grid = torch.tensor(grids, dtype=torch.int32, device="cuda")
    ops.rmsnorm_rope_bwd_raw(ops.ptr(xq), dd, ops.ptr(gq), dd, ops.ptr(dxq), dd, ops.ptr(dw), B * S, dd, ops.ptr(w),
                             1e-6, 1, ops.ptr(cos), ops.ptr(sin), 1024, D, ops.ptr(grid), S)
    assert rel_rms(dxq.float(), xc.grad) < 5e-3
    assert rel_rms(dw, wc.grad) < 1e-4


def test_adamw_and_ema_match_torch(omh):
    import importlib
    optim = importlib.import_module("omnihuman-1-hack_amd.optim")
    torch.manual_seed(0)
    p0 = torch.randn(1000, device="cuda")
    a = torch.nn.Parameter(p0.clone())
    b = torch.nn.Parameter(p0.clone())
    oa = optim.AdamW([a], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ob = torch.optim.AdamW([b], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for i in range(5):
        g = torch.randn(1000, device="cuda")
        a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)
    ema, src = torch.nn.Linear(8, 8).cuda(), torch.nn.Linear(8, 8).cuda()
    ref = [0.995 * t.detach().clone() + 0.005 * s.detach() for t, s in zip(ema.parameters(), src.parameters())]
    optim.update_ema_model(ema, src, 0.995)
    for t, r in zip(ema.parameters(), ref):
        assert torch.allclose(t, r, atol=1e-7)


def test_training_step_function_matches_reference_loss(wan_model_mod):
    """trainer.training_step keeps distilled_trainer.py's semantics (sample 0 only, broadcast loss)."""
    import importlib
    trainer = importlib.import_module("omnihuman-1-hack_amd.trainer")
    cfg, sd, m, noise, vt, cl = _setup(wan_model_mod)
    # reshape the tiny case into the trainer's batch layout: contexts as one [B, L, C] tensor
    ctx = torch.stack(cl)
    g = np.load(os.path.join(GOLD, "dit_train_t2v_L13.npz"))
    loss = trainer.training_step((noise, ctx, vt), m, num_train_timesteps=1000)
    assert abs(loss - float(g["loss"])) < 2e-2 * float(g["loss"])
    assert rel_rms(m.blocks[0].self_attn.q.weight.grad, torch.from_numpy(g["blocks.0.self_attn.q.weight"])) < TOL_GRAD


def test_optimizer_step_reaches_the_next_forward(wan_model_mod):
    """AdamW / EMA write parameters through raw pointers: the packed bf16 weight copies the forward uses must be
    rebuilt afterwards (a second step that still saw the old weights would train nothing)."""
    import importlib
    trainer = importlib.import_module("omnihuman-1-hack_amd.trainer")
    optim = importlib.import_module("omnihuman-1-hack_amd.optim")
    cfg, sd, m, noise, vt, cl = _setup(wan_model_mod)
    ctx = torch.stack(cl)
    opt = optim.AdamW(m.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    args = (noise.cuda(), torch.ones(2, device="cuda") * 1000.0, [c.cuda() for c in cl], 24)
    with torch.no_grad():
        before = torch.stack(m(*args))
    loss0 = trainer.training_step((noise, ctx, vt), m, num_train_timesteps=1000)
    opt.step()
    opt.zero_grad(set_to_none=True)
    with torch.no_grad():
        after = torch.stack(m(*args))
    assert rel_rms(after, before) > 1e-3, "the forward did not see the optimizer update"
    fresh = wan_model_mod.WanModel(num_layers=13, **__import__("oracle.make_golden", fromlist=["TINY"]).TINY)
    fresh.load_state_dict(m.state_dict())
    fresh = fresh.cuda().eval()
    with torch.no_grad():
        ref = torch.stack(fresh(*args))
    assert torch.equal(after, ref)
    # and the EMA copy likewise
    ema = wan_model_mod.WanModel(num_layers=13, **__import__("oracle.make_golden", fromlist=["TINY"]).TINY)
    ema.load_state_dict(sd)
    ema = ema.cuda().eval()
    with torch.no_grad():
        e0 = torch.stack(ema(*args))
    optim.update_ema_model(ema, m, 0.5)
    with torch.no_grad():
        e1 = torch.stack(ema(*args))
    assert rel_rms(e1, e0) > 1e-4
    loss1 = trainer.training_step((noise, ctx, vt), m, num_train_timesteps=1000)
    assert loss1 != loss0


@pytest.mark.parametrize("R,C,ld_in,batch", [(70, 130, 130, 1), (64, 64, 64, 1), (1560, 1536, 3072, 1), (97, 200, 200, 3),
                                            (513, 77, 77, 2), (40, 8960, 8960, 1)])
def test_transpose_bf16(ops, R, C, ld_in, batch):
    """omh_transpose_bf16: ragged tile edges, strided input rows, batches, aligned and odd output pitch; pad columns
    of the output stay untouched.  (A variant with 16-byte global accesses on both sides measured SLOWER on every
    training shape but the largest — 29 vs 10 us at 1560x1536, 50 vs 58 us at 6240x8960 — and was dropped.)"""
    g = torch.Generator(device="cuda").manual_seed(R * 31 + C)
    x = torch.randn(batch, R, ld_in, device="cuda", generator=g).bfloat16()
    for ld_out in ((R + 7) // 8 * 8 + 8, R if R % 2 else R + 1):        # aligned and odd row pitch
        out = torch.full((batch, C, ld_out), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.transpose_bf16_raw(ops.ptr(x), ops.ptr(out), R, C, ld_in, ld_out, batch=batch, bs_in=R * ld_in,
                               bs_out=C * ld_out)
        assert torch.equal(out[:, :, :R], x[:, :, :C].transpose(1, 2))
        assert bool((out[:, :, R:] == 7.0).all())


def test_i2v_training_gradients(wan_model_mod):
    """Training backward of the i2v backbone (image-token branch of the cross-attention with k_img / v_img /
    norm_k_img, img_emb = LayerNorm-Linear-GELU(erf)-Linear-LayerNorm on the CLIP tokens, 36 input channels):
    against gradients produced by the REAL reference (dit_train_i2v_L2.npz) and, for every parameter, against the
    autograd oracle."""
    from oracle import make_golden, wan_dit_oracle as O, detgen
    cfg, tag, xs, ctx, tt, seq_len, ys, clip = make_golden.tiny_case("i2v", 2)
    sd = O.synth_state_dict(cfg, tag)
    vt = torch.from_numpy(detgen.normalish(f"{tag}/vt", (16, 2, 6, 8)))
    t1000 = torch.tensor([1000.0, 1000.0])
    m = wan_model_mod.WanModel(model_type="i2v", in_dim=36, num_layers=2, **make_golden.TINY)
    m.load_state_dict(sd)
    m = m.cuda().train()
    out = m([u.cuda() for u in xs], t=t1000.cuda(), context=[c.cuda() for c in ctx], seq_len=seq_len,
            clip_fea=clip.cuda(), y=[u.cuda() for u in ys])
    loss = torch.nn.functional.mse_loss(out[0], vt.cuda())
    loss.backward()
    g = np.load(os.path.join(GOLD, "dit_train_i2v_L2.npz"))
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * float(g["loss"])
    params = dict(m.named_parameters())
    for name in g.files:
        if name == "loss":
            continue
        got = params[name].grad
        assert got is not None, name
        ref = torch.from_numpy(g[name])
        got = got if got.numel() <= 100000 else got[:16]
        assert rel_rms(got, ref) < TOL_GRAD, (name, rel_rms(got, ref))
    # every parameter against the autograd oracle
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oo = O.dit_forward_autograd(osd, cfg, xs, t1000, ctx, seq_len, clip_fea=clip, y=ys)
    torch.nn.functional.mse_loss(oo[0], vt).backward()
    norms = sorted(float(v.grad.norm()) for v in osd.values() if v.grad is not None and float(v.grad.abs().max()) > 0)
    floor = 1e-2 * norms[len(norms) // 2]
    bad = []
    worst = {1: 0.0, 2: 0.0}
    for name, p in m.named_parameters():
        og = osd[name].grad
        if og is None or float(og.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        err = float((p.grad.double().cpu() - og.double()).norm() / max(float(og.double().norm()), floor))
        worst[min(og.dim(), 2)] = max(worst[min(og.dim(), 2)], err)
        if err > (TOL_GRAD if og.dim() > 1 else TOL_GRAD_1D):
            bad.append((name, err))
    print(f"[measured] i2v gradients vs autograd oracle: worst matrix {worst[2]:.3e}, worst 1-D {worst[1]:.3e}")
    assert not bad, bad[:10]


def test_checkpoint_save_resume_in_the_trainers_formats(wan_model_mod, tmp_path):
    """distilled_trainer.py:153-178: save after two steps, resume into fresh objects, continue — the resumed run
    reaches the weights of the uninterrupted one (to the gradients' fp32-atomics noise), through the accelerate
    layout, through the reference's manual pytorch_model.bin alone, and with a torch.optim.AdamW optimizer state."""
    import os
    trainer = importlib.import_module("omnihuman-1-hack_amd.trainer")
    optim = importlib.import_module("omnihuman-1-hack_amd.optim")
    cfg, sd, m, noise, vt, cl = _setup(wan_model_mod, True)
    batch = (noise.cuda(), torch.stack([c for c in cl]).cuda() if cl[0].shape == cl[1].shape else None, vt.cuda())
    if batch[1] is None:
        L = max(c.shape[0] for c in cl)
        ctx = torch.zeros(len(cl), L, cl[0].shape[1])
        for i, c in enumerate(cl):
            ctx[i, :c.shape[0]] = c
        batch = (noise.cuda(), ctx.cuda(), vt.cuda())

    def fresh():
        from oracle import make_golden
        mm = wan_model_mod.WanModel(num_layers=13, **make_golden.TINY)
        mm.load_state_dict(sd)
        mm = mm.cuda().train()
        return mm, optim.AdamW(mm.parameters(), lr=1e-3, weight_decay=0.01)

    def steps(mm, oo, n):
        for _ in range(n):
            trainer.training_step(batch, mm)
            oo.step()
            oo.zero_grad(set_to_none=True)

    m_a, o_a = fresh()
    steps(m_a, o_a, 2)
    ck = trainer.save_checkpoint(str(tmp_path / "checkpoint_2"), m_a, o_a, step=2, epoch=0)
    assert {"model.safetensors", "optimizer.bin", "pytorch_model.bin", "random_states_0.pkl"} <= set(os.listdir(ck))
    ema = trainer.save_ema(str(tmp_path / "ema_model_step_2.pt"), m_a)
    steps(m_a, o_a, 2)
    # (1) resume from the accelerate layout
    m_b, o_b = fresh()
    info = trainer.load_checkpoint(ck, m_b, o_b)
    assert info["step"] == 2 and info["epoch"] == 0
    steps(m_b, o_b, 2)
    for (n, a), (_, b) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert rel_rms(b.detach(), a.detach()) < 1e-3, n
    assert int(o_b.state[next(iter(m_b.parameters()))]["step"]) == 4
    # (2) the reference's manual fallback file alone
    os.remove(os.path.join(ck, "model.safetensors"))
    os.remove(os.path.join(ck, "optimizer.bin"))
    m_c, o_c = fresh()
    assert trainer.load_checkpoint(ck, m_c, o_c)["step"] == 2
    steps(m_c, o_c, 2)
    for (n, a), (_, c) in zip(m_a.named_parameters(), m_c.named_parameters()):
        assert rel_rms(c.detach(), a.detach()) < 1e-3, n
    # (3) an optimizer state written by torch.optim.AdamW (tensor step counts) loads into the fused optimizer
    m_d, _ = fresh()
    t_opt = torch.optim.AdamW(m_d.parameters(), lr=1e-3, weight_decay=0.01)
    trainer.training_step(batch, m_d)
    t_opt.step()
    blob = {"model": m_d.state_dict(), "optimizer": t_opt.state_dict(), "scaler": None, "step": 1, "epoch": 0}
    d2 = tmp_path / "checkpoint_torch"
    d2.mkdir()
    torch.save(blob, str(d2 / "pytorch_model.bin"))
    m_e, o_e = fresh()
    trainer.load_checkpoint(str(d2), m_e, o_e)
    steps(m_e, o_e, 1)
    assert int(o_e.state[next(iter(m_e.parameters()))]["step"]) == 2
    # EMA file round trip (eval_ema.py:43-47)
    m_f, _ = fresh()
    trainer.load_ema(ema, m_f)
    assert all(torch.isfinite(p).all() for p in m_f.parameters())
