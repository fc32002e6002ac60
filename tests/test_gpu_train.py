"""Training step (distilled_trainer.py:241-316 semantics) on the HIP path:
loss and gradients against the reference's golden gradients and the autograd oracle."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import PKG, rel_rms, set_option

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# bf16 forward + bf16 backward operands, fp32 accumulation: per-tensor relative RMS error of a
# parameter gradient after 13 blocks of back-propagation.
# measured on MI355X (round 2, printed by the tests): matrices <= 1.01e-2 against the reference's own gradients and
# against the autograd oracle (t2v 13 layers, both freeze settings, i2v); 1-D parameters (bias / gain gradients: long
# sums with heavy cancellation) <= 4.5e-2.  Bounds = 2 x measured.
TOL_GRAD = 2e-2
# Round 4 (profiles/r04_measured_parity_figures.txt), bounds = 2 x measured: the 6.2e-2 / 4.5e-2 of round 3 were the
# cross-attention K biases — gradients that vanish identically (softmax shift invariance), i.e. rounding noise against a
# floor; every other 1-D parameter is within 1.9e-2 (worst: blocks.3.self_attn.k.bias).
TOL_GRAD_NULL = 1.4e-1      # identically-null gradients: 6.8e-2 measured
TOL_GRAD_1D = 4e-2          # 1.87e-2 measured


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
            assert e < TOL_GRAD, (name, e)                 # matrices and 1-D alike: 9.1e-3 / 9.0e-3 measured (round 3)
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
    worst = {0: (0.0, None), 1: (0.0, None), 2: (0.0, None)}
    norms = sorted(float(v.grad.norm()) for v in osd.values() if v.grad is not None and float(v.grad.abs().max()) > 0)
    floor = 1e-2 * norms[len(norms) // 2]      # near-null gradients (e.g. the cross-attention K bias, to which the
    for name, p in m.named_parameters():       # softmax is invariant) are compared on the absolute scale instead
        og = osd[name].grad
        if og is None or float(og.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        err = float((p.grad.double().cpu() - og.double()).norm() / max(float(og.double().norm()), floor))
        # the cross-attention K bias: its gradient is exactly zero in exact arithmetic (softmax shift invariance), what
        # is compared there is rounding noise against the floor — its own bound
        null = name.endswith("cross_attn.k.bias")
        kind = 0 if null else min(og.dim(), 2)
        if err > worst[kind][0]:
            worst[kind] = (err, name)
        # matrices: TOL_GRAD; 1-D parameters (bias / gain gradients are long sums with heavy cancellation, so the
        # same bf16 operand noise is a larger fraction of the result): TOL_GRAD_1D
        if err > (TOL_GRAD_NULL if null else TOL_GRAD if og.dim() > 1 else TOL_GRAD_1D):
            bad.append((name, err))
    print(f"[measured] all gradients vs autograd oracle (freeze={freeze}): worst matrix {worst[2]}, worst 1-D {worst[1]}, "
          f"identically-null gradients {worst[0]}")
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
    # ... and a batch in which every matrix allows 16-byte loads (widths and row pitches multiples of 8): the vector kernel
    big = torch.randn(6240, 4608, device="cuda").to(torch.bfloat16)
    mats = [big[:, :1536], big[:, 1536:], torch.randn(6240, 8960, device="cuda").to(torch.bfloat16),
            torch.randn(2048, 3072, device="cuda"), torch.randn(1, 8, device="cuda").to(torch.bfloat16),
            torch.randn(131, 264, device="cuda"), torch.randn(7, 1536, device="cuda").to(torch.bfloat16)]
    outs = [torch.full((m.shape[1],), -1.5, device="cuda") for m in mats]
    ops.colsum_accum_multi(list(zip(mats, outs)))
    for m, o in zip(mats, outs):
        assert rel_rms(o, -1.5 + m.float().sum(0)) < 2e-5, tuple(m.shape)
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
    # ONE launch for all tensors (omh_ema_update_multi) == the per-tensor kernel bit for bit: odd sizes, several chunks,
    # a tensor that starts 4 bytes off the 16-byte grid (the scalar route), twice in a row (cached pointer table)
    ops = importlib.import_module("omnihuman-1-hack_amd.ops")

    class Bag(torch.nn.Module):
        def __init__(self, sizes, seed):
            super().__init__()
            g = torch.Generator().manual_seed(seed)
            base = torch.randn(20000, generator=g).cuda()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(n, generator=g).cuda()) for n in sizes]
                                             + [torch.nn.Parameter(base[1:1 + 4099])])
    sizes = [1, 3, 4096, 4097, 8192 + 3, 70001]
    ema2, src2 = Bag(sizes, 1), Bag(sizes, 2)
    for _ in range(2):
        want = []
        for t, s_ in zip(ema2.parameters(), src2.parameters()):
            w = t.detach().clone()
            ops.ema_update(w, s_.detach().contiguous(), 0.9)
            want.append(w)
        v0 = [t._version for t in ema2.parameters()]
        optim.update_ema_model(ema2, src2, 0.9)
        for t, w, v in zip(ema2.parameters(), want, v0):
            assert torch.equal(t.detach(), w) and t._version > v


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
        null = name.endswith("cross_attn.k.bias") or name.endswith("cross_attn.k_img.bias")   # identically-null gradients
        kind = 0 if null else min(og.dim(), 2)
        worst[kind] = max(worst.get(kind, 0.0), err)
        if err > (TOL_GRAD_NULL if null else TOL_GRAD if og.dim() > 1 else TOL_GRAD_1D):
            bad.append((name, err))
    print(f"[measured] i2v gradients vs autograd oracle: worst matrix {worst[2]:.3e}, worst 1-D {worst[1]:.3e}, "
          f"identically-null {worst.get(0, 0.0):.3e}")
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
    assert {"model.safetensors", "optimizer.bin", "train_state.json", "random_states_0.pkl"} <= set(os.listdir(ck))
    assert "pytorch_model.bin" not in os.listdir(ck)                          # the model is written once
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
    assert info["rng_restored"]
    # (2) the reference's manual fallback file alone (distilled_trainer.py:166-173)
    m_a2, o_a2 = fresh()
    trainer.load_checkpoint(ck, m_a2, o_a2)                                    # the same step-2 state, rewritten
    ck2 = trainer.save_checkpoint(str(tmp_path / "checkpoint_2_manual"), m_a2, o_a2, step=2, epoch=0, manual_fallback=True)
    assert "pytorch_model.bin" in os.listdir(ck2) and "model.safetensors" not in os.listdir(ck2)
    m_c, o_c = fresh()
    assert trainer.load_checkpoint(ck2, m_c, o_c)["step"] == 2
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


# ------------------------------------------------------------------------------------------------ round 3: fused epilogues
def test_gemm_training_epilogues(ops):
    """ABI v5 epilogues of omh_gemm_bf16 against the un-fused kernels they replace: out-of-place gated residual with
    the branch output on the side, GELU with the pre-activation on the side, GELU' in the dgrad epilogue."""
    g = torch.Generator(device="cuda").manual_seed(5)
    for M, N, K, S in [(1560, 1536, 1536, 1560), (6240, 1536, 1536, 1560), (333, 264, 136, 111)]:
        a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
        b = torch.randn(N, device="cuda", generator=g)
        x = torch.randn(M, N, device="cuda", generator=g)
        g0 = torch.randn(N, device="cuda", generator=g)
        g1 = torch.randn((M + S - 1) // S, 6 * N, device="cuda", generator=g)
        # reference: in place, no aux
        ref = x.clone()
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(ref), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(b), bias_mode=ops.BIAS_N,
                     gate0=ops.ptr(g0), gate1=ops.ptr(g1, 2 * N), gate1_stride=6 * N, gate_rows=S, gate_const=0.0)
        y_ref = ops.gemm(a, w, bias=b, epilogue=ops.EPI_BF16)
        xin = x.clone()
        out = torch.full_like(x, 7.0)
        y = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(b), bias_mode=ops.BIAS_N,
                     gate0=ops.ptr(g0), gate1=ops.ptr(g1, 2 * N), gate1_stride=6 * N, gate_rows=S, gate_const=0.0,
                     c_in=ops.ptr(xin), aux=ops.ptr(y), ldaux=N)
        assert torch.equal(out, ref) and torch.equal(xin, x) and torch.equal(y, y_ref), (M, N, K)
        # GELU + pre-activation
        u_ref = ops.gemm(a, w, bias=b, epilogue=ops.EPI_GELU_BF16)
        u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(u), M, N, K, K, K, N, ops.EPI_GELU_BF16, bias=ops.ptr(b), bias_mode=ops.BIAS_N,
                     aux=ops.ptr(pre), ldaux=N)
        assert torch.equal(u, u_ref) and torch.equal(pre, y_ref)
        # GELU' epilogue: (a w^T) * gelu'(pre)  ==  gelu_bwd(bf16(a w^T), pre) up to the one rounding it saves
        d_un = ops.gelu_tanh_bwd(ops.gemm(a, w, epilogue=ops.EPI_BF16), pre)
        d_f = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(d_f), M, N, K, K, K, N, ops.EPI_GELU_BWD_BF16, aux=ops.ptr(pre), ldaux=N)
        xf = pre.float().requires_grad_(True)
        torch.nn.functional.gelu(xf, approximate="tanh").backward((a.float() @ w.float().t()))
        assert rel_rms(d_f.float(), xf.grad) < 4e-3 and rel_rms(d_f.float(), d_un.float()) < 6e-3, (M, N, K)


def test_attention_backward_prescaled_q_and_bf16_outputs(ops):
    """omh_flash_attn_bwd_d128 on the forward's pre-scaled q (q' = bf16(q scale log2 e)) with bf16 gradients written
    into the column blocks of one [rows, 3d] buffer, against the fp32 / plain-q path on the same problem."""
    torch.manual_seed(9)
    B, H, Lq, Lk, D = 2, 2, 200, 136, 128
    d = H * D
    q = torch.randn(B * Lq, d, device="cuda")
    k = torch.randn(B * Lk, d, device="cuda").bfloat16()
    v = torch.randn(B * Lk, d, device="cuda").bfloat16()
    do = torch.randn(B * Lq, d, device="cuda").bfloat16()
    klens = torch.tensor([Lk, 77], dtype=torch.int32, device="cuda")
    scale = D ** -0.5
    qs = (q * (scale * 1.4426950408889634)).bfloat16()               # what rmsnorm_rope_bf16(out_scale) hands the forward
    q_plain = (qs.float() / (scale * 1.4426950408889634)).bfloat16()  # a plain q representing (almost) the same values

    def fwd(qq, pres):
        Lp = (Lk + 63) // 64 * 64
        vt = torch.zeros(B, d, Lp, device="cuda", dtype=torch.bfloat16)
        ops.transpose_bf16_raw(ops.ptr(v), ops.ptr(vt), Lk, d, d, Lp, batch=B, bs_in=Lk * d, bs_out=d * Lp)
        o = torch.empty(B * Lq, d, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, Lq, device="cuda")
        ops.flash_attn_raw(ops.ptr(qq), ops.ptr(k), ops.ptr(vt), ops.ptr(o), ops.ptr(klens), B, H, Lq, Lk, Lq * d, d, Lk * d,
                           d, d * Lp, Lq * d, d, Lp, scale, lse=ops.ptr(lse), q_prescaled=int(pres))
        return o, lse
    o1, lse1 = fwd(qs, True)
    o0, lse0 = fwd(q_plain, False)
    assert rel_rms(o1.float(), o0.float()) < 1e-2
    dq0, dk0, dv0 = ops.flash_attn_bwd(q_plain, k, v, o0, do, lse0, klens, B, H, Lq, Lk, scale)
    buf = torch.full((B * Lq, 3 * d), 3.0, device="cuda", dtype=torch.bfloat16)
    kvb = torch.full((B * Lk, 2 * d), 3.0, device="cuda", dtype=torch.bfloat16)
    ops.flash_attn_bwd(qs, k, v, o1, do, lse1, klens, B, H, Lq, Lk, scale, q_prescaled=True,
                       out=(buf[:, d:2 * d], kvb[:, :d], kvb[:, d:]))
    assert bool((buf[:, :d] == 3.0).all()) and bool((buf[:, 2 * d:] == 3.0).all())
    for got, want, nm in ((buf[:, d:2 * d], dq0, "dq"), (kvb[:, :d], dk0, "dk"), (kvb[:, d:], dv0, "dv")):
        e = rel_rms(got.float(), want)
        assert e < 1.2e-2, (nm, e)          # bf16 rounding of the output + the 2^-9 perturbation of q between the two


@pytest.mark.parametrize("B,H,Lq,Lk,lens", [(4, 12, 1560, 1560, None), (1, 12, 1560, 1560, None),
                                            (4, 12, 1560, 512, [512, 40, 0, 300]), (1, 12, 1560, 512, [120])])
def test_attention_tail_split_matches_the_unsplit_launch(ops, monkeypatch, B, H, Lq, Lk, lens):
    """ABI v8: the last, partly filled round of workgroups of the short-sequence forward (OMH_ATTN_ALLOW_SPLIT) and of
    the dQ / dK-dV kernels (omh_attn_bwd_args.workspace) is split over the inner loop, the workers' fp32 partial results
    combined in a fixed order.  At the training step's real shapes (4 clips x 12 heads x 1560 tokens: 624 workgroups on
    512 slots; 1 clip: 156, everything split; the cross-attention with ragged key lengths): rows outside the split
    tiles equal the unsplit launch bit for bit, the split rows to fp32-association / one-bf16-ulp noise, repeatably."""
    torch.manual_seed(B * 7 + Lk)
    D = 128
    d = H * D
    scale = D ** -0.5
    q = (torch.randn(B * Lq, d, device="cuda") * (scale * 1.4426950408889634)).bfloat16()
    k = torch.randn(B * Lk, d, device="cuda").bfloat16()
    v = torch.randn(B * Lk, d, device="cuda").bfloat16()
    do = torch.randn(B * Lq, d, device="cuda").bfloat16()
    klens = None if lens is None else torch.tensor(lens, dtype=torch.int32, device="cuda")
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros(B, d, Lp, device="cuda", dtype=torch.bfloat16)
    ops.transpose_bf16_raw(ops.ptr(v), ops.ptr(vt), Lk, d, d, Lp, batch=B, bs_in=Lk * d, bs_out=d * Lp)

    def fwd(flags):
        o = torch.empty(B * Lq, d, device="cuda", dtype=torch.bfloat16)
        o32 = torch.empty(B * Lq, d, device="cuda")
        lse = torch.empty(B, H, Lq, device="cuda")
        ops.flash_attn_raw(ops.ptr(q), ops.ptr(k), ops.ptr(vt), ops.ptr(o), None if klens is None else ops.ptr(klens), B, H,
                           Lq, Lk, Lq * d, d, Lk * d, d, d * Lp, Lq * d, d, Lp, scale, lse=ops.ptr(lse), q_prescaled=1,
                           o32=ops.ptr(o32), flags=flags)
        return o, o32, lse
    # OMH_ATTN_SPLIT=tail: split the last round of ANY launch with >= 4 tiles per worker (the shipped policy splits only
    # launches that do not fill the chip once, and the forward only on long key loops: profiles/r04_attention_split_ab.txt)
    set_option("OMH_ATTN_SPLIT", "tail")
    o0, o320, lse0 = fwd(ops.ATTN_SHORT_KERNEL)
    o1, o321, lse1 = fwd(ops.ATTN_SHORT_KERNEL | ops.ATTN_ALLOW_SPLIT)
    o2, o322, lse2 = fwd(ops.ATTN_SHORT_KERNEL | ops.ATTN_ALLOW_SPLIT)
    assert torch.equal(o1, o2) and torch.equal(o321, o322) and torch.equal(lse1, lse2)           # repeatable
    assert torch.equal(o321.bfloat16(), o1)
    same_rows = (o321 == o320).view(B, Lq, H, D).all(-1)                                          # [B, Lq, H]
    frac_same = float(same_rows.float().mean())
    nwg = ((Lq + 127) // 128) * H * B
    if nwg % 512 == nwg:                                                # everything in the last round: every tile is split
        assert frac_same < 0.5
    else:
        assert frac_same > 0.5 and frac_same < 1.0                       # the full rounds are untouched, the tail did change
    live = torch.isfinite(lse0)
    assert torch.equal(torch.isfinite(lse1), live)
    assert rel_rms(o321, o320) < 2e-3 and float((lse1[live] - lse0[live]).abs().max()) < 1e-4
    set_option("OMH_ATTN_SPLIT", "0")
    o3, o323, lse3 = fwd(ops.ATTN_SHORT_KERNEL | ops.ATTN_ALLOW_SPLIT)
    assert torch.equal(o323, o320) and torch.equal(lse3, lse0)                                     # the override: no split
    # ---- backward on the unsplit forward's tensors: split (default) vs OMH_ATTN_SPLIT=0
    kw = dict(q_prescaled=True, o32=o320)
    ref = ops.flash_attn_bwd(q, k, v, None, do, lse0, klens, B, H, Lq, Lk, scale, **kw)
    set_option("OMH_ATTN_SPLIT", "tail")
    got = ops.flash_attn_bwd(q, k, v, None, do, lse0, klens, B, H, Lq, Lk, scale, **kw)
    again = ops.flash_attn_bwd(q, k, v, None, do, lse0, klens, B, H, Lq, Lk, scale, **kw)
    nosplit = ops.flash_attn_bwd(q, k, v, None, do, lse0, klens, B, H, Lq, Lk, scale, split=False, **kw)
    for g, r, a, n, nm in zip(got, ref, again, nosplit, ("dq", "dk", "dv")):
        assert torch.equal(g, a), nm                                     # fixed-order combine: bit-repeatable
        assert torch.equal(n, r), nm
        assert bool(torch.isfinite(g).all()) and rel_rms(g, r) < 1e-5, (nm, rel_rms(g, r))
        if nm != "dq" or B > 1:                                          # the split path did run (the dQ stream runs ONE workgroup per
            assert not torch.equal(g, r), nm                             # CU: 156 query blocks of one clip cannot be split 2 x on 256 CUs)
    # the phases (two streams in the training step) and bf16 column-block outputs: the same bits as the single call
    delta = torch.empty(B, H, Lq, device="cuda")
    ph = dict(delta=delta, **kw)
    ops.flash_attn_bwd(q, k, v, None, do, lse0, klens, B, H, Lq, Lk, scale, phase=1, **ph)
    dq3, _, _ = ops.flash_attn_bwd(q, k, v, None, do, lse0, klens, B, H, Lq, Lk, scale, phase=2, **ph)
    _, dk3, dv3 = ops.flash_attn_bwd(q, k, v, None, do, lse0, klens, B, H, Lq, Lk, scale, phase=3, **ph)
    assert torch.equal(dq3, got[0]) and torch.equal(dk3, got[1]) and torch.equal(dv3, got[2])
    buf = torch.full((B * Lq, 3 * d), 3.0, device="cuda", dtype=torch.bfloat16)
    kvb = torch.full((B * Lk, 2 * d), 3.0, device="cuda", dtype=torch.bfloat16)
    ops.flash_attn_bwd(q, k, v, None, do, lse0, klens, B, H, Lq, Lk, scale, out=(buf[:, d:2 * d], kvb[:, :d], kvb[:, d:]), **kw)
    assert torch.equal(buf[:, d:2 * d], got[0].bfloat16()) and torch.equal(kvb[:, :d], got[1].bfloat16())
    assert torch.equal(kvb[:, d:], got[2].bfloat16()) and bool((buf[:, :d] == 3.0).all()) and bool((buf[:, 2 * d:] == 3.0).all())
    # the shipped policy: one clip (a launch that does not fill the chip) splits dQ and dK / dV, four clips do not
    set_option("OMH_ATTN_SPLIT", None)
    dflt = ops.flash_attn_bwd(q, k, v, None, do, lse0, klens, B, H, Lq, Lk, scale, **kw)
    for g, r, t_, nm in zip(dflt, ref, got, ("dq", "dk", "dv")):
        if B == 1:
            assert torch.equal(g, t_), nm
        else:
            assert torch.equal(g, r), nm                                 # 624 (192: dK / dV of the cross-attention) workgroups:
                                                                         # the chip is (nearly) full, nothing is split
    od, _, lsed = fwd(ops.ATTN_SHORT_KERNEL | ops.ATTN_ALLOW_SPLIT)
    assert torch.equal(od, o0) and torch.equal(lsed, lse0)               # 25 / 8 key tiles: the forward is not split


@pytest.mark.parametrize("B,H,Lq,Lk,lens", [(2, 2, 200, 300, [300, 170]), (1, 2, 64, 40, None), (4, 12, 1560, 1560, None), (1, 12, 1560, 1560, None),
                                            (2, 3, 700, 512, [512, 0]), (1, 1, 3000, 129, [129])])
def test_attention_backward_streams_equal_the_hip_kernels(ops, B, H, Lq, Lk, lens):
    """The generated dK / dV instruction stream (gen_attn_bwd_w64.py, the default) performs the operations of
    attn_bwd2_dkdv_kernel<4, 1> in the same order: the results are the same BITS — fp32 and bf16 outputs, plain and
    pre-scaled q, key lengths (including 0: lse = -inf), the split workers of a launch that does not fill the chip, the
    phases."""
    D = 128
    d = H * D
    scale = D ** -0.5
    g = torch.Generator(device="cuda").manual_seed(5)
    k, v = [torch.randn(B * Lk, d, device="cuda", generator=g).bfloat16() for _ in range(2)]
    q32 = torch.randn(B * Lq, d, device="cuda", generator=g)
    do = torch.randn(B * Lq, d, device="cuda", generator=g).bfloat16()
    klens = None if lens is None else torch.tensor(lens, dtype=torch.int32, device="cuda")
    for pres in (False, True):
        q = (q32 * (scale * 1.4426950408889634)).bfloat16() if pres else q32.bfloat16()
        s = q.float().view(B, Lq, H, D).transpose(1, 2) @ k.float().view(B, Lk, H, D).transpose(1, 2).transpose(-1, -2)
        s = s * (1 / 1.4426950408889634 if pres else scale)
        if klens is not None:
            s = s.masked_fill((torch.arange(Lk, device="cuda")[None, :] >= klens[:, None])[:, None, None, :], float("-inf"))
        lse = torch.logsumexp(s, -1).contiguous()
        o32 = (torch.nan_to_num(s.softmax(-1)) @ v.float().view(B, Lk, H, D).transpose(1, 2)).transpose(1, 2).reshape(B * Lq, d).contiguous()
        res = {}
        for opt in ("0", "1"):
            set_option("OMH_ATTN_BWD_W64", opt)
            kw = dict(q_prescaled=pres, o32=o32)
            f32 = ops.flash_attn_bwd(q, k, v, None, do, lse, klens, B, H, Lq, Lk, scale, **kw)
            nosplit = ops.flash_attn_bwd(q, k, v, None, do, lse, klens, B, H, Lq, Lk, scale, split=False, **kw)
            kvb = torch.full((B * Lk, 2 * d), 3.0, device="cuda", dtype=torch.bfloat16)
            dqb = torch.empty(B * Lq, d, device="cuda", dtype=torch.bfloat16)
            ops.flash_attn_bwd(q, k, v, None, do, lse, klens, B, H, Lq, Lk, scale, out=(dqb, kvb[:, :d], kvb[:, d:]), **kw)
            delta = torch.empty(B, H, Lq, device="cuda")
            ops.flash_attn_bwd(q, k, v, None, do, lse, klens, B, H, Lq, Lk, scale, phase=1, delta=delta, **kw)
            ph3 = ops.flash_attn_bwd(q, k, v, None, do, lse, klens, B, H, Lq, Lk, scale, phase=3, delta=delta, **kw)
            ph2 = ops.flash_attn_bwd(q, k, v, None, do, lse, klens, B, H, Lq, Lk, scale, phase=2, delta=delta, **kw)
            res[opt] = (f32[1], f32[2], nosplit[1], nosplit[2], kvb, ph3[1], ph3[2], f32[0], nosplit[0], dqb, ph2[0])
        set_option("OMH_ATTN_BWD_W64", None)
        for a_, b_ in zip(res["0"][:7], res["1"][:7]):
            assert bool(torch.isfinite(b_.float()).all()) and torch.equal(a_, b_), (pres, rel_rms(b_.float(), a_.float()))
        # the dQ stream starts S from -lse' (MFMA C operand) where the HIP kernel subtracts it per score: the exponent rounds
        # differently by an ulp, some bf16 dS flip: equal to ~5e-5, bit-repeatable, bf16 rows = the fp32 rows rounded
        for i in (7, 8):
            assert bool(torch.isfinite(res["1"][i].float()).all()) and rel_rms(res["1"][i].float(), res["0"][i].float()) < 3e-4
        assert torch.equal(res["1"][9], res["1"][7].bfloat16()) and torch.equal(res["1"][10], res["1"][7])
        assert torch.equal(res["1"][4], torch.cat([res["1"][0], res["1"][1]], 1).bfloat16())      # the bf16 rows = the fp32 rows, rounded
        assert torch.equal(res["1"][5], res["1"][0]) and torch.equal(res["1"][6], res["1"][1])      # phase 3 alone


@pytest.mark.parametrize("B,H,Lq,Lk,lens", [(2, 2, 200, 136, [136, 77]), (1, 3, 1560, 1560, [1560]), (3, 1, 130, 512, [512, 0, 300]),
                                            (2, 2, 64, 64, [64, 33]), (1, 1, 257, 70, [70])])
def test_attention_backward_round3_kernels(ops, B, H, Lq, Lk, lens):
    """csrc/attention_bwd2.hip (delta from dO . o32, 64-position LDS-DMA tiles, transposed operands gathered with
    ds_read_b64_tr_b16) against round 2's kernels on the same forward — plain and pre-scaled q, fp32 and bf16 outputs,
    ragged tiles, a sample without keys, strided column-block outputs — and against autograd through a masked fp32
    softmax."""
    torch.manual_seed(B * 1000 + Lq + Lk)
    D = 128
    d = H * D
    scale = D ** -0.5
    qf32 = torch.randn(B * Lq, d, device="cuda")
    k = torch.randn(B * Lk, d, device="cuda").bfloat16()
    v = torch.randn(B * Lk, d, device="cuda").bfloat16()
    do = torch.randn(B * Lq, d, device="cuda").bfloat16()
    klens = torch.tensor(lens, dtype=torch.int32, device="cuda")
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros(B, d, Lp, device="cuda", dtype=torch.bfloat16)
    ops.transpose_bf16_raw(ops.ptr(v), ops.ptr(vt), Lk, d, d, Lp, batch=B, bs_in=Lk * d, bs_out=d * Lp)
    for pres in (False, True):
        q = (qf32 * (scale * 1.4426950408889634)).bfloat16() if pres else qf32.bfloat16()
        o = torch.empty(B * Lq, d, device="cuda", dtype=torch.bfloat16)
        o32 = torch.empty(B * Lq, d, device="cuda")
        lse = torch.empty(B, H, Lq, device="cuda")
        ops.flash_attn_raw(ops.ptr(q), ops.ptr(k), ops.ptr(vt), ops.ptr(o), ops.ptr(klens), B, H, Lq, Lk, Lq * d, d, Lk * d, d,
                           d * Lp, Lq * d, d, Lp, scale, lse=ops.ptr(lse), q_prescaled=int(pres), o32=ops.ptr(o32))
        assert torch.equal(o32.bfloat16(), o)                          # the same values before the rounding
        dq1, dk1, dv1 = ops.flash_attn_bwd(q, k, v, o, do, lse, klens, B, H, Lq, Lk, scale, q_prescaled=pres)
        dq2, dk2, dv2 = ops.flash_attn_bwd(q, k, v, o, do, lse, klens, B, H, Lq, Lk, scale, q_prescaled=pres, o32=o32)
        # the same in phases (ABI v7): delta alone, then dQ and dK / dV each alone (in either order): the same bits
        delta = torch.full((B, H, Lq), float("nan"), device="cuda")
        ph = dict(q_prescaled=pres, o32=o32, delta=delta)
        ops.flash_attn_bwd(q, k, v, o, do, lse, klens, B, H, Lq, Lk, scale, phase=1, **ph)
        _, dk3, dv3 = ops.flash_attn_bwd(q, k, v, o, do, lse, klens, B, H, Lq, Lk, scale, phase=3, **ph)
        dq3, _, _ = ops.flash_attn_bwd(q, k, v, o, do, lse, klens, B, H, Lq, Lk, scale, phase=2, **ph)
        assert torch.equal(dq3, dq2) and torch.equal(dk3, dk2) and torch.equal(dv3, dv2)
        for a_, b_, nm in ((dq2, dq1, "dq"), (dk2, dk1, "dk"), (dv2, dv1, "dv")):
            assert torch.isfinite(a_).all()
            e = rel_rms(a_, b_)
            assert e < 3e-3, (nm, pres, e)     # same P / dP / dS products; delta from bf16-P vs fp32-P sums differs by ~1e-4
        # bf16 outputs into column blocks of shared buffers
        buf = torch.full((B * Lq, 3 * d), 3.0, device="cuda", dtype=torch.bfloat16)
        kvb = torch.full((B * Lk, 2 * d), 3.0, device="cuda", dtype=torch.bfloat16)
        ops.flash_attn_bwd(q, k, v, o, do, lse, klens, B, H, Lq, Lk, scale, q_prescaled=pres, o32=o32,
                           out=(buf[:, d:2 * d], kvb[:, :d], kvb[:, d:]))
        assert torch.equal(buf[:, d:2 * d], dq2.bfloat16()) and torch.equal(kvb[:, :d], dk2.bfloat16())
        assert torch.equal(kvb[:, d:], dv2.bfloat16()) and bool((buf[:, :d] == 3.0).all()) and bool((buf[:, 2 * d:] == 3.0).all())
        again = ops.flash_attn_bwd(q, k, v, o, do, lse, klens, B, H, Lq, Lk, scale, q_prescaled=pres, o32=o32)
        assert all(torch.equal(x, y) for x, y in zip(again, (dq2, dk2, dv2)))          # bit-repeatable
    # autograd through a masked fp32 softmax on the bf16 operands (plain q)
    q = qf32.bfloat16()
    qa = q.float().view(B, Lq, H, D).transpose(1, 2).requires_grad_(True)
    ka = k.float().view(B, Lk, H, D).transpose(1, 2).requires_grad_(True)
    va = v.float().view(B, Lk, H, D).transpose(1, 2).requires_grad_(True)
    sc_ = (qa @ ka.transpose(-1, -2)) * scale
    mask = torch.arange(Lk, device="cuda")[None, None, None, :] >= klens.long()[:, None, None, None]
    pa = torch.softmax(sc_.masked_fill(mask, float("-inf")), -1)
    pa = torch.nan_to_num(pa, nan=0.0)                              # a sample without keys
    (pa @ va).backward(do.float().view(B, Lq, H, D).transpose(1, 2))
    o = torch.empty(B * Lq, d, device="cuda", dtype=torch.bfloat16)
    o32 = torch.empty(B * Lq, d, device="cuda")
    lse = torch.empty(B, H, Lq, device="cuda")
    ops.flash_attn_raw(ops.ptr(q), ops.ptr(k), ops.ptr(vt), ops.ptr(o), ops.ptr(klens), B, H, Lq, Lk, Lq * d, d, Lk * d, d,
                       d * Lp, Lq * d, d, Lp, scale, lse=ops.ptr(lse), o32=ops.ptr(o32))
    dq2, dk2, dv2 = ops.flash_attn_bwd(q, k, v, o, do, lse, klens, B, H, Lq, Lk, scale, o32=o32)
    for got, want, nm in ((dq2, qa.grad, "dq"), (dk2, ka.grad, "dk"), (dv2, va.grad, "dv")):
        w = want.transpose(1, 2).reshape(got.shape)
        assert rel_rms(got, w) < 8e-3, (nm, rel_rms(got, w))


def test_rmsnorm_rope_bwd_typed_inputs(ops):
    """omh_rmsnorm_rope_bwd_t: bf16 / fp32 x and dy in every combination, in place on dy, strided column blocks."""
    from oracle import wan_dit_oracle as O
    torch.manual_seed(4)
    B, S, N, D = 2, 9, 2, 128
    dd = N * D
    grids = [(1, 3, 3), (1, 2, 4)]
    ang = O.rope_table(D)
    cos, sin = torch.cos(ang).float().cuda(), torch.sin(ang).float().cuda()
    grid = torch.tensor(grids, dtype=torch.int32, device="cuda")
    w = torch.rand(dd, device="cuda") + 0.5
    xq = torch.randn(B * S, 2 * dd, device="cuda").bfloat16()          # q | k projection, bf16, row stride 2 dd
    gq = torch.randn(B * S, 3 * dd, device="cuda").bfloat16()          # dq | dk | dv buffer
    for x_bf in (True, False):
        for col in (0, 1):
            x_in = xq if x_bf else xq.float()
            xc = xq[:, col * dd:(col + 1) * dd].float().cpu().requires_grad_(True)
            wc = w.cpu().requires_grad_(True)
            yq = O.rope_apply(O.rms_norm(xc, wc, 1e-6).view(B, S, N, D), grids, ang)
            yq.backward(gq[:, col * dd:(col + 1) * dd].float().cpu().view(B, S, N, D))
            buf = gq.clone()
            dw = torch.zeros(dd, device="cuda")
            ops.rmsnorm_rope_bwd_t_raw(ops.ptr(x_in, col * dd), x_bf, 2 * dd, ops.ptr(buf, col * dd), True, 3 * dd,
                                       ops.ptr(buf, col * dd), 3 * dd, ops.ptr(dw), B * S, dd, ops.ptr(w), 1e-6, 1,
                                       ops.ptr(cos), ops.ptr(sin), 1024, D, ops.ptr(grid), S)
            assert rel_rms(buf[:, col * dd:(col + 1) * dd].float(), xc.grad) < 5e-3
            assert rel_rms(dw, wc.grad) < 1e-4
            other = [c for c in range(3) if c != col]
            for c in other:
                assert torch.equal(buf[:, c * dd:(c + 1) * dd], gq[:, c * dd:(c + 1) * dd])


def test_pack_weights_multi(ops):
    """One launch = bf16 copy + transposed bf16 copy of many fp32 matrices (ragged tiles, row pitches, column blocks of
    fused buffers) + the fp32 bias copies."""
    g = torch.Generator(device="cuda").manual_seed(2)
    shapes = [(1536, 1536), (8960, 1536), (1536, 8960), (64, 1536), (130, 70), (7, 5)]
    srcs = [torch.randn(r, c, device="cuda", generator=g) for r, c in shapes]
    bias = torch.randn(5000, device="cuda", generator=g)
    fused = torch.zeros(2 * 1536, 1536, device="cuda", dtype=torch.bfloat16)
    fusedT = torch.zeros(1536, 2 * 1536, device="cuda", dtype=torch.bfloat16)
    rows, outs = [], []
    for i, s in enumerate(srcs):
        r, c = s.shape
        if i == 0:
            dst, dstT, ld, ldt = fused[1536:], fusedT[:, 1536:], 1536, 3072
        else:
            dst = torch.zeros(r, c, device="cuda", dtype=torch.bfloat16)
            dstT = torch.zeros(c, r + (8 if i == 4 else 0), device="cuda", dtype=torch.bfloat16) if i != 3 else None
            ld, ldt = c, (dstT.shape[1] if dstT is not None else 0)
        outs.append((dst, dstT))
        rows.append([s.data_ptr(), dst.data_ptr(), dstT.data_ptr() if dstT is not None else 0, r, c, ld, ldt, 0, 0])
    bcopy = torch.zeros(5000, device="cuda")
    rows.append([bias.data_ptr(), bcopy.data_ptr(), 0, 1, 5000, 5000, 0, 0, 1])
    t0 = 0
    for r in rows:
        r[7] = t0
        t0 += (r[3] * r[4] + 4095) // 4096 if r[8] == 1 else ((r[3] + 63) // 64) * ((r[4] + 63) // 64)
    ops.pack_weights_multi(torch.tensor(rows, dtype=torch.int64).cuda(), len(rows), t0)
    for s, (dst, dstT) in zip(srcs, outs):
        want = s.bfloat16()
        assert torch.equal(dst, want)
        if dstT is not None:
            assert torch.equal(dstT[:, :s.shape[0]], want.t())
    assert torch.equal(bcopy, bias) and bool((fused[:1536] == 0).all()) and bool((fusedT[:, :1536] == 0).all())


@pytest.mark.parametrize("freeze", [True, False])
def test_training_forward_equals_inference_and_checkpoint_equals_kept_activations(wan_model_mod, freeze):
    """(1) the training forward is the inference forward bit for bit (same kernels, same roundings: the loss is taken
    from the values the backward differentiates); (2) use_checkpoint = False (activations kept, model.py:549-553) and
    True (block re-run in the backward, model.py:544-548) give the same gradients — both run the same kernels on the
    same inputs, only fp32 atomic summation order differs."""
    cfg, sd, m, noise, vt, cl = _setup(wan_model_mod, freeze)
    args = dict(t=torch.ones(2, device="cuda") * 1000.0, context=[c.cuda() for c in cl], seq_len=24)
    with torch.no_grad():
        want = m(list(noise.cuda()), **args)
    grads = {}
    m.checkpoint_policy = "always"                           # True = re-run the blocks, whatever HBM is free
    for ck in (False, True):
        m.use_checkpoint = ck
        for p in m.parameters():
            p.grad = None
        out = m(list(noise.cuda()), **args)
        assert m.__dict__["_kept_activations"] is (not ck)
        assert all(torch.equal(a, b) for a, b in zip(out, want)), f"training forward != inference forward (ckpt={ck})"
        sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, vt.cuda())).backward()
        grads[ck] = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}
    worst = 0.0
    for n in grads[True]:
        a, b = grads[True][n], grads[False][n]
        assert (a is None) == (b is None), n
        if a is not None:
            e = float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
            worst = max(worst, e)
            assert e < 1e-3, (n, e)
    print(f"[measured] gradients, re-run block vs kept activations (freeze={freeze}): worst relative difference {worst:.2e}")
    # the default policy: the flag is a memory policy, and these activations fit
    m.checkpoint_policy, m.use_checkpoint = "auto", True
    m(list(noise.cuda()), **args)
    assert m.__dict__["_kept_activations"] is True


def test_norm_backward_round3_kernels(ops):
    """omh_layernorm_modulate_bwd2 / omh_rmsnorm_rope_bwd2: against autograd, the fused follow-up (next branch's gated
    residual backward) against the separate kernel, two column segments in one launch — and bit-repeatable."""
    torch.manual_seed(11)
    B, S, d = 3, 37, 256                                   # 37 rows per batch element: ragged last workgroup
    R = B * S
    xx = (torch.randn(R, d, device="cuda") * 2 + 0.3).requires_grad_(True)
    mod = torch.randn(6, d, device="cuda", requires_grad=True)
    e0 = torch.randn(B, 6, d, device="cuda", requires_grad=True)
    gy = torch.randn(R, d, device="cuda")
    dx0 = torch.randn(R, d, device="cuda")
    e = (mod[None] + e0)
    y = torch.nn.functional.layer_norm(xx, (d,), eps=1e-6).view(B, S, d) * (1 + e[:, 1:2]) + e[:, 0:1]
    y.backward(gy.view(B, S, d))
    ynext = torch.randn(R, d, device="cuda").bfloat16()
    outs = []
    for gy_t in (gy, gy.bfloat16()):
        for rep in range(2):
            dx = dx0.clone()
            d_eb = torch.zeros(B, 6, d, device="cuda")
            dyn = torch.zeros(R, d, device="cuda", dtype=torch.bfloat16)
            ops.layernorm_modulate_bwd2(xx.detach(), gy_t, dx, R, d, 1e-6, 1.0, ops.ptr(mod.detach(), d),
                                        ops.ptr(e0.detach(), d), 6 * d, ops.ptr(d_eb, d), ops.ptr(d_eb, 0), 6 * d, S,
                                        dy_next=dyn, y_next=ynext, gate_const=0.0, gate0=ops.ptr(mod.detach(), 2 * d),
                                        gate1=ops.ptr(e0.detach(), 2 * d), gate1_stride=6 * d, dgate=ops.ptr(d_eb, 2 * d),
                                        dgate_stride=6 * d)
            outs.append((dx, d_eb, dyn))
        tol = 1e-4 if gy_t.dtype == torch.float32 else 6e-3
        dx, d_eb, dyn = outs[-1]
        assert rel_rms(dx - dx0, xx.grad) < tol and rel_rms(d_eb[:, :2], e0.grad[:, :2]) < tol
        assert torch.equal(outs[-1][0], outs[-2][0]) and torch.equal(outs[-1][1], outs[-2][1])     # bit-repeatable
        # the fused follow-up == omh_gated_residual_bwd on the finished dx
        dy_ref = torch.empty(R, d, device="cuda", dtype=torch.bfloat16)
        dg_ref = torch.zeros(B, 6, d, device="cuda")
        ops.gated_residual_bwd_raw(ops.ptr(dx), ops.ptr(ynext), ops.ptr(dy_ref), ops.ptr(dg_ref, 2 * d), 6 * d, R, d, 0.0,
                                   ops.ptr(mod.detach(), 2 * d), ops.ptr(e0.detach(), 2 * d), 6 * d, S)
        assert torch.equal(dyn, dy_ref) and rel_rms(d_eb[:, 2], dg_ref[:, 2]) < 1e-5
    # shared parameters (norm3: dstride 0) under per-batch row groups, follow-up without a gate
    w3 = torch.rand(d, device="cuda") + 0.5
    xr = xx.detach().clone().requires_grad_(True)
    wr = w3.clone().requires_grad_(True)
    br = torch.zeros(d, device="cuda", requires_grad=True)
    (torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-6)).backward(gy)
    dx = dx0.clone()
    dw, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    dyn = torch.zeros(R, d, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_modulate_bwd2(xx.detach(), gy, dx, R, d, 1e-6, 0.0, ops.ptr(w3), None, 0, ops.ptr(dw), ops.ptr(db), 0, S,
                                dy_next=dyn, gate_const=1.0)
    assert rel_rms(dx - dx0, xr.grad) < 1e-4 and rel_rms(dw, wr.grad) < 1e-4 and rel_rms(db, br.grad) < 1e-4
    assert torch.equal(dyn, dx.bfloat16())
    # RMSNorm + RoPE backward, q and k segments in one launch == two launches of the typed round-2 kernel
    from oracle import wan_dit_oracle as O
    N, D = 2, 128
    dd = N * D
    S2 = 9
    grids = [(1, 3, 3), (1, 2, 4), (1, 1, 5)]
    ang = O.rope_table(D)
    cos, sin = torch.cos(ang).float().cuda(), torch.sin(ang).float().cuda()
    grid = torch.tensor(grids, dtype=torch.int32, device="cuda")
    wq, wk = torch.rand(dd, device="cuda") + 0.5, torch.rand(dd, device="cuda") + 0.5
    xq = torch.randn(B * S2, 2 * dd, device="cuda").bfloat16()
    gq = torch.randn(B * S2, 3 * dd, device="cuda").bfloat16()
    ref, dws_ref = gq.clone(), []
    for col, w in ((0, wq), (1, wk)):
        dw = torch.zeros(dd, device="cuda")
        ops.rmsnorm_rope_bwd_t_raw(ops.ptr(xq, col * dd), True, 2 * dd, ops.ptr(ref, col * dd), True, 3 * dd,
                                   ops.ptr(ref, col * dd), 3 * dd, ops.ptr(dw), B * S2, dd, ops.ptr(w), 1e-6, 1,
                                   ops.ptr(cos), ops.ptr(sin), 1024, D, ops.ptr(grid), S2)
        dws_ref.append(dw)
    got = gq.clone()
    dwq, dwk = torch.zeros(dd, device="cuda"), torch.zeros(dd, device="cuda")
    ops.rmsnorm_rope_bwd2(ops.ptr(xq), True, 2 * dd, ops.ptr(got), True, 3 * dd, ops.ptr(got), 3 * dd, B * S2, dd, 1e-6, True,
                          [wq, wk], [dwq, dwk], xq.device, n_seg=2, seg_x=dd, seg_dy=dd, seg_dx=dd, rope_cos=ops.ptr(cos),
                          rope_sin=ops.ptr(sin), rope_len=1024, head_dim=D, grid=ops.ptr(grid), seq_len=S2)
    assert torch.equal(got, ref)
    assert rel_rms(dwq, dws_ref[0]) < 1e-5 and rel_rms(dwk, dws_ref[1]) < 1e-5


@pytest.mark.parametrize("tile", ["small", "big"])
def test_gemm_tn_grouped(ops, tile, monkeypatch):
    """omh_gemm_bf16_tn_grouped: several weight-gradient products in one launch == the single-problem kernel on each
    (ragged tiles, strided operands out of fused buffers, accumulation), and bit-repeatable (no split K, no atomics) —
    on 128 x 128 and on 256 x 256 tiles."""
    set_option("OMH_GEMM_TN_GROUP_TILE", tile)
    g = torch.Generator(device="cuda").manual_seed(8)
    R = 1000
    dyf = (torch.randn(R, 3 * 256, device="cuda", generator=g) * 0.3).bfloat16()       # dq | dk | dv style buffer
    xs = [(torch.randn(R, n, device="cuda", generator=g) * 0.3).bfloat16() for n in (256, 136, 520)]
    probs = [(dyf, xs[0], None), (dyf[:, 256:512], xs[1], None), (dyf[:, :136], xs[2], None), (dyf[:, 512:], xs[0], "acc")]
    want, items = [], []
    for dy, x, acc in probs:
        base = torch.randn(dy.shape[1], x.shape[1], device="cuda", generator=g) if acc else None
        ref = ops.gemm_tn(dy, x, out=base.clone() if acc else None, accumulate=bool(acc))
        want.append(ref)
        out = base.clone() if acc else torch.full((dy.shape[1], x.shape[1]), 7.0, device="cuda")
        items.append((dy, x, out, bool(acc)))
    ops.gemm_tn_grouped(items)
    for (dy, x, out, acc), ref in zip(items, want):
        assert rel_rms(out, ref) < 2e-6, (tuple(out.shape), rel_rms(out, ref))
        assert rel_rms(out, (dy.float().t() @ x.float()) + (0 if not acc else ref - dy.float().t() @ x.float())) < 1e-5
    again = [(dy, x, torch.zeros_like(out), False) for dy, x, out, acc in items[:3]]
    ops.gemm_tn_grouped(again)
    for (_, _, o2, _), (_, _, o1, _) in zip(again, items[:3]):
        assert torch.equal(o2, o1)
    many = [(dyf[:, :64], xs[0][:, :64], torch.zeros(64, 64, device="cuda"), False) for _ in range(15)]   # > one group
    ops.gemm_tn_grouped(many)
    assert all(torch.equal(m_[2], many[0][2]) for m_ in many)


def test_deterministic_mode_repeats_a_training_step_bit_for_bit(wan_model_mod, ops):
    """omh_set_deterministic (ABI v6): the launches that combine partial sums with fp32 atomics — bias-gradient column
    sums, gate gradients, split-K weight gradients, the time-embedding MLP's input gradient — hand each output element
    to one workgroup, so two backward passes from the same state give the SAME bits for every parameter; without the
    mode they agree to the atomics' rounding noise only (and to that noise the two modes agree with each other).  The
    model is wide enough (BASELINE config 3: two clips of the 1.3B model's first blocks) for split K and multi-block
    column sums to occur."""
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    trainer = importlib.import_module(PKG + ".trainer")
    torch.manual_seed(11)
    kw = dict(cfgs.dit_kwargs(cfgs.t2v_1_3B))
    kw["num_layers"] = 2
    with torch.device("cuda"):
        m = model_mod.WanModel(**kw)
        torch.nn.init.xavier_uniform_(m.head.head.weight)
    m.train()
    g = torch.Generator(device="cuda").manual_seed(9)
    batch = (torch.randn(2, 16, 1, 60, 104, device="cuda", generator=g),
             torch.randn(2, 512, 4096, device="cuda", generator=g),
             torch.randn(2, 16, 1, 60, 104, device="cuda", generator=g))

    def grads():
        for p in m.parameters():
            p.grad = None
        trainer.forward_backward(batch, m)
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    assert ops.set_deterministic() is False
    free = grads()
    try:
        assert ops.set_deterministic(True) is True
        a, b = grads(), grads()
    finally:
        ops.set_deterministic(False)
    assert a.keys() == b.keys() == free.keys() and len(a) > 20
    for n in a:
        assert torch.equal(a[n], b[n]), n
        assert rel_rms(a[n], free[n]) < 1e-4, n


def test_block_output_taps_do_not_alias_the_block_backward(wan_model_mod):
    """A forward hook on a block (the reference's discriminator features, seaweed_apt/model.py:150-155) gives the block
    output a second consumer.  The block backward updates its incoming gradient in place — only when the tensor is its
    own (produced by this package's next node); a gradient that comes from anywhere else is copied first.  (a) a tap
    whose backward hands over a buffer it keeps: the buffer is not modified; (b) tap loss + output loss: gradients =
    the sum of the two losses' separate gradients."""
    cfg, sd, m, noise, vt, cl = _setup(wan_model_mod, freeze=False)
    feats = {}
    h = m.blocks[3].register_forward_hook(lambda mod, inp, out: feats.__setitem__("f", out))
    tctx = [c.cuda() for c in cl]
    tt = torch.ones(2, device="cuda") * 1000.0

    def run(w_out, w_tap, G=None):
        for p in m.parameters():
            p.grad = None
        out = m(list(noise.cuda()), t=tt, context=tctx, seq_len=24)
        f = feats["f"]
        assert f.requires_grad and f.shape[-1] == cfg.dim
        loss = 0.0
        if w_out:
            loss = loss + w_out * sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, vt.cuda()))
        if w_tap:
            if G is not None:
                class Tap(torch.autograd.Function):
                    @staticmethod
                    def forward(ctx, x):
                        return x.sum()

                    @staticmethod
                    def backward(ctx, g):
                        return G                                    # a buffer the caller keeps using
                loss = loss + Tap.apply(f)
            else:
                loss = loss + w_tap * (f * torch.linspace(-1, 1, cfg.dim, device="cuda")).sum() / f.numel()
        loss.backward()
        return {n: (None if p.grad is None else p.grad.clone()) for n, p in m.named_parameters()}

    try:
        # (a)
        out = m(list(noise.cuda()), t=tt, context=tctx, seq_len=24)
        G = torch.randn_like(feats["f"]).contiguous()
        G0 = G.clone()
        run(0.0, 1.0, G=G)
        assert torch.equal(G, G0), "the block backward modified a gradient buffer it does not own"
        # (b)
        g_out, g_tap, g_both = run(1.0, 0.0), run(0.0, 1.0), run(1.0, 1.0)
        assert g_tap["blocks.5.self_attn.q.weight"] is None or float(g_tap["blocks.5.self_attn.q.weight"].abs().max()) == 0
        worst = (0.0, None)
        for n in g_both:
            if g_both[n] is None:
                continue
            want = g_out[n].double() + (g_tap[n].double() if g_tap[n] is not None else 0.0)
            e = float((g_both[n].double() - want).norm() / want.norm().clamp_min(1e-20))
            null = n.endswith("cross_attn.k.bias")                  # identically-null gradient: pure rounding noise
            if not null and e > worst[0]:
                worst = (e, n)
            assert e < (1.0 if null else TOL_GRAD_1D), (n, e)
        print(f"[measured] tap + output loss vs the sum of the separate gradients: worst {worst}")
    finally:
        h.remove()


def test_adamw_writes_the_operand_copies_of_the_next_forward(wan_model_mod, monkeypatch):
    """Round 4 (omh_adamw_pack_multi): the optimizer kernel writes the bf16 (and transposed) operand copies of every
    weight it updates, instead of a re-pack launch that re-reads the parameters: the copies equal a forced re-pack of the
    updated parameters bit for bit, and parameters / moments agree with the two-launch arrangement (OMH_ADAMW_PACK=0) on
    the same gradients over three steps."""
    trainer = importlib.import_module("omnihuman-1-hack_amd.trainer")
    optim = importlib.import_module("omnihuman-1-hack_amd.optim")
    mt = importlib.import_module("omnihuman-1-hack_amd.wan.modules.model_train")
    ops_ = importlib.import_module("omnihuman-1-hack_amd.ops")
    outs = []
    was = ops_.set_deterministic(None)
    ops_.set_deterministic(True)                                        # the two runs must see the same gradients, bit for bit
    for fused in ("1", "0"):
        monkeypatch.setenv("OMH_ADAMW_PACK", fused)
        cfg, sd, m, noise, vt, cl = _setup(wan_model_mod, True)
        opt = optim.AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
        batch = (noise.cuda(), torch.stack([torch.nn.functional.pad(c, (0, 0, 0, 32 - c.shape[0])) for c in cl]).cuda(), vt.cuda())
        losses = []
        for it in range(3):
            losses.append(float(trainer.forward_backward(batch, m, reference_loss_quirk=False)))
            opt.step()
            opt.zero_grad(set_to_none=True)
            if it == 0:                                                 # after ONE step both runs have seen the same gradients
                first = ({n: p.detach().clone() for n, p in m.named_parameters()},
                         {n: opt.state[p]["exp_avg_sq"].clone() for n, p in m.named_parameters() if p in opt.state})
        packs = mt.TrainPacks.of(m)
        before = [{k: (v.clone() if v is not None else None) for k, v in blk.items()} for blk in packs.blocks]
        stale = packs.sig != [p._version for p in packs.params]
        packs.sig = None                                                # force the re-pack launch over every entry
        packs.refresh(m)
        after = [{k: (v.clone() if v is not None else None) for k, v in blk.items()} for blk in packs.blocks]
        outs.append((losses, first[0], first[1], before, after, stale))
    ops_.set_deterministic(was)
    (l1, p1, v1, b1, a1, stale1), (l0, p0, v0, b0, a0, stale0) = outs
    assert stale1 is False and stale0 is True                          # fused: the copies were current when the step returned
    # the copies the optimizer kernel wrote ARE the bf16 (and transposed) images of the parameters it wrote: a forced
    # re-pack from those parameters changes nothing, bit for bit
    for blk_b, blk_a in zip(b1, a1):
        for k in blk_b:
            if blk_b[k] is not None:
                assert torch.equal(blk_b[k], blk_a[k]), k
    # ... and the update is AdamW's: against the two-launch arrangement on the same (deterministic) gradients — equal up to
    # the compiler's fma contraction inside the two kernels (measured: losses agree to 2e-5 after three steps at lr 1e-3)
    assert l1[0] == l0[0] and all(abs(a - b) < 1e-3 * abs(b) for a, b in zip(l1, l0))
    for n in p1:                                                        # parameters and second moments after the first step
        assert torch.allclose(p1[n], p0[n], rtol=1e-5, atol=1e-7), n
    for n in v1:
        assert torch.allclose(v1[n], v0[n], rtol=1e-5, atol=1e-20), n


def test_deferred_join_of_the_weight_gradient_stream(wan_model_mod, monkeypatch):
    """The join of the weight-gradient stream at the end of the backward pass (model_train._may_defer_join) instead of
    per block: same weight gradients bit for bit as with the per-block join (the grouped products have no split K), also
    with TWO forwards of the model in one pass (the second finds gradients in place and joins per block) and on a second
    backward into existing .grad tensors; refused when a foreign gradient hook is registered or a .grad exists."""
    mt = importlib.import_module(PKG + ".wan.modules.model_train")
    cfg, sd, m, noise, vt, cl = _setup(wan_model_mod, freeze=False)
    args = dict(t=torch.ones(2, device="cuda") * 1000.0, context=[c.cuda() for c in cl], seq_len=24)

    def grads(two_forwards, accumulate=False):
        if not accumulate:
            m.zero_grad(set_to_none=True)
        out = m(list(noise.cuda()), **args)
        loss = sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, vt.cuda()))
        if two_forwards:
            out2 = m(list((noise * 0.5).cuda()), **args)
            loss = loss + sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out2, vt.cuda()))
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    assert mt._DEFER_JOIN and mt._may_defer_join(m) is True
    for two in (False, True):
        monkeypatch.setattr(mt, "_DEFER_JOIN", True)
        got = grads(two)
        assert not mt._join_pending                                     # the end-of-pass callback ran
        acc = grads(two, accumulate=True)                               # second pass: into the existing .grad tensors
        monkeypatch.setattr(mt, "_DEFER_JOIN", False)
        want = grads(two)
        for n in want:
            if want[n].dim() >= 2 and n.startswith("blocks."):
                assert torch.equal(got[n], want[n]), n
                assert rel_rms(acc[n], 2 * want[n]) < 1e-6, n
            else:                                                        # column sums with fp32 atomics, and what hangs off
                assert rel_rms(got[n], want[n]) < 1e-4, n                # them (modulation -> time embedding)
    monkeypatch.setattr(mt, "_DEFER_JOIN", True)
    assert mt._may_defer_join(m) is True                                 # gradients in place: the blocks add into them
    m.direct_grad_accumulation = False
    assert mt._may_defer_join(m) is False                                # ... autograd does, on the main stream
    del m.direct_grad_accumulation
    m.zero_grad(set_to_none=True)
    p0 = m.blocks[3].self_attn.o.weight
    h = p0.register_post_accumulate_grad_hook(lambda p: None)
    assert mt._may_defer_join(m) is False                                # somebody reads gradients inside the pass
    h.remove()
    assert mt._may_defer_join(m) is True
    h = p0.register_hook(lambda g_: g_)
    assert mt._may_defer_join(m) is False
    h.remove()


def test_gradient_accumulation_adds_into_the_existing_grads(wan_model_mod, monkeypatch):
    """Gradient accumulation (distilled_trainer.py:41,116-134,289: 16 micro-steps per optimizer step).  From the second
    micro-step on the block backward adds INTO the existing .grad tensors (weight-gradient GEMMs with their accumulate
    epilogue on the second stream, 1-D sums through kernels that add into their output) and returns None to autograd:
    the .grad tensors stay the same objects / storage, the matrices equal the autograd route (fresh gradient, then
    grad += on the main stream) bit for bit, the 1-D sums to fp32 rounding; the join of the second stream stays at the end
    of the pass.  ``torch.autograd.grad`` w.r.t. parameters that hold a .grad must return the gradients and leave
    .grad alone; a bf16 / foreign-hooked parameter sends its block down the autograd route."""
    mt = importlib.import_module(PKG + ".wan.modules.model_train")
    for freeze in (False, True):
        cfg, sd, m, noise, vt, cl = _setup(wan_model_mod, freeze=freeze)
        args = dict(t=torch.ones(2, device="cuda") * 1000.0, context=[c.cuda() for c in cl], seq_len=24)

        def loss_of(scale):
            out = m(list((noise * scale).cuda()), **args)
            return sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, vt.cuda()))

        def three_micro_steps(direct):
            m.direct_grad_accumulation = direct
            m.zero_grad(set_to_none=True)
            ptrs, calls = None, []
            for k, scale in enumerate((1.0, 0.5, -0.75)):
                real = mt._grad_targets
                monkeypatch.setattr(mt, "_grad_targets", lambda *a, **kw: calls.append(real(*a, **kw)) or calls[-1])
                loss_of(scale).backward()
                monkeypatch.setattr(mt, "_grad_targets", real)
                assert not mt._join_pending
                if k == 0:
                    ptrs = {n: (id(p.grad), p.grad.data_ptr()) for n, p in m.named_parameters() if p.grad is not None}
            torch.cuda.synchronize()
            return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, ptrs, calls

        want, _, calls0 = three_micro_steps(False)
        assert all(c is None or c == {} for c in calls0)
        got, ptrs, calls1 = three_micro_steps(True)
        nblk = len(m.blocks)
        assert all(c == {} for c in calls1[:nblk]) and all(c for c in calls1[nblk:])      # micro-steps 2, 3: direct
        assert set(got) == set(want)
        for n, p in m.named_parameters():
            if p.grad is None:
                continue
            if n.startswith("blocks."):                                   # still the first micro-step's tensor
                assert (id(p.grad), p.grad.data_ptr()) == ptrs[n], n
            if want[n].dim() >= 2 and n.startswith("blocks."):
                assert torch.equal(got[n], want[n]), n
            else:
                assert rel_rms(got[n], want[n]) < 1e-5, n
        # autograd.grad with gradients in place: returned, not accumulated
        before = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        ps = [m.blocks[2].self_attn.q.weight, m.blocks[2].cross_attn.k.bias, m.blocks[0].modulation]
        gs = torch.autograd.grad(loss_of(1.0), ps)
        torch.cuda.synchronize()
        for n, p in m.named_parameters():
            if p.grad is not None:
                assert torch.equal(p.grad, before[n]), n
        m.zero_grad(set_to_none=True)
        loss_of(1.0).backward()
        torch.cuda.synchronize()
        for p, g_ in zip(ps, gs):
            assert rel_rms(g_, p.grad) < 1e-5
        # a parameter with a foreign hook: its block takes the autograd route (hook fires, values right), the others stay direct
        seen = []
        h = m.blocks[1].self_attn.o.weight.register_post_accumulate_grad_hook(lambda p: seen.append(1))
        calls = []
        real = mt._grad_targets
        monkeypatch.setattr(mt, "_grad_targets", lambda *a, **kw: calls.append(real(*a, **kw)) or calls[-1])
        loss_of(0.5).backward()
        monkeypatch.setattr(mt, "_grad_targets", real)
        h.remove()
        torch.cuda.synchronize()
        assert seen == [1] and sum(c is None for c in calls) == 1 and all(c for c in calls if c is not None)
        ref = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        m.direct_grad_accumulation = False
        m.zero_grad(set_to_none=True)
        loss_of(1.0).backward()
        loss_of(0.5).backward()
        torch.cuda.synchronize()
        for n, p in m.named_parameters():
            if p.grad is not None:
                if p.grad.dim() >= 2 and n.startswith("blocks."):
                    assert torch.equal(p.grad, ref[n]), n
                else:
                    assert rel_rms(p.grad, ref[n]) < 1e-5, n
        del m.direct_grad_accumulation


def test_cross_attention_key_bias_gradient_waits_for_the_norm_backward(wan_model_mod, monkeypatch):
    """The k | v bias gradients of the cross-attention are column sums of dkv AFTER the key norm's backward has rewritten
    the k half in place — on the second stream when the key / value gradient path runs there.  With that stream held back
    (a long sleep queued on it before the pass) a column sum launched from the main stream would read the k half too
    early: the values must equal the one-stream schedule's."""
    mt = importlib.import_module(PKG + ".wan.modules.model_train")
    cfg, sd, m, noise, vt, cl = _setup(wan_model_mod, freeze=False)
    args = dict(t=torch.ones(2, device="cuda") * 1000.0, context=[c.cuda() for c in cl], seq_len=24)

    def grads(side_kv, hold):
        monkeypatch.setattr(mt, "_SIDE_KV", side_kv)
        m.zero_grad(set_to_none=True)
        out = m(list(noise.cuda()), **args)
        loss = sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, vt.cuda()))
        torch.cuda.synchronize()
        if hold:
            with torch.cuda.stream(mt._side_stream(torch.device("cuda", torch.cuda.current_device()))):
                torch.cuda._sleep(int(2.0e8))                              # ~0.1 s: the whole backward is enqueued meanwhile
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters() if "cross_attn" in n and n.endswith(".bias")}

    want = grads(False, False)
    assert any(float(v.abs().max()) > 0 for n, v in want.items() if n.endswith("cross_attn.k.bias"))
    for hold in (False, True):
        got = grads(True, hold)
        for n in want:
            assert rel_rms(got[n], want[n]) < 1e-5, (n, hold)



def test_norm_backward_second_launches_deferred_to_one_launch(wan_model_mod, monkeypatch):
    """ABI v9 (omh_partial_reduce): the partial column sums of a block's norm backwards (LayerNorm+modulate x 3, q|k and
    cross-q RMSNorm) issued as ONE launch at the end of the block instead of five: every gradient identical bit for bit
    under the deterministic mode (ordered reductions everywhere), both freeze settings, also into existing .grad tensors."""
    mt = importlib.import_module(PKG + ".wan.modules.model_train")
    ops = importlib.import_module(PKG + ".ops")
    ops.set_deterministic(True)
    try:
        for freeze in (True, False):
            cfg, sd, m, noise, vt, cl = _setup(wan_model_mod, freeze=freeze)
            args = dict(t=torch.ones(2, device="cuda") * 1000.0, context=[c.cuda() for c in cl], seq_len=24)

            def grads(defer):
                monkeypatch.setattr(mt, "_DEFER_COLSUM", defer)
                m.zero_grad(set_to_none=True)
                calls = []
                real = ops.partial_colsum_multi
                monkeypatch.setattr(ops, "partial_colsum_multi", lambda lst: calls.append(len(lst)) or real(lst))
                for scale in (1.0, 0.5):                                   # the second pass accumulates in place
                    out = m(list((noise * scale).cuda()), **args)
                    sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, vt.cuda())).backward()
                monkeypatch.setattr(ops, "partial_colsum_multi", real)
                torch.cuda.synchronize()
                return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, calls
            want, c0 = grads(False)
            got, c1 = grads(True)
            assert c0 == [] and len(c1) == 2 * len(m.blocks) and all(3 <= n <= 5 for n in c1), c1
            for n in want:
                assert torch.equal(got[n], want[n]), n
    finally:
        ops.set_deterministic(False)
