"""The hot path at BASELINE.json's FULL sizes (Wan2.1-T2V-1.3B, 81 frames 480x832: S = 32 760 tokens, latent
[16,21,60,104]) where the CPU oracle takes minutes per block: checked through properties that do not depend on the
size — sampled rows against the oracle's arithmetic, exact homogeneity / row-permutation equivariance of the GEMM,
softmax rows summing to one, key-permutation invariance, batch / padding invariance of the DiT forward, temporal
causality of the chunked VAE — plus the sampler update on the full latent against the oracle."""
import importlib
import os
import math

import pytest
import torch

from conftest import rel_rms, set_option

pytestmark = pytest.mark.gpu
PKG = "omnihuman-1-hack_amd"
S_FULL = 21 * 30 * 52                                  # 32 760


def _bf(x):
    return x.to(torch.bfloat16)


def _vt(v, B, Lk, H, D):
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros(B, H * D, Lp, dtype=torch.bfloat16, device="cuda")
    vt[:, :, :Lk] = v.reshape(B, Lk, H * D).transpose(1, 2)
    return vt


@pytest.mark.parametrize("kernel", ["w64"])
def test_self_attention_full_size(ops, kernel, monkeypatch):
    """One self-attention launch of the benchmark (12 heads x 32 760 x 32 760, D = 128; attention.py:96-127): the
    generated stream kernel the dispatch picks at this size ("w64"); pinned so that the query slice of property (4)
    runs the same kernel."""
    set_option("OMH_ATTN_KERNEL", kernel)
    torch.manual_seed(5)
    B, H, L, D = 1, 12, S_FULL, 128
    q = _bf(torch.randn(B, L, H, D, device="cuda"))
    k = _bf(torch.randn(B, L, H, D, device="cuda"))
    v = _bf(torch.randn(B, L, H, D, device="cuda"))
    out = ops.flash_attn(q, k, _vt(v, B, L, H, D), None)
    assert out.shape == (B, L, H, D) and bool(torch.isfinite(out.float()).all())
    # (1) sampled query rows (first / last tiles and random ones) of three heads against softmax(q k^T / sqrt(D)) v
    rows = torch.tensor([0, 1, 31, 255, 256, 16383, L - 257, L - 2, L - 1] +
                        torch.randint(0, L, (40,), generator=torch.Generator().manual_seed(1)).tolist(), device="cuda")
    for h in (0, 5, 11):
        s = (q[0, rows, h].float() @ k[0, :, h].float().t()) * D ** -0.5
        ref = torch.softmax(s, -1) @ v[0, :, h].float()
        got = out[0, rows, h].float()
        assert rel_rms(got, ref) < 8e-3 and float((got - ref).abs().max()) < 3e-2, h
    # (2) rows of P sum to one: with V = 1 the output is 1 for every query (bf16 rounding of P only)
    ones = torch.ones_like(v)
    o1 = ops.flash_attn(q, k, _vt(ones, B, L, H, D), None).float()
    assert float((o1 - 1.0).abs().max()) < 8e-3
    # (3) keys are a set: permuting (k, v) rows together changes only the summation order
    perm = torch.randperm(L, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    o2 = ops.flash_attn(q, k[:, perm].contiguous(), _vt(v[:, perm].contiguous(), B, L, H, D), None)
    assert rel_rms(o2.float(), out.float()) < 6e-3
    # (4) queries are independent: a launch on a slice of the queries returns the same rows, bit for bit
    o3 = ops.flash_attn(q[:, 4096:8192].contiguous(), k, _vt(v, B, L, H, D), None)
    assert torch.equal(o3, out[:, 4096:8192])


@pytest.mark.parametrize("N,K,epi", [(8960, 1536, "gelu"), (1536, 8960, "f32"), (3072, 1536, "f32")])
def test_gemm_full_size(ops, N, K, epi):
    """The DiT's linear layers at M = S = 32 760 (model.py:176-178,253-258): sampled rows against fp32 arithmetic on
    the same bf16 operands, and two exact properties of fp32-accumulated products."""
    torch.manual_seed(N + K)
    M = S_FULL
    a = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(N, device="cuda")
    rows = torch.tensor([0, 255, 256, M - 1] + torch.randint(0, M, (60,), generator=torch.Generator().manual_seed(3)).tolist(),
                        device="cuda")
    ref = a[rows].float() @ w.float().t() + bias
    if epi == "gelu":
        out = ops.gemm(a, w, bias=bias, epilogue=ops.EPI_GELU_BF16)
        assert rel_rms(out[rows].float(), torch.nn.functional.gelu(ref, approximate="tanh")) < 5e-3
        return
    out = ops.gemm(a, w, bias=bias, epilogue=ops.EPI_F32)
    assert rel_rms(out[rows], ref) < 2e-5
    # homogeneity: scaling A by a power of two scales every partial sum exactly
    out2 = ops.gemm(_bf(a.float() * 4.0), w, epilogue=ops.EPI_F32)
    plain = ops.gemm(a, w, epilogue=ops.EPI_F32)
    assert torch.equal(out2, plain * 4.0)
    # a row of C depends on its own row of A only, and on no other row's position in the tile
    perm = torch.randperm(M, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    assert torch.equal(ops.gemm(a[perm].contiguous(), w, epilogue=ops.EPI_F32), plain[perm])


@pytest.fixture(scope="module")
def wan_1_3b():
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    torch.manual_seed(1234)
    with torch.device("cuda"):
        m = model_mod.WanModel(**cfgs.dit_kwargs(cfgs.t2v_1_3B))
        torch.nn.init.xavier_uniform_(m.head.head.weight)      # zero-init in the reference (model.py:612)
    return m.eval().requires_grad_(False)


def test_dit_forward_full_size_invariances(wan_1_3b):
    """WanModel.forward on the benchmark's latent [16,21,60,104] (model.py:502-563): a sample's output depends
    neither on the batch it is in nor on the padded sequence length."""
    model = wan_1_3b
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.randn(16, 21, 60, 104, device="cuda", generator=g)
    x2 = torch.randn(16, 21, 60, 104, device="cuda", generator=g)
    ctx = torch.randn(120, 4096, device="cuda", generator=g)
    ctx2 = torch.randn(40, 4096, device="cuda", generator=g)
    t = torch.tensor([875.0], device="cuda")
    with torch.no_grad():
        one = model([x], t, [ctx], S_FULL)[0]
        assert one.shape == (16, 21, 60, 104) and one.dtype == torch.float32 and bool(torch.isfinite(one).all())
        assert float(one.abs().mean()) > 1e-3
        again = model([x], t, [ctx], S_FULL)[0]
        assert torch.equal(one, again)                                         # repeatable bit for bit
        pair = model([x, x2], torch.cat([t, t]), [ctx, ctx2], S_FULL)
        assert torch.equal(pair[0], one)                                       # batch of two = two batches of one
        other = model([x2], t, [ctx2], S_FULL)[0]
        assert torch.equal(pair[1], other)
        # seq_len > S: zero rows, masked keys.  Not bit-identical: M changes, so some GEMMs take another tile
        # configuration (another fp32 summation order, bf16 roundings flip and propagate through 30 layers).
        # Measured 2.4e-3 (round 2) between the two evaluations, each of which sits 5e-3 from the fp32 oracle.
        padded = model([x], t, [ctx], S_FULL + 520)[0]
        assert rel_rms(padded, one) < 4e-3
        assert rel_rms(other, one) > 0.1                                       # and the inputs do matter


def test_sampler_step_full_size_matches_oracle():
    """Fused CFG + UniPC update on the full latent against the oracle scheduler (fm_solvers_unipc.py:279-739)."""
    from oracle import sampler_oracle as SO
    sched_mod = importlib.import_module(PKG + ".wan.utils.fm_solvers_unipc")
    g = torch.Generator().manual_seed(3)
    shape = (16, 21, 60, 104)
    x0 = torch.randn(shape, generator=g)
    preds = [(torch.randn(shape, generator=g), torch.randn(shape, generator=g)) for _ in range(4)]
    it = iter(preds)
    ref = SO.sample_loop(lambda x, t: next(it), x0, 4, 5.0, 5.0, solver="unipc")
    s = sched_mod.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    s.set_timesteps(4, device="cuda", shift=5.0)
    s.set_begin_index(0)
    x = x0.cuda()
    for c, u in preds:
        x = s.step_cfg(c.cuda(), u.cuda(), 5.0, x)
    assert rel_rms(x, ref) < 1e-5


def test_vae_full_size_temporal_causality():
    """Chunked causal decode / encode at 480x832 (vae.py:516-568): frame chunks see only the past, so a prefix of
    the clip decodes (encodes) to the same leading frames as the whole clip — through the sliding-window history —
    bit for bit, with the default dispatch: the convolution kernel is chosen by the layer's geometry, not by the number
    of frames a call carries (round 2 needed OMH_CONV_TILE pinned here: four batched latent frames moved the
    latent-resolution layers to another tile family and summation order)."""
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    torch.manual_seed(4321)
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(6)
    z = torch.randn(16, 4, 60, 104, device="cuda", generator=g)
    full = vae.decode([z])[0]
    assert full.shape == (3, 13, 480, 832) and bool(torch.isfinite(full).all()) and float(full.abs().max()) <= 1.0
    head = vae.decode([z[:, :2].contiguous()])[0]
    assert head.shape == (3, 5, 480, 832)
    assert torch.equal(head, full[:, :5])
    video = full.clamp(-1, 1)
    mu = vae.encode([video])[0]
    assert mu.shape == (16, 4, 60, 104) and bool(torch.isfinite(mu).all())
    mu_head = vae.encode([video[:, :5].contiguous()])[0]
    assert torch.equal(mu_head, mu[:, :2])


def test_vae_full_area_parity_against_the_oracle():
    """The VAE at the benchmark's FULL area, 480 x 832, against the CPU oracle (fp32): decode of a two-frame latent (the
    'Rep' first chunk + one steady-state chunk through both temporal upsamples, vae.py:544-568) and encode of the
    five frames it yields (:516-542).  Measured (profiles/r03_vae_full_area_parity.json): 9.2e-3 / 2.6e-3 rel-RMS — bf16
    convolution operands under an fp32 trunk; the bounds are 2x that.  ~1 min of host time for the oracle.
    Round 4: the same clip through WanVAE(dtype=torch.float) — the reference's own arithmetic class (vae.py:619-624,
    649-663), here split-bf16 operand pairs — against the same oracle outputs: <= 2e-4; a prefix of the clip still
    decodes / encodes to the same leading frames bit for bit in that mode."""
    import json
    from oracle import wan_vae_oracle as V
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    torch.manual_seed(4321)
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda")
    vae32 = vae_mod.WanVAE(vae_pth=None, device="cuda")                      # the reference's default: dtype=torch.float
    assert vae32.dtype == torch.float32 and vae32.model.compute_dtype == torch.float32
    vae32.model.load_state_dict(vae.model.state_dict())
    sd = {k: v.detach().float().cpu() for k, v in vae.model.state_dict().items()}
    cfg = V.VAEConfig(dim=96)
    z = torch.randn(16, 2, 60, 104, generator=torch.Generator().manual_seed(5))
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ref = V.vae_decode(sd, cfg, z)
    out = vae.decode([z.cuda()])[0].float().cpu()
    assert out.shape == ref.shape == (3, 5, 480, 832)
    assert rel_rms(out, ref) < 2e-2
    video = ref.clamp(-1, 1)
    mu_ref = V.vae_encode(sd, cfg, video)
    mu = vae.encode([video.cuda()])[0].float().cpu()
    assert mu.shape == mu_ref.shape == (16, 2, 60, 104)
    assert rel_rms(mu, mu_ref) < 6e-3
    # ---- the fp32-faithful mode
    out32 = vae32.decode([z.cuda()])[0]
    mu32 = vae32.encode([video.cuda()])[0]
    e_dec, e_enc = rel_rms(out32, ref), rel_rms(mu32, mu_ref)
    rec = {"decode_rel_rms_fp32_mode": e_dec, "encode_rel_rms_fp32_mode": e_enc,
           "decode_max_abs_fp32_mode": float((out32.cpu() - ref).abs().max()),
           "decode_rel_rms_bf16_mode": rel_rms(out, ref), "encode_rel_rms_bf16_mode": rel_rms(mu, mu_ref),
           "what": "WanVAE decode of a [16,2,60,104] latent -> 5 frames 480x832 and encode of those frames, vs the fp32 oracle"}
    print(f"[measured] VAE 480x832 vs fp32 oracle: fp32 mode decode {e_dec:.3e} encode {e_enc:.3e}; "
          f"bf16 mode decode {rec['decode_rel_rms_bf16_mode']:.3e} encode {rec['encode_rel_rms_bf16_mode']:.3e}")
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "vae_full_area_parity_r04.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    assert e_dec < 2e-4 and e_enc < 2e-4, (e_dec, e_enc)
    head = vae32.decode([z[:, :1].contiguous().cuda()])[0]                 # prefix = whole clip, bit for bit
    assert torch.equal(head, out32[:, :1])
    mu_head = vae32.encode([video[:, :1].contiguous().cuda()])[0]
    assert torch.equal(mu_head, mu32[:, :1])


def test_training_step_full_size_properties():
    """BASELINE config 3 at its real size (Wan2.1-T2V-1.3B, two [16,1,60,104] clips, 512-token contexts, t = 1000;
    distilled_trainer.py:241-316): the backward is homogeneous in the loss scale (x4 is exact in every bf16 / fp32
    rounding, only the fp32 atomics' order varies), repeatable to that same noise, leaves the FFNs of blocks > 10
    without a gradient (the reference's quirk, model.py:317-324) and every other parameter with a finite one."""
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    trainer = importlib.import_module(PKG + ".trainer")
    torch.manual_seed(77)
    with torch.device("cuda"):
        m = model_mod.WanModel(**cfgs.dit_kwargs(cfgs.t2v_1_3B))
        torch.nn.init.xavier_uniform_(m.head.head.weight)
    m.train()
    g = torch.Generator(device="cuda").manual_seed(9)
    batch = (torch.randn(2, 16, 1, 60, 104, device="cuda", generator=g),
             torch.randn(2, 512, 4096, device="cuda", generator=g),
             torch.randn(2, 16, 1, 60, 104, device="cuda", generator=g))

    def grads(scale):
        for p in m.parameters():
            p.grad = None
        loss = trainer.forward_backward(batch, m, loss_scale=scale)
        return float(loss), {n: p.grad for n, p in m.named_parameters()}

    l1, g1 = grads(1.0)
    l1b, g1b = grads(1.0)
    l4, g4 = grads(4.0)
    assert l1 == l1b == l4 and math.isfinite(l1) and l1 > 0          # the forward is bit-repeatable; the loss is unscaled
    none = sorted(n for n, v in g1.items() if v is None)
    assert none and all(".ffn." in n and int(n.split(".")[1]) > 10 for n in none), none[:5]
    assert len(none) == 4 * 19                                        # blocks 11..29: two Linears with weight and bias
    probe = ["blocks.0.self_attn.q.weight", "blocks.5.ffn.0.weight", "blocks.10.ffn.2.bias", "blocks.17.cross_attn.k.weight",
             "blocks.29.self_attn.o.bias", "blocks.29.modulation", "head.head.weight", "patch_embedding.weight",
             "text_embedding.0.weight", "time_embedding.2.bias", "blocks.3.self_attn.norm_q.weight"]
    for n in probe:
        a, b, c = g1[n].float(), g1b[n].float(), g4[n].float()
        assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0, n
        assert rel_rms(b, a) < 1e-4, n                                # run to run: atomics order only
        assert rel_rms(c, 4.0 * a) < 1e-4, n                          # homogeneity
    assert all(bool(torch.isfinite(v).all()) for v in g1.values() if v is not None)


# Config 3 at its REAL size against the autograd oracle (VERDICT round 3, next #1a).  Bounds = 2 x the figures measured
# on MI355X in round 4 (profiles/r04_full_size_gradient_parity.json): bf16 operands, fp32 accumulation, 30 blocks of
# back-propagation at d = 1536.
# Round 4, first run (gpurun_out/full_size_gradient_parity_freeze*.json): loss 1.1e-4, output 3.4e-3, matrices <= 9.7e-3
# (all 30 blocks), 1-D parameters <= 9.7e-3 on the probes; the worst 1-D figure over ALL parameters, 5.1e-2, is the
# cross-attention K bias — a gradient that is exactly zero in exact arithmetic (adding one vector to every key shifts
# all scores of a query row alike; softmax does not see it), so what is measured there is rounding noise against a
# floor, and it gets its own absolute-scale bound.
TOL_FULL_LOSS = 1e-3
TOL_FULL_GRAD = 2e-2
TOL_FULL_GRAD_1D = 2.2e-2       # all 1-D parameters: 1.09e-2 measured (blocks.11.cross_attn.norm_k.weight)
TOL_FULL_GRAD_NULL = 1.1e-1     # 5.1e-2 measured


def _null_gradient(name):
    """Parameters whose gradient all but vanishes: the K biases of the cross-attention.  The column sums of dK are zero
    identically (softmax shift invariance, no RoPE); what is left comes through the key RMSNorm's per-token scale
    (model.py:176-178) and is small: compared against the floor."""
    return name.endswith("cross_attn.k.bias") or name.endswith("cross_attn.k_img.bias")
FULL_SIZE_PROBES = [
    "blocks.0.self_attn.q.weight", "blocks.0.self_attn.q.bias", "blocks.0.self_attn.norm_q.weight", "blocks.0.modulation",
    "blocks.0.ffn.0.weight", "blocks.0.ffn.2.bias", "blocks.0.cross_attn.k.weight", "blocks.0.cross_attn.o.bias",
    "blocks.0.norm3.weight",
    "blocks.10.ffn.0.weight", "blocks.10.ffn.0.bias", "blocks.10.self_attn.v.weight", "blocks.10.self_attn.norm_k.weight",
    "blocks.10.modulation", "blocks.10.cross_attn.norm_q.weight",
    "blocks.11.ffn.2.weight", "blocks.11.ffn.2.bias", "blocks.11.self_attn.o.weight", "blocks.11.self_attn.k.bias",
    "blocks.11.modulation", "blocks.11.cross_attn.q.weight",
    "blocks.29.self_attn.q.weight", "blocks.29.self_attn.o.bias", "blocks.29.ffn.0.weight", "blocks.29.modulation",
    "blocks.29.self_attn.norm_q.weight", "blocks.29.cross_attn.v.weight", "blocks.29.cross_attn.v.bias",
    "patch_embedding.weight", "patch_embedding.bias", "head.head.weight", "head.head.bias", "head.modulation",
    "text_embedding.0.weight", "text_embedding.2.bias", "time_embedding.0.weight", "time_embedding.2.bias",
    "time_projection.1.weight", "time_projection.1.bias",
]


@pytest.mark.parametrize("freeze", [True, False])
def test_training_step_full_size_gradients_against_the_autograd_oracle(freeze):
    """distilled_trainer.py:268-301 on the 1.3B model: ONE [16,1,60,104] clip, 512-token context, t = 1000 —
    loss and probe gradients (first / quirk-boundary / last blocks: matrices, biases, norm gains, modulation; the
    embeddings and the head) against ``oracle.wan_dit_oracle.dit_forward_autograd`` in fp32 on the host, both settings of
    the reference's FFN-freeze quirk (model.py:317-324).  Figures are written to gpurun_out/ for profiles/."""
    import json
    import time
    from oracle import wan_dit_oracle as O
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    torch.manual_seed(177)
    with torch.device("cuda"):
        m = model_mod.WanModel(**cfgs.dit_kwargs(cfgs.t2v_1_3B))
        torch.nn.init.xavier_uniform_(m.head.head.weight)
        with torch.no_grad():
            for p in m.parameters():                                  # zero-initialised biases / gains: make them carry signal
                if p.dim() == 1 and float(p.abs().sum()) == 0:
                    p.uniform_(-0.05, 0.05)
    m.train()
    m.reference_ffn_freeze = freeze
    g = torch.Generator().manual_seed(19)
    noise = torch.randn(1, 16, 1, 60, 104, generator=g)
    ctx = torch.randn(1, 512, 4096, generator=g)
    vt = torch.randn(1, 16, 1, 60, 104, generator=g)
    # ---- the oracle, fp32 autograd on the host
    ocfg = O.DiTConfig.wan_t2v_1_3b()
    osd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    old_threads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    t0 = time.time()
    oo = O.dit_forward_autograd(osd, ocfg, [noise[0]], torch.ones(1) * 1000.0, [ctx[0]], 1560, reference_ffn_freeze=freeze)
    lo = torch.nn.functional.mse_loss(oo[0], vt[0])
    lo.backward()
    t_oracle = time.time() - t0
    torch.set_num_threads(old_threads)
    # ---- the product
    out = m(noise.cuda(), t=torch.ones(1, device="cuda") * 1000.0, context=[ctx[0].cuda()], seq_len=1560)
    lg = torch.nn.functional.mse_loss(out[0], vt[0].cuda())
    lg.backward()
    params = dict(m.named_parameters())
    rec = {"freeze": freeze, "oracle_seconds": round(t_oracle, 1), "loss_oracle": float(lo.detach()), "loss_hip": float(lg.detach()),
           "loss_rel_err": abs(float(lg) - float(lo)) / float(lo), "output_rel_rms": rel_rms(out[0], oo[0].detach()),
           "probes": {}}
    norms = sorted(float(v.grad.norm()) for v in osd.values() if v.grad is not None and float(v.grad.abs().max()) > 0)
    floor = 1e-2 * norms[len(norms) // 2]
    worst = {1: 0.0, 2: 0.0}
    bad = []
    for name in FULL_SIZE_PROBES:
        og, p = osd[name].grad, params[name]
        frozen = freeze and ".ffn." in name and int(name.split(".")[1]) > 10
        if frozen:
            assert p.grad is None and (og is None or float(og.abs().max()) == 0.0), name
            rec["probes"][name] = None
            continue
        assert p.grad is not None and og is not None, name
        err = float((p.grad.double().cpu() - og.double()).norm() / max(float(og.double().norm()), floor))
        kind = 2 if og.dim() > 1 and min(og.shape) > 1 and "modulation" not in name else 1
        rec["probes"][name] = {"rel_rms": err, "oracle_norm": float(og.norm()), "kind": "matrix" if kind == 2 else "1-D"}
        worst[kind] = max(worst[kind], err)
        if err > (TOL_FULL_GRAD if kind == 2 else TOL_FULL_GRAD_1D):
            bad.append((name, err))
    # every parameter, not only the probes: worst figures per kind (cheap: the gradients are there)
    allw = {0: (0.0, None), 1: (0.0, None), 2: (0.0, None)}          # 0: identically-null gradients (noise vs the floor)
    for name, p in params.items():
        og = osd[name].grad
        if og is None or float(og.abs().max()) == 0.0 or p.grad is None:
            continue
        err = float((p.grad.double().cpu() - og.double()).norm() / max(float(og.double().norm()), floor))
        kind = 0 if _null_gradient(name) else (2 if og.dim() > 1 and min(og.shape) > 1 and "modulation" not in name else 1)
        if err > allw[kind][0]:
            allw[kind] = (err, name)
    rec.update(worst_probe_matrix=worst[2], worst_probe_1d=worst[1], worst_all_matrix=allw[2], worst_all_1d=allw[1],
               worst_identically_null=allw[0])
    print(f"[measured] 1.3B training step vs autograd oracle (freeze={freeze}): loss {float(lg):.6f} vs {float(lo):.6f}, "
          f"output {rec['output_rel_rms']:.3e}, probes: worst matrix {worst[2]:.3e}, worst 1-D {worst[1]:.3e}; "
          f"all parameters: matrix {allw[2]}, 1-D {allw[1]}, identically-null gradients {allw[0]}; oracle {t_oracle:.0f} s")
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, f"full_size_gradient_parity_freeze{int(freeze)}.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    assert rec["loss_rel_err"] < TOL_FULL_LOSS, rec["loss_rel_err"]
    assert not bad, bad
    assert allw[2][0] < TOL_FULL_GRAD and allw[1][0] < TOL_FULL_GRAD_1D and allw[0][0] < TOL_FULL_GRAD_NULL, allw


def test_omnihuman_full_size_sampling(wan_1_3b):
    """BASELINE config 4 at its real size: OmniHumanWanT2V on the 1.3B backbone, 49 frames 480x832 (13 latent frames
    + the reference latent frame concatenated along T: S = 21 840), wav2vec-sized audio and 308-keypoint pose heat
    maps (omnihuman_wan_t2v.py:313-438).  Two annealed-CFG steps: finite, and the condition tokens matter."""
    omni = importlib.import_module(PKG + ".omnihuman_wan_t2v")
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    torch.manual_seed(4321)
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda")

    class T2V:
        model, text_encoder = wan_1_3b, None

    T2V.vae = vae
    m = omni.OmniHumanWanT2V(dict(num_frames=49, num_keypoints=308, model_dim=1536, audio_dim=1024), device_id=0, wan_t2v=T2V)
    g = torch.Generator(device="cuda").manual_seed(2)
    audio = torch.randn(1, 49, 1024, device="cuda", generator=g)
    pose = torch.rand(1, 308, 49, 64, 64, device="cuda", generator=g)
    ref_img = torch.rand(3, 1, 480, 832, device="cuda", generator=g) * 2 - 1
    kw = dict(reference_image=ref_img, num_inference_steps=2, cfg_scale=7.5, return_latent=True,
              text_context=torch.randn(120, 4096, device="cuda", generator=g),
              text_context_null=torch.randn(40, 4096, device="cuda", generator=g),
              noise=torch.randn(16, 13, 60, 104, device="cuda", generator=g))
    cond = m.prepare_conditions(audio=audio, pose=pose, reference_image=ref_img, text_context=kw["text_context"])
    assert tuple(cond["tokens"].shape) == (1, 2 * 48 + 49, 1536) and tuple(cond["reference"].shape) == (16, 1, 60, 104)
    lat = m(audio=audio, pose=pose, **kw)
    assert tuple(lat.shape) == (16, 13, 60, 104) and bool(torch.isfinite(lat).all())
    assert torch.equal(lat, m(audio=audio, pose=pose, **kw))                    # repeatable bit for bit
    plain = m(**kw)
    assert rel_rms(lat, plain) > 1e-3                                           # audio / pose tokens reach the sample
    video = vae.decode([lat])[0]
    assert tuple(video.shape) == (3, 49, 480, 832) and bool(torch.isfinite(video).all())


def test_wan_t2v_generate_full_size(wan_1_3b):
    """BASELINE config 2 end to end at its real size, shortened to two sampling steps: WanT2V.generate on the 1.3B
    backbone, 81 frames 480x832 (text2video.py:112-269) — the batch-2 CFG forward equals two forwards bit for bit,
    the seed decides the sample, the decoded video is finite and in range."""
    cfgs = importlib.import_module(PKG + ".wan.configs")
    t2v = importlib.import_module(PKG + ".wan.text2video")
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    torch.manual_seed(4321)
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda")
    pipe = t2v.WanT2V(cfgs.t2v_1_3B, checkpoint_dir="", model=wan_1_3b, vae=vae)
    g = torch.Generator().manual_seed(5)
    kw = dict(size=(832, 480), frame_num=81, shift=5.0, sampling_steps=2, guide_scale=5.0,
              context=[torch.randn(120, 4096, generator=g)], context_null=[torch.randn(40, 4096, generator=g)])
    for solver in ("unipc", "dpm++"):
        lat = pipe.generate("", seed=3, sample_solver=solver, return_latent=True, **kw)
        assert tuple(lat.shape) == (16, 21, 60, 104) and bool(torch.isfinite(lat).all())
        assert torch.equal(lat, pipe.generate("", seed=3, sample_solver=solver, return_latent=True, batched_cfg=False, **kw))
    other = pipe.generate("", seed=4, sample_solver="dpm++", return_latent=True, **kw)
    assert rel_rms(other, lat) > 0.1
    video = pipe.generate("", seed=3, **kw)
    assert tuple(video.shape) == (3, 81, 480, 832) and bool(torch.isfinite(video).all()) and float(video.abs().max()) <= 1.0


def test_self_attention_split_kv_tail_config4(ops, monkeypatch):
    """BASELINE config 4's self-attention (S = 21 840, 12 heads): 1 032 tiles of 256 queries = 4 x 256 CUs + 8, so the
    last 8 tiles are split over the keys into 16 workers each (fp32 partial results + log-sum-exp, combined by a second
    small kernel: attention_w64.hip).  The tail rows against fp32 arithmetic, the log-sum-exp of split and unsplit
    rows, and everything outside the tail bit for bit against the unsplit launch."""
    torch.manual_seed(9)
    B, H, L, D = 1, 12, 21 * 2 * 30 * 52 // 3, 128                      # 21 840 = (13 + 1) latent frames x 1 560
    assert L == 21840
    q = _bf(torch.randn(B, L, H, D, device="cuda"))
    k = _bf(torch.randn(B, L, H, D, device="cuda"))
    v = _bf(torch.randn(B, L, H, D, device="cuda"))
    vt = _vt(v, B, L, H, D)

    def run(split):
        set_option("OMH_W64_SPLIT", split)
        out = torch.empty(B, L, H, D, dtype=torch.bfloat16, device="cuda")
        lse = torch.empty(B, H, L, dtype=torch.float32, device="cuda")
        ops.flash_attn_raw(ops.ptr(q), ops.ptr(k), ops.ptr(vt), ops.ptr(out), None, B, H, L, L, q.stride(0), q.stride(1),
                           k.stride(0), k.stride(1), vt.stride(0), out.stride(0), out.stride(1), vt.stride(1), D ** -0.5,
                           lse=ops.ptr(lse))
        return out, lse
    out, lse = run("1")
    ref_o, ref_l = run("0")
    assert torch.equal(out[:, :, :11], ref_o[:, :, :11]) and torch.equal(out[:, :19968, 11], ref_o[:, :19968, 11])
    assert not torch.equal(out[:, 19968:, 11], ref_o[:, 19968:, 11])      # the tail did go through the split path
    rows = torch.tensor([19968, 19969, 20223, 20224, 21000, L - 257, L - 2, L - 1], device="cuda")
    s = (q[0, rows, 11].float() @ k[0, :, 11].float().t()) * D ** -0.5
    ref = torch.softmax(s, -1) @ v[0, :, 11].float()
    got = out[0, rows, 11].float()
    assert rel_rms(got, ref) < 8e-3 and float((got - ref).abs().max()) < 3e-2
    assert float((lse[0, 11, rows] - torch.logsumexp(s, -1)).abs().max()) < 5e-3
    assert float((lse - ref_l).abs().max()) < 5e-3
    assert torch.equal(run("1")[0], out)                                   # repeatable


@pytest.mark.parametrize("C,T,H,W", [(96, 4, 480, 832), (192, 4, 240, 416), (384, 5, 120, 208)])
def test_vae_convolutions_full_size_stream_kernel_equals_8_wave_kernel(ops, C, T, H, W, monkeypatch):
    """The decoder's residual-block convolutions at their real sizes (3 132 / 1 573 / 983 x 2 tiles): the stream kernel
    (conv_w64.hip) returns the 8-wave kw-shared kernel's values bit for bit, fp32 trunk and bf16 outputs alike."""
    torch.manual_seed(C)
    x = _bf(torch.randn(2 + T, H, W, C, device="cuda"))
    wp = _bf(torch.randn(C, 27 * C, device="cuda") / (27 * C) ** 0.5)
    bias = torch.randn(C, device="cuda")
    rf = torch.randn(T, H, W, C, device="cuda")

    def run(tile):
        set_option("OMH_CONV_TILE", tile)
        return (ops.conv_cl(x, wp, bias, T, H, W, C, 3, 3, 3, pad_h=1, pad_w=1),
                ops.conv_cl(x, wp, bias, T, H, W, C, 3, 3, 3, pad_h=1, pad_w=1, resid=rf, out_f32=True))

    got, ref = run("w64"), run("wide")
    for g, r in zip(got, ref):
        assert bool(torch.isfinite(g.float()).all()) and torch.equal(g, r)


def test_vae_81_frames_against_oracle_quarter_area():
    """The whole 81-frame clip of BASELINE config 2 through the VAE against the fp32 oracle — at a quarter of the
    area (latent [16,21,30,52] <-> 81 frames 240x416; the arithmetic per voxel is independent of H x W, the fp32
    oracle needs ~2 minutes of host time at this size and ~8 at 480x832): all 21 chunks of the decoder with their
    causal caches, the 'Rep' first chunk, every temporal up / down-sampling, then the encoder on the decoded clip.
    Prints the measured figures (bench.py only carries a 5-frame sample)."""
    from oracle import wan_vae_oracle as V
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    torch.manual_seed(4321)
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda")
    sd = {k: v.detach().float().cpu() for k, v in vae.model.state_dict().items()}
    cfg = V.VAEConfig(dim=96)
    z = torch.randn(16, 21, 30, 52, generator=torch.Generator().manual_seed(5))
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with torch.no_grad():
        ref = V.vae_decode(sd, cfg, z)
    out = vae.decode([z.cuda()])[0].float().cpu()
    assert out.shape == ref.shape == (3, 81, 240, 416)
    e_dec = rel_rms(out, ref)
    e_last = rel_rms(out[:, -4:], ref[:, -4:])                      # no drift along the 21 chunks
    video = ref.clamp(-1, 1)
    with torch.no_grad():
        ref_mu = V.vae_encode(sd, cfg, video)
    mu = vae.encode([video.cuda()])[0].float().cpu()
    assert mu.shape == ref_mu.shape == (16, 21, 30, 52)
    e_enc = rel_rms(mu, ref_mu)
    print(f"[measured] VAE 81 frames 240x416: decode rel-RMS {e_dec:.3e} (last chunk {e_last:.3e}), encode rel-RMS {e_enc:.3e}")
    # measured on MI355X (round 2, fp32 residual trunk): decode 1.03e-2 (last chunk 1.03e-2: no drift), encode 3.4e-3
    # (bf16 trunk of round 1: 1.25e-2 / 3.8e-3); bounds = 2 x
    assert e_dec < 2.1e-2 and e_last < 2.1e-2 and e_enc < 7e-3


@pytest.mark.parametrize("clips", [1, 4])
def test_gradient_accumulation_full_size_in_place_equals_autograd(clips):
    """The reference trainer's accumulation cycle (distilled_trainer.py:116-134,289) on the 1.3B model at its real shapes:
    three micro-steps of ``clips`` [16,1,60,104] clips, the block backward adding into the existing .grad tensors (the
    k-major stream's accumulate epilogue on 1536 x 1536 ... 8960 x 1536 products over 1 560 / 6 240 rows, ragged tiles
    included) against autograd's own accumulation — every matrix of the blocks bit for bit, 1-D sums to fp32 rounding."""
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    cfgs = importlib.import_module(PKG + ".wan.configs")
    torch.manual_seed(91)
    with torch.device("cuda"):
        m = model_mod.WanModel(**cfgs.dit_kwargs(cfgs.t2v_1_3B))
        torch.nn.init.xavier_uniform_(m.head.head.weight)
    m.train()
    m.reference_ffn_freeze = False
    g = torch.Generator(device="cuda").manual_seed(23)
    noise = torch.randn(3, clips, 16, 1, 60, 104, device="cuda", generator=g)
    ctx = torch.randn(clips, 512, 4096, device="cuda", generator=g)
    vt = torch.randn(clips, 16, 1, 60, 104, device="cuda", generator=g)

    def cycle(direct):
        m.direct_grad_accumulation = direct
        m.zero_grad(set_to_none=True)
        for k in range(3):
            out = m(noise[k], t=torch.ones(clips, device="cuda") * 1000.0, context=[c for c in ctx], seq_len=1560)
            (torch.nn.functional.mse_loss(torch.stack(out), vt) / 3).backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    want, got = cycle(False), cycle(True)
    del m.direct_grad_accumulation
    assert set(want) == set(got) and len(want) == len(list(m.parameters()))
    worst = 0.0
    for n in want:
        assert bool(torch.isfinite(got[n]).all()), n
        if want[n].dim() == 2 and n.startswith("blocks."):          # (modulation [1, 6, dim] is a column sum like the biases)
            assert torch.equal(got[n], want[n]), n
        else:
            worst = max(worst, rel_rms(got[n], want[n]))
    print(f"[measured] accumulation cycle at 1.3B, {clips} clip(s): block matrices bit-identical to autograd's accumulation, "
          f"worst 1-D / embedding / head difference {worst:.2e}")
    assert worst < 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# The headline configuration end to end against the REFERENCE ("weak" 1 / item 4): golden vectors of
# oracle/make_golden_full_size.py — the real WanModel.forward (model.py:502-563) at S = 32 760 with all 30 layers, and
# 50-step CFG trajectories driven as text2video.py:206-252 drives them, on detgen weights / inputs regenerated here.
# Bounds = 2 x the figures measured on MI355X (profiles/r05_full_forward_parity.json).
TOL_HEADLINE_FORWARD = 9.0e-3        # measured 4.06e-3 (lattice) / 4.44e-3 (probes) on MI355X, round 5: 30 layers, S = 32 760
TOL_TRAJECTORY = {1: 4.0e-4, 10: 3.0e-3, 25: 8.0e-3, 50: 1.2e-2}       # relative RMS of the latent after k steps; measured
#   1.6e-4 / 1.5e-3 / 3.9e-3 / 5.8e-3 on MI355X with either solver (profiles/r05_trajectory_parity_*.json)


@pytest.fixture(scope="module")
def wan_1_3b_detgen():
    """Wan2.1-T2V-1.3B with the detgen weight set of the golden vectors ("wan1.3b": 1.42 G values, ~1 min to generate)."""
    from oracle import wan_dit_oracle as O, make_golden_full_size as F
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    cfg = O.DiTConfig.wan_t2v_1_3b()
    sd = O.synth_state_dict(cfg, F.TAG)
    m = model_mod.WanModel(**{k: getattr(cfg, k) for k in ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim",
                                                            "freq_dim", "text_dim", "out_dim", "num_heads", "num_layers",
                                                            "qk_norm", "cross_attn_norm", "eps")})
    m.load_state_dict(sd)
    del sd
    return m.cuda().eval().requires_grad_(False)


def _golden(name):
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated (oracle/make_golden_full_size.py)")
    return np.load(path)


def test_headline_forward_all_30_layers_against_the_reference(wan_1_3b_detgen):
    """One whole DiT forward of BASELINE config 2 — latent [16,21,60,104], S = 32 760, 30 layers, the generated
    long-sequence attention stream, the fused q|k|v projection, every GEMM stream — against the reference's own output
    on the same weights and inputs (1/8 lattice of the output, 4 096 probes, the mean).  Writes the measured figures to
    gpurun_out/ for profiles/r05_full_forward_parity.json."""
    import json
    import numpy as np
    from oracle import make_golden_full_size as F
    g = _golden("dit_wan1_3b_c2_full_forward.npz")
    noise, t, ctx, seq_len = F.case_forward()
    m = wan_1_3b_detgen
    with torch.no_grad():
        out = m([noise.cuda()], t.cuda(), [ctx.cuda()], seq_len)[0].float().cpu()
    assert out.shape == (16, 21, 60, 104) and bool(torch.isfinite(out).all())
    lat = torch.from_numpy(g["lattice"])
    e_lat = rel_rms(out[F.LATTICE], lat)
    e_probe = rel_rms(out.flatten()[torch.from_numpy(g["probe_idx"])], torch.from_numpy(g["probe"]))
    max_abs = float((out[F.LATTICE] - lat).abs().max())
    rec = {"config": "Wan2.1-T2V-1.3B, latent [16,21,60,104], S = 32760, 30 layers, t = 625, 77 context tokens",
           "against": "the reference WanModel.forward (oracle/make_golden_full_size.py, tests/golden/dit_wan1_3b_c2_full_forward.npz)",
           "rel_rms_lattice_1_8": e_lat, "rel_rms_4096_probes": e_probe, "max_abs_lattice": max_abs,
           "reference_rms": float(g["rms"]), "mean": float(out.double().mean()), "reference_mean": float(g["mean"]),
           "oracle_vs_reference_rel_rms": float(g["oracle_vs_reference_rel_rms"]), "bound": TOL_HEADLINE_FORWARD}
    print("[measured] headline forward:", json.dumps(rec))
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                           "r05_full_forward_parity.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    assert e_lat < TOL_HEADLINE_FORWARD and e_probe < TOL_HEADLINE_FORWARD
    assert abs(float(out.double().mean()) - float(g["mean"])) < 3e-3 * float(g["abs_mean"])
    assert abs(float(out.double().abs().mean()) - float(g["abs_mean"])) < 3e-3 * float(g["abs_mean"])


@pytest.mark.parametrize("solver", ["unipc", "dpm++"])
def test_50_step_cfg_trajectory_against_the_reference(wan_1_3b_detgen, solver):
    """50 sampling steps with classifier-free guidance on a reduced clip ([16,5,30,52], S = 1 950, all 30 layers), the
    sampling loop of WanT2V.generate (text2video.py:206-252: two forwards + the fused CFG / solver step), against the
    latents the reference's WanModel + FlowUniPCMultistepScheduler / FlowDPMSolverMultistepScheduler produce after
    steps 1, 10, 25 and 50: the drift of the bf16 path over a whole trajectory, both solvers."""
    import json
    from oracle import make_golden_full_size as F
    unipc = importlib.import_module(PKG + ".wan.utils.fm_solvers_unipc")
    dpm = importlib.import_module(PKG + ".wan.utils.fm_solvers")
    g = _golden("wan1_3b_trajectory_50steps.npz")
    noise, ctx, ctx_null, seq_len, steps, shift, guide = F.case_trajectory()
    m = wan_1_3b_detgen
    dev = torch.device("cuda")
    if solver == "unipc":
        sch = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        sch.set_timesteps(steps, device=dev, shift=shift)
        timesteps = sch.timesteps
    else:
        sch = dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        timesteps, _ = dpm.retrieve_timesteps(sch, device=dev, sigmas=dpm.get_sampling_sigmas(steps, shift))
    key = "unipc" if solver == "unipc" else "dpmpp"
    assert torch.allclose(timesteps.float().cpu(), torch.from_numpy(g[key + "_timesteps"]).float())
    sch.set_begin_index(0)
    keep = {int(k): i for i, k in enumerate(g["keep_steps"])}
    with torch.no_grad():
        c_state, u_state = m.encode_context([ctx.cuda()]), m.encode_context([ctx_null.cuda()])
        x = noise.cuda()
        errs = {}
        for k, t in enumerate(timesteps):
            ts = torch.stack([t])
            cond = m([x], t=ts, context=c_state, seq_len=seq_len)[0]
            uncond = m([x], t=ts, context=u_state, seq_len=seq_len)[0]
            x = sch.step_cfg(cond, uncond, guide, x)
            if k + 1 in keep:
                errs[k + 1] = rel_rms(x, torch.from_numpy(g[key][keep[k + 1]]))
    print(f"[measured] 50-step trajectory ({solver}):", json.dumps(errs))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", f"r05_trajectory_parity_{key}.json"), "w") as fh:
        json.dump({"solver": solver, "clip": [16, 5, 30, 52], "steps": steps, "guide_scale": guide, "shift": shift,
                   "rel_rms_after_step": errs, "bounds": TOL_TRAJECTORY}, fh, indent=1)
    assert bool(torch.isfinite(x).all())
    for k, e in errs.items():
        assert e < TOL_TRAJECTORY[k], (k, e)


# Bounds of the long-sequence trajectory: 2 x the figures measured on MI355X (profiles/r06_trajectory_long_parity.json)
# measured 7.5e-4 / 3.5e-3 / 4.9e-3 (a 10-step schedule takes five times the 50-step schedule's stride per step)
TOL_TRAJECTORY_LONG = {1: 1.5e-3, 5: 7.0e-3, 10: 1.0e-2}


def test_10_step_cfg_trajectory_past_the_long_sequence_threshold(wan_1_3b_detgen):
    """VERDICT round 5 "weak" 1: the kernels the HEADLINE times — the generated long-sequence attention stream, the fused
    q | k | v projection with the transposed V, the 256 x 384 GEMM streams at M = S — pinned over MORE than one step: ten
    UniPC sampling steps with classifier-free guidance on a [16,6,60,104] clip (S = 9 360 >= the 8 192 threshold of
    model.py's fused route and of the attention dispatch), all 30 layers, against the latents the reference's WanModel +
    FlowUniPCMultistepScheduler produce after steps 1, 5 and 10 (oracle/make_golden_full_size.py trajectory_long: the
    imported reference, 39 min of CPU time; 1/8 lattice of each latent + its sums)."""
    import json
    from oracle import make_golden_full_size as F
    unipc = importlib.import_module(PKG + ".wan.utils.fm_solvers_unipc")
    ops = importlib.import_module(PKG + ".ops")
    g = _golden("wan1_3b_trajectory_long_10steps.npz")
    noise, ctx, ctx_null, seq_len, steps, shift, guide = F.case_trajectory_long()
    assert seq_len == 9360 and seq_len >= 8192 and seq_len % 8 == 0
    m = wan_1_3b_detgen
    dev = torch.device("cuda")
    sch = unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    sch.set_timesteps(steps, device=dev, shift=shift)
    timesteps = sch.timesteps
    assert torch.allclose(timesteps.float().cpu(), torch.from_numpy(g["unipc_timesteps"]).float())
    sch.set_begin_index(0)
    keep = {int(k): i for i, k in enumerate(g["keep_steps"])}
    # the long-sequence kernels really are in the loop: count the launches that take them
    seen = {"attn_long": 0, "qkv_fused": 0}
    raw_attn, raw_gemm = ops.flash_attn_raw, ops.gemm_raw

    def attn(*a, **k):
        if a[7] == a[8] == seq_len:
            seen["attn_long"] += 1
        return raw_attn(*a, **k)

    def gemm(*a, **k):
        if a[9] == ops.EPI_BF16_SPLIT_T:
            seen["qkv_fused"] += 1
        return raw_gemm(*a, **k)
    try:
        ops.flash_attn_raw, ops.gemm_raw = attn, gemm            # (the blocks call ops.* through the module: bench.py's timers do this too)
        with torch.no_grad():
            c_state, u_state = m.encode_context([ctx.cuda()]), m.encode_context([ctx_null.cuda()])
            x = noise.cuda()
            errs, sums = {}, {}
            for k, t in enumerate(timesteps):
                ts = torch.stack([t])
                cond = m([x], t=ts, context=c_state, seq_len=seq_len)[0]
                uncond = m([x], t=ts, context=u_state, seq_len=seq_len)[0]
                x = sch.step_cfg(cond, uncond, guide, x)
                if k + 1 in keep:
                    i = keep[k + 1]
                    errs[k + 1] = rel_rms(x.cpu()[F.LATTICE], torch.from_numpy(g["unipc_lattice"][i]))
                    sums[k + 1] = abs(float(x.double().norm()) - float(g["unipc_sums"][i][2])) / float(g["unipc_sums"][i][2])
    finally:
        ops.flash_attn_raw, ops.gemm_raw = raw_attn, raw_gemm
    assert seen["attn_long"] == 30 * 2 * steps and seen["qkv_fused"] == 30 * 2 * steps, seen
    print("[measured] 10-step trajectory at S = 9360:", json.dumps(errs), "norm:", json.dumps(sums))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "r06_trajectory_long_parity.json"), "w") as fh:
        json.dump({"solver": "unipc", "clip": [16, 6, 60, 104], "seq_len": seq_len, "steps": steps, "guide_scale": guide,
                   "shift": shift, "rel_rms_lattice_after_step": errs, "norm_rel_err_after_step": sums,
                   "bounds": TOL_TRAJECTORY_LONG, "long_attention_launches": seen["attn_long"],
                   "fused_qkv_launches": seen["qkv_fused"],
                   "against": "reference WanModel + FlowUniPCMultistepScheduler (tests/golden/wan1_3b_trajectory_long_10steps.npz)"},
                  fh, indent=1)
    assert bool(torch.isfinite(x).all())
    for k, e in errs.items():
        assert e < TOL_TRAJECTORY_LONG[k], (k, e)
