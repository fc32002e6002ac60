"""CPU: the C-ABI library loads and exports every symbol include/omh.h declares
(no compute without a GPU), and the host-side logic of the product package."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "omnihuman-1-hack_amd"


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "omh.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(omh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(omh):
    lib = ctypes.CDLL(os.path.join(ROOT, PKG, "lib", "libomh.so"))
    decl = _declared_symbols()
    assert len(decl) >= 17
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/omh.h but not exported by libomh.so"
    binding = importlib.import_module(PKG + "._lib")
    assert sorted(binding.EXPORTED) == decl, "ctypes signatures out of sync with the header"
    assert binding.lib.omh_abi_version() == 12 and binding.lib.omh_build_arch() == b"gfx950"


def test_argument_validation_without_gpu(omh):
    """Entry points reject bad arguments before touching the device."""
    binding = importlib.import_module(PKG + "._lib")
    lib = binding.lib
    assert lib.omh_gemm_bf16(None, None) == -1
    a = binding.GemmArgs()
    assert lib.omh_gemm_bf16(ctypes.byref(a), None) == -1                       # null pointers
    a.A, a.B, a.C, a.M, a.N, a.K, a.batch, a.lda, a.ldb, a.ldc = 16, 16, 16, 4, 4, 12, 1, 16, 16, 4
    assert lib.omh_gemm_bf16(ctypes.byref(a), None) == -2                       # K % 8 != 0
    assert lib.omh_flash_attn_fwd_d128(None, None) == -1
    assert lib.omh_conv_cl_bf16(None, None) == -1
    assert lib.omh_layernorm_modulate(16, 16, 4, 6, 1e-6, 1.0, None, None, 0, None, None, 0, 4, None) == -3


def test_no_cpu_fallback(omh, ops, wan_model_mod):
    """The product path fails loudly on CPU tensors instead of silently computing elsewhere."""
    with pytest.raises(ops.OmhError):
        ops.cast_bf16(torch.zeros(8))
    m = wan_model_mod.WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=1, text_dim=64, text_len=8, freq_dim=64)
    with pytest.raises(ops.OmhError):
        m([torch.zeros(16, 1, 4, 4)], torch.tensor([1.]), [torch.zeros(3, 64)], 4)
    # the product package never imports the oracle
    import sys
    for name, mod in list(sys.modules.items()):
        if name.startswith(PKG):
            src = getattr(mod, "__file__", None)
            if src and src.endswith(".py"):
                assert "oracle" not in re.sub(r'""".*?"""', "", open(src).read(), flags=re.S).replace("# oracle", "")


def test_oracle_is_test_infrastructure_only():
    """Nothing outside tests/, __graft_entry__.smoke() and bench.py's CPU-baseline functions imports oracle/:
    every source file of the package and every measurement script under tools/ is free of such imports, and
    bench.py imports it only inside cpu_baseline / vae_cpu_baseline."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        hits = []
        for node in ast.walk(ast.parse(open(path).read())):
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                hits.append(node.lineno)
            if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                hits.append(node.lineno)
        return hits

    files = glob.glob(os.path.join(root, "omnihuman-1-hack_amd", "**", "*.py"), recursive=True) + \
        glob.glob(os.path.join(root, "tools", "**", "*.py"), recursive=True)
    assert files
    for f in files:
        assert not oracle_imports(f), f
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        inside = [n for n in ast.walk(fn) if isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle"]
        assert not inside or fn.name in ("cpu_baseline", "cpu_config1_and_3", "vae_cpu_baseline"), fn.name
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n)]
    assert not top


def test_state_dict_contract(wan_model_mod):
    """Same parameter names / shapes as the reference WanModel."""
    from oracle import wan_dit_oracle as O
    for mt, in_dim in (("t2v", 16), ("i2v", 36)):
        cfg = O.DiTConfig(model_type=mt, in_dim=in_dim, dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64,
                          text_len=32, freq_dim=64)
        m = wan_model_mod.WanModel(model_type=mt, in_dim=in_dim, dim=256, ffn_dim=512, num_heads=2, num_layers=2,
                                   text_dim=64, text_len=32, freq_dim=64)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == O.param_shapes(cfg)
        assert "freqs" not in got and m.freqs.shape == (1024, 64) and m.freqs.dtype == torch.complex128
        assert m.patch_size == (1, 2, 2) and m.use_checkpoint is True and len(m.blocks) == 2
        import copy
        copy.deepcopy(m)
    with pytest.raises(NotImplementedError):
        wan_model_mod.WanModel(dim=128, num_heads=2, num_layers=1)                # head_dim 64: kernel is built for 128


def test_vae_state_dict_contract(omh):
    from oracle import wan_vae_oracle as V
    vae_mod = importlib.import_module(PKG + ".wan.modules.vae")
    m = vae_mod.WanVAE_(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                        temperal_downsample=[False, True, True])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == V.param_shapes(V.VAEConfig())
    assert vae_mod.count_conv3d(m.decoder) == 33 and vae_mod.count_conv3d(m.encoder) == 26


def test_unipc_host_coefficients_match_oracle(omh):
    """The scheduler folds UniPC into 7 scalars per step; emulate the kernel's formula on CPU."""
    from oracle import detgen, sampler_oracle as SO
    sch = importlib.import_module(PKG + ".wan.utils.fm_solvers_unipc")
    for n, shift in ((6, 3.0), (50, 5.0), (2, 1.0), (1, 5.0)):
        s = sch.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(n, device="cpu", shift=shift)
        o = SO.UniPCOracle(n, shift)
        assert torch.equal(s.sigmas, o.sigmas) and torch.equal(s.timesteps, o.timesteps)
        x = xo = torch.from_numpy(detgen.normalish("coef/x", (4, 5)))
        last, m1, m2, this_order, lower = None, None, None, None, 0
        for i in range(n):
            v = torch.from_numpy(detgen.normalish(f"coef/v{i}", (4, 5)))
            order_p = min(2, n - i, lower + 1)
            sigma, corr, pred = sch.unipc_coefficients(s.sigmas, i, order_p, this_order if i > 0 else None)
            mt = x - sigma * v
            xc = x
            if corr is not None:
                xc = corr[0] * last + corr[3] * mt + corr[1] * m1 + (corr[2] * m2 if m2 is not None else 0)
            xn = pred[0] * xc + pred[1] * mt + (pred[2] * m1 if m1 is not None else 0)
            m2, m1, last, this_order, lower, x = m1, mt, xc, order_p, min(lower + 1, 2), xn
            xo = o.step(v, xo)
            assert float((x - xo).abs().max()) < 2e-5, (n, i)


def test_dpmpp_host_coefficients_match_oracle(omh):
    """DPM-Solver++ folded into 3 scalars per step; emulate the kernel's formula on CPU."""
    from oracle import detgen, sampler_oracle as SO
    sch = importlib.import_module(PKG + ".wan.utils.fm_solvers")
    assert np.allclose(sch.get_sampling_sigmas(6, 3.0), [1, 0.9375, 6 / 7, 0.75, 0.6, 0.375], rtol=1e-15)
    for n, shift in ((6, 3.0), (50, 5.0), (20, 5.0), (2, 1.0), (1, 5.0)):
        s = sch.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        ts, cnt = sch.retrieve_timesteps(s, device="cpu", sigmas=sch.get_sampling_sigmas(n, shift))
        o = SO.DPMSolverOracle(n, shift)
        assert cnt == n and torch.equal(s.sigmas, o.sigmas) and torch.equal(ts, o.timesteps)
        x = xo = torch.from_numpy(detgen.normalish("coef/x", (4, 5)))
        m1, lower = None, 0
        for i in range(n):
            v = torch.from_numpy(detgen.normalish(f"coef/v{i}", (4, 5)))
            first = lower < 1 or i == n - 1
            sigma, (cx, c0, c1) = sch.dpmpp_coefficients(s.sigmas, i, 1 if first else 2)
            assert all(np.isfinite(c) for c in (sigma, cx, c0, c1))
            mt = x - sigma * v
            xn = cx * x + c0 * mt + (c1 * m1 if not first else 0)
            m1, lower, x = mt, min(lower + 1, 2), xn
            xo = o.step(v, xo)
            assert float((x - xo).abs().max()) < 2e-5, (n, i)
    with pytest.raises(ValueError):
        sch.retrieve_timesteps(s, timesteps=[1], sigmas=[1.0])


def test_configs(omh):
    cfgs = importlib.import_module(PKG + ".wan.configs")
    c = cfgs.t2v_1_3B
    assert (c.dim, c.ffn_dim, c.num_heads, c.num_layers, c.text_len) == (1536, 8960, 12, 30, 512)
    assert cfgs.i2v_14B.dim == 5120 and cfgs.SIZE_CONFIGS["480*832"] == (480, 832)
    kw = cfgs.dit_kwargs(c)
    assert kw["dim"] // kw["num_heads"] == 128


def test_from_pretrained_diffusers_layout_and_module_surface(omh, tmp_path):
    """`WanModel.from_pretrained(dir)` on the diffusers layout the reference loads (text2video.py:86: config.json
    with diffusers' private keys + sharded diffusion_pytorch_model-*.safetensors), and the module surface callers
    poke at: attributes, indexable hookable blocks, deepcopy, stable parameter order, state-dict keys."""
    import copy
    import json
    from safetensors.torch import save_file
    from oracle import wan_dit_oracle as O
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    kw = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=3, text_dim=64, text_len=32, freq_dim=64)
    sd = O.synth_state_dict(O.DiTConfig(**kw), "pretrained")
    cfg = dict(kw, model_type="t2v", patch_size=[1, 2, 2], in_dim=16, out_dim=16, window_size=[-1, -1], qk_norm=True,
               cross_attn_norm=True, eps=1e-6, _class_name="WanModel", _diffusers_version="0.30.0")
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    keys = sorted(sd)
    save_file({k: sd[k].contiguous() for k in keys[::2]}, str(tmp_path / "diffusion_pytorch_model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[1::2]}, str(tmp_path / "diffusion_pytorch_model-00002-of-00002.safetensors"))
    m = model_mod.WanModel.from_pretrained(str(tmp_path))
    got = m.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert "freqs" not in got                                     # not a buffer in the reference (model.py:484)
    assert (m.dim, m.num_heads, m.qk_norm, m.eps, tuple(m.patch_size), m.text_len, m.freq_dim) == \
        (256, 2, True, 1e-6, (1, 2, 2), 32, 64)
    assert len(m.blocks) == 3 and hasattr(m.blocks[1], "register_forward_hook") and hasattr(m, "use_checkpoint")
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in copy.deepcopy(m).named_parameters()]
    m2 = copy.deepcopy(m)
    with torch.no_grad():
        m2.head.head.weight.add_(1.0)
    assert not torch.equal(m2.head.head.weight, m.head.head.weight)   # deep copies do not share storage
    (tmp_path / "diffusion_pytorch_model-00002-of-00002.safetensors").unlink()
    with pytest.raises(RuntimeError):                             # missing shard -> strict load fails loudly
        model_mod.WanModel.from_pretrained(str(tmp_path))
    if not torch.cuda.is_available():
        with pytest.raises(Exception, match="MI355X|GPU|cuda"):
            m([torch.zeros(16, 1, 4, 4)], torch.tensor([1.0]), [torch.zeros(3, 64)], 4)   # no CPU fallback


def test_checkpoint_flag_as_memory_policy(omh, wan_model_mod, monkeypatch):
    """model.use_checkpoint (model.py:404,544-553) is honoured as a memory policy (model_train.keep_activations): False
    keeps the activations; True keeps them under the default "auto" policy when a step's worth fits in half of the free
    HBM and re-runs the blocks otherwise — or always with checkpoint_policy = "always" / OMH_CHECKPOINT_POLICY."""
    mt = importlib.import_module(PKG + ".wan.modules.model_train")
    from oracle import make_golden
    m = wan_model_mod.WanModel(num_layers=3, **make_golden.TINY)
    assert m.use_checkpoint is True and m.checkpoint_policy == "auto"
    rows = 6240
    need = mt.activation_bytes(m, rows, batch=4)
    pend = mt.pending_step_bytes(m)                    # no gradients, no optimizer state yet: 12 bytes per parameter
    assert pend == 12 * sum(p.numel() for p in m.parameters() if p.requires_grad)
    # once the optimizer holds the moments they are part of the memory in use, not of what is pending
    import weakref
    moments = [torch.zeros(1) for _ in m.parameters()]
    for p_, mo in zip(m.parameters(), moments):
        p_._omh_moments_allocated = weakref.ref(mo)
    assert mt.pending_step_bytes(m) == 4 * sum(p.numel() for p in m.parameters() if p.requires_grad)
    del moments, mo                                                   # the optimizer died: counted again
    assert mt.pending_step_bytes(m) == pend
    free = {"bytes": int(4 * need) + pend}
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda device=None: (free["bytes"], 0))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda device=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda device=None: 0)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.delenv("OMH_CHECKPOINT_POLICY", raising=False)
    dev = torch.device("cpu")
    assert mt.keep_activations(m, rows, dev, batch=4) is True                   # fits in half of the free memory
    free["bytes"] = int(1.5 * need) + pend
    assert mt.keep_activations(m, rows, dev, batch=4) is False                  # does not: recompute
    free["bytes"] = int(2.5 * need)                                   # would fit, but not beside gradients + AdamW state
    assert (mt.keep_activations(m, rows, dev, batch=4) is True) == (2.5 * need - pend > 2 * need)
    free["bytes"] = int(4 * need) + pend
    m.checkpoint_policy = "always"
    assert mt.keep_activations(m, rows, dev, batch=4) is False
    m.checkpoint_policy = "auto"
    monkeypatch.setenv("OMH_CHECKPOINT_POLICY", "always")
    assert mt.keep_activations(m, rows, dev, batch=4) is False
    m.use_checkpoint = False                                          # the reference's other branch: always kept
    assert mt.keep_activations(m, rows, dev, batch=4) is True


def test_attention_split_plans_without_a_gpu(omh):
    """ABI v8 workspace queries are pure host arithmetic (omh_tail_split_plan: 256 CUs assumed without a device): the
    training step's shapes — one clip (156 workgroups: dQ split 3 ways on 512 slots, dK / dV 3 ways on 256), four
    clips (624: nothing is split, measured not to pay), phases, the forward's stricter rule (16 key tiles per worker)."""
    import ctypes as C
    binding = importlib.import_module(PKG + "._lib")
    lib = binding.lib

    def bwd_bytes(B, H, Lq, Lk, phase=0):
        a = binding.AttnBwdArgs()
        a.B, a.H, a.Lq, a.Lk, a.phase = B, H, Lq, Lk, phase
        a.o32 = C.c_void_p(16)                                      # only its presence matters to the query
        return lib.omh_flash_attn_bwd_workspace_bytes(C.byref(a))
    tile = 128 * 128 * 4
    assert bwd_bytes(1, 12, 1560, 1560) == 156 * 3 * tile + 156 * 3 * 2 * tile
    assert bwd_bytes(1, 12, 1560, 1560, phase=2) == 156 * 3 * tile
    assert bwd_bytes(1, 12, 1560, 1560, phase=3) == 156 * 3 * 2 * tile
    assert bwd_bytes(1, 12, 1560, 1560, phase=1) == 0
    assert bwd_bytes(4, 12, 1560, 1560) == 0                        # 624 workgroups: more than a round, no split
    assert bwd_bytes(4, 12, 1560, 512) == 0                         # cross-attention dK / dV: 192 on 256, < 1/3 to gain
    # dQ (the stream: ONE workgroup per CU since round 5): 8 key tiles -> 2 workers would be 312 > 256, not split; dK/dV: 48 x 5 = 240 <= 256
    assert bwd_bytes(1, 12, 1560, 512) == 48 * 5 * 2 * tile

    def fwd_bytes(B, H, Lq, Lk, flags):
        a = binding.AttnArgs()
        a.B, a.H, a.Lq, a.Lk, a.flags = B, H, Lq, Lk, flags
        a.q_rs = a.k_rs = a.o_rs = H * 128
        return lib.omh_flash_attn_workspace_bytes(C.byref(a))
    both = binding.ATTN_SHORT_KERNEL | binding.ATTN_ALLOW_SPLIT
    assert fwd_bytes(1, 12, 1560, 1560, both) == 0                  # 25 key tiles: < 2 x 16 per worker
    assert fwd_bytes(1, 12, 1560, 4096, binding.ATTN_SHORT_KERNEL) == 0            # not allowed: never
    assert fwd_bytes(1, 12, 1560, 4096, both) == 156 * 3 * 128 * 129 * 4           # 64 key tiles: 3 workers of >= 16


def test_k_major_stream_generator_invariants():
    """csrc/gen_gemm_tn_w64.py (no GPU): both streams assemble the same k loop — 7 step bodies x 4 groups x 24 MFMAs —,
    name only the operands gemm_tn_w64.hip binds, wait on lgkmcnt before every MFMA whose fragments are in flight (the
    generator asserts the loop-carried set itself), and keep every LDS offset inside the 16-bit field."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = os.path.join(root, PKG, "csrc", "gen_gemm_tn_w64.py")
    txt = subprocess.run([sys.executable, gen], check=True, capture_output=True, text=True).stdout
    assert open(os.path.join(root, PKG, "csrc", "gemm_tn_w64_asm.inc")).read() == txt      # the committed stream is current
    streams = txt.split("#define OMH_GEMM_TN_W64_ASM_")[1:]
    assert [s.split(" ")[0] for s in streams] == ["ST", "ACC"]
    bound = {"mab", "nab", "voa0", "voa1", "vob0", "vob1", "vob2", "voc", "ra", "rb", "rc"}
    for s in streams:
        body = s.split("#define OMH_GEMM_TN_W64_CLOBBERS")[0]
        assert body.count("v_mfma_f32_32x32x16_bf16") == 7 * 4 * 24
        assert set(re.findall(r"%\[(\w+)\]", body)) == bound
        assert all(int(o) < 65536 for o in re.findall(r"ds_read_b64_tr_b16 [^\n]*offset:(\d+)", body))
        assert body.count("buffer_load_dwordx4") == 28 + 3 * 20 + 2 * 12      # prologue + 3 full step bodies + 2 last-but-one
    assert streams[0].count("buffer_store_dword ") == 24 * 16 and streams[1].count("buffer_load_dword ") == 24 * 16


def test_deferred_join_policy_without_a_gpu(omh):
    """model_train._may_defer_join (host logic only): the weight-gradient stream may be joined at the end of the pass only
    when no block parameter has a gradient in place that autograd would add to (the block backward adds into dense fp32
    ones itself) and nobody but this package's reducer hooks the parameters."""
    mt = importlib.import_module(PKG + ".wan.modules.model_train")
    par = importlib.import_module(PKG + ".parallel")

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([torch.nn.Linear(4, 4) for _ in range(2)])
            self.head = torch.nn.Linear(4, 4)
    m = Tiny()
    assert mt._may_defer_join(m) is (mt._DEFER_JOIN and mt._WGRAD_STREAM)
    m.head.weight.grad = torch.zeros(4, 4)                         # the head runs before the blocks: not a reason
    assert mt._may_defer_join(m) is True
    m.blocks[1].bias.grad = torch.zeros(4)                          # a gradient in place the block backward adds into
    assert mt._may_defer_join(m) is True
    m.direct_grad_accumulation = False                              # ... that autograd adds to, on the main stream
    assert mt._may_defer_join(m) is False
    del m.direct_grad_accumulation
    m.blocks[1].bias.grad = torch.zeros(8)[::2]                     # not something the kernels can add into (strided)
    assert mt._may_defer_join(m) is False
    m.blocks[1].bias.grad = None
    h = m.blocks[0].weight.register_post_accumulate_grad_hook(lambda p: None)
    assert mt._may_defer_join(m) is False
    h.remove()
    red = par.BucketedGradAllReduce.__new__(par.BucketedGradAllReduce)           # (its hook orders itself behind the stream)
    h = m.blocks[0].weight.register_post_accumulate_grad_hook(red._on_grad)
    assert mt._may_defer_join(m) is True
    h.remove()
    m.blocks[0].weight.requires_grad_(False)
    m.blocks[0].weight.grad = torch.zeros(4, 4)                     # frozen parameters do not count
    assert mt._may_defer_join(m) is True



def test_gradient_accumulation_targets_without_a_gpu(omh):
    """model_train._grad_targets (host logic only): which existing .grad tensors a block's backward may add into — none
    on the first micro-step, all dense fp32 ones afterwards; the autograd route (None) for a strided gradient,
    tensor hooks / foreign post-accumulate hooks, ``torch.autograd.grad`` w.r.t. the parameters, double backward, an
    initialised process group without this package's reducer, and when switched off."""
    mt = importlib.import_module(PKG + ".wan.modules.model_train")
    par = importlib.import_module(PKG + ".parallel")
    seen = []

    class Blk(torch.nn.Linear):
        pass

    class Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, model, idx, *params):
            ctx.model, ctx.idx = model, idx
            return x @ params[0].t() + params[1]

        @staticmethod
        def backward(ctx, g):
            seen.append(mt._grad_targets(ctx, ctx.model, ctx.idx))
            return g, None, None, g.t() @ g, g.sum(0)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([Blk(4, 4) for _ in range(2)])

        def forward(self, x):
            for i, b in enumerate(self.blocks):
                x = Node.apply(x, self, i, *[p for _, p in mt._block_params(self, i)])
            return x.sum()
    m = Tiny()
    x = torch.ones(3, 4)

    def run(fn=None):
        seen.clear()
        (fn or (lambda: m(x).backward()))()
        return list(seen)
    assert run() == [{}, {}]                                               # first micro-step: nothing in place
    got = run()                                                            # second: every parameter of both blocks
    assert [sorted(t) for t in got] == [["bias", "weight"]] * 2 and got[0]["weight"] is m.blocks[1].weight
    assert run(lambda: torch.autograd.grad(m(x), [m.blocks[0].weight, m.blocks[1].bias])) == [None, None]
    xg = torch.ones(3, 4, requires_grad=True)
    assert run(lambda: torch.autograd.grad(m(xg), [xg])) == [None, None]   # no AccumulateGrad node runs
    assert [sorted(t) for t in run(lambda: m(x).backward(inputs=[m.blocks[1].weight]))] == [["weight"]]
    got = run(lambda: m(x).backward(inputs=[m.blocks[0].bias]))            # block 1: gradients in place, none to accumulate
    assert got[0] is None and sorted(got[1]) == ["bias"]
    m.direct_grad_accumulation = False
    assert run() == [None, None]
    del m.direct_grad_accumulation
    h = m.blocks[1].weight.register_hook(lambda g: g)
    assert [t is None for t in run()] == [True, False]
    h.remove()
    h = m.blocks[0].bias.register_post_accumulate_grad_hook(lambda p: None)
    assert [t is None for t in run()] == [False, True]
    h.remove()
    red = par.BucketedGradAllReduce.__new__(par.BucketedGradAllReduce)
    red.enabled = False
    h = m.blocks[0].bias.register_post_accumulate_grad_hook(red._on_grad)   # this package's reducer: told by the block node
    assert all(t for t in run())
    h.remove()
    m.blocks[0].weight.grad = torch.zeros(4, 8)[:, ::2]                    # strided
    assert [t is None for t in run()] == [False, True]
    m.blocks[0].weight.grad = None                                         # (e.g. a frozen FFN: never gets a gradient)
    assert [sorted(t) for t in run()] == [["bias", "weight"], ["bias"]]
    assert run(lambda: m(x).backward(create_graph=True)) == [None, None]



def test_gemm_split_k_plan_without_a_gpu(omh, monkeypatch):
    """ABI v9: omh_gemm_workspace_bytes is pure host arithmetic — the slices of a product's contraction as a function of
    (M, N, K): four at one [16,1,60,104] clip's FFN-down / FFN-up input gradient (56 tiles of 256 x 192), two at two clips
    (104), none from four clips on, none for short contractions, other epilogues, forced kernel families, or switched off."""
    import ctypes as C
    binding = importlib.import_module(PKG + "._lib")
    from conftest import set_option
    for k_ in ("OMH_GEMM_KERNEL", "OMH_GEMM_TILE", "OMH_GEMM_SPLITK"):
        set_option(k_, None)

    def need(M, N, K, epi=binding.EPI_RESID, **kw):
        a = binding.GemmArgs()
        a.A = a.B = a.C = C.c_void_p(4096)                               # only alignment matters to the query
        a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.batch, a.epilogue = M, N, K, K, K, N, 1, epi
        for k_, v_ in kw.items():
            setattr(a, k_, v_)
        return binding.lib.omh_gemm_workspace_bytes(C.byref(a))
    assert need(1560, 1536, 8960) == 4 * 1792 * 1536 * 4
    assert need(3120, 1536, 8960) == 2 * 3328 * 1536 * 4
    assert need(1560, 1536, 8960, epi=binding.EPI_F32) == 4 * 1792 * 1536 * 4
    assert need(6240, 1536, 8960) == 0 and need(32760, 1536, 8960) == 0
    assert need(1560, 1536, 1536) == 0 and need(1560, 8960, 1536) == 0   # short contractions
    assert need(200, 1536, 8960) == 0                                    # less than one tile of rows: the 8-wave kernels
    assert need(1560, 1536, 8960, epi=binding.EPI_BF16) == 0 and need(1560, 1536, 8960, epi=binding.EPI_F32_ACCUM) == 0
    assert need(1560, 1536, 8960, batch=2) == 0 and need(1560, 1536, 8960, b_kmajor=1) == 0
    assert need(780, 776, 4416) == 3 * 1024 * 776 * 4                    # 69 k tiles: three slices of 23
    set_option("OMH_GEMM_SPLITK", "0")
    assert need(1560, 1536, 8960) == 0
    set_option("OMH_GEMM_SPLITK", None)
    set_option("OMH_GEMM_KERNEL", "8w")
    assert need(1560, 1536, 8960) == 0


def test_options_are_a_table_not_the_environment(omh, monkeypatch):
    """ABI v10: the dispatch switches live in one table that the library fills from the
    environment ONCE; after that a changed environment variable is invisible and omh_set_option is the only way in.
    Unknown keys and over-long values are refused; (NULL, NULL) restores the start-up values."""
    binding = importlib.import_module(PKG + "._lib")
    ops = importlib.import_module(PKG + ".ops")
    lib = binding.lib
    names = [lib.omh_option_name(i).decode() for i in range(lib.omh_option_count())]
    assert "GEMM_KERNEL" in names and "ATTN_KERNEL" in names and "CONV_TILE" in names and len(set(names)) == len(names)
    start = {n: ops.get_option(n) for n in names}
    monkeypatch.setenv("OMH_GEMM_KERNEL", "8w")                         # the library has been called already: not seen
    assert ops.get_option("GEMM_KERNEL") == start["GEMM_KERNEL"]
    ops.set_option("OMH_GEMM_KERNEL", "w64")                            # with or without the prefix
    assert ops.get_option("GEMM_KERNEL") == "w64"
    ops.set_option("GEMM_KERNEL", None)
    assert ops.get_option("OMH_GEMM_KERNEL") is None
    assert lib.omh_set_option(b"NO_SUCH_SWITCH", b"1") == -1 and lib.omh_get_option(b"NO_SUCH_SWITCH") is None
    assert lib.omh_set_option(b"GEMM_TILE", b"x" * 48) == -3
    with ops.options(GEMM_TILE="big", CONV_TILE="w64"):
        assert ops.get_option("GEMM_TILE") == "big" and ops.get_option("CONV_TILE") == "w64"
    assert ops.get_option("GEMM_TILE") == start["GEMM_TILE"]
    ops.set_option("LN_RPW", "2")
    ops.set_deterministic(True)
    ops.reset_options()
    assert {n: ops.get_option(n) for n in names} == start and ops.get_option("DETERMINISTIC") is None
    # the launch path holds no getenv: the one-time seeding in dit_elementwise.hip is the only call in the sources
    import glob
    hits = []
    for f in glob.glob(os.path.join(ROOT, PKG, "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, PKG, "csrc", "*.h")):
        src = re.sub(r"//[^\n]*", "", open(f).read())
        hits += [(os.path.basename(f), m.start()) for m in re.finditer(r"\bgetenv\s*\(", src)]
    assert len(hits) == 2 and all(h[0] == "dit_elementwise.hip" for h in hits), hits


def test_pack_registry_is_keyed_by_parameter_identity(omh, wan_model_mod):
    """the fused AdamW + pack path must never trust a raw address.  A parameter whose storage was
    replaced (offload / .to() without a refresh()), an unrelated tensor that the allocator placed at a packed weight's
    old address, and a dead TrainPacks all resolve to None; a re-layout and the death of the packs purge their rows."""
    import gc
    mt = importlib.import_module(PKG + ".wan.modules.model_train")
    m = wan_model_mod.WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=8, freq_dim=64)
    packs = mt.TrainPacks.of(m)
    packs._layout(m)
    w = m.blocks[0].self_attn.o.weight
    ent = mt.pack_entry_of(w)
    assert ent is not None and ent[0] is packs and packs.params[ent[1]] is w
    n_rows = len(packs.rows)
    assert sum(1 for v in mt._PACK_REGISTRY.values() if v[0]() is packs) == n_rows
    # an unrelated tensor viewing the very same storage (what a recycled address looks like) is not the parameter
    alias = torch.nn.Parameter(w.data.view(-1)[: w.numel()].view_as(w))
    assert alias.data_ptr() == w.data_ptr() and mt.pack_entry_of(alias) is None
    # the parameter's storage moved: the row's address is stale
    old = w.data
    w.data = old.clone()
    assert mt.pack_entry_of(w) is None
    w.data = old
    assert mt.pack_entry_of(w) is not None
    # a parameter that is not packed at all
    assert mt.pack_entry_of(m.head.head.weight) is None
    # re-layout: same count of rows, no stale duplicates
    packs._layout(m)
    assert sum(1 for v in mt._PACK_REGISTRY.values() if v[0]() is packs) == n_rows
    # the packs die: their rows leave the registry
    keys = list(packs._registry_keys)
    del packs, ent
    m.__dict__.pop("_train_packs")
    gc.collect()
    assert not any(k in mt._PACK_REGISTRY for k in keys)
    assert mt.pack_entry_of(w) is None
