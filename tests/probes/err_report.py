"""Prints the measured bf16-vs-fp32-oracle errors behind the tolerances stated in tests/ (GPU)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import wan_dit_oracle as O, detgen, make_golden
mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.model")
def rel(a, b): return float((a.double().cpu() - b.double()).norm() / b.double().norm())
for layers in (2, 13):
    cfg, tag, xs, ctx, tt, seq_len, _, _ = make_golden.tiny_case("t2v", layers)
    sd = O.synth_state_dict(cfg, tag)
    ref = O.dit_forward(sd, cfg, xs, tt, ctx, seq_len)
    m = mod.WanModel(num_layers=layers, **make_golden.TINY); m.load_state_dict(sd); m = m.cuda().eval().requires_grad_(False)
    out = m([u.cuda() for u in xs], tt.cuda(), [c.cuda() for c in ctx], seq_len)
    print("tiny L=%d rel-rms" % layers, [round(rel(o, r), 5) for o, r in zip(out, ref)])
if os.environ.get("FULL", "1") == "1":
    cfg = O.DiTConfig.wan_t2v_1_3b(); sd = O.synth_state_dict(cfg, "wan1.3b")
    noise = torch.from_numpy(detgen.normalish("c1/noise", (16, 1, 60, 104))); cneg = torch.from_numpy(detgen.normalish("c1/neg", (37, 4096)))
    ref = O.dit_forward(sd, cfg, [noise], torch.tensor([999.]), [cneg], 1560)[0]
    m = mod.WanModel(**{k: getattr(cfg, k) for k in ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim", "num_heads", "num_layers", "qk_norm", "cross_attn_norm", "eps")})
    m.load_state_dict(sd); m = m.cuda().eval().requires_grad_(False)
    out = m([noise.cuda()], torch.tensor([999.]).cuda(), [cneg.cuda()], 1560)[0]
    print("Wan2.1-1.3B S=1560 rel-rms", round(rel(out, ref), 5), "max-abs", float((out.cpu() - ref).abs().max()), "ref abs-mean", float(ref.abs().mean()))
