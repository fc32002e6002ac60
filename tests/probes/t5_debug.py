import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "omnihuman-1-hack_amd"
ops = importlib.import_module(PKG + ".ops"); t5 = importlib.import_module(PKG + ".wan.modules.t5")
from oracle import encoders_oracle as E, make_golden
tc, vc, ids, mask, img = make_golden.encoder_cases()
sd = E.t5_state_dict(tc, "golden/t5")
def rr(a, b): return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())
dev = "cuda"
L, H, Dh, d = 24, tc.num_heads, 64, tc.dim
x = sd["token_embedding.weight"][ids[0]]
xg = ops.gather_rows(sd["token_embedding.weight"].cuda(), ids[:1].cuda())[0]
print("gather", rr(xg, x))
n = E.t5_layernorm(x, sd["blocks.0.norm1.weight"])
nf, nb = ops.rmsnorm_f32(xg.contiguous(), sd["blocks.0.norm1.weight"].cuda(), 1e-6)
print("norm f32", rr(nf, n), "bf16", rr(nb, n))
import torch.nn.functional as F
q = F.linear(n, sd["blocks.0.attn.q.weight"]); k = F.linear(n, sd["blocks.0.attn.k.weight"]); v = F.linear(n, sd["blocks.0.attn.v.weight"])
wq, wk, wv = (sd[f"blocks.0.attn.{a}.weight"].cuda().bfloat16() for a in "qkv")
qg, kg = ops.gemm(nb, wq), ops.gemm(nb, wk)
print("q", rr(qg, q), "k", rr(kg, k), "q std", float(q.std()))
vt = torch.empty(tc.dim_attn, L, dtype=torch.bfloat16, device=dev)
ops.gemm_raw(ops.ptr(wv), ops.ptr(nb), ops.ptr(vt), tc.dim_attn, L, d, d, d, L, ops.EPI_BF16)
print("vt", rr(vt, v.t()))
s = torch.empty(H * L, L, dtype=torch.float32, device=dev)
ops.gemm_raw(ops.ptr(qg), ops.ptr(kg), ops.ptr(s), L, L, Dh, d, d, L, ops.EPI_F32, batch=H, strideA=Dh, strideB=Dh, strideC=L * L)
sref = torch.einsum("inc,jnc->nij", q.view(L, H, Dh), k.view(L, H, Dh))
sref_b = torch.einsum("inc,jnc->nij", qg.float().cpu().view(L, H, Dh), kg.float().cpu().view(L, H, Dh))
print("scores vs fp32", rr(s.view(H, L, L), sref), "vs same bf16 operands", rr(s.view(H, L, L), sref_b), "score std", float(sref.std()))
bucket = t5.relative_position_buckets(L, L, 32).cuda()
table = sd["blocks.0.pos_embedding.embedding.weight"].cuda()
p = ops.softmax_bias_rows(s, H, L, 1.0, bucket, table, L, ldy=L)
bias = sd["blocks.0.pos_embedding.embedding.weight"][E.t5_relative_buckets(L, L, 32)].permute(2, 0, 1)
pref = torch.softmax(sref + bias, -1)
pref_b = torch.softmax(s.view(H, L, L).cpu() + bias, -1)
print("p vs fp32", rr(p.view(H, L, L), pref), "vs same scores", rr(p.view(H, L, L), pref_b), "max p mean", float(pref.max(-1).values.mean()))
o = torch.empty(L, d, dtype=torch.bfloat16, device=dev)
ops.gemm_raw(ops.ptr(p), ops.ptr(vt), ops.ptr(o), L, Dh, L, L, L, d, ops.EPI_BF16, batch=H, strideA=L * L, strideB=Dh * L, strideC=Dh)
oref = torch.einsum("nij,jnc->inc", pref, v.view(L, H, Dh)).reshape(L, d)
oref_b = torch.einsum("nij,jnc->inc", p.float().cpu().view(H, L, L), vt.float().cpu().t().reshape(L, H, Dh)).reshape(L, d)
print("o vs fp32", rr(o, oref), "vs same operands", rr(o, oref_b))
