"""Bitwise run-to-run repeatability of each forward kernel at the S=1560 (one latent frame) shapes of the 1.3B
model.  GPU box only.  Prints, per op, the number of differing elements and the max abs difference over REPS runs."""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "omnihuman-1-hack_amd"
REPS = 6


def rep(name, fn, res):
    ref = fn().clone()
    worst, cnt = 0.0, 0
    for _ in range(REPS):
        o = fn()
        d = (o.float() - ref.float()).abs()
        worst = max(worst, float(d.max()))
        cnt = max(cnt, int((d > 0).sum()))
    res[name] = {"maxabs": worst, "n_diff": cnt, "numel": ref.numel()}


def main():
    ops = importlib.import_module(PKG + ".ops")
    dev = torch.device("cuda", 0)
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1560
    d, N, D, F = 1536, 12, 128, 8960
    g = torch.Generator(device=dev).manual_seed(3)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    res = {}
    x = rn(S, d)
    e = rn(1, 6, d) * 0.1
    rep("layernorm_modulate", lambda: ops.layernorm_modulate(x, 1e-6, 1.0, mul1=e[:, 1].contiguous(), add1=e[:, 0].contiguous(), rows_per_batch=S), res)
    h = ops.layernorm_modulate(x, 1e-6, 1.0, mul1=e[:, 1].contiguous(), add1=e[:, 0].contiguous(), rows_per_batch=S)
    wqk = (rn(2 * d, d) * 0.03).bfloat16()
    bqk = rn(2 * d) * 0.1
    rep("gemm_qk_bf16[N=3072]", lambda: ops.gemm(h, wqk, bias=bqk, epilogue=ops.EPI_BF16), res)
    rep("gemm_qk_f32[N=3072]", lambda: ops.gemm(h, wqk, bias=bqk, epilogue=ops.EPI_F32), res)
    w1 = (rn(d, d) * 0.03).bfloat16()
    b1 = rn(d) * 0.1
    rep("gemm_f32[N=1536]", lambda: ops.gemm(h, w1, bias=b1, epilogue=ops.EPI_F32), res)
    wf = (rn(F, d) * 0.03).bfloat16()
    bf = rn(F) * 0.1
    rep("gemm_gelu[N=8960]", lambda: ops.gemm(h, wf, bias=bf, epilogue=ops.EPI_GELU_BF16), res)
    hid = ops.gemm(h, wf, bias=bf, epilogue=ops.EPI_GELU_BF16)
    w2 = (rn(d, F) * 0.01).bfloat16()
    gate = rn(1, d) * 0.1

    def resid(a, w, bias):
        xx = x.clone()
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(xx), S, d, a.shape[1], a.shape[1], w.shape[1], d, ops.EPI_RESID,
                     bias=ops.ptr(bias), bias_mode=ops.BIAS_N, gate0=None, gate1=ops.ptr(gate), gate1_stride=d,
                     gate_rows=S, gate_const=0.0)
        return xx
    rep("gemm_resid[K=8960]", lambda: resid(hid, w2, b1), res)
    rep("gemm_resid[K=1536]", lambda: resid(h, w1, b1), res)
    # V^T GEMM (operands swapped, bias along M)
    Lp = (S + 63) // 64 * 64

    def vt():
        out = torch.zeros(1, d, Lp, dtype=torch.bfloat16, device=dev)
        ops.gemm_raw(ops.ptr(w1), ops.ptr(h), ops.ptr(out), d, S, d, d, d, Lp, ops.EPI_BF16, bias=ops.ptr(b1),
                     bias_mode=ops.BIAS_M)
        return out
    rep("gemm_vT", vt, res)
    qk = ops.gemm(h, wqk, bias=bqk, epilogue=ops.EPI_BF16)
    model_mod = importlib.import_module(PKG + ".wan.modules.model")
    freqs = torch.cat([model_mod.rope_params(1024, D - 4 * (D // 6)), model_mod.rope_params(1024, 2 * (D // 6)),
                       model_mod.rope_params(1024, 2 * (D // 6))], dim=1)
    cos, sin = model_mod._rope_tables(freqs, dev)
    grid = torch.tensor([[S // 1560 if S % 1560 == 0 else 1, 30, 52]], dtype=torch.int32, device=dev) if S % 1560 == 0 \
        else torch.tensor([[1, 1, S]], dtype=torch.int32, device=dev)
    nw = rn(d) * 0.1 + 1

    def rr(off):
        out = torch.empty(S, d, dtype=torch.bfloat16, device=dev)
        ops.rmsnorm_rope_bf16_raw(ops.ptr(qk, off), 2 * d, ops.ptr(out), S, d, ops.ptr(nw), 1e-6, 1, ops.ptr(cos),
                                  ops.ptr(sin), 1024, D, ops.ptr(grid), S)
        return out
    rep("rmsnorm_rope_bf16", lambda: rr(0), res)
    q, k = rr(0), rr(d)
    vtt = vt()
    sl = torch.tensor([S], dtype=torch.int32, device=dev)
    for kern in ("base", "w64"):
        ops.set_option("OMH_ATTN_KERNEL", kern)
        rep(f"flash_attn_self[{kern}]", lambda: ops.flash_attn(q.view(1, S, N, D), k.view(1, S, N, D), vtt, k_lens=sl), res)
    ops.set_option("OMH_ATTN_KERNEL", None)
    rep("flash_attn_self[auto]", lambda: ops.flash_attn(q.view(1, S, N, D), k.view(1, S, N, D), vtt, k_lens=sl), res)
    L = 512
    kc = (rn(1, L, N, D)).bfloat16()
    vc = torch.zeros(1, d, L, dtype=torch.bfloat16, device=dev)
    vc[:, :, :120] = rn(1, d, 120).bfloat16()
    cl = torch.tensor([120], dtype=torch.int32, device=dev)
    rep("flash_attn_cross[Lk=512,len=120]", lambda: ops.flash_attn(q.view(1, S, N, D), kc, vc, k_lens=cl), res)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
