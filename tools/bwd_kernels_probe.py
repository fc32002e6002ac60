"""Isolated timings of the training step's non-GEMM backward kernels at the step's shapes (GPU box):
    python tools/bwd_kernels_probe.py [B]"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
ptr = ops.ptr


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    S, d, H = 1560, 1536, 12
    R = B * S
    dev = "cuda"
    x = torch.randn(R, d, device=dev)
    dh = torch.randn(R, d, device=dev)
    dx = torch.randn(R, d, device=dev)
    mod = torch.randn(6, d, device=dev)
    e0 = torch.randn(B, 6, d, device=dev)
    d_eb = torch.zeros(B, 6, d, device=dev)
    y = torch.randn(R, d, device=dev).bfloat16()
    dyn = torch.empty(R, d, device=dev, dtype=torch.bfloat16)
    res = {"B": B}
    res["ln_bwd_old_us"] = timeit(lambda: ops.layernorm_modulate_bwd_raw(ptr(x), ptr(dh), ptr(dx), R, d, 1e-6, 1.0, ptr(mod, d), ptr(e0, d), 6 * d, ptr(d_eb, d), ptr(d_eb, 0), 6 * d, S))
    res["ln_bwd2_us"] = timeit(lambda: ops.layernorm_modulate_bwd2(x, dh, dx, R, d, 1e-6, 1.0, ptr(mod, d), ptr(e0, d), 6 * d, ptr(d_eb, d), ptr(d_eb, 0), 6 * d, S))
    res["ln_bwd2_next_us"] = timeit(lambda: ops.layernorm_modulate_bwd2(x, dh, dx, R, d, 1e-6, 1.0, ptr(mod, d), ptr(e0, d), 6 * d, ptr(d_eb, d), ptr(d_eb, 0), 6 * d, S, dy_next=dyn, gate_const=1.0))
    res["ln_bwd2_next_gate_us"] = timeit(lambda: ops.layernorm_modulate_bwd2(x, dh, dx, R, d, 1e-6, 1.0, ptr(mod, d), ptr(e0, d), 6 * d, ptr(d_eb, d), ptr(d_eb, 0), 6 * d, S, dy_next=dyn, y_next=y, gate_const=0.0, gate0=ptr(mod, 2 * d), gate1=ptr(e0, 2 * d), gate1_stride=6 * d, dgate=ptr(d_eb, 2 * d), dgate_stride=6 * d))
    res["gated_resid_bwd_us"] = timeit(lambda: ops.gated_residual_bwd_raw(ptr(dx), ptr(y), ptr(dyn), ptr(d_eb, 2 * d), 6 * d, R, d, 0.0, ptr(mod, 2 * d), ptr(e0, 2 * d), 6 * d, S))
    res["ln_bwd_bytes_MB"] = R * d * 16 / 1e6
    # rms
    qk = torch.randn(R, 2 * d, device=dev).bfloat16()
    dqkv = torch.randn(R, 3 * d, device=dev).bfloat16()
    w = torch.rand(d, device=dev) + 0.5
    dw = torch.zeros(2, d, device=dev)
    cos = torch.randn(1024, 64, device=dev)
    sin = torch.randn(1024, 64, device=dev)
    grid = torch.tensor([[1, 30, 52]] * B, dtype=torch.int32, device=dev)
    res["rms_bwd_old_one_seg_us"] = timeit(lambda: ops.rmsnorm_rope_bwd_t_raw(ptr(qk), True, 2 * d, ptr(dqkv), True, 3 * d, ptr(dqkv), 3 * d, ptr(dw), R, d, ptr(w), 1e-6, 1, ptr(cos), ptr(sin), 1024, 128, ptr(grid), S))
    res["rms_bwd2_two_seg_us"] = timeit(lambda: ops.rmsnorm_rope_bwd2(ptr(qk), True, 2 * d, ptr(dqkv), True, 3 * d, ptr(dqkv), 3 * d, R, d, 1e-6, True, [w, w], [dw[0], dw[1]], qk.device, n_seg=2, seg_x=d, seg_dy=d, seg_dx=d, rope_cos=ptr(cos), rope_sin=ptr(sin), rope_len=1024, head_dim=128, grid=ptr(grid), seq_len=S))
    res["rms_bwd2_one_seg_us"] = timeit(lambda: ops.rmsnorm_rope_bwd2(ptr(qk), True, 2 * d, ptr(dqkv), True, 3 * d, ptr(dqkv), 3 * d, R, d, 1e-6, True, [w], [dw[0]], qk.device))
    # transposes
    vt = torch.randn(B, d, 1600, device=dev).bfloat16()
    res["transpose_vt_to_v_us"] = timeit(lambda: ops.transpose_bf16_batched(vt, S))
    q = torch.randn(R, d, device=dev).bfloat16()
    qt = torch.empty(B, d, 1600, device=dev, dtype=torch.bfloat16)
    res["transpose_q_to_qt_us"] = timeit(lambda: ops.transpose_bf16_raw(ptr(q), ptr(qt), S, d, d, 1600, batch=B, bs_in=S * d, bs_out=d * 1600))
    # attention backward
    k = torch.randn(R, d, device=dev).bfloat16()
    v = torch.randn(R, d, device=dev).bfloat16()
    do = torch.randn(R, d, device=dev).bfloat16()
    lse = torch.randn(B, H, S, device=dev) + 8.0
    kl = torch.full((B,), S, dtype=torch.int32, device=dev)
    out = (dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:])
    res["attn_bwd_self_us"] = timeit(lambda: ops.flash_attn_bwd(q, k, v, None, do, lse, kl, B, H, S, S, out=out), 10)
    kc = torch.randn(B * 512, d, device=dev).bfloat16()
    vc = torch.randn(B * 512, d, device=dev).bfloat16()
    dkv = torch.empty(B * 512, 2 * d, device=dev, dtype=torch.bfloat16)
    klc = torch.full((B,), 512, dtype=torch.int32, device=dev)
    res["attn_bwd_cross_us"] = timeit(lambda: ops.flash_attn_bwd(q, kc, vc, None, do, lse, klc, B, H, S, 512, out=(dqkv[:, :d], dkv[:, :d], dkv[:, d:])), 10)
    o32 = torch.randn(R, d, device=dev)
    res["attn_bwd2_self_us"] = timeit(lambda: ops.flash_attn_bwd(q, k, v, None, do, lse, kl, B, H, S, S, out=out, o32=o32), 10)
    res["attn_bwd2_cross_us"] = timeit(lambda: ops.flash_attn_bwd(q, kc, vc, None, do, lse, klc, B, H, S, 512, out=(dqkv[:, :d], dkv[:, :d], dkv[:, d:]), o32=o32), 10)
    res["attn_bwd_self_gflop"] = 10 * S * S * 128 * H * B / 1e9
    print(json.dumps(res, indent=1))


main()
