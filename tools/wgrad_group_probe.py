"""A block's weight-gradient group alone on the chip: 128 x 128 tiles (two workgroups per CU) vs 256 x 256 (one)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / reps * 1e3, 1)


R = int(os.environ.get("ROWS", "6240"))
shapes = [(1536, 1536, R), (1536, 1536, R), (1536, 1536, R), (3072, 1536, 2048 * R // 6240), (4608, 1536, R)]
items = []
flop = 0
for (M, N, K) in shapes:
    items.append((torch.randn(K, M, device="cuda").bfloat16(), torch.randn(K, N, device="cuda").bfloat16(),
                  torch.empty(M, N, dtype=torch.float32, device="cuda"), False))
    flop += 2.0 * M * N * K
ffn = [(torch.randn(R, 8960, device="cuda").bfloat16(), torch.randn(R, 1536, device="cuda").bfloat16(),
        torch.empty(8960, 1536, dtype=torch.float32, device="cuda"), False),
       (torch.randn(R, 1536, device="cuda").bfloat16(), torch.randn(R, 8960, device="cuda").bfloat16(),
        torch.empty(1536, 8960, dtype=torch.float32, device="cuda"), False)]
fflop = 2 * 2.0 * 8960 * 1536 * R
for tile in ("small", "big", "w64"):
    ops.set_option("OMH_GEMM_TN_GROUP_TILE", "small" if tile == "w64" else tile)
    ops.set_option("OMH_GEMM_TN_W64", "1" if tile == "w64" else "0")
    f1 = t(lambda: (ops.gemm_tn(ffn[0][0], ffn[0][1], out=ffn[0][2]), ops.gemm_tn(ffn[1][0], ffn[1][1], out=ffn[1][2])))
    f2 = t(lambda: ops.gemm_tn_grouped(ffn))
    fall = t(lambda: ops.gemm_tn_grouped(items + ffn))
    print(tile, "FFN pair: two launches", f1, "us", round(fflop / f1 / 1e6, 1), "TFLOP/s; grouped", f2, "us; everything in one launch",
          fall, "us", round((flop + fflop) / fall / 1e6, 1), "TFLOP/s", flush=True)
    us = t(lambda: ops.gemm_tn_grouped(items))
    us2 = t(lambda: (ops.gemm_tn_grouped(items[:4]), ops.gemm_tn_grouped(items[4:])))
    print(tile, "one launch", us, "us", round(flop / us / 1e6, 1), "TFLOP/s; as {o,cq,co,ckv}+{qkv}", us2, "us", flush=True)
