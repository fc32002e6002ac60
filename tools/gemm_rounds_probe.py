"""One launch over R rounds of 256x256 tiles vs R launches of exactly one round (256 tiles) each."""
import importlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
ops.set_option("OMH_GEMM_TILE", "big")

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n

for N, K, rows_per_round in [(2048, 1536, 8192), (1536, 1536, 10752), (8960, 1536, 1792), (1536, 8960, 10752), (3072, 1536, 5376)]:
    for R in (1, 3, 4, 6):
        M = rows_per_round * R
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        one = t(lambda: ops.gemm(a, w, out=out, epilogue=ops.EPI_BF16))
        def chunks():
            for r in range(R):
                sl = slice(r * rows_per_round, (r + 1) * rows_per_round)
                ops.gemm(a[sl], w, out=out[sl], epilogue=ops.EPI_BF16)
        many = t(chunks)
        fl = 2.0 * M * N * K
        tiles = (rows_per_round // 256) * ((N + 255) // 256)
        print(f"N={N} K={K} tiles/round={tiles} R={R}: one launch {one:.1f} us ({fl/one/1e6:.0f} TF), {R} launches {many:.1f} us ({fl/many/1e6:.0f} TF)", flush=True)
