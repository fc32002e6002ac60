"""Which tile configuration (256x256 'big', 128x128 'small', 64x64 'tiny') is fastest per GEMM shape of the
S=1560 regimes (1 and 4 clips, forward / dgrad / wgrad shapes)?  Prints us per launch for each and what the
library's own dispatch picks.  GPU box only."""
import importlib, json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")


def t(M, N, K, epi, reps=30):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    odt = torch.float32 if epi == ops.EPI_F32 else torch.bfloat16
    out = torch.empty(M, N, dtype=odt, device="cuda")
    for _ in range(3):
        ops.gemm(a, w, out=out, epilogue=epi)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gemm(a, w, out=out, epilogue=epi)
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / reps * 1e3, 1)


shapes = []
for M in (1560, 3120, 6240, 12480):
    for N, K in ((1536, 1536), (3072, 1536), (8960, 1536), (1536, 8960)):
        shapes.append((M, N, K))
for R in (1560, 6240):                       # wgrad: [N_out, K_in] over R rows
    for M, N in ((1536, 1536), (3072, 1536), (8960, 1536), (1536, 8960)):
        shapes.append((M, N, (R + 7) // 8 * 8))
shapes += [(512, 1536, 1536), (2048, 1536, 1536), (1536, 512, 1536), (1536, 2048, 1536)]
res = {}
for M, N, K in shapes:
    row = {}
    for cfg in ("big", "small", "tiny", None):
        if cfg is None:
            os.environ.pop("OMH_GEMM_TILE", None)
        else:
            os.environ["OMH_GEMM_TILE"] = cfg
        row[cfg or "auto"] = t(M, N, K, ops.EPI_BF16)
    os.environ.pop("OMH_GEMM_TILE", None)
    best = min(("big", "small", "tiny"), key=lambda c: row[c])
    row["best"] = best
    row["auto_loss_pct"] = round(100 * (row["auto"] / row[best] - 1), 1)
    res[f"{M}x{N}x{K}"] = row
    print(f"{M}x{N}x{K}", row, flush=True)
