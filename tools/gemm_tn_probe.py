"""Weight-gradient shapes: omh_gemm_bf16_tn on (dy, x) as stored vs two transposes + the NT kernel."""
import importlib, json, os, sys, torch
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


for (M, N, K) in ((1536, 1536, 6240), (3072, 1536, 6240), (8960, 1536, 6240), (1536, 8960, 6240), (1536, 1536, 1560),
                  (8960, 1536, 1560), (1536, 1536, 2048), (64, 1536, 6240)):
    dy = torch.randn(K, M, device="cuda").bfloat16()
    x = torch.randn(K, N, device="cuda").bfloat16()
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")

    def nt():
        dyT, xT = ops.transpose_bf16(dy), ops.transpose_bf16(x)
        Rp = dyT.shape[1]
        ops.gemm_raw(ops.ptr(dyT), ops.ptr(xT), ops.ptr(out), M, N, Rp, Rp, Rp, N, ops.EPI_F32)

    def nt_gemm_only(dyT=ops.transpose_bf16(dy), xT=ops.transpose_bf16(x)):
        Rp = dyT.shape[1]
        ops.gemm_raw(ops.ptr(dyT), ops.ptr(xT), ops.ptr(out), M, N, Rp, Rp, Rp, N, ops.EPI_F32)
    row = {"transposes+nt": t(nt), "nt_gemm_only": t(nt_gemm_only)}
    for tile in ("big", "small"):
        ops.set_option("OMH_GEMM_TN_TILE", tile)
        row["tn_" + tile] = t(lambda: ops.gemm_tn(dy, x, out=out))
    ops.set_option("OMH_GEMM_TN_TILE", None)
    row["tn_auto"] = t(lambda: ops.gemm_tn(dy, x, out=out))
    print(f"{M}x{N}x{K}", row, flush=True)
