import importlib, json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
res = {}
for (M, N, K) in ((32760, 8960, 1536), (32760, 1536, 8960), (32760, 1536, 1536)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.gemm(a, w, out=out)
    e.record()
    torch.cuda.synchronize()
    res[f"{M}x{N}x{K}"] = round(s.elapsed_time(e) / 20 * 1e3, 1)
print(os.environ.get("OMH_LIB", "product"), json.dumps(res))
