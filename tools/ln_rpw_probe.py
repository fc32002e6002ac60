"""LayerNorm + modulate forward at the training step's shapes under OMH_LN_RPW = 1 / 2 / 4 (rows per wave)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
ptr = ops.ptr
for R in (6240, 3120, 1560, 24960, 32760, 65520):
    d = 1536
    x = torch.randn(R, d, device="cuda"); h = torch.empty(R, d, device="cuda", dtype=torch.bfloat16)
    mod = torch.randn(6, d, device="cuda"); e0 = torch.randn(max(1, R // 1560), 6, d, device="cuda")
    res = {}
    for rep in range(2):
        for rpw in ("1", "2", "4", "8", "16"):
            ops.set_option("OMH_LN_RPW", rpw)
            f = lambda: ops.layernorm_modulate_raw(ptr(x), ptr(h), R, d, 1e-6, 1.0, ptr(mod, d), ptr(e0, d), 6 * d, ptr(mod, 0), ptr(e0, 0), 6 * d, 1560)
            for _ in range(5): f()
            torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
            for _ in range(100): f()
            e.record(); torch.cuda.synchronize()
            res.setdefault(rpw, []).append(round(s.elapsed_time(e) * 10, 2))
    print(R, res, flush=True)
