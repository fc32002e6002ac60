"""us per launch of omh_rmsnorm_rope_bf16_pair at the headline's shape (32 760 rows x 2 x 1 536), the two forms interleaved."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
rows, d, D = 32760, 1536, 128
qk = (torch.randn(rows, 2 * d, device="cuda") * 1.3).bfloat16()
wq, wk = torch.rand(d, device="cuda") + 0.5, torch.rand(d, device="cuda") + 0.5
ang = torch.rand(1024, D // 2) * 6.28                        # (any table: timing only)
cos, sin = torch.cos(ang).float().cuda(), torch.sin(ang).float().cuda()
grid = torch.tensor([(21, 30, 52)], dtype=torch.int32, device="cuda")
q, k = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda"), torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
def run(): ops.rmsnorm_rope_bf16_pair_raw(ops.ptr(qk), 2 * d, d, ops.ptr(q), ops.ptr(k), rows, d, ops.ptr(wq), ops.ptr(wk), 1e-6, 1,
                                          ops.ptr(cos), ops.ptr(sin), 1024, D, ops.ptr(grid), rows, out_scale0=0.1275, out_scale1=1.0)
for rep in range(3):
    for form in ("0", "1"):
        ops.set_option("RMS_PAIR_ROW", form)
        for _ in range(5): run()
        torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
        for _ in range(50): run()
        e.record(); torch.cuda.synchronize(); us = s.elapsed_time(e) / 50 * 1e3
        print(f"RMS_PAIR_ROW={form} {us:.1f} us  {4.0 * rows * d * 2 / us / 1e6:.2f} TB/s", flush=True)
