"""Clock and power of the long-sequence attention kernel while it runs back to back (GPU box): random operands vs
all-zero operands, ~2 s each, amdgpu hwmon sampled every 50 ms.  Evidence for the "power-limited" reading of the
kernel's roofline fraction."""
import importlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
D, S, H = 128, 32760, 12
Sp = (S + 63) // 64 * 64


def run(kind):
    if kind == "random":
        q = torch.randn(1, S, H, D, device="cuda").bfloat16()
        k = torch.randn(1, S, H, D, device="cuda").bfloat16()
        vt = torch.zeros(1, H * D, Sp, dtype=torch.bfloat16, device="cuda")
        vt[:, :, :S] = torch.randn(1, H * D, S, device="cuda").bfloat16()
    else:
        q = torch.zeros(1, S, H, D, device="cuda", dtype=torch.bfloat16)
        k = torch.zeros_like(q)
        vt = torch.zeros(1, H * D, Sp, dtype=torch.bfloat16, device="cuda")
    o = torch.empty_like(q)
    for _ in range(20):
        ops.flash_attn(q, k, vt, None, out=o)
    torch.cuda.synchronize()
    tele = bench.Telemetry(0)
    tele.start()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 400
    s.record()
    for _ in range(n):
        ops.flash_attn(q, k, vt, None, out=o)
    e.record()
    torch.cuda.synchronize()
    t = tele.stop()
    ms = s.elapsed_time(e) / n
    tf = 4.0 * S * S * H * D / ms / 1e9
    r = {"operands": kind, "ms_per_launch": round(ms, 4), "tflops": round(tf, 1), "frac_of_2500": round(tf / 2500, 4), "telemetry": t}
    if t:
        r["frac_of_peak_at_measured_clock"] = round(tf / t["mfma_peak_at_mean_clock_tflops"], 4)
    return r


print(json.dumps({"kernel": "flash_attn_fwd_d128_w64_v2_kernel, 12 x 32760^2, D = 128", "runs": [run("random"), run("zeros"), run("random")]}, indent=1))
