"""Launches only the training step's self-attention backward (attention_bwd2.hip: delta, dQ, dK/dV kernels) at B clips of
S = 1560 — for rocprofv3 --pmc passes (tools/pmc_generic.sh) or, with TIME=1, a per-kernel timing."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
B, S, d, H = int(os.environ.get("B", 4)), int(os.environ.get("LK", 1560)), 1536, 12
SQ = int(os.environ.get("LQ", 1560))                           # queries per clip (the keys stay at S): the dK / dV loop length
R = B * S
g = torch.Generator(device="cuda").manual_seed(1)
k, v = [(torch.randn(R, d, device="cuda", generator=g) * 0.5).bfloat16() for _ in range(2)]
q, do = [(torch.randn(B * SQ, d, device="cuda", generator=g) * 0.5).bfloat16() for _ in range(2)]
lse = torch.randn(B, H, SQ, device="cuda", generator=g) + 8.0
kl = torch.full((B,), S, dtype=torch.int32, device="cuda")
o32 = torch.randn(B * SQ, d, device="cuda", generator=g)
dqb = torch.empty(B * SQ, d, device="cuda", dtype=torch.bfloat16)
dkv = torch.empty(R, 2 * d, device="cuda", dtype=torch.bfloat16)
out = (dqb, dkv[:, :d], dkv[:, d:])
pre = bool(int(os.environ.get("PRE", 1)))                  # the training step hands over a pre-scaled q
ph = int(os.environ.get("PHASE", 0))                        # 2: dQ only, 3: dK / dV only (after one full call)
delta = torch.empty(B, H, SQ, device="cuda")
kw = dict(out=out, o32=o32, q_prescaled=pre, delta=delta)
ops.flash_attn_bwd(q, k, v, None, do, lse, kl, B, H, SQ, S, phase=1, **kw)
run = lambda: ops.flash_attn_bwd(q, k, v, None, do, lse, kl, B, H, SQ, S, phase=ph, **kw)
for _ in range(3):
    run()
torch.cuda.synchronize()
if os.environ.get("TIME"):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        a.record()
        for _ in range(50):
            run()
        b.record(); torch.cuda.synchronize()
        print(f"attention backward, {B} clips, {SQ} queries x {S} keys, pre-scaled q {pre}, phase {ph}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call")
