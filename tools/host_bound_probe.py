"""Is the one-clip training step bound by the host?  Enqueue time of a step (no synchronisation inside) against its GPU
time, and the same with the backward's Python profile:  python tools/host_bound_probe.py [B]"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
optim = importlib.import_module("omnihuman-1-hack_amd.optim")
model.train().requires_grad_(True)
model.reference_ffn_freeze, model.use_checkpoint, model.checkpoint_policy = True, True, "auto"
opt = optim.AdamW(model.parameters(), lr=5e-6)
g = torch.Generator(device=dev).manual_seed(7)
x = torch.randn(B, 16, 1, 60, 104, device=dev, generator=g)
ctx = [torch.randn(512, 4096, device=dev, generator=g) for _ in range(B)]
tgt = torch.randn(B, 16, 1, 60, 104, device=dev, generator=g)
t = torch.full((B,), 999.0, device=dev)


def step():
    out = model(list(x), t=t, context=ctx, seq_len=1560)
    loss = sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, tgt))
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
print(f"B={B}: enqueue {sum(enq) / len(enq):.2f} ms (min {min(enq):.2f}), enqueue + drain {sum(tot) / len(tot):.2f} ms (min {min(tot):.2f})")
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"back to back: {(time.perf_counter() - t0) * 100:.2f} ms per step")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
