import importlib, os, sys, time, torch, timeit
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
n=20000
t=timeit.timeit(lambda: torch.cuda.current_stream().cuda_stream, number=n)/n*1e6
t2=timeit.timeit(lambda: torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()), number=n)/n*1e6
t3=timeit.timeit(lambda: ops._stream(), number=n)/n*1e6
a=torch.empty(1560,1536,device=dev)
t4=timeit.timeit(lambda: torch.empty(1560,1536,device=dev,dtype=torch.bfloat16), number=n)/n*1e6
t5=timeit.timeit(lambda: ops.ptr(a), number=n)/n*1e6
print(f"current_stream().cuda_stream {t:.2f} us; raw {t2:.2f} us; ops._stream {t3:.2f} us; torch.empty {t4:.2f} us; ops.ptr {t5:.2f}")
w=torch.randn(1536,1536,device=dev).bfloat16(); x=torch.randn(1560,1536,device=dev).bfloat16(); o=torch.empty(1560,1536,device=dev)
def g(): ops.gemm_raw(ops.ptr(x), ops.ptr(w), ops.ptr(o), 1560,1536,1536,1536,1536,1536, ops.EPI_F32)
for _ in range(10): g()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(2000): g()
t1=time.perf_counter(); torch.cuda.synchronize(); t2_=time.perf_counter()
print(f"gemm_raw host enqueue {(t1-t0)/2000*1e6:.2f} us per call; with drain {(t2_-t0)/2000*1e6:.2f}")
model = bench.build_model(dev)
optim = importlib.import_module("omnihuman-1-hack_amd.optim")
model.train().requires_grad_(True)
model.reference_ffn_freeze, model.use_checkpoint, model.checkpoint_policy = True, True, "auto"
opt = optim.AdamW(model.parameters(), lr=5e-6)
B=1
g_ = torch.Generator(device=dev).manual_seed(7)
x = torch.randn(B, 16, 1, 60, 104, device=dev, generator=g_)
ctx = [torch.randn(512, 4096, device=dev, generator=g_) for _ in range(B)]
tgt = torch.randn(B, 16, 1, 60, 104, device=dev, generator=g_)
tt = torch.full((B,), 999.0, device=dev)
def step(rec=None):
    torch.cuda.synchronize(); t0=time.perf_counter()
    out = model(list(x), t=tt, context=ctx, seq_len=1560)
    loss = sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, tgt))
    t1=time.perf_counter(); torch.cuda.synchronize(); t1b=time.perf_counter()
    loss.backward()
    t2=time.perf_counter(); torch.cuda.synchronize(); t2b=time.perf_counter()
    opt.step(); opt.zero_grad(set_to_none=True)
    t3=time.perf_counter(); torch.cuda.synchronize(); t3b=time.perf_counter()
    if rec is not None: rec.append(((t1-t0)*1e3,(t1b-t0)*1e3,(t2-t1b)*1e3,(t2b-t1b)*1e3,(t3-t2b)*1e3,(t3b-t2b)*1e3))
for _ in range(3): step()
rec=[]
for _ in range(8): step(rec)
import statistics
names=["fwd enqueue","fwd total","bwd enqueue","bwd total","opt enqueue","opt total"]
print({n: round(statistics.median(r[i] for r in rec),2) for i,n in enumerate(names)})
if os.environ.get("PROFILE"):
    import cProfile, pstats
    torch.autograd.set_multithreading_enabled(False)     # the backward's Python on this thread: visible to cProfile
    pr = cProfile.Profile()
    for _ in range(3):
        out = model(list(x), t=tt, context=ctx, seq_len=1560)
        loss = sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(out, tgt))
        torch.cuda.synchronize()
        pr.enable()
        loss.backward()
        pr.disable()
        opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats("ops.py|model_train.py|optim.py", 40)
