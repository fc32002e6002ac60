"""Debug aid: the accumulation legs of bench.train_bench over the forced one-rank RCCL group, with the traceback."""
import os, sys, traceback, importlib
os.environ.update(OMH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
par = importlib.import_module("omnihuman-1-hack_amd.parallel")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
model = bench.build_model(dev)
names = {id(p): n for n, p in model.named_parameters()}
R = par.BucketedGradAllReduce
orig_launch, orig_on = R._launch, R._on_grad
def launch(self, i):
    if i <= 2:
        st = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack(limit=6)[:-1]]
        print("LAUNCH bucket", i, "ready", len(self._ready[i]), "of", len(self.buckets[i]), st, flush=True)
    return orig_launch(self, i)
def on(self, p):
    if self.enabled:
        i = self._bucket_of[id(p)]
        if i == 1:
            st = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack(limit=4)[:-1]]
            print("ON", names[id(p)], "work", self._work[i] is not None, "dup", id(p) in self._ready[i], st, flush=True)
    return orig_on(self, p)
R._launch, R._on_grad = launch, on
orig_init = R.__init__
def init(self, *a, **k):
    orig_init(self, *a, **k)
    for h in self._hooks: h.remove()
    self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
    print("bucket1:", [names[id(p)] for p in self.buckets[1]], flush=True)
R.__init__ = init
for kw in (dict(bsz=2, accum=4, steps=1, warmup=1),):
    try:
        r = bench.train_bench(model, dev, 1, dist, **kw)
        print(kw, "ok", r["ms_per_step"])
    except Exception:
        print(kw, "FAILED")
        traceback.print_exc(limit=3, file=sys.stdout)
        break
