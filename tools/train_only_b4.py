"""Training steps of bench.py's training leg (reference quirks on) at OMH_TRAIN_BATCH clips per GPU (default 4) —
for rocprofv3 --kernel-trace --stats (tools/rocprof_train.sh)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
bsz = int(os.environ.get("OMH_TRAIN_BATCH", "4"))
print(bench.train_bench(model, dev, 1, None, steps=6, warmup=2, bsz=bsz, policy="always" if os.environ.get("OMH_TRAIN_CKPT", "1") == "1" else "auto"))
