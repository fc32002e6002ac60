"""Eight training steps at 4 clips per GPU with the reference's quirks (bench.py's primary training leg) — for
rocprofv3 --kernel-trace --stats."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
print(bench.train_bench(model, dev, 1, None, steps=6, warmup=2, bsz=4))
