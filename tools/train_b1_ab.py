"""One-clip training step (bench.py's primary leg at OMH_TRAIN_BATCH clips, default 1) with the attention backward streams
against the HIP kernels, alternated INSIDE one process (the leg's time differs by ~10 % from process to process)."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
bsz = int(os.environ.get("OMH_TRAIN_BATCH", "1"))
for rnd in range(3):
    for opt in ("0", "k", None):
        ops.set_option("OMH_ATTN_BWD_W64", opt)
        r = bench.train_bench(model, dev, 1, None, steps=10, warmup=2, bsz=bsz)
        print(f"round {rnd} OMH_ATTN_BWD_W64={opt}: {r['ms_per_step']:.2f} ms per step", flush=True)
