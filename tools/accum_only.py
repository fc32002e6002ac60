"""Accumulation cycles of bench.py's training leg alone (4 micro-steps of OMH_TRAIN_BATCH clips per optimizer step;
OMH_ACCUM_DIRECT=0: gradients added by autograd instead of by the block backward) — for rocprofv3 --kernel-trace --stats:
the two kernel tables differ by autograd's add kernels and the weight-gradient stream's accumulate epilogue."""
import importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
bsz = int(os.environ.get("OMH_TRAIN_BATCH", "4"))
r = bench.train_bench(model, dev, 1, None, steps=5, warmup=2, bsz=bsz, accum=4,
                      direct_accum=os.environ.get("OMH_ACCUM_DIRECT", "1") == "1")
print(json.dumps({k: r[k] for k in ("clips_per_s", "ms_per_micro_step", "grads_accumulated_in_place")}))
