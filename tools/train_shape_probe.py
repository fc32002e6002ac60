"""Where a training step's GPU time goes, per C-ABI entry point AND shape (HIP events around every ops.* call on the
stream it is launched on; the events serialise nothing — kernels already run back to back).  GPU box:
    python tools/train_shape_probe.py [batch] [ckpt|stash] > gpurun_out/train_shapes_b4.json
"""
import importlib
import json
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PKG = "omnihuman-1-hack_amd"
SKIP = {"ptr", "check", "gemm", "flash_attn", "rmsnorm_rope", "layernorm_modulate", "transpose_bf16", "colsum_accum_multi"}


def main():
    bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    mode = sys.argv[2] if len(sys.argv) > 2 else "ckpt"
    dev = torch.device("cuda", 0)
    ops = importlib.import_module(PKG + ".ops")
    trainer = importlib.import_module(PKG + ".trainer")
    optim = importlib.import_module(PKG + ".optim")
    model = bench.build_model(dev)
    model.use_checkpoint, model.checkpoint_policy = True, ("auto" if mode == "stash" else "always")
    rec = defaultdict(list)
    on = [False]

    def wrap(name, fn):
        def w(*a, **k):
            if not on[0]:
                return fn(*a, **k)
            ints = tuple(x for x in a if isinstance(x, int) and not isinstance(x, bool))[:7]
            shp = tuple(tuple(x.shape) for x in a if torch.is_tensor(x))[:3]
            extra = tuple((kk, v) for kk, v in sorted(k.items()) if isinstance(v, (int, bool)) and kk in ("batch", "b_kmajor", "accumulate"))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            rec[(name, ints, shp, extra)].append((s, e))
            return r
        return w

    for name in dir(ops):
        fn = getattr(ops, name)
        if callable(fn) and not name.startswith("_") and name not in SKIP and getattr(fn, "__module__", "") == ops.__name__:
            setattr(ops, name, wrap(name, fn))

    model.train().requires_grad_(True)
    g = torch.Generator(device=dev).manual_seed(7)
    batch = (torch.randn(bsz, 16, 1, 60, 104, device=dev, generator=g), torch.randn(bsz, 512, 4096, device=dev, generator=g),
             torch.randn(bsz, 16, 1, 60, 104, device=dev, generator=g))
    opt = optim.AdamW(model.parameters(), lr=5e-6)

    def step():
        trainer.forward_backward(batch, model)
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    on[0] = True
    nsteps = 3
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(nsteps):
        step()
    t1.record()
    torch.cuda.synchronize()
    rows = []
    for (name, ints, shp, extra), evs in rec.items():
        tot = sum(s.elapsed_time(e) for s, e in evs)
        rows.append({"op": name, "ints": ints, "shapes": shp, "kw": extra, "calls_per_step": len(evs) / nsteps,
                     "ms_per_step": tot / nsteps, "avg_us": 1e3 * tot / len(evs)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    by_op = defaultdict(float)
    for r in rows:
        by_op[r["op"]] += r["ms_per_step"]
    print(json.dumps({"batch": bsz, "mode": mode, "wall_ms_per_step_with_events": t0.elapsed_time(t1) / nsteps,
                      "sum_ms_per_step": sum(r["ms_per_step"] for r in rows),
                      "by_op": dict(sorted(by_op.items(), key=lambda kv: -kv[1])), "rows": rows[:120]}, indent=0))


if __name__ == "__main__":
    main()
