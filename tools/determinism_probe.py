"""Run-to-run determinism of the S=1560 forward (eager twice, graph twice, eager vs graph) and, if they differ,
the first block whose output differs.  GPU box only."""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PKG = "omnihuman-1-hack_amd"


def main():
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    x = [torch.randn(16, 1, 60, 104, device=dev, generator=g)]
    t = torch.tensor([999.0], device=dev)
    ctx = [torch.randn(120, 4096, device=dev, generator=g)]
    st = model.encode_context(ctx)
    res = {}
    a = model(x, t, st, 1560)[0].clone()
    b = model(x, t, st, 1560)[0].clone()
    res["eager_vs_eager_maxabs"] = float((a - b).abs().max())
    junk = [torch.full((1 << 26,), float("nan"), device=dev) for _ in range(8)]    # poison the allocator's free blocks
    del junk
    c = model(x, t, st, 1560)[0].clone()
    res["eager_after_nan_poison_maxabs"] = float((a - c).abs().max())
    res["eager_after_nan_poison_finite"] = bool(torch.isfinite(c).all())
    res["out_absmax"] = float(a.abs().max())
    # per-block hook comparison: two eager runs
    outs = [[], []]
    for run in range(2):
        hs = [blk.register_forward_hook(lambda m, i, o, run=run: outs[run].append(o.clone())) for blk in model.blocks]
        model(x, t, st, 1560)
        for h in hs:
            h.remove()
    diffs = [float((p - q).abs().max()) for p, q in zip(*outs)]
    res["first_block_differing_between_eager_runs"] = next((i for i, d in enumerate(diffs) if d > 0), None)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
