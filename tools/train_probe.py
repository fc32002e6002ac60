"""Eager vs hipGraph replay of the launch-bound regimes on the 1.3B model (run on the GPU box):
    python tools/train_probe.py [batch ...]
Prints ms per training step (fwd + recompute + bwd + AdamW) eager and graphed, and ms per S=1560 forward."""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PKG = "omnihuman-1-hack_amd"


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def main():
    dev = torch.device("cuda", 0)
    graphs = importlib.import_module(PKG + ".graphs")
    trainer = importlib.import_module(PKG + ".trainer")
    optim = importlib.import_module(PKG + ".optim")
    model = bench.build_model(dev)
    res = {}
    # ---- inference forward at S = 1560 (config 1)
    g = torch.Generator(device=dev).manual_seed(1)
    x = [torch.randn(16, 1, 60, 104, device=dev, generator=g)]
    t = torch.tensor([999.0], device=dev)
    ctx = [torch.randn(120, 4096, device=dev, generator=g)]
    st = model.encode_context(ctx)
    for _ in range(2):
        model(x, t, st, 1560)
    res["fwd_eager_ms"] = timeit(lambda: model(x, t, st, 1560), 10)
    t0 = time.perf_counter()
    gf = graphs.GraphedForward(model, x, t, st, 1560)
    res["fwd_capture_s"] = time.perf_counter() - t0
    res["fwd_graph_ms"] = timeit(lambda: gf(x, t), 10)
    want = model(x, t, st, 1560)[0]
    res["fwd_equal"] = bool(torch.equal(gf(x, t)[0], want))
    del gf
    # ---- training step
    for bsz in [int(a) for a in sys.argv[1:]] or [1, 4]:
        model.train().requires_grad_(True)
        for p in model.parameters():
            p.grad = None
        batch = (torch.randn(bsz, 16, 1, 60, 104, device=dev, generator=g),
                 torch.randn(bsz, 512, 4096, device=dev, generator=g),
                 torch.randn(bsz, 16, 1, 60, 104, device=dev, generator=g))
        opt = optim.AdamW(model.parameters(), lr=5e-6)

        def eager():
            trainer.forward_backward(batch, model)
            opt.step()
            opt.zero_grad(set_to_none=True)
        for _ in range(2):
            eager()
        res[f"train_b{bsz}_eager_ms"] = timeit(eager, 4)
        t0 = time.perf_counter()
        step = graphs.GraphedTrainingStep(model, batch, optimizer=opt)
        res[f"train_b{bsz}_capture_s"] = time.perf_counter() - t0
        step(batch)
        res[f"train_b{bsz}_graph_ms"] = timeit(lambda: step(batch), 4)
        res[f"train_b{bsz}_loss"] = float(step(batch))
        del step, opt
        model.eval().requires_grad_(False)
        torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
