"""CPU, world_size 2 over gloo: the data-parallel gradient reducer of the training step
(omnihuman-1-hack_amd/parallel.py) — bucketing, overlap hooks, unused parameters, no_sync — and the exchange step
of a CFG pair split over two ranks."""
import importlib.util
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_parallel():
    # parallel.py is pure torch.distributed: load it without importing the kernel package
    spec = importlib.util.spec_from_file_location("omh_parallel", os.path.join(ROOT, "omnihuman-1-hack_amd", "parallel.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = _load_parallel()
        torch.manual_seed(0)                                    # identical parameters on every rank
        params = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 37, 5000, 64, 300)]
        unused = torch.nn.Parameter(torch.randn(11))            # never receives a gradient (frozen-FFN case)
        red = par.BucketedGradAllReduce(params + [unused], bucket_mb=0.012)     # ~3000 floats per bucket
        assert len(red.buckets) >= 3
        data = [torch.full_like(p, float(rank + 1)) * (i + 1) for i, p in enumerate(params)]

        def step():
            for p in params:
                p.grad = None
            loss = sum((p * d).sum() for p, d in zip(params, data))
            loss.backward()
            red.finish()

        step()
        mean_scale = sum(range(1, world + 1)) / world           # grads are d_i * (rank+1) -> mean over ranks
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, torch.full_like(p, mean_scale * (i + 1))), (rank, i)
        assert unused.grad is None
        step()                                                   # second step reuses the flat buffers
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, torch.full_like(p, mean_scale * (i + 1)))
        # after finish() every p.grad is a view of its bucket's flat buffer (no copy back) ...
        flat_ptrs = {f.data_ptr(): f.numel() * f.element_size() for f in red._flat if f is not None}
        for p in params:
            assert any(b <= p.grad.data_ptr() < b + n for b, n in flat_ptrs.items())
        # ... and a backward that accumulates into those views (grads not reset) reduces correctly too
        sum((p * d).sum() for p, d in zip(params, data)).backward()
        red.finish()
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, torch.full_like(p, 2 * mean_scale * (i + 1)))
        with red.no_sync():                                      # accumulation micro-step: local gradients stay local
            for p in params:
                p.grad = None
            sum((p * d).sum() for p, d in zip(params, data)).backward()
            red.finish()
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, torch.full_like(p, float(rank + 1) * (i + 1)))
        red.remove()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_bucketed_grad_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _cfg_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = _load_parallel()
        if world != 2:
            with pytest.raises(ValueError):
                par.CFGPairSplit()
            q.put((rank, "ok"))
            return
        split = par.CFGPairSplit()
        assert split.runs_conditional == (rank == 0)
        assert split.sync_seed(1000 + rank) == 1000                      # both ranks draw rank 0's noise
        # a toy "denoising loop": each rank evaluates only its branch, both apply the same update
        x = torch.linspace(-1, 1, 16 * 2 * 6 * 8).reshape(16, 2, 6, 8)
        branch = (lambda v: torch.sin(v) * 0.5) if split.runs_conditional else (lambda v: torch.cos(v) * 0.25)
        ref = x.clone()
        for _ in range(3):
            cond, uncond = split.exchange(branch(x))
            x = x - 0.1 * (uncond + 4.0 * (cond - uncond))
            c, u = torch.sin(ref) * 0.5, torch.cos(ref) * 0.25
            ref = ref - 0.1 * (u + 4.0 * (c - u))
        assert torch.equal(x, ref)                                        # = the single-process loop, bit for bit
        cond, _ = split.exchange(torch.full((3,), float(rank)))           # a new shape re-allocates the buffer
        assert cond.shape == (3,) and float(cond[0]) == 0.0
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_cfg_pair_split_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cfg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
