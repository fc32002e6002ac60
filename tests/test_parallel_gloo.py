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


def _variant_worker(rank, world, port, q):
    """Every (collective, payload) variant of the reducer against the fp32 all-reduce on the same gradients."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = _load_parallel()
        sizes = (1000, 37, 5001, 64, 303, 7)                     # bucket totals that are not multiples of 8 x world
        g = torch.Generator().manual_seed(100 + rank)            # different gradients on every rank
        grads = [torch.randn(n, generator=g) for n in sizes]

        def reduced(**kw):
            torch.manual_seed(0)
            params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
            unused = torch.nn.Parameter(torch.zeros(11))
            red = par.BucketedGradAllReduce(params + [unused], bucket_mb=0.012, **kw)
            out = []
            for _ in range(2):                                   # second step: the flat / wire / shard buffers are reused
                for p in params:
                    p.grad = None
                sum((p * d).sum() for p, d in zip(params, grads)).backward()
                red.finish()
                out.append([p.grad.clone() for p in params])
            assert unused.grad is None
            wire = red.bytes_on_wire
            red.remove()
            return out, wire

        base, wire32 = reduced()
        want = []
        for d in grads:                                           # the mean over ranks, computed independently
            t = d.clone()
            dist.all_reduce(t)
            want.append(t / world)
        for a, w in zip(base[0], want):
            assert torch.allclose(a, w, rtol=1e-6, atol=1e-6)
        for coll in ("all_reduce", "reduce_scatter_all_gather"):
            for pay in (torch.float32, torch.bfloat16):
                got, wire = reduced(collective=coll, payload=pay)
                for step in got:
                    for a, b in zip(step, base[0]):
                        if pay == torch.float32:
                            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (coll, pay)      # same sums, other order
                        else:                                     # each rank's gradient and the sum rounded to bf16
                            assert torch.allclose(a, b, rtol=2 ** -7, atol=2 ** -7 * float(b.abs().max())), (coll, pay)
                if pay == torch.bfloat16:
                    assert wire * 2 <= wire32 + 64 * world * len(sizes), (wire, wire32)   # half the bytes (+ padding)
        # ---- gradients written straight into the bucket slices (grad_slot: what model_train._grad_slots does for the
        # weight matrices of a block): nothing is packed for them, the reduced values are the same, and a step in which
        # the set of parameters with a gradient changes re-lays the bucket out without overwriting a neighbour
        for coll, pay in (("all_reduce", torch.float32), ("reduce_scatter_all_gather", torch.float32),
                          ("all_reduce", torch.bfloat16)):
            torch.manual_seed(0)
            params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
            red = par.BucketedGradAllReduce(params, bucket_mb=0.012, collective=coll, payload=pay)
            assert all(red.grad_slot(p) is None for p in params)             # no layout before the first launch
            for p, d in zip(params, grads):
                p.grad = None
            sum((p * d).sum() for p, d in zip(params, grads)).backward()
            red.finish()
            first = [p.grad.clone() for p in params]
            for p in params:
                p.grad = None
            direct = (0, 2, 4)                                               # these write into their slot, the rest arrive fresh
            for i, (p, d) in enumerate(zip(params, grads)):
                if i in direct:
                    slot = red.grad_slot(p)
                    assert slot is not None and slot.shape == p.shape
                    slot.copy_(d)
                    p.grad = slot
                else:
                    p.grad = d.clone()
            for p in reversed(params):
                red._on_grad(p)
            red.finish()
            assert red.packed_elements == (sum(sizes[i] for i in range(len(sizes)) if i not in direct)
                                           if pay == torch.float32 else sum(sizes))        # bf16: the cast IS the pack
            for a, b in zip([p.grad for p in params], first):
                assert torch.allclose(a, b, rtol=2 ** -6 if pay == torch.bfloat16 else 1e-6, atol=2 ** -6 * float(b.abs().max())
                                      if pay == torch.bfloat16 else 1e-7), (coll, pay)
            # the composition changes: parameter 0 gets no gradient this time, the others keep living in the flat buffer
            keep = {i: params[i].grad.clone() for i in range(1, len(sizes))}
            params[0].grad = None
            for i in range(1, len(sizes)):
                slot = red.grad_slot(params[i])
                if slot is not None:
                    slot.copy_(grads[i])
                    params[i].grad = slot
                else:
                    params[i].grad = grads[i].clone()
            for p in reversed(params[1:]):
                red._on_grad(p)
            red.finish()
            assert params[0].grad is None
            for i in range(1, len(sizes)):
                assert torch.allclose(params[i].grad, keep[i], rtol=2 ** -6 if pay == torch.bfloat16 else 1e-6,
                                      atol=2 ** -6 * float(keep[i].abs().max()) if pay == torch.bfloat16 else 1e-7), (coll, pay, i)
            red.remove()
        # the environment selects the same variants (what bench.py / a launcher would set)
        os.environ.update(OMH_GRAD_COLLECTIVE="rs_ag", OMH_GRAD_PAYLOAD="bf16")
        red = par.BucketedGradAllReduce([torch.nn.Parameter(torch.zeros(4))])
        assert red.collective == "reduce_scatter_all_gather" and red.payload == torch.bfloat16
        red.remove()
        for k in ("OMH_GRAD_COLLECTIVE", "OMH_GRAD_PAYLOAD"):
            os.environ.pop(k)
        with pytest.raises(ValueError):
            par.BucketedGradAllReduce([torch.nn.Parameter(torch.zeros(4))], collective="ring")
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()[-600:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reducer_variants_match_the_fp32_all_reduce(world):
    """reduce_scatter + all_gather per bucket and the bf16 payload (SURVEY.md section 5,
    "Distributed communication backend") against the plain fp32 all-reduce, world sizes 2 and 3 (odd: padded shards)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_variant_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


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
