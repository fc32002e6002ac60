"""Data-parallel gradient synchronisation for the training step (BASELINE
config 3): one process per GPU, ``torch.distributed`` backend "nccl" (= RCCL
over xGMI) — "gloo" on CPU for the tests.

The reference relies on ``accelerate`` wrapping the model in DDP
(distilled_trainer.py:79-81).  This reducer does the same job with the knobs
that matter on a point-to-point xGMI mesh exposed: gradients are packed into
large flat buckets (default 256 MB — few, large collectives; 288 GB of HBM makes
the staging copy free) in reverse parameter order, and each bucket's
all-reduce is launched asynchronously the moment its last gradient has been
accumulated, so communication overlaps with the rest of the backward (the
block nodes of model_train.py release their gradients block by block).  A bucket is
packed with one multi-tensor copy and never unpacked: after ``finish()`` every
``p.grad`` is a view of the reduced flat buffer.
Parameters that receive no gradient (the reference's frozen FFNs of blocks
> 10) are handled without ``find_unused_parameters``: every
rank skips the same set.  With gradient accumulation, wrap the non-final
micro-steps in ``no_sync()``.
"""
import contextlib
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Done:
    """A completed piece of work (the host-staged validation path)."""

    def wait(self):
        return True


class BucketedGradAllReduce:
    """``collective``: "all_reduce" (default) or "reduce_scatter_all_gather" — the same reduction issued as its two
    halves per bucket (each rank reduces 1/world of the bucket, then the shards are gathered): on the point-to-point xGMI
    mesh both halves are direct exchanges that use all 7 links at once, and the split is the hook a sharded optimizer
    step would sit between.  ``payload``: torch.float32 (default) or torch.bfloat16 — the bucket is cast to bf16 for the
    wire (half the bytes per link; one 2^-9 rounding of each rank's gradient and of the sum, the precision of the
    operands the gradients were computed from) and widened back to fp32 for the optimizer.  Environment overrides:
    OMH_GRAD_COLLECTIVE = all_reduce | rs_ag, OMH_GRAD_PAYLOAD = fp32 | bf16.  All four combinations give the gradients
    of the fp32 all-reduce (exactly for fp32 payloads on two ranks, to bf16 rounding otherwise):
    tests/test_parallel_gloo.py."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 256.0,
                 group: Optional["dist.ProcessGroup"] = None, average: bool = True, force: bool = False,
                 collective: Optional[str] = None, payload: Optional[torch.dtype] = None):
        import os
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()        # run the collectives even on a 1-rank group (tests)
        self.average = average
        collective = collective or os.environ.get("OMH_GRAD_COLLECTIVE", "all_reduce")
        collective = {"rs_ag": "reduce_scatter_all_gather"}.get(collective, collective)
        if collective not in ("all_reduce", "reduce_scatter_all_gather"):
            raise ValueError(f"collective must be 'all_reduce' or 'reduce_scatter_all_gather', got {collective!r}")
        if payload is None:
            payload = {"fp32": torch.float32, "bf16": torch.bfloat16}[os.environ.get("OMH_GRAD_PAYLOAD", "fp32")]
        if payload not in (torch.float32, torch.bfloat16):
            raise ValueError("payload must be torch.float32 or torch.bfloat16")
        self.collective, self.payload = collective, payload
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.enabled = True
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets, cur, cur_bytes = [], [], 0
        for p in reversed(self.params):                     # gradients become ready roughly in reverse order
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > cap:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._flat = [None] * len(self.buckets)
        # Layout of a bucket = the parameters that got a gradient the last time it was launched, in bucket order (the
        # frozen FFNs never do: they take no room on the wire).  Once known, ``grad_slot`` hands the training step the
        # parameter's slice as the DESTINATION of its weight gradient (model_train._grad_slots), so the next launch finds
        # the gradient already in place and packs nothing (the 3.6 GB packing copy).
        self._layout = [None] * len(self.buckets)           # per bucket: {id(p): (offset, numel)} or None
        self._wire = [None] * len(self.buckets)             # bf16 payload: the buffer that travels
        self._shard = [None] * len(self.buckets)            # reduce_scatter_all_gather: this rank's reduced 1/world
        self.bytes_on_wire = 0                              # payload bytes handed to the collectives in the last step
        self._reset()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _reset(self):
        self._ready = [set() for _ in self.buckets]
        self._work = [None] * len(self.buckets)

    @contextlib.contextmanager
    def no_sync(self):
        """Skip the all-reduce (gradient accumulation micro-steps, distilled_trainer.py:372)."""
        old, self.enabled = self.enabled, False
        try:
            yield
        finally:
            self.enabled = old

    def grad_slot(self, p):
        """fp32 view (the parameter's shape) of ``p``'s slice of its bucket, or None while the bucket's layout is not
        known yet (first step), the parameter was not part of it, or the slice would not be what the next launch
        expects.  A gradient written there needs no packing.  NOTE: such a ``p.grad`` is a view of the persistent bucket —
        valid until the next backward pass writes the bucket again (copy it to keep a step's gradient for logging)."""
        if self.world == 1 and not self.force:
            return None
        i = self._bucket_of.get(id(p))
        if i is None or self._work[i] is not None:
            return None
        lay, flat = self._layout[i], self._flat[i]
        if lay is None or flat is None or id(p) not in lay or flat.dtype != torch.float32 or flat.device != p.device:
            return None
        off, n = lay[id(p)]
        if n != p.numel():
            return None
        return flat[off:off + n].view(p.shape)

    def _on_grad(self, p):
        if not self.enabled or (self.world == 1 and not self.force):
            return
        i = self._bucket_of[id(p)]
        if self._work[i] is not None:
            # Already launched in this pass.  The engine runs a parameter's post-accumulate hooks again after the block
            # node has told the reducer itself (torch 2.10 fires them for an undefined gradient too — measured,
            # tools/dbg_train.py), so a repeat is normal and carries no new contribution.  What must never happen is a
            # SECOND contribution landing in a bucket under its collective: the block nodes therefore take neither a
            # bucket slot (model_train._grad_slots) nor the in-place route (_grad_targets) while another forward of the
            # model awaits its backward — autograd then sums the contributions BEFORE this hook fires.
            return
        self._ready[i].add(id(p))
        if len(self._ready[i]) == len(self.buckets[i]):
            self._launch(i)

    _omh_joins_side_streams = True      # model_train._may_defer_join: this hook orders itself behind the weight-gradient stream

    def _launch(self, i):
        bucket = [p for p in self.buckets[i] if p.grad is not None]
        if not bucket:
            self._work[i] = (None, [], [], None)
            return
        if not any(self._work):
            self.bytes_on_wire = 0
            self.packed_elements = 0                        # elements copied into buckets this step (0: all in place)
        # The weight gradients of the block that just returned may still be in flight on the training step's second
        # stream (model_train: its join is deferred to the end of the backward pass).  The bucket is therefore packed and
        # its collective queued FROM that stream — behind the gradients it reads, and behind what the main stream has
        # produced so far — so the main stream never waits here; finish() makes it wait for the collectives.
        side = None
        if bucket[0].grad.is_cuda:
            from .wan.modules.model_train import reducer_stream
            side = reducer_stream(bucket[0].grad.device)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            self._pack_and_launch(i, bucket)

    def _pack_and_launch(self, i, bucket):
        n = sum(p.numel() for p in bucket)
        # the reduce-scatter halves split the bucket evenly: pad to a multiple of 8 elements per rank
        n_pad = -(-n // (8 * self.world)) * (8 * self.world) if self.collective != "all_reduce" else n
        dev, gdt = bucket[0].grad.device, bucket[0].grad.dtype
        flat = self._flat[i]
        if flat is None or flat.numel() != n_pad or flat.device != dev:
            flat = self._flat[i] = torch.zeros(n_pad, dtype=gdt, device=dev)
        wire = flat
        if self.payload != gdt:
            wire = self._wire[i]
            if wire is None or wire.numel() != n_pad or wire.device != dev:
                wire = self._wire[i] = torch.zeros(n_pad, dtype=self.payload, device=dev)
        # pack with ONE multi-tensor copy (not a launch per parameter: ~800 of them cost 10 % of a training step);
        # a gradient that already lives in its slice (accumulation into the view finish() left behind) is skipped.
        # With a bf16 payload the same copy is the cast onto the wire buffer.
        sizes = [p.numel() for p in bucket]
        views = list(flat[:n].split(sizes))
        wviews = views if wire is flat else list(wire[:n].split(sizes))
        # a gradient that lives inside this flat buffer but NOT at its slice (the set of parameters with a gradient
        # changed since the layout was recorded): copying it to its new place could overwrite a neighbour first
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        moved = [p for v, p in zip(views, bucket)
                 if lo <= p.grad.data_ptr() < hi and p.grad.data_ptr() != v.data_ptr()]
        for p in moved:
            p.grad = p.grad.clone()
        dst, src = [], []
        for v, wv, p in zip(views, wviews, bucket):
            g = p.grad.reshape(-1)
            if wire is not flat or g.data_ptr() != v.data_ptr():
                dst.append(wv)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        self.packed_elements = getattr(self, "packed_elements", 0) + sum(t.numel() for t in src)
        off, lay = 0, {}
        for p, sz in zip(bucket, sizes):
            lay[id(p)] = (off, sz)
            off += sz
        self._layout[i] = lay
        # RCCL averages in the collective; gloo (CPU tests) has no AVG, so sum now and scale in finish()
        backend = dist.get_backend(self.group)
        self._avg_in_coll = self.average and backend == "nccl"
        op = dist.ReduceOp.AVG if self._avg_in_coll else dist.ReduceOp.SUM
        self.bytes_on_wire += wire.numel() * wire.element_size()
        if backend == "gloo" and wire.is_cuda and not (self.collective == "all_reduce" and wire.dtype == torch.float32):
            # validation mode only (ranks sharing one GPU over gloo, tests/test_gpu_bench.py): gloo's device support
            # covers the fp32 all-reduce; the other variants go through the host, synchronously
            host = wire.cpu()
            if self.collective == "all_reduce":
                dist.all_reduce(host, op=op, group=self.group)
            else:
                sh = torch.empty(n_pad // self.world, dtype=host.dtype)
                dist.reduce_scatter_tensor(sh, host, op=op, group=self.group)
                dist.all_gather_into_tensor(host, sh, group=self.group)
            wire.copy_(host)
            self._work[i] = (_Done(), bucket, views, None)
            return
        if self.collective == "all_reduce":
            work = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
            self._work[i] = (work, bucket, views, None)
            return
        per = n_pad // self.world
        shard = self._shard[i]
        if shard is None or shard.numel() != per or shard.dtype != wire.dtype or shard.device != dev:
            shard = self._shard[i] = torch.empty(per, dtype=wire.dtype, device=dev)
        work = dist.reduce_scatter_tensor(shard, wire, op=op, group=self.group, async_op=True)
        gather = lambda: dist.all_gather_into_tensor(wire, shard, group=self.group, async_op=True)
        if backend == "nccl":
            # RCCL runs a group's collectives in issue order on its own stream: the gather can be queued right away
            self._work[i] = (gather(), bucket, views, None)
        else:
            self._work[i] = (work, bucket, views, gather)    # gloo: queue the gather once the scatter has completed

    def finish(self):
        """Call after ``backward()``: launches buckets that never filled (unused parameters), waits for
        all collectives and points every ``p.grad`` at its slice of the reduced flat buffer (no copy back: the
        optimizer reads the averaged gradients where the collective left them)."""
        if not self.enabled or (self.world == 1 and not self.force):
            self._reset()
            return
        for i in range(len(self.buckets)):
            if self._work[i] is None:
                self._launch(i)
        if any(w is not None and w[1] and w[1][0].grad is not None and w[1][0].grad.is_cuda for w in self._work):
            from .wan.modules.model_train import join_side_streams
            join_side_streams()                             # the packs (and host-staged copies) queued on the second stream
        for i, (work, bucket, views, then) in enumerate(self._work):
            if work is None:
                continue
            work.wait()
            if then is not None:
                then().wait()
            flat, wire = self._flat[i], self._wire[i] if self.payload != self._flat[i].dtype else self._flat[i]
            if wire is not flat:
                flat.copy_(wire)                            # widen the reduced payload back to the gradients' dtype
            if self.average and not self._avg_in_coll:
                flat.mul_(1.0 / self.world)
            for p, v in zip(bucket, views):
                p.grad = v.view_as(p)
        self._reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class CFGPairSplit:
    """One clip on TWO GPUs: the conditional and the unconditional forward of a denoising step are
    independent (text2video.py:238-241), so group rank 0 runs the conditional one, group rank 1 the unconditional
    one, and the two velocity predictions (one ``[16,T,H/8,W/8]`` fp32 latent each — 8.4 MB at 81 frames 480x832)
    are exchanged with ONE all-gather per step.  Both ranks then apply the same fused CFG + sampler update, so the
    latents stay bit-identical on both without a broadcast.  The all-gather is the only collective of the path.

    ``group`` must hold exactly two ranks (default: the world).  ``sync_seed`` makes the two ranks draw the same
    initial noise when the caller asked for a random seed."""

    def __init__(self, group: Optional["dist.ProcessGroup"] = None):
        if not dist.is_initialized():
            raise RuntimeError("CFGPairSplit needs an initialised torch.distributed process group")
        self.group = group
        if dist.get_world_size(group) != 2:
            raise ValueError(f"a CFG pair is split over exactly 2 ranks, the group has {dist.get_world_size(group)}")
        self.index = dist.get_rank(group)                  # 0: conditional branch, 1: unconditional branch
        self._buf = None

    @property
    def runs_conditional(self) -> bool:
        return self.index == 0

    def sync_seed(self, seed: int) -> int:
        box = [seed]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                   group=self.group)
        return int(box[0])

    def exchange(self, mine: torch.Tensor):
        """``mine``: this rank's prediction.  Returns ``(cond, uncond)``, identical on both ranks."""
        mine = mine.contiguous()
        if self._buf is None or self._buf.shape[1:] != mine.shape or self._buf.dtype != mine.dtype \
                or self._buf.device != mine.device:
            self._buf = torch.empty((2,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
        if mine.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo has no device all-gather: stage through the host (the 2-processes-on-one-GPU test; RCCL is the
            # production backend and gathers device to device)
            host = torch.empty(self._buf.shape, dtype=mine.dtype)
            dist.all_gather([host[0], host[1]], mine.cpu(), group=self.group)
            self._buf.copy_(host)
        else:
            dist.all_gather([self._buf[0], self._buf[1]], mine, group=self.group)
        return self._buf[0], self._buf[1]
