"""hipGraph capture of the launch-bound regimes of the path.

At S = 1 560 tokens (one latent frame: BASELINE configs 1 and 3) a DiT forward is ~600 short launches and a
training step ~4 600 per clip; issued one by one through ctypes the host, not the MI355X, sets the pace.  Both
regimes have a fixed launch schedule for a fixed geometry, so the schedule is recorded once into a hipGraph
(``torch.cuda.CUDAGraph`` = hipStreamBeginCapture on the stream every libomh.so entry point is handed) and
replayed with one host call per step:

* ``GraphedForward``       — ``WanModel.forward`` for inference (generate.py:205-229 v_teacher path, config 1);
* ``GraphedTrainingStep``  — forward + loss + backward of distilled_trainer.py:241-316 (config 3); the gradient
  all-reduce (RCCL) and AdamW stay outside the graph: AdamW's bias correction depends on the host step count.

What capture requires of the path, and how it is met: no host<->device copies (sequence-length tables are cached
device constants, model.py:_dev_ints; the loss is returned as a device scalar), no allocation outside torch's
graph-private pool (every buffer comes from torch), and weight packing as graph nodes in training
(``_Packed.always_rebuild``) because a replay cannot check parameter versions on the host.

Replays read the *static* input buffers: ``__call__`` copies the caller's tensors into them first.
"""
import contextlib
from typing import List, Optional, Sequence

import torch

from .wan.modules import model as _model


@contextlib.contextmanager
def _repacking(on: bool):
    old = _model._Packed.always_rebuild
    _model._Packed.always_rebuild = bool(on)
    try:
        yield
    finally:
        _model._Packed.always_rebuild = old


def _warm(fn, n):
    """Eager runs on a side stream before capture (PyTorch's capture protocol): lazy one-time work —
    hipFuncSetAttribute, constant tables, the allocator's first touches — must not fall inside the capture."""
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(n):
            fn()
    cur.wait_stream(side)
    torch.cuda.synchronize()


class GraphedForward:
    """One inference forward of ``model`` at a fixed geometry as a hipGraph.

    ``context`` is a list of [L, text_dim] tensors or a ``ContextState`` from ``model.encode_context`` (then the
    cross-attention K/V are constants of the graph).  ``__call__(x, t)`` returns the list of *static* output
    tensors (overwritten by the next call: clone what must survive)."""

    def __init__(self, model, x: Sequence[torch.Tensor], t: torch.Tensor, context, seq_len: int, clip_fea=None,
                 y=None, warmup: int = 2):
        dev = next(model.parameters()).device
        self.model, self.seq_len = model, seq_len
        self.x = [u.to(dev, torch.float32).clone() for u in x]
        self.t = t.to(dev).clone()
        self.context = context if isinstance(context, _model.ContextState) else [c.to(dev).clone() for c in context]
        self.clip_fea = None if clip_fea is None else clip_fea.to(dev).clone()
        self.y = None if y is None else [v.to(dev).clone() for v in y]

        def run():
            with torch.no_grad():
                return model._forward_infer(self.x, self.t, self.context, seq_len, self.clip_fea, self.y)

        _warm(run, warmup)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = run()

    def __call__(self, x: Sequence[torch.Tensor], t: torch.Tensor) -> List[torch.Tensor]:
        for dst, src in zip(self.x, x):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.t.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.out


class GraphedTrainingStep:
    """forward + loss + backward of ``trainer.forward_backward`` as one hipGraph; gradient all-reduce and the
    optimizer outside it.

        step = GraphedTrainingStep(model, example_batch, optimizer=opt, reducer=red)
        loss = step(batch)            # 0-d device tensor (un-divided loss); float(loss) to read it

    The parameters' ``.grad`` tensors are allocated by the captured backward and *rewritten* by every replay, so
    they are never zeroed or set to None between steps (``optimizer.zero_grad`` must not be called).  With a
    ``reducer`` (parallel.BucketedGradAllReduce) its per-gradient hooks are silenced during capture/replay and
    every bucket is all-reduced after the replay.  The reducer's ``finish()`` leaves every ``p.grad`` pointing at
    a slice of its reduced flat bucket, while a replay keeps writing to the captured addresses — so ``__call__``
    re-points ``p.grad`` at the captured tensors after every replay, and the reducer packs them afresh.

    A replay OVERWRITES the gradients (the captured backward starts from ``grad = None``), so micro-batch
    accumulation (distilled_trainer.py:289-301 with ``--gradient_accumulation_steps`` > 1) cannot be expressed by
    replaying this graph several times: it is refused here; use ``trainer.training_step`` eagerly (with
    ``reducer.no_sync()`` around the non-final micro-steps) for that."""

    def __init__(self, model, example_batch, optimizer=None, reducer=None, num_train_timesteps: int = 1000,
                 gradient_accumulation_steps: int = 1, loss_scale: float = 1.0, reference_loss_quirk: bool = True,
                 warmup: int = 1):
        from . import trainer
        if gradient_accumulation_steps != 1:
            raise ValueError("GraphedTrainingStep replays overwrite .grad and step the optimizer on every call; "
                             "gradient_accumulation_steps must be 1 (accumulate with the eager training_step)")
        dev = next(model.parameters()).device
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        self.accum = gradient_accumulation_steps
        self.batch = tuple(b.to(dev).clone() for b in example_batch)
        kw = dict(num_train_timesteps=num_train_timesteps, gradient_accumulation_steps=gradient_accumulation_steps,
                  loss_scale=loss_scale, reference_loss_quirk=reference_loss_quirk)

        def fb():
            return trainer.forward_backward(self.batch, model, **kw)

        def warm_once():
            fb()
            for p in model.parameters():
                p.grad = None

        sync = reducer.no_sync() if reducer is not None else contextlib.nullcontext()
        with sync, _repacking(True):
            _warm(warm_once, warmup)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = fb()
        # the captured backward left its results in .grad; keep the tensors alive at these addresses
        self.params = list(model.parameters())
        self.grads = [p.grad for p in self.params]

    def __call__(self, batch) -> torch.Tensor:
        for dst, src in zip(self.batch, batch):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # the replay wrote to the captured gradient tensors; a previous reducer.finish() may have left p.grad
        # pointing at a slice of its flat bucket (parallel.py) — that would be last step's data
        for p, g in zip(self.params, self.grads):
            p.grad = g
        if self.reducer is not None:
            self.reducer.finish()              # nothing was launched by hooks: all buckets go now
        if self.optimizer is not None:
            self.optimizer.step()
        return self.loss * self.accum
