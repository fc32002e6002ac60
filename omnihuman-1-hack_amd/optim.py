"""Optimizer side of the training step on gfx950 kernels: AdamW with the
constructor of ``torch.optim.AdamW`` (seaweed_apt/distilled_trainer.py:69-75)
and the EMA update of distilled_trainer.py:319-334 kept on the GPU."""
import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW``-compatible (lr, betas, eps, weight_decay); one fused kernel per parameter."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            by_step = {}
            keep = []                                       # keeps .contiguous() copies alive until the launch
            touched = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                # a state dict saved by torch.optim.AdamW (the 'optimizer' entry of the reference's checkpoints,
                # distilled_trainer.py:153-178) holds the step as a 0-d tensor: normalise to a Python int
                st["step"] = int(st["step"]) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(g)
                touched.append(p)
                assert p.dtype == torch.float32 and p.is_contiguous() and g.dtype == torch.float32
                by_step.setdefault((st["step"], p.device), []).append(
                    (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()))
            # one multi-tensor launch per (step count, device): ~750 parameter tensors -> 1 kernel
            for (step, dev), rows in by_step.items():
                # the pointer table is re-uploaded only when an address changed (never under a graphed step,
                # whose gradients live at fixed addresses): no host-to-device copy in the steady state
                cache = self.__dict__.setdefault("_tables", {})
                # keyed per parameter group and device; parameters that joined later (a different step count) get
                # their own table instead of evicting the main one every step
                key = (gi, dev, len(rows))
                ent = cache.get(key)
                if ent is None or ent[0] != rows:
                    ent = cache[key] = (rows, torch.tensor(rows, dtype=torch.int64).to(dev, non_blocking=False))
                table = ent[1]
                ops.adamw_multi(table, len(rows), group["lr"], b1, b2, group["eps"], group["weight_decay"], step,
                                grad_scale)
            # the kernel writes through raw pointers: tell autograd (and the packed bf16 weight copies keyed on
            # ``_version``, model.py:_Packed) that these parameters changed
            if touched:
                torch.autograd.graph.increment_version(touched)
        return loss


@torch.no_grad()
def update_ema_model(ema_model, model, decay):
    """distilled_trainer.py:319-334 without the GPU->CPU round trip (the EMA copy lives in HBM)."""
    for target, source in zip(ema_model.parameters(), model.parameters()):
        ops.ema_update(target.data, source.data.to(target.device), decay)
        torch.autograd.graph.increment_version(target)
