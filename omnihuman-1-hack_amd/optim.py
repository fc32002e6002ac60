"""Optimizer side of the training step on gfx950 kernels: AdamW with the
constructor of ``torch.optim.AdamW`` (seaweed_apt/distilled_trainer.py:69-75)
and the EMA update of distilled_trainer.py:319-334 kept on the GPU."""
import os
import weakref

import torch

from . import ops

# OMH_ADAMW_PACK=0: the bf16 operand copies of the training step in their own launch after the step (round 3), not
# written by the optimizer kernel (A/B timing)
try:
    from .wan.modules.model_train import pack_entry_of
except Exception:  # pragma: no cover
    pack_entry_of = None


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
            marks = []                                      # (TrainPacks, row): copies this step writes itself
            fuse = os.environ.get("OMH_ADAMW_PACK", "1") != "0" and pack_entry_of is not None
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    # model_train.pending_step_bytes stops counting them while they are alive (weak: a deleted optimizer
                    # frees them again)
                    p._omh_moments_allocated = weakref.ref(st["exp_avg"])
                # a state dict saved by torch.optim.AdamW (the 'optimizer' entry of the reference's checkpoints,
                # distilled_trainer.py:153-178) holds the step as a 0-d tensor: normalise to a Python int
                st["step"] = int(st["step"]) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(g)
                touched.append(p)
                assert p.dtype == torch.float32 and p.is_contiguous() and g.dtype == torch.float32
                row = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                ent = pack_entry_of(p) if fuse else None
                if ent is not None:                             # this weight has bf16 operand copies for the training step
                    packs, ri = ent
                    pr = packs.rows[ri]                          # [src, dst, dstT, rows, cols, ld_dst, ld_t, tile0, kind]
                    row = row + (pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], 0 if pr[8] == 0 else 1)
                    marks.append((packs, ri))
                by_step.setdefault((st["step"], p.device), []).append(row)
            # one multi-tensor launch per (step count, device): ~750 parameter tensors -> 1 kernel
            for (step, dev), rows in by_step.items():
                # the pointer table is re-uploaded only when an address changed (never under a graphed step,
                # whose gradients live at fixed addresses): no host-to-device copy in the steady state
                cache = self.__dict__.setdefault("_tables", {})
                # keyed per parameter group and device; parameters that joined later (a different step count) get
                # their own table instead of evicting the main one every step
                if any(len(r) > 5 for r in rows):
                    # AdamW + the bf16 operand copies in one pass (omh_adamw_pack_multi): 12-column table with the first
                    # tile of every entry (64 x 64 tiles for weights with copies, 4096-element chunks otherwise)
                    key = (gi, dev, len(rows), "pack")
                    ent = cache.get(key)
                    if ent is None or ent[0] != rows:
                        full, tile0 = [], 0
                        for r in rows:
                            if len(r) > 5 and r[11] == 0:
                                e_ = [r[0], r[1], r[2], r[3], r[5], r[6], r[7], r[8], r[9], r[10], tile0, 0]
                                tile0 += ((r[7] + 63) // 64) * ((r[8] + 63) // 64)
                            elif len(r) > 5:
                                e_ = [r[0], r[1], r[2], r[3], r[5], 0, 1, r[4], 0, 0, tile0, 1]
                                tile0 += (r[4] + 4095) // 4096
                            else:
                                e_ = [r[0], r[1], r[2], r[3], 0, 0, 1, r[4], 0, 0, tile0, 2]
                                tile0 += (r[4] + 4095) // 4096
                            full.append(e_)
                        ent = cache[key] = (rows, torch.tensor(full, dtype=torch.int64).to(dev, non_blocking=False), tile0)
                    ops.adamw_pack_multi(ent[1], len(rows), ent[2], group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                         step, grad_scale)
                    continue
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
            by_packs = {}
            for packs, ri in marks:                         # ... and these copies are already those of the new version
                by_packs.setdefault(id(packs), (packs, []))[1].append(ri)
            for packs, idx in by_packs.values():
                packs.mark_current(idx)
        return loss


_EMA_TABLES = {}


@torch.no_grad()
def update_ema_model(ema_model, model, decay):
    """distilled_trainer.py:319-334 without the GPU->CPU round trip (the EMA copy lives in HBM), and as ONE launch for all
    parameters (omh_ema_update_multi) instead of one per tensor; the pointer table is rebuilt only when an address or a
    size changes."""
    rows, touched, keep = [], [], []
    dev = None
    for target, source in zip(ema_model.parameters(), model.parameters()):
        src = source.data
        if src.device != target.device or src.dtype != torch.float32 or not src.is_contiguous():
            src = src.to(device=target.device, dtype=torch.float32).contiguous()
            keep.append(src)
        if target.dtype != torch.float32 or not target.is_contiguous() or target.numel() == 0:
            if target.numel():
                ops.ema_update(target.data, src, decay)             # an odd one out: its own launch
                touched.append(target)
            continue
        dev = target.device
        rows.append((target.data_ptr(), src.data_ptr(), target.numel()))
        touched.append(target)
    if rows:
        key = (id(ema_model), id(model))
        ent = _EMA_TABLES.get(key)
        if ent is None or ent[0] != rows:
            full, chunk0 = [], 0
            for t_, s_, n_ in rows:
                full.append([t_, s_, n_, chunk0])
                chunk0 += (n_ + 4095) // 4096
            ent = _EMA_TABLES[key] = (rows, torch.tensor(full, dtype=torch.int64).to(dev), chunk0)
            if len(_EMA_TABLES) > 8:
                _EMA_TABLES.pop(next(iter(_EMA_TABLES)))
        ops.ema_update_multi(ent[1], len(rows), ent[2], decay)
    if touched:
        torch.autograd.graph.increment_version(touched)
