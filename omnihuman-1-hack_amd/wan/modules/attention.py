"""``flash_attention`` with the reference's signature
(seaweed_apt/wan/modules/attention.py:24-130) on the gfx950 kernel
``omh_flash_attn_fwd_d128``.

The reference packs variable-length sequences and calls flash-attn's varlen
kernel; the result is softmax(q k^T * scale) v per head over the first
``k_lens[b]`` keys of each sample, for every one of the first ``q_lens[b]``
query rows (rows past it are zero; ``q_lens`` is never passed by model.py).  This wrapper keeps that contract
for head_dim 128, including flash-attn's ``causal`` / ``window_size`` band (bottom-right aligned: query i sees key j
iff i + klen - qlen - left <= j <= i + klen - qlen + right; rows with an empty band are zero) — forward only, on the
short-sequence kernel; ``dropout_p`` > 0 is rejected (a random mask has no parity to hold).  The DiT blocks do not go through it (they hand the kernel
pre-laid-out q / k / V^T buffers); it exists for callers of the reference API.
"""
import torch

from .._backend import ops

__all__ = ["flash_attention", "attention"]


def flash_attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None,
                    causal=False, window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, version=None):
    """q [B, Lq, N, 128], k/v [B, Lk, N, 128]; returns [B, Lq, N, 128] in q's dtype."""
    assert dtype in (torch.float16, torch.bfloat16)
    assert q.device.type == "cuda" and q.size(-1) <= 256
    if dropout_p != 0.:
        # the reference forwards it to flash-attn (attention.py:96-127); no caller in the repository sets it
        # (model.py:151-156,181,221-223) and a random mask cannot be held to parity: rejected, not silently ignored
        raise NotImplementedError("flash_attention on gfx950: dropout_p > 0 is not built (no caller in the reference "
                                  "uses it); q_lens, k_lens, causal and window_size are")
    # flash-attn's window (attention.py:121-126): causal bounds the right side at 0 (flash_attn_varlen_func sets
    # window_size = (left, 0) for causal=True); a negative side is unbounded
    wl, wr = int(window_size[0]), int(window_size[1])
    if causal:
        wr = 0
    window = (wl if wl >= 0 else -1, wr if wr >= 0 else -1)
    if window != (-1, -1) and torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        raise NotImplementedError("flash_attention on gfx950: causal / window_size are forward-only (the attention "
                                  "backward streams serve full attention, the only form model.py uses)")
    B, Lq, N, D = q.shape
    Lk = k.shape[1]
    if D != 128:
        raise NotImplementedError("the gfx950 attention kernel is built for head_dim 128")
    out_dtype = q.dtype
    if q_scale is not None:
        q = q * q_scale
    qb = q.to(torch.bfloat16).contiguous()
    kb = k.to(torch.bfloat16).contiguous()
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros(B, N * D, Lp, dtype=torch.bfloat16, device=q.device)
    vt[:, :, :Lk] = v.to(torch.bfloat16).reshape(B, Lk, N * D).transpose(1, 2)   # layout change only
    kl = None if k_lens is None else k_lens.to(device=q.device, dtype=torch.int32).contiguous()
    # q_lens (attention.py:55-60,79): the reference cuts the queries past q_lens[b] out of the packed batch — and can only
    # un-flatten the result when every q_lens[b] == Lq (attention.py:110); here those rows come back as zeros
    ql = None if q_lens is None else q_lens.to(device=q.device, dtype=torch.int32).contiguous()
    o = ops.flash_attn(qb, kb, vt, kl, scale=softmax_scale, q_lens=ql, window=window)
    return o.type(out_dtype)


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
              window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, fa_version=None):
    """attention.py:133-179 — same kernel; the reference's SDPA fallback is not needed here."""
    return flash_attention(q, k, v, q_lens, k_lens, dropout_p, softmax_scale, q_scale, causal, window_size,
                           deterministic, dtype, fa_version)
