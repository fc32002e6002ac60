"""Tensor-level wrappers over the libomh.so C ABI (include/omh.h).

PyTorch supplies device memory and the current HIP stream; every arithmetic
op below is one of the hand-written gfx950 kernels.  There is no CPU path:
a non-GPU tensor raises.
"""
import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (ATTN_ALLOW_SPLIT, ATTN_SHORT_KERNEL, BIAS_M, BIAS_N, BIAS_NONE, EPI_BF16, EPI_BF16_SPLIT_T, EPI_F32, EPI_F32_ACCUM, EPI_GELU_BF16,
                   EPI_GELU_BWD_BF16, EPI_GELU_ERF_BF16, EPI_RESID, AttnArgs, GemmArgs, OmhError, check, lib)

__all__ = ["gemm", "flash_attn", "layernorm_modulate", "rmsnorm_rope", "cast_bf16", "patchify", "unpatchify",
           "dense_f32", "sinusoidal_embedding", "cfg_unipc_step", "conv_cl", "rms_silu_cl", "nchw_to_cl", "cl_to_nchw",
           "softmax_rows", "OmhError",
           "EPI_BF16", "EPI_F32", "EPI_GELU_BF16", "EPI_GELU_ERF_BF16", "EPI_RESID", "EPI_F32_ACCUM", "BIAS_NONE", "BIAS_N", "BIAS_M"]


try:                                     # the raw handle of the current stream without building a torch.cuda.Stream
    _raw_stream, _cur_dev = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice      # object: 0.2 us against 3 us per
except AttributeError:                   # launch (~2 000 launches in a one-clip training step, whose backward is bound by
    _raw_stream = _cur_dev = None        # the host: tools/host_phase_probe.py)


def _stream():
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(_cur_dev()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise OmhError("omnihuman-1-hack_amd ops run on the MI355X only: got a CPU tensor "
                           "(there is no CPU fallback in the product path)")


def _p(t: Optional[torch.Tensor], byte_off: int = 0):
    return None if t is None else C.c_void_p(t.data_ptr() + byte_off)


def ptr(t: torch.Tensor, elem_off: int = 0):
    """Raw device pointer ``elem_off`` elements into ``t`` (for the *_raw ops)."""
    return C.c_void_p(t.data_ptr() + elem_off * t.element_size())


def gemm_raw(A, B, Cp, M, N, K, lda, ldb, ldc, epilogue, bias=None, bias_mode=BIAS_NONE, batch=1,
             strideA=0, strideB=0, strideC=0, gate0=None, gate1=None, gate1_stride=0, gate_rows=1,
             gate_const=0.0, b_kmajor=False, c_in=None, aux=None, ldaux=0, split_k=False, n_split=0):
    """C[m][n] = epi(sum_k A[m][k] B[n][k])  (b_kmajor: B[k][n], [K, N] row-major); all pointers are c_void_p.
    ``c_in`` / ``aux`` / ``ldaux``: the fused training epilogues of include/omh.h (ABI v5).  ``split_k``: hand the
    library a workspace so that it may cut a few-row, long contraction into slices (ABI v9: the FFN-down projection and
    the FFN-up input gradient at one or two clips; another fp32 summation order than the unsplit kernels, so only
    callers that do not need batch-invariant bits for that product ask for it).  ``n_split`` (EPI_BF16_SPLIT_T, ABI v10):
    columns from n_split on go transposed to ``aux`` (row pitch ``ldaux``) — the fused q | k | v projection."""
    a = GemmArgs(A, B, Cp, M, N, K, lda, ldb, ldc, batch, strideA, strideB, strideC, epilogue, bias_mode, bias,
                 gate0, gate1, gate1_stride, gate_rows, gate_const, int(b_kmajor), c_in, aux, ldaux, None, 0, int(n_split))
    ws = None
    # (the library's own rule again below; this is only to skip the query where it cannot say anything but 0)
    if split_k and K >= 4096 and batch == 1 and ((M + 255) // 256) * ((N + 191) // 192) <= 128:
        need = lib.omh_gemm_workspace_bytes(C.byref(a))      # few rows, long contraction: split K (include/omh.h, ABI v9)
        if need > 0:
            ws = torch.empty(need, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
            a.workspace, a.workspace_bytes = ws.data_ptr(), need
    check(lib.omh_gemm_bf16(C.byref(a), _stream()), "omh_gemm_bf16")


def gemm(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
         epilogue: int = EPI_BF16, b_kmajor: bool = False):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias[N]); a, w bf16 contiguous.  ``b_kmajor``: w is [K,N] and
    out = epi(a @ w + bias) — the input gradient of a Linear on the weight as stored."""
    _dev(a, w, out, bias)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.dim() == 2 and w.dim() == 2
    M, K = a.shape
    N = w.shape[1] if b_kmajor else w.shape[0]
    assert w.shape[0 if b_kmajor else 1] == K and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        odt = torch.bfloat16 if epilogue in (EPI_BF16, EPI_GELU_BF16, EPI_GELU_ERF_BF16) else torch.float32
        out = torch.empty(M, N, dtype=odt, device=a.device)
    gemm_raw(_p(a), _p(w), _p(out), M, N, K, a.stride(0), w.stride(0), out.stride(0), epilogue,
             bias=_p(bias), bias_mode=BIAS_N if bias is not None else BIAS_NONE, b_kmajor=b_kmajor)
    return out


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """out[M, N] (+)= a[K, M]^T @ b[K, N]  (fp32): both operands row-major with the contraction index on the ROWS
    (weight gradient dW = dy^T x without transposed copies).  a, b bf16, row stride free (multiple of 8)."""
    _dev(a, b, out)
    assert a.dtype == b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0]
    assert a.stride(1) == 1 and b.stride(1) == 1
    K, M = a.shape
    N = b.shape[1]
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    args = _lib.GemmTnArgs(_p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0), int(accumulate))
    check(lib.omh_gemm_bf16_tn(C.byref(args), _stream()), "omh_gemm_bf16_tn")
    return out


def gemm_tn_grouped(problems):
    """``problems``: list of (a [K, M] bf16, b [K, N] bf16, out fp32 [M, N], accumulate) — ``out (+)= a^T @ b`` for all of
    them in one launch per OMH_TN_GROUP_MAX entries (include/omh.h): no split K, no atomics."""
    for k0 in range(0, len(problems), _lib.TN_GROUP_MAX):
        chunk = problems[k0:k0 + _lib.TN_GROUP_MAX]
        g = _lib.GemmTnGroup()
        g.n = len(chunk)
        for i, (a, b, out, acc) in enumerate(chunk):
            _dev(a, b, out)
            assert a.dtype == b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0]
            assert a.stride(1) == 1 and b.stride(1) == 1 and out.dtype == torch.float32 and out.stride(1) == 1
            assert out.shape == (a.shape[1], b.shape[1])
            g.problem[i] = _lib.GemmTnArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[1], b.shape[1], a.shape[0],
                                           a.stride(0), b.stride(0), out.stride(0), int(acc))
        check(lib.omh_gemm_bf16_tn_grouped(C.byref(g), _stream()), "omh_gemm_bf16_tn_grouped")


def set_option(key, value=None):
    """A dispatch switch of libomh.so (include/omh.h, omh_set_option): ``value`` None unsets it.  The library reads the
    environment once, at its first call; from then on this is the only way to change a switch (tests, A/B timing)."""
    check(lib.omh_set_option(key.encode(), None if value is None else str(value).encode()), f"omh_set_option({key})")


def get_option(key):
    v = lib.omh_get_option(key.encode())
    return None if v is None else v.decode()


def reset_options():
    """Every option back to what the environment said when the library was first called."""
    check(lib.omh_set_option(None, None), "omh_set_option(NULL)")


class options:
    """``with ops.options(GEMM_KERNEL="8w", GEMM_TILE=None): ...`` — set, run, restore.

    Options are process-wide and omh_set_option is not synchronised against launches: do not change them while a
    backward pass is running (the reducer and the weight-gradient stream launch from autograd's worker threads)."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        try:
            for k, v in self.kv.items():
                old = get_option(k)
                set_option(k, v)                    # raises on an unknown key / a value over 47 characters ...
                self.old[k] = old
        except Exception:
            self.__exit__()                         # ... and the keys already set go back (__exit__ is not called then)
            raise
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def flash_attn_raw(q, k, vt, o, k_lens, B, H, Lq, Lk, q_bs, q_rs, k_bs, k_rs, vt_bs, o_bs, o_rs, ldv, scale,
                   lse=None, q_prescaled=0, o32=None, flags=0, q_lens=None, window=(-1, -1)):
    """``flags``: ATTN_SHORT_KERNEL | ATTN_ALLOW_SPLIT (include/omh.h, ABI v8): the training step pins the short-sequence
    kernel (forward and re-run take the same one) and lets it split its last round of workgroups over the keys.
    ``q_lens`` (ABI v10): int32 [B] device pointer; output rows past a sample's query length are written as zero.
    ``window`` (ABI v12): (left, right) band around the bottom-right aligned diagonal, a side < 0 unbounded (causal =
    (left, 0)); a bounded side runs the short-sequence kernel."""
    a = AttnArgs(q, k, vt, o, k_lens, B, H, Lq, Lk, q_bs, q_rs, k_bs, k_rs, vt_bs, o_bs, o_rs, ldv, scale, lse,
                 int(q_prescaled), None, 0, o32, int(flags), q_lens, int(window[0]), int(window[1]))
    need = lib.omh_flash_attn_workspace_bytes(C.byref(a))          # split-KV tail (long-sequence kernel; short one if allowed)
    ws = None
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
    check(lib.omh_flash_attn_fwd_d128(C.byref(a), _stream()), "omh_flash_attn_fwd_d128")


def flash_attn(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, k_lens: Optional[torch.Tensor] = None,
               scale: Optional[float] = None, out: Optional[torch.Tensor] = None, q_lens: Optional[torch.Tensor] = None,
               window=(-1, -1)):
    """q [B,Lq,H,128], k [B,Lk,H,128] bf16; vt [B,H*128,ldv] bf16 (V transposed,
    ldv >= roundup(Lk,64)); k_lens / q_lens int32 [B] or None.  Returns [B,Lq,H,128] bf16 (rows past q_lens: zero).
    ``window`` = (left, right): flash-attn's bottom-right aligned band (causal = (-1, 0)); (-1, -1): full attention."""
    _dev(q, k, vt, k_lens, out, q_lens)
    if q_lens is not None:
        assert q_lens.dtype == torch.int32 and q_lens.numel() == q.shape[0]
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    assert D == 128 and k.shape == (B, Lk, H, D) and vt.shape[0] == B and vt.shape[1] == H * D
    assert q.dtype == k.dtype == vt.dtype == torch.bfloat16
    assert q.stride(3) == 1 and q.stride(2) == D and k.stride(3) == 1 and k.stride(2) == D and vt.stride(2) == 1
    if k_lens is not None:
        assert k_lens.dtype == torch.int32
    if out is None:
        out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
    flash_attn_raw(_p(q), _p(k), _p(vt), _p(out), _p(k_lens), B, H, Lq, Lk, q.stride(0), q.stride(1), k.stride(0),
                   k.stride(1), vt.stride(0), out.stride(0), out.stride(1), vt.stride(1),
                   float(scale if scale is not None else D ** -0.5), q_lens=_p(q_lens), window=window)
    return out


def flash_attn_bwd(q, k, v, o, dout, lse, k_lens, B, H, Lq, Lk, scale=None, q_prescaled=False, out=None, o32=None,
                   phase=0, delta=None, split=True):
    """Fused attention backward (include/omh.h).  q, dout: bf16 [B*Lq, H*128]; k, v: bf16 [B*Lk, H*128] (row stride
    free); lse fp32 [B, H, Lq] from ``flash_attn_raw(..., lse=)``; k_lens int32 [B] or None.
    Returns fp32 dq [B*Lq, H*128], dk, dv [B*Lk, H*128] — or, with ``out=(dq, dk, dv)`` bf16 2-D tensors (row stride
    free: e.g. the three column blocks of one [rows, 3*H*128] buffer), writes bf16 gradients there.
    ``q_prescaled``: q carries scale*log2(e) as in the forward call.  ``o32``: the forward's fp32 output
    (``flash_attn_raw(..., o32=)``, fp32 [B*Lq, H*128] contiguous) — selects the round-3 kernels (no transposed copies,
    delta from dO . o32); without it round 2's kernels run (three transposes + a delta pass over the keys).
    ``phase`` (with o32): 0 everything; 1 delta only, 2 dQ only, 3 dK / dV only — 2 and 3 read the ``delta`` tensor
    (fp32 [B, H, Lq]) a phase-1 call filled and may run on two streams.  ``split`` (with o32): hand the kernels scratch so
    that a partly filled last round of workgroups is split over the inner loop (include/omh.h, ABI v8)."""
    _dev(q, k, v, dout, lse, k_lens, o32)
    d = H * 128
    for t in (q, k, v, dout):
        assert t.dtype == torch.bfloat16 and t.stride(-1) == 1 and t.shape[-1] == d
        assert t.stride(-2) % 8 == 0, "flash_attn_bwd: row strides must be multiples of 8 elements (16-byte rows)"
    assert lse.dtype == torch.float32 and lse.is_contiguous() and lse.numel() == B * H * Lq
    dev = q.device
    rs = lambda t: t.stride(-2)
    ldq, ldk = (Lq + 63) // 64 * 64, (Lk + 63) // 64 * 64
    qt = dot = kt = None
    if o32 is None:
        def padded(L, ld):                     # only the pad columns need the zeros (the transpose writes the rest)
            t = torch.empty(B, d, ld, dtype=torch.bfloat16, device=dev)
            if ld != L:
                t[:, :, L:].zero_()
            return t
        qt, dot, kt = padded(Lq, ldq), padded(Lq, ldq), padded(Lk, ldk)
        transpose_bf16_raw(ptr(q), ptr(qt), Lq, d, rs(q), ldq, batch=B, bs_in=Lq * rs(q), bs_out=d * ldq)
        transpose_bf16_raw(ptr(dout), ptr(dot), Lq, d, rs(dout), ldq, batch=B, bs_in=Lq * rs(dout), bs_out=d * ldq)
        transpose_bf16_raw(ptr(k), ptr(kt), Lk, d, rs(k), ldk, batch=B, bs_in=Lk * rs(k), bs_out=d * ldk)
    else:
        assert o32.dtype == torch.float32 and o32.is_contiguous() and o32.shape[-1] == d and rs(dout) == d
    assert phase == 0 or (o32 is not None and delta is not None)
    if delta is None:
        delta = torch.empty(B, H, Lq, dtype=torch.float32, device=dev)
    assert delta.dtype == torch.float32 and delta.is_contiguous() and delta.numel() == B * H * Lq
    if out is None:
        dq = torch.empty(B * Lq, d, dtype=torch.float32, device=dev)
        dk = torch.empty(B * Lk, d, dtype=torch.float32, device=dev)
        dv = torch.empty(B * Lk, d, dtype=torch.float32, device=dev)
        bf = 0
    else:
        dq, dk, dv = out
        for t in out:
            assert t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1 and t.shape[1] == d
        assert dk.stride(0) == dv.stride(0)
        bf = 1
    assert rs(k) == rs(v)
    a = _lib.AttnBwdArgs(_p(q), _p(k), _p(v), _p(o), _p(dout), _p(qt), _p(dot), _p(kt), _p(lse), _p(delta),
                         _p(dq), _p(dk), _p(dv), _p(k_lens), B, H, Lq, Lk,
                         Lq * rs(q), rs(q), Lk * rs(k), rs(k), Lq * rs(dout), rs(dout), Lq * dq.stride(0), dq.stride(0),
                         Lk * dk.stride(0), dk.stride(0), d * ldq, d * ldk, ldq, ldk,
                         float(scale if scale is not None else 128 ** -0.5), int(q_prescaled), bf, _p(o32), int(phase),
                         None, 0)
    need = lib.omh_flash_attn_bwd_workspace_bytes(C.byref(a)) if split else 0     # partial sums of the split last round
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
    check(lib.omh_flash_attn_bwd_d128(C.byref(a), _stream()), "omh_flash_attn_bwd_d128")
    return dq, dk, dv


def layernorm_modulate_raw(x, y, rows, dim, eps, mul_const, mul0, mul1, mul1_stride, add0, add1, add1_stride,
                           rows_per_batch):
    check(lib.omh_layernorm_modulate(x, y, rows, dim, eps, mul_const, mul0, mul1, mul1_stride, add0, add1,
                                     add1_stride, rows_per_batch, _stream()), "omh_layernorm_modulate")


def layernorm_modulate(x: torch.Tensor, eps: float, mul_const: float = 1.0, mul0=None, mul1=None, add0=None,
                       add1=None, rows_per_batch: Optional[int] = None, out=None):
    """x fp32 [rows, dim] -> bf16; see include/omh.h."""
    _dev(x, mul0, mul1, add0, add1)
    assert x.dtype == torch.float32 and x.is_contiguous()
    rows, dim = x.numel() // x.shape[-1], x.shape[-1]
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    layernorm_modulate_raw(_p(x), _p(out), rows, dim, eps, mul_const, _p(mul0), _p(mul1),
                           mul1.stride(0) if mul1 is not None else 0, _p(add0), _p(add1),
                           add1.stride(0) if add1 is not None else 0, rows_per_batch or rows)
    return out


def rmsnorm_rope_raw(x, ldx, y, rows, dim, weight, eps, do_norm, rope_cos, rope_sin, rope_len, head_dim, grid,
                     seq_len):
    check(lib.omh_rmsnorm_rope(x, ldx, y, rows, dim, weight, eps, do_norm, rope_cos, rope_sin, rope_len, head_dim,
                               grid, seq_len, _stream()), "omh_rmsnorm_rope")


def rmsnorm_rope_bf16_raw(x, ldx, y, rows, dim, weight, eps, do_norm, rope_cos, rope_sin, rope_len, head_dim, grid,
                          seq_len, out_scale=1.0):
    check(lib.omh_rmsnorm_rope_bf16(x, ldx, y, rows, dim, weight, eps, do_norm, rope_cos, rope_sin, rope_len,
                                    head_dim, grid, seq_len, float(out_scale), _stream()), "omh_rmsnorm_rope_bf16")


def rmsnorm_rope_bf16_pair_raw(x, ldx, seg_x, y0, y1, rows, dim, weight0, weight1, eps, do_norm, rope_cos, rope_sin, rope_len,
                               head_dim, grid, seq_len, out_scale0=1.0, out_scale1=1.0):
    """``rmsnorm_rope_bf16_raw`` on two column segments of the same rows in one launch (include/omh.h, ABI v9): segment 1
    reads ``x + seg_x`` elements, with its own gain / output / output scale.  Same bits as two calls
    (OMH_RMS_PAIR=0: issued as the two calls, A/B timing)."""
    if _os.environ.get("OMH_RMS_PAIR") == "0":
        rmsnorm_rope_bf16_raw(x, ldx, y0, rows, dim, weight0, eps, do_norm, rope_cos, rope_sin, rope_len, head_dim, grid,
                              seq_len, out_scale0)
        x1 = C.c_void_p(x.value + 2 * seg_x) if isinstance(x, C.c_void_p) else x + 2 * seg_x
        rmsnorm_rope_bf16_raw(x1, ldx, y1, rows, dim, weight1, eps, do_norm, rope_cos, rope_sin, rope_len, head_dim, grid,
                              seq_len, out_scale1)
        return
    check(lib.omh_rmsnorm_rope_bf16_pair(x, ldx, seg_x, y0, y1, rows, dim, weight0, weight1, eps, do_norm, rope_cos, rope_sin,
                                         rope_len, head_dim, grid, seq_len, float(out_scale0), float(out_scale1), _stream()),
          "omh_rmsnorm_rope_bf16_pair")


def rmsnorm_rope(x: torch.Tensor, weight: Optional[torch.Tensor], eps: float, do_norm: bool = True,
                 rope_cos=None, rope_sin=None, head_dim: int = 128, grid: Optional[torch.Tensor] = None,
                 seq_len: int = 0, out=None):
    """x fp32 [rows, dim] (row stride may exceed dim) -> bf16 [rows, dim]."""
    _dev(x, weight, rope_cos, rope_sin, grid)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, dim = x.shape
    if out is None:
        out = torch.empty(rows, dim, dtype=torch.bfloat16, device=x.device)
    rmsnorm_rope_raw(_p(x), x.stride(0), _p(out), rows, dim, _p(weight), eps, int(do_norm), _p(rope_cos),
                     _p(rope_sin), rope_cos.shape[0] if rope_cos is not None else 0, head_dim, _p(grid), seq_len)
    return out


def cast_bf16(x: torch.Tensor, out=None):
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.omh_cast_f32_bf16(_p(x), _p(out), x.numel(), _stream()), "omh_cast_f32_bf16")
    return out


def patchify(x: torch.Tensor, patch, Kp: int, out=None):
    """x fp32 [C,F,H,W] -> bf16 [f*h*w, Kp]."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    Cc, F, H, W = x.shape
    pt, ph, pw = patch
    n = (F // pt) * (H // ph) * (W // pw)
    if out is None:
        out = torch.empty(n, Kp, dtype=torch.bfloat16, device=x.device)
    check(lib.omh_patchify(_p(x), _p(out), Cc, F, H, W, pt, ph, pw, Kp, _stream()), "omh_patchify")
    return out


def unpatchify(tok: torch.Tensor, c_out: int, grid, patch):
    """tok fp32 [>= f*h*w, pt*ph*pw*c_out] -> fp32 [c_out, f*pt, h*ph, w*pw]."""
    _dev(tok)
    assert tok.dtype == torch.float32 and tok.is_contiguous()
    f, h, w = grid
    pt, ph, pw = patch
    out = torch.empty(c_out, f * pt, h * ph, w * pw, dtype=torch.float32, device=tok.device)
    check(lib.omh_unpatchify(_p(tok), _p(out), c_out, f, h, w, pt, ph, pw, _stream()), "omh_unpatchify")
    return out


def dense_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act_in: int = 0, act_out: int = 0):
    """y[b] = act_out(act_in(x[b]) @ w^T + bias), all fp32 (time embedding)."""
    _dev(x, w, bias)
    assert x.dtype == w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()
    B, K = x.shape
    N = w.shape[0]
    y = torch.empty(B, N, dtype=torch.float32, device=x.device)
    check(lib.omh_dense_f32(_p(x), _p(w), _p(bias), _p(y), B, N, K, act_in, act_out, _stream()), "omh_dense_f32")
    return y


def sinusoidal_embedding(t: torch.Tensor, dim: int):
    _dev(t)
    t = t.reshape(-1).to(torch.float32).contiguous()
    out = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
    check(lib.omh_sinusoidal_embedding(_p(t), _p(out), t.shape[0], dim, _stream()), "omh_sinusoidal_embedding")
    return out


def cfg_unipc_step(cond, uncond, x, last, m1, m2, mt_out, xc_out, x_next, guide, sigma, use_corr, ca, pb):
    """One fused CFG + UniPC update (include/omh.h); ca = (last, m1, m2, mt), pb = (x, mt, m1)."""
    _dev(cond, uncond, x, last, m1, m2, mt_out, xc_out, x_next)
    for t in (cond, uncond, x, last, m1, m2, mt_out, xc_out, x_next):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    check(lib.omh_cfg_unipc_step(_p(cond), _p(uncond), _p(x), _p(last), _p(m1), _p(m2), _p(mt_out), _p(xc_out),
                                 _p(x_next), x.numel(), float(guide), float(sigma), int(use_corr), float(ca[0]),
                                 float(ca[1]), float(ca[2]), float(ca[3]), float(pb[0]), float(pb[1]), float(pb[2]),
                                 _stream()), "omh_cfg_unipc_step")
    return x_next


# ----------------------------------------------------------------------------- VAE kernels
def conv_pair_supported(Tin, Hin, Win, Cin2, Tout, Hout, Wout, Cout, KT, KH, KW, stride_t=1, stride_hw=1, pad_h=0, pad_w=0,
                        up2=False):
    """Would omh_conv_cl_bf16 take this layer as a split-bf16 PAIR convolution (omh_conv_args.pair; ``Cin2`` = 2 x the real
    channel count)?  By the layer's geometry; no pointers involved."""
    a = _lib.ConvArgs(None, None, None, None, None, Tin, Hin, Win, Cin2, Tout, Hout, Wout, Cout, KT, KH, KW, stride_t,
                      stride_hw, pad_h, pad_w, int(up2), 1, 0, 1, None, None, 0, 1)
    return bool(lib.omh_conv_pair_supported(C.byref(a)))


def conv_cl(x, w, bias, Tout, Hout, Wout, Cout, KT, KH, KW, stride_t=1, stride_hw=1, pad_h=0, pad_w=0, up2=False,
            resid=None, out_f32=False, split_n=0, out=None, norm_gamma=None, norm_out=None, norm_only=False, pair=False):
    """Implicit-GEMM conv on channels-last bf16 ``x`` [Tin, Hin, Win, Cin] (history frames first);
    ``w`` bf16 [Cout, KT*KH*KW*Cin].  Returns [Tout*f, Hout, Wout, Cout/f] (f = Cout/split_n or 1).
    ``norm_gamma`` (fp32 [Cout]) + ``norm_out`` (bf16 [Tout, Hout, Wout, Cout]): also the next layer's RMS norm + SiLU of
    the output (fused into the kernel's epilogue where possible, omh.h); ``norm_only``: the caller will not read the
    returned tensor (its memory is still needed by the un-fused route)."""
    _dev(x, w, bias, resid, out, norm_gamma, norm_out)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.is_contiguous() and w.is_contiguous()
    Tin, Hin, Win, Cin = x.shape
    assert w.shape == (Cout, KT * KH * KW * Cin), (tuple(w.shape), Cout, KT, KH, KW, Cin)
    f = Cout // split_n if split_n else 1
    cch = split_n if split_n else Cout
    if out is None:
        out = torch.empty(Tout * f, Hout, Wout, cch, dtype=torch.float32 if out_f32 else torch.bfloat16,
                          device=x.device)
    assert out.is_contiguous() and (resid is None or (resid.is_contiguous() and
                                                      resid.dtype in (torch.bfloat16, torch.float32)))
    a = _lib.ConvArgs(_p(x), _p(w), _p(bias), _p(resid), _p(out), Tin, Hin, Win, Cin, Tout, Hout, Wout, Cout,
                      KT, KH, KW, stride_t, stride_hw, pad_h, pad_w, int(up2), int(out_f32), split_n,
                      int(resid is not None and resid.dtype == torch.float32),
                      _p(norm_gamma), _p(norm_out), int(bool(norm_only)), int(bool(pair)))
    if norm_gamma is not None:
        assert split_n == 0 and norm_gamma.dtype == torch.float32 and norm_gamma.numel() == Cout and norm_gamma.is_contiguous()
        assert norm_out is not None and norm_out.dtype == torch.bfloat16 and norm_out.is_contiguous() and \
            norm_out.numel() == Tout * Hout * Wout * Cout * (2 if pair else 1)
    check(lib.omh_conv_cl_bf16(C.byref(a), _stream()), "omh_conv_cl_bf16")
    return out


def rms_silu_cl(x, gamma, out=None, do_silu=True):
    """x bf16 or fp32 [..., C] channels-last -> bf16, per-voxel RMS norm (* gamma) then SiLU."""
    _dev(x, gamma, out)
    assert x.dtype in (torch.bfloat16, torch.float32) and x.is_contiguous() and gamma.dtype == torch.float32
    Cc = x.shape[-1]
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    assert out.is_contiguous() and out.dtype == torch.bfloat16
    fn, name = ((lib.omh_rms_silu_cl, "omh_rms_silu_cl") if x.dtype == torch.bfloat16 else
                (lib.omh_rms_silu_cl_f32in, "omh_rms_silu_cl_f32in"))
    check(fn(_p(x), _p(gamma), _p(out), x.numel() // Cc, Cc, int(do_silu), _stream()), name)
    return out


# ---- the fp32-faithful VAE mode (include/omh.h, ABI v8): operands as bf16 pairs in three channel blocks
def split3(x: torch.Tensor, pattern: int, Cp: Optional[int] = None, out=None):
    """x fp32 [..., C] (last dim contiguous, rows contiguous or a 2-D view with a row pitch) -> bf16 [..., 3 Cp] with
    hi = bf16(x), lo = bf16(x - hi): pattern 0 = [hi | lo | hi] (activations), 1 = [hi | hi | lo] (weights)."""
    _dev(x, out)
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    Cc = x.shape[-1]
    Cp = Cp or (Cc + 7) // 8 * 8
    if x.dim() == 2:
        rows, ldx = x.shape[0], x.stride(0)
    else:
        assert x.is_contiguous()
        rows, ldx = x.numel() // Cc, Cc
    nb = 2 if pattern == 2 else 3                             # pattern 2: pairs per 16 channels, [hi(16) | lo(16)] (omh.h)
    if out is None:
        out = torch.empty(*x.shape[:-1], nb * Cp, dtype=torch.bfloat16, device=x.device)
    assert out.dtype == torch.bfloat16 and out.shape[-1] == nb * Cp and out.numel() == rows * nb * Cp and out.is_contiguous()
    check(lib.omh_split3_f32(_p(x), ldx, _p(out), nb * Cp, rows, Cc, Cp, int(pattern), _stream()), "omh_split3_f32")
    return out


def rms_silu_cl_split3(x, gamma, out=None, do_silu=True, pair=False):
    """x fp32 [..., C] -> pattern-0 bf16 [..., 3 C]: RMS norm (* gamma), SiLU, split.  ``pair`` (or an ``out`` of 2 C
    channels): the pair layout [..., 2 C] of omh_conv_args.pair instead."""
    _dev(x, gamma, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and gamma.dtype == torch.float32
    Cc = x.shape[-1]
    if out is not None:
        pair = out.shape[-1] == 2 * Cc
    nb = 2 if pair else 3
    if out is None:
        out = torch.empty(*x.shape[:-1], nb * Cc, dtype=torch.bfloat16, device=x.device)
    assert out.is_contiguous() and out.dtype == torch.bfloat16 and out.shape[-1] == nb * Cc
    fn, name = (lib.omh_rms_silu_cl_pair, "omh_rms_silu_cl_pair") if pair else (lib.omh_rms_silu_cl_split3, "omh_rms_silu_cl_split3")
    check(fn(_p(x), _p(gamma), _p(out), x.numel() // Cc, Cc, int(do_silu), _stream()), name)
    return out


def nchw_to_cl_f32(x, T, t0, Cp, mul=None, add=None):
    """x fp32 [C, Ttot, H, W] frames [t0, t0+T) -> fp32 [T, H, W, Cp]."""
    _dev(x, mul, add)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    Cc, Ttot, H, W = x.shape
    out = torch.empty(T, H, W, Cp, dtype=torch.float32, device=x.device)
    check(lib.omh_nchw_to_cl_f32(_p(x), _p(out), Cc, T, H, W, Cp, _p(mul), _p(add), Ttot, t0, _stream()),
          "omh_nchw_to_cl_f32")
    return out


def softmax_rows_f32(x, out, L, scale):
    """x fp32 [R, >=L] -> out fp32 [R, >=L] (first L columns), row softmax of x*scale."""
    _dev(x, out)
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and x.stride(1) == 1 and out.stride(1) == 1
    check(lib.omh_softmax_rows_f32(_p(x), x.stride(0), _p(out), out.stride(0), x.shape[0], L, scale, _stream()),
          "omh_softmax_rows_f32")
    return out


def relu_bf16_(x):
    """In-place ReLU on a contiguous bf16 tensor (numel % 8 == 0)."""
    _dev(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    check(lib.omh_relu_bf16(_p(x), x.numel(), _stream()), "omh_relu_bf16")
    return x


def relu_bwd_bf16(dy, y):
    """g = y > 0 ? dy : 0 (y = the ReLU's output); contiguous bf16, numel % 8 == 0."""
    _dev(dy, y)
    assert dy.dtype == y.dtype == torch.bfloat16 and dy.is_contiguous() and y.is_contiguous() and dy.shape == y.shape
    g = torch.empty_like(dy)
    check(lib.omh_relu_bwd_bf16(_p(dy), _p(y), _p(g), dy.numel(), _stream()), "omh_relu_bwd_bf16")
    return g


def nchw_to_cl(x, T, t0, Cp, mul=None, add=None, out=None):
    """x fp32 [C, Ttot, H, W] frames [t0, t0+T) -> bf16 [T, H, W, Cp]."""
    _dev(x, mul, add, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    Cc, Ttot, H, W = x.shape
    if out is None:
        out = torch.empty(T, H, W, Cp, dtype=torch.bfloat16, device=x.device)
    check(lib.omh_nchw_to_cl(_p(x), _p(out), Cc, T, H, W, Cp, _p(mul), _p(add), Ttot, t0, _stream()),
          "omh_nchw_to_cl")
    return out


def cl_to_nchw(x, out, t0, Cc, mul=None, add=None, lo=-3.0e38, hi=3.0e38):
    """x fp32 [T, H, W, Cp] -> out fp32 [Cc, Ttot, H, W] frames [t0, t0+T)."""
    _dev(x, out, mul, add)
    assert x.dtype == torch.float32 and x.is_contiguous() and out.dtype == torch.float32 and out.is_contiguous()
    T, H, W, Cp = x.shape
    assert out.shape[0] == Cc and tuple(out.shape[2:]) == (H, W)
    check(lib.omh_cl_to_nchw(_p(x), _p(out), Cc, T, H, W, Cp, _p(mul), _p(add), lo, hi, out.shape[1], t0,
                             _stream()), "omh_cl_to_nchw")
    return out


def softmax_rows(x, out, L, scale):
    """x fp32 [R, >=L] -> out bf16 [R, >=L] (first L columns), row softmax of x*scale."""
    _dev(x, out)
    assert x.dtype == torch.float32 and out.dtype == torch.bfloat16 and x.stride(1) == 1 and out.stride(1) == 1
    check(lib.omh_softmax_rows(_p(x), x.stride(0), _p(out), out.stride(0), x.shape[0], L, scale, _stream()),
          "omh_softmax_rows")
    return out


# ----------------------------------------------------------------------------- backward kernels (raw pointer API)
def transpose_bf16_raw(src, dst, R, Cc, ld_in, ld_out, batch=1, bs_in=0, bs_out=0):
    check(lib.omh_transpose_bf16(src, dst, R, Cc, ld_in, ld_out, batch, bs_in, bs_out, _stream()),
          "omh_transpose_bf16")


def transpose_bf16(x: torch.Tensor, pad_to: int = 8):
    """x bf16 [R, C] (row stride free) -> zero-padded bf16 [C, roundup(R, pad_to)]."""
    _dev(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    out = torch.zeros(Cc, Rp, dtype=torch.bfloat16, device=x.device) if Rp != R else \
        torch.empty(Cc, Rp, dtype=torch.bfloat16, device=x.device)
    transpose_bf16_raw(_p(x), _p(out), R, Cc, x.stride(0), Rp)
    return out


def transpose_bf16_batched(xt: torch.Tensor, L: int) -> torch.Tensor:
    """xt bf16 [B, d, ld] (e.g. a V^T buffer of the attention forward) -> its first L columns transposed per batch
    element, stacked: bf16 [B*L, d]."""
    _dev(xt)
    assert xt.dtype == torch.bfloat16 and xt.dim() == 3 and xt.is_contiguous() and L <= xt.shape[2]
    B, d, ld = xt.shape
    out = torch.empty(B * L, d, dtype=torch.bfloat16, device=xt.device)
    transpose_bf16_raw(_p(xt), _p(out), d, L, ld, d, batch=B, bs_in=d * ld, bs_out=L * d)
    return out


def cast_bf16_strided(x: torch.Tensor, out: torch.Tensor):
    """fp32 contiguous [R, C] -> an existing bf16 [R, C] view with a free row stride (cast kernel + strided copy)."""
    out.copy_(cast_bf16(x))
    return out


def set_deterministic(on=None) -> bool:
    """omh_set_deterministic: with True the few launches of the backward that combine partial sums with fp32 atomics
    (bias-gradient column sums, gate gradients, split-K weight gradients, the time-embedding MLP's input gradient)
    give each output element to one workgroup — a training step then repeats bit for bit (slower; debugging aid).
    None only queries.  Returns the mode in force.  Also settable with OMH_DETERMINISTIC=1 in the environment."""
    return bool(lib.omh_set_deterministic(-1 if on is None else int(bool(on))))


def colsum_accum(x: torch.Tensor, out: torch.Tensor):
    """out[c] += sum_r x[r][c]; x bf16/fp32 [R, C]."""
    _dev(x, out)
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.float32
    check(lib.omh_colsum_accum(_p(x), int(x.dtype == torch.bfloat16), x.stride(0), _p(out), x.shape[0], x.shape[1],
                               _stream()), "omh_colsum_accum")
    return out


def colsum_accum_multi(pairs):
    """out_i[c] += sum_r x_i[r][c] for every (x_i, out_i) of ``pairs`` in one launch per 16 entries."""
    for k in range(0, len(pairs), _lib.COLSUM_MAX):
        chunk = pairs[k:k + _lib.COLSUM_MAX]
        b = _lib.ColsumBatch()
        b.n = len(chunk)
        for i, (x, out) in enumerate(chunk):
            _dev(x, out)
            assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.float32 and out.is_contiguous()
            b.x[i], b.out[i] = x.data_ptr(), out.data_ptr()
            b.ld[i], b.R[i], b.C[i] = x.stride(0), x.shape[0], x.shape[1]
            b.is_bf16[i] = int(x.dtype == torch.bfloat16)
        check(lib.omh_colsum_accum_multi(C.byref(b), _stream()), "omh_colsum_accum_multi")


def gelu_tanh(x, out=None):
    _dev(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    check(lib.omh_gelu_tanh_bf16(_p(x), _p(out), x.numel(), _stream()), "omh_gelu_tanh_bf16")
    return out


def gelu_tanh_bwd(dy, x_pre, out=None):
    _dev(dy, x_pre)
    assert dy.dtype == x_pre.dtype == torch.bfloat16 and dy.is_contiguous() and x_pre.is_contiguous()
    out = torch.empty_like(dy) if out is None else out
    check(lib.omh_gelu_tanh_bwd_bf16(_p(dy), _p(x_pre), _p(out), dy.numel(), _stream()), "omh_gelu_tanh_bwd_bf16")
    return out


def gelu_erf(x, out=None):
    _dev(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    check(lib.omh_gelu_erf_bf16(_p(x), _p(out), x.numel(), _stream()), "omh_gelu_erf_bf16")
    return out


def gelu_erf_bwd(dy, x_pre, out=None):
    _dev(dy, x_pre)
    assert dy.dtype == x_pre.dtype == torch.bfloat16 and dy.is_contiguous() and x_pre.is_contiguous()
    out = torch.empty_like(dy) if out is None else out
    check(lib.omh_gelu_erf_bwd_bf16(_p(dy), _p(x_pre), _p(out), dy.numel(), _stream()), "omh_gelu_erf_bwd_bf16")
    return out


def gated_residual_fwd_raw(xi, y, xo, rows, dim, gate_const, gate0, gate1, gate1_stride, rows_per_batch):
    check(lib.omh_gated_residual_fwd(xi, y, xo, rows, dim, gate_const, gate0, gate1, gate1_stride, rows_per_batch,
                                     _stream()), "omh_gated_residual_fwd")


def gated_residual_bwd_raw(dx, y, dy, dgate, dgate_stride, rows, dim, gate_const, gate0, gate1, gate1_stride,
                           rows_per_batch):
    check(lib.omh_gated_residual_bwd(dx, y, dy, dgate, dgate_stride, rows, dim, gate_const, gate0, gate1,
                                     gate1_stride, rows_per_batch, _stream()), "omh_gated_residual_bwd")


def layernorm_modulate_bwd_raw(x, dy, dx, rows, dim, eps, mul_const, mul0, mul1, mul1_stride, dmul, dadd, dstride,
                               rows_per_batch):
    check(lib.omh_layernorm_modulate_bwd(x, dy, dx, rows, dim, eps, mul_const, mul0, mul1, mul1_stride, dmul, dadd,
                                         dstride, rows_per_batch, _stream()), "omh_layernorm_modulate_bwd")


def rmsnorm_rope_bwd_raw(x, ldx, dy, lddy, dx, lddx, dw, rows, dim, weight, eps, do_norm, rope_cos, rope_sin,
                         rope_len, head_dim, grid, seq_len):
    check(lib.omh_rmsnorm_rope_bwd(x, ldx, dy, lddy, dx, lddx, dw, rows, dim, weight, eps, do_norm, rope_cos,
                                   rope_sin, rope_len, head_dim, grid, seq_len, _stream()), "omh_rmsnorm_rope_bwd")


def rmsnorm_rope_bwd_t_raw(x, x_bf16, ldx, dy, dy_bf16, lddy, dx, lddx, dw, rows, dim, weight, eps, do_norm, rope_cos,
                           rope_sin, rope_len, head_dim, grid, seq_len):
    """omh_rmsnorm_rope_bwd with bf16 or fp32 x / dy (include/omh.h); dx may alias dy."""
    check(lib.omh_rmsnorm_rope_bwd_t(x, int(x_bf16), ldx, dy, int(dy_bf16), lddy, dx, lddx, dw, rows, dim, weight, eps,
                                     do_norm, rope_cos, rope_sin, rope_len, head_dim, grid, seq_len, _stream()),
          "omh_rmsnorm_rope_bwd_t")


def layernorm_modulate_bwd2(x, dy, dx, rows, dim, eps, mul_const, mul0, mul1, mul1_stride, dmul, dadd, dstride,
                            rows_per_batch, dy_next=None, y_next=None, gate_const=1.0, gate0=None, gate1=None,
                            gate1_stride=0, dgate=None, dgate_stride=0, defer=None):
    """Repeatable LayerNorm+modulate backward, optionally fused with the next branch's gated-residual backward
    (include/omh.h).  x, dx fp32 tensors [rows, dim]; dy fp32 or bf16 tensor; the pointer-like arguments (mul0 ... dgate)
    are c_void_p / None as for the *_raw ops; dy_next / y_next bf16 tensors or None.  ``defer`` (a list): the second
    launch — the partial column sums into dmul / dadd / dgate — is appended to it instead of being issued; the caller
    issues the list with ``partial_colsum_multi`` on the same stream before anything reads those sums."""
    _dev(x, dy, dx, dy_next, y_next)
    need = lib.omh_layernorm_modulate_bwd2_workspace(rows, dim, rows_per_batch)
    ws = torch.empty(need, dtype=torch.float32, device=x.device)
    d = _lib.PartialReduce() if defer is not None else None
    a = _lib.LnBwdArgs(_p(x), _p(dy), int(dy.dtype == torch.bfloat16), _p(dx), rows, dim, eps, mul_const, mul0, mul1,
                       mul1_stride, dmul, dadd, dstride, rows_per_batch, _p(dy_next), _p(y_next), gate_const, gate0, gate1,
                       gate1_stride, dgate, dgate_stride, _p(ws), need, C.pointer(d) if d is not None else None)
    check(lib.omh_layernorm_modulate_bwd2(C.byref(a), _stream()), "omh_layernorm_modulate_bwd2")
    if d is not None and d.part:
        defer.append((d, ws))                                # (the partials live in ws until the deferred launch)


def partial_colsum_multi(deferred):
    """Issue the second launches collected by ``layernorm_modulate_bwd2(defer=)`` / ``rmsnorm_rope_bwd2(defer=)``: one
    launch per 8 entries, the same sums in the same order."""
    for k in range(0, len(deferred), _lib.PARTIAL_REDUCE_MAX):
        chunk = deferred[k:k + _lib.PARTIAL_REDUCE_MAX]
        b = _lib.PartialReduceBatch()
        b.n = len(chunk)
        for i, (d, _ws) in enumerate(chunk):
            b.e[i] = d
        check(lib.omh_partial_colsum_multi(C.byref(b), _stream()), "omh_partial_colsum_multi")


def rmsnorm_rope_bwd2(x, x_bf16, ldx, dy, dy_bf16, lddy, dx, lddx, rows, dim, eps, do_norm, weights, dweights, device,
                      n_seg=1, seg_x=0, seg_dy=0, seg_dx=0, rope_cos=None, rope_sin=None, rope_len=0, head_dim=128,
                      grid=None, seq_len=0, defer=None):
    """Repeatable RMSNorm(+RoPE) backward on 1 or 2 column segments (include/omh.h); x / dy / dx and the table
    pointers are c_void_p; weights / dweights: lists of fp32 tensors or None per segment.  ``defer``: as for
    ``layernorm_modulate_bwd2`` (the gains' partial sums)."""
    need = lib.omh_rmsnorm_rope_bwd2_workspace(rows, dim, n_seg) if any(d is not None for d in dweights) else 0
    ws = torch.empty(max(need, 1), dtype=torch.float32, device=device)
    W = (C.c_void_p * 2)(*[(w.data_ptr() if w is not None else None) for w in (list(weights) + [None])[:2]])
    DW = (C.c_void_p * 2)(*[(w.data_ptr() if w is not None else None) for w in (list(dweights) + [None])[:2]])
    d = _lib.PartialReduce() if defer is not None else None
    a = _lib.RmsBwdArgs(x, int(x_bf16), ldx, dy, int(dy_bf16), lddy, dx, lddx, n_seg, seg_x, seg_dy, seg_dx, W, DW, rows, dim,
                        eps, int(do_norm), rope_cos, rope_sin, rope_len, head_dim, grid, seq_len, _p(ws), need,
                        C.pointer(d) if d is not None else None)
    check(lib.omh_rmsnorm_rope_bwd2(C.byref(a), _stream()), "omh_rmsnorm_rope_bwd2")
    if d is not None and d.part:
        defer.append((d, ws))


def softmax_bwd_rows(p, dp, ds, L, scale):
    _dev(p, dp, ds)
    assert p.dtype == torch.bfloat16 and dp.dtype == torch.float32 and ds.dtype == torch.bfloat16
    check(lib.omh_softmax_bwd_rows(_p(p), p.stride(0), _p(dp), dp.stride(0), _p(ds), ds.stride(0), p.shape[0], L,
                                   scale, _stream()), "omh_softmax_bwd_rows")
    return ds


def unpatchify_bwd(g: torch.Tensor, grid, patch):
    """g fp32 [Cout, F, H, W] -> bf16 [f*h*w, pt*ph*pw*Cout]."""
    _dev(g)
    assert g.dtype == torch.float32 and g.is_contiguous()
    f, h, w = grid
    pt, ph, pw = patch
    out = torch.empty(f * h * w, pt * ph * pw * g.shape[0], dtype=torch.bfloat16, device=g.device)
    check(lib.omh_unpatchify_bwd(_p(g), _p(out), g.shape[0], f, h, w, pt, ph, pw, _stream()), "omh_unpatchify_bwd")
    return out


def dense_f32_bwd(x, w, dy, dW=None, db=None, dx=None, dx_accumulate=False, act_in=0):
    _dev(x, w, dy, dW, db, dx)
    B, K = x.shape
    N = w.shape[0]
    check(lib.omh_dense_f32_bwd(_p(x), _p(w), _p(dy), _p(dW), _p(db), _p(dx), int(dx_accumulate), B, N, K, act_in,
                                _stream()), "omh_dense_f32_bwd")


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _dev(p, g, m, v)
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous()
    check(lib.omh_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                             grad_scale, _stream()), "omh_adamw_step")


def adamw_multi(table, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """table: int64 device tensor [n, 5] = (param, grad, exp_avg, exp_avg_sq pointers, numel)."""
    _dev(table)
    assert table.dtype == torch.int64 and table.is_contiguous()
    check(lib.omh_adamw_multi(_p(table), n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, _stream()),
          "omh_adamw_multi")


def adamw_pack_multi(table, n_entries, total_tiles, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """AdamW + the bf16 operand copies of the updated weights in one launch; ``table``: int64 device tensor
    [n_entries, 12] (include/omh.h)."""
    _dev(table)
    assert table.dtype == torch.int64 and table.is_contiguous() and table.shape == (n_entries, 12)
    check(lib.omh_adamw_pack_multi(_p(table), n_entries, total_tiles, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                                   _stream()), "omh_adamw_pack_multi")


def pack_weights_multi(table, n_entries, total_tiles):
    """One launch for every bf16 operand copy (and transposed copy) of the fp32 master weights; ``table``: int64 device
    tensor [n_entries, 9] (include/omh.h)."""
    _dev(table)
    assert table.dtype == torch.int64 and table.is_contiguous() and table.shape == (n_entries, 9)
    check(lib.omh_pack_weights_multi(_p(table), n_entries, total_tiles, _stream()), "omh_pack_weights_multi")


def ema_update(ema, p, decay):
    _dev(ema, p)
    assert ema.dtype == p.dtype == torch.float32 and ema.is_contiguous() and p.is_contiguous()
    check(lib.omh_ema_update(_p(ema), _p(p), p.numel(), decay, _stream()), "omh_ema_update")


def cu_masked_stream(cus_per_32: int, high: bool = False) -> "torch.cuda.Stream":
    """A stream confined to ``cus_per_32`` of every 32 CUs (include/omh.h: omh_stream_create_cu_mask), wrapped for torch.
    The HIP stream lives as long as the process (a handful of them at most: the training step's second stream)."""
    h = C.c_void_p()
    check(lib.omh_stream_create_cu_mask(int(cus_per_32), int(bool(high)), C.cast(C.byref(h), C.c_void_p)),
          "omh_stream_create_cu_mask")
    return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", torch.cuda.current_device()))


def probe_mfma_tflops(random_operands: bool, iters: int = 400) -> float:
    """Measurement only (include/omh.h: omh_probe_mfma_tflops): TFLOP/s of back-to-back bf16 MFMAs, one wave per SIMD."""
    import ctypes as _C
    scratch = torch.empty(1024 * 256, device="cuda", dtype=torch.float32)
    out = _C.c_float(0.0)
    check(lib.omh_probe_mfma_tflops(int(random_operands), int(iters), _p(scratch), scratch.numel(),
                                    _C.cast(_C.byref(out), _C.c_void_p), _stream()), "omh_probe_mfma_tflops")
    return float(out.value)


def ema_update_multi(table, n_entries, total_chunks, decay):
    """table: device int64 [n, 4] = {ema, p, numel, first 4096-element chunk} (include/omh.h: omh_ema_update_multi)."""
    _dev(table)
    check(lib.omh_ema_update_multi(_p(table), int(n_entries), int(total_chunks), float(decay), _stream()),
          "omh_ema_update_multi")


# ----------------------------------------------------------------------------- prompt-side encoders (t5.py / clip.py)
import os as _os
_CHECK_IDS = _os.environ.get("OMH_CHECK_IDS", "1") != "0"


def gather_rows(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """nn.Embedding lookup: table fp32 or bf16 [V, dim], ids int64 [...] -> fp32 [..., dim].  ids outside [0, V) raise
    IndexError as nn.Embedding does (one host read-back of a flag: the encoders run once per prompt)."""
    _dev(table, ids)
    assert table.dtype in (torch.float32, torch.bfloat16) and table.is_contiguous() and ids.dtype == torch.int64
    idc = ids.contiguous()
    # the range check reads a flag back to the host: skipped under hipGraph capture (a sync is illegal there; the kernels
    # clamp ids, so only where the error surfaces changes) and with OMH_CHECK_IDS=0
    if idc.numel() and _CHECK_IDS and not torch.cuda.is_current_stream_capturing() and \
            bool(((idc < 0) | (idc >= table.shape[0])).any()):
        raise IndexError(f"token id outside [0, {table.shape[0]})")
    out = torch.empty(*ids.shape, table.shape[1], dtype=torch.float32, device=table.device)
    fn, name = (lib.omh_gather_rows_f32, "omh_gather_rows_f32") if table.dtype == torch.float32 else \
        (lib.omh_gather_rows_bf16, "omh_gather_rows_bf16")
    check(fn(_p(table), _p(idc), _p(out), idc.numel(), table.shape[1], table.shape[0], _stream()), name)
    return out


def rmsnorm_f32(x: torch.Tensor, weight: torch.Tensor, eps: float, want_f32=True, want_bf16=True):
    """T5LayerNorm: returns (fp32 result or None, bf16 result or None)."""
    _dev(x, weight)
    assert x.dtype == torch.float32 and x.is_contiguous()
    yf = torch.empty_like(x) if want_f32 else None
    yb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    check(lib.omh_rmsnorm_f32(_p(x), _p(weight), eps, _p(yf), _p(yb), x.numel() // x.shape[-1], x.shape[-1], _stream()),
          "omh_rmsnorm_f32")
    return yf, yb


def layernorm_f32(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    _dev(x, weight, bias)
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty_like(x)
    check(lib.omh_layernorm_f32(_p(x), _p(weight), _p(bias), eps, _p(y), x.numel() // x.shape[-1], x.shape[-1], _stream()),
          "omh_layernorm_f32")
    return y


def softmax_bias_rows(x: torch.Tensor, H: int, L: int, scale: float, bucket=None, table=None, klen=None, ldy=None):
    """x fp32 [H*L, ldx] scores -> bf16 [H*L, ldy] probabilities (include/omh.h)."""
    _dev(x, bucket, table)
    assert x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[0] == H * L
    ldy = ldy or x.shape[1]
    y = torch.empty(H * L, ldy, dtype=torch.bfloat16, device=x.device)
    check(lib.omh_softmax_bias_rows(_p(x), x.stride(0), _p(y), ldy, H, L, scale, _p(bucket), _p(table),
                                    L if klen is None else int(klen), _stream()), "omh_softmax_bias_rows")
    return y


def mul_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _dev(a, b)
    assert a.dtype == b.dtype == torch.bfloat16 and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    out = torch.empty_like(a)
    check(lib.omh_mul_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "omh_mul_bf16")
    return out


def vit_embed(tok: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """tok fp32 [B, n, dim], cls [dim], pos [n+1, dim] -> fp32 [B, n+1, dim]."""
    _dev(tok, cls, pos)
    B, n, d = tok.shape
    out = torch.empty(B, n + 1, d, dtype=torch.float32, device=tok.device)
    check(lib.omh_vit_embed(_p(tok), _p(cls), _p(pos), _p(out), B, n, d, _stream()), "omh_vit_embed")
    return out
