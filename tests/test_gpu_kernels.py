"""Parity of each gfx950 kernel (through the C ABI) against a plain fp32
PyTorch statement of the same op on the same bf16-rounded inputs."""
import math

import pytest
import torch

from conftest import rel_rms, set_option

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 72), (1560, 1536, 1536), (257, 64, 1536),
                                   (64, 8960, 256), (1000, 1536, 8960)])
@pytest.mark.parametrize("tile", ["big", "mid192", "small", "tiny"])
def test_gemm_bf16_f32_bias(ops, M, N, K, tile, monkeypatch):
    set_option("OMH_GEMM_TILE", tile)          # all three tile configurations on every (ragged) shape
    torch.manual_seed(M * 7 + N)
    a = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(N, device="cuda")
    ref = a.float() @ w.float().t() + bias
    out = ops.gemm(a, w, bias=bias, epilogue=ops.EPI_F32)
    assert rel_rms(out, ref) < 2e-5          # fp32 accumulate of identical bf16 inputs: ordering noise only
    outb = ops.gemm(a, w, bias=bias, epilogue=ops.EPI_BF16)
    assert rel_rms(outb.float(), ref) < 4e-3  # one bf16 rounding (2^-9 rel)
    outg = ops.gemm(a, w, bias=bias, epilogue=ops.EPI_GELU_BF16)
    refg = torch.nn.functional.gelu(ref, approximate="tanh")
    assert rel_rms(outg.float(), refg) < 5e-3


def test_gemm_asymmetric_layout(ops):
    # A = identity-like selector, B asymmetric: catches a transposed C write
    M = N = K = 128
    a = torch.zeros(M, K, device="cuda")
    a[torch.arange(M), torch.arange(M)] = 1.0
    w = (torch.arange(N, device="cuda").float()[:, None] * 0.5 + torch.arange(K, device="cuda").float()[None, :] * 0.0078125)
    out = ops.gemm(_bf(a), _bf(w), epilogue=ops.EPI_F32)
    ref = _bf(a).float() @ _bf(w).float().t()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("tile", ["big", "mid192", "small", "tiny"])
def test_gemm_resid_gate_and_batch(ops, tile, monkeypatch):
    set_option("OMH_GEMM_TILE", tile)
    B, S, d, K = 2, 200, 256, 320
    torch.manual_seed(1)
    a = _bf(torch.randn(B * S, K, device="cuda"))
    w = _bf(torch.randn(d, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(d, device="cuda")
    x0 = torch.randn(B * S, d, device="cuda")
    mod = torch.randn(6, d, device="cuda")
    e0 = torch.randn(B, 6, d, device="cuda")
    x = x0.clone()
    ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), B * S, d, K, K, K, d, ops.EPI_RESID, bias=ops.ptr(bias),
                 bias_mode=ops.BIAS_N, gate0=ops.ptr(mod, 2 * d), gate1=ops.ptr(e0, 2 * d), gate1_stride=6 * d,
                 gate_rows=S, gate_const=0.0)
    gate = (mod[2][None] + e0[:, 2]).repeat_interleave(S, 0)
    ref = x0 + (a.float() @ w.float().t() + bias) * gate
    assert rel_rms(x, ref) < 1e-5
    # batched "V^T" form: C[b][m][n] = W[m,:] . h[b,n,:] + bias[m]
    h = _bf(torch.randn(B, S, K, device="cuda"))
    Sp = 256
    vt = torch.zeros(B, d, Sp, dtype=torch.bfloat16, device="cuda")
    ops.gemm_raw(ops.ptr(w), ops.ptr(h), ops.ptr(vt), d, S, K, K, K, Sp, ops.EPI_BF16, bias=ops.ptr(bias),
                 bias_mode=ops.BIAS_M, batch=B, strideA=0, strideB=S * K, strideC=d * Sp)
    refv = torch.einsum("mk,bnk->bmn", w.float(), h.float()) + bias[None, :, None]
    assert rel_rms(vt[:, :, :S].float(), refv) < 4e-3
    assert float(vt[:, :, S:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K,gate_rows", [(256, 384, 256, 256), (1000, 776, 320, 300), (2000, 1536, 384, 500),
                                             (4100, 1160, 1536, 2050), (32760, 1536, 448, 16380)])
def test_gemm_w64_stream_kernel(ops, M, N, K, gate_rows, monkeypatch):
    """The 256 x 384 one-wave-per-SIMD stream kernel (gemm_w64.hip): each epilogue against fp32 torch AND bit for bit
    against the 8-wave kernel (same k order, same epilogue arithmetic) — ragged M and N, the N % 384 tail masked by
    EXEC, a strided gate table whose batch boundary falls inside a wave's 128 rows, even and odd numbers of k steps (the
    two tail paths of the stream)."""
    torch.manual_seed(M + N)
    a = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(N, device="cuda")
    nb = (M + gate_rows - 1) // gate_rows
    mod = torch.randn(6, N, device="cuda")
    e0 = torch.randn(nb, 6, N, device="cuda")
    x0 = torch.randn(M, N, device="cuda")
    ref = a.float() @ w.float().t() + bias

    def run(kernel):
        set_option("OMH_GEMM_KERNEL", kernel)
        x = x0.clone()
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(bias),
                     bias_mode=ops.BIAS_N, gate0=ops.ptr(mod, 2 * N), gate1=ops.ptr(e0, 2 * N), gate1_stride=6 * N,
                     gate_rows=gate_rows, gate_const=0.5)
        x1 = x0.clone()
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x1), M, N, K, K, K, N, ops.EPI_RESID, gate_const=1.0)
        return (ops.gemm(a, w, bias=bias, epilogue=ops.EPI_F32), ops.gemm(a, w, epilogue=ops.EPI_F32),
                ops.gemm(a, w, bias=bias, epilogue=ops.EPI_BF16), ops.gemm(a, w, bias=bias, epilogue=ops.EPI_GELU_BF16), x, x1)

    got, old = run("w64"), run("8w")
    for g, o in zip(got, old):
        assert torch.equal(g, o)
    assert rel_rms(got[0], ref) < 2e-5
    assert rel_rms(got[1], ref - bias) < 2e-5
    assert rel_rms(got[2].float(), ref) < 4e-3
    assert rel_rms(got[3].float(), torch.nn.functional.gelu(ref, approximate="tanh")) < 5e-3
    gate = (0.5 + mod[2][None] + e0[:, 2]).repeat_interleave(gate_rows, 0)[:M]
    assert rel_rms(got[4], x0 + ref * gate) < 1e-5
    assert rel_rms(got[5], x0 + ref - bias) < 1e-5


@pytest.mark.parametrize("M,N,K,gate_rows", [(256, 192, 1024, 256), (1000, 776, 1088, 300), (4100, 1160, 1536, 2050),
                                             (2000, 1536, 2048, 500), (32760, 1536, 1536, 16380)])
def test_gemm_w64_residual_stream_with_prefetched_c(ops, M, N, K, gate_rows, monkeypatch):
    """Round 4: the 256 x 192 gated-residual stream (gen_gemm_w64.py "resid192": the OLD C tile is requested during the
    first 12 k steps, so the epilogue neither waits for HBM reads nor idles the matrix pipe) — bit for bit against the
    8-wave kernel: gated and plain residual, with and without bias, ragged M / N (EXEC-masked columns, rows past M),
    a gate boundary inside a wave's rows, even and odd k-step counts from the shortest contraction it takes (16 k tiles);
    several tiles per persistent workgroup at the o-projection's real size (1024 tiles)."""
    torch.manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(N, device="cuda")
    nb = (M + gate_rows - 1) // gate_rows
    mod = torch.randn(6, N, device="cuda")
    e0 = torch.randn(nb, 6, N, device="cuda")
    x0 = torch.randn(M, N, device="cuda")

    def run(env):
        for k_, v_ in env.items():
            set_option(k_, v_)
        x = x0.clone()
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(bias),
                     bias_mode=ops.BIAS_N, gate0=ops.ptr(mod, 2 * N), gate1=ops.ptr(e0, 2 * N), gate1_stride=6 * N,
                     gate_rows=gate_rows, gate_const=0.5)
        x1 = x0.clone()
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x1), M, N, K, K, K, N, ops.EPI_RESID, gate_const=1.0)
        x2 = x0.clone()
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x2), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(bias),
                     bias_mode=ops.BIAS_N, gate0=ops.ptr(mod, 3 * N), gate_const=0.0)
        return x, x1, x2
    got = run({"OMH_GEMM_KERNEL": "w64", "OMH_GEMM_W64_R192": "1"})
    again = run({"OMH_GEMM_KERNEL": "w64", "OMH_GEMM_W64_R192": "1"})
    big = run({"OMH_GEMM_KERNEL": "w64", "OMH_GEMM_W64_R192": "0"})
    old = run({"OMH_GEMM_KERNEL": "8w"})
    for g, g2, b_, o in zip(got, again, big, old):
        assert torch.equal(g, o) and torch.equal(g, g2) and torch.equal(b_, o)
    ref = a.float() @ w.float().t()
    gate = (0.5 + mod[2][None] + e0[:, 2]).repeat_interleave(gate_rows, 0)[:M]
    assert rel_rms(got[0], x0 + (ref + bias) * gate) < 1e-5
    assert rel_rms(got[1], x0 + ref) < 1e-5


@pytest.mark.parametrize("M,N,K,S,gate_rows", [(1560, 1536, 8960, 4, 1560), (3120, 1536, 8960, 2, 1560), (300, 1536, 4096, 4, 150),
                                              (3000, 1600, 8960, 2, 1000), (780, 776, 4416, 3, 780)])
def test_gemm_split_k_for_few_row_long_contraction_products(ops, M, N, K, S, gate_rows, monkeypatch):
    """ABI v9: the contraction of a few-row, long-K product (FFN-down / FFN-up input gradient at one or two clips:
    model.py:272-274,328) in S slices on the fp32 256 x 192 stream + one combine launch (gated residual in place, out of
    place with the bf16 branch output, plain fp32 with bias).  Against the unsplit kernels (fp32 sums in another order:
    1e-6), against fp32 arithmetic, repeatable bit for bit; ragged M / N, a gate boundary inside the rows; the workspace
    query says 0 where nothing is split, and a workspace that is too small is refused."""
    import ctypes as C
    binding, omh = ops._lib, ops.lib
    for k_ in ("OMH_GEMM_KERNEL", "OMH_GEMM_TILE", "OMH_GEMM_SPLITK"):
        set_option(k_, None)
    torch.manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(N, device="cuda")
    nb = (M + gate_rows - 1) // gate_rows
    mod = torch.randn(6, N, device="cuda")
    e0 = torch.randn(nb, 6, N, device="cuda")
    x0 = torch.randn(M, N, device="cuda")

    def args(Cp, epi, **kw):
        g = binding.GemmArgs(ops.ptr(a), ops.ptr(w), Cp, M, N, K, K, K, N, 1, 0, 0, 0, epi, ops.BIAS_N, ops.ptr(bias),
                             None, None, 0, 1, 0.0, 0, None, None, 0, None, 0)
        for k_, v_ in kw.items():
            setattr(g, k_, v_)
        return g
    need = omh.omh_gemm_workspace_bytes(C.byref(args(ops.ptr(x0), ops.EPI_RESID)))
    assert need == S * ((M + 255) // 256 * 256) * N * 4
    assert omh.omh_gemm_workspace_bytes(C.byref(args(ops.ptr(x0), ops.EPI_F32))) == need
    assert omh.omh_gemm_workspace_bytes(C.byref(args(ops.ptr(x0), ops.EPI_BF16))) == 0
    small = torch.empty(need - 16, dtype=torch.uint8, device="cuda")
    rc = omh.omh_gemm_bf16(C.byref(args(ops.ptr(x0.clone()), ops.EPI_RESID, workspace=small.data_ptr(), workspace_bytes=need - 16)), None)
    assert rc == -1                                                   # OMH_E_BADARG

    def run(split):
        set_option("OMH_GEMM_SPLITK", "1" if split else "0")
        x = x0.clone()
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(bias),
                     bias_mode=ops.BIAS_N, gate0=ops.ptr(mod, 2 * N), gate1=ops.ptr(e0, 2 * N), gate1_stride=6 * N,
                     gate_rows=gate_rows, gate_const=0.5, split_k=True)
        x1 = torch.empty_like(x0)                                        # out of place + the bf16 branch output (training)
        y1 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(x1), M, N, K, K, K, N, ops.EPI_RESID, bias=ops.ptr(bias),
                     bias_mode=ops.BIAS_N, gate0=ops.ptr(mod, 3 * N), gate_const=0.0, c_in=ops.ptr(x0), aux=ops.ptr(y1), ldaux=N,
                     split_k=True)
        f, f0 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(f), M, N, K, K, K, N, ops.EPI_F32, bias=ops.ptr(bias), bias_mode=ops.BIAS_N,
                     split_k=True)
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(f0), M, N, K, K, K, N, ops.EPI_F32, split_k=True)
        nosplit = torch.empty(M, N, device="cuda")                       # a caller that does not ask: the unsplit kernels
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(nosplit), M, N, K, K, K, N, ops.EPI_F32)
        return x, x1, y1.float(), f, f0, nosplit
    got, again, old = run(True), run(True), run(False)
    set_option("OMH_GEMM_SPLITK", None)
    for g, g2, o in zip(got, again, old):
        assert torch.equal(g, g2)
        assert not torch.equal(g, o) or g is got[2] or g is got[5]      # (it did take the other path)
    assert torch.equal(got[5], old[5]) and torch.equal(got[5], old[4])
    ref = a.float() @ w.float().t()
    gate = (0.5 + mod[2][None] + e0[:, 2]).repeat_interleave(gate_rows, 0)[:M]
    assert rel_rms(got[0], old[0]) < 2e-6 and rel_rms(got[0], x0 + (ref + bias) * gate) < 1e-5
    assert rel_rms(got[1], old[1]) < 2e-6 and rel_rms(got[1], x0 + (ref + bias) * mod[3][None]) < 1e-5
    assert rel_rms(got[2], old[2]) < 1e-3 and rel_rms(got[2], ref + bias) < 4e-3      # bf16: the odd last-bit flip
    assert rel_rms(got[3], old[3]) < 2e-6 and rel_rms(got[3], ref + bias) < 2e-5
    assert rel_rms(got[4], ref) < 2e-5


@pytest.mark.parametrize("M,N,K", [(256, 192, 256), (1000, 776, 320), (6240, 1536, 1536), (6240, 1536, 4608), (4100, 1160, 448)])
def test_gemm_w64_narrow_streams(ops, M, N, K, monkeypatch):
    """The 256 x 192 fp32 / bf16 streams (the training step's M = 6 240 products: 200 tiles instead of 100 of 256 x 384):
    bit for bit against the 8-wave kernels, with and without bias, ragged shapes, odd and even k-step counts."""
    torch.manual_seed(M + K)
    a = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(N, device="cuda")

    def run(env):
        for k_ in ("OMH_GEMM_KERNEL", "OMH_GEMM_W64_N192"):
            set_option(k_, None)
        for k_, v_ in env.items():
            set_option(k_, v_)
        return (ops.gemm(a, w, bias=bias, epilogue=ops.EPI_F32), ops.gemm(a, w, epilogue=ops.EPI_F32),
                ops.gemm(a, w, bias=bias, epilogue=ops.EPI_BF16), ops.gemm(a, w, epilogue=ops.EPI_BF16))
    got, again, old = run({"OMH_GEMM_W64_N192": "1"}), run({"OMH_GEMM_W64_N192": "1"}), run({"OMH_GEMM_KERNEL": "8w"})
    for g, g2, o in zip(got, again, old):
        assert torch.equal(g, o) and torch.equal(g, g2)
    assert rel_rms(got[0], a.float() @ w.float().t() + bias) < 2e-5
    dflt = run({})                                                  # the default dispatch gives the same bits, whichever kernel
    assert all(torch.equal(d_, o) for d_, o in zip(dflt, old))


@pytest.mark.parametrize("M,N,K", [(256, 384, 256), (1000, 776, 320), (6240, 8960, 1536), (4100, 1160, 448)])
def test_gemm_w64_gelu_backward_stream(ops, M, N, K, monkeypatch):
    """Round 4: the FFN dgrad product with GELU' in its epilogue (OMH_EPI_GELU_BWD_BF16: du_pre = (dy W2) gelu'(u_pre)) on
    the 256 x 384 stream — bit for bit against the 8-wave kernel (gelu_tanh_grad operation by operation), and against
    autograd's derivative of the tanh GELU."""
    torch.manual_seed(M + N)
    a = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    pre = _bf(torch.randn(M, N, device="cuda") * 1.5)

    set_option("OMH_GEMM_W64_GBWD", "1")                    # opt-in: measured equal to the 8-wave kernel, not the default

    def run(kernel):
        set_option("OMH_GEMM_KERNEL", kernel)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), M, N, K, K, K, N, ops.EPI_GELU_BWD_BF16, aux=ops.ptr(pre), ldaux=N)
        return out
    got, again, old = run("w64"), run("w64"), run("8w")
    assert torch.equal(got, old) and torch.equal(got, again)
    xf = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(xf, approximate="tanh").sum().backward()
    ref = (a.float() @ w.float().t()) * xf.grad
    assert rel_rms(got.float(), ref) < 5e-3
    set_option("OMH_GEMM_KERNEL", None)
    assert torch.equal(run_default(ops, a, w, pre, M, N, K), old)      # whichever kernel the dispatch picks: the same bits


@pytest.mark.parametrize("M,N,K", [(256, 384, 256), (1000, 776, 320), (6240, 8960, 1536), (4100, 1160, 448), (1560, 8960, 1536)])
def test_gemm_w64_gelu_stream_with_the_pre_activation(ops, M, N, K, monkeypatch):
    """Round 4 (third part): the FFN-up projection of a training forward — C = bf16(gelu_tanh(a W^T + b)) AND the
    pre-activation aux = bf16(a W^T + b) that the GELU backward reads (OMH_EPI_GELU_BF16 with aux, ABI v5) — on the
    256 x 384 stream ("geluaux": the gelu stream with a second store per run): both outputs bit for bit against the 8-wave
    kernel, the guard band behind aux untouched, ragged M / N, several tiles per persistent workgroup."""
    torch.manual_seed(M + N + 1)
    a = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(N, device="cuda")

    def run(env):
        for k_ in ("OMH_GEMM_KERNEL", "OMH_GEMM_W64_GAUX"):
            set_option(k_, None)
        for k_, v_ in env.items():
            set_option(k_, v_)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        pre = torch.full((M + 8, N), 7.0, device="cuda", dtype=torch.bfloat16)         # 8 guard rows
        ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), M, N, K, K, K, N, ops.EPI_GELU_BF16, bias=ops.ptr(bias),
                     bias_mode=ops.BIAS_N, aux=ops.ptr(pre), ldaux=N)
        assert bool((pre[M:] == 7.0).all())
        return out, pre[:M].clone()
    got, again, old, dflt = run({"OMH_GEMM_KERNEL": "w64"}), run({"OMH_GEMM_KERNEL": "w64"}), run({"OMH_GEMM_KERNEL": "8w"}), run({})
    off = run({"OMH_GEMM_W64_GAUX": "0"})
    for g, g2, o, d_, f in zip(got, again, old, dflt, off):
        assert torch.equal(g, o) and torch.equal(g, g2) and torch.equal(d_, o) and torch.equal(f, o)
    ref = a.float() @ w.float().t() + bias
    assert rel_rms(got[1].float(), ref) < 4e-3
    assert rel_rms(got[0].float(), torch.nn.functional.gelu(ref, approximate="tanh")) < 5e-3


def run_default(ops, a, w, pre, M, N, K):
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), M, N, K, K, K, N, ops.EPI_GELU_BWD_BF16, aux=ops.ptr(pre), ldaux=N)
    return out


@pytest.mark.parametrize("M,N,K,ldc", [(256, 384, 256, 384), (1000, 776, 320, 832), (1536, 32760, 1536, 32768), (1536, 1560, 1536, 1600)])
def test_gemm_w64_per_row_bias_stream(ops, M, N, K, ldc, monkeypatch):
    """Round 4: bf16 output with a per-ROW bias (OMH_BIAS_M: V^T = Wv h^T + bv, the operands-swapped projection of
    model.py:152-153) on the 256 x 384 stream — bit for bit against the 8-wave kernel; at the real shape the last tile
    column (120 of 32 760 columns) goes to the 8-wave kernel as a second launch so that the stream's tiles are exactly
    two rounds of the persistent grid; pad columns [N, ldc) stay untouched."""
    torch.manual_seed(M + N)
    w = _bf(torch.randn(M, K, device="cuda") / math.sqrt(K))
    h = _bf(torch.randn(N, K, device="cuda"))
    bias = torch.randn(M, device="cuda")

    def run(env):
        for k_ in ("OMH_GEMM_KERNEL", "OMH_GEMM_W64_BF16M"):
            set_option(k_, None)
        for k_, v_ in env.items():
            set_option(k_, v_)
        out = torch.full((M, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm_raw(ops.ptr(w), ops.ptr(h), ops.ptr(out), M, N, K, K, K, ldc, ops.EPI_BF16, bias=ops.ptr(bias), bias_mode=ops.BIAS_M)
        return out
    got, again, old, dflt = run({"OMH_GEMM_W64_BF16M": "1"}), run({"OMH_GEMM_W64_BF16M": "1"}), run({"OMH_GEMM_KERNEL": "8w"}), run({})
    assert torch.equal(got, old) and torch.equal(got, again) and torch.equal(dflt, old)
    assert bool((got[:, N:] == 7.0).all())
    ref = w.float() @ h.float().t() + bias[:, None]
    assert rel_rms(got[:, :N].float(), ref) < 4e-3


def test_gemm_w64_random_shapes_equal_the_8_wave_kernel(ops, monkeypatch):
    """Seeded sweep: 20 random (M, N, K, epilogue, bias, gate layout) within the stream kernel's domain, whole outputs
    bit for bit against the 8-wave kernels (one to a few tiles per workgroup of the persistent grid, ragged M and N,
    odd and even numbers of k steps)."""
    import random
    rng = random.Random(7)
    for case in range(20):
        M, N, K = rng.randint(256, 5000), 8 * rng.randint(12, 300), 64 * rng.randint(4, 20)
        epi = rng.choice((ops.EPI_F32, ops.EPI_BF16, ops.EPI_GELU_BF16, ops.EPI_RESID))
        torch.manual_seed(case)
        a = _bf(torch.randn(M, K, device="cuda"))
        w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
        bias = torch.randn(N, device="cuda") if rng.random() < 0.7 else None
        gate_rows = rng.randint(128, M)
        nb = (M + gate_rows - 1) // gate_rows
        g0 = torch.randn(N, device="cuda") if rng.random() < 0.5 else None
        g1 = torch.randn(nb, 3, N, device="cuda") if rng.random() < 0.7 else None
        x0 = torch.randn(M, N, device="cuda")
        outs = {}
        for kernel in ("w64", "8w"):
            set_option("OMH_GEMM_KERNEL", kernel)
            out = x0.clone() if epi == ops.EPI_RESID else torch.empty(
                M, N, device="cuda", dtype=torch.float32 if epi == ops.EPI_F32 else torch.bfloat16)
            ops.gemm_raw(ops.ptr(a), ops.ptr(w), ops.ptr(out), M, N, K, K, K, N, epi,
                         bias=ops.ptr(bias) if bias is not None else None,
                         bias_mode=ops.BIAS_N if bias is not None else ops.BIAS_NONE,
                         gate0=ops.ptr(g0) if (g0 is not None and epi == ops.EPI_RESID) else None,
                         gate1=ops.ptr(g1, N) if (g1 is not None and epi == ops.EPI_RESID) else None,
                         gate1_stride=3 * N, gate_rows=gate_rows, gate_const=0.25)
            outs[kernel] = out
        assert torch.equal(outs["w64"], outs["8w"]), (case, M, N, K, epi)


def test_gemm_w64_is_the_default_on_the_large_shapes(ops, monkeypatch):
    """Unset OMH_GEMM_KERNEL: >= 256 tiles of 256 x 384 -> the stream kernel; its output is that of the forced call and
    (the kernels agree bit for bit) of the 8-wave kernel."""
    set_option("OMH_GEMM_KERNEL", None)
    set_option("OMH_GEMM_TILE", None)
    M, N, K = 32760, 1536, 256
    a = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    d = ops.gemm(a, w)
    set_option("OMH_GEMM_KERNEL", "8w")
    assert torch.equal(d, ops.gemm(a, w))


@pytest.mark.parametrize("M,d,K", [(520, 384, 384), (1032, 768, 512), (32760, 1536, 1536), (8200, 1536, 1536)])
def test_gemm_fused_qkv_projection_with_transposed_v(ops, M, d, K):
    """ABI v10, OMH_EPI_BF16_SPLIT_T: q | k | v as ONE product over the concatenated weights on the 256 x 384 stream —
    columns below n_split = 2 d to C (the BF16 stream), the V third through the operand-swapped stream, stored TRANSPOSED
    into V^T [d, ld] with a per-row bias.  Bit for bit against the two products it replaces (GEMM_QKV = 0 routes the
    same call through them), against fp32 arithmetic, ragged M (rows past M never written: the pad columns of V^T and
    the guard rows of q | k keep their sentinel), several tiles per persistent workgroup at the real size."""
    torch.manual_seed(M + d)
    h = _bf(torch.randn(M, K, device="cuda"))
    w = _bf(torch.randn(3 * d, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(3 * d, device="cuda")
    ld = (M + 63) // 64 * 64 + 64

    def run(mode):
        set_option("GEMM_QKV", mode)
        qk = torch.full((M + 8, 2 * d), 7.0, dtype=torch.bfloat16, device="cuda")
        vt = torch.full((d + 8, ld), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.gemm_raw(ops.ptr(h), ops.ptr(w), ops.ptr(qk), M, 3 * d, K, K, K, 2 * d, ops.EPI_BF16_SPLIT_T, bias=ops.ptr(bias),
                     bias_mode=ops.BIAS_N, aux=ops.ptr(vt), ldaux=ld, n_split=2 * d)
        return qk, vt
    got, again, old = run("1"), run("1"), run("0")
    dflt = run(None)
    for a_, b_ in ((got, again), (got, old), (got, dflt)):
        assert torch.equal(a_[0], b_[0]) and torch.equal(a_[1], b_[1])
    qk, vt = got
    assert float((qk[M:].float() - 7.0).abs().max()) == 0 and float((vt[:d, M:].float() - 7.0).abs().max()) == 0
    assert float((vt[d:].float() - 7.0).abs().max()) == 0
    ref = h.float() @ w.float().t() + bias
    assert rel_rms(qk[:M].float(), ref[:, :2 * d]) < 4e-3
    assert rel_rms(vt[:d, :M].float(), ref[:, 2 * d:].t()) < 4e-3
    # without a bias
    set_option("GEMM_QKV", "1")
    qk2 = torch.empty(M, 2 * d, dtype=torch.bfloat16, device="cuda")
    vt2 = torch.zeros(d, ld, dtype=torch.bfloat16, device="cuda")
    ops.gemm_raw(ops.ptr(h), ops.ptr(w), ops.ptr(qk2), M, 3 * d, K, K, K, 2 * d, ops.EPI_BF16_SPLIT_T, aux=ops.ptr(vt2),
                 ldaux=ld, n_split=2 * d)
    assert rel_rms(vt2[:, :M].float(), (h.float() @ w.float().t())[:, 2 * d:].t()) < 4e-3


def _attn_ref(q, k, v, k_lens, scale):
    B, Lq, H, D = q.shape
    out = torch.zeros(B, Lq, H, D, device=q.device)
    for b in range(B):
        kl = k.shape[1] if k_lens is None else int(k_lens[b])
        if kl == 0:
            continue
        s = torch.einsum("qhd,khd->hqk", q[b].float(), k[b, :kl].float()) * scale
        out[b] = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), v[b, :kl].float())
    return out


@pytest.mark.parametrize("B,H,Lq,Lk,klens", [
    (1, 1, 128, 64, None), (2, 2, 200, 200, [200, 77]), (1, 12, 1560, 1560, [1560]),
    (2, 3, 130, 512, [37, 512]), (1, 2, 64, 320, [257]), (2, 1, 100, 64, [0, 5])])
@pytest.mark.parametrize("kernel", ["base", "w64"])
def test_flash_attention(ops, B, H, Lq, Lk, klens, kernel, monkeypatch):
    set_option("OMH_ATTN_KERNEL", kernel)      # both kernels on every shape (ragged rows/keys, empty rows)
    torch.manual_seed(Lq + Lk)
    D = 128
    q = _bf(torch.randn(B, Lq, H, D, device="cuda"))
    k = _bf(torch.randn(B, Lk, H, D, device="cuda"))
    v = _bf(torch.randn(B, Lk, H, D, device="cuda"))
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros(B, H * D, Lp, dtype=torch.bfloat16, device="cuda")
    vt[:, :, :Lk] = v.reshape(B, Lk, H * D).transpose(1, 2)
    kl = None if klens is None else torch.tensor(klens, dtype=torch.int32, device="cuda")
    out = ops.flash_attn(q, k, vt, kl)
    ref = _attn_ref(q, k, v, klens, D ** -0.5)
    assert torch.isfinite(out.float()).all()
    # P is rounded to bf16 before P.V and the output to bf16: ~2^-9 relative each
    assert rel_rms(out.float(), ref) < 8e-3
    assert float((out.float() - ref).abs().max()) < 3e-2


def test_flash_attention_q_lens(ops, wan_model_mod):
    """ABI v10 / VERDICT round 4 item 10: flash_attention(q_lens=...) (attention.py:24-60,79) — query rows past a sample's
    length are pad rows of the reference's packed batch: zeros here; the rows inside are what the call without q_lens
    gives, bit for bit; also at a size whose default dispatch would take the long-sequence kernel, and through the
    reference-signature wrapper.  dropout_p is rejected loudly (causal / window_size: the next test)."""
    attn_mod = __import__("importlib").import_module(wan_model_mod.__name__.rsplit(".", 1)[0] + ".attention")
    torch.manual_seed(3)
    D = 128
    for (B, H, Lq, Lk, qlens, klens) in ((2, 2, 200, 150, [200, 77], [150, 60]), (3, 1, 129, 64, [0, 129, 1], None),
                                         (1, 12, 5601, 5601, [4000], [5000])):
        q = _bf(torch.randn(B, Lq, H, D, device="cuda"))
        k = _bf(torch.randn(B, Lk, H, D, device="cuda"))
        v = _bf(torch.randn(B, Lk, H, D, device="cuda"))
        Lp = (Lk + 63) // 64 * 64
        vt = torch.zeros(B, H * D, Lp, dtype=torch.bfloat16, device="cuda")
        vt[:, :, :Lk] = v.reshape(B, Lk, H * D).transpose(1, 2)
        kl = None if klens is None else torch.tensor(klens, dtype=torch.int32, device="cuda")
        ql = torch.tensor(qlens, dtype=torch.int32, device="cuda")
        out = ops.flash_attn(q, k, vt, kl, q_lens=ql)
        ref = _attn_ref(q, k, v, klens, D ** -0.5)
        set_option("OMH_ATTN_KERNEL", "base")
        plain = ops.flash_attn(q, k, vt, kl)
        set_option("OMH_ATTN_KERNEL", None)
        for b in range(B):
            n = qlens[b]
            assert torch.equal(out[b, :n], plain[b, :n])
            assert float(out[b, n:].float().abs().sum()) == 0.0
            if n:
                assert rel_rms(out[b, :n].float(), ref[b, :n]) < 8e-3
        w = attn_mod.flash_attention(q, k, v, q_lens=ql, k_lens=kl)
        assert torch.equal(w, out)
    with pytest.raises(NotImplementedError):
        attn_mod.flash_attention(q, k, v, dropout_p=0.1)


def _band_ref(q, k, v, qlens, klens, scale, left, right):
    """fp32 softmax attention under flash-attn's bottom-right aligned band (attention.py:96-127 -> flash_attn_varlen_func's
    causal / window_size): query i sees key j iff i + klen - qlen - left <= j <= i + klen - qlen + right (side < 0:
    unbounded); rows past qlen and rows whose band is empty are zero."""
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    out = torch.zeros(B, Lq, H, D, dtype=torch.float32, device=q.device)
    for b in range(B):
        ql = Lq if qlens is None else qlens[b]
        kl = Lk if klens is None else klens[b]
        if ql == 0 or kl == 0:
            continue
        i = torch.arange(ql, device=q.device)[:, None] + (kl - ql)
        j = torch.arange(kl, device=q.device)[None, :]
        ok = torch.ones(ql, kl, dtype=torch.bool, device=q.device)
        if left >= 0:
            ok &= j >= i - left
        if right >= 0:
            ok &= j <= i + right
        sc = torch.einsum("qhd,khd->hqk", q[b, :ql].float(), k[b, :kl].float()) * scale
        sc = sc.masked_fill(~ok[None], float("-inf"))
        p = torch.softmax(sc, dim=-1)
        p = torch.nan_to_num(p, nan=0.0)                          # a row with an empty band
        out[b, :ql] = torch.einsum("hqk,khd->qhd", p, v[b, :kl].float())
    return out


@pytest.mark.parametrize("case", [
    # B, H, Lq, Lk, qlens, klens, causal, window
    (2, 2, 200, 200, None, None, True, (-1, -1)),                 # square causal
    (2, 3, 130, 333, [130, 77], [333, 150], True, (-1, -1)),      # more keys than queries: the diagonal ends bottom-right
    (1, 2, 300, 100, None, [90], True, (-1, -1)),                 # fewer keys than queries: the first rows see nothing
    (2, 2, 257, 257, None, None, False, (40, 25)),                # a band across tiles and workgroups
    (1, 4, 1000, 1000, [900], [950], False, (128, 128)),          # the band skips whole key tiles on both sides
    (1, 2, 500, 700, None, None, False, (0, -1)),                 # left bound only
    (1, 12, 5601, 5601, None, None, True, (64, -1)),              # a size whose full-attention call takes the long-sequence stream
])
def test_flash_attention_causal_and_window(ops, wan_model_mod, case):
    """ABI v12 / VERDICT round 5 item 10: flash_attention(causal=, window_size=) (attention.py:24-60,96-127) against an fp32
    masked softmax under flash-attn's bottom-right aligned band, through the reference-signature wrapper; (-1, -1)
    through the same argument is the plain call bit for bit."""
    attn_mod = __import__("importlib").import_module(wan_model_mod.__name__.rsplit(".", 1)[0] + ".attention")
    B, H, Lq, Lk, qlens, klens, causal, window = case
    torch.manual_seed(11)
    D = 128
    q = _bf(torch.randn(B, Lq, H, D, device="cuda"))
    k = _bf(torch.randn(B, Lk, H, D, device="cuda"))
    v = _bf(torch.randn(B, Lk, H, D, device="cuda"))
    ql = None if qlens is None else torch.tensor(qlens, dtype=torch.int32, device="cuda")
    kl = None if klens is None else torch.tensor(klens, dtype=torch.int32, device="cuda")
    with torch.no_grad():
        out = attn_mod.flash_attention(q, k, v, q_lens=ql, k_lens=kl, causal=causal, window_size=window)
    left, right = window[0], (0 if causal else window[1])
    ref = _band_ref(q, k, v, qlens, klens, D ** -0.5, left, right)
    assert torch.isfinite(out.float()).all()
    live = ref.abs().sum(-1) > 0
    assert float(out.float()[~live].abs().sum()) == 0.0          # rows past q_lens / rows with an empty band: exactly zero
    assert rel_rms(out.float(), ref) < 8e-3
    assert float((out.float() - ref).abs().max()) < 3e-2
    with torch.no_grad():
        full = attn_mod.flash_attention(q, k, v, q_lens=ql, k_lens=kl, window_size=(-1, -1))
        wide = attn_mod.flash_attention(q, k, v, q_lens=ql, k_lens=kl, window_size=(Lq + Lk, Lq + Lk))
    if Lq <= 2048:                                               # (the long shape's plain call runs the other kernel)
        assert torch.equal(full, wide)                           # a band wider than the problem = full attention, same bits
    qg = q.clone().requires_grad_(True)
    with pytest.raises(NotImplementedError):
        attn_mod.flash_attention(qg, k, v, causal=True)


def test_flash_attention_long_sequence_dispatch(ops):
    """A shape that takes the long-sequence kernel through the normal dispatch (>= 512 workgroups of 256 rows),
    ragged in both rows and keys, two samples with different key lengths."""
    torch.manual_seed(77)
    B, H, Lq, Lk, D = 2, 12, 5601, 5601, 128
    klens = [5601, 4000]
    q = _bf(torch.randn(B, Lq, H, D, device="cuda"))
    k = _bf(torch.randn(B, Lk, H, D, device="cuda"))
    v = _bf(torch.randn(B, Lk, H, D, device="cuda"))
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros(B, H * D, Lp, dtype=torch.bfloat16, device="cuda")
    vt[:, :, :Lk] = v.reshape(B, Lk, H * D).transpose(1, 2)
    out = ops.flash_attn(q, k, vt, torch.tensor(klens, dtype=torch.int32, device="cuda"))
    ref = _attn_ref(q, k, v, klens, D ** -0.5)
    assert rel_rms(out.float(), ref) < 8e-3
    assert float((out.float() - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("kernel", ["base", "w64"])
def test_flash_attention_peaked_rows(ops, kernel, monkeypatch):
    """Forces large online-softmax rescales: one key dominates late in the sequence."""
    set_option("OMH_ATTN_KERNEL", kernel)
    torch.manual_seed(5)
    B, H, L, D = 1, 2, 384, 128
    q = _bf(torch.randn(B, L, H, D, device="cuda"))
    k = _bf(torch.randn(B, L, H, D, device="cuda"))
    v = _bf(torch.randn(B, L, H, D, device="cuda"))
    k[0, 300] = q[0, 17] * 4.0      # row 17 spikes at key 300 (tile 4)
    k[0, 70] = q[0, 90] * 3.0
    vt = v.reshape(B, L, H * D).transpose(1, 2).contiguous()
    out = ops.flash_attn(q, k, vt, None)
    ref = _attn_ref(q, k, v, None, D ** -0.5)
    assert rel_rms(out.float(), ref) < 8e-3


def test_flash_attention_w64_prescaled_q_lse_and_late_rescale(ops, monkeypatch):
    """The asm-owned 4 x 64 long-sequence kernel (csrc/attention_w64.hip; the shipped stream — the earlier variants are
    built for the generator's ablation runs only, round 5) as the model drives
    it: q already multiplied by softmax_scale * log2(e) by the norm kernel (omh_attn_args.q_prescaled), log-sum-exp
    requested, ragged rows and keys, masked keys, and a key that dominates LATE in the sequence with large scores so
    that the deferred running-max update (rescale of the AGPR accumulators) runs after hundreds of tiles.  Against
    fp32 softmax attention on the same bf16 operands, and bit for bit against itself."""
    set_option("OMH_ATTN_KERNEL", "w64")
    D, LOG2E = 128, 1.4426950408889634
    g = torch.Generator(device="cuda").manual_seed(11)
    for (B, H, Lq, Lk, klens, amp) in ((1, 2, 300, 200, None, 1.0), (2, 2, 777, 1000, [1000, 333], 1.0),
                                       (1, 2, 512, 4096, None, 4.0), (1, 1, 64, 100, [37], 1.0)):
        q = _bf(torch.randn(B, Lq, H, D, device="cuda", generator=g) * amp * (D ** -0.5 * LOG2E))
        k = _bf(torch.randn(B, Lk, H, D, device="cuda", generator=g) * amp)
        v = _bf(torch.randn(B, Lk, H, D, device="cuda", generator=g))
        if amp > 1:
            k[:, Lk - 70] = _bf(q[:, 5].float() / (D ** -0.5 * LOG2E))       # score ~ 16 |q|^2 / sqrt(128) at tile 62
        Lp = (Lk + 63) // 64 * 64
        vt = torch.zeros(B, H * D, Lp, dtype=torch.bfloat16, device="cuda")
        vt[:, :, :Lk] = v.reshape(B, Lk, H * D).transpose(1, 2)
        kl = None if klens is None else torch.tensor(klens, dtype=torch.int32, device="cuda")

        def run():
            out = torch.full((B, Lq, H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
            lse = torch.full((B, H, Lq), float("nan"), dtype=torch.float32, device="cuda")
            ops.flash_attn_raw(ops.ptr(q), ops.ptr(k), ops.ptr(vt), ops.ptr(out), ops.ptr(kl) if kl is not None else None,
                               B, H, Lq, Lk, q.stride(0), q.stride(1), k.stride(0), k.stride(1), vt.stride(0),
                               out.stride(0), out.stride(1), vt.stride(1), D ** -0.5, lse=ops.ptr(lse), q_prescaled=1)
            return out, lse
        out, lse = run()
        s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) / LOG2E              # natural-log scores
        if klens is not None:
            for b_, n in enumerate(klens):
                s[b_, :, :, n:] = float("-inf")
        ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float())
        assert torch.isfinite(out.float()).all()
        assert rel_rms(out.float(), ref) < 8e-3 and float((out.float() - ref).abs().max()) < 3e-2
        assert float((lse - torch.logsumexp(s, -1)).abs().max()) < 5e-3
        out2, lse2 = run()
        assert torch.equal(out, out2) and torch.equal(lse, lse2)


def test_layernorm_modulate(ops):
    torch.manual_seed(2)
    # a wave takes 4 consecutive rows: batch boundaries inside them (50 % 4 = 2; 7 rows per batch), a ragged last wave
    for B, S, d in ((2, 50, 1536), (3, 7, 1536), (1, 1, 256)):
        x = torch.randn(B * S, d, device="cuda") * 3 + 0.5
        mod = torch.randn(6, d, device="cuda")
        e0 = torch.randn(B, 6, d, device="cuda")
        y = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
        ops.layernorm_modulate_raw(ops.ptr(x), ops.ptr(y), B * S, d, 1e-6, 1.0, ops.ptr(mod, d), ops.ptr(e0, d), 6 * d,
                                   ops.ptr(mod, 0), ops.ptr(e0, 0), 6 * d, S)
        xh = torch.nn.functional.layer_norm(x, (d,), eps=1e-6)
        ref = xh * (1 + (mod[1][None] + e0[:, 1]).repeat_interleave(S, 0)) + (mod[0][None] + e0[:, 0]).repeat_interleave(S, 0)
        assert rel_rms(y.float(), ref) < 4e-3
    # rows per wave (1 / 2 / 4 / 8 / 16, picked by the row count; round 5 added 8 and 16): the same bits whichever runs —
    # batch boundaries inside a wave's rows (101 rows per batch), a ragged last wave
    B, S, d = 3, 101, 1536
    x = torch.randn(B * S, d, device="cuda") * 3 + 0.5
    mod, e0 = torch.randn(6, d, device="cuda"), torch.randn(B, 6, d, device="cuda")
    outs = []
    for rpw in ("1", "2", "4", "8", "16"):
        set_option("LN_RPW", rpw)
        y = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
        ops.layernorm_modulate_raw(ops.ptr(x), ops.ptr(y), B * S, d, 1e-6, 1.0, ops.ptr(mod, d), ops.ptr(e0, d), 6 * d,
                                   ops.ptr(mod, 0), ops.ptr(e0, 0), 6 * d, S)
        outs.append(y)
    set_option("LN_RPW", None)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    B, S, d = 2, 50, 1536
    x = torch.randn(B * S, d, device="cuda") * 3 + 0.5
    w, b = torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
    y2 = ops.layernorm_modulate(x, 1e-6, 0.0, mul0=w, add0=b)
    assert rel_rms(y2.float(), torch.nn.functional.layer_norm(x, (d,), w, b, 1e-6)) < 4e-3


def test_rmsnorm_rope_matches_oracle(ops):
    from oracle import wan_dit_oracle as O
    torch.manual_seed(3)
    B, S, N, D = 2, 30, 2, 128
    d = N * D
    grids = [(2, 3, 4), (1, 5, 6)]
    x = torch.randn(B * S, 2 * d, device="cuda")
    w = torch.rand(d, device="cuda") + 0.5
    ang = O.rope_table(D)
    cos, sin = torch.cos(ang).float().cuda(), torch.sin(ang).float().cuda()
    grid = torch.tensor(grids, dtype=torch.int32, device="cuda")
    y = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
    ops.rmsnorm_rope_raw(ops.ptr(x, d), 2 * d, ops.ptr(y), B * S, d, ops.ptr(w), 1e-6, 1, ops.ptr(cos), ops.ptr(sin),
                         1024, D, ops.ptr(grid), S)
    xc = x[:, d:].cpu()
    ref = O.rope_apply(O.rms_norm(xc, w.cpu(), 1e-6).view(B, S, N, D), grids, ang).view(B * S, d)
    assert rel_rms(y.float(), ref) < 4e-3
    # no rope, no weight
    y2 = ops.rmsnorm_rope(x[:, :d], None, 1e-6)
    assert rel_rms(y2.float(), O.rms_norm(x[:, :d].cpu(), torch.ones(d), 1e-6)) < 4e-3


@pytest.mark.parametrize("rows,d,S", [(60, 256, 30), (3120, 1536, 1560), (1001, 5120, 1001), (9362, 1536, 4681)])
def test_rmsnorm_rope_pair_equals_two_launches(ops, rows, d, S):
    """ABI v9: q and k of the self-attention normalised + rotated out of the fused q|k projection in ONE launch (two column
    segments, each with its own gain, output and output scale): bit for bit the two single-segment launches; without gains
    and without RoPE too; ragged row count (rows % 4 != 0).  Round 6: also the one-wave-per-row form that long inputs take
    (RMS_PAIR_ROW = 1 forces it at any row count; 9 362 rows take it by themselves)."""
    from oracle import wan_dit_oracle as O
    torch.manual_seed(rows)
    D = 128
    B = rows // S
    qk = (torch.randn(rows, 2 * d, device="cuda") * 1.3).bfloat16()
    wq, wk = torch.rand(d, device="cuda") + 0.5, torch.rand(d, device="cuda") + 0.5
    ang = O.rope_table(D)
    cos, sin = torch.cos(ang).float().cuda(), torch.sin(ang).float().cuda()
    gs = {30: (2, 3, 4), 1560: (1, 30, 52), 1001: (1, 25, 40), 4681: (3, 30, 52)}[S]
    grid = torch.tensor([gs] * B, dtype=torch.int32, device="cuda")
    for gains, rope in ((True, True), (False, True), (True, False)):
        rk = (ops.ptr(cos), ops.ptr(sin), 1024, D, ops.ptr(grid), S) if rope else (None, None, 0, D, None, 0)
        w0, w1 = (ops.ptr(wq), ops.ptr(wk)) if gains else (None, None)
        q0, k0 = torch.empty(rows, d, dtype=torch.bfloat16, device="cuda"), torch.empty(rows, d, dtype=torch.bfloat16, device="cuda")
        ops.rmsnorm_rope_bf16_raw(ops.ptr(qk), 2 * d, ops.ptr(q0), rows, d, w0, 1e-6, 1, *rk, out_scale=0.1275)
        ops.rmsnorm_rope_bf16_raw(ops.ptr(qk, d), 2 * d, ops.ptr(k0), rows, d, w1, 1e-6, 1, *rk, out_scale=1.0)
        for form in (None, "0", "1"):                          # the shipped dispatch, the two-workgroup form, one wave per row
            set_option("RMS_PAIR_ROW", form)
            q1, k1 = torch.full_like(q0, 3.0), torch.full_like(k0, 3.0)
            ops.rmsnorm_rope_bf16_pair_raw(ops.ptr(qk), 2 * d, d, ops.ptr(q1), ops.ptr(k1), rows, d, w0, w1, 1e-6, 1, *rk,
                                           out_scale0=0.1275, out_scale1=1.0)
            assert torch.equal(q1, q0) and torch.equal(k1, k0), (gains, rope, form)
        set_option("RMS_PAIR_ROW", None)
    assert float(q0.float().abs().mean()) > 1e-3 and not torch.equal(q0, k0)


def test_patchify_unpatchify_dense_sinusoid(ops):
    from oracle import wan_dit_oracle as O
    torch.manual_seed(4)
    x = torch.randn(16, 2, 6, 8, device="cuda")
    tok = ops.patchify(x, (1, 2, 2), 64)
    w = torch.randn(32, 16, 1, 2, 2, device="cuda")
    ref = torch.nn.functional.conv3d(x[None].to(torch.bfloat16).float(), w, stride=(1, 2, 2)).flatten(2).transpose(1, 2)[0]
    got = tok.float() @ w.flatten(1).t()
    assert rel_rms(got, ref) < 1e-5
    head = torch.randn(2 * 3 * 4, 64, device="cuda")
    u = head.cpu().view(2, 3, 4, 1, 2, 2, 16)
    refu = torch.einsum("fhwpqrc->cfphqwr", u).reshape(16, 2, 6, 8)
    assert torch.equal(ops.unpatchify(head, 16, (2, 3, 4), (1, 2, 2)).cpu(), refu)
    t = torch.tensor([0., 1., 999., 1000.], device="cuda")
    s = ops.sinusoidal_embedding(t, 256)
    assert float((s.cpu() - O.sinusoidal_embedding_1d(256, t.cpu()).float()).abs().max()) < 1e-6
    xx, W, b = torch.randn(3, 300, device="cuda"), torch.randn(70, 300, device="cuda"), torch.randn(70, device="cuda")
    y = ops.dense_f32(xx, W, b, 1, 1)
    refy = torch.nn.functional.silu(torch.nn.functional.silu(xx) @ W.t() + b)
    assert rel_rms(y, refy) < 1e-5
    c = torch.randn(1000, device="cuda")
    assert torch.equal(ops.cast_bf16(c), c.to(torch.bfloat16))


@pytest.mark.parametrize("kernel", ["base", "w64"])
def test_flash_attention_is_bitwise_repeatable(ops, kernel, monkeypatch):
    """Same inputs, same bits, every launch.  The base kernel once took its row max through an inline-asm v_max3
    that hipcc's hazard recognizer does not see: issued right behind the last K.Q^T MFMA it sometimes read scores
    missing their last k-slice — a valid softmax offset, but a different one from run to run (outputs differing
    in the last bf16 bit, a 1.3B forward at S=1560 differing by 0.017 between two runs)."""
    set_option("OMH_ATTN_KERNEL", kernel)
    g = torch.Generator(device="cuda").manual_seed(5)
    for (Lq, Lk, klen) in ((1560, 1560, 1560), (1560, 512, 120)):
        H = 12
        q = torch.randn(1, Lq, H, 128, device="cuda", generator=g).bfloat16()
        k = torch.randn(1, Lk, H, 128, device="cuda", generator=g).bfloat16()
        Lp = (Lk + 63) // 64 * 64
        vt = torch.zeros(1, H * 128, Lp, device="cuda", dtype=torch.bfloat16)
        vt[:, :, :klen] = torch.randn(1, H * 128, klen, device="cuda", generator=g).bfloat16()
        kl = torch.tensor([klen], dtype=torch.int32, device="cuda")
        ref = ops.flash_attn(q, k, vt, k_lens=kl).clone()
        for _ in range(8):
            assert torch.equal(ops.flash_attn(q, k, vt, k_lens=kl), ref)


@pytest.mark.parametrize("B,H,Lq,Lk,klens", [
    (1, 2, 64, 64, None),
    (2, 3, 200, 131, [131, 77]),          # ragged rows / keys, masked keys
    (2, 2, 97, 512, [512, 1]),            # cross-attention shape: one sample with a single valid key
    (1, 12, 1560, 1560, [1560]),          # BASELINE config 3 self-attention
    (2, 1, 40, 70, [0, 70]),              # a sample without keys: zero gradients, no NaN
])
def test_flash_attention_backward(ops, B, H, Lq, Lk, klens):
    """omh_flash_attn_bwd_d128 against autograd through a masked fp32 softmax attention on the same bf16 inputs."""
    g = torch.Generator(device="cuda").manual_seed(Lq * 7 + Lk)
    d = H * 128
    q = torch.randn(B * Lq, d, device="cuda", generator=g).bfloat16()
    k = torch.randn(B * Lk, d, device="cuda", generator=g).bfloat16()
    v = torch.randn(B * Lk, d, device="cuda", generator=g).bfloat16()
    do = torch.randn(B * Lq, d, device="cuda", generator=g).bfloat16()
    kl = None if klens is None else torch.tensor(klens, dtype=torch.int32, device="cuda")
    scale = 128 ** -0.5
    # forward on the product kernel (o and lse feed the backward)
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros(B, d, Lp, device="cuda", dtype=torch.bfloat16)
    vt[:, :, :Lk] = v.view(B, Lk, d).transpose(1, 2)
    o = torch.empty(B * Lq, d, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, Lq, device="cuda", dtype=torch.float32)
    ops.flash_attn_raw(ops.ptr(q), ops.ptr(k), ops.ptr(vt), ops.ptr(o), ops.ptr(kl) if kl is not None else None, B, H,
                       Lq, Lk, Lq * d, d, Lk * d, d, d * Lp, Lq * d, d, Lp, scale, lse=ops.ptr(lse))
    dq, dk, dv = ops.flash_attn_bwd(q, k, v, o, do, lse, kl, B, H, Lq, Lk, scale)
    # reference
    qr, kr, vr = (t.float().view(B, -1, H, 128).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bhid,bhjd->bhij", qr, kr) * scale
    if kl is not None:
        mask = torch.arange(Lk, device="cuda")[None, :] >= kl[:, None].long()
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)                      # rows without any key
    out = torch.einsum("bhij,bhjd->bhid", p, vr)
    out.backward(do.float().view(B, Lq, H, 128).transpose(1, 2))
    for got, ref, L in ((dq, qr.grad, Lq), (dk, kr.grad, Lk), (dv, vr.grad, Lk)):
        ref = torch.nan_to_num(ref, nan=0.0).transpose(1, 2).reshape(B * L, d)
        assert torch.isfinite(got).all()
        if float(ref.abs().max()) == 0:
            assert float(got.abs().max()) == 0
        else:
            assert rel_rms(got, ref) < 1.2e-2
    if klens is not None and 0 in klens:
        b0 = klens.index(0)
        assert float(dq.view(B, Lq, d)[b0].abs().max()) == 0 and float(dk.view(B, Lk, d)[b0].abs().max()) == 0
    # repeatable bit for bit (no atomics)
    dq2, dk2, dv2 = ops.flash_attn_bwd(q, k, v, o, do, lse, kl, B, H, Lq, Lk, scale)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


@pytest.mark.parametrize("M,N,K,pad", [(128, 128, 64, 0), (256, 256, 200, 0), (1536, 1536, 1560, 0), (8960, 1536, 777, 0),
                                        (1536, 3072, 333, 64), (72, 200, 130, 8), (3072, 1536, 6240, 0)])
@pytest.mark.parametrize("tile", ["big", "small", "small-split3", "big-split8"])
def test_gemm_tn(ops, M, N, K, pad, tile, monkeypatch):
    """omh_gemm_bf16_tn — C = A^T B with both operands k-major (the weight gradient on dy and x as they are):
    against fp32 matmul on the same bf16 inputs; ragged M / N / K tails, strided rows, both tile configurations,
    accumulation, and an asymmetric pattern that a transposed or permuted fragment gather would scramble."""
    set_option("OMH_GEMM_TN_TILE", tile.split("-")[0])
    set_option("OMH_GEMM_TN_SPLIT", tile.split("split")[1] if "split" in tile else "1")    # split K: fp32 atomics
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a_full = torch.randn(K, M + pad, device="cuda", generator=g).bfloat16()
    b_full = torch.randn(K, N + pad, device="cuda", generator=g).bfloat16()
    a, b = a_full[:, :M], b_full[:, :N]
    ref = a.float().t() @ b.float()
    c = ops.gemm_tn(a, b)
    assert rel_rms(c, ref) < 2e-5
    c2 = ops.gemm_tn(a, b, out=c.clone(), accumulate=True)
    assert rel_rms(c2, 2 * ref) < 2e-5
    # selector test: A = one-hot rows picks single rows of B (exact), position dependent
    sel = torch.zeros(K, M, device="cuda", dtype=torch.bfloat16)
    idx = (torch.arange(M, device="cuda") * 7 + 3) % K
    sel[idx, torch.arange(M, device="cuda")] = 1.0
    assert torch.equal(ops.gemm_tn(sel, b), b.float()[idx])


@pytest.mark.parametrize("M,N,K,pad", [(256, 384, 192, 0), (256, 384, 200, 0), (512, 776, 1000, 0), (256, 8960, 333, 0),
                                        (1536, 1536, 6240, 0), (768, 392, 640, 40), (1536, 3072, 2048, 64),
                                        (4608, 1536, 1560, 0)])
def test_gemm_tn_w64_stream_equals_the_tiled_kernel(ops, M, N, K, pad, monkeypatch):
    """gemm_tn_w64.hip (the 256 x 384 generated stream, OMH_GEMM_TN_W64=1 forces it) == gemm_tn.hip's 128 x 128 kernel
    without split K BIT FOR BIT (same MFMA, same order over k) — store and accumulate epilogues, partial last k tile
    (K % 64 = 8, 40, 13 ...), ragged last column tile (N % 384 != 0), strided operands and a strided output — and nothing
    is written outside the output."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a_full = torch.randn(K, M + pad, device="cuda", generator=g).bfloat16()
    b_full = torch.randn(K, N + pad, device="cuda", generator=g).bfloat16()
    a, b = a_full[:, :M], b_full[:, :N]
    base = torch.randn(M, N, device="cuda", generator=g)
    set_option("OMH_GEMM_TN_W64", "0")
    set_option("OMH_GEMM_TN_TILE", "small")
    set_option("OMH_GEMM_TN_SPLIT", "1")
    want = ops.gemm_tn(a, b)
    want_acc = ops.gemm_tn(a, b, out=base.clone(), accumulate=True)
    set_option("OMH_GEMM_TN_W64", "1")
    got = ops.gemm_tn(a, b)
    assert rel_rms(got, a.float().t() @ b.float()) < 2e-5
    assert torch.equal(got, want)
    assert torch.equal(ops.gemm_tn(a, b, out=base.clone(), accumulate=True), want_acc)
    # a strided output inside a guard band: columns >= N and the rows around stay untouched
    canvas = torch.full((M + 2, N + 24), -3.0, device="cuda")
    view = canvas[1:M + 1, 8:N + 8]
    ops.gemm_tn(a, b, out=view)
    assert torch.equal(view, want)
    canvas[1:M + 1, 8:N + 8] = -3.0
    assert bool((canvas == -3.0).all())
    # selector: one-hot rows of A pick single rows of B exactly
    sel = torch.zeros(K, M, device="cuda", dtype=torch.bfloat16)
    idx = (torch.arange(M, device="cuda") * 7 + 3) % K
    sel[idx, torch.arange(M, device="cuda")] = 1.0
    assert torch.equal(ops.gemm_tn(sel, b), b.float()[idx])


def test_gemm_tn_w64_stream_grouped(ops, monkeypatch):
    """A block's weight-gradient group (q|k|v out of the fused gradient buffer, o, cross k|v over the context rows, one
    accumulating) on the stream kernel == each product on the tiled kernel, bit for bit; more tiles than workgroups
    (persistent loop) included."""
    g = torch.Generator(device="cuda").manual_seed(11)
    R, d = 1000, 512
    dqkv = (torch.randn(R, 3 * d, device="cuda", generator=g) * 0.3).bfloat16()
    h1 = (torch.randn(R, d, device="cuda", generator=g) * 0.3).bfloat16()
    dy1 = (torch.randn(R, d, device="cuda", generator=g) * 0.3).bfloat16()
    ctx = (torch.randn(333, d, device="cuda", generator=g) * 0.3).bfloat16()
    dkv = (torch.randn(333, 2 * d, device="cuda", generator=g) * 0.3).bfloat16()
    u = (torch.randn(R, 8960, device="cuda", generator=g) * 0.3).bfloat16()
    base = torch.randn(d, d, device="cuda", generator=g)
    probs = [(dqkv, h1, None), (dy1, h1, base), (dkv, ctx, None), (dy1, u, None), (u[:, :8960 - 8960 % 256], dy1, None),
             (dqkv[:, d:2 * d], h1[:, :392], None), (u, u[:, :2304], None)]        # 356 tiles of 256 x 384 in all
    set_option("OMH_GEMM_TN_W64", "0")
    set_option("OMH_GEMM_TN_TILE", "small")
    set_option("OMH_GEMM_TN_SPLIT", "1")
    want = [ops.gemm_tn(dy, x, out=None if b_ is None else b_.clone(), accumulate=b_ is not None) for dy, x, b_ in probs]
    set_option("OMH_GEMM_TN_W64", "1")
    items = [(dy, x, torch.full((dy.shape[1], x.shape[1]), 7.0, device="cuda") if b_ is None else b_.clone(), b_ is not None)
             for dy, x, b_ in probs]
    ops.gemm_tn_grouped(items)
    for (dy, x, out, acc), ref in zip(items, want):
        assert torch.equal(out, ref), (tuple(out.shape), rel_rms(out, ref))


@pytest.mark.parametrize("M,N,K,pad", [(128, 128, 64, 0), (300, 200, 72, 0), (1560, 1536, 1536, 0), (6240, 1536, 8960, 0),
                                        (1000, 8960, 1536, 0), (257, 72, 1000, 24), (512, 5120, 1536, 0)])
@pytest.mark.parametrize("tile", ["big", "small", "auto"])
def test_gemm_b_kmajor(ops, M, N, K, pad, tile, monkeypatch):
    """omh_gemm_bf16 with b_kmajor — C = A B on B = [K, N] row-major (dx = dy W on the weight as stored): against
    fp32 matmul on the same bf16 inputs; ragged M / N / K tails, a strided B, both tile configurations and the cost
    rule, the three epilogues the backward uses, a batched strided call, and an exact selector pattern."""
    if tile != "auto":
        set_option("OMH_GEMM_TILE", tile)
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    b_full = (torch.randn(K, N + pad, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    b = b_full[:, :N]
    bias = torch.randn(N, device="cuda", generator=g)
    ref = a.float() @ b.float()
    c = ops.gemm(a, b, epilogue=ops.EPI_F32, b_kmajor=True)
    assert rel_rms(c, ref) < 2e-5
    cb = ops.gemm(a, b, bias=bias, epilogue=ops.EPI_BF16, b_kmajor=True)
    assert rel_rms(cb.float(), ref + bias) < 4e-3
    c2 = ops.gemm(a, b, out=c.clone(), epilogue=ops.EPI_F32_ACCUM, b_kmajor=True)
    assert rel_rms(c2, 2 * ref) < 2e-5
    # equals the row-major-B kernel on the transposed copy up to summation order
    assert rel_rms(c, ops.gemm(a, b.t().contiguous(), epilogue=ops.EPI_F32)) < 2e-6
    # selector: B = one-hot columns picks single columns of A (exact), position dependent
    sel = torch.zeros(K, N, device="cuda", dtype=torch.bfloat16)
    idx = (torch.arange(N, device="cuda") * 5 + 1) % K
    sel[idx, torch.arange(N, device="cuda")] = 1.0
    assert torch.equal(ops.gemm(a, sel, epilogue=ops.EPI_F32, b_kmajor=True), a.float()[:, idx])
    with pytest.raises(ops.OmhError):
        ops.gemm(a, b, epilogue=ops.EPI_GELU_BF16, b_kmajor=True)


def test_gemm_b_kmajor_batched_strided(ops):
    """The context-gradient call of the training step: per sample b, d_ctx[b, first:first+L] += dy[b] @ W."""
    g = torch.Generator(device="cuda").manual_seed(9)
    B, L, Lc, first, d, N = 3, 257, 600, 64, 1536, 1536
    dy = torch.randn(B * L, N, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, d, device="cuda", generator=g) / math.sqrt(N)).bfloat16()
    d_ctx = torch.randn(B, Lc, d, device="cuda", generator=g)
    ref = d_ctx.clone()
    ref[:, first:first + L] += (dy.float() @ w.float()).view(B, L, d)
    ops.gemm_raw(ops.ptr(dy), ops.ptr(w), ops.ptr(d_ctx, first * d), L, d, N, N, d, d, ops.EPI_F32_ACCUM, batch=B,
                 strideA=L * N, strideB=0, strideC=Lc * d, b_kmajor=True)
    assert rel_rms(d_ctx, ref) < 2e-5
    assert torch.equal(d_ctx[:, :first], ref[:, :first]) and torch.equal(d_ctx[:, first + L:], ref[:, first + L:])
