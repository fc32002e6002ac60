"""Parity of the HIP 3D causal VAE (conv kernel + streaming executor) against
the CPU oracle (oracle/wan_vae_oracle.py, pinned to the reference by tests/golden)."""
import importlib

import pytest
import torch

from conftest import rel_rms, set_option

pytestmark = pytest.mark.gpu

# The reference runs the VAE in fp32; this build stores activations in bf16 and
# multiplies in bf16 MFMA with fp32 accumulation.  Through the ~35 conv layers of
# the decoder the relative RMS error of the output video stays below:
TOL_VAE = 2.0e-2          # measured with the fp32 residual trunk: 9.1e-3 decode / 5.4e-3 encode at 240x416 (x 2)


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("cfg", [
    dict(Cin=16, Cout=40, T=3, H=9, W=7, KT=3, KH=3, KW=3),
    dict(Cin=96, Cout=96, T=2, H=12, W=20, KT=3, KH=3, KW=3),
    dict(Cin=24, Cout=48, T=4, H=6, W=6, KT=3, KH=1, KW=1),
    dict(Cin=32, Cout=136, T=2, H=5, W=5, KT=1, KH=1, KW=1),
    dict(Cin=8, Cout=3, T=1, H=16, W=16, KT=3, KH=3, KW=3),
    dict(Cin=96, Cout=96, T=2, H=24, W=27, KT=3, KH=3, KW=3),     # 1296 voxels: 3 ragged 512-row tiles
    dict(Cin=192, Cout=192, T=1, H=20, W=31, KT=3, KH=3, KW=3),   # one full 192-column tile
    dict(Cin=64, Cout=384, T=2, H=11, W=13, KT=3, KH=3, KW=3),    # two 192-column tiles
    dict(Cin=48, Cout=200, T=1, H=17, W=17, KT=1, KH=3, KW=3),    # ragged second column tile
    dict(Cin=32, Cout=96, T=3, H=17, W=20, KT=1, KH=3, KW=3),     # 1020 voxels = exactly two 510-voxel kw-shared tiles
    dict(Cin=32, Cout=64, T=1, H=1, W=3, KT=3, KH=3, KW=3),       # one image row of three voxels: every voxel is an edge
    dict(Cin=384, Cout=384, T=1, H=9, W=29, KT=3, KH=3, KW=3),    # 12 channel blocks per tap pair, 254-voxel tiles
    dict(Cin=96, Cout=3, T=2, H=14, W=19, KT=3, KH=3, KW=3),      # the decoder head: kw-shared kernel with a 32-column tile
    dict(Cin=32, Cout=32, T=1, H=8, W=70, KT=1, KH=3, KW=3),
])
@pytest.mark.parametrize("tile", ["small", "wide", "wide-nokw3", "w64"])
def test_conv_cl_matches_torch(ops, cfg, tile, monkeypatch):
    # every tile configuration on every shape: 128x128; wide tiles with the kw-shared kernel where it applies
    # (3x3 taps, stride 1, Cin % 32 == 0); wide tiles without it; the one-wave-per-SIMD stream kernel where IT applies
    # (kw-shared shapes with Cout = 96 or a multiple of 192), the wide kernels elsewhere
    set_option("OMH_CONV_TILE", tile.split("-")[0])
    set_option("OMH_CONV_KW3", "0" if tile.endswith("nokw3") else "1")
    torch.manual_seed(cfg["Cin"] + cfg["Cout"])
    Cin, Cout, T, H, W, KT, KH, KW = (cfg[k] for k in ("Cin", "Cout", "T", "H", "W", "KT", "KH", "KW"))
    hist = KT - 1
    x = _bf(torch.randn(hist + T, H, W, Cin, device="cuda"))
    w = _bf(torch.randn(Cout, Cin, KT, KH, KW, device="cuda") / (Cin * KT * KH * KW) ** 0.5)
    bias = torch.randn(Cout, device="cuda")
    resid = _bf(torch.randn(T, H, W, Cout, device="cuda"))
    wp = w.permute(0, 2, 3, 4, 1).contiguous().view(Cout, -1)
    y = ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, KH, KW, pad_h=KH // 2, pad_w=KW // 2, resid=resid, out_f32=True)
    xin = x.float().permute(3, 0, 1, 2)[None]                              # [1, C, T, H, W]
    ref = torch.nn.functional.conv3d(torch.nn.functional.pad(xin, (KW // 2, KW // 2, KH // 2, KH // 2)), w.float(), bias)
    ref = ref[0].permute(1, 2, 3, 0) + resid.float()
    assert y.shape == ref.shape
    assert rel_rms(y, ref) < 2e-5
    # bf16 output (the layout every inner layer uses): same values, one bf16 rounding
    yb = ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, KH, KW, pad_h=KH // 2, pad_w=KW // 2, resid=resid)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb, y.to(torch.bfloat16))
    # fp32 residual -> fp32 output: the residual trunk of the VAE executor (omh_conv_args.resid_f32)
    rf = torch.randn(T, H, W, Cout, device="cuda")
    yf = ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, KH, KW, pad_h=KH // 2, pad_w=KW // 2, resid=rf, out_f32=True)
    assert rel_rms(yf, ref - resid.float() + rf) < 2e-5
    assert float((yf - rf - (y - resid.float())).abs().max()) < 1e-5        # same accumulators, only the residual differs


@pytest.mark.parametrize("cfg", [
    dict(Cin=96, Cout=96, T=2, H=12, W=20, KT=3),        # P (512 x 96): one ragged tile, 27 stages
    dict(Cin=96, Cout=96, T=3, H=24, W=27, KT=3),        # P: 1944 voxels = 3.8 tiles of 510
    dict(Cin=160, Cout=96, T=3, H=17, W=20, KT=1),       # P: exactly two tiles, the minimum of 15 stages (no rolled loop)
    dict(Cin=64, Cout=96, T=2, H=9, W=40, KT=3),         # P: 18 stages = the 13 peeled + 3 rolled + 2
    dict(Cin=192, Cout=192, T=1, H=20, W=31, KT=3),      # Q (256 x 192)
    dict(Cin=64, Cout=384, T=2, H=11, W=13, KT=3),       # Q: two cout tiles
    dict(Cin=384, Cout=384, T=1, H=9, W=29, KT=3),       # Q: 12 channel blocks per tap pair
    dict(Cin=96, Cout=192, T=4, H=30, W=52, KT=3),       # Q: 6240 voxels, 25 tiles (the decoder's first stage)
])
def test_conv_cl_w64_equals_the_kw_shared_kernel(ops, cfg, monkeypatch):
    """conv_w64.hip's stream kernel against conv_cl_kw3_kernel on the same inputs: same accumulation order, same
    epilogue arithmetic -> torch.equal, for bf16 / fp32 outputs with and without bias and residual."""
    torch.manual_seed(cfg["Cin"] + cfg["Cout"] + cfg["H"])
    Cin, Cout, T, H, W, KT = (cfg[k] for k in ("Cin", "Cout", "T", "H", "W", "KT"))
    x = _bf(torch.randn(KT - 1 + T, H, W, Cin, device="cuda"))
    wp = _bf(torch.randn(Cout, KT * 9 * Cin, device="cuda") / (Cin * KT * 9) ** 0.5)
    bias = torch.randn(Cout, device="cuda")
    rb, rf = _bf(torch.randn(T, H, W, Cout, device="cuda")), torch.randn(T, H, W, Cout, device="cuda")

    def run(tile):
        set_option("OMH_CONV_TILE", tile)
        kw = dict(pad_h=1, pad_w=1)
        return (ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, resid=rb, **kw),
                ops.conv_cl(x, wp, None, T, H, W, Cout, KT, 3, 3, **kw),
                ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, resid=rf, out_f32=True, **kw),
                ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, out_f32=True, **kw))

    got, ref = run("w64"), run("wide")
    for g, r in zip(got, ref):
        assert torch.isfinite(g.float()).all() and torch.equal(g, r)


@pytest.mark.parametrize("cfg", [
    dict(Cin=96, Cout=96, T=2, H=12, W=20, KT=3, up2=False),      # P: 54 stages, one ragged tile
    dict(Cin=96, Cout=96, T=3, H=24, W=27, KT=3, up2=False),      # P: several tiles per workgroup at grid < tiles? (3.8 tiles)
    dict(Cin=192, Cout=192, T=1, H=20, W=31, KT=3, up2=False),    # Q
    dict(Cin=384, Cout=384, T=1, H=9, W=29, KT=3, up2=False),     # Q: 216 stages, two cout tiles
    dict(Cin=384, Cout=192, T=2, H=9, W=14, KT=1, up2=True),      # Q through the folded 2x upsample (Conv2d): 72 stages
    dict(Cin=96, Cout=96, T=4, H=30, W=52, KT=3, up2=False),      # P: 6240 voxels
    dict(Cin=128, Cout=96, T=1, H=7, W=5, KT=1, up2=False),       # P: 24 stages: the shortest rolled loop (14 + 8 + 2), tiny volume
    dict(Cin=96, Cout=96, T=1, H=5, W=3, KT=1, up2=False),        # P: 18 stages (14 + 2 + 2), image rows of 3 voxels
])
def test_conv_pair_stream_matches_the_three_block_form(ops, cfg):
    """Round 5, omh_conv_args.pair: the fp32-faithful convolution on split-bf16 PAIRS interleaved per 16 channels — three
    MFMA products per channel block that share their LDS fragments (gen_conv_w64.py: main_loop_pair) — against the same
    product as a 3 C-channel convolution over [hi | lo | hi] x [hi | hi | lo] (rounds 3-4: other fp32 summation order,
    2e-6) and against fp64 convolution of the fp32 operands (fp32 class: 2e-5); with bias and fp32 residual, without;
    repeatable bit for bit; the query says no where the stream does not take the layer."""
    Cin, Cout, T, H, W, KT, up2 = (cfg[k] for k in ("Cin", "Cout", "T", "H", "W", "KT", "up2"))
    torch.manual_seed(Cin + Cout + H + W)
    Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
    x = torch.randn(KT - 1 + T, H, W, Cin, device="cuda")
    w = torch.randn(Cout, KT, 3, 3, Cin, device="cuda") / (Cin * KT * 9) ** 0.5
    bias = torch.randn(Cout, device="cuda")
    res = torch.randn(T, Ho, Wo, Cout, device="cuda")
    taps = KT * 9
    set_option("OMH_CONV_TILE", "w64")                       # (these volumes are below the stream's default threshold)
    assert ops.conv_pair_supported(KT - 1 + T, H, W, 2 * Cin, T, Ho, Wo, Cout, KT, 3, 3, pad_h=1, pad_w=1, up2=up2)
    x3, w3 = ops.split3(x, 0), ops.split3(w.view(Cout * taps, Cin), 1).view(Cout, -1)
    x2, w2 = ops.split3(x, 2, Cp=Cin), ops.split3(w.view(Cout * taps, Cin), 2, Cp=Cin).view(Cout, -1)
    assert x2.shape[-1] == 2 * Cin and w2.shape == (Cout, taps * 2 * Cin)
    kw = dict(pad_h=1, pad_w=1, up2=up2, out_f32=True)
    for b_, r_ in ((bias, res), (None, None)):
        want = ops.conv_cl(x3, w3, b_, T, Ho, Wo, Cout, KT, 3, 3, resid=r_, **kw)
        got = ops.conv_cl(x2, w2, b_, T, Ho, Wo, Cout, KT, 3, 3, resid=r_, pair=True, **kw)
        again = ops.conv_cl(x2, w2, b_, T, Ho, Wo, Cout, KT, 3, 3, resid=r_, pair=True, **kw)
        assert torch.isfinite(got).all() and torch.equal(got, again)
        assert rel_rms(got, want) < 2e-6, rel_rms(got, want)
        xin = x.permute(3, 0, 1, 2).unsqueeze(0).double()               # [1, C, T, H, W]
        if up2:
            xin = xin.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
        ref = torch.nn.functional.conv3d(xin, w.permute(0, 4, 1, 2, 3).double(), None if b_ is None else b_.double(),
                                         padding=(0, 1, 1))[0].permute(1, 2, 3, 0)
        if r_ is not None:
            ref = ref + r_.double()
        assert rel_rms(got, ref) < 2e-5, rel_rms(got, ref)
    # not a stream layer (stride 2 / 48 channels): the query says so and the call is refused, not silently mis-computed
    assert not ops.conv_pair_supported(3, H, W, 2 * 48, 1, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1)
    assert not ops.conv_pair_supported(3, H, W, 2 * Cin, 1, H // 2, W // 2, Cout, KT, 3, 3, stride_hw=2)
    set_option("OMH_CONV_TILE", "wide")
    with pytest.raises(ops.OmhError):
        ops.conv_cl(x2, w2, None, T, Ho, Wo, Cout, KT, 3, 3, pair=True, **kw)


def test_conv_cl_w64_random_shapes_equal_the_kw_shared_kernel(ops, monkeypatch):
    """Seeded sweep: 24 random (Cin, Cout, T, H, W, KT) within the stream kernel's domain — image rows as short as 3
    voxels, volumes smaller than one tile, ragged last tiles, 15 to 36 stages — bit for bit against the 8-wave
    kernel, bf16 and fp32-trunk outputs."""
    import random
    rng = random.Random(20260929)
    for case in range(24):
        KT, Cout = rng.choice((1, 3)), rng.choice((96, 192, 384))
        Cin = rng.choice((64, 96, 128)) if KT == 3 else rng.choice((160, 192, 256))       # >= 15 stages: the stream's domain
        T, H, W = rng.randint(1, 3), rng.randint(1, 40), rng.randint(3, 40)
        torch.manual_seed(case)
        x = _bf(torch.randn(KT - 1 + T, H, W, Cin, device="cuda"))
        wp = _bf(torch.randn(Cout, KT * 9 * Cin, device="cuda") / (Cin * KT * 9) ** 0.5)
        bias = torch.randn(Cout, device="cuda")
        rf = torch.randn(T, H, W, Cout, device="cuda")
        out = {}
        for tile in ("w64", "wide"):
            set_option("OMH_CONV_TILE", tile)
            out[tile] = (ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1, resid=_bf(rf)),
                         ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1, resid=rf, out_f32=True))
        for g, r in zip(out["w64"], out["wide"]):
            assert torch.equal(g, r), (case, Cin, Cout, T, H, W, KT)


@pytest.mark.parametrize("shape", [
    dict(Cin=96, T=2, H=12, W=20),         # one ragged tile
    dict(Cin=96, T=3, H=24, W=27),         # 3.8 tiles
    dict(Cin=192, T=2, H=40, W=52),        # the 192 -> 96 block after the last upsample, 8.2 tiles
])
def test_conv_cl_fused_norm_equals_conv_then_rms_silu(ops, shape, monkeypatch):
    """ABI v7: the next layer's RMS norm + SiLU in the stream convolution's epilogue (Cout = 96) against the same
    convolution followed by omh_rms_silu_cl(_f32in) (OMH_CONV_FUSE_NORM=0 routes through it): the same instructions on
    the same values in the same order, so y and the normalised output are equal bit for bit — bf16 kind (conv1 of a
    block; also with norm_only, which writes no y), fp32-trunk kind with a residual (conv2)."""
    Cin, T, H, W = (shape[k] for k in ("Cin", "T", "H", "W"))
    Cout, KT = 96, 3
    torch.manual_seed(Cin + H)
    x = _bf(torch.randn(KT - 1 + T, H, W, Cin, device="cuda"))
    wp = _bf(torch.randn(Cout, KT * 9 * Cin, device="cuda") / (Cin * KT * 9) ** 0.5)
    bias = torch.randn(Cout, device="cuda")
    gamma = torch.rand(Cout, device="cuda") + 0.5
    rf = torch.randn(T, H, W, Cout, device="cuda")

    set_option("OMH_CONV_TILE", "w64")               # these volumes are below the stream kernel's default threshold

    def run(fuse, **kw):
        set_option("OMH_CONV_FUSE_NORM", "1" if fuse else "0")
        n = torch.full((T, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
        y = ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1, norm_gamma=gamma, norm_out=n, **kw)
        return y, n

    for kw in (dict(), dict(resid=rf, out_f32=True), dict(resid=_bf(rf))):
        (y1, n1), (y0, n0) = run(True, **kw), run(False, **kw)
        assert torch.equal(y1, y0), kw.keys()
        assert bool(torch.isfinite(n0.float()).all()) and torch.equal(n1, n0), (kw.keys(), float((n1.float() - n0.float()).abs().max()))
        want = ops.rms_silu_cl(y0, gamma)
        assert torch.equal(n0, want)
    # norm_only: the normalised output alone (y's stores fall outside an empty descriptor)
    set_option("OMH_CONV_FUSE_NORM", "1")
    n = torch.empty(T, H, W, Cout, dtype=torch.bfloat16, device="cuda")
    sentinel = torch.full((T, H, W, Cout), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.conv_cl(x, wp, bias, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1, norm_gamma=gamma, norm_out=n, norm_only=True, out=sentinel)
    _, n0 = run(False)
    assert torch.equal(n, n0) and bool((sentinel == 7.0).all())


@pytest.mark.parametrize("shape", [dict(Cin=96, T=2, H=12, W=20), dict(Cin=96, T=3, H=24, W=27), dict(Cin=192, T=2, H=40, W=52)])
def test_conv_pair_fused_norm_equals_conv_then_rms_silu_pair(ops, shape):
    """Round 5: the next layer's RMS norm + SiLU in the PAIR stream's epilogue (fp32-faithful mode, Cout = 96), written as
    split-bf16 pairs [voxel][2 x 96], against the same convolution followed by omh_rms_silu_cl_pair (CONV_FUSE_NORM = 0
    routes through it): equal bit for bit — y (fp32, with the trunk's residual) and the normalised pairs; with norm_only
    no y is written; hi + lo of the normalised output reproduce the fp32 norm to 2^-16."""
    Cin, T, H, W = (shape[k] for k in ("Cin", "T", "H", "W"))
    Cout, KT = 96, 3
    torch.manual_seed(Cin + H + 1)
    x = torch.randn(KT - 1 + T, H, W, Cin, device="cuda")
    w = torch.randn(Cout, KT * 9, Cin, device="cuda") / (Cin * KT * 9) ** 0.5
    bias = torch.randn(Cout, device="cuda")
    gamma = torch.rand(Cout, device="cuda") + 0.5
    rf = torch.randn(T, H, W, Cout, device="cuda")
    x2, w2 = ops.split3(x, 2, Cp=Cin), ops.split3(w.view(Cout * KT * 9, Cin), 2, Cp=Cin).view(Cout, -1)
    set_option("OMH_CONV_TILE", "w64")

    def run(fuse, **kw):
        set_option("OMH_CONV_FUSE_NORM", "1" if fuse else "0")
        n = torch.full((T, H, W, 2 * Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
        y = ops.conv_cl(x2, w2, bias, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1, out_f32=True, pair=True, norm_gamma=gamma,
                        norm_out=n, **kw)
        return y, n
    for kw in (dict(resid=rf), dict()):
        (y1, n1), (y0, n0) = run(True, **kw), run(False, **kw)
        assert torch.equal(y1, y0)
        assert bool(torch.isfinite(n0.float()).all()) and torch.equal(n1, n0), float((n1.float() - n0.float()).abs().max())
        assert torch.equal(n0, ops.rms_silu_cl_split3(y0, gamma, pair=True))
        # the pairs: [hi(16) | lo(16)] per 16 channels; hi + lo = the fp32 norm to 2^-16 relative
        pr = n0.float().view(T, H, W, Cout // 16, 2, 16)
        yn = y0 * torch.rsqrt(y0.pow(2).sum(-1, keepdim=True).clamp_min(1e-24)) * (Cout ** 0.5) * gamma
        ref = torch.nn.functional.silu(yn).view(T, H, W, Cout // 16, 16)
        assert rel_rms(pr[..., 0, :] + pr[..., 1, :], ref) < 3e-5
    set_option("OMH_CONV_FUSE_NORM", "1")
    n = torch.empty(T, H, W, 2 * Cout, dtype=torch.bfloat16, device="cuda")
    sentinel = torch.full((T, H, W, Cout), 7.0, dtype=torch.float32, device="cuda")
    ops.conv_cl(x2, w2, bias, T, H, W, Cout, KT, 3, 3, pad_h=1, pad_w=1, out_f32=True, pair=True, norm_gamma=gamma, norm_out=n,
                norm_only=True, out=sentinel)
    _, n0 = run(False)
    assert torch.equal(n, n0) and bool((sentinel == 7.0).all())


@pytest.mark.parametrize("Cin,Cout,T,H,W", [(192, 96, 2, 13, 21), (384, 192, 1, 9, 17), (160, 96, 3, 6, 5)])
def test_conv_cl_w64_folded_upsample(ops, Cin, Cout, T, H, W, monkeypatch):
    """The decoder's upsample convolutions (nearest-2x + Conv2d 3x3, vae.py:76-79) on the stream kernel (round 3: the slab
    loader reads input row y >> 1, column x >> 1) against torch on the upsampled tensor, and against the 8-wave kernel."""
    torch.manual_seed(Cin + H)
    x = _bf(torch.randn(T, H, W, Cin, device="cuda"))
    w = _bf(torch.randn(Cout, Cin, 3, 3, device="cuda") / (9 * Cin) ** 0.5)
    wp = w.permute(0, 2, 3, 1).contiguous().view(Cout, -1)
    bias = torch.randn(Cout, device="cuda")
    out = {}
    for tile in ("w64", "wide"):
        set_option("OMH_CONV_TILE", tile)
        out[tile] = ops.conv_cl(x, wp, bias, T, 2 * H, 2 * W, Cout, 1, 3, 3, pad_h=1, pad_w=1, up2=True, out_f32=True)
    xi = x.float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(xi, scale_factor=2.0, mode="nearest-exact"), w.float(),
                                     bias, padding=1).permute(0, 2, 3, 1)
    assert out["w64"].shape == ref.shape
    assert rel_rms(out["w64"], ref) < 2e-3 and rel_rms(out["wide"], ref) < 2e-3
    assert rel_rms(out["w64"], out["wide"]) < 1e-5


@pytest.mark.parametrize("tile", ["small", "wide"])
def test_conv_cl_upsample_downsample_stride_split(ops, tile, monkeypatch):
    set_option("OMH_CONV_TILE", tile)
    torch.manual_seed(9)
    C, H, W = 32, 6, 10
    x = _bf(torch.randn(2, H, W, C, device="cuda"))
    # nearest-2x upsample folded into a 3x3 conv (vae.py:76-79)
    w = _bf(torch.randn(16, C, 3, 3, device="cuda") / (9 * C) ** 0.5)
    wp = w.permute(0, 2, 3, 1).contiguous().view(16, -1)
    y = ops.conv_cl(x, wp, None, 2, 2 * H, 2 * W, 16, 1, 3, 3, pad_h=1, pad_w=1, up2=True, out_f32=True)
    xi = x.float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(xi, scale_factor=2.0, mode="nearest-exact"), w.float(),
                                     padding=1).permute(0, 2, 3, 1)
    assert rel_rms(y, ref) < 2e-5
    # ZeroPad2d((0,1,0,1)) + stride-2 conv (vae.py:88-90)
    w2 = _bf(torch.randn(C, C, 3, 3, device="cuda") / (9 * C) ** 0.5)
    y2 = ops.conv_cl(x, w2.permute(0, 2, 3, 1).contiguous().view(C, -1), None, 2, H // 2, W // 2, C, 1, 3, 3,
                     stride_hw=2, out_f32=True)
    ref2 = torch.nn.functional.conv2d(torch.nn.functional.pad(xi, (0, 1, 0, 1)), w2.float(), stride=2).permute(0, 2, 3, 1)
    assert rel_rms(y2, ref2) < 2e-5
    # temporal stride-2 conv over [history(1) | 4 frames] (vae.py:95-96,156-157)
    x5 = _bf(torch.randn(5, H, W, C, device="cuda"))
    w3 = _bf(torch.randn(C, C, 3, 1, 1, device="cuda") / (3 * C) ** 0.5)
    y3 = ops.conv_cl(x5, w3.permute(0, 2, 3, 4, 1).contiguous().view(C, -1), None, 2, H, W, C, 3, 1, 1, stride_t=2,
                     out_f32=True)
    ref3 = torch.nn.functional.conv3d(x5.float().permute(3, 0, 1, 2)[None], w3.float(), stride=(2, 1, 1))[0].permute(1, 2, 3, 0)
    assert rel_rms(y3, ref3) < 2e-5
    # channel -> frame interleave of the temporal upsample (vae.py:134-137)
    x4 = _bf(torch.randn(2 + 2, H, W, C, device="cuda"))
    w4 = _bf(torch.randn(2 * C, C, 3, 1, 1, device="cuda") / (3 * C) ** 0.5)
    y4 = ops.conv_cl(x4, w4.permute(0, 2, 3, 4, 1).contiguous().view(2 * C, -1), None, 2, H, W, 2 * C, 3, 1, 1,
                     split_n=C, out_f32=True)
    r = torch.nn.functional.conv3d(x4.float().permute(3, 0, 1, 2)[None], w4.float())        # [1, 2C, 2, H, W]
    r = r.reshape(1, 2, C, 2, H, W)
    r = torch.stack((r[:, 0], r[:, 1]), 3).reshape(1, C, 4, H, W)[0].permute(1, 2, 3, 0)
    assert y4.shape == (4, H, W, C)
    assert rel_rms(y4, r) < 2e-5


def test_rms_silu_softmax_layout(ops):
    torch.manual_seed(11)
    for C in (96, 192, 384, 16):
        x = _bf(torch.randn(77, C, device="cuda") * 2)
        g = torch.rand(C, device="cuda") + 0.5
        ref = torch.nn.functional.silu(torch.nn.functional.normalize(x.float(), dim=1) * C ** 0.5 * g)
        assert rel_rms(ops.rms_silu_cl(x, g).float(), ref) < 4e-3
        xf = torch.randn(77, C, device="cuda") * 2                # fp32 input (the residual trunk): omh_rms_silu_cl_f32in
        reff = torch.nn.functional.silu(torch.nn.functional.normalize(xf, dim=1) * C ** 0.5 * g)
        assert rel_rms(ops.rms_silu_cl(xf, g).float(), reff) < 3e-3
        assert torch.equal(ops.rms_silu_cl(x.float(), g), ops.rms_silu_cl(x, g))      # same values in, same bits out
    s = torch.randn(50, 333, device="cuda") * 3
    p = torch.zeros(50, 336, dtype=torch.bfloat16, device="cuda")
    ops.softmax_rows(s, p, 333, 0.7)
    assert rel_rms(p[:, :333].float(), torch.softmax(s * 0.7, -1)) < 4e-3
    assert float(p[:, 333:].abs().max()) == 0
    v = torch.randn(5, 7, 4, 6, device="cuda")
    mul, add = torch.rand(5, device="cuda") + 0.5, torch.randn(5, device="cuda")
    cl = ops.nchw_to_cl(v, 3, 2, 8, mul=mul, add=add)
    ref = (v[:, 2:5] * mul[:, None, None, None] + add[:, None, None, None]).permute(1, 2, 3, 0)
    assert rel_rms(cl[..., :5].float(), ref) < 4e-3 and float(cl[..., 5:].abs().max()) == 0
    out = torch.zeros(5, 7, 4, 6, device="cuda")
    src = torch.randn(3, 4, 6, 8, device="cuda")
    ops.cl_to_nchw(src, out, 2, 5, mul=mul, add=add, lo=-1.0, hi=1.0)
    ref2 = ((src[..., :5] + add) * mul).clamp(-1, 1).permute(3, 0, 1, 2)
    assert torch.allclose(out[:, 2:5], ref2, atol=1e-6) and float(out[:, :2].abs().max()) == 0


@pytest.mark.parametrize("window", ["default", "one-chunk", "two-chunks"])
@pytest.mark.parametrize("dim", [32, 96])
def test_vae_decode_encode_match_oracle(dim, window, monkeypatch):
    """``window``: size of the sliding history window of every conv input buffer.  The default (8 chunks) never
    reaches its end on a 9-frame clip; "one-chunk" makes the buffer exactly [history | chunk], so the history is
    copied back before every chunk (incl. the overlapping T = 1, hist = 2 case), "two-chunks" wraps every other one."""
    from oracle import wan_vae_oracle as V, detgen
    vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
    if window != "default":
        orig_slot = vae_mod._ConvState.slot
        chunks = 1 if window == "one-chunk" else 2

        def small_slot(self, T, H, W, device):
            # budget of exactly hist + chunks * T frames for this layer
            monkeypatch.setattr(vae_mod._ConvState, "_WINDOW_BYTES", (self.hist + chunks * max(T, 1)) * H * W * self.Cin * 2)
            return orig_slot(self, T, H, W, device)
        monkeypatch.setattr(vae_mod._ConvState, "slot", small_slot)
    cfg = V.VAEConfig(dim=dim)
    sd = V.synth_state_dict(cfg, f"vae{dim}")
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda", dim=dim)
    vae.model.load_state_dict(sd)
    z = torch.from_numpy(detgen.normalish("vae/z", (16, 3, 8, 8)))
    ref = V.vae_decode(sd, cfg, z)
    out = vae.decode([z.cuda()])[0]
    assert out.shape == ref.shape == (3, 9, 64, 64) and out.dtype == torch.float32
    assert float(out.abs().max()) <= 1.0
    assert rel_rms(out, ref) < TOL_VAE
    vid = torch.from_numpy(detgen.uniform("vae/vid", (3, 9, 32, 32)))
    refe = V.vae_encode(sd, cfg, vid)
    oute = vae.encode([vid.cuda()])[0]
    assert oute.shape == refe.shape == (16, 3, 4, 4)
    assert rel_rms(oute, refe) < TOL_VAE


def test_vae_decode_at_wide_tile_sizes_matches_oracle():
    """A decode large enough ([16,2,30,52] -> [3,5,240,416], dim 96) that the C=96 and C=192 stages run the
    512x96 / 256x192 convolution tiles, ragged last tiles and the frame-interleaving time conv included."""
    from oracle import wan_vae_oracle as V, detgen
    vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
    cfg = V.VAEConfig(dim=96)
    sd = V.synth_state_dict(cfg, "vae96wide")
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.bfloat16, device="cuda", dim=96)
    vae.model.load_state_dict(sd)
    z = torch.from_numpy(detgen.normalish("vae/zwide", (16, 2, 30, 52)))
    ref = V.vae_decode(sd, cfg, z)
    out = vae.decode([z.cuda()])[0]
    assert out.shape == ref.shape == (3, 5, 240, 416)
    assert rel_rms(out, ref) < TOL_VAE


TOL_VAE_F32 = 2.0e-4        # WanVAE(dtype=torch.float): split-bf16 operand pairs, fp32 accumulate


def test_split3_operands(ops):
    """omh_split3_f32: x = hi + lo to 2^-16 relative, the two block patterns, zero pad channels, strided rows; a product
    of a pattern-0 row with a pattern-1 row over the 3 Cp channels = x . w to fp32 class."""
    torch.manual_seed(5)
    x = torch.randn(37, 20, device="cuda") * 3
    a = ops.split3(x, 0)                                                   # Cp = 24
    assert a.shape == (37, 72) and a.dtype == torch.bfloat16
    hi, lo, hi2 = a[:, :24].float(), a[:, 24:48].float(), a[:, 48:].float()
    assert torch.equal(hi, hi2) and torch.equal(hi[:, :20], x.bfloat16().float())
    assert bool((a[:, 20:24] == 0).all()) and bool((a[:, 44:48] == 0).all())
    assert float(((hi + lo)[:, :20] - x).abs().max()) <= 2.0 ** -16 * float(x.abs().max())
    b = ops.split3(x, 1)
    assert torch.equal(b[:, :24], a[:, :24]) and torch.equal(b[:, 24:48], a[:, :24]) and torch.equal(b[:, 48:], a[:, 24:48])
    big = torch.randn(64, 96, device="cuda")
    v = ops.split3(big[:, 32:64], 0)                                       # a column block of a wider matrix
    assert torch.equal(v[:, :32], big[:, 32:64].bfloat16())
    # the GEMM on split operands against fp64
    A, W = torch.randn(200, 96, device="cuda"), torch.randn(64, 96, device="cuda")
    got = ops.gemm(ops.split3(A, 0), ops.split3(W, 1), epilogue=ops.EPI_F32)
    want = A.double() @ W.double().t()
    assert rel_rms(got, want) < 5e-6                                       # (one bf16 rounding of the operands: 3e-3)
    assert rel_rms(ops.gemm(A.bfloat16(), W.bfloat16(), epilogue=ops.EPI_F32), want) > 1e-3


@pytest.mark.parametrize("dim", [16, 96])
def test_vae_fp32_mode_matches_oracle(dim):
    """WanVAE(dtype=torch.float) — the reference's default arithmetic (vae.py:619-624: autocast to fp32) — against the
    fp32 oracle on the tiny clip of test_vae_decode_encode_matches_oracle: every chunk kind, both resamplers' first-chunk
    bypasses, the mid-block attention with split q / k / P / V operands."""
    from oracle import wan_vae_oracle as V, detgen
    vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
    cfg = V.VAEConfig(dim=dim)
    sd = V.synth_state_dict(cfg, f"vae{dim}")
    vae = vae_mod.WanVAE(vae_pth=None, dtype=torch.float, device="cuda", dim=dim)
    vae.model.load_state_dict(sd)
    z = torch.from_numpy(detgen.normalish("vae/z", (16, 3, 8, 8)))
    ref = V.vae_decode(sd, cfg, z)
    out = vae.decode([z.cuda()])[0]
    assert out.shape == ref.shape == (3, 9, 64, 64) and out.dtype == torch.float32
    e = rel_rms(out, ref)
    vid = torch.from_numpy(detgen.uniform("vae/vid", (3, 9, 32, 32)))
    refe = V.vae_encode(sd, cfg, vid)
    oute = vae.encode([vid.cuda()])[0]
    ee = rel_rms(oute, refe)
    print(f"[measured] fp32-mode VAE (dim {dim}) vs oracle: decode {e:.3e}, encode {ee:.3e}")
    assert e < TOL_VAE_F32 and ee < TOL_VAE_F32, (e, ee)
    with pytest.raises(ValueError):
        vae_mod.WanVAE(vae_pth=None, dtype=torch.int8, device="cuda", dim=dim)


def test_vae_fp32_mode_at_wide_tile_sizes():
    """The fp32 mode through the 512x96 / 256x192 stream tiles (3 x 96 = 288, 3 x 192 = 576 input channels per tap),
    the folded upsamples and the frame-interleaving time convolution with an fp32 result."""
    from oracle import wan_vae_oracle as V, detgen
    vae_mod = importlib.import_module("omnihuman-1-hack_amd.wan.modules.vae")
    cfg = V.VAEConfig(dim=96)
    sd = V.synth_state_dict(cfg, "vae96wide")
    vae = vae_mod.WanVAE(vae_pth=None, device="cuda", dim=96)               # default dtype: torch.float
    vae.model.load_state_dict(sd)
    z = torch.from_numpy(detgen.normalish("vae/zwide", (16, 2, 30, 52)))
    ref = V.vae_decode(sd, cfg, z)
    out = vae.decode([z.cuda()])[0]
    e = rel_rms(out, ref)
    print(f"[measured] fp32-mode VAE decode 240x416 vs oracle: {e:.3e}")
    assert out.shape == ref.shape == (3, 5, 240, 416) and e < TOL_VAE_F32
