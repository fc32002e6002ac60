"""The asm-owned 4x64 attention kernel (csrc/attention_w64.hip) against fp32 torch arithmetic on several shapes, then
interleaved timing against the 8-wave kernel at the benchmark shape.  OMH_ATTN_KERNEL is read per launch.
    python tools/attn_w64_probe.py [check|time|all]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
D = 128


def vt_of(v, Lk):
    B, _, H, _ = v.shape
    Lp = (Lk + 63) // 64 * 64
    vt = torch.zeros(B, H * D, Lp, dtype=torch.bfloat16, device="cuda")
    vt[:, :, :Lk] = v.reshape(B, Lk, H * D).transpose(1, 2)
    return vt


LOG2E = 1.4426950408889634


def run(kind, q, k, vt, k_lens=None, lse=False, pre=1):
    ops.set_option("OMH_ATTN_KERNEL", kind)
    B, Lq, H, _ = q.shape
    Lk = k.shape[1]
    out = torch.full((B, Lq, H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    l = torch.full((B, H, Lq), float("nan"), dtype=torch.float32, device="cuda") if lse else None
    ops.flash_attn_raw(ops._p(q), ops._p(k), ops._p(vt), ops._p(out), ops._p(k_lens), B, H, Lq, Lk, q.stride(0),
                       q.stride(1), k.stride(0), k.stride(1), vt.stride(0), out.stride(0), out.stride(1), vt.stride(1),
                       D ** -0.5, lse=ops._p(l) if lse else None, q_prescaled=pre)
    torch.cuda.synchronize()
    return out, l


def check():
    torch.manual_seed(0)
    ok = True
    cases = [(1, 2, 256, 64, None, 1.0), (1, 2, 256, 128, None, 1.0), (1, 1, 256, 512, None, 1.0),
             (1, 3, 300, 200, None, 1.0), (2, 2, 777, 1000, [1000, 333], 1.0), (1, 2, 512, 4096, None, 1.0),
             (1, 2, 512, 2048, None, 4.0), (1, 1, 64, 100, [37], 1.0), (1, 12, 1560, 1560, None, 1.0)]
    for (B, H, Lq, Lk, kl, amp) in cases:
        # q as the norm kernel hands it over: already multiplied by scale * log2(e), rounded once
        q = (torch.randn(B, Lq, H, D, device="cuda") * amp * (D ** -0.5 * LOG2E)).to(torch.bfloat16)
        k = (torch.randn(B, Lk, H, D, device="cuda") * amp).to(torch.bfloat16)
        v = torch.randn(B, Lk, H, D, device="cuda").to(torch.bfloat16)
        if amp > 1:                      # spike: one key row aligned with one query row far down the sequence (forces a late rescale)
            k[:, Lk - 70] = (q[:, 5].float() / (D ** -0.5 * LOG2E)).to(torch.bfloat16)
        vt = vt_of(v, Lk)
        kls = None if kl is None else torch.tensor(kl, dtype=torch.int32, device="cuda")
        got, lse = run("w64", q, k, vt, kls, lse=True)
        base, lse_b = run("base", q, k, vt, kls, lse=True)
        s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) / LOG2E        # natural-log domain
        if kl is not None:
            for b_, n in enumerate(kl):
                s[b_, :, :, n:] = float("-inf")
        ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float())
        ref_lse = torch.logsumexp(s, -1)
        def rr(a, b): return float((a.float() - b).norm() / b.norm())
        e_w, e_b = rr(got, ref), rr(base, ref)
        mx = float((got.float() - ref).abs().max())
        e_l = float((lse - ref_lse).abs().max())
        fin = bool(torch.isfinite(got.float()).all())
        good = fin and e_w < 8e-3 and mx < 3e-2 and e_l < 2e-2
        ok &= good
        print(f"B{B} H{H} Lq{Lq} Lk{Lk} kl{kl} amp{amp}: w64 rel {e_w:.3e} max {mx:.3e} lse {e_l:.2e} | base rel {e_b:.3e} "
              f"finite {fin} {'OK' if good else 'FAIL'}", flush=True)
        got2, _ = run("w64", q, k, vt, kls)
        if not torch.equal(got2, got):
            print("   NOT bit-repeatable"); ok = False
        # the un-prescaled entry (the kernel multiplies q by scale*log2e and re-rounds it): looser, error ~ |scores| 2^-9
        qu = (q.float() / (D ** -0.5 * LOG2E)).to(torch.bfloat16)
        got3, _ = run("w64", qu, k, vt, kls, pre=0)
        s3 = torch.einsum("bqhd,bkhd->bhqk", qu.float(), k.float()) * D ** -0.5
        if kl is not None:
            for b_, n in enumerate(kl):
                s3[b_, :, :, n:] = float("-inf")
        ref3 = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s3, -1), v.float())
        e3 = rr(got3, ref3)
        print(f"   un-prescaled q: rel {e3:.3e}")
        ok &= e3 < 8e-3 * amp
    print("CHECK", "PASS" if ok else "FAIL", flush=True)
    return ok


def timeit():
    S, H = int(os.environ.get("S", 32760)), 12
    q = torch.randn(1, S, H, D, device="cuda").to(torch.bfloat16)
    k = torch.randn(1, S, H, D, device="cuda").to(torch.bfloat16)
    v = torch.randn(1, S, H, D, device="cuda").to(torch.bfloat16)
    vt = vt_of(v, S)
    o = torch.empty_like(q)
    res = {}
    for rnd in range(3):
        for kind in ("w64",):
            ops.set_option("OMH_ATTN_KERNEL", kind)
            for _ in range(3):
                ops.flash_attn(q, k, vt, None, out=o)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.flash_attn(q, k, vt, None, out=o)
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 10
            res.setdefault(kind, []).append(ms)
            print(f"round {rnd} {kind}: {ms:.4f} ms {4.0 * S * S * H * D / ms / 1e9:.1f} TF", flush=True)
    # sampled-row parity at full size
    ops.set_option("OMH_ATTN_KERNEL", "w64")
    ops.flash_attn(q, k, vt, None, out=o)
    rows = torch.tensor([0, 1, 255, 256, 16383, S - 257, S - 1, 12345], device="cuda")
    for h in (0, 7):
        s_ = (q[0, rows, h].float() @ k[0, :, h].float().t()) * D ** -0.5
        ref = torch.softmax(s_, -1) @ v[0, :, h].float()
        print(f"full-size rows head {h}: rel {float((o[0, rows, h].float() - ref).norm() / ref.norm()):.3e}")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    good = True
    if mode in ("check", "all"):
        good = check()
    if mode in ("time", "all") and good:
        timeit()
