import importlib, sys, torch
sys.path.insert(0, "/root/repo")
ops = importlib.import_module("omnihuman-1-hack_amd.ops")
torch.manual_seed(0)
D = 128
for B, H, Lq, Lk, kl in ((2, 2, 200, 200, None), (2, 2, 256, 256, None), (2, 2, 200, 300, [300, 170]), (1, 2, 64, 40, None), (2, 3, 1560, 1560, None), (1, 2, 700, 512, [0])):
    d = H * D
    q = torch.randn(B * Lq, d, device="cuda").bfloat16(); k = torch.randn(B * Lk, d, device="cuda").bfloat16()
    v = torch.randn(B * Lk, d, device="cuda").bfloat16(); do = torch.randn(B * Lq, d, device="cuda").bfloat16()
    klens = None if kl is None else torch.tensor(kl, device="cuda", dtype=torch.int32)
    scale = D ** -0.5
    for pres in (False, True):
        qq = (q.float() * scale * 1.4426950408889634).bfloat16() if pres else q
        qf = qq.float().view(B, Lq, H, D).transpose(1, 2); kf = k.float().view(B, Lk, H, D).transpose(1, 2); vf = v.float().view(B, Lk, H, D).transpose(1, 2)
        s2 = (qf @ kf.transpose(-1, -2)) * (1 / 1.4426950408889634 if pres else scale)
        if kl is not None:
            m = torch.arange(Lk, device="cuda")[None, :] >= klens[:, None]
            s2 = s2.masked_fill(m[:, None, None, :], float("-inf"))
        lse2 = torch.logsumexp(s2, -1).contiguous()
        pp = torch.nan_to_num(s2.softmax(-1))
        o322 = (pp @ vf).transpose(1, 2).reshape(B * Lq, d).contiguous()
        res = {}
        for opt in ("0", "1"):
            ops.set_option("OMH_ATTN_BWD_W64", opt)
            res[opt] = ops.flash_attn_bwd(qq, k, v, None, do, lse2, klens, B, H, Lq, Lk, scale, q_prescaled=pres, o32=o322)
            res[opt + "r"] = ops.flash_attn_bwd(qq, k, v, None, do, lse2, klens, B, H, Lq, Lk, scale, q_prescaled=pres, o32=o322)
        for i, nm in ((1, "dk"), (2, "dv")):
            a, b_ = res["0"][i].float(), res["1"][i].float()
            err = float((a - b_).norm() / (a.norm() + 1e-30))
            print(B, H, Lq, Lk, kl, pres, nm, "finite", bool(torch.isfinite(b_).all()), "rel vs hip", round(err, 6), "maxabs", float((a - b_).abs().max()),
                  "repeat", bool(torch.equal(res["1"][i], res["1r"][i])), flush=True)
