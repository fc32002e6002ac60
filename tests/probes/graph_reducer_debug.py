"""Debug: GraphedTrainingStep + forced RCCL reducer vs eager, per-parameter stats (run on the GPU box)."""
import importlib, os, socket, sys
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
PKG = "omnihuman-1-hack_amd"
graphs = importlib.import_module(PKG + ".graphs"); trainer = importlib.import_module(PKG + ".trainer")
parallel = importlib.import_module(PKG + ".parallel"); wm = importlib.import_module(PKG + ".wan.modules.model")
from oracle import detgen, make_golden, wan_dit_oracle as O

def tiny(layers):
    cfg, tag, xs, ctx, tt, seq_len, _, _ = make_golden.tiny_case("t2v", layers)
    m = wm.WanModel(num_layers=layers, **make_golden.TINY); m.load_state_dict(O.synth_state_dict(cfg, tag))
    return m.cuda(), xs, ctx, tt, seq_len

def setup():
    m, xs, ctx, tt, seq_len = tiny(13)
    noise = torch.stack([xs[0], torch.from_numpy(detgen.normalish("gr/x0b", tuple(xs[0].shape)))]).cuda()
    vt = torch.from_numpy(detgen.normalish("gr/vt", tuple(noise.shape))).cuda()
    cc = torch.stack([ctx[0], torch.from_numpy(detgen.normalish("gr/c0b", tuple(ctx[0].shape)))]).cuda()
    return m.train(), (noise, cc, vt)

with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
mode = sys.argv[1] if len(sys.argv) > 1 else "reducer"
m_g, batch = setup(); m_e, _ = setup()
red = parallel.BucketedGradAllReduce(m_g.parameters(), bucket_mb=0.5, force=True) if mode == "reducer" else None
step = graphs.GraphedTrainingStep(m_g, batch, optimizer=None, reducer=red)
for it in range(4):
    b = tuple(u * (1.0 + 0.5 * it) for u in batch)
    float(step(b)); torch.cuda.synchronize()
    got = {n: p.grad.clone() for n, p in m_g.named_parameters() if p.grad is not None}
    for p in m_e.parameters(): p.grad = None
    trainer.training_step(b, m_e); torch.cuda.synchronize()
    bad = 0
    for n, p in m_e.named_parameters():
        if p.grad is None: continue
        a, r = got[n].double(), p.grad.double()
        e = float((a - r).norm() / r.norm().clamp_min(1e-30))
        if not (e < 1e-4):
            bad += 1
            if bad <= 6:
                print(f"it{it} {n}: rel {e:.3e} |a|max {float(a.abs().max()):.3e} |r|max {float(r.abs().max()):.3e} "
                      f"|a| {float(a.norm()):.3e} |r| {float(r.norm()):.3e} nonfinite a {int((~torch.isfinite(a)).sum())} r {int((~torch.isfinite(r)).sum())}")
    print(f"it{it}: {bad} parameters off", flush=True)
dist.destroy_process_group()
