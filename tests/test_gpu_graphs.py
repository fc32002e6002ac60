"""hipGraph capture of the launch-bound regimes (graphs.py): a replay must compute exactly what the eager
launch sequence computes — same kernels, same order — for the inference forward and for the training step."""
import importlib

import pytest
import torch

from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _tiny(wan_model_mod, layers=2):
    from oracle import make_golden, wan_dit_oracle as O
    cfg, tag, xs, ctx, tt, seq_len, _, _ = make_golden.tiny_case("t2v", layers)
    m = wan_model_mod.WanModel(num_layers=layers, **make_golden.TINY)
    m.load_state_dict(O.synth_state_dict(cfg, tag))
    return m.cuda(), xs, ctx, tt, seq_len


def test_graphed_forward_replays_the_eager_forward(wan_model_mod):
    graphs = importlib.import_module("omnihuman-1-hack_amd.graphs")
    m, xs, ctx, tt, seq_len = _tiny(wan_model_mod)
    m = m.eval().requires_grad_(False)
    x = [u.cuda() for u in xs]
    c = [u.cuda() for u in ctx]
    t = tt.cuda()
    eager = [o.clone() for o in m(x, t, c, seq_len)]
    g = graphs.GraphedForward(m, x, t, c, seq_len)
    out = g(x, t)
    for a, b in zip(out, eager):
        assert torch.equal(a, b)
    # new latents and timestep through the same graph; cached context state as in WanT2V.generate
    st = m.encode_context(c)
    g2 = graphs.GraphedForward(m, x, t, st, seq_len)
    x2 = [torch.randn_like(u) for u in x]
    t2 = torch.full_like(t, 417.0)
    want = [o.clone() for o in m(x2, t2, st, seq_len)]
    for gg in (g, g2):
        got = gg(x2, t2)
        for a, b in zip(got, want):
            assert torch.equal(a, b)


def test_graphed_training_step_matches_eager(wan_model_mod):
    """Gradients of a replay against the eager step (the backward's column sums use atomics: compare to 1e-5),
    then two optimizer steps through the graph against two eager steps (exercises the in-graph re-packing of
    the updated weights)."""
    graphs = importlib.import_module("omnihuman-1-hack_amd.graphs")
    trainer = importlib.import_module("omnihuman-1-hack_amd.trainer")
    optim = importlib.import_module("omnihuman-1-hack_amd.optim")
    from oracle import detgen

    def setup():
        m, xs, ctx, tt, seq_len = _tiny(wan_model_mod, 13)
        m = m.train()
        tag = "graph"
        noise = torch.stack([xs[0], torch.from_numpy(detgen.normalish(f"{tag}/x0b", tuple(xs[0].shape)))]).cuda()
        vt = torch.from_numpy(detgen.normalish(f"{tag}/vt", tuple(noise.shape))).cuda()
        L = ctx[0].shape[0]
        cc = torch.stack([ctx[0], torch.from_numpy(detgen.normalish(f"{tag}/c0b", tuple(ctx[0].shape)))]).cuda()
        return m, (noise, cc, vt)

    m_e, batch = setup()
    m_g, _ = setup()
    opt_e = optim.AdamW(m_e.parameters(), lr=1e-3)
    opt_g = optim.AdamW(m_g.parameters(), lr=1e-3)
    step = graphs.GraphedTrainingStep(m_g, batch, optimizer=opt_g)
    losses_e, losses_g = [], []
    for it in range(2):
        b = tuple(u + 0.01 * it for u in batch)
        losses_e.append(trainer.training_step(b, m_e))
        if it == 0:
            ge = {n: p.grad.clone() for n, p in m_e.named_parameters() if p.grad is not None}
        opt_e.step()
        opt_e.zero_grad(set_to_none=True)
        losses_g.append(float(step(b)))
        if it == 0:
            for n, p in m_g.named_parameters():
                assert (p.grad is None) == (n not in ge), n
                if p.grad is not None:
                    assert rel_rms(p.grad, ge[n]) < 1e-4, n
    assert losses_g[0] == pytest.approx(losses_e[0], rel=1e-6)
    assert losses_g[1] == pytest.approx(losses_e[1], rel=1e-4)
    assert losses_g[1] != losses_g[0]
    # AdamW normalises the update (|step| <= lr whatever the gradient's size): where a gradient is ~0 the atomics'
    # summation order decides its sign, so weights may differ by up to 2 lr per step there and nowhere by more
    for (n, a), (_, b) in zip(m_g.named_parameters(), m_e.named_parameters()):
        assert float((a.detach() - b.detach()).abs().max()) <= 2 * 1e-3 * 2 * 1.05, n
        assert rel_rms(a.detach(), b.detach()) < 5e-2, n


def test_graphed_training_step_rejects_accumulation(wan_model_mod):
    """A replay overwrites .grad: micro-batch accumulation cannot be expressed with this graph."""
    graphs = importlib.import_module("omnihuman-1-hack_amd.graphs")
    m, xs, ctx, tt, seq_len = _tiny(wan_model_mod, 2)
    with pytest.raises(ValueError):
        graphs.GraphedTrainingStep(m.train(), (torch.stack([xs[0]]).cuda(), torch.stack([ctx[0]]).cuda(),
                                               torch.stack([xs[0]]).cuda()), gradient_accumulation_steps=4)


def test_graphed_training_step_with_rccl_reducer_sees_fresh_gradients(wan_model_mod):
    """GraphedTrainingStep + BucketedGradAllReduce over RCCL (a one-rank "nccl" group, collectives forced):
    finish() re-points p.grad at the reduced flat bucket while replays keep writing to the captured tensors —
    step 2 must reduce and hand the optimizer step 2's gradients, not step 1's (graphs.py:131).
    Also the only place ReduceOp.AVG runs through RCCL in the GPU tests (the CPU tests use gloo + SUM)."""
    import socket
    import torch.distributed as dist
    graphs = importlib.import_module("omnihuman-1-hack_amd.graphs")
    trainer = importlib.import_module("omnihuman-1-hack_amd.trainer")
    parallel = importlib.import_module("omnihuman-1-hack_amd.parallel")
    from oracle import detgen
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        def setup():
            m, xs, ctx, tt, seq_len = _tiny(wan_model_mod, 13)
            noise = torch.stack([xs[0], torch.from_numpy(detgen.normalish("gr/x0b", tuple(xs[0].shape)))]).cuda()
            vt = torch.from_numpy(detgen.normalish("gr/vt", tuple(noise.shape))).cuda()
            cc = torch.stack([ctx[0], torch.from_numpy(detgen.normalish("gr/c0b", tuple(ctx[0].shape)))]).cuda()
            return m.train(), (noise, cc, vt)

        m_g, batch = setup()
        m_e, _ = setup()
        red = parallel.BucketedGradAllReduce(m_g.parameters(), bucket_mb=0.5, force=True)
        assert len(red.buckets) > 2
        step = graphs.GraphedTrainingStep(m_g, batch, optimizer=None, reducer=red)
        grads = []
        for it in range(3):
            b = tuple(u * (1.0 + 0.5 * it) for u in batch)
            float(step(b))
            assert red._avg_in_coll                          # RCCL: the mean is taken inside the collective
            got = {n: p.grad.clone() for n, p in m_g.named_parameters() if p.grad is not None}
            for p in m_e.parameters():
                p.grad = None
            trainer.training_step(b, m_e)
            for n, p in m_e.named_parameters():
                assert (p.grad is None) == (n not in got), n
                if p.grad is not None:
                    assert rel_rms(got[n], p.grad) < 1e-4, (it, n)
            grads.append(got)
        n0 = "blocks.0.self_attn.q.weight"
        assert rel_rms(grads[1][n0], grads[0][n0]) > 1e-2    # the inputs changed, so must the gradients
        # eager steps through the same reducer (hooks fire during backward, buckets overlap with it)
        for p in m_e.parameters():
            p.grad = None
        red_e = parallel.BucketedGradAllReduce(m_e.parameters(), bucket_mb=0.5, force=True)
        trainer.training_step(b, m_e)
        red_e.finish()
        for n, p in m_e.named_parameters():
            if p.grad is not None:
                assert rel_rms(p.grad, grads[2][n]) < 1e-4, n
        red.remove()
        red_e.remove()
    finally:
        dist.destroy_process_group()
