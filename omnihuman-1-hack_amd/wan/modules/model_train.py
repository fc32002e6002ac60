"""Training-mode forward of WanModel (autograd through the gfx950 kernels).

Config 3 of BASELINE.json (seaweed_apt/distilled_trainer.py:241-316).  Not
built yet: the inference path is complete; the backward kernels are the next
row of SURVEY.md §8 (A17)."""


def forward_train(model, x, t, context, seq_len, clip_fea=None, y=None):
    raise NotImplementedError(
        "WanModel backward on gfx950 is not built yet: call under torch.no_grad() or "
        "model.requires_grad_(False) for inference")
