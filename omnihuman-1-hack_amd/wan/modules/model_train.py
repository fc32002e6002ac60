"""Training-mode forward/backward of WanModel on the gfx950 kernels — the
student step of seaweed_apt/distilled_trainer.py:241-316 (BASELINE config 3).

The reference gets the backward from autograd over aten ops, with one
``torch.utils.checkpoint`` per block when ``model.use_checkpoint`` is set
(model.py:544-553).  Here autograd only chains three kinds of hand-written
nodes — embed -> 30 x block -> head — so a data-parallel reducer (DDP's, or
parallel.BucketedGradAllReduce) still overlaps its all-reduce with the
backward: each block's parameter gradients become ready when that block's
node finishes.

A block node's forward is ``_block_forward``: the kernels and epilogues of the
inference block (model.py:279-330 on libomh.so; same operands, same roundings,
bit-identical output wherever the inference path takes the same attention kernel: the training forward — kept
activations, the forward under use_checkpoint and its re-run alike — pins the short-sequence kernel
(OMH_ATTN_SHORT_KERNEL), the inference path switches to the long-sequence stream from 512 query tiles
on, e.g. 16 clips x 1560 tokens), with the residual stream written out of place and the
few extra tensors the backward needs emitted by the producing kernels' epilogues
(the branch outputs y = o W^T + b for the gate gradients, the FFN pre-activation
for GELU').  What happens to those tensors is ``model.use_checkpoint``'s call,
as in the reference:

* ``use_checkpoint = False`` (model.py:549-553): they are kept — 0.6 GB per
  block at 4 clips, 18 GB for the 1.3B model, nothing against 288 GB of HBM — and
  the backward starts straight away;
* ``use_checkpoint = True`` (model.py:544-548, the reference's 24 GB default):
  the flag buys memory with compute, and the two modes give the same gradients
  bit for bit, so it is honoured as what it is — a memory policy
  (``model.checkpoint_policy``, default "auto"; OMH_CHECKPOINT_POLICY overrides):
  "auto" keeps the activations when the whole step's worth of them fits in half
  of the HBM that is free when the forward starts (4 clips: 18 GB of 288) and
  otherwise — or with "always", the reference's literal behaviour — keeps only
  the block's input and first re-runs ``_block_forward`` in the backward: the same
  kernels on the same input, hence the same values the loss was computed from.

The backward applies the chain rule by hand: every dgrad / wgrad product is an
MFMA GEMM (``omh_gemm_bf16`` on a transposed weight copy, ``omh_gemm_bf16_tn`` on
dy and x as they are), the attention backward is the fused kernel pair of
csrc/attention_bwd.hip on the forward's own (pre-scaled) q, k, v and log-sum-exp,
GELU' is an epilogue of the FFN dgrad GEMM.  q, k, v share one gradient buffer
[rows, 3 dim], so their input gradient and their weight gradients are ONE GEMM
each (K = 3 dim / M = 3 dim); likewise k, v of the cross-attention.  Gradients
w.r.t. the time-embedding vector ``e0``, the text context and ``e`` are
accumulated in a per-forward state object and consumed by the embed node.

bf16 operand copies of the weights (and the transposed copies the dgrad GEMMs
read) are rebuilt by ONE launch per step (``TrainPacks``, omh_pack_weights_multi)
— after an optimizer step every copy is stale.

Reference quirk kept by default: for ``block_idx > 10``
the reference computes the FFN on the CPU under ``no_grad`` and adds
``0 * ffn_input`` (model.py:317-324), so those FFN weights — and everything
upstream *through that FFN* — receive no gradient; only the gate ``e[5]`` does.
``model.reference_ffn_freeze = False`` turns the quirk off (full gradients).

Gradient accumulation (the reference trainer's mode, distilled_trainer.py:41,116-134,289): when a parameter already
holds a dense fp32 ``.grad``, the block backward adds this micro-step's gradient INTO it (``_grad_targets``) instead of
handing autograd a fresh tensor to add — same values, one pass over the gradients less, and the weight-gradient
stream keeps its end-of-pass join.  ``model.direct_grad_accumulation = False`` restores the autograd route.

The i2v backbone trains too: the image-token branch of the cross-attention
(k_img / v_img / norm_k_img) and img_emb (LayerNorm, Linear, GELU(erf), Linear,
LayerNorm on the CLIP tokens).
"""
import math
import os

import torch

from .._backend import ops

ptr = ops.ptr
EPI_BF16, EPI_F32, EPI_ACC = ops.EPI_BF16, ops.EPI_F32, ops.EPI_F32_ACCUM
EPI_GELU_BF16, EPI_RESID, EPI_GELU_BWD = ops.EPI_GELU_BF16, ops.EPI_RESID, ops.EPI_GELU_BWD_BF16
BIAS_N, BIAS_M, BIAS_NONE = ops.BIAS_N, ops.BIAS_M, ops.BIAS_NONE
LOG2E = 1.4426950408889634


def _ru(a, b):
    return (a + b - 1) // b * b


class _State:
    """Per-forward shared state: the inference context plus gradient accumulators."""

    def __init__(self):
        self.fc = None
        self.d_e0 = self.d_e = self.d_ctx = None


# ----------------------------------------------------------------------------- bf16 weight copies, one launch per step
# id(parameter) -> (weakref to its TrainPacks, row index in TrainPacks.rows, weakref to the parameter): lets the optimizer
# write the bf16 copies of a weight in the same pass that updates it (optim.AdamW, omh_adamw_pack_multi) instead of a
# re-pack launch.  Keyed by the parameter OBJECT, never by its address: a storage that was freed or moved (offload, a
# .to() without a refresh()) can be handed to an unrelated tensor by the caching allocator.
import weakref
_PACK_REGISTRY = {}


def _purge_pack_entries(keys):
    for k in keys:
        _PACK_REGISTRY.pop(k, None)


def pack_entry_of(param):
    """(packs, row) if ``param`` IS the tensor a live TrainPacks keeps bf16 operand copies of, still at the address and
    with the shape the pack table was built for; else None (the optimizer then takes the plain AdamW row and the copies
    are rebuilt by the next refresh())."""
    ent = _PACK_REGISTRY.get(id(param))
    if ent is None:
        return None
    packs, ri = ent[0](), ent[1]
    if packs is None or ent[2]() is not param or packs.table is None or ri >= len(packs.rows):
        return None
    row = packs.rows[ri]
    if packs.params[ri] is not param or row[0] != param.data_ptr() or param.numel() != row[3] * row[4] \
            or param.dtype != torch.float32 or not param.is_contiguous():
        return None
    return packs, ri


class TrainPacks:
    """bf16 operand copies of every block's Linear weights for the training step, in persistent buffers, rebuilt by
    ONE ``omh_pack_weights_multi`` launch whenever a parameter changed (version counters): per block
      wqkv [3d, d] (+ wqkvT [d, 3d], bqkv fp32 [3d]) · wo, woT · cross wq, wqT · cross wkv [2d, d], wkvT, (i2v: wkvi,
      wkviT) · cross wo, woT · w1 [f, d], w1T · w2 [d, f], w2T     (T = the transposed copy the dgrad GEMM reads;
    not built for the FFNs the reference's quirk freezes).  The q | k | v (and cross k | v) weights sit in one buffer
    so that their input gradient and their weight gradient are one GEMM each."""

    def __init__(self, model):
        self.sig = None
        self.table = None
        self.blocks = []
        self.key = None

    @staticmethod
    def of(model):
        tp = model.__dict__.get("_train_packs")
        if tp is None:
            tp = model.__dict__["_train_packs"] = TrainPacks(model)
        return tp

    def _layout(self, model):
        dev = model.patch_embedding.weight.device
        freeze = bool(getattr(model, "reference_ffn_freeze", True))
        rows, self.blocks, self.params = [], [], []
        bf = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=dev)

        def entry(src, dst, dstT, r, c, ld_dst, ld_t, kind=0):
            assert src.dtype == torch.float32 and src.is_contiguous(), "training packs expect fp32 master weights"
            self.params.append(src)
            rows.append([src.data_ptr(), dst.data_ptr() if dst is not None else 0,
                         dstT.data_ptr() if dstT is not None else 0, r, c, ld_dst, ld_t, 0, kind])

        for idx, blk in enumerate(model.blocks):
            sa, ca = blk.self_attn, blk.cross_attn
            d, f = blk.dim, blk.ffn_dim
            P = {}
            P["wqkv"], P["wqkvT"] = bf(3 * d, d), bf(d, 3 * d)
            P["bqkv"] = torch.empty(3 * d, dtype=torch.float32, device=dev)
            for j, lin in enumerate((sa.q, sa.k, sa.v)):
                entry(lin.weight, P["wqkv"][j * d:], P["wqkvT"][:, j * d:], d, d, d, 3 * d)
                entry(lin.bias, P["bqkv"][j * d:], None, 1, d, d, 0, kind=1)
            P["wo"], P["woT"] = bf(d, d), bf(d, d)
            entry(sa.o.weight, P["wo"], P["woT"], d, d, d, d)
            P["wq_c"], P["wq_cT"] = bf(d, d), bf(d, d)
            entry(ca.q.weight, P["wq_c"], P["wq_cT"], d, d, d, d)
            P["wkv_c"], P["wkv_cT"] = bf(2 * d, d), bf(d, 2 * d)
            for j, lin in enumerate((ca.k, ca.v)):
                entry(lin.weight, P["wkv_c"][j * d:], P["wkv_cT"][:, j * d:], d, d, d, 2 * d)
            if hasattr(ca, "k_img"):
                P["wkv_i"], P["wkv_iT"] = bf(2 * d, d), bf(d, 2 * d)
                for j, lin in enumerate((ca.k_img, ca.v_img)):
                    entry(lin.weight, P["wkv_i"][j * d:], P["wkv_iT"][:, j * d:], d, d, d, 2 * d)
            P["wo_c"], P["wo_cT"] = bf(d, d), bf(d, d)
            entry(ca.o.weight, P["wo_c"], P["wo_cT"], d, d, d, d)
            frozen = freeze and idx > 10
            P["w1"], P["w2"] = bf(f, d), bf(d, f)
            P["w1T"], P["w2T"] = (None, None) if frozen else (bf(d, f), bf(f, d))
            entry(blk.ffn[0].weight, P["w1"], P["w1T"], f, d, d, f)
            entry(blk.ffn[2].weight, P["w2"], P["w2T"], d, f, f, d)
            self.blocks.append(P)
        self.rows, self.subsets = rows, {}
        self.table, self.total_tiles = self._table(range(len(rows)), dev)
        self.n = len(rows)
        _purge_pack_entries(self.__dict__.get("_registry_keys", ()))      # a re-layout drops the rows it replaces
        ref = weakref.ref(self)
        self._registry_keys = [id(p) for p in self.params]
        for i, p in enumerate(self.params):
            _PACK_REGISTRY[id(p)] = (ref, i, weakref.ref(p))
        fin = self.__dict__.get("_registry_finalizer")
        if fin is not None:
            fin.detach()
        self._registry_finalizer = weakref.finalize(self, _purge_pack_entries, self._registry_keys)

    def mark_current(self, indices):
        """The optimizer wrote the copies of rows ``indices`` itself (omh_adamw_pack_multi): their versions are current."""
        if self.sig is not None:
            for i in indices:
                self.sig[i] = self.params[i]._version

    def _table(self, idx, dev):
        """Device table of the entries ``idx`` (each with the first tile it owns in the launch) and the tile count."""
        sel, tile0 = [], 0
        for i in idx:
            r = list(self.rows[i])
            r[7] = tile0
            tile0 += (r[3] * r[4] + 4095) // 4096 if r[8] == 1 else ((r[3] + 63) // 64) * ((r[4] + 63) // 64)
            sel.append(r)
        return torch.tensor(sel, dtype=torch.int64).to(dev), tile0

    def refresh(self, model):
        """Make the copies current; returns the per-block dicts."""
        from . import model as _model_mod
        freeze = bool(getattr(model, "reference_ffn_freeze", True))
        dev = model.patch_embedding.weight.device
        if self.table is None or self.key != (freeze, str(dev)) or \
                any(p.data_ptr() != r for p, r in zip(self.params, self.ptrs)):
            self._layout(model)
            self.key, self.ptrs, self.sig = (freeze, str(dev)), [p.data_ptr() for p in self.params], None
        sig = [p._version for p in self.params]
        if self.sig is None:
            ops.pack_weights_multi(self.table, self.n, self.total_tiles)
        elif sig != self.sig:
            # only what the optimizer touched: the FFNs the reference's quirk freezes (523 M of the 1.3B model's
            # 1 419 M packed parameters) keep their copies
            changed = tuple(i for i, (a, b) in enumerate(zip(sig, self.sig)) if a != b)
            sub = self.subsets.get(changed)
            if sub is None:
                if len(self.subsets) > 8:
                    self.subsets.clear()
                sub = self.subsets[changed] = self._table(changed, dev)
            ops.pack_weights_multi(sub[0], len(changed), sub[1])
        self.sig = sig
        return self.blocks


# The weight-gradient GEMMs of a block run on a second HIP stream (OMH_WGRAD_STREAM=0: on the main one).  At 4 clips per
# GPU they are tile-poor like the dgrad GEMMs they are independent of (150 tiles on 256 CUs), so the two fill each
# other's idle CUs.  The side stream forks from the main one at every call (its inputs are ready there) and joins at
# the end of the block's backward (_wgrad_join); hipGraph capture follows the fork / join.
# Measured (round 2): 142.7 -> 138.2 ms per step at 4 clips, 428 -> 413 ms at 16.
_WGRAD_STREAM = os.environ.get("OMH_WGRAD_STREAM", "1") == "1"
# OMH_ATTN_BWD=v1: round 2's attention-backward kernels (three transposed copies + a delta pass over the keys per
# call) instead of round 3's (csrc/attention_bwd2.hip, which read the forward's fp32 output) — A/B timing
_ATTN_BWD2 = os.environ.get("OMH_ATTN_BWD", "v2") != "v1"
# The dQ and the dK / dV kernel of an attention backward on two streams (after a small delta kernel they are
# independent): at 4 clips x 1560 positions each alone leaves most of its last round of workgroups idle (624 on 512 /
# 256 slots).  Measured, one box, interleaved: 86.1 -> 85.4 ms per step at 4 clips (the weight-gradient stream already
# fills most of those gaps), 43.6 -> 44.0 at 1 clip — hence only when a kernel has more than one round of workgroups.
# OMH_ATTN_BWD_SPLIT=0 / 1 forces it off / on.
_ATTN_SPLIT = os.environ.get("OMH_ATTN_BWD_SPLIT", "auto")
# Every attention launch of the training forward — kept activations, the forward under use_checkpoint and its re-run in
# the backward alike — takes the short-sequence kernel (the choice must not depend on whether lse / o32
# are requested) and may split its last round of workgroups over the keys (4 clips: 624 workgroups on 512 slots).
_ATTN_FLAGS = ops.ATTN_SHORT_KERNEL | ops.ATTN_ALLOW_SPLIT
_side = {}
_side2 = {}
# OMH_BLOCK_DX_COPY=1: never update an incoming block gradient in place (debugging aid for callers with extra taps)
_ALWAYS_COPY_DX = os.environ.get("OMH_BLOCK_DX_COPY", "0") == "1"


def _attn_bwd(q, k, v, o, do, lse, lens, B, N, Lq, Lk, scale, out, o32, q_prescaled):
    """ops.flash_attn_bwd on the block's pre-scaled q, bf16 gradients into ``out`` — as one call, or (OMH_ATTN_BWD_SPLIT)
    as delta -> {dQ on this stream, dK / dV on a second one} -> join."""
    split = _ATTN_SPLIT == "1" or (_ATTN_SPLIT == "auto" and B * N * ((Lq + 127) // 128) > 512)
    if not (split and o32 is not None):
        return ops.flash_attn_bwd(q, k, v, o, do, lse, lens, B, N, Lq, Lk, scale, q_prescaled=q_prescaled, out=out, o32=o32)
    dev = q.device
    delta = torch.empty(B, N, Lq, dtype=torch.float32, device=dev)
    kw = dict(q_prescaled=q_prescaled, out=out, o32=o32, delta=delta)
    ops.flash_attn_bwd(q, k, v, o, do, lse, lens, B, N, Lq, Lk, scale, phase=1, **kw)
    main = torch.cuda.current_stream(dev)
    s2 = _side2.get(dev)
    if s2 is None:
        s2 = _side2[dev] = torch.cuda.Stream(device=dev)
    s2.wait_stream(main)
    try:
        with torch.cuda.stream(s2):
            ops.flash_attn_bwd(q, k, v, o, do, lse, lens, B, N, Lq, Lk, scale, phase=3, **kw)
        ops.flash_attn_bwd(q, k, v, o, do, lse, lens, B, N, Lq, Lk, scale, phase=2, **kw)
    finally:
        main.wait_stream(s2)                                 # also on an exception: nothing may stay on the second stream
    return out


# OMH_WGRAD_CU_MASK = k (1..31): the weight-gradient stream is confined to k of every 32 CUs (hipExtStreamCreateWithCUMask
# through ops.cu_masked_stream) — an A/B switch for the interference of that stream with the main stream's attention
# backward (measured and NOT adopted, DESIGN.md 4.2: the default is an ordinary stream).
_WGRAD_CU_MASK = int(os.environ.get("OMH_WGRAD_CU_MASK", "0") or 0)
_WG_LATE = os.environ.get("OMH_WG_LATE", "0") == "1"


def _side_stream(dev):
    s = _side.get(dev)
    if s is None:
        if 0 < _WGRAD_CU_MASK < 32:
            with torch.cuda.device(dev):
                s = _side[dev] = ops.cu_masked_stream(_WGRAD_CU_MASK)
        else:
            s = _side[dev] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("OMH_WGRAD_PRIO", "0")))
    return s


def _wgrad_join(dev):
    if _WGRAD_STREAM and dev in _side:
        torch.cuda.current_stream(dev).wait_stream(_side[dev])


def join_side_streams(dev=None):
    """The current stream waits for the weight-gradient stream (all devices, or ``dev``).  For code that reads parameter
    gradients INSIDE a backward pass (a gradient hook): with the deferred join (below) a block's weight gradients may still be
    in flight on the second stream when its backward returns (parallel.BucketedGradAllReduce instead packs its buckets ON
    that stream: reducer_stream).  After ``backward()`` has returned nothing is in flight."""
    for d in ([dev] if dev is not None else list(_side)):
        d = torch.device(d) if not isinstance(d, torch.device) else d
        if d in _side:
            torch.cuda.current_stream(d).wait_stream(_side[d])


def reducer_stream(dev):
    """The stream a gradient reducer should pack and launch from inside a backward pass: the weight-gradient stream, made
    to wait for what the main stream has queued so far (None when the step runs on one stream)."""
    if not _WGRAD_STREAM:
        return None
    dev = torch.device(dev) if not isinstance(dev, torch.device) else dev
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    sd = _side_stream(dev)
    sd.wait_stream(torch.cuda.current_stream(dev))
    return sd


# The join of the weight-gradient stream at the END OF THE BACKWARD PASS instead of the end of every block's backward
# (OMH_WGRAD_DEFER=0: per block, as rounds 2-3).  A block's last group (q|k|v) is launched when its backward is nearly
# done; joining there made the main stream wait for it with most of the chip idle, 30 times per step.  Deferred, the
# group of block i runs under the backward of block i-1.  Legal only when nothing reads the gradients before the pass
# ends: every trainable parameter's .grad is None (autograd then just stores the tensor) or is accumulated into by the
# block backward itself (_grad_targets below; an existing .grad that autograd has to add to is added to on the main
# stream), no tensor / post-accumulate hooks on the parameters other than this package's
# reducer (which packs its buckets on that stream: reducer_stream), and no other multi-process gradient reducer in sight (torch DDP hooks the
# accumulator nodes, which cannot be seen from here: with torch.distributed initialised on more than one rank and none
# of our reducer's hooks on the parameters the join stays per block).
_DEFER_JOIN = os.environ.get("OMH_WGRAD_DEFER", "1") == "1"
# what else rides the second stream (OMH_SIDE_KV=0 / OMH_SIDE_BIAS=0: on the main one, A/B timing): the cross-attention's
# key / value gradient path and the bias gradients' column sums — neither feeds the block's input gradient
# the norm backwards' second launches (partial column sums -> parameter / modulation gradients) of a block's main stream
# collected into ONE launch at the end of the block (ops.partial_colsum_multi).  Bit-identical, four launches fewer per
# block — and measured EQUAL (73.2 / 73.3 vs 73.1 / 73.3 ms at 4 clips, 38.3 / 38.7 vs 37.0 / 39.5 at one clip, interleaved
# on one box: the 5 us kernels were not what the main stream waits for): opt-in, OMH_DEFER_COLSUM=1.
_DEFER_COLSUM = os.environ.get("OMH_DEFER_COLSUM", "0") == "1"
_SIDE_KV = os.environ.get("OMH_SIDE_KV", "1") == "1"
_SIDE_BIAS = os.environ.get("OMH_SIDE_BIAS", "0") == "1"      # measured: 75.3 ms with, 75.1 without (one box, interleaved)
# (tried and dropped: the forward's context keys / values and the backward's V transposes on a third stream — 75.55 ms
# with, 75.4 without at 4 clips, slower at 1 clip: the extra fork / join costs what the overlap wins)
_join_pending = {}                      # (device, id(model)) -> token of the pass whose end-of-pass join is queued


def _block_params(model, idx=None):
    """[(name, parameter)] of block ``idx`` (None: the parameters of all blocks), cached on the model: walking the module
    tree 60 times per step is ~2 ms of host time, and the one-clip backward is bound by the host (tools/
    host_phase_probe.py).  Parameter OBJECTS are stable under load_state_dict / .to() / requires_grad_; a model whose
    blocks are replaced gets a new cache through the length / identity check."""
    cache = model.__dict__.get("_omh_block_params")
    blocks = model.blocks
    if cache is None or len(cache[0]) != len(blocks) or any(c[0] is not b for c, b in zip(cache[0], blocks)):
        per = [(b, list(b.named_parameters())) for b in blocks]
        cache = model.__dict__["_omh_block_params"] = (per, [p for _, lst in per for _, p in lst])
    return cache[1] if idx is None else cache[0][idx][1]


# Gradient accumulation (the reference trainer's 16 micro-steps, distilled_trainer.py:41,116-134,289): from the second
# micro-step on every parameter has a .grad, autograd would add each block's fresh gradients to it on the main stream
# (3.6 GB read + 3.6 GB written per micro-step at 1.3B) and the join of the weight-gradient stream would fall back to
# once per block.  Instead the block backward accumulates INTO the existing .grad tensors — the weight-gradient GEMMs
# through their accumulate epilogue on the second stream, the bias / gain / modulation sums through kernels that add
# into their output anyway — and hands autograd None for those parameters.  fp32 arithmetic is the same
# (old + (sum over the rows)): matrices bit-identical to the autograd route.  OMH_GRAD_ACCUM_DIRECT=0 or
# ``model.direct_grad_accumulation = False`` restore the autograd route.
_DIRECT_ACCUM = os.environ.get("OMH_GRAD_ACCUM_DIRECT", "1") == "1"


def _direct_on(model):
    return _DIRECT_ACCUM and getattr(model, "direct_grad_accumulation", True)


def _direct_usable(p):
    """``p.grad`` is something the kernels can add into in place: dense fp32 of the parameter's own shape."""
    g = p.grad
    return (g is not None and g.dtype == torch.float32 and p.dtype == torch.float32 and g.shape == p.shape
            and g.layout == torch.strided and g.is_contiguous() and g.device == p.device and not g.requires_grad)


def _grad_targets(ctx, model, idx):
    """The existing ``.grad`` tensors block ``idx``'s backward may accumulate into: {name: parameter} ({} when the block
    has none: first micro-step), or None when gradients exist but the autograd route has to be taken — a gradient of
    another dtype / layout, tensor hooks or foreign post-accumulate hooks on a parameter (they would not fire), a
    multi-process reducer that is not this package's (torch DDP hooks the accumulator nodes and would wait for them
    forever), double backward, or a pass that does not run the parameters' AccumulateGrad nodes at all
    (``torch.autograd.grad(loss, params)``: the caller wants the gradients returned, not accumulated)."""
    have = [(n, p) for n, p in _block_params(model, idx) if p.requires_grad and p.grad is not None]
    if not have:
        return {}
    if not _direct_on(model) or torch.is_grad_enabled():
        return None
    st = getattr(ctx, "st", None)
    if any(o is not st and not o.__dict__.get("bwd_done", False) for o in model.__dict__.get("_omh_live_states", ())):
        # another forward of this model awaits its backward in the same pass: a reducer may launch a bucket after the first
        # contribution and the second would be added under its collective — autograd's route sums first
        return None
    ours = False
    for _, p in have:
        if not _direct_usable(p) or getattr(p, "_backward_hooks", None):
            return None
        for h in (getattr(p, "_post_accumulate_grad_hooks", None) or {}).values():
            if not getattr(getattr(h, "__self__", None), "_omh_joins_side_streams", False):
                return None
            ours = True
    if not ours and torch.distributed.is_available() and torch.distributed.is_initialized():
        return None
    nodes = {id(fn.variable): fn for fn, _ in ctx.next_functions if fn is not None and hasattr(fn, "variable")}
    try:                                                      # (backward(inputs=[...]): only the named leaves accumulate)
        have = [(n, p) for n, p in have if id(p) in nodes and torch._C._will_engine_execute_node(nodes[id(p)])]
    except Exception:                                         # autograd.grad(..., inputs=[p]): the engine says so by raising
        return None
    return dict(have) if have else None


def _grad_slots(ctx, model, idx, tgt):
    """Destinations for the FRESH weight gradients of block ``idx``: {name: fp32 view of the parameter's shape} for the
    matrices that have no ``.grad`` yet and whose reducer (this package's parallel.BucketedGradAllReduce, found through
    its post-accumulate hook) offers their slice of a flat bucket (``grad_slot``).  The weight-gradient GEMM then writes
    straight into the bucket — nothing to pack before the collective (the 3.6 GB packing copy)
    — the block sets ``p.grad`` to the slice, tells the reducer, and hands autograd None, exactly as the accumulation
    path does for gradients that already exist.  Same conditions as _grad_targets; {} whenever they do not hold."""
    if tgt is None or not _direct_on(model) or torch.is_grad_enabled():
        return {}
    st = getattr(ctx, "st", None)
    if any(o is not st and not o.__dict__.get("bwd_done", False) for o in model.__dict__.get("_omh_live_states", ())):
        # another forward of this model still awaits its backward: its contribution would be added in
        # place into a bucket slice whose collective this pass may already have started — no slots, autograd sums
        return {}
    out, nodes = {}, None
    for n, p in _block_params(model, idx):
        if not p.requires_grad or p.grad is not None or p.dim() != 2 or p.dtype != torch.float32 or (tgt and n in tgt):
            continue
        if getattr(p, "_backward_hooks", None):
            continue
        hooks = list((getattr(p, "_post_accumulate_grad_hooks", None) or {}).values())
        if len(hooks) != 1 or not getattr(getattr(hooks[0], "__self__", None), "_omh_joins_side_streams", False):
            continue
        red = hooks[0].__self__
        slot = red.grad_slot(p) if hasattr(red, "grad_slot") else None
        if slot is None:
            continue
        if nodes is None:
            nodes = {id(fn.variable): fn for fn, _ in ctx.next_functions if fn is not None and hasattr(fn, "variable")}
        try:
            if id(p) not in nodes or not torch._C._will_engine_execute_node(nodes[id(p)]):
                continue
        except Exception:
            return {}
        out[n] = slot
    return out


def _may_defer_join(model):
    if not (_DEFER_JOIN and _WGRAD_STREAM):
        return False
    ours = False
    direct = _direct_on(model)
    for p in _block_params(model):                           # (only the blocks' products run on the second stream)
        if not p.requires_grad:
            continue
        if getattr(p, "_backward_hooks", None):
            return False
        if p.grad is not None and not (direct and _direct_usable(p)):      # autograd would add to it on the main stream
            return False
        for h in (getattr(p, "_post_accumulate_grad_hooks", None) or {}).values():
            if not getattr(getattr(h, "__self__", None), "_omh_joins_side_streams", False):
                return False
            ours = True
    if not ours and torch.distributed.is_available() and torch.distributed.is_initialized() and \
            torch.distributed.get_world_size() > 1:
        return False
    return True


def _wgrad(dy, x, out=None):
    """dW[N, K] (+)= dy[R, N]^T @ x[R, K]  (fp32) on dy and x as they are (row-major bf16, row stride free): the
    k-major GEMM of csrc/gemm_tn.hip — no transposed copies.  With ``out`` the product is ADDED to it."""
    return ops.gemm_tn(dy, x, out=out, accumulate=out is not None)


def _on_side(dev, fn, tensors=()):
    """``fn()`` on the weight-gradient stream, behind everything the main stream has queued so far (its inputs were
    produced there); ``tensors``: what fn reads or writes that may be freed before the stream gets to it."""
    if not _WGRAD_STREAM:
        return fn()
    main, sd = torch.cuda.current_stream(dev), _side_stream(dev)
    sd.wait_stream(main)
    with torch.cuda.stream(sd):
        fn()
    for tt in tensors:
        if tt is not None:
            tt.record_stream(sd)


class _WgradGroup:
    """The weight-gradient products of a block's backward, collected and launched a few at a time as ONE grouped GEMM
    (omh_gemm_bf16_tn_grouped: every tile its whole K range, no split K, no atomics) on the second stream.  Each
    product alone is tile-poor (1536 x 1536 over 6 240 rows = 144 tiles on 512 slots, 310 TFLOP/s with a 3-way split
    K and atomics); two or three of them together fill the chip."""

    def __init__(self, dev):
        self.dev, self.items = dev, []

    def add(self, dy, x, out=None, store=False):
        """``out``: added to (gradient accumulation) — or, with ``store``, simply written: a destination the caller owns
        (the parameter's slice of a reducer bucket, _grad_slots)."""
        acc = out is not None and not store
        if out is None:
            out = torch.empty(dy.shape[1], x.shape[1], dtype=torch.float32, device=self.dev)
        if dy.shape[1] * x.shape[1] >= 256 * 256 * 128:          # >= 128 tiles of 256 x 256 (the FFN weights): fills the
            self._run([(dy, x, out, acc)], single=True)          # chip alone, on the 8-wave 256 x 256 kernel, at once
        else:
            self.items.append((dy, x, out, acc))
        return out

    def _run(self, items, single=False):
        def go():
            if single:
                dy, x, out, acc = items[0]
                ops.gemm_tn(dy, x, out=out, accumulate=acc)
            else:
                ops.gemm_tn_grouped(items)
        if not _WGRAD_STREAM:
            return go()
        main, sd = torch.cuda.current_stream(self.dev), _side_stream(self.dev)
        sd.wait_stream(main)                                  # the operands were produced on the main stream
        with torch.cuda.stream(sd):
            go()
        for dy, x, out, _ in items:
            for tt in (dy, x, out):
                tt.record_stream(sd)

    def launch(self):
        if self.items:
            items, self.items = self.items, []
            self._run(items)


class _ZeroArena:
    """One zero-filled fp32 buffer per block backward, carved into the ~20 small accumulators (bias, norm-gain and
    modulation gradients) that the atomics-based kernels add into: one fill launch instead of twenty."""

    def __init__(self, n, device):
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.pos = 0
        self.pending = []                                       # deferred column sums: (dy, out) pairs

    def flush(self):
        """All bias gradients of the block in one launch (each alone is a 10-15 us latency-bound kernel)."""
        if self.pending:
            ops.colsum_accum_multi(self.pending)
            self.pending = []

    def take(self, *shape):
        n = 1
        for s_ in shape:
            n *= s_
        n_al = (n + 63) // 64 * 64                              # keep every slice 256-byte aligned
        if self.pos + n_al > self.buf.numel():                  # undersized estimate: fall back to a fresh buffer
            return torch.zeros(*shape, dtype=torch.float32, device=self.buf.device)
        out = self.buf[self.pos:self.pos + n].view(*shape)
        self.pos += n_al
        return out


def _bgrad(dy, arena=None):
    """Bias gradient = column sums of dy.  With an arena the sum is deferred to ``arena.flush()`` at the end of the
    block backward (dy stays referenced until then); the returned accumulator is complete after the flush."""
    out = arena.take(dy.shape[1]) if arena is not None else torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device)
    if arena is not None:
        arena.pending.append((dy, out))
        return out
    return ops.colsum_accum(dy, out)


def _dgrad(dy, wT, out=None, accumulate=False, epilogue=None, split_k=False):
    """dx[R, K] = dy[R, N] @ W[N, K]  with wT = W^T bf16 [K, N] (row stride free); fp32 output unless ``epilogue``.
    ``split_k``: ops.gemm_raw's (the FFN-up input gradient: N = ffn_dim)."""
    R, N = dy.shape
    K = wT.shape[0]
    epi = epilogue if epilogue is not None else (EPI_ACC if accumulate else EPI_F32)
    if out is None:
        out = torch.empty(R, K, dtype=torch.bfloat16 if epi == EPI_BF16 else torch.float32, device=dy.device)
    ops.gemm_raw(ptr(dy), ptr(wT), ptr(out), R, K, N, dy.stride(0), wT.stride(0), out.stride(0), epi, split_k=split_k)
    return out


def _wT_once(weight_bf16):
    return ops.transpose_bf16(weight_bf16)


def _wT(mod, key, weight_bf16):
    """Transposed bf16 copy [K, N] of a packed [N, K] weight outside the blocks (head), cached per weight version."""
    return mod._packed.get("T:" + key, (weight_bf16,), lambda: ops.transpose_bf16(weight_bf16))


_VT_CACHE_MAX = 256


def _vt_buffer(model, key, B, d, Lp, L, dev, keep):
    """V^T [B, d, Lp] bf16 with zeroed pad columns [L, Lp) (0 x P needs them finite).  Kept activations get their own
    tensor (two forwards may be in flight before a backward); a transient one (use_checkpoint: produced and consumed
    inside one node) comes from a per-model cache whose pad was zeroed once."""
    if keep:
        t = torch.empty(B, d, Lp, dtype=torch.bfloat16, device=dev)
        if Lp != L:
            t[:, :, L:].zero_()
        return t
    cache = model.__dict__.setdefault("_vt_cache", {})
    k = (key[1:], B, d, Lp, L, str(dev))
    t = cache.get(k)
    if t is None:
        if len(cache) > _VT_CACHE_MAX:
            cache.clear()
        t = cache[k] = torch.zeros(B, d, Lp, dtype=torch.bfloat16, device=dev)
    return t


# ----------------------------------------------------------------------------- block forward (training)
def _block_forward(model, blk, idx, st, x0, P, keep, need=True):
    """The inference block (model.py:279-330 on libomh.so, WanAttentionBlock.forward of this package) with the
    residual stream out of place and the backward's extra tensors emitted by the producing epilogues.  x0 fp32
    [B, S, d] is left untouched.  Returns (x3, S) — S holds what ``_block_backward`` reads.  ``need = False`` (the
    forward pass under use_checkpoint: the block is run again in the backward): the extras — branch outputs, FFN
    pre-activation, log-sum-exp, fp32 attention output — are not produced; the values of x3 do not depend on it."""
    fc = st.fc
    B, Sq, d = x0.shape
    R = B * Sq
    dev = x0.device
    sa, ca = blk.self_attn, blk.cross_attn
    N, D = sa.num_heads, sa.head_dim
    f = blk.ffn_dim
    mod = blk.modulation.detach()
    if mod.dtype != torch.float32:
        mod = mod.float()
    mod = mod.contiguous()
    e0 = fc.e0
    six = 6 * d
    i2v = hasattr(ca, "k_img")
    n_img = 257 if i2v else 0
    frozen_ffn = getattr(model, "reference_ffn_freeze", True) and idx > 10
    Lc = fc.Lc
    Lt = Lc - n_img
    bf = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=dev)
    S = {"x0": x0}

    def ctx_kv(lo, L, wkv, k_lin, v_lin, norm_name, key):
        """K (normalised bf16 [B*L, d], its fp32 pre-norm) and V ([B*L, d] for the backward, V^T for the forward)
        of context rows [lo, lo + L) — model.py:176-178 / 216-220 as WanT2VCrossAttention._context_kv computes them."""
        Lp = _ru(L, 64)
        kf = torch.empty(B, L, d, dtype=torch.float32, device=dev)
        ops.gemm_raw(ptr(fc.ctx, lo * d), ptr(wkv), ptr(kf), L, d, d, d, d, d, EPI_F32, bias=ptr(k_lin.bias.detach()),
                     bias_mode=BIAS_N, batch=B, strideA=Lc * d, strideB=0, strideC=L * d)
        kn = ops.rmsnorm_rope(kf.view(B * L, d), ca._norm_w(norm_name), ca.eps, do_norm=ca.qk_norm)
        vtc = _vt_buffer(model, (idx, key), B, d, Lp, L, dev, keep)
        ops.gemm_raw(ptr(wkv, d * d), ptr(fc.ctx, lo * d), ptr(vtc), d, L, d, d, d, Lp, EPI_BF16,
                     bias=ptr(v_lin.bias.detach()), bias_mode=BIAS_M, batch=B, strideA=0, strideB=Lc * d, strideC=d * Lp)
        return kf, kn, vtc, Lp

    def ln_mod(xin, shift_i, scale_i):
        h = bf(R, d)
        ops.layernorm_modulate_raw(ptr(xin), ptr(h), R, d, blk.eps, 1.0, ptr(mod, scale_i * d), ptr(e0, scale_i * d), six,
                                   ptr(mod, shift_i * d), ptr(e0, shift_i * d), six, Sq)
        return h

    def resid(xin, a, w, b, gate_i, want_y, xout=None):
        """x_out = x_in + (a w^T + b) * gate  (+ y = bf16(a w^T + b) when the gate gets a gradient)."""
        xo = torch.empty_like(xin) if xout is None else xout
        y = bf(R, d) if (want_y and need) else None
        M, K = a.shape
        kw = dict(c_in=ptr(xin) if xo.data_ptr() != xin.data_ptr() else None, aux=ptr(y) if y is not None else None, ldaux=d)
        if gate_i is None:
            ops.gemm_raw(ptr(a), ptr(w), ptr(xo), M, d, K, a.stride(0), w.stride(0), d, EPI_RESID,
                         bias=ptr(b) if b is not None else None, bias_mode=BIAS_N if b is not None else BIAS_NONE,
                         gate_const=1.0, split_k=fc.split_k, **kw)
        else:
            ops.gemm_raw(ptr(a), ptr(w), ptr(xo), M, d, K, a.stride(0), w.stride(0), d, EPI_RESID, bias=ptr(b),
                         bias_mode=BIAS_N, gate0=ptr(mod, gate_i * d), gate1=ptr(e0, gate_i * d), gate1_stride=six,
                         gate_rows=Sq, gate_const=0.0, split_k=fc.split_k, **kw)      # (same slices as the inference block: model.py)
        return xo, y

    # ---- self-attention: x1 = x0 + o(attn(LN(x0)(1+e1)+e0)) * e2                                    model.py:292-296
    h1 = ln_mod(x0, 0, 1)
    wqkv, bqkv = P["wqkv"], P["bqkv"]
    qk = bf(R, 2 * d)                                       # q | k projection, bf16 like the inference path
    ops.gemm_raw(ptr(h1), ptr(wqkv), ptr(qk), R, 2 * d, d, d, d, 2 * d, EPI_BF16, bias=ptr(bqkv), bias_mode=BIAS_N)
    q, k = bf(R, d), bf(R, d)
    nq, nk = sa._norm_w("norm_q"), sa._norm_w("norm_k")
    ops.rmsnorm_rope_bf16_pair_raw(ptr(qk), 2 * d, d, ptr(q), ptr(k), R, d, ptr(nq) if nq is not None else None,
                                   ptr(nk) if nk is not None else None, sa.eps, int(sa.qk_norm), ptr(fc.rope_cos),
                                   ptr(fc.rope_sin), fc.rope_cos.shape[0], D, ptr(fc.grid32), Sq,
                                   out_scale0=D ** -0.5 * LOG2E, out_scale1=1.0)      # q and k: one launch, as model.py
    Sp = _ru(Sq, 64)
    vt = _vt_buffer(model, (idx, "sa"), B, d, Sp, Sq, dev, keep)
    ops.gemm_raw(ptr(wqkv, 2 * d * d), ptr(h1), ptr(vt), d, Sq, d, d, d, Sp, EPI_BF16, bias=ptr(bqkv, 2 * d),
                 bias_mode=BIAS_M, batch=B, strideA=0, strideB=Sq * d, strideC=d * Sp)
    o = bf(R, d)
    f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    lse_sa = f32(B, N, Sq) if need else None
    o32_sa = f32(R, d) if (_ATTN_BWD2 and need) else None     # the output before its rounding: delta of the backward
    ops.flash_attn_raw(ptr(q), ptr(k), ptr(vt), ptr(o), ptr(fc.seq_lens32), B, N, Sq, Sq, Sq * d, d, Sq * d, d, d * Sp,
                       Sq * d, d, Sp, D ** -0.5, lse=ptr(lse_sa) if need else None, q_prescaled=1,
                       o32=ptr(o32_sa) if o32_sa is not None else None, flags=_ATTN_FLAGS)
    x1, y1 = resid(x0, o, P["wo"], sa.o.bias.detach(), 2, True)
    S.update(h1=h1, qk=qk, q=q, k=k, vt=vt, o=o, lse_sa=lse_sa, y1=y1, x1=x1, o32_sa=o32_sa)
    # ---- cross-attention: x2 = x1 + o(attn(norm3(x1), context))                                     model.py:313
    h3 = bf(R, d)
    if blk.cross_attn_norm:
        n3 = blk.norm3
        w3, b3 = n3.weight.detach().float().contiguous(), n3.bias.detach().float().contiguous()
        ops.layernorm_modulate_raw(ptr(x1), ptr(h3), R, d, n3.eps, 0.0, ptr(w3), None, 0, ptr(b3), None, 0, R)
        S.update(w3=w3)
    else:
        ops.cast_bf16(x1.view(R, d), out=h3)
    qcb = ops.gemm(h3, P["wq_c"], bias=ca.q.bias.detach(), epilogue=EPI_BF16)
    qc = bf(R, d)
    wnq = ca._norm_w("norm_q")
    ops.rmsnorm_rope_bf16_raw(ptr(qcb), d, ptr(qc), R, d, ptr(wnq) if wnq is not None else None, ca.eps, int(ca.qk_norm),
                              None, None, 0, D, None, Sq)

    kf, kc, vtc, Ltp = ctx_kv(n_img, Lt, P["wkv_c"], ca.k, ca.v, "norm_k", "ca")
    oc = bf(R, d)
    lse_ca = f32(B, N, Sq) if need else None
    o32_ca = f32(R, d) if (_ATTN_BWD2 and need and not i2v) else None
    # the reference passes the (text + 257) lengths here (model.py:223,537); keys are clipped to the text rows
    ops.flash_attn_raw(ptr(qc), ptr(kc), ptr(vtc), ptr(oc), ptr(fc.ctx_lens32), B, N, Sq, Lt, Sq * d, d, Lt * d, d,
                       d * Ltp, Sq * d, d, Ltp, D ** -0.5, lse=ptr(lse_ca) if need else None,
                       o32=ptr(o32_ca) if o32_ca is not None else None, flags=_ATTN_FLAGS)
    x2, _ = resid(x1, oc, P["wo_c"], ca.o.bias.detach(), None, False)
    S.update(h3=h3, qcb=qcb, qc=qc, kf=kf, kc=kc, vtc=vtc, oc=oc, lse_ca=lse_ca, x2=x2, o32_ca=o32_ca)
    if i2v:                                                  # model.py:189-230: extra attention over the 257 image tokens
        kfi, ki, vti, Lip = ctx_kv(0, n_img, P["wkv_i"], ca.k_img, ca.v_img, "norm_k_img", "ci")
        oi = bf(R, d)
        lse_ci = f32(B, N, Sq) if need else None
        ops.flash_attn_raw(ptr(qc), ptr(ki), ptr(vti), ptr(oi), None, B, N, Sq, n_img, Sq * d, d, n_img * d, d, d * Lip,
                           Sq * d, d, Lip, D ** -0.5, lse=ptr(lse_ci) if need else None, flags=_ATTN_FLAGS)
        resid(x2, oi, P["wo_c"], None, None, False, xout=x2)          # x2 += o_img Wo^T   (in place)
        S.update(kfi=kfi, ki=ki, vti=vti, oi=oi, lse_ci=lse_ci)
    # ---- FFN: x3 = x2 + (W2 gelu(W1 (LN(x2)(1+e4)+e3) + b1) + b2) * e5                               model.py:314-328
    h2 = ln_mod(x2, 3, 4)
    u = bf(R, f)
    u_pre = None if (frozen_ffn or not need) else bf(R, f)
    ops.gemm_raw(ptr(h2), ptr(P["w1"]), ptr(u), R, f, d, d, d, f, EPI_GELU_BF16, bias=ptr(blk.ffn[0].bias.detach()),
                 bias_mode=BIAS_N, aux=ptr(u_pre) if u_pre is not None else None, ldaux=f)
    x3, y3 = resid(x2, u, P["w2"], blk.ffn[2].bias.detach(), 5, True)
    S.update(y3=y3)
    if not frozen_ffn:
        S.update(h2=h2, u=u, u_pre=u_pre)
    return x3, S


# ----------------------------------------------------------------------------- block backward
def _block_backward(model, blk, idx, st, S, dx, P, tgt=None, slots=None):
    """Back-propagate dx (fp32 [B, S, d], updated in place) through block ``idx`` given its kept tensors ``S``.
    ``tgt`` ({name: parameter}, _grad_targets): parameters whose existing ``.grad`` the gradient is ADDED to in place
    (gradient accumulation); the returned dict then holds that ``.grad`` tensor under the name."""
    fc = st.fc
    x0 = S["x0"]
    B, Sq, d = x0.shape
    R = B * Sq
    dev = x0.device
    sa, ca = blk.self_attn, blk.cross_attn
    N, D = sa.num_heads, sa.head_dim
    f = blk.ffn_dim
    eps = blk.eps
    mod = blk.modulation.detach().float().contiguous()
    e0 = fc.e0
    six = 6 * d
    i2v = hasattr(ca, "k_img")
    n_img = 257 if i2v else 0
    frozen_ffn = getattr(model, "reference_ffn_freeze", True) and idx > 10
    Lc = fc.Lc
    Lt = Lc - n_img
    bf = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=dev)
    arena = _ZeroArena(B * six + 2 * six + 20 * d + 2 * f + 4096, dev)
    wg = _WgradGroup(dev)
    d_eb = arena.take(B, 6, d)                                        # grads of e = modulation + e0
    g = {}
    tgt = tgt or {}
    slots = slots or {}                                               # fresh weight gradients written into reducer buckets
    deferred = [] if _DEFER_COLSUM else None                          # second launches of the norm backwards (main stream)

    def acc1(name, n):
        """The fp32 accumulator of a 1-D gradient: the parameter's own .grad (added to) or a zeroed arena slice."""
        p = tgt.get(name)
        return p.grad.view(n) if p is not None else arena.take(n)

    def bgrad(dy, names, now=False):
        """Bias gradients = column sums of dy, deferred to arena.flush() at the end of the block (``now``: launched here,
        on the current stream); ``names``: the parameters stacked along dy's columns, each ``d`` wide when there are
        several.  One pair for the whole width unless a parameter accumulates into its own .grad."""
        if len(names) == 1 or not any(n in tgt for n in names):
            out = acc1(names[0], dy.shape[1]) if len(names) == 1 else arena.take(dy.shape[1])
            pairs = [(dy, out)]
            for j, n in enumerate(names):
                g[n] = out if len(names) == 1 else out[j * d:(j + 1) * d]
        else:
            pairs = []
            for j, n in enumerate(names):
                g[n] = acc1(n, d)
                pairs.append((dy[:, j * d:(j + 1) * d], g[n]))
        if now:
            ops.colsum_accum_multi(pairs)
        else:
            arena.pending.extend(pairs)

    def wgrad(dy, x, names):
        """Weight gradients dy^T x into g[names] (parameters stacked along dy's columns, ``d`` wide each): one product,
        or one per parameter — same tiles, same launch — when a parameter accumulates into its own .grad."""
        def one(dy_, n):
            p = tgt.get(n)
            if p is not None:
                return wg.add(dy_, x, out=p.grad)
            if n in slots:
                return wg.add(dy_, x, out=slots[n], store=True)
            return wg.add(dy_, x)
        if len(names) == 1:
            g[names[0]] = one(dy, names[0])
        elif not any(n in tgt or n in slots for n in names):
            out = wg.add(dy, x)
            for j, n in enumerate(names):
                g[n] = out[j * d:(j + 1) * d]
        else:
            for j, n in enumerate(names):
                g[n] = one(dy[:, j * d:(j + 1) * d], n)

    def ln_bwd(xin, dh, shift_i, scale_i, nxt=None):
        """dx += LN+modulate backward of dh; ``nxt = (y, gate_i)``: also the next branch's gated-residual backward on
        the finished rows (dy_next = bf16(dx * gate), the gate's gradient) — returns dy_next."""
        dy_next = bf(R, d) if nxt is not None else None
        kw = {}
        if nxt is not None:
            y, gi = nxt
            if gi is None:
                kw = dict(dy_next=dy_next, gate_const=1.0)
            else:
                kw = dict(dy_next=dy_next, y_next=y, gate_const=0.0, gate0=ptr(mod, gi * d), gate1=ptr(e0, gi * d),
                          gate1_stride=six, dgate=ptr(d_eb, gi * d), dgate_stride=six)
        ops.layernorm_modulate_bwd2(xin, dh, dx, R, d, eps, 1.0, ptr(mod, scale_i * d), ptr(e0, scale_i * d), six,
                                    ptr(d_eb, scale_i * d), ptr(d_eb, shift_i * d), six, Sq, defer=deferred, **kw)
        return dy_next

    def resid_bwd(y, gate_i):
        dy = bf(R, d)
        if gate_i is None:
            ops.gated_residual_bwd_raw(ptr(dx), None, ptr(dy), None, 0, R, d, 1.0, None, None, 0, Sq)
        else:
            ops.gated_residual_bwd_raw(ptr(dx), ptr(y), ptr(dy), ptr(d_eb, gate_i * d), six, R, d, 0.0,
                                       ptr(mod, gate_i * d), ptr(e0, gate_i * d), six, Sq)
        return dy

    def rms_bwd(x, x_bf16, ldx, dy, lddy, rows, weights, norm_on, rope, names, mod_, n_seg=1, seg_x=0, seg_dy=0, defer=True):
        """In place on dy (bf16): dy <- gradient of the pre-norm projection; the norm gains' gradients into g[names]
        (``defer`` False: summed at once — the call that may run on the second stream)."""
        dws = [acc1(nm, d) if norm_on else None for nm in names[:n_seg]]
        rk = dict(rope_cos=ptr(fc.rope_cos), rope_sin=ptr(fc.rope_sin), rope_len=fc.rope_cos.shape[0], grid=ptr(fc.grid32),
                  seq_len=Sq) if rope else {}
        ops.rmsnorm_rope_bwd2(x, x_bf16, ldx, dy, True, lddy, dy, lddy, rows, d, mod_.eps, norm_on,
                              [w if norm_on else None for w in weights], dws, dev, n_seg=n_seg, seg_x=seg_x, seg_dy=seg_dy,
                              seg_dx=seg_dy, head_dim=D, defer=deferred if defer else None, **rk)
        for nm, dw in zip(names, dws):
            if dw is not None:
                g[nm] = dw

    # ---- FFN branch: x3 = x2 + y3 * g5
    if not frozen_ffn:
        dy3 = resid_bwd(S["y3"], 5)
        u, u_pre, h2 = S["u"], S["u_pre"], S["h2"]
        wgrad(dy3, u, ["ffn.2.weight"]), bgrad(dy3, ["ffn.2.bias"])
        du_pre = bf(R, f)                                            # (dy3 W2) * gelu'(u_pre): GELU' in the GEMM's epilogue
        ops.gemm_raw(ptr(dy3), ptr(P["w2T"]), ptr(du_pre), R, f, d, d, d, f, EPI_GELU_BWD, aux=ptr(u_pre), ldaux=f)
        wgrad(du_pre, h2, ["ffn.0.weight"]), bgrad(du_pre, ["ffn.0.bias"])
        wg.launch()                                                  # FFN weight gradients: second stream, from here on
        dh2 = _dgrad(du_pre, P["w1T"], split_k=fc.split_k)
        # ---- cross-attention branch: x2 = x1 + y2  (its dy2 = bf16(dx) comes out of the same pass)
        dy2 = ln_bwd(S["x2"], dh2, 3, 4, nxt=(None, None))
        del du_pre, dh2
    else:
        # the reference's quirk ran this block's FFN without a graph (model.py:317-324): nothing flows back through y3, only
        # its gate gets a gradient (dx . y3) — in the same pass over dx that makes the cross-attention branch's dy2 = bf16(dx)
        dy2 = bf(R, d)
        ops.gated_residual_bwd_raw(ptr(dx), ptr(S["y3"]), ptr(dy2), ptr(d_eb, 5 * d), six, R, d, 1.0, None, None, 0, Sq)
    oc, qc, kc = S["oc"], S["qc"], S["kc"]
    wgrad(dy2, oc, ["cross_attn.o.weight"]), bgrad(dy2, ["cross_attn.o.bias"])
    doc = _dgrad(dy2, P["wo_cT"], epilogue=EPI_BF16)
    Rc = B * Lt
    ctx2 = fc.ctx.view(B * Lc, d) if not i2v else fc.ctx[:, n_img:].contiguous().view(Rc, d)
    vc = ops.transpose_bf16_batched(S["vtc"], Lt)                     # [B*Lt, d]
    dqc = bf(R, d)
    dkv = bf(Rc, 2 * d)                                               # dk | dv of the text keys, one buffer
    if not i2v:
        _attn_bwd(qc, kc, vc, oc, doc, S["lse_ca"], fc.ctx_lens32, B, N, Sq, Lt, D ** -0.5,
                  (dqc, dkv[:, :d], dkv[:, d:]), S["o32_ca"], False)
    else:                                                             # the image-token branch: same q, same dO
        oi, ki = S["oi"], S["ki"]
        wg.launch()                                                  # (the second product ADDS to the first one's result)
        wg.add(dy2, oi, out=g["cross_attn.o.weight"])
        dq32, dk32, dv32 = ops.flash_attn_bwd(qc, kc, vc, oc, doc, S["lse_ca"], fc.ctx_lens32, B, N, Sq, Lt, D ** -0.5)
        vi = ops.transpose_bf16_batched(S["vti"], n_img)
        dqi, dki, dvi = ops.flash_attn_bwd(qc, ki, vi, oi, doc, S["lse_ci"], None, B, N, Sq, n_img, D ** -0.5)
        ops.colsum_accum(dqi.view(1, R * d), dq32.view(R * d))        # dq = dq_text + dq_img
        ops.cast_bf16(dq32, out=dqc)
        ops.cast_bf16_strided(dk32, dkv[:, :d])
        ops.cast_bf16_strided(dv32, dkv[:, d:])
        Ri = B * n_img
        ctxi = fc.ctx[:, :n_img].contiguous().view(Ri, d)
        dkvi = bf(Ri, 2 * d)
        ops.cast_bf16_strided(dki, dkvi[:, :d])
        ops.cast_bf16_strided(dvi, dkvi[:, d:])
        rms_bwd(ptr(S["kfi"]), False, d, ptr(dkvi), 2 * d, Ri, [ca._norm_w("norm_k_img")], ca.qk_norm, False,
                ["cross_attn.norm_k_img.weight"], ca)
        wgrad(dkvi, ctxi, ["cross_attn.k_img.weight", "cross_attn.v_img.weight"])
        bgrad(dkvi, ["cross_attn.k_img.bias", "cross_attn.v_img.bias"])
        _dgrad_ctx(dkvi, P["wkv_iT"], st.d_ctx, 0, n_img)
        del dq32, dk32, dv32, dqi, dki, dvi
    rms_bwd(ptr(S["qcb"]), True, d, ptr(dqc), d, R, [ca._norm_w("norm_q")], ca.qk_norm, False, ["cross_attn.norm_q.weight"], ca)
    h3 = S["h3"]
    wgrad(dqc, h3, ["cross_attn.q.weight"]), bgrad(dqc, ["cross_attn.q.bias"])
    dh3 = _dgrad(dqc, P["wq_cT"])
    # The gradient of the text keys / values is off the critical path (nothing in this block reads it again): its norm
    # backward and its context-gradient GEMM go to the second stream, in front of the weight-gradient group that reads
    # dkv there (one stream: in order).  st.d_ctx is read by the embedding node, which joins first.
    # (The k | v bias gradients are column sums of dkv AFTER the norm backward has rewritten its k half in place: they are
    # launched right behind it on the same stream, not with the block's other column sums at the end of the main stream's
    # block — that flush does not wait for the second stream and would read the k half before or while it is rewritten.)
    def kv_path():
        rms_bwd(ptr(S["kf"]), False, d, ptr(dkv), 2 * d, Rc, [ca._norm_w("norm_k")], ca.qk_norm, False,
                ["cross_attn.norm_k.weight"], ca, defer=False)
        bgrad(dkv, ["cross_attn.k.bias", "cross_attn.v.bias"], now=True)
        _dgrad_ctx(dkv, P["wkv_cT"], st.d_ctx, n_img, Lt)
    if _SIDE_KV and not i2v:
        _on_side(dev, kv_path, (dkv, S["kf"], st.d_ctx, arena.buf, fc.ctx))
    else:
        kv_path()
    wgrad(dkv, ctx2, ["cross_attn.k.weight", "cross_attn.v.weight"])   # [2d, d]: k | v in one GEMM
    x1 = S["x1"]
    if blk.cross_attn_norm:
        # ---- self-attention branch: x1 = x0 + y1 * g2  (dy1 = bf16(dx * g2) and the gate's gradient: same pass)
        dw3, db3 = acc1("norm3.weight", d), acc1("norm3.bias", d)
        dy1 = bf(R, d)
        ops.layernorm_modulate_bwd2(x1, dh3, dx, R, d, blk.norm3.eps, 0.0, ptr(S["w3"]), None, 0, ptr(dw3), ptr(db3), 0, Sq,
                                    dy_next=dy1, y_next=S["y1"], gate_const=0.0, gate0=ptr(mod, 2 * d), gate1=ptr(e0, 2 * d),
                                    gate1_stride=six, dgate=ptr(d_eb, 2 * d), dgate_stride=six, defer=deferred)
        g["norm3.weight"], g["norm3.bias"] = dw3, db3
    else:
        ops.colsum_accum(dh3.view(1, R * d), dx.view(R * d))          # dx += dh3
        dy1 = resid_bwd(S["y1"], 2)
    del dy2, doc, dh3
    o, q, k, h1 = S["o"], S["q"], S["k"], S["h1"]
    wgrad(dy1, o, ["self_attn.o.weight"]), bgrad(dy1, ["self_attn.o.bias"])
    # the cross-attention's weight gradients + the self-attention's o: 3 x 144 full-K tiles + 288 short ones = about one
    # round of the 512 resident 128 x 128 tiles (with q|k|v's 432 tiles in the same launch it would be 2.2 rounds = 3)
    # OMH_WG_LATE=1 (A/B): not here but together with q|k|v's BEHIND the self-attention backward —
    # one group of 192 tiles of the k-major stream — so that the attention backward streams have the chip to themselves
    if not _WG_LATE:
        wg.launch()
    do = _dgrad(dy1, P["woT"], epilogue=EPI_BF16)
    v = ops.transpose_bf16_batched(S["vt"], Sq)                       # [B*S, d]
    dqkv = bf(R, 3 * d)                                               # dq | dk | dv, one buffer
    _attn_bwd(q, k, v, o, do, S["lse_sa"], fc.seq_lens32, B, N, Sq, Sq, D ** -0.5,
              (dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]), S["o32_sa"], True)
    qk = S["qk"]
    rms_bwd(ptr(qk), True, 2 * d, ptr(dqkv), 3 * d, R, [sa._norm_w("norm_q"), sa._norm_w("norm_k")], sa.qk_norm, True,
            ["self_attn.norm_q.weight", "self_attn.norm_k.weight"], sa, n_seg=2, seg_x=d, seg_dy=d)   # q and k: one launch
    wgrad(dqkv, h1, [f"self_attn.{nm}.weight" for nm in "qkv"])       # [3d, d]: q | k | v in one GEMM
    bgrad(dqkv, [f"self_attn.{nm}.bias" for nm in "qkv"])
    wg.launch()                                                      # the self-attention's weight gradients
    dh1 = _dgrad(dqkv, P["wqkvT"])                                      # K = 3d: dq Wq + dk Wk + dv Wv
    ln_bwd(x0, dh1, 0, 1)
    # ---- modulation / e0
    if deferred:                                                     # d_eb and the gains are complete behind this launch
        ops.partial_colsum_multi(deferred)
    dmod = acc1("modulation", six)
    ops.colsum_accum(d_eb.view(B, six), dmod)
    g["modulation"] = dmod
    ops.colsum_accum(d_eb.view(1, B * six), st.d_e0.view(B * six))      # d_e0 += d_eb
    wg.launch()
    if _SIDE_BIAS:                                                   # the bias gradients' column sums: second stream too
        _on_side(dev, arena.flush, [dy_ for dy_, _ in arena.pending] + [arena.buf])
    else:
        arena.flush()
    g["__dx__"] = dx
    return g


def _dgrad_ctx(dy, wT, d_ctx, first, L):
    """d_ctx[b, first:first+L, :] += dy[b*L:(b+1)*L, :] @ W  for every sample b (the context gradient of a K | V
    projection that reads a slice of the context rows): one batched GEMM into the strided destination."""
    B, Lc, d = d_ctx.shape
    Nn = dy.shape[1]
    ops.gemm_raw(ptr(dy), ptr(wT), ptr(d_ctx, first * d), L, d, Nn, dy.stride(0), wT.stride(0), d, EPI_ACC, batch=B,
                 strideA=L * dy.stride(0), strideB=0, strideC=Lc * d)


def activation_bytes(model, rows, batch=1):
    """Bytes a training forward over ``rows`` tokens (``batch`` clips) keeps for the backward when nothing is re-run."""
    blk = model.blocks[0]
    dim, ffn, heads = blk.dim, blk.ffn_dim, blk.self_attn.num_heads
    i2v = hasattr(blk.cross_attn, "k_img")
    # per token and block: fp32 [dim] x0, x1, x2, o32_sa, o32_ca (20 B) + bf16 [dim] h1, qk (2), q, k, vt, o, y1, h3, qcb,
    # qc, oc, y3, h2 (14 -> 28 B) (+ oi for i2v) + bf16 [ffn] u, u_pre + two (i2v: three) fp32 log-sum-exp values per head
    per_row = (48 + (2 if i2v else 0)) * dim + 4 * ffn + (12 if i2v else 8) * heads
    ctx_rows = max(1, int(batch)) * (model.text_len + (257 if i2v else 0))
    per_ctx_row = 8 * dim                                     # context K fp32 + normalised K bf16 + V^T bf16
    return int(len(model.blocks) * (rows * per_row + ctx_rows * per_ctx_row) * 1.15)


def pending_step_bytes(model):
    """Memory the rest of the step will still claim after the forward: AdamW's two moments (8 bytes per trainable
    parameter, until this package's optimizer has allocated them and said so: ``param._omh_moments_allocated``, ADVICE
    round 4 — counted again on every later step they made ``auto`` recompute although memory sufficed) and the gradients
    where none exist yet (first step, or ``zero_grad(set_to_none=True)``).  ADVICE round 3: without this the first
    step's free-memory reading is overstated by exactly what the backward and the optimizer step allocate."""
    pending = 0
    for p in model.parameters():
        if p.requires_grad:
            mref = getattr(p, "_omh_moments_allocated", None)                      # set by optim.AdamW.step
            have_moments = mref is not None and mref() is not None
            pending += p.numel() * ((0 if have_moments else 8) + (0 if p.grad is not None else 4))
    return pending


def keep_activations(model, rows, device, batch=1):
    """What ``model.use_checkpoint`` amounts to for a forward over ``rows`` tokens (module docstring): True = keep
    every block's activations for the backward, False = keep the block inputs and recompute."""
    if not getattr(model, "use_checkpoint", True):
        return True
    policy = os.environ.get("OMH_CHECKPOINT_POLICY") or getattr(model, "checkpoint_policy", "auto")
    if policy != "auto":
        return False
    if torch.cuda.is_current_stream_capturing():              # no memory queries under hipGraph capture: what the
        return bool(model.__dict__.get("_kept_activations", False))       # warm-up step before the capture decided
    need = activation_bytes(model, rows, batch)
    free, _ = torch.cuda.mem_get_info(device)
    free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)      # the allocator's own free blocks
    return need < 0.5 * (free - pending_step_bytes(model))


# ----------------------------------------------------------------------------- block node
class _BlockFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, model, st, idx, *params):
        blk = model.blocks[idx]
        keep = st.keep
        with torch.no_grad():
            x0 = x.detach()
            if not (x0.dtype == torch.float32 and x0.is_contiguous()):
                x0 = x0.float().contiguous()
            out, S = _block_forward(model, blk, idx, st, x0, st.packs[idx], keep, need=keep)
        ctx.model, ctx.st, ctx.idx = model, st, idx
        ctx.kept = S if keep else None
        ctx.save_for_backward(x0)
        return out

    @staticmethod
    def backward(ctx, dx_out):
        (x0,) = ctx.saved_tensors
        model, st, idx = ctx.model, ctx.st, ctx.idx
        blk = model.blocks[idx]
        P = st.packs[idx]
        dev = x0.device
        tgt = _grad_targets(ctx, model, idx)                 # gradient accumulation: existing .grad tensors to add into
        slots = _grad_slots(ctx, model, idx, tgt)            # fresh weight gradients: straight into the reducer's buckets
        with torch.no_grad():                                # (asked before no_grad: it looks at the pass's grad mode)
            S = ctx.kept
            ctx.kept = None
            if S is None:                                    # use_checkpoint: re-run the forward's kernels on its input
                _, S = _block_forward(model, blk, idx, st, x0, P, False)
            # The block backward updates the incoming gradient IN PLACE.  That is only legal when the buffer belongs to
            # this node: autograd hands a node the producer's own tensor when the block output has ONE consumer (the next
            # block / the head allocate a fresh dx), but a caller that also taps block outputs (the reference's
            # discriminator hooks, seaweed_apt/model.py:150-155) makes autograd accumulate, and then the same buffer
            # can be what another consumer's backward still holds.  Own-or-copy: in place only for a tensor this
            # package's own nodes produced for us (tagged below), a private copy otherwise.
            own = dx_out.dtype == torch.float32 and dx_out.is_contiguous() and \
                getattr(dx_out, "_omh_exclusive", False) and not _ALWAYS_COPY_DX
            dx = dx_out if own else dx_out.float().contiguous().clone() if (
                dx_out.dtype == torch.float32 and dx_out.is_contiguous()) else dx_out.float().contiguous()
            # the join of the weight-gradient stream: deferred to the end of this backward pass when legal — decided by the
            # first block of this forward to run in a pass (a second forward of the same model in the pass then finds
            # gradients in place and joins per block; so does a second pass over the same graph)
            key = (dev, id(model))
            tok = _join_pending.get(key)
            if tok is not None and st.__dict__.get("defer_tok") is tok:
                defer = True
            elif tok is not None:
                # another forward of this model in the same pass: the engine sums the two contributions to a parameter's
                # gradient on the main stream as soon as this node returns — behind this block's join, which also covers
                # what the other forward's blocks left on the second stream
                defer = False
            elif st.__dict__.get("no_defer", False):
                defer = False
            elif any(o is not st and not o.__dict__.get("bwd_done", False)
                     for o in model.__dict__.get("_omh_live_states", ())):
                # another forward of this model still awaits its backward: whichever order the engine picks, the sum of
                # the two contributions to a shared parameter is formed on the main stream — join per block
                st.no_defer, defer = True, False
            elif _may_defer_join(model):
                if tok is None:
                    tok = _join_pending[key] = object()

                    def _end_of_pass(dev=dev, key=key):
                        _join_pending.pop(key, None)
                        _wgrad_join(dev)
                    torch.autograd.Variable._execution_engine.queue_callback(_end_of_pass)
                st.defer_tok, defer = tok, True
            else:
                st.no_defer, defer = True, False
            done = False
            try:
                grads = _block_backward(model, blk, idx, st, S, dx, P, tgt, slots)
                done = True
            finally:
                # (tgt None: gradients exist and autograd will add this block's to them on the main stream)
                if not (done and defer and tgt is not None):
                    _wgrad_join(dev)                         # also on an exception: nothing may stay on the side stream
                if not done:
                    _join_pending.pop(key, None)             # (the engine drops its callbacks with the failed pass)
                if idx == 0:
                    st.bwd_done = True                       # the last block node of this forward's backward
        out = []
        for n, p in _block_params(model, idx):
            gg = grads.get(n) if p.requires_grad else None
            if gg is not None and slots and n in slots:      # written into the reducer's bucket: that slice IS the gradient
                p.grad = slots[n]
            if gg is not None and ((tgt and n in tgt) or (slots and n in slots)):
                gg = None                                    # in place: nothing for autograd to do, and its AccumulateGrad
                # node — the hooks on it — does not run: this package's reducer (only its hooks get this far) is told here
                for h in list((getattr(p, "_post_accumulate_grad_hooks", None) or {}).values()):
                    h(p)
            out.append(None if gg is None else gg.view(p.shape).to(p.dtype))
        gdx = grads["__dx__"]
        gdx._omh_exclusive = True                             # fresh from this node: the previous block may update it in place
        return (gdx, None, None, None, *out)


# ----------------------------------------------------------------------------- head node
class _HeadFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, model, st, grids, *params):
        with torch.no_grad():
            out = model.head(x.detach(), st.e)
            outs = model.unpatchify(out, None, _grids=grids)
        ctx.save_for_backward(x.detach())
        ctx.model, ctx.st, ctx.grids = model, st, grids
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        (x,) = ctx.saved_tensors
        model, st, grids = ctx.model, ctx.st, ctx.grids
        head = model.head
        B, S, d = x.shape
        R = B * S
        dev = x.device
        with torch.no_grad():
            mod = head.modulation.detach().float().contiguous()
            e = st.e.contiguous()
            ncol = head.head.weight.shape[0]
            dtok = torch.zeros(R, ncol, dtype=torch.bfloat16, device=dev)
            for b, (gb, grd) in enumerate(zip(gouts, grids)):
                if gb is None:
                    continue
                n = grd[0] * grd[1] * grd[2]
                dtok[b * S:b * S + n].copy_(ops.unpatchify_bwd(gb.contiguous().float(), grd, model.patch_size))
            hh = torch.empty(R, d, dtype=torch.bfloat16, device=dev)
            ops.layernorm_modulate_raw(ptr(x), ptr(hh), R, d, head.eps, 1.0, ptr(mod, d), ptr(e), d, ptr(mod, 0),
                                       ptr(e), d, S)
            w, _ = head._packed.get("head", (head.head.weight, head.head.bias), lambda: (
                ops.cast_bf16(head.head.weight.detach().float().contiguous()),
                head.head.bias.detach().float().contiguous()))
            gw, gb_ = _wgrad(dtok, hh), _bgrad(dtok)
            dhh = _dgrad(dtok, _wT(head, "head", w))
            dx = torch.zeros(R, d, dtype=torch.float32, device=dev)
            d_hb = torch.zeros(B, 2, d, dtype=torch.float32, device=dev)
            ops.layernorm_modulate_bwd2(x, dhh, dx, R, d, head.eps, 1.0, ptr(mod, d), ptr(e), d,
                                        ptr(d_hb, d), ptr(d_hb, 0), 2 * d, S)
            dmod = torch.zeros(2 * d, dtype=torch.float32, device=dev)
            ops.colsum_accum(d_hb.view(B, 2 * d), dmod)
            for b in range(B):                       # e enters both the scale and the shift (model.py:357-358)
                ops.colsum_accum(d_hb[b], st.d_e[b])
        grads = {"modulation": dmod.view(1, 2, d), "head.weight": gw, "head.bias": gb_}
        out = []
        for n, p in head.named_parameters():
            gg = grads.get(n) if p.requires_grad else None
            out.append(None if gg is None else gg.view(p.shape).to(p.dtype))
        gdx = dx.view(B, S, d)
        gdx._omh_exclusive = True                             # (see _BlockFn.backward: own-or-copy)
        return (gdx, None, None, None, *out)


# ----------------------------------------------------------------------------- embed node
_EMBED_MODULES = ("patch_embedding", "text_embedding", "time_embedding", "time_projection")


def _embed_params(model):
    out = []
    for m in _EMBED_MODULES + (("img_emb",) if hasattr(model, "img_emb") else ()):
        out += [(f"{m}.{n}", p) for n, p in getattr(model, m).named_parameters()]
    return out


def _img_emb_backward(model, clip_fea, d_img, g):
    """Backward of MLPProj (model.py:362-374: LayerNorm, Linear, GELU(erf), Linear, LayerNorm) on the CLIP tokens:
    recompute keeping the pre-activations, then the chain rule.  d_img fp32 [B*257, dim] = gradient of its output."""
    proj = model.img_emb.proj
    ln0, l1, l3, ln4 = proj[0], proj[1], proj[3], proj[4]
    dev = d_img.device
    x = clip_fea.to(device=dev, dtype=torch.float32).contiguous().view(-1, clip_fea.shape[-1])
    rows, cin = x.shape
    dim = l3.weight.shape[0]
    f = lambda t_: t_.detach().float().contiguous()
    w0, b0, w4, b4 = f(ln0.weight), f(ln0.bias), f(ln4.weight), f(ln4.bias)
    h0 = ops.layernorm_modulate(x, ln0.eps, 0.0, mul0=w0, add0=b0)
    w1, w3 = ops.cast_bf16(f(l1.weight)), ops.cast_bf16(f(l3.weight))
    z1 = ops.gemm(h0, w1, bias=f(l1.bias), epilogue=EPI_BF16)
    g1 = ops.gelu_erf(z1)
    z3 = ops.gemm(g1, w3, bias=f(l3.bias), epilogue=EPI_F32)
    # LayerNorm 4 (affine): dz3, dw4, db4
    dz3 = torch.zeros(rows, dim, dtype=torch.float32, device=dev)
    dw4, db4 = torch.zeros(dim, dtype=torch.float32, device=dev), torch.zeros(dim, dtype=torch.float32, device=dev)
    ops.layernorm_modulate_bwd_raw(ptr(z3), ptr(d_img), ptr(dz3), rows, dim, ln4.eps, 0.0, ptr(w4), None, 0, ptr(dw4),
                                   ptr(db4), 0, rows)
    dz3b = ops.cast_bf16(dz3)
    g["img_emb.proj.4.weight"], g["img_emb.proj.4.bias"] = dw4, db4
    g["img_emb.proj.3.weight"], g["img_emb.proj.3.bias"] = _wgrad(dz3b, g1), _bgrad(dz3b)
    dg1 = ops.gemm(dz3b, _wT_once(w3), epilogue=EPI_BF16)
    dz1 = ops.gelu_erf_bwd(dg1, z1)
    g["img_emb.proj.1.weight"], g["img_emb.proj.1.bias"] = _wgrad(dz1, h0), _bgrad(dz1)
    dh0 = _dgrad(dz1, _wT_once(w1))
    dx = torch.zeros(rows, cin, dtype=torch.float32, device=dev)      # (gradient w.r.t. the CLIP tokens: not needed)
    dw0, db0 = torch.zeros(cin, dtype=torch.float32, device=dev), torch.zeros(cin, dtype=torch.float32, device=dev)
    ops.layernorm_modulate_bwd_raw(ptr(x), ptr(dh0), ptr(dx), rows, cin, ln0.eps, 0.0, ptr(w0), None, 0, ptr(dw0),
                                   ptr(db0), 0, rows)
    g["img_emb.proj.0.weight"], g["img_emb.proj.0.bias"] = dw0, db0


class _EmbedFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, model, st, x_list, t, context, seq_len, clip_fea, y, extra_tokens, *params):
        with torch.no_grad():
            xs, e, fc, grids, lens, ctx_lens = model._embed(x_list, t, context, seq_len, clip_fea, y, extra_tokens)
        ctx.n_extra = 0 if extra_tokens is None else int(extra_tokens.shape[1])
        ctx.extra_dtype = None if extra_tokens is None else extra_tokens.dtype
        st.fc, st.e, st.grids, st.lens = fc, e, grids, lens
        B, d = fc.B, fc.dim
        dev = xs.device
        st.d_e0 = torch.zeros(B, 6, d, dtype=torch.float32, device=dev)
        st.d_e = torch.zeros(B, d, dtype=torch.float32, device=dev)
        st.d_ctx = torch.zeros(B, fc.Lc, d, dtype=torch.float32, device=dev)
        ctx.model, ctx.st = model, st
        ctx.inputs = (x_list, t, context, seq_len, y, clip_fea)
        return xs

    @staticmethod
    def backward(ctx, dxs):
        model, st = ctx.model, ctx.st
        x_list, t, context, seq_len, y, clip_fea = ctx.inputs
        fc = st.fc
        B, d = fc.B, fc.dim
        dev = dxs.device
        g = {}
        with torch.no_grad():
            _wgrad_join(dev)            # st.d_ctx, st.d_e0: the blocks' second-stream work adds into them
            dxs = dxs.contiguous().float()
            # ---- patch embedding: x[b,:n] = tok_b Wpe^T + bpe
            pt, ph, pw = model.patch_size
            kin = model.in_dim * pt * ph * pw
            Kp = _ru(kin, 8)
            if y is not None:
                x_list = [torch.cat([u, v], dim=0) for u, v in zip(x_list, y)]
            dW = torch.zeros(d, Kp, dtype=torch.float32, device=dev)
            db = torch.zeros(d, dtype=torch.float32, device=dev)
            for b, u in enumerate(x_list):
                n = st.lens[b]
                tok = ops.patchify(u.to(device=dev, dtype=torch.float32).contiguous(), model.patch_size, Kp)
                dxb = ops.cast_bf16(dxs[b, :n].contiguous())
                ops.gemm_tn(dxb, tok, out=dW, accumulate=True)
                ops.colsum_accum(dxs[b, :n], db)
            g["patch_embedding.weight"] = dW[:, :kin].contiguous()
            g["patch_embedding.bias"] = db
            # ---- time embedding (fp32): z0 = W0 sin + b0 ; e = W2 silu(z0) + b2 ; e0 = Wp silu(e) + bp
            te0, te2, tp1 = model.time_embedding[0], model.time_embedding[2], model.time_projection[1]
            W0, W2, Wp = (m.weight.detach().float().contiguous() for m in (te0, te2, tp1))
            sin = ops.sinusoidal_embedding(t.to(dev), model.freq_dim)
            z0 = ops.dense_f32(sin, W0, te0.bias.detach().float(), 0, 0)
            e = st.e
            dWp, dbp = torch.zeros_like(Wp), torch.zeros(Wp.shape[0], dtype=torch.float32, device=dev)
            ops.dense_f32_bwd(e, Wp, st.d_e0.view(B, 6 * d), dW=dWp, db=dbp, dx=st.d_e, dx_accumulate=True, act_in=1)
            dW2, db2 = torch.zeros_like(W2), torch.zeros(d, dtype=torch.float32, device=dev)
            dz0 = torch.empty(B, d, dtype=torch.float32, device=dev)
            ops.dense_f32_bwd(z0, W2, st.d_e, dW=dW2, db=db2, dx=dz0, act_in=1)
            dW0, db0 = torch.zeros_like(W0), torch.zeros(d, dtype=torch.float32, device=dev)
            ops.dense_f32_bwd(sin, W0, dz0, dW=dW0, db=db0, dx=None, act_in=0)
            g.update({"time_projection.1.weight": dWp, "time_projection.1.bias": dbp, "time_embedding.2.weight": dW2,
                      "time_embedding.2.bias": db2, "time_embedding.0.weight": dW0, "time_embedding.0.bias": db0})
            # ---- text embedding: ctx = W2t gelu(W0t cin + b0t) + b2t
            ctx_in = torch.zeros(B, model.text_len, model.text_dim, dtype=torch.float32, device=dev)
            for b, u in enumerate(context):
                ctx_in[b, :u.shape[0]] = u.to(device=dev, dtype=torch.float32)
            t0, t2 = model.text_embedding[0], model.text_embedding[2]
            w0 = ops.cast_bf16(t0.weight.detach().float().contiguous())
            w2 = ops.cast_bf16(t2.weight.detach().float().contiguous())
            cin = ops.cast_bf16(ctx_in).view(B * model.text_len, model.text_dim)
            pre = ops.gemm(cin, w0, bias=t0.bias.detach().float(), epilogue=EPI_BF16)
            gl = ops.gelu_tanh(pre)
            n_img = fc.Lc - model.text_len
            dctx = st.d_ctx[:, n_img:].contiguous().view(B * model.text_len, d)
            dctx_b = ops.cast_bf16(dctx)
            g["text_embedding.2.weight"], g["text_embedding.2.bias"] = _wgrad(dctx_b, gl), _bgrad(dctx_b)
            dgl = ops.gemm(dctx_b, _wT_once(w2), epilogue=EPI_BF16)
            dpre = ops.gelu_tanh_bwd(dgl, pre)
            g["text_embedding.0.weight"], g["text_embedding.0.bias"] = _wgrad(dpre, cin), _bgrad(dpre)
            # ---- image embedding (i2v): the first n_img context rows came from img_emb(clip_fea)
            if n_img and clip_fea is not None:
                _img_emb_backward(model, clip_fea, st.d_ctx[:, :n_img].contiguous().view(B * n_img, d), g)
        out = []
        for n, p in _embed_params(model):
            gg = g.get(n) if p.requires_grad else None
            out.append(None if gg is None else gg.view(p.shape).to(p.dtype))
        # condition tokens of the OmniHuman adapters (omnihuman_wan_t2v.py:453-488): they sit in front of the text
        # in the context, so their gradient is the leading rows of the context gradient
        d_extra = None
        if ctx.n_extra and ctx.needs_input_grad[8]:
            d_extra = st.d_ctx[:, :ctx.n_extra].contiguous().to(ctx.extra_dtype)
        return (None, None, None, None, None, None, None, None, d_extra, *out)


def forward_train(model, x, t, context, seq_len, clip_fea=None, y=None, extra_conditions=None):
    """WanModel.forward with autograd enabled: same outputs as the inference path, attached to a
    graph of hand-written nodes (see module docstring).  ``extra_conditions``: [B, Ne, dim] condition tokens (or a
    dict holding them under 'tokens') that receive a gradient like any other input."""
    st = _State()
    for key in [k for k in _join_pending if k[1] == id(model)]:   # a backward pass that died before its end-of-pass join
        _join_pending.pop(key, None)
        _wgrad_join(key[0])
    st.packs = TrainPacks.of(model).refresh(model)           # bf16 weight copies: one launch when anything changed
    # forwards of this model whose backward has not run yet: with more than one of them in a pass the
    # engine sums their contributions to a shared parameter on the main stream, so none of them may defer its join
    model.__dict__.setdefault("_omh_live_states", weakref.WeakSet()).add(st)
    x_list = list(x) if not isinstance(x, (list, tuple)) else list(x)
    eparams = [p for _, p in _embed_params(model)]
    tok = extra_conditions.get("tokens") if isinstance(extra_conditions, dict) else extra_conditions
    xs = _EmbedFn.apply(model, st, x_list, t, list(context), seq_len, clip_fea, y, tok, *eparams)
    if not xs.requires_grad:
        xs.requires_grad_(True)          # keeps the chain alive when the embed parameters are frozen
    st.keep = keep_activations(model, xs.shape[0] * xs.shape[1], xs.device, batch=xs.shape[0])
    model.__dict__["_kept_activations"] = st.keep            # what the last training forward did (bench / tests)
    for i, blk in enumerate(model.blocks):
        if blk._forward_pre_hooks:
            # the block nodes take their other inputs from the forward state, so a pre-hook could only rewrite x — and a
            # rewritten x is something the reference never does: refuse loudly rather than skip it silently
            raise NotImplementedError("forward pre-hooks on WanAttentionBlock are not honoured by the training forward; "
                                      "register a forward hook (block outputs) instead")
        x_in = xs
        xs = _BlockFn.apply(xs, model, st, i, *[p for _, p in _block_params(model, i)])
        # forward hooks registered on a block (the reference's discriminator taps block outputs this way,
        # seaweed_apt/model.py:150-155) see the block's output as under nn.Module.__call__; the extra consumer they
        # create is why _BlockFn.backward owns-or-copies its incoming gradient
        for hid, hook in list(blk._forward_hooks.items()):
            if hid in getattr(blk, "_forward_hooks_with_kwargs", ()):      # register_forward_hook(..., with_kwargs=True)
                r = hook(blk, (x_in,), {}, xs)
            else:
                r = hook(blk, (x_in,), xs)
            if r is not None:
                xs = r
    outs = _HeadFn.apply(xs, model, st, st.grids, *list(model.head.parameters()))
    return list(outs)
