"""HuggingFace-style boundary classes backed by the native gfx950 encoder.

What the reference calls today and what replaces it here:

  AutoModel / BertModel.from_pretrained(path)(input_ids=, attention_mask=)      -> CocoBertModel
      ANCE/model/models.py:225-229 ``self.bert(...)[0][:, 0]``; README.md:101-116
  BertForSequenceClassification subclass BertDot_NLL_LN (triplet NLL)           -> BertDotNLL
      ANCE/model/models.py:194-262
  CoCondenserForPretraining (encoder -> [CLS] -> all_gather -> contrastive)     -> CoCondenserForPretraining
      COCO/modeling.py:162-248  (Condenser head + MLM losses are SURVEY 8(f1) "next")

Design (MI355X first, not a port of the torch module tree):
  * all parameters live in TWO flat fp32 nn.Parameters (`flat_decay`: embeddings + weight matrices,
    `flat_nodecay`: biases + LayerNorm), laid out so consecutive layers sit at a uniform stride.
    The optimizer therefore updates two tensors (one fused AdamW kernel each), DDP all-reduces two
    large buckets over xGMI, and the native backward writes all layers' weight gradients with one
    grouped launch per matrix.  ``state_dict()`` / ``load_state_dict()`` still speak the HF BERT key
    names, so reference checkpoints load and ``save_pretrained`` round-trips.
  * one autograd.Function wraps the whole encoder: forward = one C call, backward = one C call.
  * a bf16 shadow of the weight matrices is refreshed by one cast kernel whenever the fp32 master
    changes (tracked through the tensor version counter).
Dropout (hf hidden_dropout_prob / attention_probs_dropout_prob) is active in train() mode as in the reference
(ANCE/drivers/run_ann.py:293; COCO keeps the backbone in eval, COCO/modeling.py:198, so only its Condenser head drops):
counter-based masks fused into the kernels that produce the dropped tensors, regenerated in the backward
(include/cocodr.h "Dropout").  torch's Philox stream cannot be matched bit-wise (SURVEY 7 iv); the distribution is.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import _native as N
from . import ops
from ._native import check, lib, ptr, stream_ptr
from .flatparams import FlatParamsMixin

__all__ = ["CocoBertConfig", "CocoBertModel", "BertDotNLL", "CoCondenserForPretraining", "EncoderOutput"]


# =============================================================================== config
class CocoBertConfig:
    """The BertConfig fields the hot path reads (config.json compatible)."""
    model_type = "bert"

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12,
                 pad_token_id=0, **extra):
        self.vocab_size = int(vocab_size)
        self.hidden_size = int(hidden_size)
        self.num_hidden_layers = int(num_hidden_layers)
        self.num_attention_heads = int(num_attention_heads)
        self.intermediate_size = int(intermediate_size)
        self.hidden_act = hidden_act
        self.hidden_dropout_prob = float(hidden_dropout_prob)
        self.attention_probs_dropout_prob = float(attention_probs_dropout_prob)
        self.max_position_embeddings = int(max_position_embeddings)
        self.type_vocab_size = int(type_vocab_size)
        self.initializer_range = float(initializer_range)
        self.layer_norm_eps = float(layer_norm_eps)
        self.pad_token_id = pad_token_id
        self.extra = dict(extra)
        if self.hidden_act != "gelu":
            raise ValueError(f"hidden_act={self.hidden_act!r}: only the exact-erf 'gelu' of BERT is implemented")
        if self.hidden_size != 64 * self.num_attention_heads:
            raise ValueError("head_dim must be 64 (hidden_size == 64 * num_attention_heads)")
        if self.hidden_size % 128 or self.intermediate_size % 128 or self.hidden_size > 1024:
            raise ValueError("hidden_size / intermediate_size must be multiples of 128 and hidden_size <= 1024")

    @classmethod
    def base(cls, **kw):
        return cls(**kw)

    @classmethod
    def large(cls, **kw):
        return cls(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, **kw)

    def to_dict(self) -> dict:
        d = dict(self.extra)
        d.update(model_type="bert", vocab_size=self.vocab_size, hidden_size=self.hidden_size,
                 num_hidden_layers=self.num_hidden_layers, num_attention_heads=self.num_attention_heads,
                 intermediate_size=self.intermediate_size, hidden_act=self.hidden_act,
                 hidden_dropout_prob=self.hidden_dropout_prob,
                 attention_probs_dropout_prob=self.attention_probs_dropout_prob,
                 max_position_embeddings=self.max_position_embeddings, type_vocab_size=self.type_vocab_size,
                 initializer_range=self.initializer_range, layer_norm_eps=self.layer_norm_eps,
                 pad_token_id=self.pad_token_id)
        return d

    @classmethod
    def from_pretrained(cls, path: str, **kw):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d.update(kw)
        d.pop("model_type", None)
        return cls(**d)

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @classmethod
    def coerce(cls, config) -> "CocoBertConfig":
        """A transformers ``BertConfig`` (what the reference passes as ``config=`` to ``from_pretrained``:
        COCO/modeling.py:100-101, ANCE/drivers/run_ann.py:889-901) or a plain dict -> this class, with its validation."""
        if isinstance(config, cls):
            return config
        d = dict(config) if isinstance(config, dict) else dict(config.to_dict())
        d.pop("model_type", None)
        return cls(**d)


# =============================================================================== flat parameter layout
class _Layout:
    """Offsets (in elements) of every HF-named tensor inside two flat parameters (decay / no-decay): a preamble of
    free-form tensors followed by ``n_layers`` BertLayer blocks at a uniform stride."""

    def __init__(self, cfg: CocoBertConfig, n_layers: Optional[int] = None, layer_prefix: str = "encoder.layer.",
                 decay_pre=None, nodecay_pre=None):
        H, I = cfg.hidden_size, cfg.intermediate_size
        NL = cfg.num_hidden_layers if n_layers is None else n_layers
        V, P, T = cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size
        if decay_pre is None:
            decay_pre = [("embeddings.word_embeddings.weight", (V, H)), ("embeddings.position_embeddings.weight", (P, H)),
                         ("embeddings.token_type_embeddings.weight", (T, H))]
        if nodecay_pre is None:
            nodecay_pre = [("embeddings.LayerNorm.weight", (H,)), ("embeddings.LayerNorm.bias", (H,))]
        self.cfg = cfg
        self.n_layers = NL
        d = OrderedDict()  # name -> (flat index 0/1, offset, shape)

        def numel(shape):
            n = 1
            for x in shape:
                n *= x
            return n

        o = 0
        for name, shape in decay_pre:
            d[name] = (0, o, tuple(shape))
            o += numel(shape)
            if not name.startswith("embeddings."):
                o = (o + 63) // 64 * 64  # the embedding tables stay contiguous (native side derives pos/type from word)
        self.emb_decay_end = o
        o = (o + 63) // 64 * 64
        self.mat_begin = o
        self.mat_stride = 3 * H * H + H * H + I * H + H * I
        self.off_wqkv, self.off_wo, self.off_w1, self.off_w2 = 0, 3 * H * H, 4 * H * H, 4 * H * H + I * H
        for l in range(NL):
            b = self.mat_begin + l * self.mat_stride
            p = f"{layer_prefix}{l}."
            d[p + "attention.self.query.weight"] = (0, b, (H, H))
            d[p + "attention.self.key.weight"] = (0, b + H * H, (H, H))
            d[p + "attention.self.value.weight"] = (0, b + 2 * H * H, (H, H))
            d[p + "attention.output.dense.weight"] = (0, b + self.off_wo, (H, H))
            d[p + "intermediate.dense.weight"] = (0, b + self.off_w1, (I, H))
            d[p + "output.dense.weight"] = (0, b + self.off_w2, (H, I))
        self.decay_numel = self.mat_begin + NL * self.mat_stride
        # --- no-decay flat: LayerNorm + biases
        o = 0
        for name, shape in nodecay_pre:
            d[name] = (1, o, tuple(shape))
            o += numel(shape)
            if not name.startswith("embeddings."):
                o = (o + 63) // 64 * 64
        self.vec_begin = (o + 3) // 4 * 4
        self.vec_stride = 3 * H + H + I + H + 4 * H
        self.off_bqkv, self.off_bo, self.off_b1, self.off_b2 = 0, 3 * H, 4 * H, 4 * H + I
        self.off_ln1g, self.off_ln1b, self.off_ln2g, self.off_ln2b = (5 * H + I, 6 * H + I, 7 * H + I, 8 * H + I)
        for l in range(NL):
            b = self.vec_begin + l * self.vec_stride
            p = f"{layer_prefix}{l}."
            d[p + "attention.self.query.bias"] = (1, b, (H,))
            d[p + "attention.self.key.bias"] = (1, b + H, (H,))
            d[p + "attention.self.value.bias"] = (1, b + 2 * H, (H,))
            d[p + "attention.output.dense.bias"] = (1, b + self.off_bo, (H,))
            d[p + "intermediate.dense.bias"] = (1, b + self.off_b1, (I,))
            d[p + "output.dense.bias"] = (1, b + self.off_b2, (H,))
            d[p + "attention.output.LayerNorm.weight"] = (1, b + self.off_ln1g, (H,))
            d[p + "attention.output.LayerNorm.bias"] = (1, b + self.off_ln1b, (H,))
            d[p + "output.LayerNorm.weight"] = (1, b + self.off_ln2g, (H,))
            d[p + "output.LayerNorm.bias"] = (1, b + self.off_ln2b, (H,))
        self.nodecay_numel = self.vec_begin + NL * self.vec_stride
        self.names = d

    def view(self, flats, name: str) -> torch.Tensor:
        which, off, shape = self.names[name]
        n = 1
        for s in shape:
            n *= s
        return flats[which][off:off + n].view(shape)

    def layer_structs(self, shadow_ptr: int, shadow_begin: int, nodecay_ptr: int, grads=None):
        """ctypes arrays of cocodr_layer_params (bf16 shadow matrices + fp32 vectors) and, with ``grads`` = (decay
        grad ptr, nodecay grad ptr), of cocodr_layer_grads for the layer blocks of this layout."""
        arr = (N.LayerParams * self.n_layers)()
        for l in range(self.n_layers):
            mb = shadow_ptr + 2 * (self.mat_begin - shadow_begin + l * self.mat_stride)
            vb = nodecay_ptr + 4 * (self.vec_begin + l * self.vec_stride)
            arr[l] = N.LayerParams(mb + 2 * self.off_wqkv, mb + 2 * self.off_wo, mb + 2 * self.off_w1, mb + 2 * self.off_w2,
                                   vb + 4 * self.off_bqkv, vb + 4 * self.off_bo, vb + 4 * self.off_b1, vb + 4 * self.off_b2,
                                   vb + 4 * self.off_ln1g, vb + 4 * self.off_ln1b, vb + 4 * self.off_ln2g, vb + 4 * self.off_ln2b)
        if grads is None:
            return arr, None
        gd, gn = grads
        garr = (N.LayerGrads * self.n_layers)()
        for l in range(self.n_layers):
            mb = gd + 4 * (self.mat_begin + l * self.mat_stride)
            vb = gn + 4 * (self.vec_begin + l * self.vec_stride)
            garr[l] = N.LayerGrads(mb + 4 * self.off_wqkv, mb + 4 * self.off_wo, mb + 4 * self.off_w1, mb + 4 * self.off_w2,
                                   vb + 4 * self.off_bqkv, vb + 4 * self.off_bo, vb + 4 * self.off_b1, vb + 4 * self.off_b2,
                                   vb + 4 * self.off_ln1g, vb + 4 * self.off_ln1b, vb + 4 * self.off_ln2g, vb + 4 * self.off_ln2b)
        return arr, garr


class EncoderOutput:
    """Minimal stand-in for transformers' BaseModelOutputWithPooling: attribute access, ``out[0]`` and
    tuple-unpacking (ANCE/model/models.py:228 indexes ``outputs1[0]``)."""

    def __init__(self, last_hidden_state, hidden_states=None, cls_fp32=None):
        # ``last_hidden_state`` may be a zero-argument callable (packed batches: the padded [B, L, H] view of the last layer is
        # only materialised for callers that read it - every reference wrapper consumes the [CLS] rows alone)
        self._last = last_hidden_state
        self.pooler_output = None
        self.hidden_states = hidden_states
        self.cls_fp32 = cls_fp32

    @property
    def last_hidden_state(self):
        if callable(self._last):
            self._last = self._last()
        return self._last

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output, self.hidden_states)[i]

    def __iter__(self):
        return iter((self.last_hidden_state, self.pooler_output))


# =============================================================================== the encoder Function
class _EncoderFn(torch.autograd.Function):
    """(flat_decay, flat_nodecay, ids, mask) -> (last_hidden bf16 [B,L,H], cls fp32 [B,H][, hidden_states[0 .. N-1]]).
    forward = cocodr_encoder_fwd, backward = cocodr_encoder_bwd.  With ``taps`` the intermediate hidden states are outputs
    of the Function too, so a head that reads ``hidden_states[i]`` (the reference's Condenser head reads
    ``hidden_states[skip_from]``, COCO/modeling.py:212-216) back-propagates into the backbone: the backward then walks the
    layer stack in ranges and adds each such gradient where its layer range ends."""

    @staticmethod
    def forward(ctx, flat_decay, flat_nodecay, ids, mask, model, grad_mode=True, taps=False, cls_only=False):
        # grad_mode: torch.is_grad_enabled() at the call site (always False in here); inference keeps no activations and never drops
        training = bool(grad_mode and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        arena, lay = model._run_forward(ids, mask, training, cls_tail=cls_only and not taps)
        ctx.tail = arena._cocodr_tail
        B, L = ids.shape
        H, NL = model.config.hidden_size, model.config.num_hidden_layers
        M = B * L
        hidden = arena[lay.hidden: lay.hidden + (NL + 1) * M * H * 2].view(torch.bfloat16).view(NL + 1, B, L, H)
        cls = arena[lay.cls_f32: lay.cls_f32 + B * H * 4].view(torch.float32).view(B, H)
        ctx.model = model
        ctx.training = training
        ctx.arena = arena if training else None
        ctx.lay = lay
        ctx.ids, ctx.mask = ids, mask
        ctx.set_materialize_grads(False)
        if ctx.tail:  # the last layer ran on its [CLS] rows only: there is no last_hidden_state (and no hidden_states tuple)
            return None, cls.clone(), None
        # the third output is the [N + 1, B, L, H] stack of hidden states itself (not differentiable: a caller that wants gradients
        # through an intermediate state asks for taps) - handed to THIS call's caller, not parked on the module, so that
        # overlapping forwards of one module (side streams, BertDotNLL with merge_passes = False) cannot read each other's
        stack = hidden.detach()
        ctx.mark_non_differentiable(stack)
        last = hidden[NL]
        if taps and training:  # views of the saved activations (read-only for the caller, like any saved tensor)
            return (last, cls.clone(), stack) + tuple(hidden[l] for l in range(NL))
        return last, cls.clone(), stack

    @staticmethod
    def backward(ctx, d_last, d_cls, _d_stack, *d_taps):
        model = ctx.model
        if not ctx.training or ctx.arena is None:
            raise RuntimeError("encoder backward called but the forward ran without saved activations")
        B, L = ctx.ids.shape
        H = model.config.hidden_size
        taps = {l: d for l, d in enumerate(d_taps) if d is not None}
        none = (None,) * 8
        if ctx.tail:
            if d_cls is None:
                return none
            gd, gn = model._run_backward(ctx.ids, ctx.mask, d_cls.to(torch.bfloat16).contiguous(), ctx.arena)  # [B,H]: the [CLS] rows' gradient
            ctx.arena = None
            return (gd, gn) + none[2:]
        if d_last is None and d_cls is None and not taps:
            return none
        if d_last is None and d_cls is None:
            d16 = torch.zeros((B * L, H), dtype=torch.bfloat16, device=ctx.ids.device)
        elif d_last is None:  # gradient enters at the [CLS] rows only (every reference wrapper)
            d16 = ops.scatter_cls_grad(d_cls.float().contiguous(), L)
        else:
            d16 = d_last.reshape(B * L, H).to(torch.bfloat16).contiguous()
            if d_cls is not None:
                if d16.data_ptr() == d_last.data_ptr():
                    d16 = d16.clone()
                d16.view(B, L, H)[:, 0] += d_cls.to(torch.bfloat16)
        if taps:
            gd, gn = model._run_backward_taps(ctx.ids, ctx.mask, d16, ctx.arena, ctx.lay, taps)
        else:
            gd, gn = model._run_backward(ctx.ids, ctx.mask, d16, ctx.arena)
        ctx.arena = None
        return (gd, gn) + none[2:]


# =============================================================================== packed (variable-length) batches
#: rows a packed sequence's extent is rounded up to.  1 (default): every sequence is stored on exactly max(length, 1) rows - no
#: padding rows at all but the <= 31 that make the batch's row count a multiple of 32; 32: the layout of rounds 3-4 (every
#: extent a multiple of 32: ~16 padding rows per sequence, 17 % of an MS MARCO-shaped batch of 64 x 128)
PACK_ALIGN = 1


def packed_extents(lengths, B: int, L: int, align: Optional[int] = None):
    """Host arithmetic of the packed layout: sequence b gets an extent of max(len_b, 1) rows rounded up to ``align``
    (``PACK_ALIGN``; a fully masked sequence keeps one masked row); the rows that make the total a multiple of 32 go to the
    last sequences that have room below ``ceil32(L)``.  Returns (int32 [2B + 1] = lengths followed by the B + 1 row offsets,
    T, longest extent rounded up to a multiple of 32)."""
    import numpy as np
    align = PACK_ALIGN if align is None else int(align)
    lens = np.ascontiguousarray(np.asarray(lengths).reshape(-1), dtype=np.int64)
    if lens.shape[0] != B or (lens < 0).any() or (lens > L).any():
        raise ValueError(f"packed batch: lengths must be {B} integers in [0, {L}]")
    ext = (np.maximum(lens, 1) + align - 1) // align * align
    pad = int(-ext.sum() % 32)
    cap = (L + 31) // 32 * 32
    b = B - 1
    while pad > 0:  # (cap - ext sums to a number congruent to pad modulo 32 and >= 0: there is always room)
        assert b >= 0, "packed_extents: no room for the rows that make T a multiple of 32"
        give = min(pad, int(cap - ext[b]))
        ext[b] += give
        pad -= give
        b -= 1
    host = np.zeros(2 * B + 1, np.int32)
    host[:B] = lens
    np.cumsum(ext, out=host[B + 1:])
    return host, int(host[-1]), int((ext.max() + 31) // 32 * 32)


class NotPrefixMask(ValueError):
    """A batch whose layout was planned on the device turned out not to be packable (some attention mask is not 1 .. 1 0 .. 0)."""


class PackedIndex:
    """Device-side description of a batch stored back to back (include/cocodr.h "Packed batches"): sequence b owns rows
    [seq_off[b], seq_off[b+1]): its length (``PACK_ALIGN`` = 1), see ``packed_extents``.  ``src`` maps packed row -> row of the padded [B*L] layout.

    Built per batch by ONE native launch (``cocodr_pack_index``) from the padded ids and the B lengths.  The row count T must
    reach the host (it sizes every GEMM of the step):
      * ``lengths`` known on the host - the reference's collators pad on the CPU (COCO/data.py:135-144,
        ANCE/data/msmarco_data.py:381-382), ``collate.CoCondenserCollator`` emits them - : extents and offsets are a numpy
        cumsum, one small pinned host -> device copy, nothing waits for the device queue;
      * only the device mask (the reference's batch unchanged): the whole layout is planned on the device (``from_mask``:
        ``cocodr_mask_lengths`` -> ``cocodr_pack_plan`` -> ``cocodr_pack_index`` queued back to back) and 16 bytes - T, the
        longest extent, "all prefix masks" - come back through a pinned buffer; the host waits for the stream to reach the
        plan kernel (unavoidable: the mask may have been produced by the work queued just before) and nothing else."""

    def __init__(self, ids: torch.Tensor, lens_host, L: Optional[int] = None):
        ids, B, L_in, Lp = self._check_ids(ids, L)
        host, self.T, self.max_len = packed_extents(lens_host, B, L_in)
        dev = ids.device
        self.B, self.L = B, Lp
        self.lengths = host[:B]
        # attention launches hand the sequences to workgroups longest first (the partial last round over the CUs gets the cheap ones)
        import numpy as np
        order = np.argsort(-np.diff(host[B:].astype(np.int64)), kind="stable").astype(np.int32)
        staged = torch.from_numpy(np.concatenate([host, order])).pin_memory().to(dev, non_blocking=True)
        self._finish(ids, staged, self.T)

    @staticmethod
    def _check_ids(ids: torch.Tensor, L: Optional[int]):
        if ids.dim() != 2 or ids.dtype not in (torch.int32, torch.int64):
            raise ValueError("PackedIndex: ids must be an int32 / int64 [B, L] tensor")
        if not ids.is_cuda:
            raise RuntimeError("PackedIndex describes a batch in HBM: move the ids with .to('cuda') (there is no CPU fallback)")
        if ids.stride(1) != 1:
            ids = ids.contiguous()
        B, L_in = ids.shape
        return ids, B, L_in, ((L_in + 31) // 32 * 32 if L is None else int(L))

    def _finish(self, ids: torch.Tensor, staged: torch.Tensor, rows: int, launch: bool = True):
        """``staged`` = lens [B] | seq_off [B + 1] | seq_order [B] on the device; ``rows`` = the row count the [T] arrays are
        allocated for (T, or its upper bound B x L when T is still on its way to the host)."""
        B, Lp, dev = self.B, self.L, ids.device
        self.seq_off = staged[B:2 * B + 1]
        self.seq_order = staged[2 * B + 1:]
        buf = torch.empty(4 * rows, dtype=torch.int32, device=dev)
        self._rows_buf = (buf, torch.empty(rows, dtype=torch.int64, device=dev), rows)
        self._keep = (staged, buf)
        if launch:
            self._launch_index(ids, staged)
            self._bind(self.T)

    def _launch_index(self, ids: torch.Tensor, staged: torch.Tensor):
        buf, src, rows = self._rows_buf
        check(lib().cocodr_pack_index(ptr(ids), ids.element_size(), ids.stride(0), ptr(staged), ptr(self.seq_off), self.B, self.L, ptr(buf),
                                      ptr(buf[rows:]), ptr(buf[2 * rows:]), ptr(buf[3 * rows:]), ptr(src), stream_ptr()), "pack_index")

    def _bind(self, T: int):
        buf, src, rows = self._rows_buf
        self.ids, self.positions, self.mask, self.cls_slot = buf[:T], buf[rows:rows + T], buf[2 * rows:2 * rows + T], buf[3 * rows:3 * rows + T]
        self.src = src[:T]
        self.c_struct = N.PackedBatch(self.ids.data_ptr(), self.positions.data_ptr(), self.mask.data_ptr(), self.seq_off.data_ptr(),
                                      self.cls_slot.data_ptr(), self.B, T, self.max_len, self.L, self.seq_order.data_ptr())

    @property
    def cls_rows(self) -> torch.Tensor:
        return self.seq_off[:-1].to(torch.int64)

    #: sequences per batch up to which the layout of a mask-only batch is planned on the device (cocodr_pack_plan: one workgroup)
    PLAN_MAX_B = 4096

    @classmethod
    def from_mask(cls, ids: torch.Tensor, mask: torch.Tensor, lazy: bool = False) -> Optional["PackedIndex"]:
        """The reference's batch unchanged - padded ids + attention mask in HBM, no lengths on the host (COCO/data.py:150-154,
        ANCE/data/msmarco_data.py:381-382): lengths, extents, offsets, order AND the row arrays are all built on the device
        (``cocodr_mask_lengths`` -> ``cocodr_pack_plan`` -> ``cocodr_pack_index``, queued back to back); the host reads back 16
        bytes - {T, longest extent, prefix masks?} - through a pinned buffer behind an event recorded in front of the last
        launch, because T sizes every GEMM of the step.  The wait ends when the stream reaches the plan kernel (it cannot end
        earlier: the mask may be the result of work queued just before).  ``lazy``: return at once and wait at the first use of
        ``T`` / the [T] arrays / ``c_struct`` - the encoder forward prepares everything that does not depend on T (weight structs,
        the arena at its padded bound, the autograd node) first, so ~20 us of host work separate the wait from the first launch;
        a batch that turns out not to be packable then raises ``NotPrefixMask`` at that use (``CocoBertModel.forward`` falls back to
        the padded execution).  Not lazy: None when some mask is not a prefix mask."""
        ids, B, L_in, Lp = cls._check_ids(ids, None)
        dev = ids.device
        self = cls.__new__(cls)
        self.B, self.L, self.lengths = B, Lp, None
        scratch = torch.empty(2 * B + 4, dtype=torch.int32, device=dev)  # lens | prefix_ok | result
        staged = torch.empty(3 * B + 1, dtype=torch.int32, device=dev)
        check(lib().cocodr_mask_lengths(ptr(mask), mask.element_size(), B, L_in, mask.stride(0), ptr(scratch), ptr(scratch[B:]), stream_ptr()),
              "mask_lengths")
        check(lib().cocodr_pack_plan(ptr(scratch), ptr(scratch[B:]), B, Lp, ptr(staged), ptr(scratch[2 * B:]), stream_ptr()), "pack_plan")
        host = torch.empty(4, dtype=torch.int32, pin_memory=True)
        host.copy_(scratch[2 * B:], non_blocking=True)
        ready = torch.cuda.Event()
        ready.record()
        self._finish(ids, staged, B * Lp, launch=False)
        self._launch_index(ids, staged)  # (takes lens / offsets from the plan: queued before the host knows T)
        self._keep = self._keep + (scratch,)
        self._pending = (ready, host)
        if lazy:  # T, max_len, the [T] arrays and c_struct resolve at first use (the encoder prepares everything else first)
            return self
        return self if self.resolve() else None

    #: attributes that exist once T has reached the host
    _LAZY = frozenset(("T", "max_len", "ids", "positions", "mask", "cls_slot", "src", "c_struct"))

    def resolve(self) -> bool:
        """Wait (if still pending) for the 16 bytes of a device-planned layout; False: the masks are not prefix masks."""
        pend = self.__dict__.get("_pending")
        if pend is None:
            return self.__dict__.get("_ok", True)
        ready, host = pend
        ready.synchronize()
        T, max_len, ok, _ = (int(v) for v in host.tolist())
        self._pending, self._ok = None, bool(ok)
        if ok:
            self.T, self.max_len = T, max_len
            self._bind(T)
        return self._ok

    def __getattr__(self, name):  # (only reached when the attribute is not set yet)
        if name in PackedIndex._LAZY:
            if self.__dict__.get("_pending") is not None and self.resolve():
                return self.__dict__[name]
            if self.__dict__.get("_ok") is False:
                raise NotPrefixMask("the batch cannot be packed: an attention mask is not a prefix mask (1 .. 1 0 .. 0)")
        raise AttributeError(name)

    @staticmethod
    def build(ids: torch.Tensor, mask: Optional[torch.Tensor] = None, lengths=None, lazy: bool = False) -> Optional["PackedIndex"]:
        """``lengths`` (host integers, = attention_mask.sum(1) of prefix masks): no device -> host traffic.  Otherwise the layout
        is planned on the device from the mask and 16 bytes come back (``from_mask``); None when a mask is not a prefix mask
        (the reference pads at the end, COCO/data.py:135-144; anything else runs padded)."""
        if lengths is not None and torch.is_tensor(lengths) and lengths.is_cuda:
            # (a trainer moved the batch to the device: these are no longer host-known - the mask route plans on the device, or they do)
            lengths = None if mask is not None else lengths.cpu()
        if lengths is not None:
            lengths = lengths.numpy() if torch.is_tensor(lengths) else lengths
            pk = PackedIndex(ids, lengths)
            if mask is not None and PackedIndex.check_lengths:
                pk.assert_lengths_match(mask)
            return pk
        B, L = ids.shape
        if mask is None:
            import numpy as np
            return PackedIndex(ids, np.full(B, L, np.int64))
        if mask.shape != ids.shape:
            raise ValueError("attention_mask shape must match input_ids")
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        if mask.element_size() not in (1, 4, 8) or mask.is_floating_point() or mask.stride(1) != 1:
            mask = mask.to(torch.int32).contiguous()
        if B <= PackedIndex.PLAN_MAX_B and PACK_ALIGN == 1:
            return PackedIndex.from_mask(ids, mask, lazy)
        out = torch.empty(2 * B, dtype=torch.int32, device=ids.device)
        check(lib().cocodr_mask_lengths(ptr(mask), mask.element_size(), B, L, mask.stride(0), ptr(out), ptr(out[B:]), stream_ptr()), "mask_lengths")
        host = out.cpu().numpy()
        if not host[B:].all():
            return None
        return PackedIndex(ids, host[:B])

    #: debug switch: with host ``lengths`` AND a mask, compare them on the device (one launch + a read-back per call)
    check_lengths = False

    def assert_lengths_match(self, mask: torch.Tensor) -> None:
        """Host-passed ``lengths`` are trusted over the mask (no read-back on the fast path); this check - ``PackedIndex.check_lengths
        = True``, or call it directly - raises when they are not the mask's prefix lengths (lengths counted without [CLS] / [SEP], a
        mask with holes: the packed run would silently attend a different token set than the reference's padded run)."""
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        if mask.element_size() not in (1, 4, 8) or mask.is_floating_point() or mask.stride(1) != 1:
            mask = mask.to(torch.int32).contiguous()
        B, L = mask.shape
        out = torch.empty(2 * B, dtype=torch.int32, device=mask.device)
        check(lib().cocodr_mask_lengths(ptr(mask), mask.element_size(), B, L, mask.stride(0), ptr(out), ptr(out[B:]), stream_ptr()), "mask_lengths")
        host = out.cpu().numpy()
        import numpy as np
        if not host[B:].all() or not np.array_equal(host[:B], np.asarray(self.lengths).reshape(-1)):
            raise ValueError("PackedIndex: `lengths` are not the prefix lengths of `attention_mask` (lengths must equal "
                             "attention_mask.sum(1) and every mask must be 1 .. 1 0 .. 0)")

    def unpack(self, x: torch.Tensor) -> torch.Tensor:
        """[T, H] -> padded [B, L, H]; rows past a sequence's extent are zeros (the padded path leaves masked garbage there)."""
        out = x.new_zeros((self.B * self.L, x.shape[-1]))
        out[self.src] = x
        return out.view(self.B, self.L, x.shape[-1])


class _PackedEncoderFn(torch.autograd.Function):
    """The encoder on a packed batch: (flat_decay, flat_nodecay, PackedIndex) -> (last hidden [B,L,H] bf16, cls fp32 [B,H])."""

    @staticmethod
    def forward(ctx, flat_decay, flat_nodecay, model, pk: PackedIndex, grad_mode: bool, cls_only: bool = False):
        training = bool(grad_mode and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        arena, lay = model._run_forward_packed(pk, training, cls_tail=cls_only)
        ctx.tail = arena._cocodr_tail
        H, NL = model.config.hidden_size, model.config.num_hidden_layers
        hidden = arena[lay.hidden: lay.hidden + (NL + 1) * pk.T * H * 2].view(torch.bfloat16).view(NL + 1, pk.T, H)
        cls = arena[lay.cls_f32: lay.cls_f32 + pk.B * H * 4].view(torch.float32).view(pk.B, H)
        ctx.model, ctx.pk, ctx.training = model, pk, training
        ctx.arena = arena if training else None
        ctx.set_materialize_grads(False)
        if ctx.tail:
            return None, cls.clone(), None
        stack = hidden.detach()   # (see _EncoderFn.forward: the stack travels with the call)
        ctx.mark_non_differentiable(stack)
        return hidden[NL], cls.clone(), stack  # the last layer stays packed [T, H]; PackedIndex.unpack (differentiable) pads it on demand

    @staticmethod
    def backward(ctx, d_last, d_cls, _d_stack=None):
        model, pk = ctx.model, ctx.pk
        if not ctx.training or ctx.arena is None:
            raise RuntimeError("encoder backward called but the forward ran without saved activations")
        if d_last is None and d_cls is None:
            return None, None, None, None, None, None
        H = model.config.hidden_size
        if ctx.tail:
            gd, gn = model._run_backward_packed(pk, d_cls.to(torch.bfloat16).contiguous(), ctx.arena)
            ctx.arena = None
            return gd, gn, None, None, None, None
        if d_last is None:
            d16 = torch.zeros((pk.T, H), dtype=torch.bfloat16, device=d_cls.device)
        else:
            d16 = d_last.to(torch.bfloat16).contiguous()
            if d_cls is not None and d16.data_ptr() == d_last.data_ptr():
                d16 = d16.clone()
        if d_cls is not None:
            d16[pk.cls_rows] += d_cls.to(torch.bfloat16)
        gd, gn = model._run_backward_packed(pk, d16, ctx.arena)
        ctx.arena = None
        return gd, gn, None, None, None, None


# =============================================================================== model
def _resize_rows(t: torch.Tensor, n: int) -> torch.Tensor:
    """first dimension cut or zero-padded to n (hf pads a resized decoder bias with zeros)"""
    out = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    keep = min(n, t.shape[0])
    out[:keep] = t[:keep]
    return out


class CocoBertModel(FlatParamsMixin, nn.Module):
    """BertModel (no pooler) on the native gfx950 kernels.  HF-compatible: ``from_pretrained``,
    ``save_pretrained``, ``state_dict`` key names, ``forward(input_ids=, attention_mask=)`` ->
    object with ``[0]`` / ``.last_hidden_state`` / ``.hidden_states``."""

    def __init__(self, config: CocoBertConfig, device: Optional[torch.device] = None):
        super().__init__()
        config = CocoBertConfig.coerce(config)
        self.config = config
        self.layout = _Layout(config)
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self.flat_decay = nn.Parameter(torch.zeros(self.layout.decay_numel, dtype=torch.float32, device=dev))
        self.flat_nodecay = nn.Parameter(torch.zeros(self.layout.nodecay_numel, dtype=torch.float32, device=dev))
        self._shadow = None          # bf16 copy of the weight-matrix region
        self._shadow_version = -1
        self._extra_state: Dict[str, torch.Tensor] = {}  # checkpoint tensors outside the encoder (pooler, heads ...)
        self.dropout_seed: Optional[int] = None  # None: torch.initial_seed() at the first dropout forward
        self._dropout_calls = 0
        # store batches back to back (no padding rows beyond 32-token alignment) instead of padded to one length: same
        # outputs at the real tokens, ~1/3 fewer rows on MS MARCO-shaped batches (include/cocodr.h "Packed batches").  The
        # default since round 4; False = every kernel over all B x L rows, as the reference computes
        self.pack_sequences = True
        self.cls_tail = True  # encode_cls(): last layer on the [CLS] rows only (cocodr_config.cls_tail); False = always the full layer
        self.reset_parameters()
        self._build_views()  # HF-named nn.Parameter views of the flats: what parameters() / named_parameters() yield

    # ---------------------------------------------------------------- init / HF naming
    def reset_parameters(self):
        """HF ``_init_weights``: normal(0, initializer_range) for Linear / Embedding weights, zeros for
        biases, ones / zeros for LayerNorm (ANCE/model/models.py:54-60 restates the same rule)."""
        with torch.no_grad():
            self.flat_decay.normal_(0.0, self.config.initializer_range)
            self.flat_nodecay.zero_()
            for name in self.layout.names:
                if name.endswith("LayerNorm.weight"):
                    self.hf_view(name).fill_(1.0)
            pad = self.config.pad_token_id
            if pad is not None and 0 <= pad < self.config.vocab_size:
                self.hf_view("embeddings.word_embeddings.weight")[pad].zero_()  # nn.Embedding(padding_idx=0)

    def hf_view(self, name: str) -> torch.Tensor:
        return self.layout.view((self.flat_decay.data, self.flat_nodecay.data), name)

    def hf_named_parameters(self):
        """(HF name, plain tensor view into the flat parameter) pairs.  ``named_parameters()`` yields the same names with
        ``nn.Parameter`` views (cocodr_amd.flatparams) whose ``.grad`` aliases the flat gradient."""
        for name in self.layout.names:
            yield name, self.hf_view(name)

    def hf_named_grads(self):
        flats = (self.flat_decay.grad, self.flat_nodecay.grad)
        for name in self.layout.names:
            if flats[self.layout.names[name][0]] is not None:
                yield name, self.layout.view(flats, name)

    def param_groups(self, weight_decay: float = 0.0) -> List[dict]:
        """AdamW groups as the HF Trainer builds them (COCO/trainer.py:66-70 -> Trainer.create_optimizer):
        no decay on biases and LayerNorm weights."""
        return [{"params": [self.flat_decay], "weight_decay": weight_decay},
                {"params": [self.flat_nodecay], "weight_decay": 0.0}]

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = OrderedDict() if destination is None else destination
        for name, v in self.hf_named_parameters():
            sd[prefix + name] = v if keep_vars else v.detach().clone()
        for name, v in self._extra_state.items():
            sd[prefix + name] = v
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        missing, unexpected = [], []
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("bert."):
                k = k[len("bert."):]
            sd[k] = v
        with torch.no_grad():
            for name, dst in self.hf_named_parameters():
                if name in sd:
                    src = sd.pop(name)
                    if tuple(src.shape) != tuple(dst.shape):
                        raise RuntimeError(f"size mismatch for {name}: checkpoint {tuple(src.shape)} vs model {tuple(dst.shape)}")
                    dst.copy_(src.to(dst.dtype))
                else:
                    missing.append(name)
        for k, v in sd.items():
            if k == "embeddings.position_ids":
                continue
            unexpected.append(k)
            self._extra_state[k] = v
        if strict and missing:
            raise RuntimeError(f"missing keys in state_dict: {missing[:8]}{' ...' if len(missing) > 8 else ''}")
        self._shadow_version = -1
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    @classmethod
    def from_pretrained(cls, path: str, config: Optional[CocoBertConfig] = None, device=None, **unused):
        config = config or CocoBertConfig.from_pretrained(path)
        model = cls(config, device=device)
        st = os.path.join(path, "model.safetensors")
        pt = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(pt):
            sd = torch.load(pt, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
        model.load_state_dict(sd, strict=False)
        return model

    def save_pretrained(self, path: str, prefix: str = "") -> None:
        """``prefix`` ("bert.") writes the encoder tensors under the base-model name a head-carrying HF class expects
        (BertForMaskedLM / BertForSequenceClassification); tensors this model only carries along keep their names."""
        from safetensors.torch import save_file
        self.config.save_pretrained(path)
        own = {name for name, _ in self.hf_named_parameters()}
        sd = {(prefix + k if k in own else k): v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})

    def resize_token_embeddings(self, n: Optional[int] = None):
        """hf ``PreTrainedModel.resize_token_embeddings`` (COCO/run_coco_pre_training.py:158 calls it with ``len(tokenizer)``,
        before the optimizer exists): the word table grows or shrinks to ``n`` rows - kept rows are copied, new rows drawn
        like ``_init_weights`` draws an embedding (normal(0, initializer_range)) - and whoever shares the vocabulary (the
        tied MLM decoder bias of a Condenser head, ``_vocab_listeners``) follows.  The flat parameters are re-created, so
        optimizers and ``enable_grad_allreduce`` must be set up afterwards, as in the reference."""
        old = self.config.vocab_size
        if n is None or int(n) == old:
            return self
        n = int(n)
        if n <= 0:
            raise ValueError(f"resize_token_embeddings: vocabulary size must be positive, got {n}")
        if getattr(self, "_dp_hooks", None):
            raise RuntimeError("resize_token_embeddings after enable_grad_allreduce: resize first (the flat parameters are re-created)")
        old_views = {name: v.detach().clone() for name, v in self.hf_named_parameters()}
        dev = self.flat_decay.device
        self.config.vocab_size = n
        self.layout = _Layout(self.config)
        self.flat_decay = nn.Parameter(torch.zeros(self.layout.decay_numel, dtype=torch.float32, device=dev))
        word = "embeddings.word_embeddings.weight"
        with torch.no_grad():
            for name, dst in self.hf_named_parameters():
                if name == word:
                    keep = min(old, n)
                    dst[:keep].copy_(old_views[name][:keep])
                    if n > keep:
                        dst[keep:].normal_(0.0, self.config.initializer_range)
                else:
                    dst.copy_(old_views[name])
        for k in ("cls.predictions.decoder.weight", "cls.predictions.decoder.bias"):  # tied tensors a checkpoint carried along:
            self._extra_state.pop(k, None)                                              # transformers re-ties them on load
        bias = self._extra_state.get("cls.predictions.bias")
        if bias is not None and bias.shape[0] == old:
            self._extra_state["cls.predictions.bias"] = _resize_rows(bias, n)
        self._shadow, self._shadow_version = None, -1
        self._build_views()
        for fn in getattr(self, "_vocab_listeners", []):
            fn(old, n)
        return self

    def get_extended_attention_mask(self, attention_mask, input_shape=None, device=None):
        """Additive key-padding mask [B,1,1,L] (0 / finfo.min) - COCO/modeling.py:57-61,193-197."""
        m = attention_mask[:, None, None, :].to(torch.float32)
        return (1.0 - m) * torch.finfo(torch.float32).min

    # ---------------------------------------------------------------- native plumbing
    def _c_config(self, drop=None, cls_tail: bool = False) -> N.Config:
        """``drop`` = (hidden p, attention p, seed, call) of a training forward with dropout, or None (no dropout);
        ``cls_tail``: the last layer computes its [CLS] rows only (cocodr_config.cls_tail)."""
        c = self.config
        ph, pa, seed, call = drop if drop is not None else (0.0, 0.0, 0, 0)
        return N.Config(c.hidden_size, c.num_attention_heads, c.num_hidden_layers, c.intermediate_size, c.vocab_size,
                        c.max_position_embeddings, c.layer_norm_eps, ph, pa, seed, call, int(cls_tail))

    # ---------------------------------------------------------------- dropout (hf nn.Dropout under model.train())
    def _next_dropout_peek(self) -> bool:
        """True when a training forward of this module would drop (train() mode and a positive probability)."""
        c = self.config
        return bool(self.training and torch.is_grad_enabled() and (c.hidden_dropout_prob > 0 or c.attention_probs_dropout_prob > 0))

    def _next_dropout(self, training: bool):
        """The (p_hidden, p_attention, seed, call) tuple of the next training forward, or None when nothing drops: eval
        mode, torch.no_grad() inference, or both probabilities 0.  Every call gets fresh masks (the counter), the
        backward of a forward regenerates the same masks from the same tuple; the seed is torch's at first use unless
        ``dropout_seed`` was set."""
        c = self.config
        if not (training and self.training) or (c.hidden_dropout_prob <= 0 and c.attention_probs_dropout_prob <= 0):
            return None
        if self.dropout_seed is None:
            self.dropout_seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        self._dropout_calls += 1
        return (float(c.hidden_dropout_prob), float(c.attention_probs_dropout_prob), int(self.dropout_seed), self._dropout_calls)

    def _ensure_shadow(self):
        lo = self.layout
        n = lo.decay_numel - lo.mat_begin
        if self._shadow is None or self._shadow.device != self.flat_decay.device:
            self._shadow = torch.empty(n, dtype=torch.bfloat16, device=self.flat_decay.device)
            self._shadow_version = -1

    def _shadow_target(self):
        """(bf16 shadow, first element of flat_decay it mirrors) - what an optimizer pass writes next to the fp32 master"""
        self._ensure_shadow()
        return self._shadow, self.layout.mat_begin

    def _refresh_shadow(self):
        lo = self.layout
        self._ensure_shadow()
        if self._shadow_stale():
            ops.cast_f32_bf16(self.flat_decay.data[lo.mat_begin:], self._shadow)
            self._shadow_mark_fresh()

    def _param_structs(self, grads: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        lo, cfg = self.layout, self.config
        H = cfg.hidden_size
        fd, fn = self.flat_decay.data, self.flat_nodecay.data
        pd, pn, ps = fd.data_ptr(), fn.data_ptr(), self._shadow.data_ptr()
        emb = N.EmbedParams(pd, pd + 4 * cfg.vocab_size * H, pd + 4 * (cfg.vocab_size + cfg.max_position_embeddings) * H,
                            pn, pn + 4 * H)
        if grads is None:
            arr, _ = lo.layer_structs(ps, lo.mat_begin, pn)
            return emb, arr, None, None
        gd, gn = grads[0].data_ptr(), grads[1].data_ptr()
        arr, garr = lo.layer_structs(ps, lo.mat_begin, pn, (gd, gn))
        eg = N.EmbedGrads(gd, gd + 4 * cfg.vocab_size * H, gd + 4 * (cfg.vocab_size + cfg.max_position_embeddings) * H,
                          gn, gn + 4 * H)
        return emb, arr, eg, garr

    def _layout_for(self, B: int, L: int, training: bool) -> N.EncoderLayout:
        lay = N.EncoderLayout()
        cfg = self._c_config()
        check(lib().cocodr_encoder_layout(C.byref(cfg), B, L, int(training), C.byref(lay)), "encoder_layout")
        return lay

    def _run_forward(self, ids: torch.Tensor, mask: torch.Tensor, training: bool, cls_tail: bool = False):
        if not self.flat_decay.is_cuda:
            raise RuntimeError("CocoBertModel runs on an MI355X only: move it with .to('cuda') (there is no CPU fallback)")
        B, L = ids.shape
        self._refresh_shadow()
        if training:
            self._dp_note_forward()
        lay = self._layout_for(B, L, training)
        arena = torch.empty(lay.total_bytes, dtype=torch.uint8, device=ids.device)
        emb, arr, _, _ = self._param_structs()
        arena._cocodr_drop = self._next_dropout(training)  # travels with the arena to whoever runs its backward
        arena._cocodr_tail = bool(cls_tail and arena._cocodr_drop is None and (not training or B % 8 == 0))  # (its weight gradients contract over B rows)
        cfg = self._c_config(arena._cocodr_drop, arena._cocodr_tail)
        check(lib().cocodr_encoder_fwd(C.byref(cfg), C.byref(emb), arr, ptr(ids), ptr(mask), B, L, int(training), ptr(arena),
                                       arena.numel(), stream_ptr()), "encoder_fwd")
        return arena, lay

    def _run_forward_packed(self, pk: "PackedIndex", training: bool, cls_tail: bool = False):
        if not self.flat_decay.is_cuda:
            raise RuntimeError("CocoBertModel runs on an MI355X only: move it with .to('cuda') (there is no CPU fallback)")
        # Everything that does not depend on the row count T comes first: a layout planned on the device (PackedIndex.from_mask,
        # the reference's batch without host lengths) is still on its way to the host, and the wait for it should be followed by
        # as little host work as possible (the GPU is idle from the moment it has produced T until the first launch below)
        self._refresh_shadow()
        emb, arr, _, _ = self._param_structs()
        c = self.config  # (what _next_dropout will decide; it also advances the call counter, so it runs after the wait)
        drops = bool(training and self.training and (c.hidden_dropout_prob > 0 or c.attention_probs_dropout_prob > 0))
        tail = bool(cls_tail and not drops and (not training or pk.B % 8 == 0))
        # T changes with every batch; the arena is allocated at the size of the PADDED batch (B x L rows, the upper bound of T) so
        # that the caching allocator hands back the same block step after step instead of growing / splitting a 10-20 GB block
        # whenever a batch is a little longer than any before it (a hipMalloc of that size inside a step costs tens of ms)
        cap = N.EncoderLayout()
        check(lib().cocodr_encoder_layout_packed(C.byref(self._c_config(None, tail)), pk.B * pk.L, pk.B, int(training), C.byref(cap)), "encoder_layout_packed")
        arena = torch.empty(cap.total_bytes, dtype=torch.uint8, device=self.flat_decay.device)
        if not pk.resolve():  # <- the only wait of a step without host lengths
            raise NotPrefixMask("the batch cannot be packed: an attention mask is not a prefix mask (1 .. 1 0 .. 0)")
        if training:
            self._dp_note_forward()
        arena_drop = self._next_dropout(training)
        if (arena_drop is not None) != drops:   # (a real check: `assert` disappears under python -O)
            raise RuntimeError("internal: the dropout setting predicted for the arena does not match the one drawn for this forward")
        cfg = self._c_config(arena_drop, tail)
        lay = N.EncoderLayout()
        check(lib().cocodr_encoder_layout_packed(C.byref(cfg), pk.T, pk.B, int(training), C.byref(lay)), "encoder_layout_packed")
        if lay.total_bytes > cap.total_bytes:
            arena = torch.empty(lay.total_bytes, dtype=torch.uint8, device=self.flat_decay.device)
        arena._cocodr_drop = arena_drop
        arena._cocodr_tail = tail
        check(lib().cocodr_encoder_fwd_packed(C.byref(cfg), C.byref(emb), arr, C.byref(pk.c_struct), int(training), ptr(arena),
                                              arena.numel(), stream_ptr()), "encoder_fwd_packed")
        return arena, lay

    def _run_backward_packed(self, pk: "PackedIndex", d16: torch.Tensor, arena: torch.Tensor):
        lo, NL = self.layout, self.config.num_hidden_layers
        gd = torch.empty_like(self.flat_decay.data)
        gn = torch.empty_like(self.flat_nodecay.data)
        ops.zero_f32(gd[:lo.mat_begin])
        emb, arr, eg, garr = self._param_structs((gd, gn))
        cfg = self._c_config(getattr(arena, "_cocodr_drop", None), getattr(arena, "_cocodr_tail", False))

        def call(l_hi, l_lo, d_in, do_embed):
            check(lib().cocodr_encoder_bwd_packed(C.byref(cfg), C.byref(emb), arr, C.byref(eg), garr, C.byref(pk.c_struct),
                                                  ptr(d_in) if d_in is not None else None, ptr(arena), arena.numel(), l_hi, l_lo,
                                                  int(do_embed), stream_ptr()), "encoder_bwd_packed")

        if not self._dp_overlap_ok():
            call(NL, 0, d16, True)
            return gd, gn
        bounds = self._dp_bounds()
        works = []
        for ci in reversed(range(len(bounds) - 1)):
            l_lo, l_hi = bounds[ci], bounds[ci + 1]
            call(l_hi, l_lo, d16 if l_hi == NL else None, l_lo == 0)
            works += self._dp_reduce_async(self._grad_range(gd, gn, l_lo, l_hi))
        self._dp_finish(works)
        self._dp_mark_reduced(self.flat_decay, self.flat_nodecay)
        return gd, gn

    def _dp_bounds(self):
        """layer boundaries of the overlapped backward ranges, ascending, balanced by gradient BYTES: the range that ends at layer
        0 also carries the embedding tables (its all-reduce is the one nothing can hide), so it gets correspondingly fewer layers"""
        lo, NL = self.layout, self.config.num_hidden_layers
        nchunk = min(self._dp_chunks, NL)
        emb_units = lo.mat_begin / float(lo.mat_stride)
        bounds = [0] + [min(NL, max(1, round(i * (NL + emb_units) / nchunk - emb_units))) for i in range(1, nchunk)] + [NL]
        return sorted(set(bounds))

    def enable_grad_allreduce(self, group=None, chunks: int = 4, extra_params=()) -> None:
        """Data-parallel gradient averaging without DDP (ANCE/drivers/run_ann.py:177-184, HF Trainer for COCO).

        * ONE encoder pass in the step (the COCO contrastive step): the backward walks the layer stack in ``chunks``
          ranges (top-down); as soon as a range's kernels are enqueued its slice of the two flat gradient tensors is
          all-reduced asynchronously (RCCL over xGMI, on the process group's stream) while the next range computes.
        * SEVERAL passes through the same weights (ANCE: query pass + passage pass, iDRO: up to three): every pass runs its
          plain backward, autograd sums the passes, and the summed gradient is reduced ONCE from a post-accumulate hook -
          never one all-reduce of the whole model per pass.
        * the full coCondenser step reduces the head's gradients under the backbone's backward and the upper backbone range
          under the lower one (condenser._CondenserStepFn.backward).
        ``no_sync()`` suspends all of it for gradient-accumulation micro-steps (DDP semantics, the pattern of
        ANCE/drivers/run_ann.py:318-341): gradients of those micro-steps stay local and un-reduced in ``.grad``; the next
        synchronised backward then does NOT reduce in flight (that would average only its own share) but lets the
        post-accumulate hook reduce the accumulated ``.grad`` once.
        ``extra_params``: further flat parameters that take part (the Condenser head's; ``condenser_step`` adopts them itself)."""
        self._dp_group = group
        self._dp_chunks = max(1, int(chunks))
        self._dp_enabled = True
        self._dp_fwd_live = 0        # training forwards of the current step (since the parameters last changed) without a backward
        self._dp_live_version = -1   # flat_decay._version those forwards were counted at
        self._dp_skip = set()        # id(param): its next hook call is a no-op, the backward reduced this pass in flight
        self._dp_unsynced = set()    # id(param): .grad holds local gradient of no_sync micro-steps
        if not getattr(self, "_dp_hooks", None):
            self._dp_hooks = {}
        self._dp_adopt(self.flat_decay, self.flat_nodecay, *extra_params)

    def _dp_adopt(self, *params) -> None:
        """register the reduce-on-accumulate hook on flat parameters that do not have it yet (idempotent)"""
        for p in params:
            if id(p) not in self._dp_hooks:
                self._dp_hooks[id(p)] = p.register_post_accumulate_grad_hook(self._dp_post_accumulate)

    def _dp_adopt_ddp_wrapper(self) -> None:
        """``torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
        find_unused_parameters=True)`` is how every reference driver makes the model data-parallel
        (ANCE/drivers/run_ann.py:177-184, ANCE/drivers/run_ann_data_gen.py:146-153; HF Trainer for COCO, COCO/trainer.py:181-182).
        torch's reducer hangs its hooks on the per-tensor parameters - here HF-named VIEWS of two flat tensors that autograd never
        visits (the native backward writes the flat gradients), so on its own the wrapper would reduce nothing and the ranks would
        drift apart without an error.  When a training forward finds itself inside such a wrapper (DDP publishes the running
        wrapper in ``_active_ddp_module``) the model therefore takes the job over: ``enable_grad_allreduce`` on the wrapper's
        process group (same averaged gradients, reduced range by range under the backward), the wrapper's own reducer is switched
        to its pass-through state (``require_backward_grad_sync = False``: what its ``no_sync()`` sets), and the wrapper's
        ``no_sync()`` is rebound to this model's, so gradient-accumulation loops written against DDP
        (ANCE/drivers/run_ann.py:318-341) keep their meaning.  tests/test_gpu_distributed.py::test_reference_ddp_wrap_line_*."""
        from torch.nn.parallel import DistributedDataParallel as DDP
        if not hasattr(DDP, "_active_ddp_module"):  # (a private attribute of torch's wrapper: say so when a torch release drops it)
            if not getattr(CocoBertModel, "_warned_ddp_api", False) and torch.distributed.is_available() and torch.distributed.is_initialized():
                CocoBertModel._warned_ddp_api = True
                import warnings
                warnings.warn("cocodr_amd: this torch has no DistributedDataParallel._active_ddp_module - a model wrapped in DDP cannot "
                              "detect its wrapper and its flat gradients would NOT be reduced; call model.enable_grad_allreduce() instead")
            return
        ddp = DDP._active_ddp_module
        if ddp is None:
            return
        adopted = ddp.__dict__.get("_cocodr_adopted")
        if adopted is None:
            # the wrapper's reducer is about to be made passive: that is only right when everything it wraps is a view of adopted
            # flat storage.  An ordinary nn.Parameter in the wrapped module (a projection head, DRO weights) would silently never be
            # reduced - refuse instead
            from .flatparams import ViewParameter
            plain = [n for n, q in ddp.module.named_parameters() if q.requires_grad and not isinstance(q, ViewParameter)]
            if plain:
                raise RuntimeError("DistributedDataParallel wraps ordinary parameters next to the cocodr_amd encoder (" + ", ".join(plain[:4]) +
                                   (", ..." if len(plain) > 4 else "") + "): the encoder reduces its own flat gradients and keeps torch's "
                                   "reducer passive, so these would never be averaged.  Wrap them in their own DDP module, or reduce the "
                                   "encoder with model.enable_grad_allreduce() and leave it out of the wrapper")
            adopted = ddp.__dict__["_cocodr_adopted"] = {"models": [], "in_no_sync": False}
            import contextlib

            @contextlib.contextmanager
            def no_sync(_state=adopted):
                with contextlib.ExitStack() as stack:
                    for m in _state["models"]:
                        stack.enter_context(m.no_sync())
                    _state["in_no_sync"] = True
                    try:
                        yield
                    finally:
                        _state["in_no_sync"] = False
            in_no_sync = not ddp.require_backward_grad_sync  # a first forward already inside the wrapper's own no_sync()
            ddp.no_sync = no_sync
        else:
            in_no_sync = adopted["in_no_sync"]
        if not any(m is self for m in adopted["models"]):
            adopted["models"].append(self)
            if not hasattr(self, "_dp_unsynced"):
                self.enable_grad_allreduce(group=ddp.process_group, chunks=2)
            elif self._dp_group is not ddp.process_group and self._dp_group is not None:
                raise RuntimeError("CocoBertModel: enable_grad_allreduce() was set up on a different process group than the "
                                   "DistributedDataParallel wrapper around this model")
        ddp.require_backward_grad_sync = False  # the wrapper's reducer has nothing to reduce: keep it passive
        if not adopted["in_no_sync"] and not getattr(self, "_dp_user_no_sync", 0):  # (a no_sync() the user opened on the model itself stands)
            self._dp_enabled = not in_no_sync

    def _dp_note_forward(self) -> None:
        """count a training forward of the current step; the count restarts whenever the parameters have changed since the last
        counted forward (an optimizer step: a forward whose backward never ran must not haunt the next step)"""
        self._dp_adopt_ddp_wrapper()
        if not getattr(self, "_dp_enabled", False):
            return
        v = self.flat_decay._version
        if v != self._dp_live_version:
            self._dp_live_version, self._dp_fwd_live = v, 0
            self._dp_skip.clear()
        self._dp_fwd_live += 1

    def _dp_overlap_ok(self) -> bool:
        """may this backward all-reduce its own gradient ranges in flight?  Only when it is the step's single pass through the
        weights AND no local, un-reduced gradient of earlier no_sync micro-steps sits in ``.grad``."""
        return bool(getattr(self, "_dp_enabled", False) and self._dp_fwd_live == 1 and not self._dp_unsynced)

    def _dp_mark_reduced(self, *params) -> None:
        self._dp_skip = {id(p) for p in params}

    def no_sync(self):
        """Context manager: gradients accumulate locally (DDP's ``no_sync`` for accumulation micro-steps)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            was = getattr(self, "_dp_enabled", False)
            self._dp_enabled = False
            self._dp_user_no_sync = getattr(self, "_dp_user_no_sync", 0) + 1
            try:
                yield
            finally:
                self._dp_user_no_sync -= 1
                self._dp_enabled = was
        return ctx()

    def _dp_reduce_async(self, tensors):
        import torch.distributed as dist
        native = dist.get_backend(self._dp_group) == "nccl"
        op = dist.ReduceOp.AVG if native else dist.ReduceOp.SUM  # gloo (CPU tests, two ranks on one GPU) has no AVG
        return [(dist.all_reduce(t, op=op, group=self._dp_group, async_op=True), t, native) for t in tensors if t.numel()]

    def _dp_finish(self, works) -> None:
        import torch.distributed as dist
        for w, _t, _n in works:
            w.wait()  # stream-level: the current stream waits for the collective, the host does not
        W = dist.get_world_size(self._dp_group)
        for _w, t, native in works:
            if not native:
                t.div_(W)

    def _dp_post_accumulate(self, p) -> None:
        if p.grad is None or not hasattr(self, "_dp_unsynced"):
            return
        if not getattr(self, "_dp_enabled", False):  # under no_sync(): remember that .grad now holds local gradient
            self._dp_unsynced.add(id(p))
            return
        self._dp_fwd_live = 0
        if id(p) in self._dp_skip:  # this pass was reduced in flight (only possible with nothing un-reduced in .grad)
            self._dp_skip.discard(id(p))
            return
        self._dp_unsynced.discard(id(p))
        n, k = p.grad.numel(), self._dp_chunks
        step = (n // k + 1023) // 1024 * 1024 if k > 1 else n
        self._dp_finish(self._dp_reduce_async([p.grad[a:a + step] for a in range(0, n, max(step, 1))]))

    def _run_backward(self, ids, mask, d_last16, arena):
        B, L = ids.shape
        lo = self.layout
        NL = self.config.num_hidden_layers
        gd = torch.empty_like(self.flat_decay.data)
        gn = torch.empty_like(self.flat_nodecay.data)
        ops.zero_f32(gd[:lo.mat_begin])  # embedding tables: sparse word rows are accumulated, unused position rows stay zero
        emb, arr, eg, garr = self._param_structs((gd, gn))
        cfg = self._c_config(getattr(arena, "_cocodr_drop", None), getattr(arena, "_cocodr_tail", False))
        if not self._dp_overlap_ok():  # several passes share the weights / local gradient pending: reduced once, from the hook
            check(lib().cocodr_encoder_bwd(C.byref(cfg), C.byref(emb), arr, C.byref(eg), garr, ptr(ids), ptr(mask), ptr(d_last16),
                                           B, L, ptr(arena), arena.numel(), stream_ptr()), "encoder_bwd")
            return gd, gn
        bounds = self._dp_bounds()
        nchunk = len(bounds) - 1
        works = []
        for ci in reversed(range(nchunk)):
            l_lo, l_hi = bounds[ci], bounds[ci + 1]
            top = l_hi == NL
            check(lib().cocodr_encoder_bwd_range(C.byref(cfg), C.byref(emb), arr, C.byref(eg), garr, ptr(ids), ptr(mask),
                                                 ptr(d_last16) if top else None, B, L, ptr(arena), arena.numel(), l_hi, l_lo,
                                                 int(l_lo == 0), stream_ptr()), "encoder_bwd_range")
            works += self._dp_reduce_async(self._grad_range(gd, gn, l_lo, l_hi))
        self._dp_finish(works)
        self._dp_mark_reduced(self.flat_decay, self.flat_nodecay)
        return gd, gn

    def _run_backward_taps(self, ids, mask, d_last16, arena, lay, taps: Dict[int, torch.Tensor]):
        """backward with gradients arriving at intermediate hidden states too: ``taps[l]`` = dL/d hidden_states[l] (the input
        of layer l; l = 0 is the embedding output).  Top-down in ranges that end at every tapped layer; the range leaves its
        input gradient in the arena, the tap is added there, the next range continues from it.  (Data-parallel runs reduce
        such a backward from the post-accumulate hook - no overlapped ranges here.)"""
        B, L = ids.shape
        lo, NL, H = self.layout, self.config.num_hidden_layers, self.config.hidden_size
        M = B * L
        gd = torch.empty_like(self.flat_decay.data)
        gn = torch.empty_like(self.flat_nodecay.data)
        ops.zero_f32(gd[:lo.mat_begin])
        emb, arr, eg, garr = self._param_structs((gd, gn))
        cfg = self._c_config(getattr(arena, "_cocodr_drop", None))
        dx_view = arena[lay.bwd_dx: lay.bwd_dx + M * H * 2].view(torch.bfloat16).view(B, L, H)

        def bwd_range(hi, lo_, d_in, do_embed):
            check(lib().cocodr_encoder_bwd_range(C.byref(cfg), C.byref(emb), arr, C.byref(eg), garr, ptr(ids), ptr(mask),
                                                 ptr(d_in) if d_in is not None else None, B, L, ptr(arena), arena.numel(), hi, lo_,
                                                 int(do_embed), stream_ptr()), "encoder_bwd_range")

        hi, d_in = NL, d_last16
        for l in sorted(taps, reverse=True):
            if not (0 <= l < NL) or tuple(taps[l].shape) != (B, L, H):
                raise ValueError(f"gradient of hidden_states[{l}] has shape {tuple(taps[l].shape)}, expected {(B, L, H)}")
            bwd_range(hi, l, d_in, False)  # leaves dL/d hidden_states[l] (through the layers above) in the arena
            dx_view += taps[l].to(torch.bfloat16)
            hi, d_in = l, None
        bwd_range(hi, 0, d_in, True)
        return gd, gn

    def _grad_range(self, gd, gn, l_lo: int, l_hi: int):
        """the slices of the two flat gradients that layers [l_lo, l_hi) own (+ the embedding blocks when l_lo == 0)"""
        lo = self.layout
        d0 = lo.mat_begin + l_lo * lo.mat_stride if l_lo > 0 else 0
        n0 = lo.vec_begin + l_lo * lo.vec_stride if l_lo > 0 else 0
        return [gd[d0:lo.mat_begin + l_hi * lo.mat_stride], gn[n0:lo.vec_begin + l_hi * lo.vec_stride]]

    # ---------------------------------------------------------------- public forward
    @staticmethod
    def _prep(input_ids, attention_mask):
        if input_ids.dim() != 2:
            raise ValueError(f"input_ids must be [B, L], got {tuple(input_ids.shape)}")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if attention_mask.shape != input_ids.shape:
            raise ValueError("attention_mask shape must match input_ids")
        B, L = input_ids.shape
        Lp = (L + 31) // 32 * 32
        ids = input_ids.to(torch.int32)
        mask = attention_mask.to(torch.int32)
        if Lp != L:  # pad to the kernels' 32-token granularity with masked [PAD]
            ids = torch.nn.functional.pad(ids, (0, Lp - L))
            mask = torch.nn.functional.pad(mask, (0, Lp - L))
        return ids.contiguous(), mask.contiguous(), L

    def pack(self, input_ids, attention_mask=None, lengths=None, lazy: bool = False) -> Optional["PackedIndex"]:
        """The packed-layout description of a batch (or None when its masks are not prefix masks), for callers that reuse a
        batch: ``forward`` builds it per call when ``pack_sequences`` is set (one native launch; see ``PackedIndex`` for what
        knowing the ``lengths`` on the host saves)."""
        if input_ids.dim() != 2:
            raise ValueError(f"input_ids must be [B, L], got {tuple(input_ids.shape)}")
        if input_ids.dtype not in (torch.int32, torch.int64):
            input_ids = input_ids.to(torch.int32)
        return PackedIndex.build(input_ids, attention_mask, lengths, lazy)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None,
                output_hidden_states: bool = False, return_dict: bool = True, packed_index: Optional["PackedIndex"] = None,
                cls_only: bool = False, lengths=None, **unused):
        """``cls_only`` (what ``encode_cls`` passes): the caller reads ``cls_fp32`` alone - the last layer then runs its output
        projection, LayerNorms and FFN on the [CLS] rows only (identical values, ~6 % less work; not with dropout, where the full
        layer runs); ``last_hidden_state`` / ``hidden_states`` are None.
        ``lengths``: the B sequence lengths on the HOST (= attention_mask.sum(1); what a collator that pads on the CPU knows) -
        with ``pack_sequences`` the packed layout is then built without reading anything back from the device."""
        self._check_token_types(token_type_ids)
        if position_ids is not None:
            raise NotImplementedError("custom position_ids are not on the reference path")
        if input_ids.dim() != 2:
            raise ValueError(f"input_ids must be [B, L], got {tuple(input_ids.shape)}")
        B, L = input_ids.shape
        if L > self.config.max_position_embeddings:
            raise ValueError(f"sequence length {L} exceeds max_position_embeddings={self.config.max_position_embeddings}")
        if attention_mask is not None and attention_mask.shape != input_ids.shape:
            raise ValueError("attention_mask shape must match input_ids")
        pk = packed_index
        if pk is not None and (pk.B, pk.L) != (B, (L + 31) // 32 * 32):
            raise ValueError(f"packed_index describes a {pk.B} x {pk.L} batch, the inputs are {tuple(input_ids.shape)}")
        # a training forward whose caller may read (and back-propagate through) ANY hidden_states[i] - the Condenser head under
        # the reference wrapper reads hidden_states[skip_from], COCO/modeling.py:212-216: only the padded Function makes the
        # intermediate states differentiable outputs ("taps"), so such a call runs padded
        want_taps = bool(output_hidden_states and torch.is_grad_enabled() and self.flat_decay.requires_grad)
        if want_taps:
            pk = None
        elif pk is None and self.pack_sequences:
            if not self.flat_decay.is_cuda:
                raise RuntimeError("CocoBertModel runs on an MI355X only: move it with .to('cuda') (there is no CPU fallback)")
            pk = self.pack(input_ids, attention_mask, lengths, lazy=True)  # None: not prefix masks -> padded
        cls_only = bool(cls_only and not output_hidden_states and self.cls_tail)
        stack = None
        if pk is not None:
            try:
                last, cls, stack = _PackedEncoderFn.apply(*self._flat_leaves(), self, pk, torch.is_grad_enabled(), cls_only)
            except NotPrefixMask:  # (a device-planned layout, resolved inside the forward: some mask has holes - run padded)
                pk = None
        ids = mask = None
        if pk is None:
            if attention_mask is None and lengths is not None:  # padded run of a batch described by its lengths alone
                lens_dev = torch.as_tensor(lengths, dtype=torch.int64).reshape(-1).to(input_ids.device)
                attention_mask = (torch.arange(L, device=input_ids.device)[None] < lens_dev[:, None]).to(torch.int32)
            ids, mask, L = self._prep(input_ids, attention_mask)
            outs = _EncoderFn.apply(*self._flat_leaves(), ids, mask, self, torch.is_grad_enabled(), want_taps, cls_only)
            last, cls, stack, taps = outs[0], outs[1], outs[2], outs[3:]
        if last is None:  # [CLS] tail
            out = EncoderOutput(None, None, cls)
            return out if return_dict else (None, None)
        hs = None
        if pk is not None:
            last_packed = last
            last = (lambda: pk.unpack(last_packed)[:, :L])
            if output_hidden_states:
                last = last()
                hs = tuple(pk.unpack(h)[:, :L] for h in stack[:-1].unbind(0)) + (last,)
            out = EncoderOutput(last, hs, cls)
            return out if return_dict else (out.last_hidden_state, None)
        if output_hidden_states:  # in a training forward every entry is differentiable (a head may read any layer)
            below = taps if taps else stack[:-1].unbind(0)
            hs = tuple(h[:, :L] for h in below) + (last[:, :L],)
        out = EncoderOutput(last[:, :L], hs, cls)
        return out if return_dict else (out.last_hidden_state, None)

    _TT_MSG = "token_type_ids != 0: the reference never passes segment ids (COCO/data.py:140)"

    def _check_token_types(self, token_type_ids) -> None:
        """A tokenizer's (all-zero) ``token_type_ids`` are accepted (README.md:101-116 passes them); non-zero ones are not on the
        reference path.  A host tensor is checked on the spot.  A device tensor is checked WITHOUT stopping the host: its
        "any non-zero" flag is copied to a pinned byte behind an event, and every later call (or ``check_inputs()``) raises for the
        flags that have arrived - so a wrong batch is reported a call late instead of every call paying a device round trip."""
        self._resolve_token_type_flags(block=False)
        if token_type_ids is None:
            return
        if not token_type_ids.is_cuda:
            if bool(token_type_ids.any()):
                raise NotImplementedError(self._TT_MSG)
            return
        pend = self.__dict__.setdefault("_tt_pending", [])
        if len(pend) >= 8:  # (a caller that never lets the device catch up: wait for the oldest)
            self._resolve_token_type_flags(block=True)
        host = torch.empty(1, dtype=torch.uint8).pin_memory()
        host.copy_(token_type_ids.ne(0).any().reshape(1).to(torch.uint8), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pend.append((host, ev))

    def _resolve_token_type_flags(self, block: bool) -> None:
        pend = self.__dict__.get("_tt_pending")
        while pend and (block or pend[0][1].query()):
            host, ev = pend.pop(0)
            ev.synchronize()
            if int(host[0]):
                pend.clear()
                raise NotImplementedError(self._TT_MSG + " (seen in an earlier call: device tensors are checked a call late)")

    def check_inputs(self) -> None:
        """Waits for the deferred input checks of earlier forward calls (device-side ``token_type_ids``) and raises what they found."""
        self._resolve_token_type_flags(block=True)

    def _flat_leaves(self):
        """The two tensors a forward attaches its autograd node to: the flat parameters - or, inside ``side_stream_aliases()``, views
        of them that were created on the caller's main stream."""
        return getattr(self, "_flat_alias", None) or (self.flat_decay, self.flat_nodecay)

    def side_stream_aliases(self):
        """Context for an encoder pass that runs on a SIDE stream next to another pass (``BertDotNLL``'s two-pass step): autograd
        replays a pass's backward on its forward stream, and a gradient that arrives at a leaf's AccumulateGrad node from another
        stream than the leaf's makes torch warn and synchronise there.  Entered on the MAIN stream, this hands the pass views of the
        flats whose ViewBackward nodes belong to the main stream: the engine joins the side stream in front of them (the join the
        step needs anyway) and AccumulateGrad only ever sees main-stream producers."""
        import contextlib
        model = self

        @contextlib.contextmanager
        def ctx():
            model._flat_alias = (model.flat_decay.view(-1), model.flat_nodecay.view(-1)) if torch.is_grad_enabled() else None
            try:
                yield
            finally:
                model._flat_alias = None
        return ctx()

    def encode_cls(self, input_ids, attention_mask=None, packed_index=None, lengths=None) -> torch.Tensor:
        """fp32 last-layer [CLS] rows [B,H] with autograd (what every reference wrapper consumes)."""
        return self.forward(input_ids, attention_mask, packed_index=packed_index, cls_only=True, lengths=lengths).cls_fp32


# =============================================================================== ANCE wrapper
class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, a, b, weights):
        loss, rows, logits, dq, da, db = ops.triplet_nll_fwd_bwd(q.contiguous(), a.contiguous(), b.contiguous(), weights)
        ctx.save_for_backward(dq, da, db)
        ctx.mark_non_differentiable(rows, logits)
        return loss[0], rows, logits

    @staticmethod
    def backward(ctx, g, _r, _l):
        dq, da, db = ctx.saved_tensors
        return dq * g, da * g, db * g, None


class BertDotNLL(nn.Module):
    """``BertDot_NLL_LN`` (ANCE/model/models.py:194-262): shared-weight bi-encoder, embedding = raw
    last-layer [CLS] (:225-229), triplet NLL over [q.pos, q.neg] (:97-106), ``(loss*weights).mean()``
    (:260-261).  The dead ``embeddingHead`` / ``norm`` / ``classifier`` parameters of the reference class
    are carried through checkpoints untouched."""

    def __init__(self, config: CocoBertConfig, model_argobj=None, device=None):
        super().__init__()
        config = CocoBertConfig.coerce(config)
        self.config = config
        self.bert = CocoBertModel(config, device=device)
        self.total = 0
        # with bert.pack_sequences (both defaults since round 4): queries + positives + negatives go through the encoder as ONE
        # packed pass - one forward, one backward, one weight-gradient pass over all rows - instead of the reference's three
        # (ANCE/model/models.py:84-86); same embeddings, loss and gradients.  False: a query pass next to a passage pass
        self.merge_passes = True
        self.dro_type, self.loss = "erm", None

    def add_group_loss(self, args=None, n_groups: int = 0, dro_type: str = "idro", alpha: float = 0.0, eps: float = 0.1,
                       ema: float = 0.1, rho: float = 0.1, weight_ema: bool = True):
        """ANCE/model/models.py:211-223.  ``args.model_size == 'large'`` re-weights the last 2 layers, otherwise the last 3
        (ANCE/model/dro_loss.py:177-181)."""
        from .idro import DROGreedyLoss, IDROLoss
        dev = self.bert.flat_decay.device
        if dro_type == "idro":
            self.loss = IDROLoss(n_groups, alpha, eps, ema, rho, model_size=getattr(args, "model_size", "base"), device=dev)
        elif dro_type == "dro-greedy":
            self.loss = DROGreedyLoss(n_groups, alpha, eps, ema, weight_ema, device=dev)
        else:
            raise ValueError("dro_type must be 'idro' or 'dro-greedy'")
        self.dro_type = dro_type
        self.n_groups = n_groups

    @classmethod
    def from_pretrained(cls, path, config=None, device=None, **unused):
        config = config or CocoBertConfig.from_pretrained(path)
        m = cls(config, device=device)
        m.bert = CocoBertModel.from_pretrained(path, config=config, device=device)
        return m

    def save_pretrained(self, path):
        self.bert.save_pretrained(path)

    def _side_stream(self):
        s = getattr(self, "_side", None)
        if s is None or s.device != self.bert.flat_decay.device:
            s = self._side = torch.cuda.Stream(device=self.bert.flat_decay.device)
        return s

    def query_emb(self, input_ids, attention_mask):
        return self.bert.encode_cls(input_ids, attention_mask)

    def _forward_merged(self, query_ids, attention_mask_q, input_ids_a, attention_mask_a, input_ids_b, attention_mask_b, lengths=None):
        """Queries, positives and negatives as ONE packed batch (``merge_passes`` with ``bert.pack_sequences``): the packed layout
        stores every sequence at its own length, so the [B, 64] queries and the [2B, 128] passages need not be separate encoder
        passes - one forward, one backward (one weight-gradient pass over all rows, no second full-size gradient to add), GEMMs at
        the row count of the whole step.  Returns None when the masks are not prefix masks (the caller runs the two passes)."""
        Lq, Lp = query_ids.shape[1], input_ids_a.shape[1]
        pad = lambda t: torch.nn.functional.pad(t, (0, Lp - Lq)) if Lp > Lq else t
        ids = torch.cat([pad(query_ids), input_ids_a, input_ids_b])
        if lengths is not None:  # host-known lengths of (queries, positives, negatives): no read-back of the masks
            import numpy as np
            lengths, mask = np.concatenate([np.asarray(x).reshape(-1) for x in lengths]), None
        else:
            mask = torch.cat([pad(attention_mask_q), attention_mask_a, attention_mask_b])
        pk = self.bert.pack(ids, mask, lengths)
        if pk is None:
            return None
        calls = lambda: self.bert._dropout_calls if self.bert._next_dropout_peek() else 0
        e = self.bert.encode_cls(ids, mask, packed_index=pk)
        self.last_passes = [("qab", calls())]
        B = query_ids.shape[0]
        return e[:B], e[B:2 * B], e[2 * B:]

    def body_emb(self, input_ids, attention_mask):
        return self.query_emb(input_ids, attention_mask)

    def forward(self, query_ids, attention_mask_q, input_ids_a=None, attention_mask_a=None, input_ids_b=None,
                attention_mask_b=None, is_query=True, group_ids=None, weights=None, lengths=None):
        """``lengths`` (optional, beyond the reference signature): the host-known lengths ``(len_q, len_a, len_b)`` of the three
        padded inputs - the merged packed pass is then laid out without reading the masks back from the device."""
        if input_ids_b is None:
            return self.query_emb(query_ids, attention_mask_q) if is_query else self.body_emb(query_ids, attention_mask_q)
        if group_ids is not None and getattr(self, "loss", None) is None:
            raise RuntimeError("group_ids need add_group_loss(...) first (ANCE/drivers/run_ann.py:904-906)")
        if group_ids is not None and self.dro_type == "idro":  # ANCE/model/models.py:259-273
            from .idro import idro_triplet_step
            robust, rows, logits, group_losses, group_counts = idro_triplet_step(
                self.bert, self.loss, query_ids, attention_mask_q, input_ids_a, attention_mask_a, input_ids_b, attention_mask_b,
                group_ids)
            self.total += rows.shape[0] * (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1)
            return robust, torch.argmax(logits, dim=1), group_losses, group_counts
        if self.merge_passes and self.bert.pack_sequences and input_ids_a.shape == input_ids_b.shape \
                and query_ids.shape[1] <= input_ids_a.shape[1] and query_ids.shape[0] == input_ids_a.shape[0]:
            merged = self._forward_merged(query_ids, attention_mask_q, input_ids_a, attention_mask_a, input_ids_b, attention_mask_b, lengths)
            if merged is not None:
                q, a, b = merged
                B = q.shape[0]
                w = None if weights is None else weights.to(torch.float32).contiguous()
                if group_ids is not None:
                    loss, rows, logits = _TripletFn.apply(q, a, b, self.loss.row_weights(group_ids, w).contiguous())
                    group_losses, group_counts = self.loss.update(rows, group_ids, w)
                    self.total += B * (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1)
                    return loss, torch.argmax(logits, dim=1), group_losses, group_counts
                loss, rows, logits = _TripletFn.apply(q, a, b, w)
                self.total += B * (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1)
                return loss, torch.argmax(logits, dim=1), logits
        # The short query pass (B x 64 tokens) fills a fraction of the CUs; it runs on a side stream next to the passage
        # pass.  Autograd replays each pass's backward on its forward stream, so the two backward passes overlap as well.
        main = torch.cuda.current_stream()
        side = self._side_stream()
        # the bf16 weight shadow both passes read is brought up to date HERE, on the main stream: refreshed inside the query
        # pass it would be written on the side stream while the passage pass (main stream) already believes it fresh
        self.bert._refresh_shadow()
        side.wait_stream(main)
        # which batches went through the encoder together, with the dropout call number each pass drew (0 = no dropout):
        # where the reference runs three passes (models.py:84-86), positives and negatives share one here when shapes allow
        calls = lambda: self.bert._dropout_calls if self.bert._next_dropout_peek() else 0
        self.last_passes = []
        with self.bert.side_stream_aliases(), torch.cuda.stream(side):
            q = self.query_emb(query_ids, attention_mask_q)
        self.last_passes.append(("q", calls()))
        B = q.shape[0]
        if input_ids_a.shape == input_ids_b.shape:  # one encoder pass for positives and negatives
            ab = self.body_emb(torch.cat([input_ids_a, input_ids_b]), torch.cat([attention_mask_a, attention_mask_b]))
            a, b = ab[:B], ab[B:]
            self.last_passes.append(("ab", calls()))
        else:
            a = self.body_emb(input_ids_a, attention_mask_a)
            self.last_passes.append(("a", calls()))
            b = self.body_emb(input_ids_b, attention_mask_b)
            self.last_passes.append(("b", calls()))
        main.wait_stream(side)
        q.record_stream(main)
        w = None if weights is None else weights.to(torch.float32).contiguous()
        if group_ids is not None:  # dro-greedy (ANCE/model/dro_loss.py:50-90): weights of the previous step, then the update
            loss, rows, logits = _TripletFn.apply(q, a, b, self.loss.row_weights(group_ids, w).contiguous())
            group_losses, group_counts = self.loss.update(rows, group_ids, w)
            self.total += B * (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1)
            return loss, torch.argmax(logits, dim=1), group_losses, group_counts
        loss, rows, logits = _TripletFn.apply(q, a, b, w)
        self.total += B * (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1)
        return loss, torch.argmax(logits, dim=1), logits


# =============================================================================== COCO wrapper
class _SimCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E_all, world, row0, m_local):
        loss, rows, dE = ops.simce_fwd_bwd(E_all.contiguous(), world, row0, m_local)
        ctx.save_for_backward(dE)
        ctx.row0, ctx.m_local, ctx.M = row0, m_local, E_all.shape[0]
        ctx.mark_non_differentiable(rows)
        ctx.set_materialize_grads(False)  # (no zero tensor for the per-row losses' missing gradient: one fill kernel per step)
        return loss[0], rows

    @staticmethod
    def backward(ctx, g, _rows):
        (dE,) = ctx.saved_tensors
        if g is None:
            return None, None, None, None
        if ctx.row0 == 0 and ctx.m_local == ctx.M:  # one rank: the local rows are all rows (no zero-filled [M, H] + slice copy)
            return dE * g, None, None, None
        full = dE.new_zeros((ctx.M, dE.shape[1]))
        full[ctx.row0:ctx.row0 + ctx.m_local] = dE * g
        return full, None, None, None


class _GatherRows(torch.autograd.Function):
    """COCO/modeling.py:182-190 ``gather_tensors``: all_gather the [2b,H] [CLS] block of every rank into
    one [W*2b,H] matrix; only the local slot carries gradient (:185), so the backward is a slice - no
    collective.  One contiguous RCCL all-gather instead of W buffers + cat + slot overwrite."""

    @staticmethod
    def forward(ctx, t):
        import torch.distributed as dist
        W, r = dist.get_world_size(), dist.get_rank()
        out = torch.empty((W * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous())
        ctx.r, ctx.m = r, t.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.r * ctx.m:(ctx.r + 1) * ctx.m]


class CoCondenserForPretraining(nn.Module):
    """``CoCondenserForPretraining`` (COCO/modeling.py:162-248): encoder -> last-layer [CLS] (:206) -> cross-rank
    gather (:207-208) -> span-pair InfoNCE (:244-248) -> ``.mean()`` (:229), plus - when ``model_args.n_head_layers``
    > 0 and ``labels`` are given - the Condenser head and the MLM losses (:212-224, cocodr_amd.condenser).
    ``forward(model_input, labels)`` keeps the reference signature and returns the summed loss."""

    def __init__(self, bert: CocoBertModel, model_args=None, data_args=None, train_args=None):
        super().__init__()
        self.lm = bert
        self.model_args, self.data_args, self.train_args = model_args, data_args, train_args
        n_head = int(getattr(model_args, "n_head_layers", 0) or 0) if model_args is not None else 0
        self.c_head = None
        if n_head > 0:
            from .condenser import CondenserHead
            self.c_head = CondenserHead(bert.config, n_head, device=bert.flat_decay.device)
            if not hasattr(bert, "_vocab_listeners"):
                bert._vocab_listeners = []
            bert._vocab_listeners.append(self.c_head.resize_vocab)  # lm.resize_token_embeddings also resizes lm.cls's decoder bias

    @staticmethod
    def _world_size():
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @classmethod
    def from_pretrained(cls, model_args, data_args, train_args, path, **kw):
        model = cls(CocoBertModel.from_pretrained(path, **kw), model_args, data_args, train_args)
        if model.c_head is not None:
            # the reference keeps the MLM head (cls.predictions.*) inside the HF checkpoint (AutoModelForMaskedLM) and the
            # Condenser layers (c_head.*) in model.pt next to it (COCO/modeling.py:103-107, 123-131)
            sd = {k: v for k, v in model.lm._extra_state.items() if k.startswith("cls.predictions.")}
            extra = os.path.join(path, "model.pt")
            if os.path.exists(extra):
                sd.update(torch.load(extra, map_location="cpu", weights_only=True))
            model.c_head.load_state_dict(sd, strict=False)
        return model

    def save_pretrained(self, output_dir: str):
        if self.c_head is not None:
            head = {k: v.detach().cpu() for k, v in self.c_head.state_dict().items()}
            for k, v in head.items():  # MLM head into the HF checkpoint
                if k.startswith("cls.predictions."):
                    self.lm._extra_state[k] = v
            for k in ("cls.predictions.decoder.weight", "cls.predictions.decoder.bias"):
                self.lm._extra_state.pop(k, None)  # tied to the word embeddings / cls.predictions.bias: transformers re-ties on load
        self.lm.save_pretrained(output_dir, prefix="bert." if self.c_head is not None else "")
        if self.c_head is not None:
            torch.save({k: v for k, v in head.items() if k.startswith("c_head.")}, os.path.join(output_dir, "model.pt"))

    def param_groups(self, weight_decay: float = 0.0):
        groups = self.lm.param_groups(weight_decay)
        if self.c_head is not None:
            groups += self.c_head.param_groups(weight_decay)
        return groups

    def compute_contrastive_loss(self, co_cls_hiddens: torch.Tensor) -> torch.Tensor:
        """Per-row loss [M] (already multiplied by world size), COCO/modeling.py:244-248."""
        W = self._world_size()
        _, rows = _SimCEFn.apply(co_cls_hiddens.float(), W, 0, co_cls_hiddens.shape[0])
        return rows

    def forward(self, model_input, labels=None, **unused):
        ids, mask = model_input["input_ids"], model_input.get("attention_mask")
        mlm_loss = None
        self.lm.eval()  # COCO/modeling.py:198: the backbone never drops; the Condenser head layers follow self.training
        if self.c_head is not None and labels is not None:
            from .condenser import condenser_step
            skip_from = int(getattr(self.model_args, "skip_from", 2))
            late_mlm = bool(getattr(self.model_args, "late_mlm", False))
            mlm_loss, cls = condenser_step(self.lm, self.c_head, ids, mask, labels, skip_from, late_mlm, lengths=model_input.get("lengths"))
        else:
            cls = self.lm.encode_cls(ids, mask, packed_index=model_input.get("packed_index"), lengths=model_input.get("lengths"))  # [2b, H] fp32
        W = self._world_size()
        force = bool(getattr(self, "force_gather", False)) and torch.distributed.is_initialized()  # (tests: the N > 1 path on a 1-rank group)
        if W > 1 or force:
            import torch.distributed as dist
            E = _GatherRows.apply(cls)
            row0 = dist.get_rank() * cls.shape[0]
        else:
            E, row0 = cls, 0
        loss, _rows = _SimCEFn.apply(E, W, row0, cls.shape[0])
        return loss if mlm_loss is None else loss + mlm_loss
