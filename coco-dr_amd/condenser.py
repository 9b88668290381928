"""The rest of the coCondenser pre-training step: the Condenser head and the two masked-LM losses on
top of the encoder / contrastive path.

Reference: COCO/modeling.py:192-235 -
    lm_out   = lm(**input, labels, output_hidden_states=True)               # backbone + "late" MLM loss
    hiddens  = cat(hidden_states[-1][:, :1], hidden_states[skip_from][:, 1:])
    hiddens  = c_head layers (n_head_layers BertLayers, :43-46)(hiddens)
    loss     = CE(lm.cls(hiddens), labels) [+ lm_out.loss if late_mlm] + contrastive.mean()
`lm.cls` is hf BertOnlyMLMHead: dense + erf-GELU + LayerNorm, then a decoder tied to the word embeddings (+ bias).

Native mapping: the head layers reuse the encoder's layer kernels through `cocodr_stack_fwd` /
`cocodr_encoder_bwd_range`; the MLM head runs ONLY on the labelled rows (~15 % of the tokens; the reference forms the
full [B*L, V] logits and lets CE ignore -100 - same loss, 6x less work), both applications of `lm.cls` batched into
one set of GEMMs; the vocabulary CE is `cocodr_ce_fwd_bwd`.  The head's gradient re-enters the backbone at
`skip_from` between two ranges of the ranged encoder backward.  Row gather / scatter and the [CLS] splice are torch
indexing ops (plumbing on a few MB).
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Optional

import torch
from torch import nn

from . import _native as N
from . import ops
from ._native import check, lib, ptr, stream_ptr
from .flatparams import FlatParamsMixin
from .modeling import CocoBertConfig, CocoBertModel, _Layout

__all__ = ["CondenserHead", "condenser_step"]

_NEG = -1e30


class CondenserHead(FlatParamsMixin, nn.Module):
    """Parameters of `c_head` (COCO/modeling.py:43-46) and of `lm.cls` (hf BertOnlyMLMHead) in two flat tensors.
    State-dict names follow the reference: ``c_head.{i}.attention.self.query.weight`` ...,
    ``cls.predictions.transform.dense.weight`` ..., ``cls.predictions.bias`` (the decoder weight is the backbone's
    word-embedding table)."""

    #: (a fused decoder GEMM + cross entropy was built in round 4 and measured slower than logits + ``cocodr_ce_fwd_bwd`` - 235 against
    #: 173 us, profiles/r04_decoder_ce.md; the code is archived in tools/experiments/decoder_ce_removed.hip.txt)
    #: contraction slices of the decoder's input-gradient GEMM ([n, H] = dlogits . Word over K = 30 720: 30 tiles of 256 rows would
    #: leave most CUs idle; slices run as the batch items of one launch).  Measured 2 / 4 / 8: profiles/r04_decoder_ce.md
    decoder_split_k = 8

    def __init__(self, config: CocoBertConfig, n_head_layers: int = 2, device=None):
        super().__init__()
        H, V = config.hidden_size, config.vocab_size
        self.config = config
        self.n_head_layers = int(n_head_layers)
        self.layout = self._build_layout()
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self.flat_decay = nn.Parameter(torch.zeros(self.layout.decay_numel, dtype=torch.float32, device=dev))
        self.flat_nodecay = nn.Parameter(torch.zeros(self.layout.nodecay_numel, dtype=torch.float32, device=dev))
        self._shadow = None
        self._shadow_version = -1
        self.dropout_seed = None  # None: torch.initial_seed() at the first dropout forward
        self._dropout_calls = 0
        self.reset_parameters()
        self._build_views()

    def _build_layout(self):
        H, V = self.config.hidden_size, self.config.vocab_size
        self.vpad = (V + 127) // 128 * 128
        return _Layout(
            self.config, self.n_head_layers, "c_head.",
            decay_pre=[("cls.predictions.transform.dense.weight", (H, H))],
            nodecay_pre=[("cls.predictions.transform.dense.bias", (H,)), ("cls.predictions.transform.LayerNorm.weight", (H,)),
                         ("cls.predictions.transform.LayerNorm.bias", (H,)), ("cls.predictions.bias", (V,))])

    def resize_vocab(self, old: int, new: int) -> None:
        """follow ``lm.resize_token_embeddings`` (the config object is the backbone's and already carries the new size): the
        decoder bias is cut or zero-padded like hf's ``_get_resized_lm_head`` does, everything else is copied"""
        keep = {name: self.hf_view(name).detach().clone() for name in self.layout.names}
        self.layout = self._build_layout()
        self.flat_nodecay = nn.Parameter(torch.zeros(self.layout.nodecay_numel, dtype=torch.float32, device=self.flat_nodecay.device))
        with torch.no_grad():
            for name in self.layout.names:
                dst = self.hf_view(name)
                if name == "cls.predictions.bias":
                    n = min(old, new)
                    dst[:n].copy_(keep[name][:n])
                else:
                    dst.copy_(keep[name])
        self._shadow_version = -1
        self._build_views()

    def reset_parameters(self):
        with torch.no_grad():
            self.flat_decay.normal_(0.0, self.config.initializer_range)
            self.flat_nodecay.zero_()
            for name in self.layout.names:
                if name.endswith("LayerNorm.weight"):
                    self.hf_view(name).fill_(1.0)

    def hf_view(self, name):
        return self.layout.view((self.flat_decay.data, self.flat_nodecay.data), name)

    def _next_dropout(self):
        """(p_hidden, p_attention, seed, call) of the next forward of the c_head BertLayers, or None in eval mode / with both
        probabilities 0.  The reference's c_head layers are the only part of the COCO step that drops
        (COCO/modeling.py:198 puts the backbone in eval, trainer.py:146 the rest in train)."""
        c = self.config
        if not self.training or (c.hidden_dropout_prob <= 0 and c.attention_probs_dropout_prob <= 0):
            return None
        if self.dropout_seed is None:
            self.dropout_seed = (int(torch.initial_seed()) ^ 0x5DEECE66D) & (2 ** 63 - 1)
        self._dropout_calls += 1
        return (float(c.hidden_dropout_prob), float(c.attention_probs_dropout_prob), int(self.dropout_seed), self._dropout_calls)

    def hf_named_grads(self):
        flats = (self.flat_decay.grad, self.flat_nodecay.grad)
        for name in self.layout.names:
            if flats[self.layout.names[name][0]] is not None:
                yield name, self.layout.view(flats, name)

    def param_groups(self, weight_decay: float = 0.0):
        return [{"params": [self.flat_decay], "weight_decay": weight_decay},
                {"params": [self.flat_nodecay], "weight_decay": 0.0}]

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = OrderedDict() if destination is None else destination
        for name in self.layout.names:
            v = self.hf_view(name)
            sd[prefix + name] = v if keep_vars else v.detach().clone()
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        missing = []
        with torch.no_grad():
            for name in self.layout.names:
                if name in state_dict:
                    self.hf_view(name).copy_(state_dict[name].to(torch.float32))
                else:
                    missing.append(name)
        if strict and missing:
            raise RuntimeError(f"missing keys in state_dict: {missing[:8]}")
        self._shadow_version = -1
        return torch.nn.modules.module._IncompatibleKeys(missing, [])

    def _shadow_target(self):
        if self._shadow is None or self._shadow.device != self.flat_decay.device:
            self._shadow = torch.empty(self.layout.decay_numel, dtype=torch.bfloat16, device=self.flat_decay.device)
            self._shadow_version = -1
        return self._shadow, 0

    def _refresh_shadow(self):
        self._shadow_target()
        if self._shadow_stale():
            ops.cast_f32_bf16(self.flat_decay.data, self._shadow)
            self._shadow_mark_fresh()


class _CondenserStepFn(torch.autograd.Function):
    """(backbone flats, head flats, ids, mask, labels) -> (mlm loss scalar, fp32 [CLS] rows).  The contrastive loss
    is applied on the returned [CLS] rows by the caller (it needs the cross-rank gather)."""

    @staticmethod
    def forward(ctx, fd, fn, hd, hn, ids, mask, labels, bert: CocoBertModel, head: CondenserHead, skip_from: int, late_mlm: bool, pk=None):
        """``pk`` (a PackedIndex): backbone and head run on the packed layout - T stored rows instead of B x L; ``labels`` are
        then the labels of the packed rows [T].  Same arithmetic per real token, same losses and gradients."""
        cfg = bert.config
        B, L = ids.shape
        H, V, NL = cfg.hidden_size, cfg.vocab_size, cfg.num_hidden_layers
        M = pk.T if pk is not None else B * L
        nh = head.n_head_layers
        dev = ids.device
        rows = torch.nonzero(labels.reshape(-1) != -100).squeeze(1)  # host sync: the GEMM row count must be known
        n_lab = int(rows.numel())
        if n_lab == 0:
            raise ValueError("condenser step: no labelled positions in the batch (cross_entropy would be NaN)")
        lab = labels.reshape(-1)[rows].to(torch.int32)
        # first row of every sequence in the [M, H] activations
        cls_rows = pk.cls_rows if pk is not None else torch.arange(B, device=dev, dtype=torch.int64) * L
        # ---- backbone
        if pk is not None:
            arena, lay = bert._run_forward_packed(pk, True)
        else:
            arena, lay = bert._run_forward(ids, mask, True)
        hidden = arena[lay.hidden: lay.hidden + (NL + 1) * M * H * 2].view(torch.bfloat16).view(NL + 1, M, H)
        cls = arena[lay.cls_f32: lay.cls_f32 + B * H * 4].view(torch.float32).view(B, H).clone()
        last = hidden[NL]
        # ---- Condenser head on cat(cls of the last layer, skip_from states without their first token)
        head._refresh_shadow()
        hdrop = head._next_dropout() or (0.0, 0.0, 0, 0)
        hcfg = N.Config(H, cfg.num_attention_heads, nh, cfg.intermediate_size, V, cfg.max_position_embeddings, cfg.layer_norm_eps, *hdrop)
        hlay = N.EncoderLayout()
        if pk is not None:
            check(lib().cocodr_encoder_layout_packed(C.byref(hcfg), pk.T, B, 1, C.byref(hlay)), "encoder_layout_packed(head)")
            hcap = N.EncoderLayout()  # (allocated at the padded upper bound, as the backbone's arena: CocoBertModel._run_forward_packed)
            check(lib().cocodr_encoder_layout_packed(C.byref(hcfg), B * pk.L, B, 1, C.byref(hcap)), "encoder_layout_packed(head)")
            harena = torch.empty(max(hlay.total_bytes, hcap.total_bytes), dtype=torch.uint8, device=dev)
        else:
            check(lib().cocodr_encoder_layout(C.byref(hcfg), B, L, 1, C.byref(hlay)), "encoder_layout(head)")
            harena = torch.empty(hlay.total_bytes, dtype=torch.uint8, device=dev)
        hhidden = harena[hlay.hidden: hlay.hidden + (nh + 1) * M * H * 2].view(torch.bfloat16).view(nh + 1, M, H)
        hhidden[0].copy_(hidden[skip_from])
        ops.scatter_rows(last[cls_rows].contiguous(), cls_rows, hhidden[0])
        hlo = head.layout
        harr, _ = hlo.layer_structs(head._shadow.data_ptr(), 0, head.flat_nodecay.data_ptr())
        if pk is not None:
            check(lib().cocodr_stack_fwd_packed(C.byref(hcfg), harr, C.byref(pk.c_struct), 1, ptr(harena), harena.numel(), stream_ptr()), "stack_fwd_packed")
        else:
            check(lib().cocodr_stack_fwd(C.byref(hcfg), harr, ptr(mask), B, L, 1, ptr(harena), harena.numel(), stream_ptr()), "stack_fwd")
        head_out = hhidden[nh]
        # ---- lm.cls on the labelled rows of the head output (and of the last backbone layer: "late" MLM)
        # each group of rows is padded to a multiple of 64 with copies of row 0 that carry scale 0 (their dlogits,
        # and therefore every gradient they touch, are exactly zero): the wgrad GEMMs contract over this axis
        n_pad = (n_lab + 63) // 64 * 64
        rows_p = torch.cat([rows, rows.new_zeros(n_pad - n_lab)])
        lab_p = torch.cat([lab, lab.new_zeros(n_pad - n_lab)])
        scale_p = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        scale_p[:n_lab] = 1.0 / n_lab
        n2 = 2 * n_pad if late_mlm else n_pad
        xg = torch.empty((n2, H), dtype=torch.bfloat16, device=dev)  # labelled rows of the head output [| of the last backbone layer]
        ops.gather_rows(head_out.reshape(M, H), rows_p, xg[:n_pad])
        if late_mlm:
            ops.gather_rows(last.reshape(M, H), rows_p, xg[n_pad:])
        wt = head._shadow[: H * H].view(H, H)
        b_t = head.hf_view("cls.predictions.transform.dense.bias")
        g_act, a_pre = ops.gemm(xg.contiguous(), wt, bias=b_t, epi=N.EPI_GELU)
        t, t_mean, t_rstd = ops.ln_fwd(g_act, head.hf_view("cls.predictions.transform.LayerNorm.weight"),
                                       head.hf_view("cls.predictions.transform.LayerNorm.bias"), cfg.layer_norm_eps)
        # tied decoder weight; rows padded to a multiple of 512 (bias -1e30 there: probability exactly 0, gradient exactly 0): whole
        # 256-column tiles for the logits GEMM, and a contraction the backward can cut into 8 slices of 64-wide steps
        vp = (V + 511) // 512 * 512
        word16 = torch.empty((vp, H), dtype=torch.bfloat16, device=dev)
        word16[V:].zero_()
        ops.cast_f32_bf16(bert.hf_view("embeddings.word_embeddings.weight"), word16[:V])
        dec_bias = torch.full((vp,), _NEG, dtype=torch.float32, device=dev)
        dec_bias[:V].copy_(head.hf_view("cls.predictions.bias"))
        scale = torch.cat([scale_p, scale_p]) if late_mlm else scale_p
        lab2 = torch.cat([lab_p, lab_p]) if late_mlm else lab_p
        logits = ops.gemm(t, word16, bias=dec_bias, out_f32=True)  # [n2, vp] fp32
        loss_rows = torch.empty(n2, dtype=torch.float32, device=dev)
        dlogits = torch.empty((n2, vp), dtype=torch.bfloat16, device=dev)
        check(lib().cocodr_ce_fwd_bwd(ptr(logits), ptr(lab2), ptr(scale), n2, V, vp, ptr(loss_rows), ptr(dlogits),
                                      stream_ptr()), "ce_fwd_bwd")
        del logits
        mlm_loss = (loss_rows * scale).sum()  # mean over the head rows + mean over the late rows
        ctx.bert, ctx.head = bert, head
        ctx.skip_from, ctx.late_mlm, ctx.n_lab, ctx.n_pad = skip_from, late_mlm, n_lab, n_pad
        ctx.arena, ctx.lay, ctx.harena, ctx.hlay, ctx.hcfg = arena, lay, harena, hlay, hcfg
        ctx.ids, ctx.mask, ctx.pk, ctx.cls_rows = ids, mask, pk, cls_rows
        ctx.saved = (rows, xg, a_pre, g_act, t, t_mean, t_rstd, word16, dlogits)
        ctx.set_materialize_grads(False)
        return mlm_loss, cls

    @staticmethod
    def backward(ctx, g_mlm, d_cls):
        bert, head = ctx.bert, ctx.head
        cfg = bert.config
        B, L = ctx.ids.shape
        pk, cls_rows = ctx.pk, ctx.cls_rows
        H, V, NL = cfg.hidden_size, cfg.vocab_size, cfg.num_hidden_layers
        M = pk.T if pk is not None else B * L
        nh, n_lab, skip_from = head.n_head_layers, ctx.n_lab, ctx.skip_from
        rows, xg, a_pre, g_act, t, t_mean, t_rstd, word16, dlogits = ctx.saved
        dev = ctx.ids.device
        hlo, lo = head.layout, bert.layout
        ghd = torch.zeros_like(head.flat_decay.data)
        ghn = torch.zeros_like(head.flat_nodecay.data)
        gv = lambda name: hlo.view((ghd, ghn), name)
        d_head_out = torch.zeros((M, H), dtype=torch.bfloat16, device=dev)
        d_last = torch.zeros((M, H), dtype=torch.float32, device=dev)
        # backbone gradient flats (allocated here: the tied decoder's weight gradient is written straight into the word-table rows)
        bgd = torch.empty_like(bert.flat_decay.data)
        bgn = torch.empty_like(bert.flat_nodecay.data)
        vp = word16.shape[0]
        dword_mlm = None
        word_rows = 0  # rows of the embedding region the decoder's weight-gradient GEMM has already written (no zeroing there)
        if g_mlm is not None:
            # upstream scale (1.0 in the reference step; a device scalar, no host sync): applied to the SMALL operands - t for the
            # decoder's weight gradient, dt and the bias sums on the way out - instead of a pass over the [n2, vp] dlogits
            gs = g_mlm.to(torch.float32)
            dlog = dlogits
            # decoder (tied to the word embeddings) and its bias
            # [n2,H] = dlogits . Word: 30 output tiles of 256 rows over K = 30 720: eight K slices in one launch fill the CUs
            dt = ops.gemm(dlog, word16, trans_b=True, split_k=head.decoder_split_k).mul_(gs.to(torch.bfloat16))
            # [vp,H] = dlogits^T . (g t), written over the word-table rows of the gradient flat (rows >= V of it: exact zeros over the
            # first position rows, which the embedding backward overwrites / accumulates into like the zeros they replace)
            if vp * H <= lo.mat_begin:
                ops.gemm(dlog, t * gs.to(torch.bfloat16), trans_a=True, trans_b=True, out_f32=True, out=bgd[: vp * H].view(vp, H))
                word_rows = vp
            else:
                dword_mlm = ops.gemm(dlog, t * gs.to(torch.bfloat16), trans_a=True, trans_b=True, out_f32=True)
            gv("cls.predictions.bias").copy_(ops.colsum(dlog)[:V] * gs)
            # transform: LayerNorm, erf-GELU, dense
            dg, dlnw, dlnb = ops.ln_bwd(dt, g_act, head.hf_view("cls.predictions.transform.LayerNorm.weight"), t_mean, t_rstd)
            gv("cls.predictions.transform.LayerNorm.weight").copy_(dlnw)
            gv("cls.predictions.transform.LayerNorm.bias").copy_(dlnb)
            da = ops.mul_bf16(dg, a_pre)  # a_pre holds GELU'(pre-activation) (EPI_GELU's C2)
            gv("cls.predictions.transform.dense.weight").copy_(ops.gemm(da, xg, trans_a=True, trans_b=True, out_f32=True))
            gv("cls.predictions.transform.dense.bias").copy_(ops.colsum(da))
            wt = head._shadow[: H * H].view(H, H)
            dxg = ops.gemm(da, wt, trans_b=True)                                        # [n2,H]
            ops.scatter_rows(dxg[:n_lab], rows, d_head_out)
            if ctx.late_mlm:
                ops.scatter_rows(dxg[ctx.n_pad:ctx.n_pad + n_lab], rows, d_last)
        # ---- Condenser head backward (layers nh-1 .. 0), input gradient left at hlay.bwd_dx
        harr, hgarr = hlo.layer_structs(head._shadow.data_ptr(), 0, head.flat_nodecay.data_ptr(), (ghd.data_ptr(), ghn.data_ptr()))
        if pk is not None:
            check(lib().cocodr_encoder_bwd_packed(C.byref(ctx.hcfg), None, harr, None, hgarr, C.byref(pk.c_struct), ptr(d_head_out),
                                                  ptr(ctx.harena), ctx.harena.numel(), nh, 0, 0, stream_ptr()), "encoder_bwd_packed(head)")
        else:
            check(lib().cocodr_encoder_bwd_range(C.byref(ctx.hcfg), None, harr, None, hgarr, None, ptr(ctx.mask), ptr(d_head_out), B, L,
                                                 ptr(ctx.harena), ctx.harena.numel(), nh, 0, 0, stream_ptr()), "encoder_bwd_range(head)")
        # in flight only when this is the step's single pass and nothing un-reduced waits in .grad; otherwise the hooks on the
        # backbone AND head flats (adopted in condenser_step) reduce the accumulated gradients
        dp = bert._dp_overlap_ok()
        works = bert._dp_reduce_async([ghd, ghn]) if dp else []  # the head's gradients travel under the whole backbone backward
        d_hin = ctx.harena[ctx.hlay.bwd_dx: ctx.hlay.bwd_dx + M * H * 2].view(torch.bfloat16).view(M, H)
        d_last[cls_rows] += d_hin[cls_rows].float()
        if d_cls is not None:
            d_last[cls_rows] += d_cls.float()
        d_skip = d_hin.clone()
        d_skip[cls_rows] = 0
        # ---- backbone backward in two ranges; the head's gradient joins at hidden_states[skip_from]
        ops.zero_f32(bgd[word_rows * H: lo.mat_begin])
        emb, arr, eg, garr = bert._param_structs((bgd, bgn))
        bcfg = bert._c_config(getattr(ctx.arena, "_cocodr_drop", None))
        d_last16 = d_last.to(torch.bfloat16)
        dx_view = ctx.arena[ctx.lay.bwd_dx: ctx.lay.bwd_dx + M * H * 2].view(torch.bfloat16).view(M, H)

        def bwd_range(hi, lo_, d_in, do_embed):
            if pk is not None:
                check(lib().cocodr_encoder_bwd_packed(C.byref(bcfg), C.byref(emb), arr, C.byref(eg), garr, C.byref(pk.c_struct),
                                                      ptr(d_in) if d_in is not None else None, ptr(ctx.arena), ctx.arena.numel(),
                                                      hi, lo_, int(do_embed), stream_ptr()), "encoder_bwd_packed")
            else:
                check(lib().cocodr_encoder_bwd_range(C.byref(bcfg), C.byref(emb), arr, C.byref(eg), garr, ptr(ctx.ids), ptr(ctx.mask),
                                                     ptr(d_in) if d_in is not None else None, B, L, ptr(ctx.arena), ctx.arena.numel(),
                                                     hi, lo_, int(do_embed), stream_ptr()), "encoder_bwd_range")

        if skip_from >= NL or skip_from == 0:  # the head reads the last layer / the embedding output: one range
            if skip_from >= NL:
                bwd_range(NL, 0, (d_last + d_skip.float()).to(torch.bfloat16), True)
            else:
                bwd_range(NL, 0, d_last16, False)
                dx_view += d_skip
                bwd_range(0, 0, None, True)
            lower = (0, NL)
        else:
            bwd_range(NL, skip_from, d_last16, False)
            if dp:  # layers [skip_from, NL) are final: reduced while the lower range computes
                works += bert._dp_reduce_async(bert._grad_range(bgd, bgn, skip_from, NL))
            dx_view += d_skip
            bwd_range(skip_from, 0, None, True)
            lower = (0, skip_from)
        if dword_mlm is not None:
            bgd[: V * H].view(V, H).add_(dword_mlm[:V])
        if dp:
            works += bert._dp_reduce_async(bert._grad_range(bgd, bgn, *lower))
            bert._dp_finish(works)
            bert._dp_mark_reduced(bert.flat_decay, bert.flat_nodecay, head.flat_decay, head.flat_nodecay)
        ctx.arena = ctx.harena = None
        ctx.saved = None
        return bgd, bgn, ghd, ghn, None, None, None, None, None, None, None, None


def condenser_step(bert: CocoBertModel, head: CondenserHead, input_ids, attention_mask, labels, skip_from: int, late_mlm: bool,
                   lengths=None):
    """Returns (mlm_loss, cls_fp32): the MLM part of COCO/modeling.py:222-224 and the last-layer [CLS] rows for the
    contrastive part (:206-210, :226-230).  With ``bert.pack_sequences`` (the default) backbone and head run on the packed
    layout (``lengths``: the B lengths on the host, when the collator has them - no read-back of the mask)."""
    if not (0 <= skip_from <= bert.config.num_hidden_layers):
        raise ValueError(f"skip_from={skip_from} outside [0, {bert.config.num_hidden_layers}]")
    if torch.is_grad_enabled():
        bert._dp_adopt_ddp_wrapper()  # (a torch DistributedDataParallel wrapper around the model: the model reduces, see there)
    if getattr(bert, "_dp_hooks", None) is not None and hasattr(bert, "_dp_unsynced"):
        bert._dp_adopt(head.flat_decay, head.flat_nodecay)  # the head's gradients are averaged with the backbone's, in flight or by hook
    # (hidden-state dropout in the head layers indexes token rows: a packed and a padded run of one batch draw different - equally
    #  valid - masks there; the placement tests run padded, include/cocodr.h "Packed batches")
    pk = bert.pack(input_ids, attention_mask, lengths) if bert.pack_sequences else None
    if pk is not None:
        B, L = input_ids.shape
        lab = labels
        if pk.L != L:
            lab = torch.nn.functional.pad(labels, (0, pk.L - L), value=-100)
        lab_packed = torch.where(pk.mask > 0, lab.reshape(-1)[pk.src], torch.full_like(pk.src, -100))
        ids32 = input_ids if input_ids.dtype == torch.int32 else input_ids.to(torch.int32)
        return _CondenserStepFn.apply(bert.flat_decay, bert.flat_nodecay, head.flat_decay, head.flat_nodecay, ids32, None,
                                      lab_packed.contiguous(), bert, head, int(skip_from), bool(late_mlm), pk)
    ids, mask, L = bert._prep(input_ids, attention_mask)
    if L != ids.shape[1]:
        labels = torch.nn.functional.pad(labels, (0, ids.shape[1] - L), value=-100)
    return _CondenserStepFn.apply(bert.flat_decay, bert.flat_nodecay, head.flat_decay, head.flat_nodecay, ids, mask,
                                  labels.contiguous(), bert, head, int(skip_from), bool(late_mlm), None)
