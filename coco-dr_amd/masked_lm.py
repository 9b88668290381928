"""`CocoBertForMaskedLM`: the object `AutoModelForMaskedLM.from_pretrained(path)` hands to the reference
(COCO/modeling.py:96-108) - a BERT encoder under hf's `BertOnlyMLMHead` - on the native kernels.

What the reference's `CondenserForPretraining` / `CoCondenserForPretraining` touch on that object:

* ``lm(**model_input, labels=labels, output_hidden_states=True, return_dict=True)`` -> ``.loss`` (MLM cross entropy, mean over
  the labelled positions, ignore_index -100) and ``.hidden_states`` (N + 1 tensors, every one differentiable: the Condenser
  head reads ``hidden_states[skip_from]``), COCO/modeling.py:199-204, 212-216, 224;
* ``lm.cls(hiddens)`` -> vocabulary logits of arbitrary hidden states (the head's output), COCO/modeling.py:85-93;
* ``lm.bert.get_extended_attention_mask``, ``lm.config``, ``lm.resize_token_embeddings``, ``lm.save_pretrained``,
  ``state_dict`` names ``bert.*`` / ``cls.predictions.*``.

`cocodr_amd.modeling.CoCondenserForPretraining` fuses all of this into one native step and is the fast path; this class is
the literal drop-in for code that keeps the reference's own wrapper.  The MLM head is native: two GEMMs (the transform
with the erf-GELU epilogue, the decoder tied to the word table), LayerNorm, and `cocodr_ce_fwd_bwd` on the labelled rows -
``.loss`` never materialises the [B, L, V] logits (``.logits`` does, lazily, as hf would)."""
import os
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from . import _native as N
from . import ops
from ._native import check, lib, ptr, stream_ptr
from .condenser import CondenserHead
from .modeling import CocoBertConfig, CocoBertModel


class _MLMHeadFn(torch.autograd.Function):
    """(x [M,H] bf16, head flats, word table [V,H] fp32) -> logits fp32 [M, vpad] (columns >= V: padding, bias ``pad_bias``)."""

    @staticmethod
    def forward(ctx, x, hd, hn, word, head: CondenserHead, pad_bias: float):
        cfg = head.config
        H, V = cfg.hidden_size, cfg.vocab_size
        head._refresh_shadow()
        wt = head._shadow[: H * H].view(H, H)
        g_act, a_pre = ops.gemm(x, wt, bias=head.hf_view("cls.predictions.transform.dense.bias"), epi=N.EPI_GELU)
        t, t_mean, t_rstd = ops.ln_fwd(g_act, head.hf_view("cls.predictions.transform.LayerNorm.weight"),
                                       head.hf_view("cls.predictions.transform.LayerNorm.bias"), cfg.layer_norm_eps)
        word16 = torch.zeros((head.vpad, H), dtype=torch.bfloat16, device=x.device)  # tied decoder weight, rows padded to 128
        ops.cast_f32_bf16(word.detach().contiguous(), word16[:V])
        dec_bias = torch.full((head.vpad,), float(pad_bias), dtype=torch.float32, device=x.device)
        dec_bias[:V].copy_(head.hf_view("cls.predictions.bias"))
        logits = ops.gemm(t, word16, bias=dec_bias, out_f32=True)
        ctx.head = head
        ctx.saved = (x, a_pre, g_act, t, t_mean, t_rstd, word16)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        head = ctx.head
        cfg = head.config
        H, V = cfg.hidden_size, cfg.vocab_size
        x, a_pre, g_act, t, t_mean, t_rstd, word16 = ctx.saved
        hlo = head.layout
        ghd = torch.zeros_like(head.flat_decay.data)
        ghn = torch.zeros_like(head.flat_nodecay.data)
        gv = lambda name: hlo.view((ghd, ghn), name)
        dlog = dlogits.to(torch.bfloat16).contiguous()
        if head.vpad > V:
            dlog[:, V:].zero_()
        dt = ops.gemm(dlog, word16, trans_b=True)                                      # [M,H] = dlogits . Word
        dword = ops.gemm(dlog, t, trans_a=True, trans_b=True, out_f32=True)[:V]         # [V,H] = dlogits^T . t
        gv("cls.predictions.bias").copy_(ops.colsum(dlog)[:V])
        dg, dlnw, dlnb = ops.ln_bwd(dt, g_act, head.hf_view("cls.predictions.transform.LayerNorm.weight"), t_mean, t_rstd)
        gv("cls.predictions.transform.LayerNorm.weight").copy_(dlnw)
        gv("cls.predictions.transform.LayerNorm.bias").copy_(dlnb)
        da = ops.mul_bf16(dg, a_pre)  # a_pre holds GELU'(pre-activation) (EPI_GELU's second output)
        gv("cls.predictions.transform.dense.weight").copy_(ops.gemm(da, x, trans_a=True, trans_b=True, out_f32=True))
        gv("cls.predictions.transform.dense.bias").copy_(ops.colsum(da))
        dx = ops.gemm(da, head._shadow[: H * H].view(H, H), trans_b=True)
        ctx.saved = None
        return dx, ghd, ghn, dword, None, None


class _SparseCEFn(torch.autograd.Function):
    """mean cross entropy of the first n rows of fp32 logits [n_pad, ld] (columns >= V padding) against int32 labels."""

    @staticmethod
    def forward(ctx, logits, labels, n: int, V: int):
        n_pad, ld = logits.shape
        scale = torch.zeros(n_pad, dtype=torch.float32, device=logits.device)
        scale[:n] = 1.0 / n
        loss_rows = torch.empty(n_pad, dtype=torch.float32, device=logits.device)
        dlogits = torch.empty((n_pad, ld), dtype=torch.bfloat16, device=logits.device)
        check(lib().cocodr_ce_fwd_bwd(ptr(logits), ptr(labels), ptr(scale), n_pad, V, ld, ptr(loss_rows), ptr(dlogits), stream_ptr()),
              "ce_fwd_bwd")
        ctx.dlogits = dlogits
        return (loss_rows * scale).sum()

    @staticmethod
    def backward(ctx, g):
        d = ctx.dlogits.float() * g
        ctx.dlogits = None
        return d, None, None, None


class CocoMLMHead(nn.Module):
    """``lm.cls`` (hf BertOnlyMLMHead): callable on hidden states of any leading shape, returns fp32 logits [..., V]."""

    def __init__(self, bert: CocoBertModel):
        super().__init__()
        self._bert = [bert]  # not a sub-module of the head: the owner registers it as `.bert`
        self.params = CondenserHead(bert.config, 0, device=bert.flat_decay.device)
        if not hasattr(bert, "_vocab_listeners"):
            bert._vocab_listeners = []
        bert._vocab_listeners.append(self.params.resize_vocab)

    def _word(self):
        bert = self._bert[0]
        V, H = bert.config.vocab_size, bert.config.hidden_size
        return bert.flat_decay[: V * H].view(V, H)  # a differentiable view: the decoder's gradient lands in the word table

    def padded_logits(self, x2d: torch.Tensor, pad_bias: float = 0.0) -> torch.Tensor:
        """fp32 [M, vpad] for bf16 rows [M, H], M a multiple of 8"""
        p = self.params
        return _MLMHeadFn.apply(x2d, p.flat_decay, p.flat_nodecay, self._word(), p, pad_bias)

    def forward(self, hidden: torch.Tensor) -> torch.Tensor:
        H, V = self.params.config.hidden_size, self.params.config.vocab_size
        lead = hidden.shape[:-1]
        x = hidden.reshape(-1, H).to(torch.bfloat16)
        M = x.shape[0]
        Mp = (M + 7) // 8 * 8
        if Mp != M:
            x = torch.nn.functional.pad(x, (0, 0, 0, Mp - M))
        logits = self.padded_logits(x.contiguous())
        return logits[:M, :V].reshape(*lead, V)


class MaskedLMOutput:
    """transformers' MaskedLMOutput, as far as the reference reads it: ``.loss``, ``.hidden_states``, ``.logits`` (computed
    on first access), ``out[0]``."""

    def __init__(self, loss, logits_fn, hidden_states):
        self.loss = loss
        self.hidden_states = hidden_states
        self.attentions = None
        self._logits_fn, self._logits = logits_fn, None

    @property
    def logits(self):
        if self._logits is None:
            self._logits = self._logits_fn()
        return self._logits

    def __getitem__(self, i):
        return tuple(v for v in (self.loss, self.logits, self.hidden_states) if v is not None)[i]


class CocoBertForMaskedLM(nn.Module):
    """hf ``BertForMaskedLM`` on the native path: ``.bert`` (`CocoBertModel`), ``.cls`` (`CocoMLMHead`), hf state-dict names."""

    def __init__(self, config: CocoBertConfig, device=None):
        super().__init__()
        config = CocoBertConfig.coerce(config)
        self.config = config
        self.bert = CocoBertModel(config, device=device)
        self.cls = CocoMLMHead(self.bert)

    # ---------------------------------------------------------------- checkpoints (hf names)
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = OrderedDict() if destination is None else destination
        self.bert.state_dict(destination=sd, prefix=prefix + "bert.", keep_vars=keep_vars)
        for k in [k for k in sd if k.startswith(prefix + "bert.cls.")]:  # MLM tensors a backbone checkpoint carried along
            del sd[k]
        self.cls.params.state_dict(destination=sd, prefix=prefix, keep_vars=keep_vars)
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        r = self.bert.load_state_dict({k: v for k, v in state_dict.items() if not k.startswith("cls.")}, strict=strict)
        head = {k: v for k, v in state_dict.items() if k.startswith("cls.")}
        if "cls.predictions.bias" not in head and "cls.predictions.decoder.bias" in head:  # the same tensor under its tied name
            head["cls.predictions.bias"] = head["cls.predictions.decoder.bias"]
        h = self.cls.params.load_state_dict(head, strict=strict)
        return torch.nn.modules.module._IncompatibleKeys(list(r.missing_keys) + list(h.missing_keys), list(r.unexpected_keys))

    @classmethod
    def from_pretrained(cls, path: str, config: Optional[CocoBertConfig] = None, device=None, **unused):
        config = config or CocoBertConfig.from_pretrained(path)
        model = cls(config, device=device)
        st, pt = os.path.join(path, "model.safetensors"), os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(pt):
            sd = torch.load(pt, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
        model.load_state_dict(sd, strict=False)
        return model

    def save_pretrained(self, path: str) -> None:
        from safetensors.torch import save_file
        self.config.save_pretrained(path)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()
              if k not in ("cls.predictions.decoder.weight", "cls.predictions.decoder.bias")}  # tied: transformers re-ties on load
        save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})

    def resize_token_embeddings(self, n: Optional[int] = None):
        self.bert.resize_token_embeddings(n)
        return self

    def get_extended_attention_mask(self, *a, **kw):
        return self.bert.get_extended_attention_mask(*a, **kw)

    def param_groups(self, weight_decay: float = 0.0):
        return self.bert.param_groups(weight_decay) + self.cls.params.param_groups(weight_decay)

    # ---------------------------------------------------------------- forward
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, labels=None,
                output_hidden_states: bool = False, return_dict: bool = True, **unused):
        out = self.bert(input_ids=input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids,
                        output_hidden_states=output_hidden_states)
        seq = out.last_hidden_state
        loss = None
        if labels is not None:
            loss = self.mlm_loss(seq, labels)
        res = MaskedLMOutput(loss, lambda: self.cls(seq), out.hidden_states)
        if return_dict:
            return res
        return tuple(v for v in (loss, res.logits, out.hidden_states) if v is not None)

    def mlm_loss(self, hidden: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """``CrossEntropyLoss()(cls(hidden).view(-1, V), labels.view(-1))`` (ignore_index -100) on the labelled rows only."""
        H, V = self.config.hidden_size, self.config.vocab_size
        flat = labels.reshape(-1)
        rows = torch.nonzero(flat != -100).squeeze(1)  # host sync: the GEMM row count must be known
        n = int(rows.numel())
        if n == 0:
            raise ValueError("masked-LM loss: no labelled positions in the batch (cross_entropy would be NaN)")
        n_pad = (n + 63) // 64 * 64
        rows_p = torch.cat([rows, rows.new_zeros(n_pad - n)])
        lab = torch.cat([flat[rows].to(torch.int32), torch.zeros(n_pad - n, dtype=torch.int32, device=flat.device)])
        x = hidden.reshape(-1, H).index_select(0, rows_p).to(torch.bfloat16).contiguous()
        logits = self.cls.padded_logits(x, pad_bias=-1e30)
        return _SparseCEFn.apply(logits, lab, n, V)
