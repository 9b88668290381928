"""``named_parameters()`` / ``parameters()`` yield HF-named per-tensor ``nn.Parameter``s (views of the flat
storage, ``.grad`` = views of the flat gradient), so the reference's own per-tensor code runs unchanged on the model:

  * HF Trainer's name-based weight-decay grouping + ``torch.optim.AdamW``      COCO/trainer.py:66-70 -> Trainer.create_optimizer
  * a per-tensor LAMB over ``model.parameters()`` (one trust ratio per tensor)  ANCE/utils/lamb.py:71-121, run_ann.py:128-147
  * ``iDROLoss._params`` name filter (``layer.9`` ...)                           ANCE/model/dro_loss.py:174-190
  * ``clip_grad_norm_(model.parameters())``                                      ANCE/drivers/run_ann.py:345-347

Everything here runs on CPU (parameters and gradients only; the forward is native and needs the GPU): the flat gradient is
filled by hand, exactly where the native backward writes it."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

import cocodr_amd  # noqa: F401
from cocodr_amd.modeling import BertDotNLL, CocoBertConfig, CocoBertModel
from oracle import optim_oracle as OO  # checker


def _cfg(layers=3):
    return CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, hidden_size=128,
                          num_hidden_layers=layers, num_attention_heads=2, intermediate_size=256, max_position_embeddings=64)


def _fill_grads(m, seed):
    g = torch.Generator().manual_seed(seed)
    for p in m.flat_parameters():
        p.grad = torch.randn(p.shape, generator=g) * 0.01


def test_named_parameters_are_hf_named_views_of_the_flats():
    torch.manual_seed(0)
    m = CocoBertModel(_cfg())
    named = dict(m.named_parameters())
    assert sorted(named) == sorted(m.layout.names)  # (module-tree order, as transformers' BertModel yields them)
    assert "encoder.layer.2.attention.self.query.weight" in named and "embeddings.LayerNorm.bias" in named
    assert all(isinstance(p, nn.Parameter) and p.requires_grad and p.is_leaf for p in named.values())
    # each weight exactly once; the flats themselves are not registered parameters
    assert sum(p.numel() for p in m.parameters()) == sum(int(np.prod(s)) for _, _, s in m.layout.names.values())
    assert "flat_decay" not in named and "flat_nodecay" not in named
    # views alias the flat storage both ways
    q = named["encoder.layer.1.attention.self.query.weight"]
    off = m.layout.names["encoder.layer.1.attention.self.query.weight"][1]
    with torch.no_grad():
        q[3, 5] = 7.0
    assert float(m.flat_decay.data[off + 3 * 128 + 5]) == 7.0
    with torch.no_grad():
        m.flat_decay.data[off] = -2.0
    assert float(q.detach()[0, 0]) == -2.0
    # .grad: None without a flat gradient, a view of it otherwise
    assert q.grad is None
    _fill_grads(m, 1)
    assert q.grad.data_ptr() == m.flat_decay.grad.data_ptr() + 4 * off and tuple(q.grad.shape) == (128, 128)
    # the module tree is the HF one: LayerNorm shells are nn.LayerNorm (what HF's decay grouping tests), dense shells nn.Linear
    mods = dict(m.named_modules())
    assert isinstance(mods["encoder.layer.0.output.LayerNorm"], nn.LayerNorm)
    assert isinstance(mods["encoder.layer.0.intermediate.dense"], nn.Linear)
    assert isinstance(mods["embeddings.word_embeddings"], nn.Embedding)
    # wrapper modules prefix the names like the reference class does (ANCE/model/models.py:226 ``self.bert``)
    w = BertDotNLL(_cfg())
    assert "bert.encoder.layer.0.output.dense.bias" in dict(w.named_parameters())


def test_reference_idro_name_filter_selects_the_last_layers():
    """ANCE/model/dro_loss.py:174-190 restated: parameters whose name contains ``layer.{9,10,11}`` (base)."""
    w = BertDotNLL(_cfg(layers=12))
    picked = [n for n, p in w.named_parameters() if any(f"layer.{i}." in n for i in (9, 10, 11))]
    assert len(picked) == 3 * 16 and all(n.startswith("bert.encoder.layer.") for n in picked)
    lo = w.bert.layout
    lo_off = lo.mat_begin + 9 * lo.mat_stride
    assert sum(p.numel() for n, p in w.named_parameters() if n in picked and p._which == 0) == lo.decay_numel - lo_off


def _hf_decay_groups(model, weight_decay):
    """transformers.Trainer.create_optimizer's grouping, restated: decay = every parameter that does not live in an
    nn.LayerNorm module and whose name does not contain 'bias' (trainer_pt_utils.get_parameter_names)."""
    def names(mod, prefix=""):
        out = []
        for n, child in mod.named_children():
            if not isinstance(child, nn.LayerNorm):
                out += names(child, prefix + n + ".")
        out += [prefix + n for n in mod._parameters]
        return out
    decay = [n for n in names(model) if "bias" not in n]
    return [{"params": [p for n, p in model.named_parameters() if n in decay], "weight_decay": weight_decay},
            {"params": [p for n, p in model.named_parameters() if n not in decay], "weight_decay": 0.0}]


def test_name_based_adamw_over_views_equals_adamw_over_the_flats():
    torch.manual_seed(0)
    a, b = CocoBertModel(_cfg()), CocoBertModel(_cfg())
    b.load_state_dict(a.state_dict())
    groups = _hf_decay_groups(a, 0.01)
    # the name-based grouping lands exactly on the flat split: decay = flat_decay's tensors, no-decay = flat_nodecay's
    assert {p._which for p in groups[0]["params"]} == {0} and {p._which for p in groups[1]["params"]} == {1}
    assert len(groups[0]["params"]) + len(groups[1]["params"]) == len(a.layout.names)
    opt_a = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    opt_b = torch.optim.AdamW(b.param_groups(0.01), lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    for step in range(3):
        _fill_grads(a, 10 + step)
        _fill_grads(b, 10 + step)
        v0 = a._params_version()
        torch.nn.utils.clip_grad_norm_(a.parameters(), 0.05)
        torch.nn.utils.clip_grad_norm_(b.flat_parameters(), 0.05)
        opt_a.step()
        opt_b.step()
        assert a._params_version() != v0  # the bf16 weight shadow sees per-tensor updates
        opt_a.zero_grad()  # set_to_none: every view cleared -> the flat gradient is dropped
        opt_b.zero_grad()
        assert a.flat_decay.grad is None and a.flat_nodecay.grad is None
    # padding between tensors is never touched by the per-tensor optimizer; everything that is a parameter agrees
    for name in a.layout.names:
        assert torch.allclose(a.hf_view(name), b.hf_view(name), rtol=1e-6, atol=1e-8), name


class _RefLamb(torch.optim.Optimizer):
    """per-tensor LAMB written the way ANCE/utils/lamb.py:71-121 is: a Python loop over ``group['params']`` reading ``p.grad``
    and updating ``p.data`` in place - what runs when the reference driver builds ``Lamb(model.parameters())``."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad.data
                st = self.state[p]
                if not st:
                    st["m"], st["v"] = torch.zeros_like(p.data), torch.zeros_like(p.data)
                b1, b2 = group["betas"]
                st["m"].mul_(b1).add_(grad, alpha=1 - b1)
                st["v"].mul_(b2).addcmul_(grad, grad, value=1 - b2)
                upd = st["m"] / (st["v"].sqrt() + group["eps"])
                if group["weight_decay"]:
                    upd = upd + group["weight_decay"] * p.data
                wn, un = p.data.pow(2).sum().sqrt().clamp(0, 10), upd.pow(2).sum().sqrt()
                trust = 1.0 if (wn == 0 or un == 0) else float(wn / un)
                p.data.add_(upd, alpha=-group["lr"] * trust)


def test_per_tensor_lamb_over_parameters_matches_the_oracle_per_tensor():
    torch.manual_seed(1)
    m = CocoBertModel(_cfg(layers=2))
    names = [n for n, _ in m.named_parameters()]
    P = [p.detach().double().numpy().copy() for p in m.parameters()]
    M = [np.zeros_like(x) for x in P]
    V = [np.zeros_like(x) for x in P]
    opt = _RefLamb(m.parameters(), lr=1e-3, weight_decay=0.01)
    trust_seen = None
    for step in range(2):
        _fill_grads(m, 20 + step)
        G = [p.grad.detach().double().numpy().copy() for p in m.parameters()]
        m.__dict__["_views_dirty"] = False
        opt.step()
        assert m._shadow_stale()  # the reference's `p.data.add_` bypasses version counters: handing out .data marks the shadow stale
        trust_seen = OO.lamb_step(P, G, M, V, lr=1e-3, weight_decay=0.01)
        m.zero_grad()
        assert m.flat_decay.grad is None
    assert len(set(np.round(trust_seen, 6))) > 10  # one trust ratio per TENSOR (two flats would give two)
    for n, p, want in zip(names, m.parameters(), P):
        assert np.allclose(p.detach().numpy(), want, rtol=2e-5, atol=1e-7), n


def test_partial_zeroing_and_assignment_of_view_gradients():
    m = CocoBertModel(_cfg(layers=2))
    _fill_grads(m, 3)
    named = dict(m.named_parameters())
    q, k = named["encoder.layer.0.attention.self.query.weight"], named["encoder.layer.0.attention.self.key.weight"]
    keep = k.grad.clone()
    q.grad = None  # one tensor only
    assert q.grad is None and m.flat_decay.grad is not None and torch.equal(k.grad, keep)
    q.grad = torch.ones_like(q)  # assignment writes through
    off = m.layout.names["encoder.layer.0.attention.self.query.weight"][1]
    assert float(m.flat_decay.grad[off]) == 1.0 and torch.equal(q.grad, torch.ones_like(q))
    # zero_grad(set_to_none=False) through an optimizer zeroes the flat in place
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    opt.zero_grad(set_to_none=False)
    assert float(m.flat_decay.grad.abs().sum()) == 0.0 and float(m.flat_nodecay.grad.abs().sum()) == 0.0


def test_views_follow_to_resize_and_deepcopy():
    m = CocoBertModel(_cfg(layers=2))
    m2 = m.to(torch.float32)  # _apply: flats converted in place, views rebuilt
    assert m2 is m and dict(m.named_parameters())["embeddings.LayerNorm.weight"].data_ptr() == m.flat_nodecay.data_ptr()
    m.resize_token_embeddings(320)
    assert dict(m.named_parameters())["embeddings.word_embeddings.weight"].shape == (320, 128)
    c = copy.deepcopy(m)
    assert [n for n, _ in c.named_parameters()] == [n for n, _ in m.named_parameters()]
    with torch.no_grad():
        dict(c.named_parameters())["embeddings.LayerNorm.weight"].fill_(3.0)
    assert float(c.flat_nodecay.data[0]) == 3.0 and float(m.flat_nodecay.data[0]) == 1.0
    m.requires_grad_(False)
    assert not m.flat_decay.requires_grad and not any(p.requires_grad for p in m.parameters())
