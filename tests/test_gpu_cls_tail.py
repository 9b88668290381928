"""[CLS] tail (cocodr_config.cls_tail, what encode_cls() runs): the last layer's output projection, LayerNorms and FFN on the [CLS]
rows only.  Every reference wrapper of the contrastive / ANCE / inference paths consumes hidden_states[-1][:, 0] alone
(COCO/modeling.py:199-204, ANCE/model/models.py:225-232), so the values must be those of the full layer: the [CLS] rows bit for bit
(per-row arithmetic is unchanged), the gradients up to the summation order of the top layer's three small weight gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402,F401
from cocodr_amd.modeling import CocoBertConfig, CocoBertModel  # noqa: E402

DEV = "cuda"


def rel_l2(a, b):
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def batch(B, L, V, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.clip(np.rint(rng.normal(0.6 * L, 0.25 * L, B)), 3, L).astype(np.int64)
    lens[0] = L
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    ids = rng.integers(5, V, (B, L)) * mask
    return torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)


def model(layers=3, H=256, heads=4, I=512, V=900, **kw):
    torch.manual_seed(0)
    cfg = dict(vocab_size=V, hidden_size=H, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=I,
               max_position_embeddings=512, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg.update(kw)
    m = CocoBertModel(CocoBertConfig(**cfg)).to(DEV)
    with torch.no_grad():  # biases / LayerNorm parameters away from their init values: their gradients must be exercised
        m.flat_nodecay.add_(0.05 * torch.randn_like(m.flat_nodecay))
    return m


def step(m, ids, mask, packed, tail):
    m.cls_tail = tail
    m.pack_sequences = packed  # (packed is the model's default: the padded runs must say so)
    m.zero_grad(set_to_none=True)
    pk = m.pack(ids, mask) if packed else None
    cls = m.encode_cls(ids, mask, packed_index=pk)
    w = torch.linspace(-1.0, 1.0, cls.numel(), device=DEV).view_as(cls)
    (cls * w).sum().backward()
    return cls.detach().clone(), m.flat_decay.grad.clone(), m.flat_nodecay.grad.clone()


# (the last two rows: BERT-base and BERT-large WIDTH, two layers - VERDICT r03 item 9)
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("B,L,layers,wide", [(8, 64, 3, None), (64, 128, 2, None), (5, 96, 1, None), (16, 128, 2, (768, 12, 3072)),
                                             (16, 128, 2, (1024, 16, 4096))])
def test_cls_tail_equals_the_full_last_layer(packed, B, L, layers, wide):
    m = model(layers=layers) if wide is None else model(layers=layers, H=wide[0], heads=wide[1], I=wide[2])
    ids, mask = batch(B, L, 900, 3)
    cls_f, gd_f, gn_f = step(m, ids, mask, packed, False)
    cls_t, gd_t, gn_t = step(m, ids, mask, packed, True)
    assert torch.equal(cls_t, cls_f)  # the same arithmetic on the same rows
    assert rel_l2(gd_t, gd_f) < 2e-3 and rel_l2(gn_t, gn_f) < 2e-3
    lo = m.layout
    for name, (which, off, shape) in lo.names.items():  # per tensor: nothing of the top layer (or below it) is missed
        n = int(np.prod(shape))
        a, b = ((gd_t, gd_f) if which == 0 else (gn_t, gn_f))
        a, b = a[off:off + n], b[off:off + n]
        if float(b.norm()) > 0:
            assert rel_l2(a, b) < 1e-2, name
        else:
            assert float(a.norm()) == 0.0, name
    with torch.no_grad():  # inference: same rows, lean arena
        m.cls_tail = True
        m.pack_sequences = packed
        e_t = m.encode_cls(ids, mask, packed_index=m.pack(ids, mask) if packed else None)
        m.cls_tail = False
        e_f = m.encode_cls(ids, mask, packed_index=m.pack(ids, mask) if packed else None)
    assert torch.equal(e_t, e_f) and torch.equal(e_t, cls_f)


def test_cls_tail_output_has_no_hidden_states_and_forward_is_unchanged():
    m = model()
    ids, mask = batch(4, 64, 900, 5)
    with torch.no_grad():
        out = m(ids, mask, cls_only=True)
        assert out.last_hidden_state is None and out.hidden_states is None and out.cls_fp32.shape == (4, 256)
        full = m(ids, mask, output_hidden_states=True, cls_only=True)  # hidden states asked for: the full layer runs
        assert full.last_hidden_state.shape == (4, 64, 256) and len(full.hidden_states) == 4
        assert torch.equal(full.cls_fp32, out.cls_fp32)


def test_cls_tail_steps_aside_under_dropout():
    m = model(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    m.train()
    ids, mask = batch(4, 64, 900, 7)
    cls = m.encode_cls(ids, mask)  # masks are indexed by token row: the full layer runs
    cls.sum().backward()
    assert torch.isfinite(m.flat_decay.grad).all() and float(m.flat_decay.grad.norm()) > 0
    m.eval()
    with torch.no_grad():
        a = m.encode_cls(ids, mask)
        m.cls_tail = False
        assert torch.equal(a, m.encode_cls(ids, mask))
