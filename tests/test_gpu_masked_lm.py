"""`CocoBertForMaskedLM`: the drop-in for the object `AutoModelForMaskedLM.from_pretrained` hands to the reference
(COCO/modeling.py:96-108), against transformers' own BertForMaskedLM (fp32, eager attention, CPU) loaded from the same
checkpoint - `.loss`, `.hidden_states`, `.logits`, `lm.cls(hiddens)` and the gradients, including the way the reference's
Condenser wrapper uses them (COCO/modeling.py:199-224: hidden_states[skip_from] and the last layer into `lm.cls`, two MLM
losses).  Tolerances: bf16 activations against fp32 - hidden states rel-L2 2e-2, loss 1e-2 relative, gradients 8e-2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402
from cocodr_amd.masked_lm import CocoBertForMaskedLM  # noqa: E402

DEV = "cuda"


def rel_l2(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _setup(tmp_path, vocab=500):
    transformers = pytest.importorskip("transformers")
    cfg = transformers.BertConfig(vocab_size=vocab, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                                  max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                                  attn_implementation="eager")
    torch.manual_seed(11)
    hf = transformers.BertForMaskedLM(cfg).eval()
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if "LayerNorm" not in n:
                p.mul_(3.0)  # the default 0.02 init gives nearly input-independent hidden states
        hf.cls.predictions.bias.normal_(0.0, 0.3)
    d = tmp_path / "hf"
    hf.save_pretrained(str(d))
    m = CocoBertForMaskedLM.from_pretrained(str(d)).to(DEV)
    rng = np.random.Generator(np.random.PCG64(12))
    B, L = 6, 40
    ids = torch.from_numpy(rng.integers(5, vocab, (B, L)))
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[1, 25:] = 0
    mask[4, 9:] = 0
    ids = ids * mask
    labels = torch.full((B, L), -100, dtype=torch.int64)
    pick = (torch.from_numpy(rng.random((B, L))) < 0.2) & (mask > 0)
    pick[:, 0] = False
    labels[pick] = ids[pick]
    return transformers, hf, m, ids, mask, labels


def test_masked_lm_outputs_and_gradients_match_transformers(tmp_path):
    transformers, hf, m, ids, mask, labels = _setup(tmp_path)
    ref = hf(input_ids=ids, attention_mask=mask, labels=labels, output_hidden_states=True, return_dict=True)
    ref.loss.backward()
    out = m(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), output_hidden_states=True, return_dict=True)
    out.loss.backward()
    assert abs(float(out.loss.detach()) - float(ref.loss.detach())) < 1e-2 * abs(float(ref.loss.detach()))
    valid = mask.bool()
    assert len(out.hidden_states) == len(ref.hidden_states) == 4
    for a, b in zip(out.hidden_states, ref.hidden_states):
        assert rel_l2(a.float().cpu()[valid], b[valid]) < 2e-2
    logits = out.logits
    assert logits.shape == ref.logits.shape == (6, 40, 500) and logits.dtype == torch.float32
    assert rel_l2(logits.cpu()[valid], ref.logits[valid]) < 3e-2
    G = {"bert." + k: v for k, v in m.bert.hf_named_grads()}
    G.update(dict(m.cls.params.hf_named_grads()))
    R = {k: p.grad for k, p in hf.named_parameters() if p.grad is not None}
    rows = torch.unique(torch.cat([ids[valid], labels[labels >= 0]]))
    for name in ("cls.predictions.bias", "cls.predictions.transform.dense.weight", "cls.predictions.transform.dense.bias",
                 "cls.predictions.transform.LayerNorm.weight", "bert.encoder.layer.0.attention.self.query.weight",
                 "bert.encoder.layer.2.output.dense.weight", "bert.embeddings.position_embeddings.weight",
                 "bert.embeddings.word_embeddings.weight"):
        assert rel_l2(G[name], R[name]) < 8e-2, (name, rel_l2(G[name], R[name]))
    # the tied decoder: word rows that only occur as (wrong or right) predictions still receive gradient through the logits
    gw = G["bert.embeddings.word_embeddings.weight"].cpu()
    unused = torch.ones(500, dtype=torch.bool)
    unused[rows] = False
    assert float(gw[unused].abs().sum()) > 0 and rel_l2(gw[unused], R["bert.embeddings.word_embeddings.weight"][unused]) < 8e-2
    # tuple form and the checkpoint written back
    tup = m(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), return_dict=False)
    assert len(tup) == 2 and abs(float(tup[0].detach()) - float(out.loss.detach())) < 1e-6 and tup[1].shape == (6, 40, 500)
    d2 = tmp_path / "native"
    m.save_pretrained(str(d2))
    back = transformers.BertForMaskedLM.from_pretrained(str(d2), attn_implementation="eager").eval()
    for (n1, p1), (n2, p2) in zip(sorted(hf.state_dict().items()), sorted(back.state_dict().items())):
        assert n1 == n2 and torch.allclose(p1, p2, rtol=0, atol=0), n1


def test_condenser_style_use_of_hidden_states_and_cls(tmp_path):
    """COCO/modeling.py:199-224 with the c_head layers left out: cat([CLS] of the last layer, hidden_states[skip_from] without
    its first token) -> lm.cls -> cross entropy, plus lm's own MLM loss; the gradient reaches the backbone through
    hidden_states[skip_from], the last layer and the tied word table."""
    _, hf, m, ids, mask, labels = _setup(tmp_path, vocab=384)  # a vocabulary that needs no padding columns
    skip_from = 1

    def step(lm, ids, mask, labels):
        out = lm(input_ids=ids, attention_mask=mask, labels=labels, output_hidden_states=True, return_dict=True)
        cls_hiddens = out.hidden_states[-1][:, :1]
        skip_hiddens = out.hidden_states[skip_from]
        hiddens = torch.cat([cls_hiddens, skip_hiddens[:, 1:]], dim=1)
        scores = lm.cls(hiddens)
        loss = torch.nn.functional.cross_entropy(scores.float().view(-1, scores.shape[-1]), labels.view(-1))
        return loss + out.loss

    ref = step(hf, ids, mask, labels)
    ref.backward()
    got = step(m, ids.to(DEV), mask.to(DEV), labels.to(DEV))
    got.backward()
    assert abs(float(got.detach()) - float(ref.detach())) < 1e-2 * abs(float(ref.detach()))
    G = {"bert." + k: v for k, v in m.bert.hf_named_grads()}
    G.update(dict(m.cls.params.hf_named_grads()))
    R = {k: p.grad for k, p in hf.named_parameters() if p.grad is not None}
    for name in ("cls.predictions.bias", "cls.predictions.transform.dense.weight", "bert.encoder.layer.0.output.dense.weight",
                 "bert.encoder.layer.1.attention.self.value.weight", "bert.encoder.layer.2.intermediate.dense.weight",
                 "bert.embeddings.word_embeddings.weight", "bert.embeddings.LayerNorm.weight"):
        assert rel_l2(G[name], R[name]) < 8e-2, (name, rel_l2(G[name], R[name]))


def test_resize_and_eval_mode(tmp_path):
    _, hf, m, ids, mask, labels = _setup(tmp_path)
    m.resize_token_embeddings(517)
    assert m.config.vocab_size == 517 and m.cls.params.hf_view("cls.predictions.bias").shape == (517,)
    with torch.no_grad():
        out = m(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV))
        assert out.logits.shape == (6, 40, 517) and torch.isfinite(out.loss)
        ref = hf(input_ids=ids, attention_mask=mask, labels=labels)
    # the new columns only add to the partition function: with a zero-padded bias and N(0, 0.02) rows the loss barely moves
    assert abs(float(out.loss) - float(ref.loss)) < 0.15 * abs(float(ref.loss))
    with pytest.raises(ValueError):
        m(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=torch.full_like(labels, -100).to(DEV))
