"""`resize_token_embeddings` (COCO/run_coco_pre_training.py:158 calls it on `model.lm` with len(tokenizer)): the flat parameter
layout is rebuilt around a word table of the new size.  Host logic only, CPU tensors; pinned against transformers' own
BertForMaskedLM.resize_token_embeddings for what is kept, what is padded and what a saved checkpoint looks like."""
import types

import numpy as np
import pytest
import torch

import cocodr_amd  # noqa: F401
from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel


def _cfg(**kw):
    d = dict(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
             max_position_embeddings=48, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    d.update(kw)
    return CocoBertConfig(**d)


@pytest.mark.parametrize("new", [307, 300, 256, 1024])
def test_resize_keeps_rows_and_every_other_tensor(new):
    torch.manual_seed(0)
    m = CocoBertModel(_cfg())
    with torch.no_grad():
        m.flat_nodecay.normal_()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    assert m.resize_token_embeddings(new) is m
    after = m.state_dict()
    assert m.config.vocab_size == new and after["embeddings.word_embeddings.weight"].shape == (new, 128)
    keep = min(new, 300)
    assert torch.equal(after["embeddings.word_embeddings.weight"][:keep], before["embeddings.word_embeddings.weight"][:keep])
    for k, v in before.items():
        if k != "embeddings.word_embeddings.weight":
            assert torch.equal(after[k], v), k
    if new > 300:  # new rows: drawn like _init_weights draws an embedding
        fresh = after["embeddings.word_embeddings.weight"][300:]
        assert torch.isfinite(fresh).all() and (fresh != 0).any()
        if new - 300 >= 512:
            assert abs(float(fresh.std()) - m.config.initializer_range) < 0.15 * m.config.initializer_range
    # the native side derives the position / type tables from the word table's end: the layout must stay contiguous
    lo = m.layout
    assert lo.names["embeddings.position_embeddings.weight"][1] == new * 128
    assert lo.decay_numel == m.flat_decay.numel() and lo.mat_begin % 64 == 0
    assert m.resize_token_embeddings(None) is m and m.config.vocab_size == new


def test_resize_matches_transformers_on_kept_rows_bias_padding_and_checkpoint(tmp_path):
    transformers = pytest.importorskip("transformers")
    hf_cfg = transformers.BertConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                     max_position_embeddings=48)
    torch.manual_seed(1)
    hf = transformers.BertForMaskedLM(hf_cfg)
    with torch.no_grad():
        hf.cls.predictions.bias.normal_()
    d0 = tmp_path / "hf"
    hf.save_pretrained(str(d0))
    args = types.SimpleNamespace(n_head_layers=2, skip_from=1, late_mlm=False)
    model = CoCondenserForPretraining.from_pretrained(args, None, None, str(d0))
    old_bias = model.c_head.hf_view("cls.predictions.bias").clone()
    old_dense = model.c_head.hf_view("cls.predictions.transform.dense.weight").clone()
    old_head_q = model.c_head.hf_view("c_head.1.attention.self.query.bias").clone()
    model.lm.resize_token_embeddings(311)  # the reference's call site: model.lm.resize_token_embeddings(len(tokenizer))
    hf.resize_token_embeddings(311, mean_resizing=False)
    w, w_hf = model.lm.hf_view("embeddings.word_embeddings.weight"), hf.bert.embeddings.word_embeddings.weight.detach()
    assert w.shape == w_hf.shape == (311, 128) and torch.equal(w[:300], w_hf[:300])
    b, b_hf = model.c_head.hf_view("cls.predictions.bias"), hf.cls.predictions.bias.detach()
    assert b.shape == b_hf.shape == (311,) and torch.equal(b, b_hf) and torch.equal(b[:300], old_bias) and not b[300:].any()
    assert torch.equal(model.c_head.hf_view("cls.predictions.transform.dense.weight"), old_dense)
    assert torch.equal(model.c_head.hf_view("c_head.1.attention.self.query.bias"), old_head_q)
    assert model.c_head.vpad == 384
    # the resized model's checkpoint loads into transformers with the new vocabulary
    d1 = tmp_path / "native"
    model.save_pretrained(str(d1))
    back = transformers.BertForMaskedLM.from_pretrained(str(d1))
    assert back.config.vocab_size == 311
    assert torch.equal(back.bert.embeddings.word_embeddings.weight.detach(), w)
    assert torch.equal(back.cls.predictions.bias.detach(), b)
    assert back.cls.predictions.decoder.weight.data_ptr() == back.bert.embeddings.word_embeddings.weight.data_ptr()  # still tied
    # shrinking cuts both
    model.lm.resize_token_embeddings(290)
    assert model.lm.hf_view("embeddings.word_embeddings.weight").shape == (290, 128)
    assert torch.equal(model.c_head.hf_view("cls.predictions.bias"), old_bias[:290])


def test_resize_rejects_bad_sizes_and_late_calls():
    m = CocoBertModel(_cfg())
    with pytest.raises(ValueError):
        m.resize_token_embeddings(0)
    m._dp_hooks = [object()]  # as after enable_grad_allreduce
    with pytest.raises(RuntimeError):
        m.resize_token_embeddings(400)


def test_transformers_config_objects_are_accepted():
    """The reference passes `config=AutoConfig.from_pretrained(...)` into `from_pretrained` (COCO/modeling.py:100-101,
    ANCE/drivers/run_ann.py:889-901): a transformers BertConfig (or a dict) becomes a validated CocoBertConfig."""
    transformers = pytest.importorskip("transformers")
    from cocodr_amd.modeling import BertDotNLL
    hf = transformers.BertConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                 max_position_embeddings=48, hidden_dropout_prob=0.05)
    for m in (CocoBertModel(hf), BertDotNLL(hf).bert, CocoBertModel(hf.to_dict())):
        assert isinstance(m.config, CocoBertConfig)
        assert (m.config.vocab_size, m.config.hidden_size, m.config.num_hidden_layers, m.config.hidden_dropout_prob) == (300, 128, 2, 0.05)
    bad = transformers.BertConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)
    with pytest.raises(ValueError):
        CocoBertModel(bad)  # head_dim 32: the validation still applies
    with pytest.raises(ValueError):
        CocoBertModel(transformers.BertConfig(vocab_size=300, hidden_size=128, num_attention_heads=2, intermediate_size=256,
                                              hidden_act="relu"))
