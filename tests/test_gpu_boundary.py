"""Boundary papercuts of the HuggingFace-style forward: a tokenizer's all-zero ``token_type_ids`` on the device do
not stop the host (README.md:101-116 passes them on every call); non-zero ones are still refused - a call late, or at check_inputs();
the stack of hidden states travels with the call that produced it, not on the module."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402,F401
from cocodr_amd.modeling import CocoBertConfig, CocoBertModel  # noqa: E402

DEV = "cuda"


def small_model():
    cfg = CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=500, hidden_size=128, num_hidden_layers=2,
                         num_attention_heads=2, intermediate_size=512, max_position_embeddings=64)
    torch.manual_seed(0)
    m = CocoBertModel(cfg).to(DEV)
    with torch.no_grad():
        m.flat_decay.normal_(0, 0.02)
    return m.eval()


def batch(seed=0, B=4, L=32):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(5, 500, (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[1, 20:] = 0
    return ids.to(DEV), mask.to(DEV)


def test_device_token_type_zeros_are_accepted_without_a_host_sync_and_nonzero_is_refused_late():
    m = small_model()
    ids, mask = batch()
    with torch.no_grad():
        a = m(input_ids=ids, attention_mask=mask).last_hidden_state
        b = m(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids)).last_hidden_state
        m.check_inputs()
        assert torch.equal(a, b)
        m(input_ids=ids, attention_mask=mask, token_type_ids=torch.ones_like(ids))   # accepted now ...
        with pytest.raises(NotImplementedError, match="token_type_ids"):
            m.check_inputs()                                                             # ... refused here (or by the next forward)
        m.check_inputs()  # the refusal cleared the queue
        with pytest.raises(NotImplementedError, match="token_type_ids"):  # a host tensor is checked on the spot
            m(input_ids=ids, attention_mask=mask, token_type_ids=torch.ones(ids.shape, dtype=torch.int64))


def test_hidden_states_come_from_the_call_not_from_the_module():
    m = small_model()
    ids, mask = batch(1)
    with torch.no_grad():
        for packed in (True, False):
            m.pack_sequences = packed
            out = m(input_ids=ids, attention_mask=mask, output_hidden_states=True)
            assert len(out.hidden_states) == 3 and out.hidden_states[-1].shape == (4, 32, 128)
            assert torch.equal(out.hidden_states[-1], out.last_hidden_state)
    assert not hasattr(m, "_last_hidden_states")
