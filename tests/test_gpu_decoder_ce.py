"""cocodr_decoder_ce (include/cocodr.h): the MLM decoder GEMM fused with the vocabulary cross entropy - the reference's
``lm.cls`` decoder + ``CrossEntropyLoss`` (COCO/modeling.py:87-93, :221-227) without the fp32 [n, V] logits in HBM.  Checked against
a torch fp32 restatement on the same bf16 operands, and against the two-kernel path (GEMM to fp32 logits + cocodr_ce_fwd_bwd)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd import ops  # noqa: E402
from cocodr_amd._native import check, lib, ptr, stream_ptr  # noqa: E402

DEV = "cuda"
NEG = -1e30


def problem(n, H, V, seed, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    vpad = (V + 255) // 256 * 256
    t = (torch.randn(n, H, generator=g) * spread).to(torch.bfloat16)
    W = torch.zeros(vpad, H, dtype=torch.bfloat16)
    W[:V] = (torch.randn(V, H, generator=g) * 0.2).to(torch.bfloat16)
    bias = torch.full((vpad,), NEG)
    bias[:V] = torch.randn(V, generator=g) * 0.5
    labels = torch.randint(0, V, (n,), generator=g, dtype=torch.int32)
    labels[0], labels[-1] = V - 1, 0            # last real column / first column
    scale = torch.rand(n, generator=g) / n
    scale[n // 2] = 0.0                          # a padding row (condenser.py pads the labelled rows to 64)
    return [x.to(DEV) for x in (t, W, bias, labels, scale)], vpad


def torch_reference(t, W, bias, labels, scale, V):
    logits = t.float() @ W.float().T + bias
    logits = logits[:, :V].double()
    lse = torch.logsumexp(logits, 1)
    loss = lse - logits.gather(1, labels.long()[:, None])[:, 0]
    p = torch.exp(logits - lse[:, None])
    p[torch.arange(len(labels)), labels.long()] -= 1.0
    return loss.float(), (p * scale.double()[:, None]).float()


@pytest.mark.parametrize("n,H,V", [(64, 128, 900), (192, 768, 30522), (1216, 768, 30522), (320, 1024, 30522), (64, 64, 256), (37, 128, 1000)])
def test_fused_decoder_ce_matches_fp32_reference_and_the_two_kernel_path(n, H, V):
    (t, W, bias, labels, scale), vpad = problem(n, H, V, seed=n + V)
    loss, dlog = ops.decoder_ce(t, W, bias, labels, scale)
    torch.cuda.synchronize()
    ref_loss, ref_d = torch_reference(t, W, bias, labels, scale, V)
    assert torch.allclose(loss, ref_loss, rtol=2e-5, atol=2e-5), float((loss - ref_loss).abs().max())
    d = dlog.float()
    assert not d[:, V:].any()                   # padding columns: probability exactly 0
    assert not d[n // 2].any()                  # scale 0 row: exactly 0
    err = (d[:, :V] - ref_d).abs().max() / ref_d.abs().max()
    assert float(err) < 6e-3, float(err)        # bf16 rounding of the stored gradient
    # the two-kernel path on the same operands
    logits = ops.gemm(t, W, bias=bias, out_f32=True)
    loss2 = torch.empty(n, dtype=torch.float32, device=DEV)
    dlog2 = torch.empty((n, vpad), dtype=torch.bfloat16, device=DEV)
    check(lib().cocodr_ce_fwd_bwd(ptr(logits), ptr(labels), ptr(scale), n, V, vpad, ptr(loss2), ptr(dlog2), stream_ptr()), "ce_fwd_bwd")
    assert torch.allclose(loss, loss2, rtol=2e-5, atol=2e-5)
    both = (dlog.float() - dlog2.float()).abs().max() / dlog2.float().abs().max()
    assert float(both) < 8e-3, float(both)      # (one bf16 ulp of the largest entries: exp evaluated in a different order)


def test_fused_decoder_ce_is_stable_for_large_logits_and_deterministic():
    (t, W, bias, labels, scale), vpad = problem(128, 256, 5000, seed=5, spread=12.0)   # logits of several hundred
    loss, dlog = ops.decoder_ce(t, W, bias, labels, scale)
    ref_loss, ref_d = torch_reference(t, W, bias, labels, scale, 5000)
    assert torch.isfinite(loss).all() and torch.isfinite(dlog.float()).all()
    assert torch.allclose(loss, ref_loss, rtol=1e-4, atol=1e-3)
    loss_b, dlog_b = ops.decoder_ce(t, W, bias, labels, scale)
    assert torch.equal(loss, loss_b) and torch.equal(dlog, dlog_b)


def test_decoder_ce_argument_errors():
    (t, W, bias, labels, scale), vpad = problem(64, 128, 900, seed=1)
    with pytest.raises(ValueError):
        ops.decoder_ce(t, W[:900].contiguous(), bias[:900].contiguous(), labels, scale)     # vpad % 256
    with pytest.raises(ValueError):
        ops.decoder_ce(t, W, bias, labels.long(), scale)
    with pytest.raises(ValueError):
        ops.decoder_ce(t.float(), W, bias, labels, scale)
    # the epilogue forms are refused outside the 256 x 256-tile NT pipeline
    g = N.GemmArgs()
    out = torch.empty((64, vpad), dtype=torch.bfloat16, device=DEV)
    g.A, g.B, g.C, g.bias = ptr(t), ptr(W), ptr(out), ptr(bias)
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.batch = 64, vpad, 128, 128, 128, vpad, 1
    g.epi = N.EPI_CE_GRAD
    import ctypes as C
    assert lib().cocodr_gemm(C.byref(g), stream_ptr()) != 0     # no row_label / row_lse


@pytest.mark.parametrize("late", [True, False])
def test_condenser_step_with_the_fused_decoder_equals_the_two_kernel_step(late):
    """The full coCondenser step (COCO/modeling.py:192-235) with ``CondenserHead.fused_ce`` on and off: same MLM + contrastive loss,
    same gradients of backbone (incl. the tied word table) and head."""
    import types
    from cocodr_amd.condenser import CondenserHead
    from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel
    cfgd = dict(vocab_size=700, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, max_position_embeddings=512,
                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    rng = np.random.Generator(np.random.PCG64(3))
    lens = rng.integers(5, 65, 8)
    mask = (np.arange(64)[None] < lens[:, None]).astype(np.int64)
    ids = rng.integers(5, 700, (8, 64)) * mask
    pick = (rng.random(ids.shape) < 0.2) & (mask > 0)
    pick[:, 0] = False
    pick[0, 1] = True
    labels, inp = np.where(pick, ids, -100), np.where(pick, 103, ids)
    res = {}
    try:
        for fused in (False, True):
            CondenserHead.fused_ce = fused
            torch.manual_seed(0)
            bert = CocoBertModel(CocoBertConfig(**cfgd)).to(DEV)
            with torch.no_grad():
                bert.flat_decay.mul_(2.0)
            model = CoCondenserForPretraining(bert, types.SimpleNamespace(n_head_layers=2, skip_from=1, late_mlm=late)).to(DEV).eval()
            batch = {"input_ids": torch.from_numpy(inp).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
            loss = model(batch, torch.from_numpy(labels).to(DEV))
            loss.backward()
            res[fused] = (float(loss.detach()), [p.grad.detach().clone() for p in (bert.flat_decay, bert.flat_nodecay, model.c_head.flat_decay, model.c_head.flat_nodecay)])
    finally:
        CondenserHead.fused_ce = False
    assert abs(res[True][0] - res[False][0]) < 1e-5 * abs(res[False][0])
    for g, r in zip(res[True][1], res[False][1]):
        d = float((g - r).norm() / r.norm())
        assert d < 5e-3, d
