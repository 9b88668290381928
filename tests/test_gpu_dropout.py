"""Dropout on the native path (hf nn.Dropout sites under model.train(), ANCE/drivers/run_ann.py:293).
Kernel level: every fused dropout site against a plain fp32 torch evaluation that multiplies with the ORACLE's mask of the
same keys - the positions must agree exactly (a dropped element is an exact zero contribution), the values within the
bf16 tolerances of the un-dropped kernels.  Model level: train()-mode forward + backward against the numpy oracle driven
with the same (seed, call) - the oracle's placement is pinned to the reference in tests/test_dropout_cpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd.modeling import BertDotNLL, CoCondenserForPretraining, CocoBertConfig, CocoBertModel  # noqa: E402
import oracle as O  # noqa: E402  (checker only)
from oracle import dropout_oracle as D  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def mult(shape, p, seed, call, layer, kind):
    return torch.from_numpy(D.multiplier(tuple(shape), p, seed, call, layer, kind)).to(DEV)


def make_mask(B, L, seed=0):
    g = np.random.Generator(np.random.PCG64(seed))
    m = np.zeros((B, L), np.int32)
    for b in range(B):
        m[b, : (L if b == 0 else int(g.integers(3, L + 1)))] = 1
    return torch.from_numpy(m).to(DEV)


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("impl", [0, 1, 3, 9, 13])
@pytest.mark.parametrize("M,Nn,K", [(512, 256, 128), (1000, 1024, 256)])
def test_gemm_residual_epilogue_drops_the_dense_output(M, Nn, K, impl):
    ops.gemm_set_impl(impl)
    try:
        a, w, bias, r = rnd(M, K, seed=1), rnd(Nn, K, scale=0.05, seed=2), rnd(Nn, seed=3, dtype=torch.float32), rnd(M, Nn, seed=4)
        p, key = 0.1, (11, 5, 3, ops.KIND_ATTN_OUT)
        dm = ops.dropout_mask(p, *key)
        out = ops.gemm(a, w, bias=bias, epi=N.EPI_ADD, r=r, drop=dm)
    finally:
        ops.gemm_set_impl(0)
    m = mult((M, Nn), p, *key)
    dense = a.float() @ w.float().T + bias
    ref = dense * m + r.float()
    assert rel_l2(out, ref) < 5e-3
    dropped = m == 0
    assert 0.08 < float(dropped.float().mean()) < 0.12
    assert torch.equal(out[dropped], r[dropped])  # a dropped element leaves exactly the residual
    # threshold 0 is the plain epilogue
    plain = ops.gemm(a, w, bias=bias, epi=N.EPI_ADD, r=r)
    assert torch.equal(ops.gemm(a, w, bias=bias, epi=N.EPI_ADD, r=r, drop=ops.dropout_mask(0.0, *key)), plain)
    with pytest.raises(ValueError):
        ops.gemm(a, w, bias=bias, drop=dm)  # dropout belongs to the residual epilogue


@pytest.mark.parametrize("B,L,H", [(3, 32, 128), (2, 64, 768), (2, 32, 1024)])
def test_embedding_dropout_fwd_bwd(B, L, H):
    V = 300
    ids = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(1)).to(torch.int32).to(DEV)
    word, pos, type0 = rnd(V, H, seed=2, dtype=torch.float32), rnd(64, H, seed=3, dtype=torch.float32), rnd(H, seed=4, dtype=torch.float32)
    g, b = 1 + 0.1 * rnd(H, seed=5, dtype=torch.float32), rnd(H, seed=6, dtype=torch.float32)
    p, key = 0.1, (5, 2, 0, ops.KIND_EMBED)
    dm = ops.dropout_mask(p, *key)
    out, mean, rstd = ops.embed_ln_fwd(ids, word, pos, type0, g, b, drop=dm)
    plain, mean0, rstd0 = ops.embed_ln_fwd(ids, word, pos, type0, g, b)
    m = mult((B * L, H), p, *key)
    assert torch.equal(mean, mean0) and torch.equal(rstd, rstd0)  # statistics are the LayerNorm's, dropout sits behind it
    assert torch.equal(out[m == 0], torch.zeros_like(out[m == 0]))
    ref = torch.nn.functional.layer_norm(word[ids.long()] + pos[:L][None] + type0, (H,), g, b, 1e-12).reshape(B * L, H) * m
    assert rel_l2(out, ref) < 5e-3
    # backward: the gradient is masked and scaled in front of the LayerNorm backward
    dout = rnd(B * L, H, seed=7)
    got = ops.embed_ln_bwd(dout, ids, word, pos, type0, g, mean, rstd, drop=dm)
    want = ops.embed_ln_bwd((dout.float() * m).to(torch.bfloat16), ids, word, pos, type0, g, mean, rstd)
    for a_, b_ in zip(got, want):
        assert rel_l2(a_, b_) < 1e-2  # one extra bf16 rounding on the reference side


@pytest.mark.parametrize("M,H", [(64, 128), (1000, 768), (515, 1024), (300, 512)])
def test_layernorm_backward_with_dropped_dense_output(M, H):
    y, g = rnd(M, H, seed=15), 1 + 0.1 * rnd(H, seed=16, dtype=torch.float32)
    b = rnd(H, seed=17, dtype=torch.float32)
    _, mean, rstd = ops.ln_fwd(y, g, b)
    dout = rnd(M, H, seed=18)
    p, key = 0.1, (1, 9, 4, ops.KIND_FFN_OUT)
    dm = ops.dropout_mask(p, *key)
    dy, dyd, dg, db, cs = ops.ln_bwd(dout, y, g, mean, rstd, colsum=True, drop=dm)
    dy0, dg0, db0 = ops.ln_bwd(dout, y, g, mean, rstd)
    assert torch.equal(dy, dy0) and torch.equal(dg, dg0) and torch.equal(db, db0)  # the residual branch sees no mask
    m = mult((M, H), p, *key)
    assert torch.equal(dyd[m == 0], torch.zeros_like(dyd[m == 0]))
    yf = y.float().requires_grad_(True)
    torch.nn.functional.layer_norm(yf, (H,), g, b, 1e-12).backward(dout.float())
    ref = yf.grad * m
    assert rel_l2(dyd, ref) < 5e-3
    assert (cs - ref.sum(0)).abs().max() < 1e-3 * max(1.0, float(ref.sum(0).abs().max()))


def ref_attention_drop(qkv, mask, B, L, heads, m):
    H = heads * 64
    q, k, v = [t.reshape(B, L, heads, 64).permute(0, 2, 1, 3) for t in qkv.float().split(H, dim=1)]
    s = (q @ k.transpose(-1, -2)) / 8.0
    s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    lse = torch.logsumexp(s, -1)
    p = torch.softmax(s, -1) * m
    return (p @ v).permute(0, 2, 1, 3).reshape(B * L, H), lse


@pytest.mark.parametrize("B,L,heads", [(2, 32, 2), (3, 64, 2), (2, 128, 12), (2, 256, 4), (2, 288, 2), (2, 512, 2)])
def test_attention_probability_dropout_fwd_bwd(B, L, heads):
    H = heads * 64
    qkv = rnd(B * L, 3 * H, seed=12)
    mask = make_mask(B, L, seed=L)
    p, key = 0.15, (21, 3, 1, ops.KIND_ATTN_PROBS)
    dm = ops.dropout_mask(p, *key)
    m = mult((B, heads, L, L), p, *key)
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, heads, drop=dm)
    _, lse0 = ops.attn_fwd(qkv, mask, B, L, heads)
    assert torch.equal(lse, lse0)  # the normaliser is the un-dropped softmax's
    q = qkv.float().clone().requires_grad_(True)
    rctx, _ = ref_attention_drop(q, mask, B, L, heads, m)
    assert rel_l2(ctx, rctx) < 1.5e-2
    # the mask must be THIS one: the same reference with the mask of another call is far away
    other, _ = ref_attention_drop(qkv, mask, B, L, heads, mult((B, heads, L, L), p, 21, 4, 1, ops.KIND_ATTN_PROBS))
    assert rel_l2(ctx, other) > 0.1
    dctx = rnd(B * L, H, seed=14)
    rctx.backward(dctx.float())
    dqkv, part = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, heads, qk_bias=True, drop=dm)
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        assert rel_l2(dqkv[:, sl], q.grad[:, sl]) < 2.5e-2, name
    part = part.view(B, 4, 2 * H).sum(1)
    want = q.grad[:, :H].view(B, L, H).sum(1)
    assert rel_l2(part[:, :H], want) < 2.5e-2
    assert float(part[:, H:].abs().max()) == 0.0  # rows of dS still sum to zero under dropout: key-bias gradient is 0


def test_attention_dropout_zero_probability_is_the_plain_kernel():
    B, L, heads = 2, 64, 2
    qkv, mask = rnd(B * L, 3 * heads * 64, seed=1), make_mask(B, L, 3)
    c0, l0 = ops.attn_fwd(qkv, mask, B, L, heads)
    c1, l1 = ops.attn_fwd(qkv, mask, B, L, heads, drop=ops.dropout_mask(0.0, 1, 1, 0, 0))
    assert torch.equal(c0, c1) and torch.equal(l0, l1)


# ------------------------------------------------------------------------------------------------ model level
def small_cfg(**kw):
    base = dict(vocab_size=600, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, max_position_embeddings=64)
    base.update(kw)
    return base


def build(cfgd, P, ph, pa):
    cfg = CocoBertConfig(hidden_dropout_prob=ph, attention_probs_dropout_prob=pa, **cfgd)
    m = CocoBertModel(cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    return m.to(DEV)


def batch(B, L, V, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = rng.integers(5, V, (B, L))
    mask = np.ones((B, L), np.int64)
    for b in range(1, B):
        mask[b, int(rng.integers(4, L + 1)):] = 0
    return ids * mask, mask


@pytest.mark.parametrize("ph,pa", [(0.1, 0.1), (0.2, 0.0), (0.0, 0.3)])
def test_train_mode_step_matches_oracle_with_the_same_masks(ph, pa):
    cfgd = small_cfg()
    ocfg = O.OracleConfig(**cfgd)
    P = O.make_params(ocfg, 21, std=0.08)
    m = build(cfgd, P, ph, pa).train()
    m.dropout_seed = 1234
    ids, mask = batch(6, 32, cfgd["vocab_size"], 3)
    t = lambda a: torch.from_numpy(a).to(DEV)
    for call in (1, 2):  # every forward draws fresh masks: the call counter of the keys
        m.zero_grad(set_to_none=True)
        out = m(input_ids=t(ids), attention_mask=t(mask), output_hidden_states=True)
        E = out.cls_fp32
        loss, _rows, dE = ops.simce_fwd_bwd(E.detach().contiguous())
        E.backward(dE)
        drop = dict(p_hidden=ph, p_attn=pa, seed=1234, call=call)
        hs, cache = O.encoder_fwd(P, ocfg, ids, mask, keep_cache=True, dropout=drop)
        valid = mask.astype(bool)
        for i, h in enumerate(out.hidden_states):
            assert rel_l2(h.detach().float().cpu().numpy()[valid], hs[i][valid]) < 2e-2, (call, i)
        ref_loss, rdE = O.contrastive_loss_grad(O.cls_embedding(hs[-1]).copy(), 1)
        assert abs(float(loss) - ref_loss) < 1e-2 * abs(ref_loss)
        d_last = np.zeros_like(hs[-1])
        d_last[:, 0] = rdE
        G = O.encoder_bwd(P, ocfg, cache, d_last)
        got = {k: v.detach().float().cpu().numpy() for k, v in m.hf_named_grads()}
        for name, ref in G.items():
            if name.endswith("key.bias") or name == "embeddings.word_embeddings.weight":
                continue
            assert rel_l2(got[name], ref) < 8e-2, (call, name, rel_l2(got[name], ref))
        rows = np.unique(ids[valid])
        assert rel_l2(got["embeddings.word_embeddings.weight"][rows], G["embeddings.word_embeddings.weight"][rows]) < 8e-2
        if call == 1:
            first = out.cls_fp32.detach().clone()
        else:
            assert rel_l2(out.cls_fp32, first) > 1e-2  # a second forward drops other elements


def test_eval_and_no_grad_do_not_drop_and_seed_reproduces():
    cfgd = small_cfg()
    P = O.make_params(O.OracleConfig(**cfgd), 22, std=0.08)
    ids, mask = batch(4, 32, cfgd["vocab_size"], 5)
    t = lambda a: torch.from_numpy(a).to(DEV)
    m = build(cfgd, P, 0.1, 0.1)
    m0 = build(cfgd, P, 0.0, 0.0)
    ref = m0(input_ids=t(ids), attention_mask=t(mask)).cls_fp32
    assert torch.equal(m.eval()(input_ids=t(ids), attention_mask=t(mask)).cls_fp32, ref)
    with torch.no_grad():  # inference inside a train()-mode module: no saved activations, no dropout
        assert torch.equal(m.train()(input_ids=t(ids), attention_mask=t(mask)).cls_fp32, ref)
    m.dropout_seed = 7
    a = m.train()(input_ids=t(ids), attention_mask=t(mask)).cls_fp32
    assert not torch.equal(a, ref)
    m2 = build(cfgd, P, 0.1, 0.1).train()
    m2.dropout_seed = 7
    assert torch.equal(m2(input_ids=t(ids), attention_mask=t(mask)).cls_fp32, a)  # same seed, same call number -> same masks
    m2.dropout_seed = 8
    assert not torch.equal(m2(input_ids=t(ids), attention_mask=t(mask)).cls_fp32, a)


def test_ance_triplet_train_mode_matches_oracle_pass_by_pass():
    """BertDotNLL under model.train() (ANCE/drivers/run_ann.py:293): the query pass and the passage pass(es) each draw their
    own call number; the oracle is driven with the same per-pass keys."""
    cfgd = small_cfg(num_hidden_layers=2)
    ocfg = O.OracleConfig(**cfgd)
    P = O.make_params(ocfg, 23, std=0.08)
    last = f"encoder.layer.{ocfg.num_hidden_layers - 1}.output.LayerNorm."
    for k in (last + "weight", last + "bias"):
        P[k] = (P[k] * 0.2).astype(np.float32)  # logits O(5), as in the triplet goldens
    model = BertDotNLL(CocoBertConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **cfgd))
    model.merge_passes = model.bert.pack_sequences = False  # pass by pass on the padded layout: the oracle's mask indices
    model.bert.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    model.to(DEV).train()
    model.bert.dropout_seed = 99
    B = 4
    q_ids, q_mask = batch(B, 32, cfgd["vocab_size"], 1)
    a_ids, a_mask = batch(B, 32, cfgd["vocab_size"], 2)
    b_ids, b_mask = batch(B, 32, cfgd["vocab_size"], 3)
    t = lambda a: torch.from_numpy(a).to(DEV)
    loss, _acc, logits = model(t(q_ids), t(q_mask), t(a_ids), t(a_mask), t(b_ids), t(b_mask))
    loss.backward()
    torch.cuda.synchronize()
    # the product's pass structure: which batches went through the encoder together, in call order
    passes = model.last_passes  # list of (name, call) recorded by BertDotNLL.forward
    embs, caches = {}, []
    for name, call in passes:
        ids_, mask_ = {"q": (q_ids, q_mask), "a": (a_ids, a_mask), "b": (b_ids, b_mask),
                       "ab": (np.concatenate([a_ids, b_ids]), np.concatenate([a_mask, b_mask]))}[name]
        hs, cache = O.encoder_fwd(P, ocfg, ids_, mask_, keep_cache=True, dropout=dict(p_hidden=0.1, p_attn=0.1, seed=99, call=call))
        e = O.cls_embedding(hs[-1])
        if name == "ab":
            embs["a"], embs["b"] = e[:B], e[B:]
        else:
            embs[name] = e
        caches.append((name, cache, hs[-1].shape))
    ref_loss, dq, da, db = O.triplet_nll_grad(embs["q"], embs["a"], embs["b"])
    assert abs(float(loss) - ref_loss) < 1e-2 * abs(ref_loss)
    G = {}
    for name, cache, shape in caches:
        d_last = np.zeros(shape, np.float32)
        d_last[:, 0] = {"q": dq, "a": da, "b": db, "ab": np.concatenate([da, db])}[name]
        for k, v in O.encoder_bwd(P, ocfg, cache, d_last).items():
            G[k] = G.get(k, 0) + v
    got = {k: v.detach().float().cpu().numpy() for k, v in model.bert.hf_named_grads()}
    for name in ("encoder.layer.0.attention.self.query.weight", "encoder.layer.0.attention.output.dense.weight",
                 "encoder.layer.1.intermediate.dense.weight", "encoder.layer.1.output.dense.weight", "encoder.layer.0.output.dense.bias",
                 "encoder.layer.0.attention.output.LayerNorm.weight", "embeddings.LayerNorm.weight", "embeddings.position_embeddings.weight"):
        assert rel_l2(got[name], G[name]) < 8e-2, (name, rel_l2(got[name], G[name]))


def test_condenser_head_drops_while_the_backbone_stays_in_eval():
    """COCO/modeling.py:198 puts only ``lm`` in eval: the c_head BertLayers keep their dropout under trainer.py:146's
    model.train().  Checked against the oracle's full step with the head's masks."""
    import types
    from cocodr_amd.condenser import CondenserHead
    cfgd = small_cfg(num_hidden_layers=3)
    ocfg = O.OracleConfig(**cfgd)
    P = O.make_params(ocfg, 31, std=0.08)
    bert = build(cfgd, P, 0.1, 0.1)
    margs = types.SimpleNamespace(n_head_layers=2, skip_from=2, late_mlm=True)
    model = CoCondenserForPretraining(bert, margs).to(DEV).train()
    head = model.c_head
    Ph = {k: v.detach().cpu().numpy().copy() for k, v in head.state_dict().items()}
    head.dropout_seed = 4242
    ids, mask = batch(6, 32, cfgd["vocab_size"], 9)
    rng = np.random.Generator(np.random.PCG64(10))
    pick = (rng.random(ids.shape) < 0.2) & (mask > 0)
    pick[:, 0] = False
    labels = np.where(pick, ids, -100)
    t = lambda a: torch.from_numpy(a).to(DEV)
    loss = model({"input_ids": t(ids), "attention_mask": t(mask)}, t(labels))
    loss.backward()
    assert not bert.training and head.training
    drop = dict(p_hidden=0.1, p_attn=0.1, seed=4242, call=1)
    total, parts, G, Gh = O.condenser_step(P, Ph, ocfg, ids, mask, labels, 2, 2, True, head_dropout=drop)
    assert abs(float(loss) - total) < 1e-2 * abs(total), (float(loss), total, parts)
    Gh0 = O.condenser_step(P, Ph, ocfg, ids, mask, labels, 2, 2, True)[3]
    got = {k: v.detach().float().cpu().numpy() for k, v in head.hf_named_grads()}
    # the head's masks are visible: the same step without them has clearly different head gradients (the loss itself is
    # dominated by log(vocab) at random init and barely moves)
    assert rel_l2(got["c_head.0.attention.self.value.weight"], Gh0["c_head.0.attention.self.value.weight"]) > 0.1
    for name in ("c_head.0.attention.self.value.weight", "c_head.1.output.dense.weight", "c_head.0.attention.output.dense.bias",
                 "c_head.1.output.LayerNorm.weight"):
        assert rel_l2(got[name], Gh[name]) < 8e-2, (name, rel_l2(got[name], Gh[name]))
    gb = {k: v.detach().float().cpu().numpy() for k, v in bert.hf_named_grads()}
    for name in ("encoder.layer.0.attention.self.query.weight", "encoder.layer.1.output.dense.weight", "encoder.layer.2.intermediate.dense.weight"):
        assert rel_l2(gb[name], G[name]) < 8e-2, (name, rel_l2(gb[name], G[name]))


def test_idro_paths_agree_under_dropout():
    """The per-sequence route (one partial backward per pass) and the reference-shaped per-group route re-run backward
    ranges over the same arenas: with dropout both must regenerate the forward's masks."""
    import types
    from cocodr_amd.idro import IDROLoss  # noqa: F401
    cfgd = small_cfg(num_hidden_layers=3)
    B, G = 8, 4
    ids = [batch(B, 32, cfgd["vocab_size"], s) for s in (1, 2, 3)]
    groups = torch.tensor([0, 1, 1, 3, 0, 3, 1, 0], device=DEV)
    t = lambda a: torch.from_numpy(a).to(DEV)
    res = []
    for per_group in (False, True):
        torch.manual_seed(0)
        model = BertDotNLL(CocoBertConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **cfgd)).to(DEV).train()
        model.bert.dropout_seed = 5
        model.add_group_loss(args=types.SimpleNamespace(model_size="base"), n_groups=G, dro_type="idro", alpha=0.25, eps=0.01, ema=0.1, rho=0.05)
        model.loss.per_group_backward = per_group
        robust, _acc, gl, gc = model(t(ids[0][0]), t(ids[0][1]), t(ids[1][0]), t(ids[1][1]), t(ids[2][0]), t(ids[2][1]), group_ids=groups)
        robust.backward()
        res.append((float(robust), model.loss.h_fun.detach().clone(), model.bert.flat_decay.grad.detach().clone()))
    assert abs(res[0][0] - res[1][0]) < 1e-5 * max(1.0, abs(res[0][0]))
    assert torch.allclose(res[0][1], res[1][1], rtol=2e-2, atol=1e-4)
    assert rel_l2(res[0][2], res[1][2]) < 1e-3
