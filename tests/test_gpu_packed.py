"""Packed (variable-length) batches: sequences stored back to back, every one on its own length, instead of padded to one length
(include/cocodr.h "Packed batches", SURVEY 7 iii).  The arithmetic per real token is the padded path's, so the packed path is
checked against the padded one (tight tolerance) and against the numpy oracle (the usual bf16 tolerances)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd._native import check, lib, ptr, stream_ptr  # noqa: E402
from cocodr_amd.modeling import BertDotNLL, CoCondenserForPretraining, CocoBertConfig, CocoBertModel, PackedIndex  # noqa: E402
import oracle as O  # noqa: E402  (checker only)

DEV = "cuda"


def rel_l2(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def ragged_batch(B, L, V, seed, lens=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    if lens is None:
        lens = np.clip(np.rint(rng.normal(0.6 * L, 0.25 * L, B)), 3, L).astype(np.int64)
        lens[0] = L
    ids = rng.integers(5, V, (B, L))
    mask = (np.arange(L)[None] < np.asarray(lens)[:, None]).astype(np.int64)
    return ids * mask, mask, np.asarray(lens)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def cfg_small(**kw):
    d = dict(vocab_size=700, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256, max_position_embeddings=512,
             hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    d.update(kw)
    return d


def build(cfgd, P):
    m = CocoBertModel(CocoBertConfig(**cfgd))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    return m.to(DEV)


def test_packed_index_layout():
    ids, mask, lens = ragged_batch(5, 64, 700, 1, lens=[64, 1, 33, 32, 7])
    pk = PackedIndex.build(t(ids).int(), t(mask).int())
    # every sequence on its own length; 137 rows -> 160: the 23 rows that make T a multiple of 32 go to the last sequence
    assert pk.seq_off.cpu().tolist() == [0, 64, 65, 98, 130, 160] and pk.T == 160 and pk.max_len == 64
    pos = pk.positions.cpu().numpy()
    assert pos[:64].tolist() == list(range(64)) and pos[64] == 0 and pos[65:98].tolist() == list(range(33)) and pos[130:160].tolist() == list(range(30))
    m = pk.mask.cpu().numpy()
    assert m[:130].all() and m[130:137].all() and not m[137:160].any()
    assert pk.cls_slot.cpu().numpy()[[0, 64, 65, 98, 130]].tolist() == [0, 1, 2, 3, 4] and (pk.cls_slot.cpu().numpy() >= 0).sum() == 5


@pytest.mark.parametrize("L,heads", [(64, 2), (128, 4), (256, 2), (384, 2)])
def test_packed_attention_matches_per_sequence_padded_attention(L, heads):
    B, H = 5, heads * 64
    _, mask, lens = ragged_batch(B, L, 100, L)
    pk = PackedIndex.build(t(mask).int(), t(mask).int())
    g = torch.Generator().manual_seed(L)
    qkv_pad = (torch.randn(B * L, 3 * H, generator=g)).to(torch.bfloat16).to(DEV)
    dctx_pad = (torch.randn(B * L, H, generator=g)).to(torch.bfloat16).to(DEV)
    qkv = qkv_pad[pk.src].contiguous()
    dctx = dctx_pad[pk.src].contiguous()
    ctx = torch.empty((pk.T, H), dtype=torch.bfloat16, device=DEV)
    lse = torch.empty((heads, pk.T), dtype=torch.float32, device=DEV)
    check(lib().cocodr_attn_fwd_packed(ptr(qkv), ptr(pk.mask), ptr(ctx), ptr(lse), ptr(pk.seq_off), ptr(pk.seq_order), B, pk.T, pk.max_len, heads, None, L,
                                       stream_ptr()), "attn_fwd_packed")
    rctx, rlse = ops.attn_fwd(qkv_pad, t(mask).int(), B, L, heads)
    real = pk.mask.bool()
    assert torch.equal(ctx[real], rctx[pk.src][real])  # same arithmetic per real token: masked key blocks contribute exact zeros
    rl = rlse.permute(1, 0, 2).reshape(heads, B * L)[:, pk.src]
    assert torch.allclose(lse[:, real], rl[:, real], atol=1e-6, rtol=1e-6)
    dqkv = torch.empty_like(qkv)
    part = torch.empty((4 * B, 2 * H), dtype=torch.float32, device=DEV)
    check(lib().cocodr_attn_bwd_packed(ptr(qkv), ptr(pk.mask), ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), ptr(part), ptr(pk.seq_off), None if L == 128 else ptr(pk.seq_order), B, pk.T,
                                       pk.max_len, heads, None, L, stream_ptr()), "attn_bwd_packed")
    # the padded reference needs zero upstream gradient on its padding rows (the packed layout does not have them)
    dpad = torch.zeros_like(dctx_pad)
    dpad[pk.src] = dctx
    rd, rpart = ops.attn_bwd(qkv_pad, t(mask).int(), rctx, dpad, rlse, B, L, heads, qk_bias=True)
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        assert rel_l2(dqkv[real][:, sl], rd[pk.src][real][:, sl]) < 2e-3, name
    assert rel_l2(part.view(B, 4, 2 * H).sum(1), rpart.view(B, 4, 2 * H).sum(1)) < 2e-3


@pytest.mark.parametrize("L", [64, 128])
def test_packed_forward_equals_padded_forward(L):
    cfgd = cfg_small()
    P = O.make_params(O.OracleConfig(**{k: v for k, v in cfgd.items() if "dropout" not in k}), 5, std=0.08)
    m = build(cfgd, P).eval()
    ids, mask, lens = ragged_batch(9, L, cfgd["vocab_size"], 3)
    with torch.no_grad():
        m.pack_sequences = False
        ref = m(input_ids=t(ids), attention_mask=t(mask), output_hidden_states=True)
        m.pack_sequences = True
        got = m(input_ids=t(ids), attention_mask=t(mask), output_hidden_states=True)
    assert torch.equal(got.cls_fp32, ref.cls_fp32)
    valid = t(mask).bool()
    for a, b in zip(got.hidden_states, ref.hidden_states):
        assert torch.equal(a[valid], b[valid])
    # odd lengths / L not a multiple of 32 (the wrapper pads to 32 first) and a batch that cannot be packed
    ids2, mask2, _ = ragged_batch(4, 50, cfgd["vocab_size"], 4)
    with torch.no_grad():
        got2 = m(input_ids=t(ids2), attention_mask=t(mask2))
        m.pack_sequences = False
        ref2 = m(input_ids=t(ids2), attention_mask=t(mask2))
    assert got2.last_hidden_state.shape == ref2.last_hidden_state.shape == (4, 50, cfgd["hidden_size"])
    assert torch.equal(got2.cls_fp32, ref2.cls_fp32)


def test_packed_training_step_matches_padded_step_and_oracle():
    cfgd = cfg_small()
    ocfg = O.OracleConfig(**{k: v for k, v in cfgd.items() if "dropout" not in k})
    P = O.make_params(ocfg, 6, std=0.08)
    ids, mask, lens = ragged_batch(8, 64, cfgd["vocab_size"], 7)
    res = {}
    for packed in (False, True):
        m = build(cfgd, P)
        m.pack_sequences = packed
        model = CoCondenserForPretraining(m)
        loss = model({"input_ids": t(ids), "attention_mask": t(mask)}, None)
        loss.backward()
        res[packed] = (float(loss), {k: v.detach().float().cpu().numpy() for k, v in m.hf_named_grads()})
    assert abs(res[True][0] - res[False][0]) < 1e-5 * abs(res[False][0])
    for name, ref in res[False][1].items():
        if name.endswith("key.bias"):
            continue
        assert rel_l2(res[True][1][name], ref) < 5e-3, (name, rel_l2(res[True][1][name], ref))
    hs, cache = O.encoder_fwd(P, ocfg, ids, mask, keep_cache=True)
    ref_loss, dE = O.contrastive_loss_grad(O.cls_embedding(hs[-1]).copy(), 1)
    assert abs(res[True][0] - ref_loss) < 2e-2 * abs(ref_loss)  # raw [CLS] logits O(100) at H = 128: the padded path sits at the same 1 % here
    d_last = np.zeros_like(hs[-1])
    d_last[:, 0] = dE
    G = O.encoder_bwd(P, ocfg, cache, d_last)
    for name in ("encoder.layer.0.attention.self.query.weight", "encoder.layer.2.output.dense.weight", "embeddings.position_embeddings.weight",
                 "embeddings.LayerNorm.weight", "encoder.layer.1.intermediate.dense.bias", "embeddings.token_type_embeddings.weight"):
        assert rel_l2(res[True][1][name], G[name]) < 8e-2, (name, rel_l2(res[True][1][name], G[name]))
    rows = np.unique(ids[mask.astype(bool)])
    assert rel_l2(res[True][1]["embeddings.word_embeddings.weight"][rows], G["embeddings.word_embeddings.weight"][rows]) < 8e-2


def test_packed_ance_step_and_attention_dropout_draw_the_padded_masks():
    """BertDotNLL on packed batches; with dropout on the attention probabilities only, the packed run reproduces the padded
    run (the probability masks are indexed on the padded length in both layouts)."""
    cfgd = cfg_small(num_hidden_layers=2, attention_probs_dropout_prob=0.2)
    B = 4
    q = ragged_batch(B, 32, cfgd["vocab_size"], 1)
    a = ragged_batch(B, 64, cfgd["vocab_size"], 2)
    b = ragged_batch(B, 64, cfgd["vocab_size"], 3)
    res = {}
    for packed in (False, True):
        torch.manual_seed(0)
        model = BertDotNLL(CocoBertConfig(**cfgd)).to(DEV).train()
        model.bert.dropout_seed = 11
        model.bert.pack_sequences = packed
        model.merge_passes = False  # the reference's pass structure (a query pass and a passage pass), packed or padded
        loss, _acc, logits = model(t(q[0]), t(q[1]), t(a[0]), t(a[1]), t(b[0]), t(b[1]))
        loss.backward()
        res[packed] = (float(loss), logits.detach().clone(), model.bert.flat_decay.grad.detach().clone())
    assert abs(res[True][0] - res[False][0]) < 1e-4 * max(1.0, abs(res[False][0]))
    assert torch.allclose(res[True][1], res[False][1], rtol=1e-4, atol=1e-4)
    assert rel_l2(res[True][2], res[False][2]) < 5e-3


def test_merged_single_pass_triplet_step_equals_the_separate_passes():
    """`merge_passes`: queries [B, 32], positives and negatives [B, 64] as ONE packed encoder pass - same embeddings (each
    sequence attends to itself only), same loss, same gradients as the query pass + passage pass; with dropout it still trains
    (one call number for the whole step)."""
    cfgd = cfg_small(num_hidden_layers=2)
    B = 5
    q = ragged_batch(B, 32, cfgd["vocab_size"], 11)
    a = ragged_batch(B, 64, cfgd["vocab_size"], 12)
    b = ragged_batch(B, 64, cfgd["vocab_size"], 13)
    res = {}
    for mode in ("padded", "packed", "merged"):
        torch.manual_seed(0)
        model = BertDotNLL(CocoBertConfig(**cfgd)).to(DEV).train()
        with torch.no_grad():
            model.bert.flat_decay.mul_(2.0)
        model.bert.pack_sequences = mode != "padded"
        model.merge_passes = mode == "merged"
        loss, _acc, logits = model(t(q[0]), t(q[1]), t(a[0]), t(a[1]), t(b[0]), t(b[1]))
        loss.backward()
        res[mode] = (float(loss), logits.detach().clone(), model.bert.flat_decay.grad.detach().clone(), list(model.last_passes))
    assert [p[0] for p in res["merged"][3]] == ["qab"] and [p[0] for p in res["packed"][3]] == ["q", "ab"]
    for ref in ("padded", "packed"):
        assert abs(res["merged"][0] - res[ref][0]) < 1e-4 * max(1.0, abs(res[ref][0]))
        assert torch.allclose(res["merged"][1], res[ref][1], rtol=1e-4, atol=1e-4)
        assert rel_l2(res["merged"][2], res[ref][2]) < 5e-3
    # masks with a hole cannot be packed: the step falls back to the two passes
    model = BertDotNLL(CocoBertConfig(**cfgd)).to(DEV).train()
    model.bert.pack_sequences = model.merge_passes = True
    bad = q[1].copy()
    bad[0, 2] = 0
    loss, _acc, _l = model(t(q[0]), t(bad), t(a[0]), t(a[1]), t(b[0]), t(b[1]))
    assert torch.isfinite(loss) and [p[0] for p in model.last_passes] == ["q", "ab"]
    # dropout on: one pass, one call number, finite loss and gradients
    model = BertDotNLL(CocoBertConfig(**cfg_small(num_hidden_layers=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1))).to(DEV).train()
    model.bert.pack_sequences = model.merge_passes = True
    model.bert.dropout_seed = 3
    loss, _acc, _l = model(t(q[0]), t(q[1]), t(a[0]), t(a[1]), t(b[0]), t(b[1]))
    loss.backward()
    assert model.last_passes == [("qab", 1)] and torch.isfinite(loss) and torch.isfinite(model.bert.flat_decay.grad).all()


@pytest.mark.parametrize("skip_from,late", [(1, True), (2, False), (0, True), (3, True)])
def test_packed_condenser_step_equals_padded_step(skip_from, late):
    """The full coCondenser step (COCO/modeling.py:192-235: backbone + Condenser head + head / late MLM losses + contrastive) on
    the packed layout against the same step on the padded layout: same losses, same gradients of backbone and head."""
    import types
    cfgd = cfg_small()
    ids, mask, lens = ragged_batch(8, 64, cfgd["vocab_size"], 21)
    rng = np.random.Generator(np.random.PCG64(3))
    pick = (rng.random(ids.shape) < 0.2) & (mask > 0)
    pick[:, 0] = False
    pick[0, 1] = True
    labels = np.where(pick, ids, -100)
    inp = np.where(pick, 103, ids)
    res = {}
    for mode in ("padded", "packed", "packed_host_lengths"):
        torch.manual_seed(0)
        bert = CocoBertModel(CocoBertConfig(**cfgd)).to(DEV)
        with torch.no_grad():
            bert.flat_decay.mul_(2.0)
        model = CoCondenserForPretraining(bert, types.SimpleNamespace(n_head_layers=2, skip_from=skip_from, late_mlm=late)).to(DEV).eval()
        bert.pack_sequences = mode != "padded"
        batch = {"input_ids": t(inp), "attention_mask": t(mask)}
        if mode == "packed_host_lengths":
            batch["lengths"] = torch.from_numpy(lens)
        loss = model(batch, t(labels))
        loss.backward()
        res[mode] = (float(loss.detach()), [p.grad.detach().clone() for p in (bert.flat_decay, bert.flat_nodecay, model.c_head.flat_decay, model.c_head.flat_nodecay)])
    for mode in ("packed", "packed_host_lengths"):
        assert abs(res[mode][0] - res["padded"][0]) < 1e-4 * abs(res["padded"][0]), (mode, res[mode][0], res["padded"][0])
        for g, r in zip(res[mode][1], res["padded"][1]):
            assert rel_l2(g, r) < 6e-3, (mode, rel_l2(g, r))


# (the 200-sequence row: ~17.9 k stored rows = 280 tiles of 256 x 256 at N = 1024 - the forward / dgrad GEMMs with K >= 2048 run
#  with their last partial round cut into contraction slices, gemm_pp.hip launch_split - against 400 whole tiles in the padded run)
@pytest.mark.parametrize("H,heads,I,B", [(768, 12, 3072, 32), (1024, 16, 4096, 32), (1024, 16, 4096, 200)])
def test_packed_step_equals_padded_step_at_full_width(H, heads, I, B):
    """Packed against padded at BERT-base / BERT-large WIDTH (two layers, 32 x 128 tokens, MS MARCO-shaped lengths): forward
    bit-identical at the real tokens (while both layouts take the same GEMM route), loss within 1e-5, every parameter gradient within 5e-3."""
    cfgd = cfg_small(hidden_size=H, num_attention_heads=heads, intermediate_size=I, num_hidden_layers=2, vocab_size=3000)
    ids, mask, lens = ragged_batch(B, 128, 3000, 77)
    res, cls = {}, {}
    for packed in (False, True):
        torch.manual_seed(0)
        m = CocoBertModel(CocoBertConfig(**cfgd)).to(DEV)
        with torch.no_grad():
            s_ln = float(np.sqrt(5.0 / H))  # [CLS] logits O(5): a conditioned InfoNCE (as tests/test_gpu_large_shapes.py)
            for k in ("weight", "bias"):
                m.hf_view(f"encoder.layer.1.output.LayerNorm.{k}").mul_(s_ln)
            m.flat_nodecay.add_(0.02)
        m.pack_sequences = packed
        model = CoCondenserForPretraining(m)
        batch = {"input_ids": t(ids), "attention_mask": t(mask)}
        if packed:
            batch["lengths"] = torch.from_numpy(lens)
        with torch.no_grad():
            cls[packed] = m.encode_cls(t(ids), t(mask)).clone()
        loss = model(batch, None)
        loss.backward()
        res[packed] = (float(loss.detach()), {k: v.detach().clone() for k, v in m.hf_named_grads()})
    if B <= 32:
        assert torch.equal(cls[True], cls[False])
    else:  # the packed row count puts the K = 4096 GEMM in the split-tail window (its last tiles sum K in slices): rounding-level
        assert rel_l2(cls[True], cls[False]) < 2e-3, rel_l2(cls[True], cls[False])
    assert abs(res[True][0] - res[False][0]) < 1e-5 * abs(res[False][0])
    for name, ref in res[False][1].items():
        if name.endswith("key.bias") or float(ref.norm()) == 0:
            continue
        # B = 200: different K-summation order in the split-tail tiles = different bf16 roundings of the hidden states; both steps
        # sit 2-3 % from the fp64 oracle's gradients and <= 1.7 % from each other (profiles/r04_split_tail_noise.md)
        assert rel_l2(res[True][1][name], ref) < (5e-3 if B <= 32 else 2.5e-2), (name, rel_l2(res[True][1][name], ref))


@pytest.mark.parametrize("B,L,seed", [(2, 32, 0), (6, 50, 1), (10, 100, 2), (14, 128, 3), (4, 200, 4), (2, 512, 5), (26, 64, 6), (8, 33, 7)])
def test_packed_default_step_equals_padded_step_over_odd_batch_shapes(B, L, seed):
    """The default (packed) step against the padded one over the shapes a data loader really produces: batch sizes that are not
    multiples of 8 (last batch of an epoch), lengths that are not multiples of 32, sequences of one token, every sequence full,
    L up to max_position_embeddings.  Forward bit-identical at the real tokens, loss and gradients equal."""
    cfgd = cfg_small(num_hidden_layers=2)
    rng = np.random.Generator(np.random.PCG64(100 + seed))
    lens = rng.integers(1, L + 1, B)
    lens[0] = 1
    lens[-1] = L
    if seed == 3:
        lens[:] = L                     # nothing to drop
    ids, mask, lens = ragged_batch(B, L, cfgd["vocab_size"], seed, lens=lens)
    res, cls = {}, {}
    for packed in (False, True):
        torch.manual_seed(0)
        m = CocoBertModel(CocoBertConfig(**cfgd)).to(DEV)
        with torch.no_grad():
            m.flat_decay.mul_(2.0)
            m.flat_nodecay.add_(0.02)
            for k in ("weight", "bias"):
                m.hf_view(f"encoder.layer.1.output.LayerNorm.{k}").mul_(0.2)   # a conditioned InfoNCE
        m.pack_sequences = packed
        model = CoCondenserForPretraining(m)
        with torch.no_grad():
            cls[packed] = m.encode_cls(t(ids), t(mask)).clone()
            hs = m(input_ids=t(ids), attention_mask=t(mask), output_hidden_states=True).hidden_states
            cls[("hs", packed)] = [h.clone() for h in hs]
        batch = {"input_ids": t(ids), "attention_mask": t(mask)}
        if packed and seed % 2 == 0:
            batch["lengths"] = torch.from_numpy(lens)
        loss = model(batch, None)
        loss.backward()
        res[packed] = (float(loss.detach()), {k: v.detach().clone() for k, v in m.hf_named_grads()})
    assert torch.equal(cls[True], cls[False])
    valid = t(mask).bool()
    for hp, hd in zip(cls[("hs", True)], cls[("hs", False)]):
        assert hp.shape == hd.shape and torch.equal(hp[valid], hd[valid])
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * abs(res[False][0]), (res[True][0], res[False][0])
    for name, ref in res[False][1].items():
        if name.endswith("key.bias") or float(ref.norm()) == 0:
            continue
        assert rel_l2(res[True][1][name], ref) < 6e-3, (name, rel_l2(res[True][1][name], ref))


@pytest.mark.parametrize("B,Lq,Lp,seed", [(1, 32, 64, 0), (3, 20, 100, 1), (8, 64, 128, 2), (11, 64, 50, 3), (2, 64, 512, 4)])
def test_default_triplet_step_equals_the_padded_two_pass_step_over_odd_batch_shapes(B, Lq, Lp, seed):
    """The ANCE triplet step (ANCE/model/models.py:80-115) as it runs by default - one merged packed pass - against the reference's
    structure (query pass + passage pass, padded) for batch sizes and lengths a loader really produces, one-token sequences included."""
    cfgd = cfg_small(num_hidden_layers=2)
    rng = np.random.Generator(np.random.PCG64(200 + seed))
    def side(L, s):
        lens = rng.integers(1, L + 1, B)
        lens[0] = 1 if s != 1 else L
        return ragged_batch(B, L, cfgd["vocab_size"], 10 * seed + s, lens=lens)
    q, a, b = side(Lq, 0), side(Lp, 1), side(Lp, 2)
    res = {}
    for mode in ("padded", "default"):
        torch.manual_seed(0)
        model = BertDotNLL(CocoBertConfig(**cfgd)).to(DEV).train()
        with torch.no_grad():
            model.bert.flat_decay.mul_(2.0)
        if mode == "padded":
            model.bert.pack_sequences = model.merge_passes = False
        else:
            assert model.bert.pack_sequences and model.merge_passes
        loss, _acc, logits = model(t(q[0]), t(q[1]), t(a[0]), t(a[1]), t(b[0]), t(b[1]))
        loss.backward()
        res[mode] = (float(loss.detach()), logits.detach().clone(), model.bert.flat_decay.grad.detach().clone(), model.bert.flat_nodecay.grad.detach().clone())
    assert abs(res["default"][0] - res["padded"][0]) < 1e-4 * max(1.0, abs(res["padded"][0]))
    assert torch.allclose(res["default"][1], res["padded"][1], rtol=1e-4, atol=1e-4)
    assert rel_l2(res["default"][2], res["padded"][2]) < 6e-3 and rel_l2(res["default"][3], res["padded"][3]) < 6e-3


@pytest.mark.parametrize("L,heads", [(128, 2), (256, 2), (384, 1)])
def test_packed_attention_of_a_sequence_does_not_depend_on_its_neighbours(L, heads):
    """Sequences are stored on exactly their lengths, so the last 32-row block of one sequence overlaps the first rows of the next:
    the kernels must neither use those rows nor write them.  Changing every OTHER sequence's q / k / v / dctx rows (to huge values
    and NaN-free garbage) leaves a sequence's context, log-sum-exp and gradients bit-identical - forward, one-pass backward
    (L <= 128), two-phase backward (L <= 256) and the two-kernel backward (L > 256)."""
    B, H = 5, heads * 64
    rng = np.random.Generator(np.random.PCG64(L))
    lens = np.array([L, 1, 33, L - 5, 17])
    ids, mask, lens = ragged_batch(B, L, 100, L, lens=lens)
    pk = PackedIndex.build(t(mask).int(), t(mask).int())
    off = pk.seq_off.cpu().numpy()
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(pk.T, 3 * H, generator=g).to(torch.bfloat16).to(DEV)
    dctx = torch.randn(pk.T, H, generator=g).to(torch.bfloat16).to(DEV)

    def run(qkv_, dctx_):
        ctx = torch.full((pk.T, H), 7.0, dtype=torch.bfloat16, device=DEV)
        lse = torch.full((heads, pk.T), 7.0, dtype=torch.float32, device=DEV)
        check(lib().cocodr_attn_fwd_packed(ptr(qkv_), ptr(pk.mask), ptr(ctx), ptr(lse), ptr(pk.seq_off), ptr(pk.seq_order), B, pk.T, pk.max_len, heads,
                                           None, L, stream_ptr()), "attn_fwd_packed")
        dqkv = torch.full_like(qkv_, 7.0)
        part = torch.empty((4 * B, 2 * H), dtype=torch.float32, device=DEV)
        check(lib().cocodr_attn_bwd_packed(ptr(qkv_), ptr(pk.mask), ptr(ctx), ptr(dctx_), ptr(lse), ptr(dqkv), ptr(part), ptr(pk.seq_off),
                                           ptr(pk.seq_order), B, pk.T, pk.max_len, heads, None, L, stream_ptr()), "attn_bwd_packed")
        return ctx, lse, dqkv

    ctx0, lse0, dq0 = run(qkv, dctx)
    assert not (ctx0 == 7.0).all(dim=1).any() and not (dq0 == 7.0).all(dim=1).any()   # every stored row was written
    for b in range(B):
        q2, d2 = qkv.clone(), dctx.clone()
        other = torch.ones(pk.T, dtype=torch.bool, device=DEV)
        other[off[b]:off[b + 1]] = False
        q2[other] = (torch.randn(int(other.sum()), 3 * H, generator=g) * 300.0).to(torch.bfloat16).to(DEV)
        d2[other] = (torch.randn(int(other.sum()), H, generator=g) * 300.0).to(torch.bfloat16).to(DEV)
        ctx1, lse1, dq1 = run(q2, d2)
        rows = slice(int(off[b]), int(off[b + 1]))
        assert torch.equal(ctx1[rows], ctx0[rows]) and torch.equal(lse1[:, rows], lse0[:, rows]) and torch.equal(dq1[rows], dq0[rows]), b
