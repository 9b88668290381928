"""Pin the numpy oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np

from conftest import cfg_from_golden, load_golden, params_from_golden
import oracle as O


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _check_grad(name, got, ref):
    if name.endswith("key.bias"):
        # d/d(key bias) is identically zero (softmax is shift invariant along keys): both sides are round-off
        assert np.abs(got).max() < 1e-6 and np.abs(ref).max() < 1e-6
    else:
        assert _rel(got, ref) < 2e-3, (name, _rel(got, ref))


def test_encoder_hidden_states_match_reference(golden_coco):
    g = golden_coco
    cfg = cfg_from_golden(g)
    P = O.make_params(cfg, int(g["seed"]), std=float(g["std"]))
    hs, _ = O.encoder_fwd(P, cfg, g["input_ids"], g["attention_mask"])
    ref = g["hidden_states"]
    assert len(hs) == cfg.num_hidden_layers + 1 == ref.shape[0]
    for i, h in enumerate(hs):
        assert _rel(h, ref[i]) < 1e-5, (i, _rel(h, ref[i]))


def test_co_target_matches_reference(golden_coco, golden_loss):
    assert np.array_equal(O.co_target(len(golden_coco["co_target"])), golden_coco["co_target"])
    for M in (8, 16, 64):
        assert np.array_equal(O.co_target(M), golden_loss[f"target_{M}"])


def test_contrastive_loss_standalone(golden_loss):
    g = golden_loss
    for M in (8, 16, 64):
        E, W = g[f"E_{M}"], int(g[f"W_{M}"])
        rows = O.contrastive_loss(E.copy(), W)
        np.testing.assert_allclose(rows, g[f"rows_{M}"], rtol=2e-5, atol=2e-5)
        loss, dE = O.contrastive_loss_grad(E.copy(), W)
        np.testing.assert_allclose(dE, g[f"dE_{M}"], rtol=1e-4, atol=1e-5)
        assert abs(loss - g[f"rows_{M}"].mean()) < 1e-5


def test_coco_loss_and_grads_through_encoder(golden_coco):
    g = golden_coco
    cfg = cfg_from_golden(g)
    P = O.make_params(cfg, int(g["seed"]), std=float(g["std"]))
    hs, cache = O.encoder_fwd(P, cfg, g["input_ids"], g["attention_mask"], keep_cache=True)
    E = O.cls_embedding(hs[-1])
    for W in (1, 2):
        rows = O.contrastive_loss(E.copy(), W)
        np.testing.assert_allclose(rows, g[f"loss_rows_w{W}"], rtol=1e-4, atol=2e-4)
    loss, dE = O.contrastive_loss_grad(E.copy(), 1)
    assert abs(loss - float(g["loss_w1"])) < 2e-4
    d_last = np.zeros_like(hs[-1])
    d_last[:, 0] = dE
    G = O.encoder_bwd(P, cfg, cache, d_last)
    for key in g.files:
        if key.startswith("grad:"):
            name = key[5:]
            _check_grad(name, G[name], g[key])
    rows = g["grad_rows:embeddings.word_embeddings.weight"]
    assert _rel(G["embeddings.word_embeddings.weight"][:64], rows) < 2e-3
    names = [str(n) for n in g["gradsum_names"]]
    for n, (s, a) in zip(names, g["gradsums"]):
        if n not in G or n.endswith("key.bias"):
            continue  # cls.* / pooler params are outside the encoder; key.bias grads are pure round-off
        assert abs(np.abs(G[n]).sum(dtype=np.float64) - a) <= 2e-3 * a + 1e-7, n


def test_ance_triplet_matches_reference(golden_ance):
    g = golden_ance
    cfg = cfg_from_golden(g)
    P = params_from_golden(g)
    embs, caches = [], []
    for ids, mask in ((g["q_ids"], g["q_mask"]), (g["a_ids"], g["a_mask"]), (g["b_ids"], g["b_mask"])):
        hs, cache = O.encoder_fwd(P, cfg, ids, mask, keep_cache=True)
        embs.append(O.cls_embedding(hs[-1]))
        caches.append((cache, hs[-1].shape))
    for e, k in zip(embs, ("q_emb", "a_emb", "b_emb")):
        assert _rel(e, g[k]) < 1e-5
    rows, logits = O.triplet_nll(*embs)
    np.testing.assert_allclose(logits, g["logits"], rtol=1e-5)
    assert np.array_equal(np.argmax(logits, 1), g["acc"])
    loss, dq, da, db = O.triplet_nll_grad(*embs, weights=g["weights"])
    assert abs(loss - float(g["loss"])) < 1e-4
    G = {}
    for (cache, shape), de in zip(caches, (dq, da, db)):
        d_last = np.zeros(shape, np.float32)
        d_last[:, 0] = de
        for k, v in O.encoder_bwd(P, cfg, cache, d_last).items():
            G[k] = G.get(k, 0) + v
    for key in g.files:
        if key.startswith("grad:"):
            name = key[5:]
            _check_grad(name, G[name], g[key])


def test_mrr_matches_reference_script():
    g = load_golden("msmarco_mrr.npz")
    ranked = {q: [int(x) for x in row] for q, row in enumerate(g["ranked"])}
    relevant = {q: [int(x) for x in row if x >= 0] for q, row in enumerate(g["relevant"])}
    assert abs(O.mrr_at_10(relevant, ranked) - float(g["mrr10"])) < 1e-12


def test_topk_and_metrics_hand_cases():
    Q = np.array([[1.0, 0.0], [0.0, 1.0]], np.float32)
    P = np.array([[0.5, 0.1], [0.9, 0.2], [0.5, 0.3], [0.1, 0.8]], np.float32)
    D, I = O.score_topk(Q, P, 3)
    assert I.tolist() == [[1, 0, 2], [3, 2, 1]]  # tie 0.5/0.5 -> lower position first
    np.testing.assert_allclose(D[0], [0.9, 0.5, 0.5])
    # k > Np pads with -1
    D, I = O.score_topk(Q, P, 6)
    assert I[0, 4:].tolist() == [-1, -1]
    # shard merge == global search
    rng = np.random.Generator(np.random.PCG64(3))
    Q = rng.standard_normal((5, 16)).astype(np.float32)
    P = rng.standard_normal((101, 16)).astype(np.float32)
    Dg, Ig = O.score_topk(Q, P, 10)
    Ds, Is = [], []
    for r in range(4):
        idx = O.shard_indices(101, r, 4)
        d, i = O.score_topk(Q, P[idx], 10)
        Ds.append(d)
        Is.append(idx[i])
    Dm, Im = O.merge_topk(Ds, Is, 10)
    assert np.array_equal(Im, Ig)
    # nDCG hand case: relevant doc at rank 3 of one relevant -> 1/log2(4)
    assert abs(O.ndcg_cut([7, 8, 9], {9: 1}, 10) - 0.5) < 1e-12
    assert abs(O.ndcg_cut([9, 7, 8], {9: 2, 7: 1}, 10) - 1.0) < 1e-12
    assert O.recip_rank([7, 8, 9], {9: 1}) == 1 / 3
    assert O.merged_order(7, 3).tolist() == [0, 3, 6, 1, 4, 2, 5]


def test_eval_dev_query_and_negatives():
    q2id = [10, 11]
    p2id = [100, 101, 102, 100, 103]  # position 3 duplicates pid 100
    I = np.array([[0, 3, 1, 2], [4, 2, 1, 0]])
    qrels = {10: {101: 1}, 11: {103: 1}}
    ndcg, mrr, n, pred = O.eval_dev_query(q2id, p2id, qrels, I, 4)
    assert pred[10] == {100: -1, 101: -2, 102: -3}
    assert n == 2 and abs(mrr - (0.5 + 1.0) / 2) < 1e-12
    assert abs(ndcg - (1 / np.log2(3) + 1.0) / 2) < 1e-12
    negs, rr = O.generate_negatives(q2id, p2id, {10: 101, 11: 103}, I, 2)
    assert negs == {10: [100], 11: [102, 101]}  # top (negative_sample+1) window, positive and dups dropped
    assert rr.tolist() == [1 / 3, 1.0]


def test_full_condenser_step_matches_reference():
    """Condenser head + both MLM losses + contrastive loss through the reference's
    CoCondenserForPretraining.forward (3 disclosed harness shims, tests/golden/make_golden.py)."""
    g = load_golden("coco_condenser_tiny.npz")
    cfg = cfg_from_golden(g)
    P = O.make_params(cfg, int(g["seed"]), std=float(g["std"]))
    Ph = O.make_head_params(cfg, int(g["n_head_layers"]), int(g["seed_head"]), std=float(g["std"]))
    total, parts, G, Gh = O.condenser_step(P, Ph, cfg, g["input_ids"], g["attention_mask"], g["labels"],
                                           int(g["n_head_layers"]), int(g["skip_from"]), late_mlm=True)
    assert abs(total - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert parts["mlm_head"] > 0 and parts["mlm_late"] > 0 and parts["co"] > 0
    for key in g.files:
        if key.startswith("grad:"):
            _check_grad(key[5:], G[key[5:]], g[key])
        elif key.startswith("hgrad:"):
            name = key[6:]
            if name.endswith("key.bias"):
                continue
            assert _rel(Gh[name], g[key]) < 2e-3, (name, _rel(Gh[name], g[key]))
    assert _rel(G["embeddings.word_embeddings.weight"][:64], g["grad_rows:embeddings.word_embeddings.weight"]) < 2e-3


def test_lamb_and_clip_oracle_match_reference_lamb_golden():
    """oracle.lamb_step / clip_grad_norm vs three steps of the reference's Lamb class (tests/golden/lamb_steps.npz)."""
    z = load_golden("lamb_steps.npz")
    for wd, tag in ((0.0, "wd0"), (0.01, "wd01")):
        ps = [z[f"p0_{i}"].astype(np.float64) for i in range(5)]
        m = [np.zeros_like(p) for p in ps]
        v = [np.zeros_like(p) for p in ps]
        for step in range(3):
            gs = [z[f"{tag}_g{step}_{i}"].astype(np.float64) for i in range(5)]
            norm, coef = O.clip_grad_norm(gs, 1.0)
            assert abs(norm - float(z[f"{tag}_norm{step}"])) <= 1e-5 * norm
            trust = O.lamb_step(ps, [g * coef for g in gs], m, v, lr=2e-3, eps=1e-6, weight_decay=wd)
            np.testing.assert_allclose(trust, z[f"{tag}_trust{step}"], rtol=2e-4)
            for i in range(5):
                np.testing.assert_allclose(ps[i], z[f"{tag}_p{step + 1}_{i}"], rtol=1e-4, atol=2e-7)


def test_idro_oracle_matches_reference_idro_golden():
    """Two steps of the reference's iDROLoss through BertDot_NLL_LN (tests/golden/idro_steps.npz): robust loss, group
    statistics, the updated group weights and gradients of the re-weighted loss."""
    z = load_golden("idro_steps.npz")
    cfg = cfg_from_golden(z)
    P = params_from_golden(z, np.float64)
    G, alpha, eps, ema, rho = (float(x) for x in z["hyper"])
    G = int(G)
    h = np.ones(G)
    for step in range(2):
        batch = tuple(z[f"s{step}_{k}"] for k in ("q_ids", "q_mask", "a_ids", "a_mask", "b_ids", "b_mask"))
        robust, gl, cnt, h, grads = O.idro_step(P, cfg, batch, z[f"s{step}_groups"], h, G, alpha, eps, ema, rho)
        assert abs(robust - float(z[f"s{step}_robust"])) <= 2e-5 * abs(robust)
        np.testing.assert_allclose(gl, z[f"s{step}_group_losses"], rtol=2e-5, atol=1e-6)
        np.testing.assert_array_equal(cnt, z[f"s{step}_group_counts"])
        np.testing.assert_allclose(h, z[f"s{step}_h_fun"], rtol=2e-4, atol=1e-6)
        for key in z.files:
            if key.startswith(f"s{step}_grad:"):
                name = key.split(":", 1)[1]
                if name.endswith("key.bias"):  # identically zero in exact arithmetic (softmax rows are shift invariant)
                    assert np.abs(grads[name]).max() < 1e-6 and np.abs(z[key]).max() < 1e-6
                else:
                    assert _rel(grads[name], z[key]) < 2e-4, name


def test_dro_greedy_oracle_matches_reference_golden():
    """Three steps of the reference's DROGreedyLoss (both h_fun update rules): robust loss, EMA buffers, weights, and the
    gradient of the re-weighted loss on step 1."""
    z = load_golden("dro_greedy_steps.npz")
    cfg = cfg_from_golden(z)
    P = params_from_golden(z, np.float64)
    G, alpha, eps, ema = (float(x) for x in z["hyper"])
    G = int(G)
    w = z["weights"].astype(np.float64)
    for wema, tag in ((False, "hard"), (True, "ema")):
        st = O.DROGreedyState(G)
        for step in range(3):
            enc = []
            for ids, mask in ((z[f"s{step}_q_ids"], z[f"s{step}_q_mask"]), (z[f"s{step}_a_ids"], z[f"s{step}_a_mask"]),
                              (z[f"s{step}_b_ids"], z[f"s{step}_b_mask"])):
                hs, cache = O.encoder_fwd(P, cfg, ids, mask, keep_cache=True)
                enc.append((hs[-1], cache))
            q, a, b = (O.cls_embedding(e[0]) for e in enc)
            rows, logits = O.triplet_nll(q, a, b)
            robust, row_w, gl, cnt = O.dro_greedy_forward(st, rows, z[f"s{step}_groups"], w, G, alpha, eps, ema, wema)
            assert abs(robust - float(z[f"{tag}_s{step}_robust"])) <= 2e-5 * abs(robust)
            np.testing.assert_allclose(gl, z[f"{tag}_s{step}_group_losses"], rtol=2e-5, atol=1e-7)
            np.testing.assert_array_equal(cnt, z[f"{tag}_s{step}_group_counts"])
            np.testing.assert_allclose(st.h_fun, z[f"{tag}_s{step}_h_fun"], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(st.sum_losses, z[f"{tag}_s{step}_sum_losses"], rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(st.count_cat, z[f"{tag}_s{step}_count_cat"], rtol=1e-6)
            if step == 1:
                zl = logits - logits.max(1, keepdims=True)
                p = np.exp(zl) / np.exp(zl).sum(1, keepdims=True)
                dl = p.copy(); dl[:, 0] -= 1.0
                dl *= row_w[:, None]
                dE = (dl[:, :1] * a + dl[:, 1:] * b, dl[:, :1] * q, dl[:, 1:] * q)
                tot = {}
                for (last, cache), d in zip(enc, dE):
                    d_last = np.zeros_like(last); d_last[:, 0] = d
                    for k, v in O.encoder_bwd(P, cfg, cache, d_last).items():
                        tot[k] = tot.get(k, 0) + v
                for key in z.files:
                    if key.startswith(f"{tag}_s1_grad:"):
                        assert _rel(tot[key.split(":", 1)[1]], z[key]) < 2e-4, key


def test_collator_oracle_matches_reference_methods():
    """Word grouping, greedy whole-word selection and the truncation window against the reference collator's own methods
    driven with recorded permutations / offsets (tests/golden/collator_cases.npz)."""
    z = load_golden("collator_cases.npz")
    for i in range(int(z["n_cases"])):
        sub = z[f"c{i}_sub"].astype(bool)
        spec = z[f"c{i}_special"].astype(bool) if f"c{i}_special" in z.files else None  # the cases with [UNK] / [SEP] inside the span
        groups = O.whole_word_groups(list(sub), None if spec is None else list(spec))
        assert [len(g) for g in groups] == list(z[f"c{i}_groups_len"])
        assert [t for g in groups for t in g] == list(z[f"c{i}_groups_flat"])
        mask = O.whole_word_mask(groups, list(z[f"c{i}_order"]), len(sub), float(z[f"c{i}_prob"]))
        assert mask == list(z[f"c{i}_mask"]), i
    # truncation: window [left, left + max_seq_length - 2) - collate_span takes `left` from its generator; check the slicing
    ex = list(range(100, 140))
    for j in range(3):
        left = int(z[f"trunc{j}_left"])
        assert ex[left:left + 14] == list(z[f"trunc{j}"])
    assert list(z["trunc_short"]) == ex[:9]


def test_collate_span_invariants_and_rates():
    """collate_span end to end: layout, whole words masked together, budget, label / id consistency, 80/10/10 rates."""
    rng = np.random.Generator(np.random.PCG64(1))
    V, L = 1000, 64
    is_sub = (rng.random(V) < 0.3).astype(np.uint8)
    is_sub[:110] = 0
    n_mask = n_tok = n_repl = n_rand = n_keep = 0
    for ex in range(300):
        n = int(rng.integers(1, 90))
        toks = rng.integers(110, V, n)
        ids, labels, att = O.collate_span(toks, is_sub, seed=7, ex=ex, L=L, cls_id=101, sep_id=102, pad_id=0, mask_id=103,
                                          mlm_probability=0.15)
        m = min(n, L - 2)
        assert ids[0] == 101 and ids[m + 1] == 102 and (ids[m + 2:] == 0).all()
        assert att.sum() == m + 2 and (att[:m + 2] == 1).all()
        lab = labels[1:m + 1]
        picked = lab != -100
        want = min(512, max(1, int(round(m * 0.15))))
        assert 0 <= picked.sum() <= want and labels[0] == -100 and (labels[m + 1:] == -100).all()
        # the window is a contiguous slice of the span and unmasked positions keep their token
        win = ids[1:m + 1].copy()
        win[picked] = lab[picked]
        starts = [s for s in range(n - m + 1) if (toks[s:s + m] == win).all()]
        assert starts
        # whole words: a "##" piece is masked iff the token in front of it (same word) is
        for i in range(1, m):
            if is_sub[win[i]] == 1:
                assert picked[i] == picked[i - 1]
        n_tok += m; n_mask += int(picked.sum())
        cur = ids[1:m + 1][picked]
        n_repl += int((cur == 103).sum()); n_keep += int((cur == lab[picked]).sum()); n_rand += int(((cur != 103) & (cur != lab[picked])).sum())
    assert 0.10 < n_mask / n_tok < 0.16
    assert abs(n_repl / n_mask - 0.8) < 0.05 and abs(n_rand / n_mask - 0.1) < 0.04 and abs(n_keep / n_mask - 0.1) < 0.04
