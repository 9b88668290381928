#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE in the build
container (CPU, fp32).  Run once from the repo root:

    python tests/golden/make_golden.py            # every target, each in its own interpreter
    python tests/golden/make_golden.py coco ance  # selected targets in this interpreter

It imports /root/reference/COCO/modeling.py and /root/reference/ANCE/model/models.py
(read-only) on top of the installed transformers BertModel (eager attention, fp32, eval),
loads seeded weights from oracle.make_params, and stores ONLY inputs + expected outputs
(data, no reference source).  /root/reference does not exist on the GPU box; the tests read
the .npz files committed next to this script.

Harness-side shims (never written into /root/reference):
  1. ``data_args.train_method`` is set because COCO/modeling.py:179 reads a field that
     COCO/arguments.py does not define.
  2. ``model._world_size`` is overridden per instance to emulate world sizes > 1 for the
     loss scaling at COCO/modeling.py:247 (the real call needs W processes).
Neither touches the encoder -> [CLS] -> contrastive path arithmetic.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import OracleConfig, make_params  # noqa: E402

REF = "/root/reference"
OUT = os.environ.get("COCODR_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))  # (tests regenerate into a scratch directory)
STD = 0.08  # larger than HF's 0.02 so that [CLS] rows differ visibly between inputs
# The [CLS] embedding is a LayerNorm output: |q|^2 ~ H whatever the weight scale, so raw dot-product logits sit near 130
# at H = 128 and a bf16-vs-fp32 comparison of them says little about the loss.  The triplet / DRO fixtures therefore
# shrink the LAST LayerNorm (gain and bias x 0.2): logits O(5), logit gaps O(0.1-1), a loss that reacts to errors.
FINAL_LN_SCALE = 0.2


def scale_final_ln(P, cfg, s=FINAL_LN_SCALE):
    last = f"encoder.layer.{cfg.num_hidden_layers - 1}.output.LayerNorm."
    P[last + "weight"] = (P[last + "weight"] * s).astype(P[last + "weight"].dtype)
    P[last + "bias"] = (P[last + "bias"] * s).astype(P[last + "bias"].dtype)
    return P


def hf_config(cfg: OracleConfig):
    from transformers import BertConfig
    return BertConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
        max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager")


def load_into(bert, P):
    sd = bert.state_dict()
    for k, v in P.items():
        assert k in sd, k
        sd[k].copy_(torch.from_numpy(v))


def synth_batch(rng, B, L, V, lo=4):
    ids = np.zeros((B, L), np.int64)
    mask = np.zeros((B, L), np.int64)
    for b in range(B):
        n = L if b == 0 else int(rng.integers(lo, L + 1))
        ids[b, :n] = rng.integers(5, V, n)
        ids[b, 0] = 1  # [CLS]-like
        ids[b, n - 1] = 2  # [SEP]-like
        mask[b, :n] = 1
    return ids, mask


def selected_grads(named_params, prefix):
    keep = ("embeddings.position_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias",
            "encoder.layer.0.attention.self.query.weight", "encoder.layer.0.attention.self.key.bias", "encoder.layer.0.attention.self.query.bias",
            "encoder.layer.0.attention.self.value.weight", "encoder.layer.0.attention.output.dense.weight",
            "encoder.layer.0.attention.output.LayerNorm.weight", "encoder.layer.1.intermediate.dense.weight",
            "encoder.layer.1.intermediate.dense.bias", "encoder.layer.1.output.dense.weight",
            "encoder.layer.1.output.LayerNorm.bias", "embeddings.token_type_embeddings.weight")
    out = {}
    sums = {}
    for name, p in named_params:
        if not name.startswith(prefix):
            continue
        short = name[len(prefix):]
        if p.grad is None:
            continue
        g = p.grad.detach().numpy()
        sums[short] = np.array([g.sum(dtype=np.float64), np.abs(g).sum(dtype=np.float64)])
        if short in keep:
            out["grad:" + short] = g.copy()
        if short == "embeddings.word_embeddings.weight":
            out["grad_rows:" + short] = g[:64].copy()  # ids < 64 are hit often in the synthetic batch
    names = sorted(sums)
    out["gradsum_names"] = np.array(names)
    out["gradsums"] = np.stack([sums[n] for n in names])
    return out


def golden_coco():
    """COCO path: BertForMaskedLM encoder -> hidden_states -> [CLS] -> compute_contrastive_loss -> backward.
    Follows COCO/modeling.py:199-208,229,244-248 using the reference's own module objects."""
    sys.path.insert(0, os.path.join(REF, "COCO"))
    import modeling as coco_modeling  # reference
    from arguments import ModelArguments, DataTrainingArguments  # reference
    from transformers import BertForMaskedLM

    cfg = OracleConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                       intermediate_size=512, max_position_embeddings=64)
    seed = 1234
    P = make_params(cfg, seed, std=STD)
    torch.manual_seed(0)
    lm = BertForMaskedLM(hf_config(cfg))
    load_into(lm.bert, P)
    B, L = 6, 32
    rng = np.random.Generator(np.random.PCG64(99))
    ids, mask = synth_batch(rng, B, L, cfg.vocab_size)

    model_args = ModelArguments(n_head_layers=0, skip_from=1, late_mlm=False)
    data_args = DataTrainingArguments()
    data_args.train_method = "coco"  # shim 1
    train_args = types.SimpleNamespace(per_device_train_batch_size=B // 2, local_rank=-1)
    model = coco_modeling.CoCondenserForPretraining(lm, model_args, data_args, train_args)
    model.lm.eval()  # COCO/modeling.py:198
    out = {}
    res = {}
    for W in (1, 2):
        model.zero_grad()
        lm_out = model.lm(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                          output_hidden_states=True, return_dict=True)
        cls_hiddens = lm_out.hidden_states[-1][:, :1]
        co_cls = cls_hiddens.squeeze()
        model._world_size = (lambda w=W: w)  # shim 2
        rows = model.compute_contrastive_loss(co_cls)
        loss = rows.mean()
        loss.backward()
        res[W] = (rows.detach().numpy().copy(), float(loss))
        if W == 1:
            out["hidden_states"] = np.stack([h.detach().numpy() for h in lm_out.hidden_states])
            out["co_target"] = model.co_target.numpy().copy()
            out.update(selected_grads(model.lm.named_parameters(), "bert."))
    out.update(dict(input_ids=ids, attention_mask=mask, seed=np.int64(seed), std=np.float64(STD),
                    cfg=np.array([cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                                  cfg.intermediate_size, cfg.max_position_embeddings, cfg.type_vocab_size]),
                    loss_rows_w1=res[1][0], loss_w1=np.float64(res[1][1]),
                    loss_rows_w2=res[2][0], loss_w2=np.float64(res[2][1])))
    np.savez_compressed(os.path.join(OUT, "coco_contrastive_tiny.npz"), **out)
    print("coco golden: loss", res[1][1], "hs", out["hidden_states"].shape)

    # stand-alone loss goldens at several M / world sizes on random E (no encoder)
    stand = {}
    for M, W, H in ((8, 1, 32), (16, 2, 48), (64, 8, 96)):
        E = (np.random.Generator(np.random.PCG64(M)).standard_normal((M, H)) * 0.7).astype(np.float32)
        ta = types.SimpleNamespace(per_device_train_batch_size=M // 2, local_rank=-1)
        m2 = coco_modeling.CoCondenserForPretraining(lm, model_args, data_args, ta)
        m2._world_size = (lambda w=W: w)
        Et = torch.from_numpy(E).requires_grad_(True)
        rows = m2.compute_contrastive_loss(Et)
        rows.mean().backward()
        stand[f"E_{M}"] = E
        stand[f"W_{M}"] = np.int64(W)
        stand[f"rows_{M}"] = rows.detach().numpy()
        stand[f"dE_{M}"] = Et.grad.numpy().copy()
        stand[f"target_{M}"] = m2.co_target.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "contrastive_loss.npz"), **stand)
    sys.path.pop(0)


def golden_ance():
    """ANCE path: BertDot_NLL_LN triplet forward/backward - ANCE/model/models.py:97-106,225-232,234-262."""
    sys.path.insert(0, os.path.join(REF, "ANCE"))
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    from model.models import BertDot_NLL_LN  # reference

    cfg = OracleConfig(vocab_size=800, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                       intermediate_size=256, max_position_embeddings=64)
    seed = 4321
    P = scale_final_ln(make_params(cfg, seed, std=STD), cfg)
    torch.manual_seed(0)
    model = BertDot_NLL_LN(hf_config(cfg))
    load_into(model.bert, P)
    model.eval()  # dropout p=0 anyway (config) - parity tests run without dropout (SURVEY 7 iv)
    rng = np.random.Generator(np.random.PCG64(7))
    B = 4
    q_ids, q_mask = synth_batch(rng, B, 16, cfg.vocab_size)
    a_ids, a_mask = synth_batch(rng, B, 32, cfg.vocab_size)
    b_ids, b_mask = synth_batch(rng, B, 32, cfg.vocab_size)
    weights = np.array([1.0, 0.5, 2.0, 1.0], np.float32)
    t = torch.from_numpy
    model.zero_grad()
    loss, acc, logits = model(t(q_ids), t(q_mask), t(a_ids), t(a_mask), t(b_ids), t(b_mask),
                              weights=t(weights))
    loss.backward()
    with torch.no_grad():
        qe = model.query_emb(t(q_ids), t(q_mask)).numpy()
        ae = model.body_emb(t(a_ids), t(a_mask)).numpy()
        be = model.body_emb(t(b_ids), t(b_mask)).numpy()
    out = dict(q_ids=q_ids, q_mask=q_mask, a_ids=a_ids, a_mask=a_mask, b_ids=b_ids, b_mask=b_mask,
               weights=weights, seed=np.int64(seed), std=np.float64(STD), final_ln_scale=np.float64(FINAL_LN_SCALE),
               loss=np.float64(float(loss)), logits=logits.detach().numpy(),
               acc=acc.numpy(), q_emb=qe, a_emb=ae, b_emb=be,
               cfg=np.array([cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                             cfg.intermediate_size, cfg.max_position_embeddings, cfg.type_vocab_size]))
    out.update(selected_grads(model.named_parameters(), "bert."))
    np.savez_compressed(os.path.join(OUT, "ance_triplet_tiny.npz"), **out)
    print("ance golden: loss", float(loss), "logits", logits.detach().numpy())
    sys.path.pop(0)


def golden_dropout():
    """Where the reference drops: the reference's own BertDot_NLL_LN in train() mode (ANCE/drivers/run_ann.py:293) on top of
    transformers' BertModel with hidden_dropout_prob = 0.1 / attention_probs_dropout_prob = 0.15, three encoder passes
    (query, positive, negative).  torch.nn.functional.dropout - what nn.Dropout.forward and eager_attention_forward call -
    is replaced by a multiplication with the oracle's counter-based mask of the site (oracle/dropout_oracle.py): the site is
    identified by call order (embedding output, then per layer: attention probabilities, attention-output dense, FFN-output
    dense), the pass number is the `call` of the keys.  Everything else - which tensors are dropped, on which side of the
    residual add / LayerNorm, how the loss and the gradients follow - is the reference's and transformers' own code."""
    sys.path.insert(0, os.path.join(REF, "ANCE"))
    import torch.distributed as dist
    import torch.nn.functional as F
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    from model.models import BertDot_NLL_LN  # reference
    from oracle import dropout_oracle as D

    cfg = OracleConfig(vocab_size=800, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                       intermediate_size=256, max_position_embeddings=64)
    seed, drop_seed, p_hidden, p_attn = 2468, 777, 0.1, 0.15
    P = scale_final_ln(make_params(cfg, seed, std=STD), cfg)
    hc = hf_config(cfg)
    hc.hidden_dropout_prob, hc.attention_probs_dropout_prob = p_hidden, p_attn
    torch.manual_seed(0)
    model = BertDot_NLL_LN(hc)
    load_into(model.bert, P)
    model.train()
    rng = np.random.Generator(np.random.PCG64(11))
    B = 4
    q_ids, q_mask = synth_batch(rng, B, 32, cfg.vocab_size)
    a_ids, a_mask = synth_batch(rng, B, 32, cfg.vocab_size)
    b_ids, b_mask = synth_batch(rng, B, 32, cfg.vocab_size)
    weights = np.array([1.0, 0.5, 2.0, 1.0], np.float32)

    state = dict(call=0, site=0)
    sites_per_pass = 1 + 3 * cfg.num_hidden_layers
    seen = []

    def fake_dropout(x, p=0.5, training=True, inplace=False):
        assert training and p > 0
        k = state["site"] % sites_per_pass
        if k == 0:
            state["call"] += 1
            layer, kind = 0, D.KIND_EMBED
        else:
            layer, kind = (k - 1) // 3, (k - 1) % 3
        state["site"] += 1
        assert abs(p - (p_attn if kind == D.KIND_ATTN_PROBS else p_hidden)) < 1e-9, (p, kind)
        seen.append((state["call"], layer, kind, tuple(x.shape)))
        m = D.multiplier(tuple(x.shape), p, drop_seed, state["call"], layer, kind)
        return x * torch.from_numpy(m)

    real = F.dropout
    F.dropout = fake_dropout
    torch.nn.functional.dropout = fake_dropout
    try:
        t = torch.from_numpy
        model.zero_grad()
        loss, acc, logits = model(t(q_ids), t(q_mask), t(a_ids), t(a_mask), t(b_ids), t(b_mask), weights=t(weights))
        loss.backward()
    finally:
        F.dropout = real
        torch.nn.functional.dropout = real
    assert state["site"] == 3 * sites_per_pass and state["call"] == 3, state
    assert [s[3] for s in seen[:4]] == [(B, 32, 128), (B, 2, 32, 32), (B, 32, 128), (B, 32, 128)], seen[:4]
    out = dict(q_ids=q_ids, q_mask=q_mask, a_ids=a_ids, a_mask=a_mask, b_ids=b_ids, b_mask=b_mask, weights=weights,
               seed=np.int64(seed), std=np.float64(STD), final_ln_scale=np.float64(FINAL_LN_SCALE), drop_seed=np.int64(drop_seed),
               p_hidden=np.float64(p_hidden), p_attn=np.float64(p_attn), loss=np.float64(float(loss)),
               logits=logits.detach().numpy(),
               cfg=np.array([cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                             cfg.intermediate_size, cfg.max_position_embeddings, cfg.type_vocab_size]))
    out.update(selected_grads(model.named_parameters(), "bert."))
    np.savez_compressed(os.path.join(OUT, "dropout_sites.npz"), **out)
    print("dropout golden: loss", float(loss), "logits", logits.detach().numpy())
    sys.path.pop(0)


def golden_mrr():
    """evaluate/evaluation/msmarco_eval.py:109-139 compute_metrics on a seeded synthetic run."""
    sys.path.insert(0, os.path.join(REF, "evaluate", "evaluation"))
    import msmarco_eval  # reference, pure stdlib
    rng = np.random.Generator(np.random.PCG64(5))
    nq = 50
    ranked = {q: [int(x) for x in 1 + rng.permutation(40)[:20]] + [0] * 980 for q in range(nq)}
    relevant = {q: [int(x) for x in rng.integers(1, 41, int(rng.integers(1, 4)))] for q in range(nq)}
    m = msmarco_eval.compute_metrics(relevant, ranked)
    np.savez_compressed(os.path.join(OUT, "msmarco_mrr.npz"),
                        ranked=np.array([ranked[q][:20] for q in range(nq)]),
                        relevant=np.array([relevant[q] + [-1] * (3 - len(relevant[q])) for q in range(nq)]),
                        mrr10=np.float64(m["MRR @10"]))
    print("mrr golden:", m)
    sys.path.pop(0)


def golden_condenser():
    """Full coCondenser step: Condenser head + both MLM losses + contrastive loss, forward/backward
    through the reference's CoCondenserForPretraining.forward (COCO/modeling.py:192-235).  Harness-side shims
    (none touches /root/reference): (1) data_args.train_method; (2) get_extended_attention_mask is
    wrapped to drop the 3rd positional `device` argument that transformers 5.x reads as `dtype`; (3) every c_head
    layer is wrapped to return a 1-tuple, restoring the <=4.x BertLayer return convention `layer_out[0]` relies on."""
    sys.path.insert(0, os.path.join(REF, "COCO"))
    import modeling as coco_modeling  # reference
    from arguments import ModelArguments, DataTrainingArguments  # reference
    from transformers import BertForMaskedLM
    from oracle import make_head_params, layer_names

    cfg = OracleConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                       intermediate_size=512, max_position_embeddings=64)
    seed, seed_h, n_head, skip_from = 2468, 1357, 2, 1
    P = make_params(cfg, seed, std=STD)
    Ph = make_head_params(cfg, n_head, seed_h, std=STD)
    torch.manual_seed(0)
    lm = BertForMaskedLM(hf_config(cfg))
    load_into(lm.bert, P)
    with torch.no_grad():
        tr = lm.cls.predictions.transform
        tr.dense.weight.copy_(torch.from_numpy(Ph["cls.predictions.transform.dense.weight"]))
        tr.dense.bias.copy_(torch.from_numpy(Ph["cls.predictions.transform.dense.bias"]))
        tr.LayerNorm.weight.copy_(torch.from_numpy(Ph["cls.predictions.transform.LayerNorm.weight"]))
        tr.LayerNorm.bias.copy_(torch.from_numpy(Ph["cls.predictions.transform.LayerNorm.bias"]))
        lm.cls.predictions.bias.copy_(torch.from_numpy(Ph["cls.predictions.bias"]))
        lm.cls.predictions.decoder.bias = lm.cls.predictions.bias
    assert lm.cls.predictions.decoder.weight is lm.bert.embeddings.word_embeddings.weight  # tied
    B, L = 6, 32
    rng = np.random.Generator(np.random.PCG64(77))
    ids, mask = synth_batch(rng, B, L, cfg.vocab_size)
    labels = np.full((B, L), -100, np.int64)
    for b in range(B):
        n = int(mask[b].sum())
        pick = 1 + rng.permutation(n - 2)[: max(1, int(0.15 * n))]
        labels[b, pick] = ids[b, pick]
        ids[b, pick] = 3  # [MASK]-like
    model_args = ModelArguments(n_head_layers=n_head, skip_from=skip_from, late_mlm=True)
    data_args = DataTrainingArguments()
    data_args.train_method = "coco"  # shim 1
    train_args = types.SimpleNamespace(per_device_train_batch_size=B // 2, local_rank=-1)
    model = coco_modeling.CoCondenserForPretraining(lm, model_args, data_args, train_args)
    with torch.no_grad():
        for i in range(n_head):
            sd = model.c_head[i].state_dict()
            names = layer_names(i, "c_head.")
            for k in sd:
                sd[k].copy_(torch.from_numpy(Ph[f"c_head.{i}." + k]))
    orig_mask_fn = model.lm.get_extended_attention_mask
    model.lm.get_extended_attention_mask = lambda m, shape, device=None: orig_mask_fn(m, shape)  # shim 2

    class TupleOut(torch.nn.Module):  # shim 3
        def __init__(self, layer):
            super().__init__()
            self.layer = layer

        def forward(self, hidden, attention_mask):
            out = self.layer(hidden, attention_mask)
            return out if isinstance(out, tuple) else (out,)

    model.c_head = torch.nn.ModuleList([TupleOut(l) for l in model.c_head])
    model.zero_grad()
    loss = model({"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}, torch.from_numpy(labels))
    loss.backward()
    out = dict(input_ids=ids, attention_mask=mask, labels=labels, seed=np.int64(seed), seed_head=np.int64(seed_h),
               std=np.float64(STD), n_head_layers=np.int64(n_head), skip_from=np.int64(skip_from), loss=np.float64(float(loss)),
               cfg=np.array([cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                             cfg.intermediate_size, cfg.max_position_embeddings, cfg.type_vocab_size]))
    out.update(selected_grads(model.lm.named_parameters(), "bert."))
    head = {}
    for name, p_ in model.named_parameters():
        if name.startswith("c_head.") and p_.grad is not None:
            short = name.replace(".layer.", ".", 1)  # undo the TupleOut nesting: c_head.0.layer.attention... -> c_head.0.attention...
            if any(t in short for t in ("0.attention.self.query.weight", "1.intermediate.dense.weight", "1.output.LayerNorm.weight",
                                        "0.attention.output.dense.bias", "1.attention.self.value.weight")):
                head["hgrad:" + short] = p_.grad.numpy().copy()
        if name.startswith("lm.cls.") and p_.grad is not None and "decoder" not in name:
            head["hgrad:" + name[len("lm."):]] = p_.grad.numpy().copy()
    out.update(head)
    np.savez_compressed(os.path.join(OUT, "coco_condenser_tiny.npz"), **out)
    print("condenser golden: loss", float(loss), "head grads", sorted(head)[:3], len(head))
    sys.path.pop(0)


def golden_token_cache():
    """A small token cache written with the reference's record formula (ANCE/data/msmarco_data.py:279) using its own
    ``pad_input_ids`` and read back with its own ``EmbeddingCache`` (ANCE/utils/util.py:316-370).  Harness shim: an
    empty stand-in module for the absent ``pytrec_eval`` so that ANCE/utils/util.py imports (the cache code never
    touches it)."""
    sys.modules.setdefault("pytrec_eval", types.ModuleType("pytrec_eval"))
    sys.path.insert(0, os.path.join(REF, "ANCE"))
    from utils.util import EmbeddingCache, pad_input_ids  # reference
    import json
    import tempfile
    rng = np.random.Generator(np.random.PCG64(21))
    L, n = 16, 9
    recs = []
    blob = b""
    for i in range(n):
        toks = [int(x) for x in rng.integers(1, 30000, int(rng.integers(1, 25)))]
        ln = min(len(toks), L)
        blob += ln.to_bytes(4, "big") + np.array(pad_input_ids(toks, L), np.int32).tobytes()
    d = tempfile.mkdtemp()
    path = os.path.join(d, "passages")
    with open(path, "wb") as f:
        f.write(blob)
    with open(path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": n, "embedding_size": L}, f)
    lens, toks = [], []
    with EmbeddingCache(path) as cache:
        for i in range(n):
            ln, t = cache[i]
            lens.append(ln)
            toks.append(np.array(t))
    np.savez_compressed(os.path.join(OUT, "token_cache.npz"), blob=np.frombuffer(blob, np.uint8), max_len=np.int64(L),
                        lengths=np.array(lens), tokens=np.stack(toks))
    print("token cache golden:", lens)
    sys.path.pop(0)


def golden_training_rows():
    """The ANCE training stream: the reference's own ``GetTripletTrainingDataProcessingFn`` / ``GetTrainingDataProcessingFn``
    (ANCE/data/msmarco_data.py:328-384) over its ``StreamingDataset`` (ANCE/utils/util.py:372-399) and a DataLoader, on a small
    query / passage cache read with its ``EmbeddingCache``.  World sizes 1 and 2 (``dist`` is stubbed on the dataset module: the
    sharding rule only asks it for rank and world size).  Same ``pytrec_eval`` stand-in as the token-cache golden."""
    sys.modules.setdefault("pytrec_eval", types.ModuleType("pytrec_eval"))
    sys.path.insert(0, os.path.join(REF, "ANCE"))
    sys.path.insert(0, os.path.join(REF, "ANCE", "data"))
    import json
    import tempfile
    import utils.util as U  # reference
    import msmarco_data as MD  # reference
    from torch.utils.data import DataLoader
    rng = np.random.Generator(np.random.PCG64(31))
    d = tempfile.mkdtemp()

    def make_cache(name, n, L):
        blob = b""
        for i in range(n):
            toks = [int(x) for x in rng.integers(1, 30000, int(rng.integers(1, L + 6)))]
            blob += min(len(toks), L).to_bytes(4, "big") + np.array(U.pad_input_ids(toks, L), np.int32).tobytes()
        path = os.path.join(d, name)
        with open(path, "wb") as f:
            f.write(blob)
        with open(path + "_meta", "w") as f:
            json.dump({"type": "int32", "total_number": n, "embedding_size": L}, f)
        return path, np.frombuffer(blob, np.uint8)

    Lq, Lp = 8, 12
    qpath, qblob = make_cache("queries", 7, Lq)
    ppath, pblob = make_cache("passages", 20, Lp)
    lines = []
    for qid in (3, 0, 5, 6, 1):
        negs = [int(x) for x in rng.choice(20, int(rng.integers(1, 4)), replace=False)]
        lines.append("{}\t{}\t{}\n".format(qid, int(rng.integers(0, 20)), ",".join(map(str, negs))))
    args = types.SimpleNamespace(max_query_length=Lq, max_seq_length=Lp)
    out = dict(q_blob=qblob, p_blob=pblob, Lq=np.int64(Lq), Lp=np.int64(Lp), lines=np.array(lines), batch_size=np.int64(3))

    class FakeDist:
        def __init__(self, rank, world):
            self.rank, self.world = rank, world

        def is_initialized(self):
            return self.world > 1

        def get_world_size(self):
            return self.world

        def get_rank(self):
            return self.rank

    real_dist = U.dist
    with U.EmbeddingCache(qpath) as qc, U.EmbeddingCache(ppath) as pc:
        for world in (1, 2):
            for rank in range(world):
                U.dist = FakeDist(rank, world)
                ds = U.StreamingDataset(lines, MD.GetTripletTrainingDataProcessingFn(args, qc, pc), size=-1)
                batches = list(DataLoader(ds, batch_size=3))
                for bi, b in enumerate(batches):
                    for name, j in (("q_ids", 0), ("q_mask", 1), ("a_ids", 3), ("a_mask", 4), ("b_ids", 6), ("b_mask", 7)):
                        out[f"trip_w{world}_r{rank}_b{bi}_{name}"] = b[j].long().numpy()
                out[f"trip_w{world}_r{rank}_nb"] = np.int64(len(batches))
        U.dist = FakeDist(0, 1)
        ds = U.StreamingDataset(lines, MD.GetTrainingDataProcessingFn(args, qc, pc), size=-1)
        recs = list(ds)
        out["pair_q_ids"] = np.stack([r[0].long().numpy() for r in recs])
        out["pair_p_ids"] = np.stack([r[3].long().numpy() for r in recs])
        out["pair_p_mask"] = np.stack([r[4].long().numpy() for r in recs])
        out["pair_label"] = np.array([int(r[6]) for r in recs])
    U.dist = real_dist
    np.savez_compressed(os.path.join(OUT, "training_rows.npz"), **out)
    print("training rows golden:", {k: v for k, v in out.items() if k.endswith("_nb")}, len(recs), "pair records")
    sys.path.pop(0)
    sys.path.pop(0)


def golden_coco_dataset():
    """``CoCondenserDataset.__getitem__`` (COCO/data.py:169-183) under a seeded Python ``random``: which two spans of each
    document become the positive pair."""
    import random
    sys.path.insert(0, os.path.join(REF, "COCO"))
    from data import CoCondenserDataset  # reference
    rng = np.random.Generator(np.random.PCG64(41))
    docs = []
    for n_spans in (1, 2, 5, 3, 1, 8):
        docs.append({"spans": [[int(x) for x in rng.integers(1000, 30000, int(rng.integers(2, 9)))] for _ in range(n_spans)]})
    ds = CoCondenserDataset(docs, None)
    random.seed(1234)
    picks = []
    for epoch in range(2):
        for i in range(len(ds)):
            picks.append(ds[i]["span"])
    flat = [s for d in docs for s in d["spans"]]
    np.savez_compressed(os.path.join(OUT, "coco_dataset.npz"),
                        span_tokens=np.concatenate([np.asarray(s, np.int64) for s in flat]),
                        span_lens=np.asarray([len(s) for s in flat]), doc_spans=np.asarray([len(d["spans"]) for d in docs]),
                        seed=np.int64(1234),
                        pick_tokens=np.concatenate([np.asarray(s, np.int64) for p in picks for s in p]),
                        pick_lens=np.asarray([len(s) for p in picks for s in p]))
    print("coco dataset golden:", len(picks), "items")
    sys.path.pop(0)


def golden_idro():
    """f2: two training steps of the reference's iDRO re-weighting (ANCE/model/dro_loss.py:160-254) driven through
    BertDot_NLL_LN.forward(group_ids=...) (ANCE/model/models.py:234-273) on a 12-layer toy BERT (iDROLoss selects
    layer.9-11 by name, :177-189).  1-rank gloo group for its all_reduce.  No shims: the class runs unmodified."""
    sys.path.insert(0, os.path.join(REF, "ANCE"))
    import types
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29534", rank=0, world_size=1)
    from model.models import BertDot_NLL_LN  # reference

    cfg = OracleConfig(vocab_size=400, hidden_size=128, num_hidden_layers=12, num_attention_heads=2,
                       intermediate_size=256, max_position_embeddings=64)
    seed = 777
    P = scale_final_ln(make_params(cfg, seed, std=STD), cfg)
    torch.manual_seed(0)
    model = BertDot_NLL_LN(hf_config(cfg))
    load_into(model.bert, P)
    model.train()  # iDROLoss.forward only defines its group statistics in training mode (:226); dropout is 0 by config
    G, alpha, eps, ema, rho = 5, 0.25, 0.01, 0.1, 0.1
    args = types.SimpleNamespace(model_size="base", local_rank=0)
    model.add_group_loss(args=args, n_groups=G, dro_type="idro", alpha=alpha, eps=eps, ema=ema, rho=rho, weight_ema=True)
    rng = np.random.Generator(np.random.PCG64(17))
    B = 6
    t = torch.from_numpy
    out = dict(seed=np.int64(seed), std=np.float64(STD), final_ln_scale=np.float64(FINAL_LN_SCALE), hyper=np.array([G, alpha, eps, ema, rho]),
               cfg=np.array([cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                             cfg.intermediate_size, cfg.max_position_embeddings, cfg.type_vocab_size]))
    groups = [np.array([0, 2, 2, 4, 0, 2]), np.array([1, 1, 3, 0, 4, 4])]  # group 3 / 1 absent in step 0, 2 in step 1
    for step in range(2):
        q_ids, q_mask = synth_batch(rng, B, 16, cfg.vocab_size)
        a_ids, a_mask = synth_batch(rng, B, 32, cfg.vocab_size)
        b_ids, b_mask = synth_batch(rng, B, 32, cfg.vocab_size)
        g = groups[step]
        model.zero_grad()
        robust, acc, group_losses, group_counts = model(t(q_ids), t(q_mask), t(a_ids), t(a_mask), t(b_ids), t(b_mask),
                                                        group_ids=t(g))
        robust.backward()
        out.update({f"s{step}_q_ids": q_ids, f"s{step}_q_mask": q_mask, f"s{step}_a_ids": a_ids, f"s{step}_a_mask": a_mask,
                    f"s{step}_b_ids": b_ids, f"s{step}_b_mask": b_mask, f"s{step}_groups": g,
                    f"s{step}_robust": np.float64(float(robust)), f"s{step}_group_losses": group_losses.numpy().astype(np.float64),
                    f"s{step}_group_counts": group_counts.numpy().astype(np.float64),
                    f"s{step}_h_fun": model.loss.h_fun.detach().numpy().astype(np.float64)})
        sg = selected_grads(model.named_parameters(), "bert.")
        out.update({f"s{step}_{k}": v for k, v in sg.items()})
        for name, p in model.named_parameters():  # a few last-layer gradients as well (the re-weighted layers)
            if name in ("bert.encoder.layer.11.output.dense.weight", "bert.encoder.layer.9.attention.self.value.weight",
                        "bert.encoder.layer.10.intermediate.dense.bias"):
                out[f"s{step}_grad:{name[5:]}"] = p.grad.detach().numpy().copy()
        print(f"idro golden step {step}: robust {float(robust):.6f} h_fun {model.loss.h_fun.numpy()}")
    np.savez_compressed(os.path.join(OUT, "idro_steps.npz"), **out)
    sys.path.pop(0)


def golden_dro_greedy():
    """f2 (second strategy): three steps of the reference's DROGreedyLoss (ANCE/model/dro_loss.py:11-126, the driver's
    default --dro_type) through BertDot_NLL_LN.forward(group_ids=..., weights=...), for both h_fun update rules
    (weight_ema False / True).  2-layer toy BERT, 1-rank gloo group for its all_gather.  Runs unmodified."""
    sys.path.insert(0, os.path.join(REF, "ANCE"))
    import types
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29535", rank=0, world_size=1)
    from model.models import BertDot_NLL_LN  # reference

    cfg = OracleConfig(vocab_size=400, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                       intermediate_size=256, max_position_embeddings=64)
    seed = 555
    P = scale_final_ln(make_params(cfg, seed, std=STD), cfg)
    G, alpha, eps, ema = 4, 0.5, 0.05, 0.3
    t = torch.from_numpy
    out = dict(seed=np.int64(seed), std=np.float64(STD), final_ln_scale=np.float64(FINAL_LN_SCALE), hyper=np.array([G, alpha, eps, ema]),
               cfg=np.array([cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                             cfg.intermediate_size, cfg.max_position_embeddings, cfg.type_vocab_size]))
    B = 6
    groups = [np.array([0, 2, 2, 3, 0, 2]), np.array([1, 1, 3, 0, 3, 3]), np.array([2, 0, 1, 1, 0, 3])]
    weights = np.array([1.0, 0.5, 2.0, 1.0, 1.5, 1.0], np.float32)
    rng = np.random.Generator(np.random.PCG64(23))
    batches = []
    for step in range(3):
        batches.append((synth_batch(rng, B, 16, cfg.vocab_size), synth_batch(rng, B, 32, cfg.vocab_size),
                        synth_batch(rng, B, 32, cfg.vocab_size)))
        (q, qm), (a, am), (b, bm) = batches[-1]
        out.update({f"s{step}_q_ids": q, f"s{step}_q_mask": qm, f"s{step}_a_ids": a, f"s{step}_a_mask": am,
                    f"s{step}_b_ids": b, f"s{step}_b_mask": bm, f"s{step}_groups": groups[step]})
    out["weights"] = weights
    for wema, tag in ((False, "hard"), (True, "ema")):
        torch.manual_seed(0)
        model = BertDot_NLL_LN(hf_config(cfg))
        load_into(model.bert, P)
        model.train()
        args = types.SimpleNamespace(model_size="base", local_rank=0)
        model.add_group_loss(args=args, n_groups=G, dro_type="dro-greedy", alpha=alpha, eps=eps, ema=ema, rho=0.1, weight_ema=wema)
        for step in range(3):
            (q, qm), (a, am), (b, bm) = batches[step]
            model.zero_grad()
            robust, acc, group_losses, group_counts = model(t(q), t(qm), t(a), t(am), t(b), t(bm), group_ids=t(groups[step]),
                                                            weights=t(weights))
            robust.backward()
            out.update({f"{tag}_s{step}_robust": np.float64(float(robust.detach())),
                        f"{tag}_s{step}_group_losses": group_losses.numpy().astype(np.float64),
                        f"{tag}_s{step}_group_counts": group_counts.numpy().astype(np.float64),
                        f"{tag}_s{step}_h_fun": model.loss.h_fun.detach().numpy().astype(np.float64),
                        f"{tag}_s{step}_sum_losses": model.loss.sum_losses.detach().numpy().astype(np.float64),
                        f"{tag}_s{step}_count_cat": model.loss.count_cat.detach().numpy().astype(np.float64)})
            if step == 1:  # h_fun is no longer uniform here: the gradient shows the re-weighting
                for name, prm in model.named_parameters():
                    if name in ("bert.encoder.layer.1.output.dense.weight", "bert.encoder.layer.0.attention.self.value.weight",
                                "bert.embeddings.LayerNorm.weight"):
                        out[f"{tag}_s1_grad:{name[5:]}"] = prm.grad.detach().numpy().copy()
            print(f"dro-greedy golden [{tag}] step {step}: robust {float(robust.detach()):.6f} h_fun {model.loss.h_fun.numpy()}")
    np.savez_compressed(os.path.join(OUT, "dro_greedy_steps.npz"), **out)
    sys.path.pop(0)


def golden_collate():
    """f3: the reference collator's own methods (COCO/data.py:44-55 word grouping, :68-99 whole-word mask, :101-117
    truncation) called unbound on a stub ``self`` (the class itself cannot be constructed offline: it needs a tokenizer
    with a vocabulary file, and its ``__call__`` uses the removed ``encode_plus``).  ``random.shuffle`` /
    ``random.randint`` are replaced by recorded permutations / offsets so the algorithm - not Python's RNG stream - is
    what the fixture pins."""
    import types
    sys.path.insert(0, os.path.join(REF, "COCO"))
    import data as coco_data  # reference
    C = coco_data.CondenserCollator
    rng = np.random.Generator(np.random.PCG64(99))
    cases = []
    for case in range(12):
        n = int(rng.integers(1, 60)) if case else 1
        sub = rng.random(n) < 0.35
        sub[0] = bool(case % 2) and n > 1 and case > 6  # a leading "##" piece starts a word of its own (:50)
        toks = [("##" if s_ else "") + f"w{i}" for i, s_ in enumerate(sub)]
        stub = types.SimpleNamespace(specials=["[CLS]", "[SEP]", "[PAD]", "[MASK]", "[UNK]"], mlm_probability=0.15 if case % 3 else 0.3)
        stub._whole_word_cand_indexes_bert = lambda t, stub=stub: C._whole_word_cand_indexes_bert(stub, t)
        groups = C._whole_word_cand_indexes_bert(stub, toks)
        order = rng.permutation(len(groups))

        def fake_shuffle(lst, order=order):
            lst[:] = [lst[i] for i in order]
        coco_data.random.shuffle = fake_shuffle
        mask = C._whole_word_mask(stub, toks)
        cases.append(dict(sub=sub.astype(np.uint8), order=order.astype(np.int64), mask=np.array(mask, np.int64),
                          prob=np.float64(stub.mlm_probability),
                          groups_flat=np.array([i for g in groups for i in g], np.int64),
                          groups_len=np.array([len(g) for g in groups], np.int64)))
    # spans that contain special tokens ([UNK], a stray [SEP]): skipped by the grouping (:47-48), never masked; own generator
    # so the twelve cases above keep their values
    rng2 = np.random.Generator(np.random.PCG64(199))
    for case in range(5):
        n = int(rng2.integers(6, 50))
        sub = rng2.random(n) < 0.35
        sub[0] = False
        spec = rng2.random(n) < 0.2
        spec[1] = True
        if case == 0:
            spec[0] = True           # the span opens with a special token
        sub[2] = True                # a "##" piece right behind a special token joins the word in FRONT of the special one
        toks = [("[UNK]" if k % 2 else "[SEP]") if spec[k] else (("##" if sub[k] else "") + f"w{k}") for k in range(n)]
        stub = types.SimpleNamespace(specials=["[CLS]", "[SEP]", "[PAD]", "[MASK]", "[UNK]"], mlm_probability=0.15 if case % 2 else 0.4)
        stub._whole_word_cand_indexes_bert = lambda t, stub=stub: C._whole_word_cand_indexes_bert(stub, t)
        groups = C._whole_word_cand_indexes_bert(stub, toks)
        order = rng2.permutation(len(groups))

        def fake_shuffle2(lst, order=order):
            lst[:] = [lst[i] for i in order]
        coco_data.random.shuffle = fake_shuffle2
        mask = C._whole_word_mask(stub, toks)
        cases.append(dict(sub=(sub & ~spec).astype(np.uint8), special=spec.astype(np.uint8), order=order.astype(np.int64),
                          mask=np.array(mask, np.int64), prob=np.float64(stub.mlm_probability),
                          groups_flat=np.array([i for g in groups for i in g], np.int64),
                          groups_len=np.array([len(g) for g in groups], np.int64)))
    out = {}
    for i, c in enumerate(cases):
        out.update({f"c{i}_{k}": v for k, v in c.items()})
    out["n_cases"] = np.int64(len(cases))
    # truncation window
    import random as _random
    tstub = types.SimpleNamespace(max_seq_length=16, tokenizer=types.SimpleNamespace(num_special_tokens_to_add=lambda pair: 2))
    ex = list(range(100, 140))
    for j, left in enumerate((0, 7, 26)):
        coco_data.random.randint = lambda a, b, left=left: left
        out[f"trunc{j}"] = np.array(C._truncate(tstub, list(ex)), np.int64)
        out[f"trunc{j}_left"] = np.int64(left)
    out["trunc_short"] = np.array(C._truncate(tstub, ex[:9]), np.int64)
    coco_data.random.shuffle, coco_data.random.randint = _random.shuffle, _random.randint
    np.savez_compressed(os.path.join(OUT, "collator_cases.npz"), **out)
    print("collator golden:", len(cases), "mask cases; truncation", out["trunc1"][:3], "...")
    sys.path.pop(0)


def golden_lamb():
    """Three steps of the reference's own ``Lamb`` (ANCE/utils/lamb.py) behind ``torch.nn.utils.clip_grad_norm_`` -
    the optimizer half of the ANCE step (ANCE/drivers/run_ann.py:345-356).  Harness shim (disclosed): lamb.py imports
    ``tensorboardX`` at module level only for its logging helper; the package is absent here, so an empty stand-in
    module with a ``SummaryWriter`` name is registered before the import.  Five tensors exercise the corner cases:
    a matrix, a vector, an all-zero tensor (trust ratio 1), a tensor with ||w|| > 10 (clamp) and a tiny one."""
    import types
    import importlib.machinery
    tb = types.ModuleType("tensorboardX")
    tb.SummaryWriter = object
    tb.__spec__ = importlib.machinery.ModuleSpec("tensorboardX", None)  # transformers probes find_spec() of optional packages
    sys.modules.setdefault("tensorboardX", tb)
    sys.path.insert(0, os.path.join(REF, "ANCE", "utils"))
    from lamb import Lamb
    sys.path.pop(0)
    g = torch.Generator().manual_seed(11)
    shapes = [(24, 16), (64,), (32,), (40, 8), (4,)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g) * 0.05) for s in shapes]
    with torch.no_grad():
        params[2].zero_()
        params[3].mul_(40.0)  # ||w|| ~ 36 -> clamped to 10
    out = {f"p0_{i}": p.detach().numpy().copy() for i, p in enumerate(params)}
    for wd, tag in ((0.0, "wd0"), (0.01, "wd01")):
        ps = [torch.nn.Parameter(p.detach().clone()) for p in params]
        opt = Lamb(ps, lr=2e-3, eps=1e-6, weight_decay=wd)
        gg = torch.Generator().manual_seed(12)
        for step in range(3):
            for i, p in enumerate(ps):
                p.grad = torch.randn(p.shape, generator=gg) * (3.0 if step == 0 else 0.02)  # step 0 is clipped, the others not
                if i == 2 and step < 2:
                    p.grad.zero_()  # zero weights AND zero update -> trust ratio 1 branch
                out[f"{tag}_g{step}_{i}"] = p.grad.numpy().copy()
            norm = torch.nn.utils.clip_grad_norm_(ps, 1.0)
            out[f"{tag}_norm{step}"] = np.float64(float(norm))
            opt.step()
            for i, p in enumerate(ps):
                out[f"{tag}_p{step + 1}_{i}"] = p.detach().numpy().copy()
            out[f"{tag}_trust{step}"] = np.array([float(opt.state[p]["trust_ratio"]) for p in ps])
    np.savez_compressed(os.path.join(OUT, "lamb_steps.npz"), **out)
    print("lamb golden: norms", [out[f"wd0_norm{i}"] for i in range(3)], "trust", out["wd0_trust0"])


def _reference_functions(path, names, namespace):
    """Compile selected top-level function definitions of a reference script that cannot be imported whole (it parses
    sys.argv, opens data files and imports faiss / pytrec_eval at module level) into ``namespace`` - the reference's own
    code objects, executed here only; nothing of their text is stored."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in keep} == set(names), (names, [n.name for n in keep])
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), namespace)
    return namespace


def golden_evaldev():
    """a18: the BEIR script's own ``EvalDevQuery`` + ``convert_to_string_id`` (evaluate/evaluation/evaluate_beir.py:89-194)
    on a seeded synthetic run with duplicated pids, unjudged passages and two ArguAna-style self matches; MS MARCO MRR from
    the reference's msmarco_eval.compute_metrics.  Harness shim (disclosed): pytrec_eval is absent, so the evaluator object
    is ``oracle.TrecEvaluatorStandIn`` - the prediction dictionary, hole rates, eval_query_cnt and ms_mrr stored here are
    reference outputs; the four trec_eval means are the stand-in's and are stored only as a cross-check."""
    from oracle import TrecEvaluatorStandIn
    sys.path.insert(0, os.path.join(REF, "evaluate", "evaluation"))
    import msmarco_eval  # reference, pure stdlib
    rng = np.random.Generator(np.random.PCG64(31))
    nq, npass, k, topN = 24, 400, 120, 100
    q2id = rng.permutation(5000)[:nq] + 1
    p2id = rng.integers(1, 260, npass)  # several vectors per document
    I = np.stack([rng.permutation(npass)[:k] for _ in range(nq)])
    qrels = {}
    for i, q in enumerate(q2id):
        walked = p2id[I[i, :topN]]
        rel = {int(p): int(rng.integers(0, 3)) for p in rng.choice(walked, 3)}  # some judged with rel 0
        rel.update({int(p): 1 for p in rng.integers(1, 260, 2)})
        qrels[int(q)] = rel
    off_q = {int(q2id[0]): "doc-a", int(q2id[1]): "doc-b"}
    off_p = {int(p2id[I[0, 0]]): "doc-a", int(p2id[I[1, 4]]): "doc-b", int(p2id[I[2, 1]]): "doc-zzz"}
    ns = {"pytrec_eval": types.SimpleNamespace(RelevanceEvaluator=TrecEvaluatorStandIn), "compute_metrics": msmarco_eval.compute_metrics,
          "offset2qchar": off_q, "offset2pchar": off_p}
    _reference_functions(os.path.join(REF, "evaluate", "evaluation", "evaluate_beir.py"), ["EvalDevQuery", "convert_to_string_id"], ns)
    (ndcg, cnt, Map, mrr, recall, hole, ms_mrr, ahole, result, prediction, mrrs, ndcgs) = ns["EvalDevQuery"](
        [int(x) for x in q2id], [int(x) for x in p2id], qrels, I, topN)
    pred_q, pred_p, pred_s = [], [], []
    for q, docs in prediction.items():
        for pid, sc in docs.items():
            pred_q.append(q); pred_p.append(pid); pred_s.append(sc)
    qr = [(q, p_, r) for q, d in qrels.items() for p_, r in d.items()]
    np.savez_compressed(os.path.join(OUT, "evaldev_beir.npz"), q2id=q2id, p2id=p2id, I=I, topN=np.int64(topN),
                        qrels=np.array(qr, np.int64), off_q=np.array(sorted(off_q), np.int64), off_q_char=np.array([off_q[k_] for k_ in sorted(off_q)]),
                        off_p=np.array(sorted(off_p), np.int64), off_p_char=np.array([off_p[k_] for k_ in sorted(off_p)]),
                        pred=np.array([pred_q, pred_p, pred_s], np.int64), n_queries=np.int64(cnt), hole_rate=np.float64(hole),
                        ahole_rate=np.float64(ahole), ms_mrr10=np.float64(ms_mrr["MRR @10"]), ms_ranked=np.int64(ms_mrr["QueriesRanked"]),
                        standin_means=np.array([ndcg, Map, mrr, recall]))
    print("evaldev golden:", cnt, "queries; hole", hole, ahole, "ms_mrr", ms_mrr, "stand-in ndcg/map/mrr/recall", ndcg, Map, mrr, recall)
    sys.path.pop(0)


def golden_negatives():
    """a19: the ANCE driver's own ``GenerateNegativePassaageID`` (ANCE/drivers/run_ann_data_gen.py:497-570), both branches:
    ``--ann_measure_topk_mrr`` and the default shuffled walk.  ``random.shuffle`` is replaced by recorded permutations (as
    for the collator fixture) so the selection rule - not Python's RNG stream - is what the fixture pins; ``trange`` is
    ``range``."""
    rng = np.random.Generator(np.random.PCG64(41))
    nq, npass, k, n_neg = 30, 500, 60, 7
    q2id = rng.permutation(9000)[:nq]
    p2id = rng.integers(0, 300, npass)
    I = np.stack([rng.permutation(npass)[:k] for _ in range(nq)])
    pos = {int(q): (int(p2id[I[i, int(rng.integers(0, k))]]) if i % 5 else 100000 + i) for i, q in enumerate(q2id)}  # every 5th positive is not retrieved
    eff = set(int(q) for i, q in enumerate(q2id) if i % 3 != 1)
    perms = [rng.permutation(k) for _ in range(nq)]
    state = {"j": 0}

    def fake_shuffle(lst):
        lst[:] = [int(x) for x in perms[state["j"]]]
        state["j"] += 1

    ns = {"np": np, "trange": range, "random": types.SimpleNamespace(shuffle=fake_shuffle), "print": lambda *a, **k_: None}
    _reference_functions(os.path.join(REF, "ANCE", "drivers", "run_ann_data_gen.py"), ["GenerateNegativePassaageID"], ns)
    out = dict(q2id=q2id, p2id=p2id, I=I, pos=np.array([[q, p_] for q, p_ in pos.items()], np.int64), eff=np.array(sorted(eff), np.int64),
               negative_sample=np.int64(n_neg), perms=np.stack(perms))
    for topk, tag in ((True, "topk"), (False, "shuffle")):
        state["j"] = 0
        args = types.SimpleNamespace(ann_measure_topk_mrr=topk, negative_sample=n_neg, rank=0)
        negs, rr = ns["GenerateNegativePassaageID"](args, [int(x) for x in q2id], [int(x) for x in p2id], pos, I, eff)
        qs = list(negs.keys())
        out[f"{tag}_qids"] = np.array(qs, np.int64)
        out[f"{tag}_negs"] = np.array([negs[q] + [-1] * (n_neg - len(negs[q])) for q in qs], np.int64)
        out[f"{tag}_rr"] = np.asarray(rr, np.float64)
        out[f"{tag}_nperm"] = np.int64(state["j"])
    np.savez_compressed(os.path.join(OUT, "hard_negatives.npz"), **out)
    print("negatives golden:", len(out["topk_qids"]), "queries;", out["shuffle_negs"][0], out["shuffle_nperm"])


if __name__ == "__main__":
    torch.set_num_threads(4)
    ALL = ["coco", "ance", "mrr", "condenser", "cache", "lamb", "idro", "dro_greedy", "collate", "evaldev", "negatives", "dropout",
           "training_rows", "coco_dataset"]
    which = sys.argv[1:]
    if not which:
        # every target in its own interpreter: the reference trees shadow each other's top-level modules (COCO/data.py vs
        # ANCE/data/, two different `model` / `utils` packages), and a target that has imported one poisons sys.modules for the next
        import subprocess
        for t in ALL:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), t])
            if r.returncode != 0:
                sys.exit(f"make_golden: target {t!r} failed with exit code {r.returncode}")
        sys.exit(0)
    unknown = [t for t in which if t not in ALL]
    if unknown:
        sys.exit(f"make_golden: unknown target(s) {unknown}; known: {ALL}")
    if "evaldev" in which:
        golden_evaldev()
    if "negatives" in which:
        golden_negatives()
    if "collate" in which:
        golden_collate()
    if "dro_greedy" in which:
        golden_dro_greedy()
    if "idro" in which:
        golden_idro()
    if "lamb" in which:
        golden_lamb()
    if "cache" in which:
        golden_token_cache()
    if "training_rows" in which:
        golden_training_rows()
    if "coco_dataset" in which:
        golden_coco_dataset()
    if "coco" in which:
        golden_coco()
    if "condenser" in which:
        golden_condenser()
    if "ance" in which:
        golden_ance()
    if "mrr" in which:
        golden_mrr()
    if "dropout" in which:
        golden_dropout()
