"""Is the packed-vs-padded gradient gap at 200 sequences (split-tail GEMM route) rounding noise?  The fp64 oracle's gradients are the
truth; the padded step, the packed step and the packed step without the split tail (COCODR_GEMM_NOTAIL=1, a second process) are
each measured against it.  Usage: packed_b200_oracle.py oracle | gpu"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import oracle as O

H, heads, I, B, L, V = 1024, 16, 4096, 200, 128, 3000
ocfg = O.OracleConfig(vocab_size=V, hidden_size=H, num_hidden_layers=2, num_attention_heads=heads, intermediate_size=I, max_position_embeddings=128)
P = O.make_params(ocfg, 21, std=0.04)
s_ln = float(np.sqrt(5.0 / H))
for k in ("weight", "bias"):
    n = f"encoder.layer.1.output.LayerNorm.{k}"
    P[n] = (P[n] * s_ln).astype(P[n].dtype)
rng = np.random.Generator(np.random.PCG64(77))
lens = np.clip(np.rint(rng.normal(0.6 * L, 0.25 * L, B)), 3, L).astype(np.int64)
lens[0] = L
ids = rng.integers(5, V, (B, L))
mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
ids = ids * mask
path = "/tmp/oracle_b200.npz"
if sys.argv[1] == "oracle":
    hs, cache = O.encoder_fwd(P, ocfg, ids, mask, keep_cache=True)
    E = O.cls_embedding(hs[-1])
    ref_loss, dE = O.contrastive_loss_grad(E.copy(), 1)
    d_last = np.zeros_like(hs[-1])
    d_last[:, 0] = dE
    G = O.encoder_bwd(P, ocfg, cache, d_last)
    np.savez(path, loss=ref_loss, **{k: np.asarray(v, np.float32) for k, v in G.items()})
    print("oracle loss", ref_loss)
else:
    import torch
    import cocodr_amd
    from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel
    ref = np.load(path)
    cfg = CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=V, hidden_size=H, num_hidden_layers=2,
                         num_attention_heads=heads, intermediate_size=I, max_position_embeddings=128, type_vocab_size=ocfg.type_vocab_size)
    out = {}
    for packed in (False, True):
        m = CocoBertModel(cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
        m = m.to("cuda")
        m.pack_sequences = packed
        model = CoCondenserForPretraining(m)
        b = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
        if packed:
            b["lengths"] = torch.from_numpy(lens)
        loss = model(b, None)
        loss.backward()
        out[packed] = (float(loss), {k: v.detach().float().cpu().numpy() for k, v in m.hf_named_grads()})
    tag = "NOTAIL" if os.environ.get("COCODR_GEMM_NOTAIL") else "tail"
    print(f"[{tag}] loss oracle {float(ref['loss']):.6f} padded {out[False][0]:.6f} packed {out[True][0]:.6f}")
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-30))
    print(f"{'tensor':58s} padded-vs-oracle packed-vs-oracle packed-vs-padded")
    worst = [0, 0, 0]
    for k in ref.files:
        if k == "loss" or k.endswith("key.bias"):
            continue
        r = (rel(out[False][1][k], ref[k]), rel(out[True][1][k], ref[k]), rel(out[True][1][k], out[False][1][k]))
        worst = [max(a, b) for a, b in zip(worst, r)]
        print(f"{k:58s} {r[0]:.2e} {r[1]:.2e} {r[2]:.2e}")
    print(f"[{tag}] worst: padded-vs-oracle {worst[0]:.2e} packed-vs-oracle {worst[1]:.2e} packed-vs-padded {worst[2]:.2e}")
