"""Kernel-level parity tests, all through the C ABI (ctypes).  Floating-point kernels are compared with
a plain fp32 torch reference of the same op on identical (bf16-rounded) inputs; tolerances are stated
per test.  Run on the GPU box:  python -m pytest tests -m gpu"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402
from cocodr_amd import _native as N  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------ hardware layout probes
def test_native_library_is_loaded():
    assert "gfx950" in ops.build_info()


def test_probe_mfma_32x32x16_layout():
    a, b = rnd(32, 16, seed=1), rnd(32, 16, seed=2)
    out = ops.probe_mfma32(a, b)
    ref = a.float() @ b.float().T  # asymmetric operands: a transposed C-write would show
    assert torch.allclose(out, ref, atol=1e-4, rtol=1e-5), (out - ref).abs().max()


def test_probe_ds_read_tr16_layout():
    tile = torch.arange(16 * 64, dtype=torch.int16, device=DEV).reshape(16, 64)
    out = ops.probe_tr16(tile.view(torch.bfloat16)).cpu().numpy()
    t = tile.cpu().numpy()
    for l in range(64):
        g, c = l >> 4, l & 15
        r0, c0 = 4 * g, 16 * ((g + 1) & 3)
        assert out[l].tolist() == [int(t[r0 + j, c0 + c]) for j in range(4)], (l, out[l])


# ------------------------------------------------------------------ GEMM
@pytest.fixture(params=[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 18, 14, 15], ids=["regstage", "glds128x64", "glds256x64", "glds128x32", "glds256x32", "glds256x32w4", "glds256x64w4", "glds128x192", "glds256x64ld4", "glds256x64w4ld4", "glds256x256", "glds256x96ld4", "pingpong256x256", "pingpong256x256fat", "onewave256x256", "onewave256x256walk"], autouse=False)
def gemm_impl(request):
    ops.gemm_set_impl(request.param)
    yield request.param
    ops.gemm_set_impl(0)


@pytest.mark.parametrize("M,Nn,K", [(256, 128, 64), (192, 384, 128), (8192, 768, 768), (1000, 256, 200), (1000, 512, 64), (300, 256, 192)])
def test_gemm_nt_bias(M, Nn, K, gemm_impl):
    a, w, bias = rnd(M, K, seed=3), rnd(Nn, K, scale=0.05, seed=4), rnd(Nn, seed=5, dtype=torch.float32)
    out = ops.gemm(a, w, bias=bias)
    ref = a.float() @ w.float().T + bias
    assert rel_l2(out, ref) < 5e-3  # bf16 output rounding only (fp32 accumulate)


def test_gemm_nt_gelu_and_residual(gemm_impl):
    M, Nn, K = 384, 512, 128
    a, w, bias, r = rnd(M, K, seed=3), rnd(Nn, K, scale=0.1, seed=4), rnd(Nn, seed=5, dtype=torch.float32), rnd(M, Nn, seed=6)
    h, gp = ops.gemm(a, w, bias=bias, epi=N.EPI_GELU)
    pre = (a.float() @ w.float().T + bias).requires_grad_(True)
    ref = torch.nn.functional.gelu(pre)
    assert rel_l2(h, ref) < 5e-3
    ref.sum().backward()  # second output: GELU'(pre-activation), what the backward multiplies by
    assert rel_l2(gp, pre.grad) < 5e-3
    pre = pre.detach()
    out = ops.gemm(a, w, bias=bias, epi=N.EPI_ADD, r=r)
    assert rel_l2(out, pre + r.float()) < 5e-3


@pytest.mark.parametrize("M,Nn,K", [(256, 128, 128), (200, 256, 384), (8192, 768, 3072)])
def test_gemm_nn_dgrad(M, Nn, K, gemm_impl):
    dy, w = rnd(M, K, seed=7), rnd(K, Nn, scale=0.05, seed=8)  # w stored [K,N] = Linear weight [out=K, in=N]
    out = ops.gemm(dy, w, trans_b=True)
    assert rel_l2(out, dy.float() @ w.float()) < 5e-3
    gp = rnd(M, Nn, seed=9)  # the saved GELU' values (any bf16 matrix: the epilogue is an element-wise product)
    out = ops.gemm(dy, w, trans_b=True, epi=N.EPI_DGELU, r=gp)
    assert rel_l2(out, (dy.float() @ w.float()) * gp.float()) < 5e-3


@pytest.mark.parametrize("M,Nn,K", [(200, 256, 384), (8192, 768, 768), (1000, 3072, 768)])
def test_gemm_fused_column_sums(M, Nn, K, gemm_impl):
    """colsum = column sums of the fp32 epilogue result (bias gradient produced next to dX), all epilogue forms."""
    dy, w, u = rnd(M, K, seed=31), rnd(K, Nn, scale=0.05, seed=32), rnd(M, Nn, seed=33)
    out, cs = ops.gemm(dy, w, trans_b=True, colsum=True)
    ref = dy.float() @ w.float()
    assert torch.equal(out, ops.gemm(dy, w, trans_b=True))
    assert (cs - ref.sum(0)).abs().max() < 2e-3 * float(ref.abs().sum(0).max())
    out, cs = ops.gemm(dy, w, trans_b=True, epi=N.EPI_DGELU, r=u, colsum=True)
    gp = u.float()
    assert (cs - (ref * gp).sum(0)).abs().max() < 2e-3 * float((ref * gp).abs().sum(0).max())


@pytest.mark.parametrize("nb,Mtok,No,Ni", [(1, 256, 128, 128), (3, 200, 384, 256), (2, 8192, 768, 768)])
def test_gemm_tn_wgrad_batched_fp32(nb, Mtok, No, Ni, gemm_impl):
    dy, x = rnd(nb, Mtok, No, seed=10), rnd(nb, Mtok, Ni, seed=11)
    out = ops.gemm(dy, x, trans_a=True, trans_b=True, out_f32=True)
    ref = torch.einsum("bmo,bmi->boi", dy.float(), x.float())
    assert out.dtype == torch.float32 and rel_l2(out, ref) < 1e-4  # fp32 out: accumulate-order noise only


def test_gemm_split_k_matches_the_single_pass():
    """the decoder-gradient shape of the MLM head: few output tiles, very long K, cut into batch items of one launch"""
    dy, w = rnd(512, 3072, seed=41), rnd(3072, 256, scale=0.05, seed=42)
    ref = dy.float() @ w.float()
    for s in (2, 4):
        assert rel_l2(ops.gemm(dy, w, trans_b=True, split_k=s), ref) < 5e-3
        assert rel_l2(ops.gemm(dy, w, trans_b=True, split_k=s, out_f32=True), ref) < 1e-4
    wt = w.t().contiguous()
    assert rel_l2(ops.gemm(dy, wt, split_k=2, out_f32=True), ref) < 1e-4
    with pytest.raises(ValueError):
        ops.gemm(dy, w, trans_b=True, split_k=5)


def test_gemm_rejects_bad_shapes():
    with pytest.raises(ValueError):
        ops.gemm(rnd(64, 64), rnd(100, 64))  # N not a multiple of 128
    with pytest.raises(ValueError):
        ops.gemm(rnd(64, 64).cpu(), rnd(128, 64))


# ------------------------------------------------------------------ attention
def ref_attention(qkv, mask, B, L, heads):
    H = heads * 64
    q, k, v = [t.reshape(B, L, heads, 64).permute(0, 2, 1, 3) for t in qkv.float().split(H, dim=1)]
    s = (q @ k.transpose(-1, -2)) / 8.0
    s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    lse = torch.logsumexp(s, -1)
    p = torch.softmax(s, -1)
    ctx = (p @ v).permute(0, 2, 1, 3).reshape(B * L, H)
    return ctx, lse


def make_mask(B, L, seed=0):
    g = np.random.Generator(np.random.PCG64(seed))
    m = np.zeros((B, L), np.int32)
    for b in range(B):
        m[b, : (L if b == 0 else int(g.integers(3, L + 1)))] = 1
    return torch.from_numpy(m).to(DEV)


@pytest.mark.parametrize("B,L,heads", [(2, 32, 2), (3, 64, 2), (2, 128, 12), (2, 256, 4), (1, 512, 2)])
def test_attention_fwd(B, L, heads):
    qkv = rnd(B * L, 3 * heads * 64, seed=12)
    mask = make_mask(B, L, seed=L)
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, heads)
    rctx, rlse = ref_attention(qkv, mask, B, L, heads)
    assert rel_l2(ctx, rctx) < 1e-2  # P and ctx are rounded to bf16
    assert torch.allclose(lse, rlse, atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("B,L,heads", [(2, 32, 2), (3, 64, 2), (2, 128, 12), (2, 256, 4), (2, 288, 2), (3, 512, 4)])
def test_attention_bwd(B, L, heads):
    H = heads * 64
    qkv = rnd(B * L, 3 * H, seed=13)
    mask = make_mask(B, L, seed=L + 1)
    dctx = rnd(B * L, H, seed=14)
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, heads)
    dqkv, part = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, heads, qk_bias=True)
    plain = ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, heads)
    assert torch.equal(plain, dqkv)
    q = qkv.float().clone().requires_grad_(True)
    rctx, _ = ref_attention(q, mask, B, L, heads)
    rctx.backward(dctx.float())
    # fused query / key bias partials: column sums of dQ | dK (four partial rows per sequence), taken from the fp32 accumulators (the stored
    # values are their bf16 roundings); the key half is an exact zero (softmax shift invariance), written as such
    part = part.view(B, 4, 2 * H).sum(1)
    stored = dqkv[:, :2 * H].float().view(B, L, 2 * H).sum(1)
    want = q.grad[:, :2 * H].view(B, L, 2 * H).sum(1)
    scale = float(want[:, :H].abs().max())
    assert rel_l2(part[:, :H], want[:, :H]) < 2e-2
    assert float((part[:, :H] - stored[:, :H]).abs().max()) < 2e-2 * scale
    assert float(part[:, H:].abs().max()) == 0.0 and float(want[:, H:].abs().max()) < 1e-4 * scale  # identically zero
    for name, sl in (("dq", slice(0, H)), ("dk", slice(H, 2 * H)), ("dv", slice(2 * H, 3 * H))):
        assert rel_l2(dqkv[:, sl], q.grad[:, sl]) < 2e-2, name


# ------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("M,H", [(64, 128), (1000, 768), (515, 1024)])
def test_layernorm_fwd_bwd(M, H):
    y, g, b = rnd(M, H, seed=15), 1 + 0.1 * rnd(H, seed=16, dtype=torch.float32), rnd(H, seed=17, dtype=torch.float32)
    out, mean, rstd = ops.ln_fwd(y, g, b)
    yf = y.float().requires_grad_(True)
    gf, bf = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(yf, (H,), gf, bf, 1e-12)
    assert rel_l2(out, ref) < 5e-3
    assert torch.allclose(mean, yf.mean(-1), atol=1e-5)
    dout = rnd(M, H, seed=18)
    ref.backward(dout.float())
    dy, dg, db = ops.ln_bwd(dout, y, g, mean, rstd)
    assert rel_l2(dy, yf.grad) < 5e-3
    assert rel_l2(dg, gf.grad) < 1e-4 and rel_l2(db, bf.grad) < 1e-4
    # fused column sums of dy (the bias gradient of the Linear in front of the LayerNorm)
    dy2, dg2, db2, cs = ops.ln_bwd(dout, y, g, mean, rstd, colsum=True)
    assert torch.equal(dy2, dy) and torch.equal(dg2, dg) and torch.equal(db2, db)
    ref_cs = yf.grad.sum(0)
    assert (cs - ref_cs).abs().max() < 1e-3 * max(1.0, float(ref_cs.abs().max()))


def test_layernorm_cls_rows_fp32():
    L, B, H = 32, 5, 768
    y, g, b = rnd(B * L, H, seed=19), 1 + 0.1 * rnd(H, seed=20, dtype=torch.float32), rnd(H, seed=21, dtype=torch.float32)
    out, mean, rstd, cls = ops.ln_fwd(y, g, b, cls_stride=L)
    ref = torch.nn.functional.layer_norm(y.float(), (H,), g, b, 1e-12)[::L]
    assert cls.shape == (B, H) and torch.allclose(cls, ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("B,L,H,V", [(3, 32, 128, 1000), (16, 128, 768, 30522)])
def test_embedding_ln_fwd_bwd(B, L, H, V):
    g = torch.Generator().manual_seed(22)
    ids = torch.randint(0, V, (B, L), generator=g, dtype=torch.int32)
    ids[:, 0] = 1  # repeated ids -> scatter-add collisions
    ids = ids.to(DEV)
    word, pos, typ = rnd(V, H, scale=0.05, seed=23, dtype=torch.float32), rnd(512, H, scale=0.05, seed=24, dtype=torch.float32), \
        rnd(2, H, scale=0.05, seed=25, dtype=torch.float32)
    gam, bet = 1 + 0.1 * rnd(H, seed=26, dtype=torch.float32), rnd(H, seed=27, dtype=torch.float32)
    out, mean, rstd = ops.embed_ln_fwd(ids, word, pos, typ[0].contiguous(), gam, bet)
    wf, pf, tf = word.clone().requires_grad_(True), pos.clone().requires_grad_(True), typ.clone().requires_grad_(True)
    gf, bf = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(wf[ids.long()] + pf[None, :L] + tf[0], (H,), gf, bf, 1e-12).reshape(B * L, H)
    assert rel_l2(out, ref) < 5e-3
    dout = rnd(B * L, H, seed=28)
    ref.backward(dout.float())
    dword, dpos, dtype0, dgam, dbet = ops.embed_ln_bwd(dout, ids, word, pos, typ[0].contiguous(), gam, mean, rstd)
    assert rel_l2(dword, wf.grad) < 1e-4  # fp32 atomics: order noise only
    assert rel_l2(dpos, pf.grad) < 1e-4 and rel_l2(dtype0, tf.grad[0]) < 1e-4
    assert rel_l2(dgam, gf.grad) < 1e-4 and rel_l2(dbet, bf.grad) < 1e-4


def test_colsum_and_cast():
    x = rnd(3, 1000, 384, seed=29)
    assert rel_l2(ops.colsum(x), x.float().sum(1)) < 1e-5
    x2 = rnd(8192, 3072, seed=30)
    assert rel_l2(ops.colsum(x2), x2.float().sum(0)) < 1e-5
    src = rnd(12345, seed=31, dtype=torch.float32)
    assert torch.equal(ops.cast_f32_bf16(src), src.to(torch.bfloat16))  # bit-exact round-to-nearest-even
    dE = rnd(4, 768, seed=32, dtype=torch.float32)
    d_last = ops.scatter_cls_grad(dE, 32)
    assert torch.equal(d_last[::32], dE.to(torch.bfloat16)) and float(d_last.float().abs().sum()) == float(d_last[::32].float().abs().sum())


def test_flat_adamw_matches_torch_adamw_and_updates_shadow():
    from cocodr_amd.modeling import CocoBertConfig, CocoBertModel
    from cocodr_amd.optim import FlatAdamW
    cfg = CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         max_position_embeddings=64)
    torch.manual_seed(0)
    m = CocoBertModel(cfg).to(DEV)
    ref = [p.detach().clone().requires_grad_(True) for p in (m.flat_decay, m.flat_nodecay)]
    opt = FlatAdamW.for_model(m, lr=3e-3, weight_decay=0.01)
    ropt = torch.optim.AdamW([{"params": [ref[0]], "weight_decay": 0.01}, {"params": [ref[1]], "weight_decay": 0.0}], lr=3e-3)
    g = torch.Generator(device="cpu").manual_seed(1)
    for step in range(4):
        for p, r in zip((m.flat_decay, m.flat_nodecay), ref):
            grad = (torch.randn(p.shape, generator=g) * 0.1).to(DEV)
            p.grad = grad.clone()
            r.grad = grad.clone()
        opt.step()
        ropt.step()
    for p, r in zip((m.flat_decay, m.flat_nodecay), ref):
        assert torch.allclose(p, r, rtol=1e-5, atol=1e-7), float((p - r).abs().max())  # fp32 op-order differences only
    lo = m.layout
    assert torch.equal(m._shadow, m.flat_decay.data[lo.mat_begin:].to(torch.bfloat16))
    assert not m._shadow_stale()  # the optimizer pass wrote the shadow and marked it fresh


def test_flat_lamb_and_device_clip_match_reference_lamb_golden():
    """The native LAMB + clip coefficient on a flat parameter holding the golden's five tensors (with alignment padding
    between them) reproduces three steps of the reference's own Lamb class behind clip_grad_norm_."""
    import os
    from cocodr_amd.optim import FlatLamb, clip_grad_norm_
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lamb_steps.npz"))
    shapes = [z[f"p0_{i}"].shape for i in range(5)]
    offs, o = [], 0
    for s in shapes:
        offs.append(o)
        o = (o + int(np.prod(s)) + 63) // 64 * 64  # padded like the model's flat layout
    for wd, tag in ((0.0, "wd0"), (0.01, "wd01")):
        flat = torch.zeros(o, dtype=torch.float32, device=DEV)
        for i, s in enumerate(shapes):
            flat[offs[i]: offs[i] + int(np.prod(s))] = torch.from_numpy(z[f"p0_{i}"].ravel()).to(DEV)
        p = torch.nn.Parameter(flat)
        opt = FlatLamb([p], [offs], lr=2e-3, eps=1e-6, weight_decay=wd)
        for step in range(3):
            g = torch.zeros(o, dtype=torch.float32, device=DEV)
            for i, s in enumerate(shapes):
                g[offs[i]: offs[i] + int(np.prod(s))] = torch.from_numpy(z[f"{tag}_g{step}_{i}"].ravel()).to(DEV)
            p.grad = g
            clip = clip_grad_norm_([p], 1.0)
            opt.step(clip=clip)
            norm, coef = (float(x) for x in clip)
            ref_norm = float(z[f"{tag}_norm{step}"])
            assert abs(norm - ref_norm) <= 1e-5 * ref_norm and abs(coef - min(1.0, 1.0 / (ref_norm + 1e-6))) < 1e-6
            np.testing.assert_allclose(opt.state[p]["trust_ratio"].cpu().numpy(), z[f"{tag}_trust{step}"], rtol=5e-4)
            for i, s in enumerate(shapes):
                got = p.data[offs[i]: offs[i] + int(np.prod(s))].cpu().numpy().reshape(s)
                np.testing.assert_allclose(got, z[f"{tag}_p{step + 1}_{i}"], rtol=2e-4, atol=5e-7)
        # padding between the tensors never moves
        mask = torch.ones(o, dtype=torch.bool, device=DEV)
        for i, s in enumerate(shapes):
            mask[offs[i]: offs[i] + int(np.prod(s))] = False
        assert float(p.data[mask].abs().max()) == 0.0


def test_flat_lamb_on_the_model_matches_the_oracle_per_tensor():
    """FlatLamb.for_model: trust ratios per HF-named tensor inside the two flats, bf16 shadow refreshed in the pass."""
    import oracle as O
    from cocodr_amd.modeling import CocoBertConfig, CocoBertModel
    from cocodr_amd.optim import FlatLamb, clip_grad_norm_
    cfg = CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         max_position_embeddings=64)
    torch.manual_seed(0)
    m = CocoBertModel(cfg).to(DEV)
    names = list(m.layout.names)
    ref = {n: m.hf_view(n).detach().cpu().numpy().astype(np.float64).copy() for n in names}
    rm = {n: np.zeros_like(v) for n, v in ref.items()}
    rv = {n: np.zeros_like(v) for n, v in ref.items()}
    opt = FlatLamb.for_model(m, lr=1e-3, eps=1e-6)
    gen = torch.Generator(device="cpu").manual_seed(3)
    for step in range(3):
        grads = {}
        for p in (m.flat_decay, m.flat_nodecay):
            p.grad = torch.zeros_like(p)
        for n in names:
            gview = m.layout.view((m.flat_decay.grad, m.flat_nodecay.grad), n)
            gn = torch.randn(gview.shape, generator=gen) * (0.5 if step == 0 else 0.01)
            gview.copy_(gn.to(DEV))
            grads[n] = gn.numpy().astype(np.float64)
        clip = clip_grad_norm_([m.flat_decay, m.flat_nodecay], 1.0)
        opt.step(clip=clip)
        norm, coef = O.clip_grad_norm(list(grads.values()), 1.0)
        assert abs(float(clip[0]) - norm) <= 1e-5 * norm
        O.lamb_step([ref[n] for n in names], [grads[n] * coef for n in names], [rm[n] for n in names], [rv[n] for n in names],
                    lr=1e-3, eps=1e-6)
    for n in names:
        np.testing.assert_allclose(m.hf_view(n).detach().cpu().numpy(), ref[n], rtol=3e-4, atol=1e-6, err_msg=n)
    lo = m.layout
    assert torch.equal(m._shadow, m.flat_decay.data[lo.mat_begin:].to(torch.bfloat16))


def test_one_pass_lamb_equals_the_two_pass_kernels_and_the_oracle_on_matrix_sized_tensors():
    """cocodr_lamb_step_fused (the weight matrices in ONE pass: w and u stay in registers between the norm and the update) against the
    two-pass kernels on the same flat - BERT-large's matrix shapes, a tensor with a ragged tail, small tensors and an over-sized one
    that stay on the two-pass path - and against the fp64 oracle; three steps with clipping and weight decay; the rendezvous never
    timed out."""
    import oracle as O
    from cocodr_amd.optim import FlatLamb, LAMB_FUSED_MIN, clip_grad_norm_
    from cocodr_amd._native import lib
    cap = int(lib().cocodr_lamb_fused_capacity())
    assert cap >= 4096 * 1024, cap  # BERT-large's FFN matrices fit
    sizes = [1024, 3072 * 1024, 1024 * 1024, 64, 4096 * 1024, 4096 * 1024 - 12, LAMB_FUSED_MIN, LAMB_FUSED_MIN - 64, cap + 4, 300 * 1024 + 20]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o = (o + n + 63) // 64 * 64
    g = torch.Generator().manual_seed(4)
    p0 = torch.randn(o, generator=g) * 0.05
    for i, n in enumerate(sizes):  # padding between tensors holds zeros
        p0[offs[i] + n: (offs + [o])[i + 1]] = 0
    p0[offs[3]: offs[3] + 64] = 0  # a tensor of zeros: trust ratio 1
    runs = {}
    for one_pass in (True, False):
        p = torch.nn.Parameter(p0.clone().to(DEV))
        opt = FlatLamb([p], [offs], lr=2e-3, eps=1e-6, weight_decay=0.01)
        opt.one_pass = one_pass
        gg = torch.Generator().manual_seed(5)
        grads = []
        for step in range(3):
            gr = torch.randn(o, generator=gg) * (1e-3 if step else 2e-5)
            for i, n in enumerate(sizes):
                gr[offs[i] + n: (offs + [o])[i + 1]] = 0
            grads.append(gr)
            p.grad = gr.to(DEV)
            opt.step(clip=clip_grad_norm_([p], 1.0))
        assert not opt.one_pass_error()
        runs[one_pass] = (p.data.clone(), opt.state[p]["trust_ratio"].clone(), opt.state[p]["weight_norm"].clone(), opt.state[p]["adam_norm"].clone(),
                          opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone())
    # which tensors took the one-pass kernel
    plan, fplan = opt._plan(p, offs)[:2]
    p1 = torch.nn.Parameter(p0.clone().to(DEV))
    o1 = FlatLamb([p1], [offs], lr=2e-3, eps=1e-6)
    assert o1._plan(p1, offs)[1].nfused == 6  # the four matrices, the 2^18 tensor and the 300 K one
    a, b = runs[True], runs[False]
    assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5])                    # m, v: the same arithmetic per element
    for k in (1, 2, 3):
        torch.testing.assert_close(a[k], b[k], rtol=2e-6, atol=0)               # norms: another summation order
    torch.testing.assert_close(a[0], b[0], rtol=1e-6, atol=1e-9)
    # oracle (fp64), tensor by tensor
    P = [p0[offs[i]: offs[i] + n].double().numpy().copy() for i, n in enumerate(sizes)]
    M = [np.zeros_like(x) for x in P]
    V = [np.zeros_like(x) for x in P]
    for step in range(3):
        G = [grads[step][offs[i]: offs[i] + n].double().numpy() for i, n in enumerate(sizes)]
        _, coef = O.clip_grad_norm(G, 1.0)
        trust = O.lamb_step(P, [x * coef for x in G], M, V, lr=2e-3, eps=1e-6, weight_decay=0.01)
    np.testing.assert_allclose(a[1].cpu().numpy(), np.asarray(trust), rtol=2e-4)
    for i, n in enumerate(sizes):
        np.testing.assert_allclose(a[0][offs[i]: offs[i] + n].cpu().numpy(), P[i], rtol=3e-4, atol=1e-7, err_msg=str(i))


def test_one_pass_lamb_error_flag_skips_the_update_and_falls_back():
    """a one-pass grid that is not co-resident must not leave wrongly scaled weights behind, and a training step must
    notice.  The flag word is raised by hand here (a device that holds the grid never raises it): the one-pass tensors keep their
    weights in that step (trust ratio 0), FlatLamb.step reads the flag on its first step, warns and stays on the two-pass kernels."""
    import warnings
    from cocodr_amd.optim import FlatLamb, LAMB_FUSED_MIN
    n = 4 * LAMB_FUSED_MIN
    p = torch.nn.Parameter((torch.randn(n + 64, generator=torch.Generator().manual_seed(1)) * 0.05).to(DEV))
    opt = FlatLamb([p], [[0, n]], lr=1e-2, eps=1e-6)
    assert opt.one_pass
    plan, fplan, _keep, _ws, _stats, fws, _t0 = opt._plan(p, [0, n])
    assert fplan is not None and fplan.nfused == 1
    from cocodr_amd._native import lib
    fws[int(lib().cocodr_lamb_fused_error_index(fplan.nfused))] = torch.tensor([1], dtype=torch.int32).view(torch.float32).item()
    before = p.data.clone()
    p.grad = torch.randn(n + 64, generator=torch.Generator().manual_seed(2)).to(DEV) * 1e-2
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        opt.step()
    assert any("co-resident" in str(x.message) for x in w) and opt.one_pass is False
    assert torch.equal(p.data[:n], before[:n])           # the one-pass tensor skipped its update
    assert not torch.equal(p.data[n:], before[n:])       # the small tensor behind it went through the two-pass kernels
    opt.step()                                           # two-pass from here on
    assert not torch.equal(p.data[:n], before[:n])


def test_one_pass_lamb_on_a_bert_base_width_model_equals_the_two_pass_step():
    """FlatLamb.for_model at BERT-base width (one layer): the six weight matrices (Wq, Wk, Wv are tensors of their own) AND the position table (393 K elements, in
    front of the bf16 shadow's range: the one-pass kernel must not write a shadow for it) take the one-pass kernel, the word table
    (too large here? no: 1000 x 768 - it is fused as well), vectors and the token-type table the two-pass kernels; parameters, optimizer
    state and the bf16 shadow after three clipped steps equal the all-two-pass run."""
    import copy
    from cocodr_amd.modeling import CocoBertConfig, CocoBertModel
    from cocodr_amd.optim import FlatLamb, clip_grad_norm_
    cfg = CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=1000, hidden_size=768, num_hidden_layers=1,
                         num_attention_heads=12, intermediate_size=3072, max_position_embeddings=512)
    torch.manual_seed(0)
    models = [CocoBertModel(cfg).to(DEV)]
    models.append(copy.deepcopy(models[0]))
    outs = []
    for m, one_pass in zip(models, (True, False)):
        opt = FlatLamb.for_model(m, lr=1e-3, eps=1e-6, weight_decay=0.01)
        opt.one_pass = one_pass
        gen = torch.Generator().manual_seed(3)
        for step in range(3):
            for p in (m.flat_decay, m.flat_nodecay):
                p.grad = (torch.randn(p.shape, generator=gen) * (0.3 if step == 0 else 0.01)).to(DEV)
            opt.step(clip=clip_grad_norm_([m.flat_decay, m.flat_nodecay], 1.0))
        assert not opt.one_pass_error()
        nf = [pl[1].nfused if pl[1] is not None else 0 for pl in opt._plans.values()]
        outs.append((m.flat_decay.data.clone(), m.flat_nodecay.data.clone(), m._shadow.clone(), opt.state[m.flat_decay]["exp_avg"].clone(),
                     opt.state[m.flat_decay]["trust_ratio"].clone(), sorted(nf)))
    assert outs[0][5] == [0, 8] and outs[1][5] == [0, 0]  # word + position tables, Wq, Wk, Wv, Wo, W1, W2 in one pass; the vector flat never
    a, b = outs
    assert torch.equal(a[3], b[3])                                     # m: the same arithmetic per element
    torch.testing.assert_close(a[4], b[4], rtol=2e-6, atol=0)          # trust ratios: another summation order
    torch.testing.assert_close(a[0], b[0], rtol=1e-6, atol=1e-9)
    assert torch.equal(a[1], b[1])                                     # vectors: two-pass both times
    lo = models[0].layout
    assert torch.equal(a[2], a[0][lo.mat_begin:].to(torch.bfloat16))   # the shadow follows the one-pass update, over its own range only
    assert float((a[2].float() - b[2].float()).abs().max()) <= 2.0 ** -7 * float(b[2].float().abs().max())


@pytest.mark.parametrize("layers,M,H,I", [(6, 2048, 768, 3072), (2, 1024, 256, 512), (3, 4096, 1024, 4096), (12, 1024, 768, 3072)])
def test_gemm_multi_weight_gradients_in_one_launch(layers, M, H, I):
    """cocodr_gemm_multi: the four weight-gradient problems of a layer range (different shapes, one contraction length) as one
    launch - the same numbers as four cocodr_gemm calls (bit for bit: same pipeline, same K order) and as fp32 torch.  The
    smallest case stays below the merged pipeline's tile threshold and takes the four plain calls."""
    import ctypes as C
    from cocodr_amd import _native as N
    from cocodr_amd._native import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(layers * M)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    shapes = [(3 * H, H), (H, H), (I, H), (H, I)]  # (out features, in features) of Wqkv, Wo, W1, W2
    dys = [mk(layers, M, o) for o, _ in shapes]
    xs = [mk(layers, M, i) for _, i in shapes]
    outs = [torch.empty(layers, o, i, dtype=torch.float32, device=DEV) for o, i in shapes]
    outs1 = [torch.empty_like(t) for t in outs]

    def problem(dy, x, out):
        o, i = out.shape[1], out.shape[2]
        a = N.GemmArgs()
        a.A, a.B, a.C = ptr(dy), ptr(x), ptr(out)
        a.M, a.N, a.K, a.lda, a.ldb, a.ldc = o, i, M, o, i, i
        a.trans_a = a.trans_b = a.out_f32 = 1
        a.epi, a.batch = N.EPI_NONE, layers
        a.strideA, a.strideB, a.strideC = M * o, M * i, o * i
        return a

    arr = (N.GemmArgs * 4)(*[problem(d, x, o) for d, x, o in zip(dys, xs, outs)])
    nws = lib().cocodr_gemm_multi_workspace_floats()
    ws = torch.empty(nws, dtype=torch.float32, device=DEV)
    check(lib().cocodr_gemm_multi(arr, 4, ptr(ws), nws, stream_ptr()), "gemm_multi")
    for d, x, o in zip(dys, xs, outs1):
        a = problem(d, x, o)
        check(lib().cocodr_gemm(C.byref(a), stream_ptr()), "gemm")
    for d, x, o, o1 in zip(dys, xs, outs, outs1):
        ref = torch.bmm(d.float().transpose(1, 2), x.float())
        assert float((o - ref).norm() / ref.norm()) < 2e-3
        assert float((o - o1).norm() / ref.norm()) < 1e-6  # (equal up to the pipelines' tile shapes; both accumulate K in order)
    # a problem outside the merged form (bf16 result) makes the call fall back to plain launches - same results
    out16 = torch.empty(layers, shapes[1][0], shapes[1][1], dtype=torch.bfloat16, device=DEV)
    a0, a1 = problem(dys[0], xs[0], outs[0]), problem(dys[1], xs[1], out16)
    a1.out_f32 = 0
    outs[0].zero_()
    arr2 = (N.GemmArgs * 2)(a0, a1)
    check(lib().cocodr_gemm_multi(arr2, 2, None, 0, stream_ptr()), "gemm_multi(fallback)")
    assert float((outs[0] - outs1[0]).abs().max()) == 0.0
    assert float((out16.float() - outs1[1]).norm() / outs1[1].norm()) < 5e-3
    assert lib().cocodr_gemm_multi(arr2, 5, None, 0, stream_ptr()) != 0  # more than four problems
    # without a workspace the merged launch runs whole tiles only - the same numbers up to the summation order of the cut tiles
    outs2 = [torch.empty_like(t) for t in outs]
    arr3 = (N.GemmArgs * 4)(*[problem(d, x, o) for d, x, o in zip(dys, xs, outs2)])
    check(lib().cocodr_gemm_multi(arr3, 4, None, 0, stream_ptr()), "gemm_multi(no workspace)")
    check(lib().cocodr_gemm_multi(arr, 4, ptr(ws), nws, stream_ptr()), "gemm_multi")
    for o, o2 in zip(outs, outs2):
        assert float((o - o2).norm() / o2.norm()) < 1e-6


def test_row_gather_scatter_and_bf16_product_are_exact():
    """cocodr_gather_rows / cocodr_scatter_rows / cocodr_mul_bf16 (the label-sparse MLM head's plumbing) against torch indexing:
    copies are bit-exact, the fp32 scatter-add adds exactly the bf16 values, the product is one fp32 multiply rounded once."""
    g = torch.Generator().manual_seed(4)
    M, H, n = 700, 768, 131
    src = torch.randn(M, H, generator=g).to(torch.bfloat16).to(DEV)
    idx = torch.randperm(M, generator=g)[:n].to(DEV)
    got = ops.gather_rows(src, idx)
    assert torch.equal(got, src.index_select(0, idx))
    rows = torch.randn(n, H, generator=g).to(torch.bfloat16).to(DEV)
    dst = torch.zeros(M, H, dtype=torch.bfloat16, device=DEV)
    ops.scatter_rows(rows, idx, dst)
    want = torch.zeros_like(dst)
    want.index_copy_(0, idx, rows)
    assert torch.equal(dst, want)
    acc = torch.randn(M, H, generator=g).to(DEV)
    ref = acc.clone()
    ref.index_add_(0, idx, rows.float())
    ops.scatter_rows(rows, idx, acc)
    assert torch.equal(acc, ref)
    a = torch.randn(n, H, generator=g).to(torch.bfloat16).to(DEV)
    assert torch.equal(ops.mul_bf16(a, rows), (a.float() * rows.float()).to(torch.bfloat16))
    with pytest.raises(ValueError):
        ops.gather_rows(src, idx.to(torch.int32))
