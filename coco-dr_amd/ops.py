"""Tensor-level wrappers over the C ABI (one function per kernel entry point).  torch supplies device
memory and the current stream only; all arithmetic happens in libcocodr_hip.so.  Shape / dtype /
contiguity problems raise ValueError here, mirroring where ATen would raise for the reference."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _native as N
from ._native import check, lib, ptr, stream_ptr

BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64


def _req(t: torch.Tensor, dtype, name: str, dims: Optional[int] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU (got {t.device}); this path has no CPU fallback")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if dims is not None and t.dim() != dims:
        raise ValueError(f"{name}: expected {dims} dims, got shape {tuple(t.shape)}")
    return t


def build_info() -> str:
    return lib().cocodr_build_info().decode()


# ----------------------------------------------------------------------------------------------- dropout masks
KIND_ATTN_PROBS, KIND_ATTN_OUT, KIND_FFN_OUT, KIND_EMBED = 0, 1, 2, 3


def dropout_mask(p: float, seed: int, call: int, layer: int, kind: int) -> N.DropoutMask:
    """Keys / threshold / scale of one dropout site of one forward call (cocodr_dropout_mask_for; host arithmetic only)."""
    dm = N.DropoutMask()
    check(lib().cocodr_dropout_mask_for(float(p), int(seed) & (2 ** 64 - 1), int(call) & (2 ** 64 - 1), int(layer), int(kind), C.byref(dm)),
          "dropout_mask_for")
    return dm


def _dm_ref(drop: Optional[N.DropoutMask]):
    return C.byref(drop) if drop is not None else None


# ----------------------------------------------------------------------------------------------- GEMM
def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False,
         bias: Optional[torch.Tensor] = None, epi: int = N.EPI_NONE, r: Optional[torch.Tensor] = None,
         out_f32: bool = False, out: Optional[torch.Tensor] = None, colsum: bool = False, split_k: int = 1,
         drop: Optional[N.DropoutMask] = None, split_ws: Optional[torch.Tensor] = None):
    """C = epi(op(a) @ op(b)); a, b bf16 2-D (or 3-D batched with equal batch).  See cocodr_gemm.
    colsum=True (unbatched) also returns the fp32 column sums of C as the last element of the result tuple.
    split_k > 1 (plain 2-D product, trans_a=False): the contraction is cut into split_k slices that run as the batch items of
    ONE launch (fp32 partial products, summed afterwards) - for few-tile outputs over a very long K, which would otherwise
    leave most CUs idle."""
    if split_k > 1:
        return _gemm_split_k(a, b, trans_b, out_f32, split_k)
    batched = a.dim() == 3
    _req(a, BF16, "a", 3 if batched else 2)
    _req(b, BF16, "b", 3 if batched else 2)
    nb = a.shape[0] if batched else 1
    a2, b2 = (a[0], b[0]) if batched else (a, b)
    if batched and b.shape[0] != nb:
        raise ValueError("gemm: batch mismatch")
    M, K = (a2.shape[1], a2.shape[0]) if trans_a else (a2.shape[0], a2.shape[1])
    Nn, Kb = (b2.shape[1], b2.shape[0]) if trans_b else (b2.shape[0], b2.shape[1])
    if K != Kb:
        raise ValueError(f"gemm: contraction mismatch {K} vs {Kb}")
    shape = (nb, M, Nn) if batched else (M, Nn)
    if out is None:
        out = torch.empty(shape, dtype=F32 if out_f32 else BF16, device=a.device)
    else:
        _req(out, F32 if out_f32 else BF16, "out")
        if tuple(out.shape) != shape:
            raise ValueError("gemm: out shape mismatch")
    c2 = torch.empty(shape, dtype=BF16, device=a.device) if epi == N.EPI_GELU else None
    g = N.GemmArgs()
    g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    g.C2 = c2.data_ptr() if c2 is not None else None
    if bias is not None:
        _req(bias, F32, "bias")
        if bias.shape[-1] != Nn:
            raise ValueError("gemm: bias length mismatch")
        g.bias = bias.data_ptr()
        g.strideBias = Nn if bias.dim() == 2 else 0
    if r is not None:
        _req(r, BF16, "r")
        if tuple(r.shape) != shape:
            raise ValueError("gemm: r shape mismatch")
        g.R = r.data_ptr()
        g.ldr = Nn
        g.strideR = M * Nn
    g.M, g.N, g.K = M, Nn, K
    g.lda, g.ldb, g.ldc = a2.shape[1], b2.shape[1], Nn
    g.trans_a, g.trans_b, g.epi, g.out_f32 = int(trans_a), int(trans_b), int(epi), int(out_f32)
    g.batch = nb
    g.strideA, g.strideB, g.strideC = a2.numel(), b2.numel(), M * Nn
    if drop is not None:  # EPI_ADD: out = dropout(a @ b + bias) + r
        g.drop = drop
    if split_ws is not None:  # fp32 workspace: the 256 x 256-tile pipeline may cut its last partial round into contraction slices
        _req(split_ws, F32, "split_ws")
        g.split_ws, g.split_ws_floats = split_ws.data_ptr(), split_ws.numel()
    cs = None
    if colsum:
        if batched:
            raise ValueError("gemm: colsum needs an unbatched call")
        cs = torch.empty(Nn, dtype=F32, device=a.device)
        part = torch.empty(lib().cocodr_gemm_colsum_partial_floats(M, Nn), dtype=F32, device=a.device)
        g.colsum, g.colsum_partial = cs.data_ptr(), part.data_ptr()
    check(lib().cocodr_gemm(C.byref(g), stream_ptr()), "gemm")
    res = (out, c2) if epi == N.EPI_GELU else (out,)
    if colsum:
        res = res + (cs,)
    return res if len(res) > 1 else res[0]


def _gemm_split_k(a, b, trans_b: bool, out_f32: bool, s: int):
    _req(a, BF16, "a", 2)
    _req(b, BF16, "b", 2)
    M, K = a.shape
    Nn, Kb = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if K != Kb or K % (64 * s) != 0:
        raise ValueError(f"gemm(split_k={s}): K={K} must match and be a multiple of {64 * s}")
    Ks = K // s
    part = torch.empty((s, M, Nn), dtype=F32, device=a.device)
    g = N.GemmArgs()
    g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), part.data_ptr()
    g.M, g.N, g.K = M, Nn, Ks
    g.lda, g.ldb, g.ldc = K, b.shape[1], Nn
    g.trans_a, g.trans_b, g.epi, g.out_f32 = 0, int(trans_b), N.EPI_NONE, 1
    g.batch = s
    g.strideA, g.strideB, g.strideC = Ks, (Ks * Nn if trans_b else Ks), M * Nn  # slice z: columns z*Ks.. of a, rows / columns of b
    check(lib().cocodr_gemm(C.byref(g), stream_ptr()), "gemm(split_k)")
    out = part.sum(0)
    return out if out_f32 else out.to(BF16)


def gemm_set_impl(impl: int) -> None:
    check(lib().cocodr_gemm_set_impl(int(impl)), "gemm_set_impl")


# ----------------------------------------------------------------------------------------------- attention
def attn_fwd(qkv: torch.Tensor, mask: torch.Tensor, B: int, L: int, heads: int,
             drop: Optional[N.DropoutMask] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    _req(qkv, BF16, "qkv", 2)
    _req(mask, I32, "mask", 2)
    H = heads * 64
    if tuple(qkv.shape) != (B * L, 3 * H) or tuple(mask.shape) != (B, L):
        raise ValueError(f"attn_fwd: qkv {tuple(qkv.shape)} / mask {tuple(mask.shape)} do not match B={B} L={L} heads={heads}")
    ctx = torch.empty((B * L, H), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, heads, L), dtype=F32, device=qkv.device)
    check(lib().cocodr_attn_fwd_drop(ptr(qkv), ptr(mask), ptr(ctx), ptr(lse), B, L, heads, _dm_ref(drop), stream_ptr()), "attn_fwd")
    return ctx, lse


def attn_bwd(qkv, mask, ctx, dctx, lse, B: int, L: int, heads: int, qk_bias: bool = False, drop: Optional[N.DropoutMask] = None):
    """``qk_bias=True`` also returns the [4 B, 2H] fp32 partial column sums of dQ | dK (four rows per sequence; their sum over
    all rows is the query / key bias gradient)."""
    H = heads * 64
    _req(qkv, BF16, "qkv", 2); _req(mask, I32, "mask", 2); _req(ctx, BF16, "ctx", 2); _req(dctx, BF16, "dctx", 2)
    _req(lse, F32, "lse", 3)
    if tuple(qkv.shape) != (B * L, 3 * H) or tuple(ctx.shape) != (B * L, H) or tuple(dctx.shape) != (B * L, H):
        raise ValueError("attn_bwd: shape mismatch")
    dqkv = torch.empty_like(qkv)
    part = torch.empty((4 * B, 2 * H), dtype=F32, device=qkv.device) if qk_bias else None
    check(lib().cocodr_attn_bwd_drop(ptr(qkv), ptr(mask), ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), ptr(part) if qk_bias else None,
                                     B, L, heads, _dm_ref(drop), stream_ptr()), "attn_bwd")
    return (dqkv, part) if qk_bias else dqkv


# ----------------------------------------------------------------------------------------------- row kernels
def embed_ln_fwd(ids, word, pos, type0, gamma, beta, eps: float = 1e-12, drop: Optional[N.DropoutMask] = None):
    _req(ids, I32, "ids", 2)
    for t, n in ((word, "word"), (pos, "pos"), (type0, "type0"), (gamma, "gamma"), (beta, "beta")):
        _req(t, F32, n)
    B, L = ids.shape
    V, H = word.shape
    if L > pos.shape[0]:
        raise ValueError(f"embed_ln_fwd: L={L} exceeds max_position_embeddings={pos.shape[0]}")
    out = torch.empty((B * L, H), dtype=BF16, device=ids.device)
    mean = torch.empty(B * L, dtype=F32, device=ids.device)
    rstd = torch.empty_like(mean)
    check(lib().cocodr_embed_ln_fwd_drop(ptr(ids), ptr(word), ptr(pos), ptr(type0), ptr(gamma), ptr(beta), ptr(out), ptr(mean),
                                         ptr(rstd), B, L, H, V, eps, _dm_ref(drop), stream_ptr()), "embed_ln_fwd")
    return out, mean, rstd


def embed_ln_bwd(dout, ids, word, pos, type0, gamma, mean, rstd, drop: Optional[N.DropoutMask] = None):
    _req(dout, BF16, "dout", 2); _req(ids, I32, "ids", 2)
    B, L = ids.shape
    V, H = word.shape
    dev = dout.device
    dword = torch.zeros_like(word)
    dpos = torch.zeros_like(pos)
    dtype0 = torch.empty(H, dtype=F32, device=dev)
    dgamma = torch.empty(H, dtype=F32, device=dev)
    dbeta = torch.empty(H, dtype=F32, device=dev)
    partial = torch.empty(lib().cocodr_embed_bwd_partial_floats(L, H), dtype=F32, device=dev)
    check(lib().cocodr_embed_ln_bwd_drop(ptr(dout), ptr(ids), ptr(word), ptr(pos), ptr(type0), ptr(gamma), ptr(mean), ptr(rstd),
                                         ptr(dword), ptr(dpos), ptr(dtype0), ptr(dgamma), ptr(dbeta), ptr(partial), B, L, H, V,
                                         _dm_ref(drop), stream_ptr()), "embed_ln_bwd")
    return dword, dpos, dtype0, dgamma, dbeta


def ln_fwd(y, gamma, beta, eps: float = 1e-12, cls_stride: int = 0):
    _req(y, BF16, "y", 2); _req(gamma, F32, "gamma", 1); _req(beta, F32, "beta", 1)
    M, H = y.shape
    out = torch.empty_like(y)
    mean = torch.empty(M, dtype=F32, device=y.device)
    rstd = torch.empty_like(mean)
    cls = torch.empty((M // cls_stride, H), dtype=F32, device=y.device) if cls_stride > 0 else None
    check(lib().cocodr_ln_fwd(ptr(y), ptr(gamma), ptr(beta), ptr(out), ptr(mean), ptr(rstd), ptr(cls), cls_stride, M, H, eps,
                              stream_ptr()), "ln_fwd")
    return (out, mean, rstd, cls) if cls_stride > 0 else (out, mean, rstd)


def ln_bwd(dout, y, gamma, mean, rstd, colsum: bool = False, drop: Optional[N.DropoutMask] = None):
    """(dy, dgamma, dbeta[, column sums of dy]); with ``drop`` (the LayerNorm input was dropout(dense) + residual):
    (dy, dy_drop, dgamma, dbeta[, column sums of dy_drop])"""
    _req(dout, BF16, "dout", 2); _req(y, BF16, "y", 2)
    M, H = y.shape
    dy = torch.empty_like(y)
    dgamma = torch.empty(H, dtype=F32, device=y.device)
    dbeta = torch.empty(H, dtype=F32, device=y.device)
    dcs = torch.empty(H, dtype=F32, device=y.device) if colsum else None
    partial = torch.empty(lib().cocodr_ln_bwd_partial_floats(M, H), dtype=F32, device=y.device)
    dyd = torch.empty_like(y) if drop is not None else None
    check(lib().cocodr_ln_bwd_drop(ptr(dout), ptr(y), ptr(gamma), ptr(mean), ptr(rstd), ptr(dy), ptr(dyd), ptr(dgamma), ptr(dbeta),
                                   ptr(dcs) if colsum else None, ptr(partial), M, H, _dm_ref(drop), stream_ptr()), "ln_bwd")
    head = (dy, dyd) if drop is not None else (dy,)
    return head + ((dgamma, dbeta, dcs) if colsum else (dgamma, dbeta))


def colsum(x: torch.Tensor) -> torch.Tensor:
    batched = x.dim() == 3
    _req(x, BF16, "x", 3 if batched else 2)
    nb = x.shape[0] if batched else 1
    M, Nn = x.shape[-2], x.shape[-1]
    out = torch.empty((nb, Nn) if batched else (Nn,), dtype=F32, device=x.device)
    partial = torch.empty(lib().cocodr_colsum_partial_floats(M, Nn, nb), dtype=F32, device=x.device)
    check(lib().cocodr_colsum(ptr(x), ptr(out), ptr(partial), M, Nn, Nn, nb, M * Nn, Nn, stream_ptr()), "colsum")
    return out


def gram(a: torch.Tensor) -> torch.Tensor:
    """a @ a.T for a short, very wide fp32 matrix [G <= 64, D] (iDRO's per-group gradient gram): one streaming pass, deterministic"""
    _req(a, F32, "a", 2)
    if a.stride(1) != 1:
        raise ValueError("gram: rows must be contiguous")
    G, D = a.shape
    out = torch.empty((G, G), dtype=F32, device=a.device)
    ws = torch.empty(lib().cocodr_gram_f32_workspace_floats(G, D), dtype=F32, device=a.device)
    check(lib().cocodr_gram_f32(ptr(a), a.stride(0), G, D, ptr(out), ptr(ws), stream_ptr()), "gram_f32")
    return out


def zero_f32(t: torch.Tensor) -> torch.Tensor:
    """t[...] = 0 for a contiguous fp32 CUDA tensor (cocodr_zero_f32)."""
    _req(t, F32, "t")
    check(lib().cocodr_zero_f32(ptr(t), t.numel(), stream_ptr()), "zero_f32")
    return t


def gather_rows(src: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = src[idx[r]]: bf16 rows [*, H], idx int64 [n] (cocodr_gather_rows)."""
    _req(src, BF16, "src", 2); _req(idx, I64, "idx", 1)
    n, H = idx.shape[0], src.shape[1]
    if out is None:
        out = torch.empty((n, H), dtype=BF16, device=src.device)
    else:
        _req(out, BF16, "out", 2)
        if tuple(out.shape) != (n, H):
            raise ValueError("gather_rows: out shape mismatch")
    check(lib().cocodr_gather_rows(ptr(src), ptr(idx), ptr(out), n, H, stream_ptr()), "gather_rows")
    return out


def scatter_rows(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[idx[r]] = src[r] (dst bf16) or dst[idx[r]] += src[r] (dst fp32; idx unique) - cocodr_scatter_rows."""
    _req(src, BF16, "src", 2); _req(idx, I64, "idx", 1)
    if dst.dtype not in (BF16, F32) or not dst.is_cuda or not dst.is_contiguous() or dst.dim() != 2 or dst.shape[1] != src.shape[1]:
        raise ValueError("scatter_rows: dst must be a contiguous CUDA [M, H] tensor (bf16 or fp32) of src's width")
    if idx.shape[0] > src.shape[0]:
        raise ValueError("scatter_rows: more indices than source rows")
    check(lib().cocodr_scatter_rows(ptr(src), ptr(idx), ptr(dst), idx.shape[0], src.shape[1], int(dst.dtype == F32), stream_ptr()), "scatter_rows")
    return dst


def mul_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """element-wise bf16 product (fp32 multiply, one rounding) - cocodr_mul_bf16."""
    _req(a, BF16, "a"); _req(b, BF16, "b")
    if a.shape != b.shape:
        raise ValueError("mul_bf16: shape mismatch")
    out = torch.empty_like(a)
    check(lib().cocodr_mul_bf16(ptr(a), ptr(b), ptr(out), a.numel(), stream_ptr()), "mul_bf16")
    return out


def cast_f32_bf16(src: torch.Tensor, dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(src, F32, "src")
    if dst is None:
        dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    else:
        _req(dst, BF16, "dst")
        if dst.numel() != src.numel():
            raise ValueError("cast: size mismatch")
    check(lib().cocodr_cast_f32_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr()), "cast_f32_bf16")
    return dst


def scatter_cls_grad(dE: torch.Tensor, L: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(dE, F32, "dE", 2)
    B, H = dE.shape
    if out is None:
        out = torch.empty((B * L, H), dtype=BF16, device=dE.device)
    check(lib().cocodr_scatter_cls_grad(ptr(dE), ptr(out), B, L, H, stream_ptr()), "scatter_cls_grad")
    return out


# ----------------------------------------------------------------------------------------------- losses
def simce_fwd_bwd(E: torch.Tensor, world: int = 1, row0: int = 0, m_local: Optional[int] = None):
    """Returns (loss scalar tensor, loss_rows [M], dE_local [m_local, H]) - COCO/modeling.py:244-248."""
    _req(E, F32, "E", 2)
    M, H = E.shape
    m_local = M if m_local is None else m_local
    dev = E.device
    rows = torch.empty(M, dtype=F32, device=dev)
    loss = torch.empty(1, dtype=F32, device=dev)
    dE = torch.empty((m_local, H), dtype=F32, device=dev)
    ws = torch.empty(lib().cocodr_simce_workspace_floats(M), dtype=F32, device=dev)
    check(lib().cocodr_simce_fwd_bwd(ptr(E), M, H, world, row0, m_local, ptr(rows), ptr(loss), ptr(dE), ptr(ws), stream_ptr()),
          "simce_fwd_bwd")
    return loss, rows, dE


def triplet_nll_fwd_bwd(q, a, b, weights: Optional[torch.Tensor] = None):
    """Returns (loss [1], loss_rows [B], logits [B,2], dq, da, db) - ANCE/model/models.py:97-106,260-261."""
    _req(q, F32, "q", 2); _req(a, F32, "a", 2); _req(b, F32, "b", 2)
    if a.shape != q.shape or b.shape != q.shape:
        raise ValueError("triplet: q/a/b shape mismatch")
    if weights is not None:
        _req(weights, F32, "weights", 1)
    B, H = q.shape
    dev = q.device
    rows = torch.empty(B, dtype=F32, device=dev)
    logits = torch.empty((B, 2), dtype=F32, device=dev)
    loss = torch.empty(1, dtype=F32, device=dev)
    dq, da, db = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    check(lib().cocodr_triplet_nll_fwd_bwd(ptr(q), ptr(a), ptr(b), ptr(weights), B, H, ptr(rows), ptr(logits), ptr(loss),
                                           ptr(dq), ptr(da), ptr(db), stream_ptr()), "triplet_nll_fwd_bwd")
    return loss, rows, logits, dq, da, db


# ----------------------------------------------------------------------------------------------- search
def score_topk(Q: torch.Tensor, P: torch.Tensor, k: int, id_offset: int = 0, workspace: Optional[torch.Tensor] = None,
               p_resident: bool = False):
    """(D [Nq,k] fp32 descending, I [Nq,k] int64) = IndexFlatIP(P).search(Q, k).  ``p_resident``: ``workspace`` still holds the
    passage image of an identical earlier call (cocodr_score_topk_resident; ``retrieval.FlatIPIndex`` keeps that book)."""
    _req(Q, F32, "Q", 2); _req(P, F32, "P", 2)
    if Q.shape[1] != P.shape[1]:
        raise ValueError("score_topk: dim mismatch")
    Nq, H = Q.shape
    Np = P.shape[0]
    need = lib().cocodr_score_topk_workspace_bytes_dim(Nq, Np, H, k)
    if not p_resident and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = torch.empty(need, dtype=torch.uint8, device=Q.device)
    D = torch.empty((Nq, k), dtype=F32, device=Q.device)
    I = torch.empty((Nq, k), dtype=I64, device=Q.device)
    if p_resident and (workspace is None or workspace.numel() * workspace.element_size() < need):
        raise ValueError("score_topk: p_resident needs the workspace of the earlier call")
    check(lib().cocodr_score_topk_resident(ptr(Q), ptr(P), Nq, Np, H, k, id_offset, ptr(D), ptr(I), ptr(workspace),
                                           workspace.numel() * workspace.element_size(), int(bool(p_resident)), stream_ptr()), "score_topk")
    return D, I


def topk_merge(D: torch.Tensor, I: torch.Tensor, shard_offset: torch.Tensor, k_out: int):
    """Merge W sorted per-shard top-k lists per query (cocodr_topk_merge): D fp32 / I int32 [W, Nq, k] with shard-local
    positions, shard_offset int64 [W] -> (D [Nq, k_out] fp32, I [Nq, k_out] int64 global positions)."""
    _req(D, F32, "D", 3); _req(I, I32, "I", 3); _req(shard_offset, I64, "shard_offset", 1)
    if D.shape != I.shape or shard_offset.shape[0] != D.shape[0]:
        raise ValueError("topk_merge: D / I must be [W, Nq, k] and shard_offset [W]")
    W, Nq, k = D.shape
    outD = torch.empty((Nq, k_out), dtype=F32, device=D.device)
    outI = torch.empty((Nq, k_out), dtype=I64, device=D.device)
    check(lib().cocodr_topk_merge(ptr(D), ptr(I), ptr(shard_offset), W, Nq, k, Nq * k, ptr(outD), ptr(outI), int(k_out), stream_ptr()),
          "topk_merge")
    return outD, outI


def score_filter_plan(Nq: int, Np: int, H: int, k: int) -> dict:
    """How cocodr_score_topk would search these sizes in the current score mode (include/cocodr.h: the filtered search)."""
    out = (C.c_longlong * 8)()
    check(lib().cocodr_score_filter_plan(int(Nq), int(Np), int(H), int(k), C.cast(out, C.c_void_p)), "score_filter_plan")
    keys = ("filtered", "sample_passages", "sample_stride", "threshold_rank", "block_slots", "rows_per_pass", "rows_per_exhaustive_pass",
            "handed_back_count_offset")
    return {k_: int(v) for k_, v in zip(keys, out)}


def score_set_mode(mode: int) -> None:
    """0 = split-precision scores on the 16-bit matrix pipe (default: fp32-accurate), 1 = exact fp32 MFMA scores, 2 = half-precision
    scores (opt-in: one product of the operands rounded to IEEE half, a third of mode 0's matrix work; include/cocodr.h)."""
    check(lib().cocodr_score_set_mode(int(mode)), "score_set_mode")


# ----------------------------------------------------------------------------------------------- profiling hooks
def prof_begin(kind: int) -> None:
    check(lib().cocodr_prof_begin(kind), "prof_begin")


def prof_pause(paused: bool) -> None:
    check(lib().cocodr_prof_pause(int(bool(paused))), "prof_pause")


def prof_end():
    n, ms, fl = C.c_int(0), C.c_double(0.0), C.c_double(0.0)
    check(lib().cocodr_prof_end(C.byref(n), C.byref(ms), C.byref(fl)), "prof_end")
    return n.value, ms.value, fl.value


def prof_event_overhead_us() -> float:
    """microseconds one begin / end event pair adds to the launch it brackets (cocodr_prof_event_overhead_us, include/cocodr.h)"""
    us = C.c_double(0.0)
    check(lib().cocodr_prof_event_overhead_us(stream_ptr(), C.byref(us)), "prof_event_overhead_us")
    return us.value


def probe_mfma32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty((32, 32), dtype=F32, device=a.device)
    check(lib().cocodr_probe_mfma32(ptr(a), ptr(b), ptr(out), stream_ptr()), "probe_mfma32")
    return out


def probe_tr16(tile: torch.Tensor) -> torch.Tensor:
    out = torch.empty((64, 4), dtype=torch.int16, device=tile.device)
    check(lib().cocodr_probe_tr16(ptr(tile), ptr(out), stream_ptr()), "probe_tr16")
    return out
