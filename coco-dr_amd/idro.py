"""iDRO re-weighting of the ANCE triplet step: ``iDROLoss`` (ANCE/model/dro_loss.py:160-254) behind
``BertDot_NLL_LN.forward(group_ids=...)`` (ANCE/model/models.py:211-223, 259-273).

Per step: per-row triplet losses -> per-group mean losses L_g; the gradient of every present group's L_g w.r.t. the
parameters of the last layers (dro_loss.py:177-205: layers 9-11 of a 12-layer model, the last 2 of a large one);
cross-rank SUM of the [G, D] gradient matrix (:234); cosine gram x (L^alpha outer product) -> multiplicative update of
the group weights h (:236-252); the step's loss is sum_g h_g L_g with the weights from BEFORE the update (:229).

Native structure: the three encoder passes keep their activation arenas; a group's gradient is ONE partial backward
(``cocodr_encoder_bwd_range`` over the selected layers only, [CLS] gradient of the group's rows scaled by 1/count_g)
per pass instead of an autograd.grad through the whole graph, and the gram / weight update are a handful of small
device ops.  The final backward is the ordinary one with row weights h_{g(i)} / count_{g(i)}.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from . import ops
from ._native import check, lib, ptr, stream_ptr

__all__ = ["IDROLoss", "DROGreedyLoss", "idro_triplet_step"]


class IDROLoss(torch.nn.Module):
    """State and update rule of ``iDROLoss`` (hyper-parameters as in ``add_group_loss``, models.py:211-218).  A module with
    the reference's buffer name, so ``state_dict()`` of the wrapping model carries ``loss.h_fun`` across checkpoints."""

    def __init__(self, n_groups: int, alpha: float, eps: float, ema: float = 0.1, rho: float = 0.1, model_size: str = "base",
                 device=None):
        super().__init__()
        self.n_groups, self.alpha, self.eps, self.ema, self.rho = int(n_groups), float(alpha), float(eps), float(ema), float(rho)
        self.model_size = model_size
        self.register_buffer("h_fun", torch.ones(self.n_groups, dtype=torch.float32, device=device))  # dro_loss.py:28
        self.per_group_backward = False   # True: one partial backward per present group (the reference's structure)
        self.last_path = None

    def selected_layers(self, n_layers: int) -> Tuple[int, int]:
        """[lo, hi) of the re-weighted layers: the reference matches the names layer.9/10/11 (base) or layer.22/23
        (large) (:177-181), i.e. the top 3 / top 2 of the stack."""
        k = 2 if self.model_size == "large" else 3
        if n_layers < k:
            raise ValueError("iDRO needs at least %d encoder layers" % k)
        return n_layers - k, n_layers

    @torch.no_grad()
    def update(self, group_losses: torch.Tensor, counts: torch.Tensor, all_grads: torch.Tensor) -> None:
        mask = (counts > 0).to(torch.float32)
        # cosine gram (:236-238) from ONE pass over the [G, D] matrix: raw gram, norms from its diagonal
        raw = _gram(all_grads)
        nrm = 1e-12 + torch.sqrt(torch.clamp(torch.diagonal(raw), min=0.0))
        RTG = raw / (nrm[:, None] * nrm[None, :])
        gl = torch.pow(group_losses.unsqueeze(-1), self.alpha)                              # :240
        RTG = (gl * gl.T) * RTG              # :241 gl [G,1]: the outer product, as a broadcast                  
        ex = self.rho * RTG.mean(dim=0) * mask                                              # :242-244
        ex = ex - ex.max()                                                                  # :246
        h = torch.pow(self.h_fun, self.ema) * torch.exp(ex) * (counts != 0).to(torch.float32)  # :248-250
        h = h / h.sum()
        self.h_fun = torch.clamp(h, min=self.eps)                                           # :252


def _gram(a: torch.Tensor) -> torch.Tensor:
    """a @ a.T for the [groups, D] gradient matrix (ANCE/model/dro_loss.py:236): the native streaming gram (ops.gram), in
    blocks of 64 groups should there ever be more"""
    G = a.shape[0]
    if G <= 64:
        return ops.gram(a.contiguous())
    out = torch.empty((G, G), dtype=torch.float32, device=a.device)
    for i in range(0, G, 32):
        for j in range(i, G, 32):
            ni, nj = min(32, G - i), min(32, G - j)
            if j == i:
                out[i:i + ni, i:i + ni] = ops.gram(a[i:i + ni].contiguous())
            else:  # ni == 32 here: the cross block of the stacked [32 + nj] rows
                out[i:i + ni, j:j + nj] = ops.gram(torch.cat([a[i:i + ni], a[j:j + nj]]).contiguous())[:ni, ni:ni + nj]
                out[j:j + nj, i:i + ni] = out[i:i + ni, j:j + nj].T
    return out


class DROGreedyLoss(torch.nn.Module):
    """``DROGreedyLoss`` (ANCE/model/dro_loss.py:11-126, the driver's default ``--dro_type``): the step's loss is
    ``sum_i h[g_i] * w_i * loss_i / B`` with the weights of the previous step; afterwards the EMA group losses / counts
    (gathered over all ranks) choose the worst groups whose cumulative EMA fraction stays below ``alpha``: weight
    ``1/alpha`` for them, the left-over mass for the next one, ``eps`` for the rest (``update_mw``)."""

    def __init__(self, n_groups: int, alpha: float, eps: float, ema: float = 0.1, weight_ema: bool = False, device=None):
        super().__init__()
        self.n_groups, self.alpha, self.eps, self.ema, self.weight_ema = int(n_groups), float(alpha), float(eps), float(ema), bool(weight_ema)
        self.register_buffer("h_fun", torch.ones(self.n_groups, dtype=torch.float32, device=device))        # dro_loss.py:28-30
        self.register_buffer("sum_losses", torch.zeros(self.n_groups, dtype=torch.float32, device=device))
        self.register_buffer("count_cat", torch.ones(self.n_groups, dtype=torch.float32, device=device))

    def row_weights(self, groups: torch.Tensor, weights: Optional[torch.Tensor]) -> torch.Tensor:
        """What ``(loss * row_weights).mean()`` must use so that it equals the robust loss (dro_loss.py:51-60)."""
        hw = self.h_fun[groups.to(torch.int64)]
        return hw if weights is None else hw * weights.to(torch.float32)

    @torch.no_grad()
    def update(self, loss_rows: torch.Tensor, groups: torch.Tensor, weights: Optional[torch.Tensor]):
        """Post-step bookkeeping (dro_loss.py:62-90).  Returns the LOCAL (group mean losses, group counts)."""
        import torch.distributed as dist
        G = self.n_groups
        g = groups.to(torch.int64)
        losses = loss_rows if weights is None else loss_rows * weights.to(torch.float32)
        ga, la = g, losses
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:   # gather_tensors, :128-136
            W = dist.get_world_size()
            ga = torch.empty(W * g.numel(), dtype=g.dtype, device=g.device)
            la = torch.empty(W * losses.numel(), dtype=losses.dtype, device=losses.device)
            dist.all_gather_into_tensor(ga, g.contiguous())
            dist.all_gather_into_tensor(la, losses.contiguous())
        zero = torch.zeros(G, dtype=torch.float32, device=losses.device)
        cnt_agg = zero.scatter_add(0, ga, torch.ones_like(la))
        mean_agg = zero.scatter_add(0, ga, la) / (cnt_agg + (cnt_agg == 0).float())
        valid = cnt_agg > 0
        self.sum_losses = torch.where(valid, self.sum_losses * (1 - self.ema) + self.ema * mean_agg, self.sum_losses)  # :75
        self.count_cat = self.count_cat * (1 - self.ema) + self.ema * cnt_agg                                         # :78-79
        self._update_mw()
        cnt = zero.scatter_add(0, g, torch.ones_like(losses))
        return zero.scatter_add(0, g, losses) / (cnt + (cnt == 0).float()), cnt

    def _update_mw(self):  # dro_loss.py:93-126
        frac = self.count_cat / self.count_cat.sum()
        _sorted, sort_id = torch.sort(self.sum_losses, descending=True, stable=True)
        sfrac = frac[sort_id]
        n = sfrac.numel()
        cutoff = torch.clamp((torch.cumsum(sfrac, 0) < self.alpha).sum(), max=n - 1)   # device scalar, no host sync
        pos = torch.arange(n, device=sfrac.device)
        head = pos < cutoff
        leftover = 1.0 - (sfrac * head).sum() / self.alpha
        tie = torch.clamp(leftover / sfrac[cutoff], min=self.eps)
        vals = torch.where(head, torch.full_like(sfrac, 1.0 / self.alpha), torch.full_like(sfrac, self.eps))
        vals = torch.where(pos == cutoff, tie, vals)
        h = torch.empty_like(self.h_fun)
        h[sort_id] = vals
        if self.weight_ema:
            self.h_fun = self.h_fun * (1 - self.ema) + torch.clamp(h, min=self.eps) * self.ema
        else:
            self.h_fun = h


def _per_sequence_group_grads(bert, passes, arenas, unit, g, inv, all_grads, layers, slices, scratch, structs) -> bool:
    """All group gradients from ONE un-weighted partial backward per encoder pass.  The backward of a sequence depends
    only on its own upstream gradient (attention and LayerNorm never mix sequences), so a single pass with every row's
    d(row loss)/d[CLS] leaves, per layer of the range, output-gradient matrices whose rows are each sequence's own
    flow.  The weight gradient of group g is then sum_{i in g} dY_i^T X_i / count_g: batched per-SEQUENCE weight-gradient
    GEMMs (batch = sequence, contraction over its L tokens) + an index_add by group; bias / LayerNorm gradients come from
    per-sequence column sums and from the LayerNorm backward's partial rows (which cover whole fractions of a sequence).
    Returns False (nothing written) when the partial rows do not align with sequences; the caller then falls back."""
    from . import _native as N
    lo, cfg = bert.layout, bert.config
    H, I = cfg.hidden_size, cfg.intermediate_size
    l_lo, l_hi = layers
    d0, d1, n0, n1 = slices
    if l_hi - l_lo < 2:
        return False
    sd, sn = scratch
    emb, arr, eg, garr, ccfg = structs
    Dd = d1 - d0
    lays = []
    for ids, _mask in passes:
        Bp, L = ids.shape
        bl = N.EncoderBwdLayout()
        check(lib().cocodr_encoder_bwd_layout(C.byref(ccfg[0]), Bp, L, C.byref(bl)), "encoder_bwd_layout")
        if L % bl.ln_rows != 0 or bl.ln_rows * bl.ln_blocks != Bp * L:
            return False
        lays.append(bl)
    B = g.shape[0]
    for p, (ids, mask) in enumerate(passes):
        Bp, L = ids.shape
        M = Bp * L
        arena, bl, fl = arenas[p], lays[p], bert._layout_for(Bp, L, True)
        d16 = ops.scatter_cls_grad(unit[p].contiguous(), L)
        check(lib().cocodr_encoder_bwd_range(C.byref(ccfg[p]), C.byref(emb), arr, C.byref(eg), garr, ptr(ids), ptr(mask), ptr(d16), Bp, L,
                                             ptr(arena), arena.numel(), l_hi, l_lo, 0, stream_ptr()), "encoder_bwd_range(idro)")
        seq_g = g if Bp == B else torch.cat([g, g])

        def act(off, layer, width):  # bf16 [Bp, L, width] view of a per-layer activation / gradient block
            b0 = off + layer * M * width * 2
            return arena[b0: b0 + M * width * 2].view(torch.bfloat16).view(Bp, L, width)

        def add(col0, per_seq):      # all_grads[group, col0 : col0 + n] += per_seq[i] for every sequence i of the group
            n = per_seq.shape[1]     # (SUMS; the caller applies the 1 / count_g of the group MEAN once, before the cross-rank sum)
            all_grads[:, col0: col0 + n].index_add_(0, seq_g, per_seq)

        per = L // bl.ln_rows
        for l in range(l_lo, l_hi):
            li = l - l_lo
            cd, cn = li * lo.mat_stride, Dd + li * lo.vec_stride
            dqkv, dy1 = act(bl.dqkv, l, 3 * H), act(bl.dy1, l, H)
            du, dy2 = act(bl.du, l, I), act(bl.dy2, l, H)
            x_in, ctx = act(fl.hidden, l, H), act(fl.ctx, l, H)
            x1, hh = act(fl.x1, l, H), act(fl.h, l, I)
            for col, a_, b_ in ((lo.off_wqkv, dqkv, x_in), (lo.off_wo, dy1, ctx), (lo.off_w1, du, x1), (lo.off_w2, dy2, hh)):
                gw = ops.gemm(a_, b_, trans_a=True, trans_b=True, out_f32=True)      # [Bp, out, in] fp32, one per sequence
                add(cd + col, gw.view(Bp, -1))
                del gw
            add(cn + lo.off_bqkv, ops.colsum(dqkv))
            add(cn + lo.off_b1, ops.colsum(du))
            for off, slot_off, names in ((bl.ln1_partial, li, (lo.off_ln1g, lo.off_ln1b, lo.off_bo)),
                                         (bl.ln2_partial, li, (lo.off_ln2g, lo.off_ln2b, lo.off_b2))):
                nfl = bl.ln_blocks * 3 * H
                b0 = off + slot_off * nfl * 4
                part = arena[b0: b0 + nfl * 4].view(torch.float32).view(Bp, per, 3, H).sum(1)   # [Bp, 3, H]
                for k, col in enumerate(names):
                    add(cn + col, part[:, k].contiguous())
    return True


class _IDROStepFn(torch.autograd.Function):
    """(flat_decay, flat_nodecay) -> robust loss; everything else rides along un-differentiated."""

    @staticmethod
    def forward(ctx, fd, fn, passes, groups, bert, dro: IDROLoss):
        # passes: list of (ids int32 [Bp,L], mask int32 [Bp,L]); rows of q / a / b are located by `slots`
        cfg = bert.config
        H, NL = cfg.hidden_size, cfg.num_hidden_layers
        G = dro.n_groups
        dev = fd.device
        arenas, cls_rows = [], []
        for ids, mask in passes:
            arena, lay = bert._run_forward(ids, mask, True)
            Bp = ids.shape[0]
            cls_rows.append(arena[lay.cls_f32: lay.cls_f32 + Bp * H * 4].view(torch.float32).view(Bp, H).clone())
            arenas.append(arena)
        q = cls_rows[0]
        B = q.shape[0]
        if len(passes) == 2:   # positives and negatives share one pass
            a, b = cls_rows[1][:B], cls_rows[1][B:]
        else:
            a, b = cls_rows[1], cls_rows[2]
        _mean, rows, logits, dq, da, db = ops.triplet_nll_fwd_bwd(q.contiguous(), a.contiguous(), b.contiguous(), None)
        dq, da, db = dq * B, da * B, db * B            # d(row loss)/d(q, a, b): the kernel folds in the 1/B of the mean
        unit = [dq, torch.cat([da, db])] if len(passes) == 2 else [dq, da, db]
        # ---- group statistics (dro_loss.py:220-229)
        g = groups.to(torch.int64)
        counts = torch.zeros(G, dtype=torch.float32, device=dev).scatter_add_(0, g, torch.ones(B, dtype=torch.float32, device=dev))
        sums = torch.zeros(G, dtype=torch.float32, device=dev).scatter_add_(0, g, rows)
        group_losses = sums / (counts + (counts == 0).to(torch.float32))
        h_old = dro.h_fun.clone()
        robust = (group_losses * h_old).sum()
        # ---- per-group gradients of the selected layers (dro_loss.py:192-205, 231-234)
        lo = bert.layout
        l_lo, l_hi = dro.selected_layers(NL)
        d0, d1 = lo.mat_begin + l_lo * lo.mat_stride, lo.mat_begin + l_hi * lo.mat_stride
        n0, n1 = lo.vec_begin + l_lo * lo.vec_stride, lo.vec_begin + l_hi * lo.vec_stride
        all_grads = torch.zeros((G, (d1 - d0) + (n1 - n0)), dtype=torch.float32, device=dev)
        sd, sn = torch.empty_like(fd), torch.empty_like(fn)   # scratch gradient flats: only the selected slices are written
        emb, arr, eg, garr = bert._param_structs((sd, sn))
        # one config per pass: each training forward drew its own dropout call counter (it travels with the arena)
        ccfg = [bert._c_config(getattr(a_, "_cocodr_drop", None)) for a_ in arenas]
        inv = 1.0 / counts.clamp(min=1.0)
        fast = (not dro.per_group_backward) and _per_sequence_group_grads(
            bert, passes, arenas, unit, g, inv, all_grads, (l_lo, l_hi), (d0, d1, n0, n1), (sd, sn), (emb, arr, eg, garr, ccfg))
        dro.last_path = "per-sequence" if fast else "per-group"
        if fast:
            # per-rank group MEAN gradients, as the reference forms them (dro_loss.py:192-205): on one rank the cosine gram
            # ignores a row scale, but the cross-rank SUM below (:234) mixes ranks whose counts of a group differ
            all_grads.mul_(inv[:, None])
        present = [] if fast else torch.nonzero(counts > 0).flatten().tolist()   # fallback: one partial backward per group
        for gi in present:
            w = (g == gi).to(torch.float32) * inv[gi]
            for p, (ids, mask) in enumerate(passes):
                Bp, L = ids.shape
                wp = w if Bp == B else torch.cat([w, w])
                d16 = ops.scatter_cls_grad((unit[p] * wp[:, None]).contiguous(), L)
                check(lib().cocodr_encoder_bwd_range(C.byref(ccfg[p]), C.byref(emb), arr, C.byref(eg), garr, ptr(ids), ptr(mask), ptr(d16),
                                                     Bp, L, ptr(arenas[p]), arenas[p].numel(), l_hi, l_lo, 0, stream_ptr()),
                      "encoder_bwd_range(idro)")
                all_grads[gi, : d1 - d0] += sd[d0:d1]
                all_grads[gi, d1 - d0:] += sn[n0:n1]
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(all_grads)                                                   # :234 (SUM)
        dro.update(group_losses.detach(), counts, all_grads)
        # ---- what the ordinary backward needs: row weights h_{g(i)} / count_{g(i)} of the PRE-update weights
        ctx.bert, ctx.passes, ctx.arenas, ctx.unit = bert, passes, arenas, unit
        ctx.row_w = (h_old * inv)[g]
        ctx.B = B
        ctx.mark_non_differentiable(rows, logits, group_losses, counts)
        return robust, rows, logits, group_losses, counts

    @staticmethod
    def backward(ctx, g_robust, *_unused):
        bert = ctx.bert
        gd_tot = gn_tot = None
        for p, (ids, mask) in enumerate(ctx.passes):
            Bp, L = ids.shape
            wp = ctx.row_w if Bp == ctx.B else torch.cat([ctx.row_w, ctx.row_w])
            d16 = ops.scatter_cls_grad((ctx.unit[p] * (wp * g_robust)[:, None]).contiguous(), L)
            gd, gn = bert._run_backward(ids, mask, d16, ctx.arenas[p])
            gd_tot = gd if gd_tot is None else gd_tot.add_(gd)
            gn_tot = gn if gn_tot is None else gn_tot.add_(gn)
        ctx.arenas = None
        return gd_tot, gn_tot, None, None, None, None


def idro_triplet_step(bert, dro: IDROLoss, query_ids, attention_mask_q, input_ids_a, attention_mask_a, input_ids_b,
                      attention_mask_b, group_ids):
    """-> (robust_loss, loss_rows, logits, group_losses, group_counts); ``dro.h_fun`` is updated in place."""
    q_ids, q_mask, _ = bert._prep(query_ids, attention_mask_q)
    a_ids, a_mask, _ = bert._prep(input_ids_a, attention_mask_a)
    b_ids, b_mask, _ = bert._prep(input_ids_b, attention_mask_b)
    if a_ids.shape == b_ids.shape:
        passes = [(q_ids, q_mask), (torch.cat([a_ids, b_ids]), torch.cat([a_mask, b_mask]))]
    else:
        passes = [(q_ids, q_mask), (a_ids, a_mask), (b_ids, b_mask)]
    return _IDROStepFn.apply(bert.flat_decay, bert.flat_nodecay, passes, group_ids, bert, dro)
