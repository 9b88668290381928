"""World-size-2 `gloo` tests (CPU) of the multi-process logic on the hot path: the [CLS] gather with its
local-slot gradient rule, the local-row gradient identity the simce kernel relies on, and the sharded search
merge.  No GPU kernels are called here (the native search is replaced by an injected numpy search)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(fn, world, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q, args)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    errs = [o for o in out if isinstance(o, str)]
    assert not errs, errs
    return dict(out)


def _entry(fn, rank, world, port, q, args):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(1)
        q.put((rank, fn(rank, world, *args)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))


def _reference_pattern_local_grad(rank, world, E_full):
    """The reference's gather (COCO/modeling.py:182-190: all_gather into fresh buffers, overwrite slot `rank` with the
    autograd tensor, cat) followed by compute_contrastive_loss/.mean() - executed for real in 2 processes."""
    import torch.nn.functional as F
    m = E_full.shape[0] // world
    t = torch.from_numpy(E_full[rank * m:(rank + 1) * m].copy()).requires_grad_(True)
    all_t = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(all_t, t.detach())
    all_t[rank] = t
    E = torch.cat(all_t)
    S = E @ E.T
    S.fill_diagonal_(float("-inf"))
    target = torch.arange(E.shape[0]).view(-1, 2).flip([1]).flatten()
    loss = (F.cross_entropy(S, target, reduction="none") * world).mean()
    loss.backward()
    return float(loss), t.grad.numpy()


def test_oracle_local_gradient_matches_two_process_reference_pattern():
    rng = np.random.Generator(np.random.PCG64(0))
    E = (rng.standard_normal((12, 16)) * 0.8).astype(np.float32)
    res = _run(_reference_pattern_local_grad, 2, E)
    ref_loss, _ = O.contrastive_loss_grad(E.copy(), 2)
    for r in range(2):
        loss, g = res[r]
        assert abs(loss - ref_loss) < 1e-5
        np.testing.assert_allclose(g, O.contrastive_local_grad(E.copy(), 2, r), rtol=1e-4, atol=1e-6)
    # the identity the native kernel uses: dE_i = (W/M) sum_j [e^{S_ij-lse_i} + e^{S_ij-lse_j} - 2 [j == i^1]] E_j
    E64 = E.astype(np.float64)
    S = E64 @ E64.T
    np.fill_diagonal(S, -np.inf)
    lse = np.log(np.exp(S).sum(1))
    Gs = np.exp(S - lse[:, None]) + np.exp(S - lse[None, :])
    Gs[np.arange(12), np.arange(12) ^ 1] -= 2.0
    np.fill_diagonal(Gs, 0.0)
    full = (2 / 12) * Gs @ E64
    for r in range(2):
        np.testing.assert_allclose(full[r * 6:(r + 1) * 6], res[r][1], rtol=1e-4, atol=1e-6)


def _gather_rows(rank, world):
    import cocodr_amd
    from cocodr_amd.modeling import _GatherRows
    t = (torch.arange(6, dtype=torch.float32).view(3, 2) + 10 * rank).requires_grad_(True)
    E = _GatherRows.apply(t)
    w = torch.arange(E.numel(), dtype=torch.float32).view_as(E)
    (E * w).sum().backward()
    return E.detach().numpy(), t.grad.numpy()


def test_gather_rows_forward_is_rank_major_and_backward_is_local_slice():
    res = _run(_gather_rows, 2)
    w = np.arange(12, dtype=np.float32).reshape(6, 2)
    for r in range(2):
        E, g = res[r]
        assert E.shape == (6, 2)
        np.testing.assert_array_equal(E[:3], np.arange(6).reshape(3, 2))
        np.testing.assert_array_equal(E[3:], np.arange(6).reshape(3, 2) + 10)
        np.testing.assert_array_equal(g, w[r * 3:(r + 1) * 3])  # no collective in the backward


def _sharded(rank, world, Q, P, k):
    import cocodr_amd
    from cocodr_amd import retrieval as R

    def numpy_search(q, p, kk, off):  # stands in for the native kernel on CPU
        D, I = O.score_topk(q.numpy(), p.numpy(), kk)
        I = np.where(I >= 0, I + off, I)
        return torch.from_numpy(D), torch.from_numpy(I)

    def numpy_merge(Dw, Iw, offs, kk):  # stands in for cocodr_topk_merge: [W, Nq, k] sorted lists, shard-local int32 positions
        Dw, Iw, offs = Dw.numpy(), Iw.numpy().astype(np.int64), offs.numpy()
        Ig = [np.where(Iw[w] >= 0, Iw[w] + offs[w], -1) for w in range(Dw.shape[0])]
        D, I = O.merge_topk([Dw[w] for w in range(Dw.shape[0])], Ig, kk)
        return torch.from_numpy(D), torch.from_numpy(I)

    qi = R.shard_indices(Q.shape[0], rank, world)
    pi = R.shard_indices(P.shape[0], rank, world)
    D, I = R.sharded_search(torch.from_numpy(Q)[qi], torch.from_numpy(P)[pi], k, local_search=numpy_search, local_merge=numpy_merge)
    Db, Ib, (lo, hi) = R.sharded_search(torch.from_numpy(Q)[qi], torch.from_numpy(P)[pi], k, local_search=numpy_search,
                                        local_merge=numpy_merge, gather=False)
    assert np.array_equal(Db.numpy(), D.numpy()[lo:hi]) and np.array_equal(Ib.numpy(), I.numpy()[lo:hi])  # the block form
    return D.numpy(), I.numpy()


@pytest.mark.parametrize("nq,npass,k", [(7, 101, 10), (4, 9, 6)])
def test_sharded_search_equals_search_over_merged_corpus(nq, npass, k):
    rng = np.random.Generator(np.random.PCG64(npass))
    Q = rng.standard_normal((nq, 8)).astype(np.float32)
    P = rng.standard_normal((npass, 8)).astype(np.float32)
    res = _run(_sharded, 2, Q, P, k)
    import cocodr_amd
    from cocodr_amd import retrieval as R
    order_p = R.merged_order(npass, 2).numpy()
    order_q = R.merged_order(nq, 2).numpy()
    Dr, Ir = O.score_topk(Q[order_q], P[order_p], k)  # reference semantics: search the rank-major merged arrays
    assert np.array_equal(order_p, O.merged_order(npass, 2))
    for r in range(2):
        D, I = res[r]
        np.testing.assert_allclose(D, Dr, rtol=1e-6)
        assert np.array_equal(I, Ir)


def _dro_greedy_two_ranks(rank, world, losses, groups, G):
    """DROGreedyLoss.update in 2 processes: every rank must end with the weights computed from ALL ranks' rows
    (ANCE/model/dro_loss.py:62-80 gathers group ids and losses before the EMA / weight update)."""
    import cocodr_amd  # noqa: F401
    from cocodr_amd.idro import DROGreedyLoss
    m = losses.shape[0] // world
    dro = DROGreedyLoss(G, alpha=0.5, eps=0.05, ema=0.3, weight_ema=False, device="cpu")
    out = []
    for step in range(losses.shape[1]):
        rows = torch.from_numpy(losses[rank * m:(rank + 1) * m, step].copy())
        g = torch.from_numpy(groups[rank * m:(rank + 1) * m, step].copy())
        gl, cnt = dro.update(rows, g, None)
        out.append((dro.h_fun.numpy().copy(), dro.sum_losses.numpy().copy(), dro.count_cat.numpy().copy(), gl.numpy().copy()))
    return out


def test_dro_greedy_weight_update_aggregates_over_ranks():
    rng = np.random.Generator(np.random.PCG64(3))
    G, B, steps = 5, 8, 3
    losses = rng.random((B, steps)).astype(np.float32) * 3
    groups = rng.integers(0, G, (B, steps)).astype(np.int64)
    res = _run(_dro_greedy_two_ranks, 2, losses, groups, G)
    st = O.DROGreedyState(G)
    for step in range(steps):
        for rank in range(2):
            sl = slice(rank * 4, rank * 4 + 4)
            st_r = O.DROGreedyState(G)
            st_r.h_fun, st_r.sum_losses, st_r.count_cat = st.h_fun.copy(), st.sum_losses.copy(), st.count_cat.copy()
            _rob, _rw, gl, _cnt = O.dro_greedy_forward(st_r, losses[sl, step].astype(np.float64), groups[sl, step], None, G, 0.5, 0.05, 0.3,
                                                       False, all_losses=losses[:, step].astype(np.float64), all_groups=groups[:, step])
            h, sl_, cc, gl_got = res[rank][step]
            np.testing.assert_allclose(h, st_r.h_fun, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(sl_, st_r.sum_losses, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(cc, st_r.count_cat, rtol=1e-6)
            np.testing.assert_allclose(gl_got, gl, rtol=1e-5, atol=1e-6)
        st = st_r  # both ranks hold the same global state


# ---- the private torch API the DDP adoption leans on (CocoBertModel._dp_adopt_ddp_wrapper, modeling.py): checked against THIS torch
class _SeesItsWrapper(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(4))
        self.seen = []

    def forward(self, x):
        from torch.nn.parallel import DistributedDataParallel as DDP
        self.seen.append(DDP._active_ddp_module)
        return (x * self.w).sum()


def _ddp_publishes_its_wrapper(rank, world):
    from torch.nn.parallel import DistributedDataParallel as DDP
    inner = _SeesItsWrapper()
    wrapped = DDP(inner, find_unused_parameters=True)
    wrapped(torch.ones(4)).backward()
    inside = inner.seen[-1] is wrapped                       # published while the wrapped forward runs ...
    outside = DDP._active_ddp_module is None                 # ... and withdrawn afterwards
    # the switches the adoption flips / rebinds exist under these names
    return bool(inside and outside and isinstance(wrapped.require_backward_grad_sync, bool) and callable(wrapped.no_sync)
                and hasattr(wrapped, "process_group"))


def test_torch_ddp_still_offers_what_the_adoption_of_a_ddp_wrapper_needs():
    """`DistributedDataParallel._active_ddp_module`, `require_backward_grad_sync`, `no_sync` and `process_group` are what
    `_dp_adopt_ddp_wrapper` uses to find the reference's DDP wrap line (ANCE/drivers/run_ann.py:177-184) and take over the gradient
    reduction.  The first is private: a torch release that drops or renames it makes this test fail here, loudly, instead of leaving
    the ranks of a wrapped model un-reduced behind a warning."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    assert hasattr(DDP, "_active_ddp_module"), "torch %s: DistributedDataParallel._active_ddp_module is gone" % torch.__version__
    out = _run(_ddp_publishes_its_wrapper, 2)
    assert out == {0: True, 1: True}
