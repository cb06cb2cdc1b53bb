"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: concept-shard bookkeeping + K/Zc all-gather in
request order, caption-shard partition + second-moment all-reduce.  The kernels themselves need a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from emcid_amd import runningstats as rs
        from emcid_amd.edit_engine import ConceptShard, _all_gather_rows
        from emcid_amd.emcid_main import _shard_from_env

        sh = _shard_from_env(None)
        assert (sh.rank, sh.world) == (rank, world)
        # K rows of this rank's requests -> all ranks hold the full stack in request order (uneven shards)
        full = torch.arange(n_total * 6, dtype=torch.float32).reshape(n_total, 6)
        lo, hi = sh.bounds(n_total)
        got = _all_gather_rows(full[lo:hi].clone(), SimpleNamespace(shard=sh, n_total=n_total))
        assert torch.equal(got, full), (rank, got)
        # caption sample partition: union of shards == the single-process sample, disjoint
        s = rs.FixedRandomSubsetSampler(range(500), end=123, seed=1)
        mine = list(s.shard(rank, world))
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        assert sum(gathered, []) == list(s)
        # second-moment all-reduce: sum of per-rank Grams and counts
        x = torch.randn(40, 8, generator=torch.Generator().manual_seed(7))
        part = x[rank::world]
        st = rs.CombinedStat(mom2=rs.SecondMoment())
        st.load_state_dict({"mom2.count": part.shape[0], "mom2.mom2": (part.t() @ part).numpy()})
        st.all_reduce_()
        assert st.mom2.count == 40
        torch.testing.assert_close(st.mom2.mom2, x.t() @ x, rtol=1e-5, atol=1e-5)
        # tally(shard=...) : rank 0 writes the cache after the reduce
        cache = os.path.join(tmp, "s.npz")
        st2 = rs.CombinedStat(mom2=rs.SecondMoment())
        data = torch.arange(20, dtype=torch.float32).reshape(20, 1)
        seen = []
        for (b,) in rs.tally(st2, data, cache=cache, shard=(rank, world), batch_size=4, sample_size=20, quiet=True):
            seen.append(b)
            if st2.mom2.mom2 is None:   # emulate add() on CPU: load a running state
                st2.load_state_dict({"mom2.count": 0, "mom2.mom2": np.zeros((1, 1), np.float32)})
            cur = st2.mom2
            cur.load_state_dict({"count": cur.count + b.shape[0], "mom2": (cur.mom2 + b.t() @ b).numpy()})
        counts = [None] * world
        dist.all_gather_object(counts, sum(t.shape[0] for t in seen))
        assert sum(counts) == 20 and max(counts) - min(counts) <= 1          # every item once, shards even to within one
        dist.barrier()
        with np.load(cache) as z:
            assert int(z["mom2.count"]) == 20 and float(z["mom2.mom2"][0, 0]) == float((data ** 2).sum())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8])
def test_world2_gloo(tmp_path, n_total):
    mp.spawn(_worker, args=(2, _free_port(), n_total, str(tmp_path)), nprocs=2, join=True)


def test_world8_gloo_uneven_shards(tmp_path):
    """The skeleton at the node's size: 8 ranks, 1 003 concepts (shards of 126 / 125), caption shards, the reduce."""
    mp.spawn(_worker, args=(8, _free_port(), 1003, str(tmp_path)), nprocs=8, join=True)


def test_concept_shard_bounds_cover_all():
    from emcid_amd.edit_engine import ConceptShard
    for n in (1, 7, 8, 1000):
        for w in (1, 2, 4, 8):
            spans = [ConceptShard(r, w).bounds(n) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


class _TorchColsBackend:
    """Torch-CPU restatement of the two stages of the column-sharded solve (include/emcid_hip.h, "COLUMN-SHARDED"), so that
    the real skeleton (hip.edit_layer_dual_cols: tile lists, the two all-reduces, the order of the steps) runs under gloo."""

    def __init__(self, K, Zc, zs_t, C, lam, ew, layers_left):
        s = (ew / 0.5) ** 0.5
        self.Kt = K.double() * s
        self.Rt = (zs_t - Zc).double() * s / layers_left
        M = lam * (C * (1 - ew) / 0.5).double()
        self.X = torch.linalg.inv(torch.linalg.cholesky(M))          # X = inv(L), lower triangular
        self.Yc = None

    def stage1(self, tiles):
        self.Yc = torch.cat([self.Kt @ self.X[128 * t:128 * t + 128].t() for t in tiles], dim=1)
        self.S = self.Yc @ self.Yc.t()
        return self.S

    def stage2(self, tiles):
        n = self.S.shape[0]
        Z = torch.linalg.solve(self.S + torch.eye(n, dtype=torch.float64), self.Rt)
        V = Z.t() @ self.Yc
        self.U = sum(V[:, 128 * i:128 * i + 128] @ self.X[128 * t:128 * t + 128] for i, t in enumerate(tiles))
        return self.U

    def apply(self, U, W0, W, want_dw):
        W.copy_(W0 + U.float())
        return U.float()


def _cols_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emcid_amd import hip
        from emcid_amd.edit_engine import _all_reduce_sum
        from emcid_amd.emcid_main import _sdxl_split, _broadcast_, sdxl_rank_split
        from emcid_amd.edit_engine import ConceptShard
        from oracle import emcid_oracle as orc
        torch.set_num_threads(1)
        N, d, h, lam, ew, left = 24, (3072 if world == 8 else 512), 16, 40.0, 0.6, 2      # world 8: 24 column tiles over 8 ranks
        g = torch.Generator().manual_seed(5)                      # identical inputs on every rank (after the K all-gather)
        K = torch.randn(N, d, generator=g) * 0.3
        Zc = torch.randn(N, h, generator=g)
        zs = torch.randn(h, N, generator=g)
        x = torch.randn(2 * d, d, generator=g)
        C = (x.t() @ x) / (2 * d)
        W0 = torch.randn(h, d, generator=g) * 0.02
        _, _, upd = orc.closed_form_layer(K, Zc, zs, C, lam, ew, left)
        tiles = hip.column_tiles(rank, world, d // 128)
        gathered = [None] * world
        dist.all_gather_object(gathered, tiles)
        assert sorted(sum(gathered, [])) == list(range(d // 128))          # the ranks' tiles partition the columns
        W = torch.empty(h, d)
        out = hip.edit_layer_dual_cols(K, Zc, zs.t().contiguous(), None, 0, ew, left, W0, W, tiles,
                                       lambda t: _all_reduce_sum(t, None), backend=_TorchColsBackend(K, Zc, zs.t(), C, lam, ew, left))
        scale = upd.abs().max().item()
        assert (out["dW"].double() - upd).abs().max().item() <= 1e-6 * scale
        assert (W - (W0 + upd.float())).abs().max().item() <= 1e-6 * max(scale, 1.0)
        every = [None] * world
        dist.all_gather_object(every, W.numpy().tobytes())
        assert all(b == every[0] for b in every)                           # bit-identical weights on every rank
        # ---- SDXL: TE1 on the first group, TE2 on the rest; the roots broadcast their encoder's weights ----------------
        which, sub, (g1, g2) = sdxl_rank_split(rank, world)
        split = _sdxl_split(ConceptShard(rank, world, None))
        assert split["which"] == which and split["shard"].rank == sub and split["shard"].world == (g1 if which == 1 else g2)
        t = torch.full((3,), float(rank))
        _all_reduce_sum(t, split["shard"].group)                           # a collective inside the encoder's own group
        members = list(range(g1)) if which == 1 else list(range(g1, world))
        assert t[0].item() == float(sum(members))
        w1, w2 = torch.full((4,), -1.0), torch.full((4,), -1.0)
        if rank == split["roots"][0]:
            w1.fill_(11.0)
        if rank == split["roots"][1]:
            w2.fill_(22.0)
        _broadcast_(w1, split["roots"][0])
        _broadcast_(w2, split["roots"][1])
        assert w1[0].item() == 11.0 and w2[0].item() == 22.0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_column_sharded_solve_and_sdxl_groups_gloo(world):
    """The column-sharded layer solve (partial S and U summed over the ranks) equals the single-process closed form and
    leaves bit-identical weights on every rank; the SDXL TE1 / TE2 rank groups and their weight broadcasts."""
    mp.spawn(_cols_worker, args=(world, _free_port()), nprocs=world, join=True)


def test_column_tiles_are_balanced():
    from emcid_amd import hip
    for world in (1, 2, 3, 4, 7, 8):
        for n_tiles in (24, 40):
            tiles = [hip.column_tiles(r, world, n_tiles) for r in range(world)]
            assert sorted(sum(tiles, [])) == list(range(n_tiles))
            cost = [sum(t + 1 for t in ts) for ts in tiles]                 # tile t of the triangular factor costs ~ t + 1
            assert max(cost) <= 1.25 * (sum(cost) / world) + 1, (world, n_tiles, cost)


def _empty_shard_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emcid_amd import runningstats as rs
        st = rs.SecondMoment()
        if rank == 0:       # rank 1 was dealt no caption: its statistic is empty
            st.load_state_dict({"count": 3, "mom2": np.eye(4, dtype=np.float32)})
        try:
            st.all_reduce_()
            verdict = "returned"
        except RuntimeError as e:
            verdict = "raised" if "every rank" in str(e) else f"other: {e}"
        with open(os.path.join(tmp, f"verdict{rank}.txt"), "w") as f:
            f.write(verdict)
    finally:
        dist.destroy_process_group()


def test_empty_caption_shard_fails_on_every_rank(tmp_path):
    """A rank without data must not leave its peers inside the all-reduce: every rank raises the same error (round-2 review)."""
    mp.spawn(_empty_shard_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f"verdict{r}.txt").read() for r in range(2)] == ["raised", "raised"]
