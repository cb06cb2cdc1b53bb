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
        assert sum(t.shape[0] for t in seen) == 10
        dist.barrier()
        with np.load(cache) as z:
            assert int(z["mom2.count"]) == 20 and float(z["mom2.mom2"][0, 0]) == float((data ** 2).sum())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8])
def test_world2_gloo(tmp_path, n_total):
    mp.spawn(_worker, args=(2, _free_port(), n_total, str(tmp_path)), nprocs=2, join=True)


def test_concept_shard_bounds_cover_all():
    from emcid_amd.edit_engine import ConceptShard
    for n in (1, 7, 8, 1000):
        for w in (1, 2, 4, 8):
            spans = [ConceptShard(r, w).bounds(n) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
