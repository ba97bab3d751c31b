"""World-size-2 (and 3) gloo tests of the multi-GPU plumbing on CPU: batch partitioning without a collective, and
reference-set sharding with one all-gather + merge.  The HIP search/merge kernels are replaced by NumPy stand-ins of
the SAME contract (they are checked against the real kernels in tests/test_gpu_parity.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vision3d_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _np_search(x, y, k, offset):
    d = torch.cdist(x.double(), y.double())
    order = torch.argsort(d, dim=1, stable=True)[:, :k]
    dist_k = torch.gather(d, 1, order).float()
    idx = order + offset
    if order.shape[1] < k:  # fewer references than k on this shard: pad with (inf, -1)
        pad = k - order.shape[1]
        dist_k = torch.cat([dist_k, torch.full((x.shape[0], pad), float("inf"))], 1)
        idx = torch.cat([idx, torch.full((x.shape[0], pad), -1, dtype=torch.long)], 1)
    return dist_k, idx


def _np_merge(parts_d, parts_i):
    g, rows, k = parts_d.shape
    d = parts_d.permute(1, 0, 2).reshape(rows, g * k)
    i = parts_i.permute(1, 0, 2).reshape(rows, g * k)
    invalid = i < 0
    key_i = torch.where(invalid, torch.full_like(i, 2 ** 62), i)
    key_d = torch.where(invalid, torch.full_like(d, float("inf")), d)
    order = np.lexsort((key_i.numpy(), key_d.numpy()), axis=1)[:, :k]
    order = torch.from_numpy(order)
    return torch.gather(d, 1, order), torch.gather(i, 1, order)


def _np_select(row_i, row_d, col_i, col_d, mutual):
    """NumPy stand-in of ext.mutual_select_lists (checked against the kernel in tests/test_gpu_next_rows.py)."""
    n, m = row_i.shape[0], col_i.shape[0]
    rows = torch.zeros((n, m), dtype=torch.bool)
    cols = torch.zeros((n, m), dtype=torch.bool)
    dmat = torch.zeros((n, m))
    for r in range(n):
        for j in range(row_i.shape[1]):
            if row_i[r, j] >= 0:
                rows[r, row_i[r, j]] = True
                dmat[r, row_i[r, j]] = row_d[r, j]
    for c in range(m):
        for j in range(col_i.shape[1]):
            if col_i[c, j] >= 0:
                cols[col_i[c, j], c] = True
                dmat[col_i[c, j], c] = col_d[c, j]
    corr = (rows & cols) if mutual else (rows | cols)
    r, c = torch.nonzero(corr, as_tuple=True)
    return r, c, dmat[r, c]


def _matching_worker(rank, world, port, n_ref, k, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1)
        q = torch.randn(30, 6, generator=g)
        refs = torch.cat([q[:12] + 0.01 * torch.randn(12, 6, generator=g), torch.randn(n_ref - 12, 6, generator=g)])
        b, e = sharding.batch_bounds(n_ref, world, rank)
        ok = True
        for mutual in (True, False):
            got = sharding.mutual_matching_sharded(q, refs[b:e], k, b, n_ref, mutual=mutual, search_fn=_np_search,
                                                   merge_fn=_np_merge, select_fn=_np_select)
            rd, ri = _np_search(q, refs, k, 0)
            cd, ci = _np_search(refs, q, k, 0)
            want = _np_select(ri, rd, ci, cd, mutual)
            ok = ok and all(torch.equal(a, w) if a.dtype == torch.long else torch.allclose(a, w)
                            for a, w in zip(got, want))
            if mutual:  # the planted pairs are found
                ok = ok and int((got[0] < 12).sum()) >= 12
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_ref,k", [(2, 41, 1), (2, 40, 2), (3, 31, 1)])
def test_sharded_mutual_matching_equals_single_shot(world, n_ref, k):
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_matching_worker, args=(world, port, n_ref, k, out), nprocs=world, join=True)
        assert all(out.get(r) for r in range(world)), dict(out)


def _worker(rank, world, port, n_ref, k, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)  # same data on every rank
        q = torch.randn(40, 8, generator=g)
        refs = torch.randn(n_ref, 8, generator=g)
        refs[7] = refs[3]  # an exact tie across the shard boundary region
        b, e = sharding.batch_bounds(n_ref, world, rank)
        d, i = sharding.knn_reference_sharded(q, refs[b:e], k, b, search_fn=_np_search, merge_fn=_np_merge)
        full_d, full_i = _np_search(q, refs, k, 0)
        ok = torch.equal(i, full_i) and torch.allclose(d, full_d)
        # batch sharding: the union of the shards is the batch, in order, with no communication
        batch = torch.arange(11 * 3).view(11, 3)
        mine = sharding.shard_batch(batch)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ok = ok and torch.equal(torch.cat(gathered), batch)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_ref,k", [(2, 50, 3), (2, 9, 1), (3, 31, 4)])
def test_reference_set_sharding_equals_single_shot(world, n_ref, k):
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, n_ref, k, out), nprocs=world, join=True)
        assert all(out.get(r) for r in range(world)), dict(out)


def _pack_worker(rank, world, port, out):
    """Pack-mode batch sharding (BASELINE config 4): every rank searches its own clouds with an injected CPU search (the
    oracle's restatement of ops/radius_search.py:44-71); the per-rank tables, globalized and gathered, are the table one
    process computes over the whole batch.  Clouds of different sizes, one of them empty, fewer clouds than ranks."""
    import oracle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def cpu_search(q, s, ql, sl, radius, limit):
            return torch.from_numpy(oracle.radius_search_pack(q.numpy(), s.numpy(), ql.numpy(), sl.numpy(), radius, limit))

        ok = True
        for q_lens, s_lens in (((40, 0, 25, 31, 17), (90, 12, 60, 75, 33)), ((50,), (120,))):
            g = torch.Generator().manual_seed(3)  # same batch on every rank
            ql, sl = torch.tensor(q_lens), torch.tensor(s_lens)
            q = torch.rand((int(ql.sum()), 3), generator=g)
            s = torch.rand((int(sl.sum()), 3), generator=g)
            table, qs, ss = sharding.radius_search_pack_sharded(q, s, ql, sl, 0.25, 9, search_fn=cpu_search)
            ok = ok and table.shape == (qs.row_end - qs.row_begin, 9)
            ok = ok and bool((table <= ss.row_end - ss.row_begin).all())  # local numbering, local sentinel
            whole = sharding.gather_pack_tables(table, ss, s.shape[0])
            want = cpu_search(q, s, ql, sl, 0.25, 9)
            ok = ok and torch.equal(whole, want)
            # the shards tile the batch: clouds and rows, in rank order, nothing twice
            spans = [None] * world
            dist.all_gather_object(spans, (qs.cloud_begin, qs.cloud_end, qs.row_begin, qs.row_end))
            ok = ok and spans[0][0] == 0 and spans[0][2] == 0 and spans[-1][1] == len(q_lens) and spans[-1][3] == q.shape[0]
            ok = ok and all(spans[r][1] == spans[r + 1][0] and spans[r][3] == spans[r + 1][2] for r in range(world - 1))
            ok = ok and torch.equal(qs.points, q[qs.row_begin:qs.row_end]) and torch.equal(qs.lengths, ql[qs.cloud_begin:qs.cloud_end])
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pack_mode_batch_sharding_equals_single_shot(world):
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_pack_worker, args=(world, port, out), nprocs=world, join=True)
        assert all(out.get(r) for r in range(world)), dict(out)


def test_shard_pack_bounds_and_globalize():
    lengths = torch.tensor([5, 0, 9, 3, 7])
    points = torch.arange(24 * 3, dtype=torch.float32).view(24, 3)
    for world in (1, 2, 3, 8):
        shards = [sharding.shard_pack(points, lengths, world, r) for r in range(world)]
        assert torch.equal(torch.cat([s.points for s in shards]), points)
        assert torch.equal(torch.cat([s.lengths for s in shards]), lengths)
        assert all(int(s.lengths.sum()) == s.row_end - s.row_begin == s.points.shape[0] for s in shards)
    table = torch.tensor([[0, 2, 4], [4, 4, 4]])
    assert sharding.globalize_pack_table(table, 4, 10, 99).tolist() == [[10, 12, 99], [99, 99, 99]]
    with pytest.raises(ValueError):
        sharding.shard_pack(points[:20], lengths, 2, 1)


def test_batch_bounds_cover_everything():
    for total in (0, 1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [sharding.batch_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
