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


def test_batch_bounds_cover_everything():
    for total in (0, 1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [sharding.batch_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
