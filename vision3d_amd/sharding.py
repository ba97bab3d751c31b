"""Sharding of the operator hot path across the GPUs of one node.

Two cases, one process per GPU (``torch.distributed``, backend ``nccl`` == RCCL over xGMI):

* Batched operators (FPS, ball query, kNN, three_nn, interpolation, gathers): clouds are independent, so the batch
  dimension is split contiguously across ranks -- ``batch_bounds`` / ``shard_batch`` -- and NO collective touches
  the data path.
* Scene-scale matching (N_q x N_s feature distance + top-k): the reference/support set is split into ``world``
  contiguous blocks; every rank runs the fused distance + top-k over its block (indices offset to global), then ONE
  exchange step -- an all-gather of the (N_q, k) partial lists, ``N_q * k * 12`` bytes per rank -- and a local
  merge by (distance, index), which reproduces the single-GPU result exactly.  On MI355X's fully connected xGMI
  mesh every peer's shard arrives on its own link, so the exchange is one step, not a ring of hops.

The search and merge kernels are injected (``search_fn`` / ``merge_fn``) only so the collective plumbing can be
exercised with ``gloo`` on CPU in the tests; the defaults are the HIP kernels and there is no CPU fallback.
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def batch_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of ``total`` items owned by ``rank``; the first ``total % world`` ranks get one more."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_batch(t: torch.Tensor, world: Optional[int] = None, rank: Optional[int] = None) -> torch.Tensor:
    """The slice of the leading (cloud) dimension this rank works on."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    b, e = batch_bounds(t.shape[0], world, rank)
    return t[b:e]


def _hip_search(x, y, k, index_offset):
    from . import ext

    d = x.new_empty((1, x.shape[0], k))
    i = torch.empty((1, x.shape[0], k), dtype=torch.long, device=x.device)
    ext.knn_feat(x.unsqueeze(0).contiguous(), y.unsqueeze(0).contiguous(), d, i, k, index_offset=index_offset)
    return d[0], i[0]


def _hip_merge(parts_d, parts_i):
    from . import ext

    out_d = parts_d.new_empty(parts_d.shape[1:])
    out_i = torch.empty(parts_i.shape[1:], dtype=torch.long, device=parts_i.device)
    ext.topk_merge(parts_d.contiguous(), parts_i.contiguous(), out_d, out_i)
    return out_d, out_i


def _all_gather_rows(part: torch.Tensor, group=None) -> torch.Tensor:
    """(rows, k) on every rank -> (world * rows, k), rank-major.  RCCL gathers device tensors directly; gloo (CPU tests,
    one-GPU rehearsals of the rank plumbing) only moves host memory, so device tensors are staged through the host."""
    world = dist.get_world_size(group)
    part = part.contiguous()
    if part.is_cuda and dist.get_backend(group) == "gloo":
        host = part.cpu()
        out = host.new_empty((world * host.shape[0],) + tuple(host.shape[1:]))
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(part.device)
    out = part.new_empty((world * part.shape[0],) + tuple(part.shape[1:]))
    dist.all_gather_into_tensor(out, part, group=group)
    return out


def knn_reference_sharded(
    queries: torch.Tensor,
    local_refs: torch.Tensor,
    k: int,
    ref_offset: int,
    group=None,
    search_fn: Callable = _hip_search,
    merge_fn: Callable = _hip_merge,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Global k nearest references of every query when the reference set is partitioned across ranks.

    queries (N_q, C) replicated on every rank; local_refs (N_local, C) this rank's block, whose first row has
    global index ``ref_offset``.  Returns (distances (N_q, k), global indices (N_q, k)) on every rank, identical
    to a single-device search over the concatenated reference set (ties broken by the lower global index).
    """
    part_d, part_i = search_fn(queries, local_refs, k, ref_offset)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return part_d, part_i
    world = dist.get_world_size(group)
    rows = part_d.shape[0]
    # the single exchange step of the path: two small all-gathers (fp32 distances, int64 indices); the rank-major
    # concatenation is the (world, rows, k) layout the merge kernel reads
    all_d = _all_gather_rows(part_d, group)
    all_i = _all_gather_rows(part_i, group)
    return merge_fn(all_d.view(world, rows, k), all_i.view(world, rows, k))


def _hip_select(row_i, row_d, col_i, col_d, mutual):
    from . import ext

    n, k = row_i.shape
    m = col_i.shape[0]
    bound = max(k * (min(n, m) if mutual else n + m), 1)
    idx = torch.empty((2, bound), dtype=torch.long, device=row_i.device)
    dist_ = torch.empty((bound,), dtype=torch.float32, device=row_i.device)
    count = torch.zeros((), dtype=torch.long, device=row_i.device)
    ext.mutual_select_lists(row_i.unsqueeze(0).contiguous(), row_d.unsqueeze(0).contiguous(),
                            col_i.unsqueeze(0).contiguous(), col_d.unsqueeze(0).contiguous(), m, out_row=idx[0],
                            out_col=idx[1], out_score=dist_, num_out=count, largest=False, mutual=mutual)
    c = int(count.item())
    return idx[0, :c], idx[1, :c], dist_[:c]


def mutual_topk_reference_sharded(queries, local_refs, k, ref_offset, group=None, search_fn=_hip_search,
                                  merge_fn=_hip_merge):
    """Row side of ops/mutual_topk_select.py:33 under reference-set sharding: see knn_reference_sharded.  The column
    side (top-k queries of every local reference) needs no exchange because columns are not split."""
    row = knn_reference_sharded(queries, local_refs, k, ref_offset, group, search_fn, merge_fn)
    col = search_fn(local_refs, queries, k, 0)
    return row, col


def mutual_matching_sharded(queries, local_refs, k, ref_offset, num_refs_total, mutual=True, group=None,
                            search_fn=_hip_search, merge_fn=_hip_merge, select_fn=_hip_select):
    """Mutual (or union) top-k matching of ``queries`` against a reference set partitioned across ranks -- BASELINE
    config 5 on a node: (query indices, global reference indices, distances), identical on every rank and identical to
    the single-device ``ops.extract_correspondences_from_feats`` / ``mutual_topk_select`` result.

    Exchange steps: the all-gather of the (N_q, k) partial row lists inside ``knn_reference_sharded`` and ONE all-gather
    of the column lists (every rank owns the complete top-k of ITS references, so they are concatenated, not merged:
    12 * k bytes per reference).  Blocks may differ in size by one row; they are padded to the largest for the
    collective and trimmed after it.
    """
    (row_d, row_i), (col_d, col_i) = mutual_topk_reference_sharded(queries, local_refs, k, ref_offset, group, search_fn,
                                                                   merge_fn)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        sizes = [batch_bounds(num_refs_total, world, r) for r in range(world)]
        widest = max(e - b for b, e in sizes)
        pad = widest - col_d.shape[0]
        if pad:
            col_d = torch.cat([col_d, col_d.new_full((pad, k), float("inf"))], 0)
            col_i = torch.cat([col_i, col_i.new_full((pad, k), -1)], 0)
        all_d = _all_gather_rows(col_d, group)
        all_i = _all_gather_rows(col_i, group)
        keep = torch.cat([torch.arange(r * widest, r * widest + (e - b)) for r, (b, e) in enumerate(sizes)])
        keep = keep.to(all_d.device)
        col_d, col_i = all_d[keep], all_i[keep]
    return select_fn(row_i, row_d, col_i, col_d, mutual)
