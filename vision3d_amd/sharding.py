"""Sharding of the operator hot path across the GPUs of one node.

Two cases, one process per GPU (``torch.distributed``, backend ``nccl`` == RCCL over xGMI):

* Batched operators (FPS, ball query, kNN, three_nn, interpolation, gathers): clouds are independent, so the batch
  dimension is split contiguously across ranks -- ``batch_bounds`` / ``shard_batch`` -- and NO collective touches
  the data path.  Pack-mode inputs (``points (sum N_b, 3)`` + ``lengths (B,)``: KPConv / GeoTransformer pyramids,
  BASELINE config 4) split the same way by ``lengths`` -- ``shard_pack`` -- each rank builds the neighbour tables of ITS
  clouds; a table row holds indices into that rank's packed array (sentinel = that rank's row count), and
  ``globalize_pack_table`` maps it onto the whole batch's numbering when a caller wants to compare or concatenate.
* Scene-scale matching (N_q x N_s feature distance + top-k): the reference/support set is split into ``world``
  contiguous blocks; every rank runs the fused distance + top-k over its block (indices offset to global), then ONE
  exchange step -- an all-gather of the (N_q, k) partial lists, ``N_q * k * 12`` bytes per rank -- and a local
  merge by (distance, index), which reproduces the single-GPU result exactly.  On MI355X's fully connected xGMI
  mesh every peer's shard arrives on its own link, so the exchange is one step, not a ring of hops.

The search and merge kernels are injected (``search_fn`` / ``merge_fn``) only so the collective plumbing can be
exercised with ``gloo`` on CPU in the tests; the defaults are the HIP kernels and there is no CPU fallback.
"""
from typing import Callable, NamedTuple, Optional, Tuple

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


class PackShard(NamedTuple):
    """This rank's contiguous clouds of a pack-mode batch."""
    points: torch.Tensor   # (sum of this rank's lengths, C): a view of the batch's packed rows
    lengths: torch.Tensor  # (clouds of this rank,): a view of the batch's lengths
    cloud_begin: int       # first cloud of this rank in the batch
    cloud_end: int
    row_begin: int         # first packed row of this rank in the batch
    row_end: int


def shard_pack(points: torch.Tensor, lengths: torch.Tensor, world: Optional[int] = None,
               rank: Optional[int] = None) -> PackShard:
    """Pack-mode batch sharding ("pack-mode: split by ``lengths``"; the reference gives every DDP rank its
    own samples through ``DistributedSampler``, utils/dataloader.py:90-93, and builds the pyramid per rank,
    models/graph_pyramid/graph_pyramid.py:9-63).  Clouds [cloud_begin, cloud_end) = ``batch_bounds`` over
    ``len(lengths)``; their packed rows are the contiguous range [row_begin, row_end) from a cumulative sum of the
    lengths.  No collective; the two row bounds are read from ``lengths`` once (collate hands the lengths over on the
    host; a device tensor costs one read-back)."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    cb, ce = batch_bounds(int(lengths.shape[0]), world, rank)
    ends = torch.cumsum(lengths.to(torch.long), dim=0)
    bounds = torch.stack([ends[cb - 1] if cb > 0 else ends.new_zeros(()), ends[ce - 1] if ce > 0 else ends.new_zeros(())])
    rb, re = (int(v) for v in bounds.tolist())
    if int(points.shape[0]) < re:
        raise ValueError(f"shard_pack: lengths sum past row {re} but points has {int(points.shape[0])} rows")
    return PackShard(points[rb:re], lengths[cb:ce], cb, ce, rb, re)


def globalize_pack_table(table: torch.Tensor, local_rows: int, row_begin: int, global_rows: int) -> torch.Tensor:
    """A neighbour table built on one rank's shard (entries in [0, local_rows), absent = ``local_rows``) in the numbering
    of the whole batch (entries + ``row_begin``, absent = ``global_rows``): the per-rank tables of a sharded search,
    globalized with the SUPPORT side's bounds and concatenated in rank order, equal the single-process table."""
    return torch.where(table >= local_rows, torch.full_like(table, global_rows), table + row_begin)


def _hip_radius_search(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit):
    from .ops.radius_search import radius_search_pack_mode

    return radius_search_pack_mode(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit)


def radius_search_pack_sharded(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit,
                               world: Optional[int] = None, rank: Optional[int] = None,
                               search_fn: Callable = _hip_radius_search):
    """``ops.radius_search_pack_mode`` (reference: ops/radius_search.py:44-71) on this rank's clouds of a pack-mode batch.
    Returns (table over the rank's queries, in the rank's own support numbering; the query ``PackShard``; the support
    ``PackShard``).  No collective: a query only ever sees supports of its own cloud."""
    qs = shard_pack(q_points, q_lengths, world, rank)
    ss = shard_pack(s_points, s_lengths, world, rank)
    table = search_fn(qs.points, ss.points, qs.lengths, ss.lengths, radius, neighbor_limit)
    return table, qs, ss


def gather_pack_tables(table: torch.Tensor, supports: PackShard, global_rows: int, group=None) -> torch.Tensor:
    """Off the data path (evaluation, tests): the whole batch's table on every rank from the per-rank tables of
    ``radius_search_pack_sharded`` -- globalize, pad to the largest shard, ONE all-gather, trim.  Equal to the table a
    single process computes over the whole batch."""
    local_rows = supports.row_end - supports.row_begin
    table = globalize_pack_table(table, local_rows, supports.row_begin, global_rows)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return table
    world = dist.get_world_size(group)
    counts = [None] * world
    dist.all_gather_object(counts, int(table.shape[0]), group=group)
    widest = max(counts)
    if table.shape[0] < widest:
        table = torch.cat([table, table.new_full((widest - table.shape[0], table.shape[1]), global_rows)], 0)
    everything = _all_gather_rows(table, group)
    return torch.cat([everything[r * widest:r * widest + counts[r]] for r in range(world)], 0)


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
