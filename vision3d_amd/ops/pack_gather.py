"""PACK-mode row gathers -- mirror ``vision3d/ops/index_select.py:4-33``, ``ops/pooling.py:6-39``,
``ops/nearest_interpolate.py:6-20`` and ``ops/knn_interpolate.py:43-76`` (same names, signatures, returns).

The reference concatenates a zero row to the features, gathers an (M,K,C) tensor with ``index_select`` and reduces it.
Here each operator is one or two kernels that reduce in registers (``ext.local_maxpool_pack`` etc.): neither the padded
copy nor the (M,K,C) intermediate is written.  Backward is atomic-free: the neighbour table is inverted once into a CSR
(``ext.rows_csr_build``), cached on the table tensor itself so that every layer sharing the table (all KPConv blocks of
a pyramid stage) reuses it, and each source row is summed in ascending table position -- bit-reproducible.
"""
from typing import Optional

import torch
from torch import Tensor

from .. import ext


def _native(*tensors) -> bool:
    return all(t.is_cuda for t in tensors)


def table_csr(indices: Tensor, num_sources: int):
    """CSR of ``indices`` over ``num_sources`` rows, cached on the tensor object (invalidated by in-place edits)."""
    cached = getattr(indices, "_v3d_csr", None)
    if cached is not None and cached[0] == (indices._version, int(num_sources), indices.data_ptr()):
        return cached[1]
    csr = ext.rows_csr_build(indices, num_sources)
    try:
        indices._v3d_csr = ((indices._version, int(num_sources), indices.data_ptr()), csr)
    except AttributeError:  # pragma: no cover  (tensor subclasses with slots)
        pass
    return csr


class _IndexSelectRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sources, indices):
        sources = sources.contiguous()
        indices = indices.contiguous()
        out = sources.new_empty(tuple(indices.shape) + (sources.shape[1],))
        ext.index_select_rows(sources, indices, out)
        ctx.table = indices
        ctx.num_sources = sources.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad):
        indices, n = ctx.table, ctx.num_sources
        c = grad.shape[-1]
        if grad.dtype != torch.float32 or c % 4 != 0:
            g = torch.zeros((n + 1, c), dtype=grad.dtype, device=grad.device)
            clamped = torch.where((indices >= 0) & (indices < n), indices, torch.full_like(indices, n))
            g.index_add_(0, clamped.view(-1), grad.reshape(-1, c))
            return g[:n], None
        grad = grad.contiguous().view(-1, c)
        out = grad.new_empty((n, c))
        flat = indices.view(-1, 1)
        ext.rows_scatter(grad, table_csr(indices, n), out, (flat.shape[0], 1), mode=0)
        return out, None


def index_select(inputs: Tensor, indices: Tensor, dim: int) -> Tensor:
    """``inputs`` indexed along ``dim`` by an N-D ``indices`` (index_select.py:4-33).

    2-D CUDA inputs indexed along dim 0 -- the pack-mode gather of every KPConv / EdgeConv / pooling layer -- run on the
    native row-gather kernel, where an index equal to ``inputs.shape[0]`` (the reference's appended zero row) or outside
    the valid range reads as zeros.  Other layouts keep the reference's formulation (``Tensor.index_select`` + view).
    """
    if inputs.dim() == 2 and dim in (0, -2) and _native(inputs, indices) and indices.dtype == torch.long:
        return _IndexSelectRows.apply(inputs, indices)
    outputs = inputs.index_select(dim, indices.reshape(-1))
    if indices.dim() > 1:
        if dim < 0:
            dim += inputs.dim()
        output_shape = inputs.shape[:dim] + indices.shape + inputs.shape[dim + 1:]
        outputs = outputs.view(*output_shape)
    return outputs


def _require_cuda(name, *tensors):
    if not _native(*tensors):
        raise RuntimeError(f"{name}: CUDA tensors required (no CPU fallback in vision3d_amd)")


class _LocalMaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, neighbor_indices):
        # decided before .contiguous(): grad mode is off in here, so the copy of a non-contiguous input has
        # requires_grad == False although the input takes part in autograd
        need_grad = ctx.needs_input_grad[0] and neighbor_indices.shape[1] <= 255
        feats = feats.contiguous()
        neighbor_indices = neighbor_indices.contiguous()
        m, k = neighbor_indices.shape
        pooled = feats.new_empty((m, feats.shape[1]))
        argk = torch.empty((m, feats.shape[1]), dtype=torch.uint8, device=feats.device) if need_grad else None
        ext.local_maxpool_pack(feats, neighbor_indices, pooled, argk)
        ctx.table, ctx.argk, ctx.num_sources = neighbor_indices, argk, feats.shape[0]
        return pooled

    @staticmethod
    def backward(ctx, grad):
        if ctx.argk is None:
            raise RuntimeError("local_maxpool_pack_mode: backward needs K <= 255")
        grad = grad.contiguous()
        out = grad.new_empty((ctx.num_sources, grad.shape[1]))
        ext.rows_scatter(grad, table_csr(ctx.table, ctx.num_sources), out, tuple(ctx.table.shape), mode=2,
                         argk=ctx.argk)
        return out, None


def local_maxpool_pack_mode(feats: Tensor, neighbor_indices: Tensor) -> Tensor:
    """feats (N, C), neighbor_indices (M, K) -> (M, C) max over the neighbours (absent ones count as zeros)."""
    _require_cuda("local_maxpool_pack_mode", feats, neighbor_indices)
    if feats.dtype != torch.float32 or feats.shape[1] % 4 != 0:
        padded = torch.cat((feats, torch.zeros_like(feats[:1, :])), 0)
        return index_select(padded, neighbor_indices, dim=0).max(1)[0]
    return _LocalMaxPool.apply(feats, neighbor_indices)


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, lengths):
        feats = feats.contiguous()
        lengths = lengths.contiguous()
        pooled = feats.new_empty((lengths.shape[0], feats.shape[1]))
        ext.global_avgpool_pack(feats, lengths, pooled)
        ctx.save_for_backward(lengths)
        return pooled

    @staticmethod
    def backward(ctx, grad):
        (lengths,) = ctx.saved_tensors
        per_row = grad / lengths.to(grad.dtype).unsqueeze(1)
        return per_row.repeat_interleave(lengths, dim=0), None


def global_avgpool_pack_mode(feats: Tensor, lengths: Tensor) -> Tensor:
    """feats (N, C), lengths (B,) -> per-cloud mean (B, C); no ``.item()`` loop over clouds."""
    _require_cuda("global_avgpool_pack_mode", feats)
    lengths = lengths.to(device=feats.device, dtype=torch.long)
    if feats.dtype != torch.float32:
        return _GlobalAvgPool.apply(feats.float(), lengths).to(feats.dtype)
    return _GlobalAvgPool.apply(feats, lengths)


def nearest_interpolate_pack_mode(inputs: Tensor, neighbor_indices: Tensor) -> Tensor:
    """inputs (n1, d), neighbor_indices (n2, K): features of the FIRST neighbour of every query (n2, d)."""
    _require_cuda("nearest_interpolate_pack_mode", inputs, neighbor_indices)
    return index_select(inputs, neighbor_indices[:, 0].contiguous(), dim=0)


class _RowsWeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, neighbor_indices, weights):
        feats = feats.contiguous()
        out = feats.new_empty((neighbor_indices.shape[0], feats.shape[1]))
        ext.rows_weighted_sum(feats, neighbor_indices, weights, out)
        ctx.table, ctx.weights, ctx.num_sources = neighbor_indices, weights, feats.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad):
        grad = grad.contiguous()
        out = grad.new_empty((ctx.num_sources, grad.shape[1]))
        ext.rows_scatter(grad, table_csr(ctx.table, ctx.num_sources), out, tuple(ctx.table.shape), mode=1,
                         weights=ctx.weights)
        return out, None, None


def knn_interpolate_pack_mode(
    q_points: Tensor,
    s_points: Tensor,
    s_feats: Tensor,
    neighbor_indices: Tensor,
    k: Optional[int] = None,
    eps: float = 1e-8,
) -> Tensor:
    """Inverse-squared-distance interpolation of ``s_feats`` (M, C) at ``q_points`` (N, 3) over the first ``k`` columns of
    ``neighbor_indices`` (N, K); entries equal to M are absent neighbours (knn_interpolate.py:43-76).

    Gradients flow to ``s_feats`` (the learned quantity); the point coordinates are treated as constants.
    """
    _require_cuda("knn_interpolate_pack_mode", q_points, s_points, s_feats, neighbor_indices)
    table = neighbor_indices.contiguous()
    k_used = table.shape[1] if k is None else min(int(k), table.shape[1])
    weights = torch.empty((table.shape[0], k_used), dtype=torch.float32, device=table.device)
    ext.knn_interp_weights_pack(q_points.detach().float().contiguous(), s_points.detach().float().contiguous(), table,
                                weights, eps)
    if s_feats.dtype != torch.float32 or s_feats.shape[1] % 4 != 0:
        padded = torch.cat((s_feats, torch.zeros_like(s_feats[:1, :])), 0)
        knn_feats = index_select(padded, table[:, :k_used].contiguous(), dim=0)
        return (knn_feats * weights.to(s_feats.dtype).unsqueeze(-1)).sum(dim=1)
    return _RowsWeightedSum.apply(s_feats, table, weights)


def knn_interpolate(
    q_points: Tensor,
    s_points: Tensor,
    s_feats: Tensor,
    k: int = 3,
    eps: float = 1e-10,
    distance_limit: Optional[float] = None,
) -> Tensor:
    """Inverse-distance interpolation of ``s_feats`` (N, C) at ``q_points`` (M, 3) over the ``k`` nearest supports
    (knn_interpolate.py:10-40).  The search is the native 3-D kNN (the reference: pykeops), the (M, k, C) gather and its
    weighted sum are one fused kernel; gradients flow to ``s_feats`` through the ordered row scatter."""
    from .knn import keops_knn

    _require_cuda("knn_interpolate", q_points, s_points, s_feats)
    knn_distances, knn_indices = keops_knn(q_points.detach(), s_points.detach(), k)  # :32
    if distance_limit is not None:  # :33-35
        knn_distances = knn_distances.masked_fill(torch.gt(knn_distances, distance_limit), 1e10)
    weights = 1.0 / (knn_distances + eps)  # :36
    weights = weights / weights.sum(dim=1, keepdim=True)  # :37
    if s_feats.dtype != torch.float32 or s_feats.shape[1] % 4 != 0:
        knn_feats = index_select(s_feats, knn_indices, dim=0)  # :38
        return (knn_feats * weights.to(s_feats.dtype).unsqueeze(-1)).sum(dim=1)  # :39
    return _RowsWeightedSum.apply(s_feats, knn_indices.contiguous(), weights.contiguous())
