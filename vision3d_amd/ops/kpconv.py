"""KPConv neighbourhood aggregation -- the data path of ``KPConv.forward`` (``vision3d/layers/kpconv.py:97-149``).

``kpconv_gather`` returns ``weighted_feats`` (M, K, C) and ``neighbor_num`` (M,) straight from the inputs; ``kpconv``
adds the layer's remaining arithmetic (the einsum with the convolution weights -- a plain GEMM, left to the BLAS --, the
density normalisation and the bias), so a KPConv layer can call it with its own parameters.  Forward: one fused kernel
(``ext.kpconv_gather``: fp32-input MFMA, the gathered (M,H,C) tensor and the (M,H,K,3) differences never exist).
Backward w.r.t. the features: ``ext.kpconv_gather_backward`` recomputes the kernel-point influences and forms
``grad_neighbor_feats = influences^T @ grad_weighted`` on the MFMA, then the atomic-free row scatter of
``ops.pack_gather`` sums it per source in table order.
"""
from typing import Optional

import torch
from torch import Tensor

from .. import ext
from .pack_gather import table_csr


def _influences(q_points, s_points, neighbor_indices, kernel_points, sigma):
    """(M, K, H) kernel-point influences, the reference formulation (kpconv.py:100-110) with absent neighbours at 0."""
    n = s_points.shape[0]
    valid = (neighbor_indices >= 0) & (neighbor_indices < n)
    safe = torch.where(valid, neighbor_indices, torch.zeros_like(neighbor_indices))
    neighbors = s_points[safe] - q_points.unsqueeze(1)  # (M, H, 3)
    differences = neighbors.unsqueeze(2) - kernel_points  # (M, H, K, 3)
    sq = (differences ** 2).sum(dim=3)
    w = torch.clamp(1 - torch.sqrt(sq) / sigma, min=0.0) * valid.unsqueeze(2).to(sq.dtype)
    return w.transpose(1, 2)


class _KPConvGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_points, s_points, s_feats, neighbor_indices, kernel_points, sigma):
        q_points, s_points = q_points.contiguous(), s_points.contiguous()
        s_feats, neighbor_indices = s_feats.contiguous(), neighbor_indices.contiguous()
        kernel_points = kernel_points.contiguous()
        m, k, c = neighbor_indices.shape[0], kernel_points.shape[0], s_feats.shape[1]
        weighted = s_feats.new_empty((m, k, c))
        num = torch.empty((m,), dtype=torch.long, device=s_feats.device)
        ext.kpconv_gather(q_points, s_points, s_feats, neighbor_indices, kernel_points, weighted, num, sigma)
        ctx.save_for_backward(q_points, s_points, kernel_points)
        ctx.table, ctx.sigma, ctx.num_sources = neighbor_indices, float(sigma), s_feats.shape[0]
        ctx.mark_non_differentiable(num)
        return weighted, num

    @staticmethod
    def backward(ctx, grad_weighted, _grad_num):
        q_points, s_points, kernel_points = ctx.saved_tensors
        m, h = ctx.table.shape
        c = grad_weighted.shape[2]
        grad_neighbor = grad_weighted.new_empty((m, h, c))
        ext.kpconv_gather_backward(q_points, s_points, ctx.table, kernel_points, grad_weighted.contiguous(),
                                   grad_neighbor, ctx.sigma)
        out = grad_weighted.new_empty((ctx.num_sources, c))
        ext.rows_scatter(grad_neighbor.view(m * h, c), table_csr(ctx.table, ctx.num_sources), out, (m * h, 1), mode=0)
        return None, None, out, None, None, None


def kpconv_gather(q_points: Tensor, s_points: Tensor, s_feats: Tensor, neighbor_indices: Tensor, kernel_points: Tensor,
                  sigma: float):
    """q_points (M,3), s_points (N,3), s_feats (N,C), neighbor_indices (M,H), kernel_points (K,3) ->
    (weighted_feats (M,K,C), neighbor_num (M,) int64)."""
    if not s_feats.is_cuda:
        raise RuntimeError("kpconv_gather: CUDA tensors required (no CPU fallback in vision3d_amd)")
    h, k, c = neighbor_indices.shape[1], kernel_points.shape[0], s_feats.shape[1]
    if s_feats.dtype == torch.float32 and h <= 64 and k <= 16 and c % 4 == 0:
        return _KPConvGather.apply(q_points, s_points, s_feats, neighbor_indices, kernel_points, sigma)
    # shapes outside the fused kernel: the reference formulation over the native row gather
    from .pack_gather import index_select

    w = _influences(q_points, s_points, neighbor_indices, kernel_points, sigma).to(s_feats.dtype)
    neighbor_feats = index_select(s_feats, neighbor_indices, dim=0)
    weighted = torch.matmul(w, neighbor_feats)
    num = torch.gt(neighbor_feats.sum(dim=-1), 0.0).sum(dim=-1)
    return weighted, torch.max(num, torch.ones_like(num))


def kpconv(q_points: Tensor, s_points: Tensor, s_feats: Tensor, neighbor_indices: Tensor, kernel_points: Tensor,
           weights: Tensor, sigma: float, bias: Optional[Tensor] = None, groups: int = 1) -> Tensor:
    """Rigid KPConv forward (kpconv.py:97-149): ``weights`` (K, C_in, C_out) or (K, G, C_in/G, C_out/G)."""
    weighted, num = kpconv_gather(q_points, s_points, s_feats, neighbor_indices, kernel_points, sigma)
    if groups == 1:
        out = torch.einsum("mkc,kcd->md", weighted, weights)  # :119
    else:
        k = kernel_points.shape[0]
        grouped = weighted.view(-1, k, groups, weighted.shape[2] // groups)
        out = torch.einsum("mkgc,kgcd->mgd", grouped, weights).reshape(weighted.shape[0], -1)  # :122-124
    out = out / num.unsqueeze(1)  # :140
    if bias is not None:
        out = out + bias  # :148-149
    return out
