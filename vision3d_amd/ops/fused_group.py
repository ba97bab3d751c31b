"""Fused neighbourhood grouping for the batch-mode layers.

``sa_group`` computes what ``SAConv.forward`` builds before its MLP (``vision3d/layers/pointnet2.py:44-47``) and
``edge_group`` what ``EdgeConv.forward`` builds (``layers/edge_conv.py:59-61``) -- each in ONE kernel that writes the
MLP input directly instead of group_gather + subtract + cat (three HBM passes over the largest tensor of the layer).
Values are bit-identical to the reference formulation.  Backward reuses the atomic-free group-gather backward.
"""
import torch
from torch import Tensor

from .. import ext


class _SAGroup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_points, s_points, s_feats, neighbor_indices):
        q_points, s_points = q_points.contiguous(), s_points.contiguous()
        s_feats = s_feats.contiguous() if s_feats is not None else None
        neighbor_indices = neighbor_indices.contiguous()
        b, m, k = neighbor_indices.shape
        c = s_feats.shape[1] if s_feats is not None else 0
        out = s_points.new_empty((b, c + 3, m, k))
        ext.sa_group(q_points, s_points, s_feats, neighbor_indices, out)
        ctx.save_for_backward(neighbor_indices)
        ctx.sizes = (c, s_points.shape[2])
        return out

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        c, n = ctx.sizes
        b = grad.shape[0]
        g_feats = g_points = g_q = None
        g_off = grad[:, c:].contiguous()
        if ctx.needs_input_grad[2] and c > 0:
            g_feats = grad.new_empty((b, c, n))
            ext.group_gather_backward(grad[:, :c].contiguous(), idx, g_feats, n)
        if ctx.needs_input_grad[1]:
            g_points = grad.new_empty((b, 3, n))
            ext.group_gather_backward(g_off, idx, g_points, n)
        if ctx.needs_input_grad[0]:
            g_q = -g_off.sum(dim=3)
        return g_q, g_points, g_feats, None


def sa_group(q_points: Tensor, s_points: Tensor, s_feats, neighbor_indices: Tensor) -> Tensor:
    """q_points (B,3,M), s_points (B,3,N), s_feats (B,C,N) or None, neighbor_indices (B,M,K) -> (B, C+3, M, K) =
    ``cat([group_gather(s_feats, idx), group_gather(s_points, idx) - q_points.unsqueeze(3)], dim=1)``."""
    if not s_points.is_cuda:
        raise RuntimeError("sa_group: CUDA tensors required (no CPU fallback in vision3d_amd)")
    return _SAGroup.apply(q_points, s_points, s_feats, neighbor_indices)


class _EdgeGroup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_feats, s_feats, neighbor_indices):
        q_feats, s_feats = q_feats.contiguous(), s_feats.contiguous()
        neighbor_indices = neighbor_indices.contiguous()
        b, m, k = neighbor_indices.shape
        c = s_feats.shape[1]
        out = s_feats.new_empty((b, 2 * c, m, k))
        ext.edge_group(q_feats, s_feats, neighbor_indices, out)
        ctx.save_for_backward(neighbor_indices)
        ctx.sizes = (c, s_feats.shape[2])
        return out

    @staticmethod
    def backward(ctx, grad):
        (idx,) = ctx.saved_tensors
        c, n = ctx.sizes
        b = grad.shape[0]
        g_q = g_s = None
        g_diff = grad[:, c:].contiguous()
        if ctx.needs_input_grad[1]:
            g_s = grad.new_empty((b, c, n))
            ext.group_gather_backward(g_diff, idx, g_s, n)
        if ctx.needs_input_grad[0]:
            g_q = (grad[:, :c] - g_diff).sum(dim=3)
        return g_q, g_s, None


def edge_group(q_feats: Tensor, s_feats: Tensor, neighbor_indices: Tensor) -> Tensor:
    """q_feats (B,C,M), s_feats (B,C,N), neighbor_indices (B,M,K) -> (B, 2C, M, K) =
    ``cat([q.expand, group_gather(s_feats, idx) - q.expand], dim=1)``."""
    if not s_feats.is_cuda:
        raise RuntimeError("edge_group: CUDA tensors required (no CPU fallback in vision3d_amd)")
    return _EdgeGroup.apply(q_feats, s_feats, neighbor_indices)
