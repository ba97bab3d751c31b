"""``three_interpolate`` -- mirrors ``vision3d/ops/three_interpolate.py:10-66``."""
from torch import Tensor
from torch.autograd import Function

from .. import ext
from ._scatter_lists import scatter_lists


class _ThreeInterpolateFunction(Function):
    """q_feats[b, c, n] = sum_j s_feats[b, c, indices[b, n, j]] * weights[b, n, j], j = 0..2."""

    @staticmethod
    def forward(ctx, s_feats: Tensor, indices: Tensor, weights: Tensor, transposed: bool = False) -> Tensor:
        src = s_feats.transpose(1, 2) if transposed else s_feats  # (B, M, C) -> (B, C, M)
        src = src.contiguous()
        indices = indices.contiguous()
        weights = weights.contiguous()
        ctx.save_for_backward(indices, weights)
        ctx.num_supports = src.shape[-1]
        ctx.transposed = transposed
        q_feats = src.new_empty((src.shape[0], src.shape[1], indices.shape[1]))  # (B, C, N)
        ext.three_interpolate_forward(src, indices, weights, q_feats)
        return q_feats.transpose(1, 2).contiguous() if transposed else q_feats

    @staticmethod
    def backward(ctx, q_grads: Tensor):
        indices, weights = ctx.saved_tensors
        grads = q_grads.transpose(1, 2) if ctx.transposed else q_grads
        grads = grads.contiguous()
        s_grads = grads.new_empty((grads.shape[0], grads.shape[1], ctx.num_supports))  # (B, C, M)
        # (the reference calls ext.three_interpolate_backward here, :57; same sums, the lists built once per table)
        ext.scatter_apply(grads, scatter_lists(indices, ctx.num_supports, weights), s_grads)
        if ctx.transposed:
            s_grads = s_grads.transpose(1, 2).contiguous()
        return s_grads, None, None, None


three_interpolate = _ThreeInterpolateFunction.apply
