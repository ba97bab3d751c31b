"""``pairwise_distance`` -- mirrors ``vision3d/ops/pairwise_distance.py:5-58``.

The reference computes torch.matmul plus three or four elementwise passes over the (N, M) matrix; here the whole
expression is one gfx950 kernel (fp32-input MFMA for x.y^T, norms / clamp / sqrt fused in the epilogue), or one
VALU kernel of the exact difference form for ``strict=True``.  Differentiable (analytic backward of the same
expression).  Only CUDA float32 tensors are supported -- there is deliberately no CPU / other-dtype fallback.
"""
import torch
from torch import Tensor
from torch.autograd import Function

from .. import ext


class _PairwiseDistanceFunction(Function):
    @staticmethod
    def forward(ctx, x: Tensor, y: Tensor, normalized: bool, squared: bool, strict: bool, eps: float) -> Tensor:
        # x (B, N, C), y (B, M, C) contiguous fp32
        out = x.new_empty((x.shape[0], x.shape[1], y.shape[1]))
        ext.pairwise_distance(x, y, out, normalized=normalized, squared=squared, eps=eps, strict=strict)
        ctx.save_for_backward(x, y, out)
        ctx.flags = (normalized, squared, strict, eps)
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        x, y, out = ctx.saved_tensors
        normalized, squared, strict, eps = ctx.flags
        g = grad_out.contiguous()
        if strict:
            # out = ||x - y|| (or its square): d out^2 / dx = 2 (x - y)
            if not squared:
                g = torch.where(out > 0, g / (2.0 * out), torch.zeros_like(g))
        else:
            # clamp(min=0) (pairwise_distance.py:53) blocks the gradient where it clipped.  A clipped cell holds exactly
            # 0, resp. exactly sqrt(0 + eps) in fp32 (correctly rounded in the kernel and in torch alike): the mask is
            # a comparison with that one value, not a rounding-dependent out*out > eps.  Cells whose unclamped value
            # is exactly 0 (or within eps rounding of it) are masked too, where torch's clamp lets the gradient pass:
            # a set of measure zero, and for coinciding points the gradient 2 g (x - y) vanishes either way.
            if not squared:
                floor = torch.sqrt(torch.tensor(eps, dtype=out.dtype, device=out.device))
                g = g / (2.0 * out)  # out = sqrt(clamp(d) + eps) > 0
                live = out > floor
            else:
                live = out > 0
            g = torch.where(live, g, torch.zeros_like(g))
        grad_x = grad_y = None
        if normalized and not strict:  # d = 2 - 2 x.y
            if ctx.needs_input_grad[0]:
                grad_x = -2.0 * torch.matmul(g, y)
            if ctx.needs_input_grad[1]:
                grad_y = -2.0 * torch.matmul(g.transpose(-1, -2), x)
        else:  # d = |x|^2 - 2 x.y + |y|^2
            if ctx.needs_input_grad[0]:
                grad_x = 2.0 * (g.sum(dim=-1, keepdim=True) * x - torch.matmul(g, y))
            if ctx.needs_input_grad[1]:
                grad_y = 2.0 * (g.sum(dim=-2).unsqueeze(-1) * y - torch.matmul(g.transpose(-1, -2), x))
        return grad_x, grad_y, None, None, None, None


def pairwise_distance(
    x: Tensor,
    y: Tensor,
    normalized: bool = False,
    transposed: bool = False,
    squared: bool = True,
    strict: bool = False,
    eps: float = 1e-8,
) -> Tensor:
    """Pairwise (squared) distance of two (batched) point / feature sets.

    x (*, N, C), y (*, M, C) -- or (*, C, N), (*, C, M) if ``transposed`` -- -> (*, N, M).
    ``normalized``: unit-norm inputs, d = 2 - 2 x.y.  ``squared=False``: sqrt(d + eps).  ``strict``: exact
    difference form (no cancellation), like the reference's broadcast + norm branch.
    """
    if not (x.is_cuda and y.is_cuda):
        raise RuntimeError("pairwise_distance: x and y must be CUDA tensors (no CPU fallback in vision3d_amd)")
    if x.dtype != torch.float32 or y.dtype != torch.float32:
        raise RuntimeError("pairwise_distance: x and y must be float tensors")
    if transposed:
        x = x.transpose(-1, -2)
        y = y.transpose(-1, -2)
    lead = x.shape[:-2]
    if y.shape[:-2] != lead:
        lead = torch.broadcast_shapes(x.shape[:-2], y.shape[:-2])
        x = x.expand(lead + x.shape[-2:])
        y = y.expand(lead + y.shape[-2:])
    n, c = x.shape[-2:]
    m = y.shape[-2]
    xb = x.reshape((-1, n, c)).contiguous()
    yb = y.reshape((-1, m, c)).contiguous()
    out = _PairwiseDistanceFunction.apply(xb, yb, bool(normalized), bool(squared), bool(strict), float(eps))
    return out.reshape(tuple(lead) + (n, m))
