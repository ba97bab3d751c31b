"""``group_gather`` -- mirrors ``vision3d/ops/group_gather.py:10-72``."""
from torch import Tensor
from torch.autograd import Function

from .. import ext
from ._scatter_lists import scatter_lists


class GroupGatherFunction(Function):
    """outputs[b, c, j, k] = inputs[b, c, indices[b, j, k]]  (``transposed``: (B,N,C) -> (B,M,K,C))."""

    @staticmethod
    def forward(ctx, inputs: Tensor, indices: Tensor, transposed: bool = False):
        src = inputs.transpose(1, 2) if transposed else inputs
        src = src.contiguous()
        indices = indices.contiguous()
        ctx.save_for_backward(indices)
        ctx.num_inputs = src.shape[-1]
        ctx.transposed = transposed
        outputs = src.new_empty((src.shape[0], src.shape[1], indices.shape[1], indices.shape[2]))  # (B, C, M, K)
        ext.group_gather_forward(src, indices, outputs)
        return outputs.permute(0, 2, 3, 1).contiguous() if transposed else outputs

    @staticmethod
    def backward(ctx, grad_outputs: Tensor):
        (indices,) = ctx.saved_tensors
        grads = grad_outputs.permute(0, 3, 1, 2) if ctx.transposed else grad_outputs  # -> (B, C, M, K)
        grads = grads.contiguous()
        grad_inputs = grads.new_empty((grads.shape[0], grads.shape[1], ctx.num_inputs))  # (B, C, N)
        # (the reference calls ext.group_gather_backward here, :63; same sums, the lists built once per index tensor)
        ext.scatter_apply(grads, scatter_lists(indices, ctx.num_inputs), grad_inputs)
        if ctx.transposed:
            grad_inputs = grad_inputs.transpose(1, 2).contiguous()
        return grad_inputs, None, None


group_gather = GroupGatherFunction.apply
