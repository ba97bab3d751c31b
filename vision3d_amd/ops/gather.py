"""``gather`` -- mirrors ``vision3d/ops/gather.py:10-70`` (GatherFunction + ``gather = GatherFunction.apply``)."""
from torch import Tensor
from torch.autograd import Function

from .. import ext
from ._scatter_lists import scatter_lists


class GatherFunction(Function):
    """outputs[b, c, j] = inputs[b, c, indices[b, j]]  (``transposed``: channel-last in and out)."""

    @staticmethod
    def forward(ctx, inputs: Tensor, indices: Tensor, transposed: bool = False) -> Tensor:
        src = inputs.transpose(1, 2) if transposed else inputs  # (B, N, C) -> (B, C, N)
        src = src.contiguous()
        indices = indices.contiguous()
        ctx.save_for_backward(indices)
        ctx.num_inputs = src.shape[-1]
        ctx.transposed = transposed
        # the kernel overwrites every element, so the reference's zero fill (gather.py:40) is not needed
        outputs = src.new_empty((src.shape[0], src.shape[1], indices.shape[1]))  # (B, C, M)
        ext.gather_forward(src, indices, outputs)
        return outputs.transpose(1, 2).contiguous() if transposed else outputs

    @staticmethod
    def backward(ctx, grad_outputs: Tensor):
        (indices,) = ctx.saved_tensors
        grads = grad_outputs.transpose(1, 2) if ctx.transposed else grad_outputs
        grads = grads.contiguous()
        grad_inputs = grads.new_empty((grads.shape[0], grads.shape[1], ctx.num_inputs))  # (B, C, N)
        # (the reference calls ext.gather_backward here, :61; same sums, the lists built once per index tensor)
        ext.scatter_apply(grads, scatter_lists(indices, ctx.num_inputs), grad_inputs)
        if ctx.transposed:
            grad_inputs = grad_inputs.transpose(1, 2).contiguous()
        return grad_inputs, None, None


gather = GatherFunction.apply
