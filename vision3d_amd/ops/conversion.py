"""Pack <-> batch layout conversion -- mirrors ``vision3d/ops/conversion.py:7-83`` (same results), built from
one cumulative sum on the device instead of a Python loop over the clouds."""
from typing import Optional, Tuple

import torch
from torch import Tensor


def _get_indices_from_lengths(lengths: Tensor, num_items: int, total: Optional[int] = None) -> Tensor:
    """Flat indices, in a (B * num_items) batch layout, of the valid items given per-cloud ``lengths``.  ``total`` =
    sum(lengths) when the caller already knows it (the packed tensor's row count): then nothing is read back from the
    device; without it the sum is read once, because it is the shape of the result."""
    lengths = lengths.to(dtype=torch.long)
    starts = torch.cumsum(lengths, dim=0) - lengths  # first packed index of every cloud
    if total is None:
        total = int(lengths.sum().item())
    cloud = torch.repeat_interleave(torch.arange(lengths.shape[0], device=lengths.device), lengths, output_size=total)
    within = torch.arange(total, device=lengths.device) - starts[cloud]
    return cloud * num_items + within


def batch_to_pack_with_lengths(batch_tensor: Tensor, lengths: Tensor) -> Tuple[Tensor, Tensor]:
    """(B, N, C) + lengths (B,) -> packed (sum(lengths), C), lengths."""
    batch_size, num_items = batch_tensor.shape[0], batch_tensor.shape[1]
    flat = batch_tensor.view(batch_size * num_items, -1)
    return flat[_get_indices_from_lengths(lengths, num_items)], lengths


def batch_to_pack(batch_tensor: Tensor, masks: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """(B, N, C) or (B, N) with masks (B, N) -> packed (M, C) or (M,), lengths (B,)."""
    if masks is None:
        lengths = torch.full((batch_tensor.shape[0],), batch_tensor.shape[1], dtype=torch.long,
                             device=batch_tensor.device)
        return batch_tensor, lengths
    return batch_tensor[masks], masks.sum(dim=1)


def pack_to_batch(pack_tensor: Tensor, lengths: Tensor, max_length=None, fill_value=0.0) -> Tuple[Tensor, Tensor]:
    """packed (M, C) + lengths (B,) -> (B, max_length, C) filled with ``fill_value`` and masks (B, max_length)."""
    batch_size = lengths.shape[0]
    if max_length is None:
        max_length = int(lengths.max().item())
    tgt = _get_indices_from_lengths(lengths, max_length, total=pack_tensor.shape[0])
    channels = pack_tensor.shape[1]
    batch_tensor = pack_tensor.new_full((batch_size * max_length, channels), fill_value)
    batch_tensor[tgt] = pack_tensor
    masks = torch.zeros((batch_size * max_length,), dtype=torch.bool, device=pack_tensor.device)
    masks[tgt] = True
    return batch_tensor.view(batch_size, max_length, channels), masks.view(batch_size, max_length)
