"""Per-index-tensor cache of the backward lists (``ext.scatter_build``), kept on the tensor object itself exactly as
``ops/pack_gather.py`` keeps a table's CSR: the xyz gather and the feature gather of one set-abstraction layer share one
``indices`` (reference: ``layers/pointnet2.py:44-46``), and so do the scales of a feature pyramid that reuse a table, so
one inversion serves every gradient that flows through it.  An in-place edit of the tensor (``_version``) or another
allocation behind it (``data_ptr``) invalidates the entry."""
from .. import ext


def scatter_lists(indices, num_sources, weights=None):
    key = (indices._version, indices.data_ptr(), int(num_sources), tuple(indices.shape),
           None if weights is None else (weights._version, weights.data_ptr(), weights.dtype))
    cached = getattr(indices, "_v3d_scatter_lists", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    lists = ext.scatter_build(indices, num_sources, weights)
    try:
        indices._v3d_scatter_lists = (key, lists)
    except (AttributeError, RuntimeError):  # a tensor type that takes no attributes: build per call
        pass
    return lists
