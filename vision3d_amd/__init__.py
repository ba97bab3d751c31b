"""vision3d_amd -- MI355X (gfx950) implementation of the vision3d point-cloud operator hot path.

Layout:
  csrc/        hand-written HIP kernels + the C ABI (``include/v3d_hip.h``) -> ``libv3d_hip.so``
  _lib.py      ctypes binding of that library (fails loudly when it is missing: no fallback exists)
  ext.py       drop-in for the CUDA entry points of the reference's ``vision3d.ext`` module
  ops/         host-side mirror of the reference's ``vision3d.ops`` wrappers for this path
  sharding.py  batch sharding and reference-set sharding (RCCL all-gather merge) across one node

``install()`` registers ``ext`` and ``ops`` under the reference's module names so unchanged model code
(``from vision3d.ops import ball_query``, ``load_ext("vision3d.ext", [...])``) runs on this implementation.
"""
import sys
import types

__version__ = "0.1.0"


def install(force: bool = False) -> None:
    """Expose this package as ``vision3d.ext`` / ``vision3d.ops``.

    If a real ``vision3d`` package is importable, only its ``ext`` submodule and the hot-path names of its
    ``ops`` are overridden; otherwise a minimal ``vision3d`` namespace is created.
    """
    from . import ext, ops  # noqa: WPS433  (imports the ctypes binding lazily, not the GPU)

    root = sys.modules.get("vision3d")
    if root is None:
        try:
            import vision3d as root  # type: ignore  # a real checkout on sys.path
        except Exception:  # ModuleNotFoundError, or its own import of vision3d.ext failing
            root = types.ModuleType("vision3d")
            root.__path__ = []  # mark as package
            sys.modules["vision3d"] = root
    if force or "vision3d.ext" not in sys.modules:
        sys.modules["vision3d.ext"] = ext
        root.ext = ext
    existing_ops = sys.modules.get("vision3d.ops")
    if existing_ops is None or force:
        sys.modules["vision3d.ops"] = ops
        root.ops = ops
    else:  # patch the hot-path names into the already-imported reference package
        for name in ops.__all__:
            setattr(existing_ops, name, getattr(ops, name))
