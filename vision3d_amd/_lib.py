"""ctypes binding of ``libv3d_hip.so`` (C ABI declared in ``include/v3d_hip.h``).

The library is the product: there is NO CPU or PyTorch fallback.  If it is missing, or a call fails, the
error is raised to the caller (``RuntimeError``), never papered over.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libv3d_hip.so")

_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_f32 = ctypes.c_float
_c_ptr = ctypes.c_void_p
_c_size = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/v3d_hip.h one to one (tests/test_abi.py checks both against
# the header text).
_TAIL = [_c_int, _c_ptr]  # device, stream
_WS = [_c_ptr, _c_size]  # workspace, workspace_bytes
SIGNATURES = {
    "v3d_version": (_c_int, []),
    "v3d_last_error_string": (ctypes.c_char_p, []),
    "v3d_error_name": (ctypes.c_char_p, [_c_int]),
    "v3d_device_count": (_c_int, [ctypes.POINTER(_c_int)]),
    "v3d_fps_ref_block": (_c_int, [_c_int]),
    "v3d_fps_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_fps_resume_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_ball_query_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_f32, _c_int] + _TAIL),
    "v3d_ball_query_f64": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_f32, _c_int] + _TAIL),
    "v3d_ball_query_multi_f32": (
        _c_int, [_c_ptr, _c_ptr, ctypes.POINTER(_c_ptr), ctypes.POINTER(_c_f32), ctypes.POINTER(_c_int), _c_int, _c_int,
                 _c_int, _c_int] + _TAIL),
    "v3d_knn_points_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_knn_points_f64": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_knn_lex_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_three_nn_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_three_nn_f64": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_radius_count_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_f32] + _TAIL),
    "v3d_three_interp_fwd_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_three_interp_fwd_f64": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_scatter_workspace_bytes": (_c_size, [_c_int, _c_int, _c_i64]),
    "v3d_three_interp_bwd_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _WS + _TAIL),
    "v3d_three_interp_bwd_f64": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _WS + _TAIL),
    "v3d_three_interp_bwd_noalloc_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_gather_fwd": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_group_gather_fwd": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_gather_bwd_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _WS + _TAIL),
    "v3d_gather_bwd_f64": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int] + _WS + _TAIL),
    "v3d_group_gather_bwd_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int] + _WS + _TAIL),
    "v3d_group_gather_bwd_f64": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int] + _WS + _TAIL),
    "v3d_scatter_build": (_c_int, [_c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_i64] + _WS + _TAIL),
    "v3d_scatter_apply_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_i64] + _WS + _TAIL),
    "v3d_scatter_apply_f64": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_i64] + _WS + _TAIL),
    "v3d_pairwise_distance_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32] + _TAIL),
    "v3d_knn_feat_workspace_bytes": (_c_size, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "v3d_knn_feat_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_i64] + _WS + _TAIL),
    "v3d_topk_merge_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_i64, _c_int] + _TAIL),
    "v3d_grid_subsample_workspace_bytes": (_c_size, [_c_int, _c_i64, _c_int]),
    "v3d_grid_subsample_pack_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_i64, _c_f32] + _WS + _TAIL),
    "v3d_voxelize_pack_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_i64, _c_f32] + _WS
        + _TAIL),
    "v3d_radius_search_workspace_bytes": (_c_size, [_c_int, _c_i64]),
    "v3d_radius_search_pack_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_i64, _c_i64, _c_f32, _c_int] + _WS + _TAIL),
    "v3d_radius_grid_build_f32": (_c_int, [_c_ptr, _c_ptr, _c_int, _c_i64, _c_f32] + _WS + _TAIL),
    "v3d_radius_search_grid_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_i64, _c_i64, _c_f32, _c_f32, _c_int] + _WS + _TAIL),
    "v3d_knn_pack_workspace_bytes": (_c_size, [_c_int, _c_i64, _c_int]),
    "v3d_knn_pack_uses_cell_list": (_c_int, [_c_int, _c_i64, _c_i64, _c_int]),
    "v3d_knn_pack_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_i64, _c_i64, _c_int] + _WS + _TAIL),
    "v3d_radius_count_pack_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_i64, _c_i64, _c_f32] + _WS + _TAIL),
    "v3d_mutual_topk_select_workspace_bytes": (_c_size, [_c_int, _c_int, _c_int, _c_int]),
    "v3d_mutual_select_lists_workspace_bytes": (_c_size, [_c_int, _c_int, _c_int, _c_int]),
    "v3d_mutual_topk_select_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                 _c_ptr, _c_ptr, _c_i64] + _WS + _TAIL),
    "v3d_mutual_select_lists_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32, _c_ptr,
                 _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64] + _WS + _TAIL),
    "v3d_threshold_select_workspace_bytes": (_c_size, [_c_i64, _c_i64]),
    "v3d_global_topk_select_workspace_bytes": (_c_size, [_c_i64, _c_i64, _c_i64]),
    "v3d_threshold_select_f32": (
        _c_int, [_c_ptr, _c_i64, _c_i64, _c_f32, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64] + _WS + _TAIL),
    "v3d_global_topk_select_f32": (
        _c_int, [_c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64] + _WS + _TAIL),
    "v3d_index_select_rows": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64] + _TAIL),
    "v3d_local_maxpool_pack_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_int] + _TAIL),
    "v3d_knn_interp_weights_pack_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_int, _c_f32] + _TAIL),
    "v3d_rows_weighted_sum_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_global_avgpool_pack_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int] + _TAIL),
    "v3d_sa_group_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_edge_group_f32": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_int, _c_int] + _TAIL),
    "v3d_kpconv_gather_workspace_bytes": (_c_size, [_c_i64]),
    "v3d_kpconv_gather_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_f32]
        + _WS + _TAIL),
    "v3d_point_to_node_workspace_bytes": (_c_size, [_c_i64, _c_i64]),
    "v3d_point_to_node_partition_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr] + _WS + _TAIL),
    "v3d_point_to_node_knn": (_c_int, [_c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr] + _TAIL),
    "v3d_kpconv_gather_bwd_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_f32] + _TAIL),
    "v3d_rows_csr_bytes": (_c_size, [_c_i64, _c_i64]),
    "v3d_rows_csr_build": (_c_int, [_c_ptr, _c_i64, _c_i64] + _WS + _TAIL),
    "v3d_rows_scatter_f32": (
        _c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_i64, _c_int, _c_int, _c_int] + _TAIL),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 into ``vision3d_amd/libv3d_hip.so`` (in-tree)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    """The loaded library; raises RuntimeError when it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"vision3d_amd: {LIB_PATH} is missing -- build it with `make -C vision3d_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU / PyTorch fallback for these operators."
            )
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # e.g. no ROCm runtime on this host
            raise RuntimeError(f"vision3d_amd: cannot load {LIB_PATH}: {e}") from e
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError = ABI drift, let it surface
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    """Turn a non-zero C-ABI return code into the RuntimeError the reference's TORCH_CHECK would raise."""
    if rc != 0:
        handle = lib()
        name = handle.v3d_error_name(rc).decode()
        msg = handle.v3d_last_error_string().decode()
        raise RuntimeError(f"{what}: {name} ({rc}): {msg}")
