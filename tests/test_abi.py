"""The C-ABI library loads without a GPU and exports exactly what include/v3d_hip.h declares."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "v3d_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # drop comments
    decls = {}
    for m in re.finditer(r"\b([A-Za-z_][\w \*]*?)\b(v3d_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else args.count(",") + 1
        decls[m.group(2)] = n
    return decls


@pytest.fixture(scope="module")
def lib_module():
    from vision3d_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def test_header_declares_the_expected_entry_points():
    decls = _declared()
    for name in ("v3d_fps_f32", "v3d_ball_query_f32", "v3d_ball_query_f64", "v3d_knn_points_f32",
                 "v3d_knn_points_f64", "v3d_three_nn_f32", "v3d_three_nn_f64", "v3d_three_interp_fwd_f32",
                 "v3d_three_interp_bwd_f32", "v3d_gather_fwd", "v3d_group_gather_fwd", "v3d_gather_bwd_f32",
                 "v3d_group_gather_bwd_f32", "v3d_pairwise_distance_f32", "v3d_knn_feat_f32", "v3d_topk_merge_f32"):
        assert name in decls
    assert len(decls) >= 28


def test_library_exports_every_declared_symbol(lib_module):
    handle = ctypes.CDLL(lib_module.LIB_PATH)
    for name in _declared():
        assert hasattr(handle, name), f"{name} declared in v3d_hip.h but not exported"


def test_binding_table_matches_header(lib_module):
    decls = _declared()
    assert set(lib_module.SIGNATURES) == set(decls)
    for name, (_, argtypes) in lib_module.SIGNATURES.items():
        assert len(argtypes) == decls[name], name


def test_no_compute_entry_points_without_gpu(lib_module):
    """Loads and answers the pure-host queries; compute calls are not attempted here (no GPU)."""
    lib = lib_module.lib()
    assert lib.v3d_version() >= 100
    assert lib.v3d_fps_ref_block(100) == 64 and lib.v3d_fps_ref_block(4096) == 512
    assert lib.v3d_error_name(-2) == b"V3D_ERR_K_TOO_LARGE"
    assert lib.v3d_scatter_workspace_bytes(2, 10, 100) >= 2 * (11 + 10 + 200) * 4
    assert lib.v3d_knn_feat_workspace_bytes(1, 128, 128, 64, 4) > 0
    # validation happens before any HIP call: k too large is rejected on any host
    rc = lib.v3d_knn_points_f32(None, None, None, None, 1, 1, 1, 101, 0, None)
    assert rc == -2 and b"exceeds" in lib.v3d_last_error_string()
    assert lib.v3d_ball_query_f32(None, None, None, -1, 1, 1, 0.1, 1, 0, None) == -1


def test_fps_ref_block_agrees_with_oracle(lib_module, oracle_mod):
    lib = lib_module.lib()
    for n in list(range(1, 70)) + [100, 127, 128, 255, 256, 257, 511, 512, 513, 1000, 1024, 4096, 30000, 100000]:
        assert lib.v3d_fps_ref_block(n) == oracle_mod.fps_block(n), n
