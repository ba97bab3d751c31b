"""The entry point that allocates nothing (v3d_three_interp_bwd_noalloc_f32, include/v3d_hip.h): called through the C ABI,
its output must equal -- bit for bit, the bar of the other backward tests -- the output of v3d_three_interp_bwd_f32 on the
same inputs, on two calls in a row that share the built-in scratch array, the second on a smaller input; shapes beyond
the array are refused.  Collected last (file name) so that a failure here cannot hide another test behind `pytest -x`."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from vision3d_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libv3d_hip.so must be built in-tree before the GPU run"
    return _lib.lib()


def _case(seed, B, C, S, Q):
    g = torch.Generator().manual_seed(seed)
    grad = torch.randn((B, C, Q), generator=g).to(DEV)
    idx = torch.randint(0, S, (B, Q, 3), generator=g).to(DEV)
    idx[:, ::7, 1] = idx[:, ::7, 0]  # repeated sources inside a target
    idx[0, :, 2] = 3 % S  # one source named by every target of a cloud
    w = torch.rand((B, Q, 3), generator=g).to(DEV)
    return grad, idx, w


def _both(lib, grad, idx, w, S):
    B, C, Q = grad.shape
    dev = torch.cuda.current_device()
    stream = torch.cuda.current_stream().cuda_stream
    want = torch.full((B, C, S), float("nan"), device=DEV)
    ws = torch.empty((lib.v3d_scatter_workspace_bytes(B, S, Q * 3),), dtype=torch.uint8, device=DEV)
    rc = lib.v3d_three_interp_bwd_f32(grad.data_ptr(), idx.data_ptr(), w.data_ptr(), want.data_ptr(), B, C, S, Q,
                                      ws.data_ptr(), ws.numel(), dev, stream)
    assert rc == 0, lib.v3d_last_error_string()
    got = torch.full((B, C, S), float("nan"), device=DEV)
    rc = lib.v3d_three_interp_bwd_noalloc_f32(grad.data_ptr(), idx.data_ptr(), w.data_ptr(), got.data_ptr(), B, C, S, Q,
                                              dev, stream)
    assert rc == 0, lib.v3d_last_error_string()
    torch.cuda.synchronize()
    return got, want


def test_no_alloc_backward_equals_workspace_backward_twice(lib):
    # first call: BASELINE C2's feature-propagation shape (B=32, C=128, 4096 targets <- 1024 sources)
    grad, idx, w = _case(1, 32, 128, 1024, 4096)
    got, want = _both(lib, grad, idx, w, 1024)
    assert torch.equal(got, want)
    # second call right after it, smaller input, same scratch array (which still holds the first call's lists)
    grad, idx, w = _case(2, 3, 20, 100, 777)
    got, want = _both(lib, grad, idx, w, 100)
    assert torch.equal(got, want)
    # and a cloud with more sources than the in-LDS path takes (the generic CSR layout of the same scratch)
    grad, idx, w = _case(3, 2, 8, 5000, 3000)
    got, want = _both(lib, grad, idx, w, 5000)
    assert torch.equal(got, want)


def test_no_alloc_backward_refuses_shapes_beyond_its_scratch(lib):
    B, C, S, Q = 64, 4, 4096, 200000  # needs far more than the 16 MiB array
    assert lib.v3d_scatter_workspace_bytes(B, S, Q * 3) > (16 << 20)
    grad = torch.zeros((1,), device=DEV)  # never dereferenced: the size check comes first
    rc = lib.v3d_three_interp_bwd_noalloc_f32(grad.data_ptr(), grad.data_ptr(), grad.data_ptr(), grad.data_ptr(), B, C, S,
                                              Q, torch.cuda.current_device(), None)
    assert rc != 0 and lib.v3d_error_name(rc).decode() == "V3D_ERR_SIZE_LIMIT"
    assert b"scratch" in lib.v3d_last_error_string()


def test_no_alloc_backward_really_allocates_nothing(lib):
    """The property itself, two ways.  (1) HIP's default memory pool keeps a high-water mark of stream-ordered
    allocations (hipMemPoolAttrUsedMemHigh): reset to zero, it must still read zero after the call, while the workspace
    entry point called with a NULL workspace -- which takes hipMallocAsync -- moves it (so the probe can see what it
    looks for).  (2) The call is captured into a HIP graph on a side stream under torch's capture mode "global" (any
    hipMalloc on the capturing thread is then an error) after one eager warm-up call (first use per device takes
    hipGetSymbolAddress / hipFuncSetAttribute, to be done outside capture, as include/v3d_hip.h says), replayed on new
    inputs written into the same buffers, and compared with the eager result."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    USED_MEM_HIGH = 0x8  # hipMemPoolAttrUsedMemHigh, hip_runtime_api.h
    dev = torch.cuda.current_device()
    pool = ctypes.c_void_p()
    assert hip.hipDeviceGetDefaultMemPool(ctypes.byref(pool), dev) == 0

    def reset_mark():
        zero = ctypes.c_uint64(0)
        assert hip.hipMemPoolSetAttribute(pool, USED_MEM_HIGH, ctypes.byref(zero)) == 0

    def mark():
        v = ctypes.c_uint64(0)
        assert hip.hipMemPoolGetAttribute(pool, USED_MEM_HIGH, ctypes.byref(v)) == 0
        return v.value

    B, C, S, Q = 4, 32, 512, 2048
    grad, idx, w = _case(5, B, C, S, Q)
    out = torch.empty((B, C, S), device=DEV)
    stream = torch.cuda.current_stream().cuda_stream

    def call():
        return lib.v3d_three_interp_bwd_noalloc_f32(grad.data_ptr(), idx.data_ptr(), w.data_ptr(), out.data_ptr(), B, C, S,
                                                    Q, dev, torch.cuda.current_stream().cuda_stream)

    assert call() == 0, lib.v3d_last_error_string()  # warm-up: first call on the device
    torch.cuda.synchronize()
    reset_mark()
    assert call() == 0
    torch.cuda.synchronize()
    assert mark() == 0, "the no-alloc entry point made a stream-ordered allocation"
    ref = out.clone()
    # the probe does see an allocation when there is one: NULL workspace on the ordinary entry point
    other = torch.empty_like(out)
    torch.cuda.synchronize()
    reset_mark()
    rc = lib.v3d_three_interp_bwd_f32(grad.data_ptr(), idx.data_ptr(), w.data_ptr(), other.data_ptr(), B, C, S, Q, None, 0,
                                      dev, stream)
    assert rc == 0, lib.v3d_last_error_string()
    torch.cuda.synchronize()
    assert mark() > 0 and torch.equal(other, ref)

    # (2) under graph capture
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    out.fill_(float("nan"))
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="global"):
        rc = call()
    assert rc == 0, lib.v3d_last_error_string()
    grad2, idx2, w2 = _case(6, B, C, S, Q)
    grad.copy_(grad2), idx.copy_(idx2), w.copy_(w2)
    graph.replay()
    torch.cuda.synchronize()
    replayed = out.clone()
    assert call() == 0
    torch.cuda.synchronize()
    assert torch.equal(replayed, out) and not torch.equal(replayed, ref)
