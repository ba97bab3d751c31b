#!/usr/bin/env python3
"""Per-operator CPU baselines beside the GPU numbers ("CPU baseline beside it"), on this host's cores.

Three contenders per operator, identical seeded inputs (BASELINE config C2 unless noted):
  (i)   the reference's CPU path restated against scipy directly: cKDTree.query / query_ball_point
        (vision3d/array_ops/knn.py:11-12, array_ops/ball_query.py:8-9), workers=1 and workers=-1;
  (ii)  pure-torch CPU formulations: pairwise_distance expansion + torch.topk, gather via torch.gather,
        scatter via index_add_ (torch threads = all cores and 1);
  (iii) the bit-exact CPU oracle (oracle/v3d_oracle.c), single thread and OpenMP over all cores.
Timing: time.perf_counter, 1 warm-up, median of 3 (the slowest contenders take seconds per call).  A tuning / reporting
aid like tools/opbench.py: test infrastructure may import oracle/, the product path never does.

usage: python tools/cpu_baselines.py > profiles/r01_cpu_baselines.txt
"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

B, N, M = 32, 4096, 1024


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def line(name, contender, seconds, work, unit):
    print(f"{name:34s} {contender:38s} {seconds * 1e3:10.1f} ms  {work / seconds / 1e6:10.3f} {unit}", flush=True)


def main():
    from scipy.spatial import cKDTree

    cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))  # the GPU box grants a 16-core share
    omp = min(oracle.max_threads(), cores)
    print(f"# host cores used = {cores}; OpenMP threads used by the oracle = {omp}; torch threads = "
          f"{torch.get_num_threads()}")
    g = torch.Generator().manual_seed(0)
    pts = torch.rand((B, N, 3), generator=g).numpy()
    feats = torch.randn((B, 64, N), generator=g).numpy()
    sfeats = torch.randn((B, 128, M), generator=g).numpy()
    grad = torch.randn((B, 128, N), generator=g).numpy()
    oracle.set_threads(omp)
    idx = oracle.furthest_point_sample(pts, M)
    q = np.ascontiguousarray(np.take_along_axis(pts, idx[..., None], axis=1))

    def both_thread_counts(name, fn, work, unit):
        for t in (1, omp):
            oracle.set_threads(t)
            line(name, f"(iii) oracle, {t} thread(s)", timed(fn), work, unit)
        oracle.set_threads(omp)

    # ---- furthest point sampling (reference CPU twin: csrc/cpu/node_sampling, a different traversal; oracle only)
    both_thread_counts("fps B32 N4096 M1024", lambda: oracle.furthest_point_sample(pts, M), B * N, "Mpts/s")
    # ---- ball query r=0.2 k=32
    both_thread_counts("ball_query r0.2 k32", lambda: oracle.ball_query(q, pts, 32, 0.2), B * M, "Mq/s")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        trees = [cKDTree(pts[b]) for b in range(B)]
        for w in (1, -1):
            line("ball_query r0.2 (all in-radius)", f"(i) scipy query_ball_point workers={w}",
                 timed(lambda: [trees[b].query_ball_point(q[b], 0.2, workers=w) for b in range(B)]), B * M, "Mq/s")
        # ---- kNN on points k=32 and three_nn
        both_thread_counts("knn_points k32", lambda: oracle.knn_points(q, pts, 32), B * M, "Mq/s")
        for w in (1, -1):
            line("knn_points k32", f"(i) scipy cKDTree.query workers={w}",
                 timed(lambda: [trees[b].query(q[b], k=32, workers=w) for b in range(B)]), B * M, "Mq/s")
        both_thread_counts("three_nn Q4096 S1024", lambda: oracle.three_nn(pts, q), B * N, "Mq/s")
        qtrees = [cKDTree(q[b]) for b in range(B)]
        for w in (1, -1):
            line("three_nn Q4096 S1024", f"(i) scipy cKDTree.query k=3 workers={w}",
                 timed(lambda: [qtrees[b].query(pts[b], k=3, workers=w) for b in range(B)]), B * N, "Mq/s")
    # ---- group gather k=32 C=64 forward / backward
    bi = oracle.ball_query(q, pts, 32, 0.2)
    tb = torch.from_numpy(bi)
    tf = torch.from_numpy(feats)
    gbytes = (B * M * 32 * 8 + 4 * B * 64 * (N + M * 32)) / 1e3
    both_thread_counts("group_gather fwd k32 C64", lambda: oracle.group_gather_forward(feats, bi), gbytes, "GB/s")
    up = np.random.default_rng(0).standard_normal((B, 64, M, 32)).astype(np.float32)
    both_thread_counts("group_gather bwd k32 C64", lambda: oracle.group_gather_backward(up, bi, N), gbytes, "GB/s")
    for t in (1, cores):
        torch.set_num_threads(t)
        flat = tb.view(B, 1, M * 32).expand(-1, 64, -1)
        line("group_gather fwd k32 C64", f"(ii) torch.gather, {t} thread(s)", timed(lambda: torch.gather(tf, 2, flat)),
             gbytes, "GB/s")
        tup = torch.from_numpy(up).view(B, 64, M * 32)
        line("group_gather bwd k32 C64", f"(ii) torch scatter_add_, {t} thread(s)",
             timed(lambda: torch.zeros((B, 64, N)).scatter_add_(2, flat, tup)), gbytes, "GB/s")
    # ---- three_interpolate fwd / bwd C=128
    d3, i3 = oracle.three_nn(pts, q)
    w3 = 1.0 / (d3 + 1e-5)
    w3 = (w3 / w3.sum(-1, keepdims=True)).astype(np.float32)
    ibytes = 88604672 / 1e3
    both_thread_counts("three_interpolate fwd C128", lambda: oracle.three_interpolate_forward(sfeats, i3, w3), ibytes,
                       "GB/s")
    both_thread_counts("three_interpolate bwd C128", lambda: oracle.three_interpolate_backward(grad, i3, w3, M), ibytes,
                       "GB/s")
    # ---- pairwise distance + top-k (config C3: 32 x 1024 x 1024 x 64, k = 20)
    x = torch.randn((32, 1024, 64), generator=g)
    flops = 2.0 * 32 * 1024 * 1024 * 64 / 1e6
    for t in (1, cores):
        torch.set_num_threads(t)

        def torch_knn():
            d = (x * x).sum(-1, keepdim=True) - 2 * x @ x.transpose(1, 2) + (x * x).sum(-1).unsqueeze(1)
            return d.clamp_(min=0).topk(20, dim=2, largest=False)
        line("feature kNN C3 k20", f"(ii) torch matmul expansion + topk, {t} thread(s)", timed(torch_knn), flops,
             "TFLOP/s")
    xn = x.numpy()
    both_thread_counts("feature kNN C3 k20", lambda: oracle.knn_feat(xn, xn, 20), flops, "TFLOP/s")
    oracle.set_threads(1)


if __name__ == "__main__":
    main()
