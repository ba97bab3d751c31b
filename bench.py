#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json config[1] on N GPUs of one node, plus one short line per other config.

usage: python bench.py --gpus N --steps K --warmup W [--workload c2|c5_refset|c4_pyramid]
  Under a launcher (RANK / LOCAL_RANK / WORLD_SIZE set, e.g. torch.distributed.run) this process IS one rank.  Without
  one and with --gpus N > 1 this process is only a parent: it starts N rank processes itself (before anything touches
  the GPU), relays rank 0's JSON line and exits with the worst child code.

Workload "pointnet2_msg_fp_s3dis" (default), one STEP = one pass over one batch of B=32 synthetic
S3DIS-sized blocks of N=4096 points, inputs already resident in HBM, outputs pre-allocated:
    FPS 4096 -> 1024 | gather centroids | for (r,k) in (0.1,16),(0.2,32),(0.4,128): ball_query,
    group_gather(xyz, 3 ch), group_gather(features, 64 ch) | three_nn 4096 <- 1024 | interpolation weights
    (torch elementwise, as the reference's FeaturePropagate does) | three_interpolate fwd + bwd (128 ch).
Every operator call goes through vision3d_amd.ext -> the C ABI -> hand-written gfx950 kernels.

Metric (BASELINE.json): neighbor queries/sec = (3*B*M ball queries + B*Q three_nn queries) / step time, whole job
over all ranks; FPS points/sec and per-operator timings ride along in the same JSON line.  Multi-GPU: independent
clouds shard by batch (one process per GPU, no data-path collective), so scaling is "weak": every rank runs its
own B=32 batch; the only collectives around the timed region are the barrier and a MAX all-reduce of the elapsed time.

`roofline` describes the dominant HBM kernel (group_gather forward, k=128, C=64): algorithmic bytes per launch / its
average duration from a HIP event pair around that launch on EVERY timed step (the other operators carry events on
every EVENT_EVERY-th step only).  `cpu_baseline` = the CPU oracle (port of the reference kernels' semantics, OpenMP) on
the same workload; `cpu_baseline_reference` = the reference's own CPU path for the query operators (scipy cKDTree as
vision3d/array_ops/{ball_query,knn}.py call it) on the same inputs.

`per_config` (after the timed region, excluded from ms_per_step) carries the other BASELINE configs: C1 step, C3 feature
kNN, C4 radius search + grid subsampling, C5 one-GPU shard, and -- when N > 1 -- C5 with the reference set sharded over
the ranks: partial top-k per rank, ONE all-gather over RCCL, merge, compared on the device with the single-GPU result.
`--workload c5_refset` makes that sharded search the timed step (scaling "strong": 100k queries in total).  When N > 1
`per_config` also carries C4 under pack-mode batch sharding (`c4_kpconv_pyramid_sharded`: 16 clouds of 30,000 points per
rank split by `lengths`, the 4-stage grid + radius pyramid per rank, no collective, checked against the single-process
tables); `--workload c4_pyramid` makes that the timed step (scaling "weak").
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable by a float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (no TF32 on gfx950)

B, N, M, C_FEAT, C_INTERP = 32, 4096, 1024, 64, 128
SCALES = ((0.1, 16), (0.2, 32), (0.4, 128))
QUERIES_PER_STEP = len(SCALES) * B * M + B * N  # ball queries + three_nn queries
EVENT_EVERY = 5  # timed steps whose launches ALL carry HIP events (per-operator durations)
DOMINANT = "group_gather_feat_k128"  # carries an event pair on every timed step (roofline)
C5_ROWS, C5_DIM = 100000, 256


# ----------------------------------------------------------------------------------------------------------------------
# Parent mode: start one process per GPU.  Nothing here may initialise the GPU (no torch.cuda call that creates a
# context; the children are fresh processes, never an exec of a process that touched the card).
# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    world = args.gpus
    port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if rank == 0 else subprocess.DEVNULL  # rank 0 prints the JSON line; stderr of every rank is kept
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    worst = 0
    for p in procs:
        rc = p.wait()
        worst = worst or rc
    return worst


# ----------------------------------------------------------------------------------------------------------------------
def make_inputs(seed, device):
    import torch

    g = torch.Generator().manual_seed(seed)
    pts = torch.rand((B, N, 3), generator=g)  # ~U[0,1)^3: a 1 m block
    feats = torch.randn((B, C_FEAT, N), generator=g)
    sfeats = torch.randn((B, C_INTERP, M), generator=g)
    grad = torch.randn((B, C_INTERP, N), generator=g)
    return {k: v.to(device) for k, v in dict(pts=pts, feats=feats, sfeats=sfeats, grad=grad).items()}


class Step:
    """Pre-allocated buffers + the operator sequence; `timers` collects (name, start_event, end_event)."""

    def __init__(self, inp, device, fused_msg=False):
        import torch

        from vision3d_amd import ext

        self.torch = torch
        self.fused_msg = fused_msg
        self.ext = ext
        self.inp = inp
        dev = device
        self.pts = inp["pts"]
        self.pts_cf = self.pts.transpose(1, 2).contiguous()  # (B,3,N) channel-first copy the models already hold
        self.fps_dist = torch.empty((B, N), device=dev)
        self.fps_idx = torch.zeros((B, M), dtype=torch.long, device=dev)
        self.q_cf = torch.empty((B, 3, M), device=dev)
        self.ball_idx = [torch.zeros((B, M, k), dtype=torch.long, device=dev) for _, k in SCALES]
        self.g_xyz = [torch.empty((B, 3, M, k), device=dev) for _, k in SCALES]
        self.g_feat = [torch.empty((B, C_FEAT, M, k), device=dev) for _, k in SCALES]
        self.tnn_d = torch.empty((B, N, 3), device=dev)
        self.tnn_i = torch.zeros((B, N, 3), dtype=torch.long, device=dev)
        self.interp = torch.empty((B, C_INTERP, N), device=dev)
        self.sgrad = torch.empty((B, C_INTERP, M), device=dev)
        self.timers = None  # list: every launch carries events
        self.dom_timers = None  # list: only the dominant launch carries events

    def _timed(self, name, fn):
        sink = self.timers
        if sink is None and name == DOMINANT:
            sink = self.dom_timers
        if sink is None:
            fn()
            return
        torch = self.torch
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        sink.append((name, s, e))

    def run(self):
        ext, inp = self.ext, self.inp
        self.fps_dist.fill_(1e10)
        self._timed("fps", lambda: ext.furthest_point_sample(self.pts, self.fps_dist, self.fps_idx, M))
        self._timed("gather_centroids", lambda: ext.gather_forward(self.pts_cf, self.fps_idx, self.q_cf))
        q = self.q_cf.transpose(1, 2).contiguous()  # (B,M,3)
        if self.fused_msg:  # one pass over the pairs for the three MSG scales (not the reference's call pattern)
            for t in self.ball_idx:
                t.zero_()
            self._timed("ball_query_msg3", lambda: ext.ball_query_multi(q, self.pts, self.ball_idx,
                                                                        [r for r, _ in SCALES], [k for _, k in SCALES]))
        for i, (r, k) in enumerate(SCALES):
            if not self.fused_msg:
                self.ball_idx[i].zero_()
                self._timed(f"ball_query_k{k}", lambda: ext.ball_query(q, self.pts, self.ball_idx[i], r, k))
            self._timed(f"group_gather_xyz_k{k}",
                        lambda: ext.group_gather_forward(self.pts_cf, self.ball_idx[i], self.g_xyz[i]))
            self._timed(f"group_gather_feat_k{k}",
                        lambda: ext.group_gather_forward(inp["feats"], self.ball_idx[i], self.g_feat[i]))
        self._timed("three_nn", lambda: ext.three_nn(self.pts, q, self.tnn_d, self.tnn_i))
        w = 1.0 / (self.tnn_d + 1e-5)  # layers/feature_propagate.py:43-44
        w = w / w.sum(dim=2, keepdim=True)
        self._timed("three_interp_fwd",
                    lambda: ext.three_interpolate_forward(inp["sfeats"], self.tnn_i, w, self.interp))
        self._timed("three_interp_bwd",
                    lambda: ext.three_interpolate_backward(inp["grad"], self.tnn_i, w, self.sgrad, M))


class PipelinedStep:
    """The same operator sequence on the same inputs for ONE batch, the sampling cut into `pieces` launches
    (ext.furthest_point_sample_rounds): the ball queries and gathers of the first centroids run on a second
    stream while the sampling -- a serial chain on B of the 256 CUs -- goes on.  Outputs per piece; three_nn and the
    interpolation need all centroids and follow the last piece.  Few, larger launches (one ball_query_multi per piece,
    one zero fill per scale): issued from Python a launch costs ~10 us of host time, and a HIP graph of the two-stream
    schedule replays slower than the plain launches (tools/pipe_probe.py).  Not the reference's call pattern (one launch
    samples all M): reported beside the headline as `ms_per_step_single_batch_pipelined`, never as `value`."""

    def __init__(self, inp, device, cuts=(0, M // 2, M)):
        import torch

        from vision3d_amd import ext

        # the consumers of a piece run under the sampling of the pieces after it; what follows the last piece is exposed.
        # Two halves measured best (tools/pipe_probe.py: 0.85 ms against 0.92 for 512 + 256 + 256 and 0.91 for four
        # quarters): every piece pays ~0.1 ms in launches that are too small to fill the chip
        assert cuts[0] == 0 and cuts[-1] == M and all(a < b and a % 64 == 0 for a, b in zip(cuts[:-1], cuts[1:]))
        self.torch, self.ext, self.inp, self.cuts = torch, ext, inp, tuple(cuts)
        self.pieces = len(cuts) - 1
        dev = device
        self.pts = inp["pts"]
        self.pts_cf = self.pts.transpose(1, 2).contiguous()
        self.fps_dist = torch.empty((B, N), device=dev)
        self.fps_idx = torch.zeros((B, M), dtype=torch.long, device=dev)
        # one flat index buffer per scale (one zero fill), carved into the pieces' (B, m_piece, k) tensors
        self.ball_flat = [torch.zeros((B * M * k,), dtype=torch.long, device=dev) for _, k in SCALES]
        self.ball_idx = [[f[B * a * k:B * b * k].view(B, b - a, k) for a, b in zip(cuts[:-1], cuts[1:])]
                         for f, (_, k) in zip(self.ball_flat, SCALES)]
        self.g_xyz = [[torch.empty((B, 3, b - a, k), device=dev) for _, k in SCALES] for a, b in zip(cuts[:-1], cuts[1:])]
        self.g_feat = [[torch.empty((B, C_FEAT, b - a, k), device=dev) for _, k in SCALES]
                       for a, b in zip(cuts[:-1], cuts[1:])]
        self.tnn_d = torch.empty((B, N, 3), device=dev)
        self.tnn_i = torch.zeros((B, N, 3), dtype=torch.long, device=dev)
        self.interp = torch.empty((B, C_INTERP, N), device=dev)
        self.sgrad = torch.empty((B, C_INTERP, M), device=dev)
        self.side = torch.cuda.Stream(device=dev)
        self.start = torch.cuda.Event()
        self.sampled = [torch.cuda.Event() for _ in range(self.pieces)]
        self.grouped = torch.cuda.Event()
        self.radii, self.ks = [r for r, _ in SCALES], [k for _, k in SCALES]

    def run(self):
        torch, ext, inp, cuts = self.torch, self.ext, self.inp, self.cuts
        main = torch.cuda.current_stream()
        self.start.record(main)
        self.fps_dist.fill_(1e10)
        for p in range(self.pieces):
            ext.furthest_point_sample_rounds(self.pts, self.fps_dist, self.fps_idx, M, cuts[p], cuts[p + 1])
            self.sampled[p].record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.start)
            for t in self.ball_flat:
                t.zero_()
            for p in range(self.pieces):
                self.side.wait_event(self.sampled[p])
                idx = self.fps_idx[:, cuts[p]:cuts[p + 1]]
                q = torch.gather(self.pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3))  # (B, m_piece, 3) centroids
                ext.ball_query_multi(q, self.pts, [t[p] for t in self.ball_idx], self.radii, self.ks)
                for i in range(len(SCALES)):
                    ext.group_gather_forward(self.pts_cf, self.ball_idx[i][p], self.g_xyz[p][i])
                    ext.group_gather_forward(inp["feats"], self.ball_idx[i][p], self.g_feat[p][i])
            self.grouped.record(self.side)
        # feature propagation needs every centroid: after the last piece, on the main stream (beside the last grouping)
        q_all = torch.gather(self.pts, 1, self.fps_idx.unsqueeze(-1).expand(-1, -1, 3))
        ext.three_nn(self.pts, q_all, self.tnn_d, self.tnn_i)
        w = 1.0 / (self.tnn_d + 1e-5)
        w = w / w.sum(dim=2, keepdim=True)
        ext.three_interpolate_forward(inp["sfeats"], self.tnn_i, w, self.interp)
        ext.three_interpolate_backward(inp["grad"], self.tnn_i, w, self.sgrad, M)
        main.wait_event(self.grouped)


def algorithmic_bytes():
    """SURVEY.md 8(d) formulas, per launch."""
    out = {}
    for _, k in SCALES:
        out[f"group_gather_feat_k{k}"] = B * M * k * 8 + B * C_FEAT * N * 4 + B * C_FEAT * M * k * 4
        out[f"group_gather_xyz_k{k}"] = B * M * k * 8 + B * 3 * N * 4 + B * 3 * M * k * 4
        out[f"ball_query_k{k}"] = B * (M + N) * 12 + B * M * k * 8
    out["three_interp_fwd"] = B * C_INTERP * M * 4 + B * N * 3 * 8 + B * N * 3 * 4 + B * C_INTERP * N * 4
    out["three_interp_bwd"] = out["three_interp_fwd"]
    out["three_nn"] = B * (N + M) * 12 + B * N * 3 * 12
    out["fps"] = B * (12 * N + 8 * M)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# CPU legs (rank 0, N = 1 only).  The only place bench.py touches oracle/.
# ----------------------------------------------------------------------------------------------------------------------
def _cpu_inputs(seed):
    import numpy as np
    import torch

    g = torch.Generator().manual_seed(seed)
    pts = torch.rand((B, N, 3), generator=g).numpy()
    feats = torch.randn((B, C_FEAT, N), generator=g).numpy()
    sfeats = torch.randn((B, C_INTERP, M), generator=g).numpy()
    grad = torch.randn((B, C_INTERP, N), generator=g).numpy()
    return pts, feats, sfeats, grad, np


def cpu_baseline(seed, budget_s=12.0):
    """The same step on the CPU oracle (port of the reference kernel semantics), all host cores."""
    import oracle

    cores = min(oracle.max_threads(), os.cpu_count() or 1)
    oracle.set_threads(cores)
    pts, feats, sfeats, grad, np = _cpu_inputs(seed)
    pts_cf = np.ascontiguousarray(np.swapaxes(pts, 1, 2))
    reps, t0 = 0, time.perf_counter()
    while True:
        idx = oracle.furthest_point_sample(pts, M)
        q = np.ascontiguousarray(np.take_along_axis(pts, idx[..., None], axis=1))
        for r, k in SCALES:
            bi = oracle.ball_query(q, pts, k, r)
            oracle.group_gather_forward(pts_cf, bi)
            oracle.group_gather_forward(feats, bi)
        d3, i3 = oracle.three_nn(pts, q)
        w = 1.0 / (d3 + 1e-5)
        w = (w / w.sum(-1, keepdims=True)).astype(np.float32)
        oracle.three_interpolate_forward(sfeats, i3, w)
        oracle.three_interpolate_backward(grad, i3, w, M)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 400:
            break
    oracle.set_threads(1)
    return {
        "value": QUERIES_PER_STEP * reps / el,
        "unit": "queries/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{reps} full step(s) of the same B={B},N={N} workload on the CPU oracle "
                  f"(oracle/v3d_oracle.c, OpenMP x{cores}), {el:.1f} s",
    }, q


def cpu_baseline_reference(seed, q, budget_s=10.0):
    """The reference's CPU path for the QUERY operators of the step, as vision3d/array_ops/ball_query.py:8-9
    (cKDTree(s).query_ball_point(q, r)) and array_ops/knn.py:11-12 (cKDTree(s).query(q, k)) run them, all host cores
    (workers=-1), tree construction included as in the reference.  No gathers / interpolation: array_ops has none."""
    import warnings

    from scipy.spatial import cKDTree

    pts = _cpu_inputs(seed)[0]
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
    done, t0 = 0, time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for b in range(B):  # bounded sample: as many whole clouds as fit the budget
            tree = cKDTree(pts[b])
            for r, _ in SCALES:
                tree.query_ball_point(q[b], r, workers=-1)
            cKDTree(q[b]).query(pts[b], k=3, workers=-1)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
    el = time.perf_counter() - t0
    out = {
        "value": (len(SCALES) * M + N) * done / el,
        "unit": "queries/s",
        "cores": cores,
        "kind": "reference CPU path restated on scipy (array_ops/ball_query.py:8-9, array_ops/knn.py:11-12)",
        "sample": f"{done} of the {B} clouds of the same inputs: 3 query_ball_point scales + k=3 query per cloud, "
                  f"workers=-1, {el:.1f} s",
    }
    out["fps"] = cpu_fps_reference(pts[0])
    return out


def cpu_fps_reference(cloud, budget_s=4.0):
    """The reference's own CPU sampling, array_ops/furthest_point_sample.py:22-66 (`native_furthest_point_sample`),
    restated as it is written there: a Python list of unselected indices, one float64 `np.linalg.norm` per (round, point),
    running minima, strict `>` for the farthest.  One thread.  Bounded sample: as many ROUNDS of one cloud of the same
    inputs as fit the budget (a whole M = 1024 sampling of 4096 points takes minutes this way); the rate is distance
    updates per second, and `points_per_sec_extrapolated` = N / (time of a whole sampling at that rate), the unit of
    `fps_points_per_sec`."""
    import numpy as np

    pts = np.asarray(cloud, dtype=np.float64)
    n = pts.shape[0]
    unselected = list(range(n))
    running = np.full((n,), np.inf)
    best, rounds, updates, t0 = 0, 0, 0, time.perf_counter()
    while rounds < M - 1:
        sel = unselected.pop(best)
        best, best_d = -1, 0.0
        for pos, p in enumerate(unselected):
            d = min(running[p], np.linalg.norm(pts[p] - pts[sel]))
            running[p] = d
            if d > best_d:
                best, best_d = pos, d
        rounds += 1
        updates += len(unselected)
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    whole = sum(n - 1 - r for r in range(M - 1))  # updates of a whole sampling
    return {"value": updates / el, "unit": "distance updates/s", "cores": 1,
            "points_per_sec_extrapolated": n / (whole / (updates / el)),
            "kind": "reference NumPy FPS restated (array_ops/furthest_point_sample.py:22-66)",
            "sample": f"{rounds} of the {M - 1} rounds of one cloud of {n} points, {el:.1f} s"}


# ----------------------------------------------------------------------------------------------------------------------
# The other BASELINE configs: short loops after the timed region.
# ----------------------------------------------------------------------------------------------------------------------
def _loop_ms(torch, fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def c5_features(torch, device, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.nn.functional.normalize(torch.randn((C5_ROWS, C5_DIM), generator=g), dim=1).to(device)
    y = torch.nn.functional.normalize(torch.randn((C5_ROWS, C5_DIM), generator=g), dim=1).to(device)
    return x, y


def per_config_single(torch, device):
    """C1, C3, C4 and the C5 one-GPU shard on this rank's GPU (N = 1 run)."""
    from vision3d_amd import ext

    out = {}
    g = torch.Generator().manual_seed(11)
    # ---- C1: PointNet++ SSG set abstraction, B=16, N=1024 unit-sphere cloud, FPS -> 512, r=0.2 k=32, C=64
    b1, n1, m1, k1, c1 = 16, 1024, 512, 32, 64
    p = torch.randn((b1, n1, 3), generator=g)
    p = p - p.mean(dim=1, keepdim=True)
    p = (p / p.norm(dim=2).amax(dim=1).view(b1, 1, 1)).to(device)
    p_cf = p.transpose(1, 2).contiguous()
    f = torch.randn((b1, c1, n1), generator=g).to(device)
    dist = torch.empty((b1, n1), device=device)
    idx = torch.zeros((b1, m1), dtype=torch.long, device=device)
    q_cf = torch.empty((b1, 3, m1), device=device)
    bi = torch.zeros((b1, m1, k1), dtype=torch.long, device=device)
    gx = torch.empty((b1, 3, m1, k1), device=device)
    gf = torch.empty((b1, c1, m1, k1), device=device)

    def c1_step():
        dist.fill_(1e10)
        ext.furthest_point_sample(p, dist, idx, m1)
        ext.gather_forward(p_cf, idx, q_cf)
        q = q_cf.transpose(1, 2).contiguous()
        bi.zero_()
        ext.ball_query(q, p, bi, 0.2, k1)
        ext.group_gather_forward(p_cf, bi, gx)
        ext.group_gather_forward(f, bi, gf)

    def c1_fps():
        dist.fill_(1e10)
        ext.furthest_point_sample(p, dist, idx, m1)

    ms = _loop_ms(torch, c1_step, 50)
    ms_fps = _loop_ms(torch, c1_fps, 50)
    out["c1_pointnet2_ssg"] = {"workload": "B16 N1024 FPS->512, ball_query r0.2 k32, group_gather C64 (+xyz)",
                               "ms_per_step": ms, "queries_per_sec": b1 * m1 / (ms * 1e-3),
                               "fps_points_per_sec": b1 * n1 / (ms_fps * 1e-3), "fps_ms": ms_fps}
    # ---- C2 queries one by one: the MSG shapes (B=32, N=4096, M=1024) through ball_query / knn_points / three_nn
    b2, n2, m2 = 32, 4096, 1024
    p2 = torch.rand((b2, n2, 3), generator=g).to(device)
    q2 = p2[:, :m2].contiguous()
    c2q = {"workload": "B32 N4096 M1024 uniform cloud: one launch per entry"}
    for k2, r2 in ((16, 0.1), (32, 0.2)):
        bi2 = torch.zeros((b2, m2, k2), dtype=torch.long, device=device)
        kd2 = torch.empty((b2, m2, k2), device=device)
        ms_b = _loop_ms(torch, lambda: ext.ball_query(q2, p2, bi2, r2, k2), 30)
        ms_k = _loop_ms(torch, lambda: ext.knn_points(q2, p2, kd2, bi2, k2), 30)
        ms_l = _loop_ms(torch, lambda: ext.knn_lex(q2, p2, kd2, bi2, k2), 30)
        c2q[f"k{k2}"] = {"ball_query_ms": ms_b, "knn_points_ms": ms_k, "knn_points_queries_per_sec": b2 * m2 / (ms_k * 1e-3),
                         "knn_index_order_ms": ms_l}
    td2 = torch.empty((b2, n2, 3), device=device)
    ti2 = torch.empty((b2, n2, 3), dtype=torch.long, device=device)
    ms_t = _loop_ms(torch, lambda: ext.three_nn(p2, q2, td2, ti2), 30)
    c2q["three_nn_4096_from_1024"] = {"ms": ms_t, "queries_per_sec": b2 * n2 / (ms_t * 1e-3)}
    out["c2_queries"] = c2q
    del p2, q2, td2, ti2
    # ---- C3: DGCNN feature-space kNN, B=32, N=1024, C=64, k=20 (self included)
    b3, n3, c3, k3 = 32, 1024, 64, 20
    x = torch.randn((b3, n3, c3), generator=g).to(device)
    d = torch.empty((b3, n3, k3), device=device)
    i = torch.empty((b3, n3, k3), dtype=torch.long, device=device)
    ms = _loop_ms(torch, lambda: ext.knn_feat(x, x, d, i, k3), 50)
    tf = 2.0 * b3 * n3 * n3 * c3 / (ms * 1e-3) / 1e12
    out["c3_dgcnn_feature_knn"] = {"workload": "B32 N1024 C64 k20, fused fp32-MFMA distance + top-k", "ms": ms,
                                   "queries_per_sec": b3 * n3 / (ms * 1e-3), "tflops": tf,
                                   "mfma_frac": tf / MFMA_F32_PEAK_TFLOPS}
    # ---- C4: 16 clouds x 30,000 points in a 3 x 3 x 2.5 m box: radius search (r 0.0625, limit 38) + grid subsampling
    nc, n4, lim = 16, 30000, 38
    pts = (torch.rand((nc * n4, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(device)
    lengths = torch.full((nc,), n4, dtype=torch.long, device=device)
    table = torch.empty((nc * n4, lim), dtype=torch.long, device=device)
    o, ol = torch.empty_like(pts), torch.empty_like(lengths)
    num = torch.zeros((), dtype=torch.long, device=device)
    ms_r = _loop_ms(torch, lambda: ext.radius_search_pack(pts, pts, lengths, lengths, table, 0.0625, lim), 30)
    ms_g = _loop_ms(torch, lambda: ext.grid_subsample_pack(pts, lengths, o, ol, num, 0.025), 30)
    out["c4_kpconv_pyramid_stage"] = {"workload": "16 clouds x 30000 pts, radius 0.0625 limit 38, voxel 0.025",
                                      "radius_search_ms": ms_r, "radius_queries_per_sec": nc * n4 / (ms_r * 1e-3),
                                      "grid_subsample_ms": ms_g, "grid_points_per_sec": nc * n4 / (ms_g * 1e-3)}
    # decoder side of the same stage: every point's nearest / three nearest points of the next (coarser) stage, pack mode
    sub = (torch.rand((nc * n4 // 4, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(device)
    sub_lengths = torch.full((nc,), n4 // 4, dtype=torch.long, device=device)
    for kk in (1, 3):
        kd4 = torch.empty((nc * n4, kk), device=device)
        ki4 = torch.empty((nc * n4, kk), dtype=torch.long, device=device)
        ms_k = _loop_ms(torch, lambda: ext.knn_pack(pts, sub, lengths, sub_lengths, kd4, ki4, kk), 30)
        out["c4_kpconv_pyramid_stage"][f"knn_pack_k{kk}_ms"] = ms_k
        out["c4_kpconv_pyramid_stage"][f"knn_pack_k{kk}_queries_per_sec"] = nc * n4 / (ms_k * 1e-3)
    del pts, table, o, sub
    job = C4Pyramid(torch, None, device, 1, 0)
    job.count_queries()
    ms_p = _loop_ms(torch, job.step, 5, warm=2)
    out["c4_kpconv_pyramid_stage"]["pyramid"] = job.report(ms_p, job.queries_per_step)
    del job
    # ---- C5: one GPU's share of the 100k x 100k x 256 matching when the reference set is split 8 ways
    xq, y = c5_features(torch, device)
    ys = y[: C5_ROWS // 8].contiguous()
    d = torch.empty((1, C5_ROWS, 3), device=device)
    i = torch.empty((1, C5_ROWS, 3), dtype=torch.long, device=device)
    ms = _loop_ms(torch, lambda: ext.knn_feat(xq.unsqueeze(0), ys.unsqueeze(0), d, i, 3), 5, warm=1)
    tf = 2.0 * C5_ROWS * (C5_ROWS // 8) * C5_DIM / (ms * 1e-3) / 1e12
    out["c5_matching_one_gpu_shard"] = {"workload": "100000 queries x 12500 references (1/8 of the set) x 256, k=3",
                                        "ms": ms, "queries_per_sec": C5_ROWS / (ms * 1e-3), "tflops": tf,
                                        "mfma_frac": tf / MFMA_F32_PEAK_TFLOPS}
    return out


C4_CLOUDS_PER_RANK, C4_POINTS = 16, 30000  # 8 samples per GPU x 2 clouds of ~30k points
C4_STAGES, C4_VOXEL, C4_RADIUS, C4_LIMITS = 4, 0.025, 0.0625, (38, 36, 36, 38)


class C4Pyramid:
    """BASELINE config 4 under pack-mode batch sharding: the batch = 16 clouds per rank; every rank
    generates the SAME whole batch from one seed (what a collate function would hand out), keeps its clouds through
    sharding.shard_pack and builds the 4-stage grid + radius pyramid of those clouds
    (ops.build_grid_and_radius_graph_pyramid; reference: models/graph_pyramid/graph_pyramid.py:9-63 under DDP,
    utils/distributed.py:11-18).  No data-path collective.  verify(): this rank's first-stage table and first subsampling,
    put into the whole batch's numbering, equal the rows a single process computes over the whole batch."""

    def __init__(self, torch, dist, device, world, rank):
        from vision3d_amd import ops, sharding

        self.torch, self.dist, self.ops, self.sharding = torch, dist, ops, sharding
        self.world, self.rank, self.device = world, rank, device
        g = torch.Generator().manual_seed(11)
        nc = C4_CLOUDS_PER_RANK * world
        self.all_points = (torch.rand((nc * C4_POINTS, 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(device)
        self.all_lengths_host = torch.full((nc,), C4_POINTS, dtype=torch.long)
        self.all_lengths = self.all_lengths_host.to(device)
        shard = sharding.shard_pack(self.all_points, self.all_lengths_host, world, rank)
        self.shard = shard
        self.points = shard.points.contiguous()
        self.lengths = self.all_lengths[shard.cloud_begin:shard.cloud_end].contiguous()
        self.queries_per_step = None

    def step(self):
        return self.ops.build_grid_and_radius_graph_pyramid(self.points, self.lengths, C4_STAGES, C4_VOXEL, C4_RADIUS,
                                                            list(C4_LIMITS))

    def count_queries(self):
        pyr = self.step()
        rows = sum(t.shape[0] for key in ("neighbors", "subsampling", "upsampling") for t in pyr[key])
        self.queries_per_step = rows
        self.stage_points = [int(p.shape[0]) for p in pyr["points"]]
        return rows

    def verify(self):
        torch, sh = self.torch, self.sharding
        ok = True
        if self.world > 1:
            total = int(self.all_points.shape[0])
            local_rows = self.shard.row_end - self.shard.row_begin
            mine = self.ops.radius_search_pack_mode(self.points, self.points, self.lengths, self.lengths, C4_RADIUS,
                                                    C4_LIMITS[0])
            whole = self.ops.radius_search_pack_mode(self.all_points, self.all_points, self.all_lengths, self.all_lengths,
                                                     C4_RADIUS, C4_LIMITS[0])
            ok = torch.equal(sh.globalize_pack_table(mine, local_rows, self.shard.row_begin, total),
                             whole[self.shard.row_begin:self.shard.row_end])
            del whole, mine
            sp, sl = self.ops.grid_subsample_pack_mode(self.points, self.lengths, 2 * C4_VOXEL)
            wp, wl = self.ops.grid_subsample_pack_mode(self.all_points, self.all_lengths, 2 * C4_VOXEL)
            sub = sh.shard_pack(wp, wl.cpu(), self.world, self.rank)
            ok = ok and torch.equal(sp, sub.points) and torch.equal(sl.cpu(), sub.lengths)
        flag = torch.tensor([1 if ok else 0], device=self.device)
        if self.world > 1:
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(flag.item())

    def report(self, ms, total_queries):
        return {"workload": f"{C4_CLOUDS_PER_RANK} clouds x {C4_POINTS} pts per rank x {self.world} rank(s): {C4_STAGES}-stage "
                            f"pyramid, voxel {C4_VOXEL} x2 per stage, radius {C4_RADIUS} x2 per stage, limits {list(C4_LIMITS)} "
                            "(3 grid subsamplings + 10 radius searches per step)",
                "sharding": f"pack-mode batch x{self.world} by lengths (sharding.shard_pack), no data-path collective",
                "pyramid_ms": ms, "radius_queries_per_step_all_ranks": total_queries,
                "radius_queries_per_sec": total_queries / (ms * 1e-3),
                "input_points_per_sec": C4_CLOUDS_PER_RANK * C4_POINTS * self.world / (ms * 1e-3),
                "stage_points_rank0": self.stage_points}


def run_c4(torch, dist, device, world, rank, iters, max_over_ranks, bracket, warm=2):
    """Times the sharded pyramid: barrier + synchronize either side, MAX over ranks, queries summed over ranks."""
    job = C4Pyramid(torch, dist, device, world, rank)
    ok = job.verify()
    rows = job.count_queries()
    for _ in range(warm):
        job.step()
    bracket()
    t0 = time.perf_counter()
    for _ in range(iters):
        job.step()
    bracket()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    total = rows
    if dist is not None:
        t = torch.tensor([rows], dtype=torch.int64, device=device)
        dist.all_reduce(t)
        total = int(t.item())
    rep = dict(job.report(elapsed / iters * 1e3, total), equals_single_gpu_result=ok)
    return job, elapsed, total, rep


class C5Sharded:
    """BASELINE config 5 on `world` ranks: queries replicated, reference set split by sharding.batch_bounds, one
    all-gather of the partial top-k (RCCL) + merge; verified on the device against the single-GPU search."""

    def __init__(self, torch, dist, device, world, rank, k=3):
        from vision3d_amd import sharding

        self.torch, self.dist, self.sharding, self.k = torch, dist, sharding, k
        self.world, self.rank = world, rank
        self.x, self.y = c5_features(torch, device)  # same seed on every rank: replicated inputs
        self.begin, self.end = sharding.batch_bounds(C5_ROWS, world, rank)
        self.local = self.y[self.begin:self.end].contiguous()

    def step(self):
        return self.sharding.knn_reference_sharded(self.x, self.local, self.k, self.begin)

    def verify(self):
        torch = self.torch
        d, i = self.step()
        fd, fi = self.sharding._hip_search(self.x, self.y, self.k, 0)  # the single-GPU result, on this rank
        ok = bool(torch.equal(i, fi)) and bool(torch.equal(d, fd))
        flag = torch.tensor([1 if ok else 0], device=self.x.device)
        if self.world > 1:
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(flag.item())

    def exchange_ms(self, iters=10):
        """The collective alone: all-gather of (rows,k) fp32 + int64 partial lists."""
        torch, dist = self.torch, self.dist
        if self.world == 1:
            return 0.0
        pd = torch.zeros((C5_ROWS, self.k), device=self.x.device)
        pi = torch.zeros((C5_ROWS, self.k), dtype=torch.long, device=self.x.device)

        def fn():  # exactly the two calls knn_reference_sharded makes
            self.sharding._all_gather_rows(pd)
            self.sharding._all_gather_rows(pi)
        return _loop_ms(torch, fn, iters)

    def report(self, ms):
        flops = 2.0 * C5_ROWS * C5_ROWS * C5_DIM
        return {"workload": f"100000 x 100000 x 256 k={self.k}, reference set split over {self.world} rank(s)",
                "ms": ms, "queries_per_sec": C5_ROWS / (ms * 1e-3), "tflops_whole_job": flops / (ms * 1e-3) / 1e12,
                "mfma_frac_per_gpu": flops / self.world / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                "allgather_bytes_per_rank": C5_ROWS * self.k * 12,
                "allgather_bytes_received": self.world * C5_ROWS * self.k * 12}


# ----------------------------------------------------------------------------------------------------------------------
def _trace(msg):
    if os.environ.get("V3D_BENCH_TRACE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def rank_main(args):
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the operators have no CPU fallback)"
    torch.cuda.set_device(local_rank)  # reference: utils/distributed.py:14
    device = torch.device("cuda", local_rank)
    dist, rccl_world = None, 1
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":  # RCCL; reference: utils/distributed.py:15
            dist.init_process_group(backend="nccl", device_id=device)
        else:  # rehearsal of the rank plumbing on a one-GPU box (gloo stages device tensors through the host)
            dist.init_process_group(backend="gloo")
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        rccl_world = int(ones.item())
        assert rccl_world == dist.get_world_size() == world
        _trace(f"process group up: backend {args.backend}, world {rccl_world}")

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def bracket():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    line = {"higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "unit": "queries/s", "metric": "neighbor_queries_per_sec"}
    if dist is not None:
        line["collective_backend"] = "rccl (torch nccl)" if args.backend == "nccl" else args.backend
        line["rccl_world_size"] = rccl_world

    if args.workload == "c5_refset":
        job = C5Sharded(torch, dist, device, world, rank, k=args.c5_k)
        _trace("c5 inputs ready")
        ok = job.verify()
        _trace(f"c5 verified: {ok}")
        for _ in range(args.warmup):
            job.step()
        bracket()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            job.step()
        bracket()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        ex_ms = max_over_ranks(job.exchange_ms())
        if rank == 0:
            ms = elapsed / args.steps * 1e3
            rep = job.report(ms)
            line.update(value=C5_ROWS * args.steps / elapsed, ms_per_step=ms, scaling="strong",
                        config={"workload": "geotransformer_matching_100k_refset_sharded", "k": args.c5_k,
                                "sharding": f"reference set x{world}, one all-gather of the partial top-k + merge"},
                        equals_single_gpu_result=ok, allgather_ms=ex_ms, **{k: rep[k] for k in rep if k != "ms"})
            tf = rep["tflops_whole_job"] / world
            line["roofline"] = {"bound": "mfma", "kernel": "pairwise_distance_kernel (fp32 MFMA) inside knn_feat",
                                "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                                "note": "whole knn_feat call per GPU (distance + top-k + exact refine), not the GEMM alone"}
            assert ok, "sharded result differs from the single-GPU result"
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    if args.workload == "c4_pyramid":
        job, elapsed, total, rep = run_c4(torch, dist, device, world, rank, args.steps, max_over_ranks, bracket,
                                          warm=args.warmup)
        if rank == 0:
            line.update(value=total * args.steps / elapsed, ms_per_step=elapsed / args.steps * 1e3, scaling="weak",
                        metric="neighbor_queries_per_sec",
                        config={"workload": "kpconv_pyramid_pack_mode", "sharding": rep["sharding"],
                                "clouds_per_gpu": C4_CLOUDS_PER_RANK, "points_per_cloud": C4_POINTS},
                        **{k: rep[k] for k in rep if k != "sharding"})
            assert rep["equals_single_gpu_result"], "sharded pyramid differs from the single-process result"
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    inp = make_inputs(1234 + rank, device)  # every rank owns different clouds (batch sharding)
    # `--streams S` batches are in flight at once, each on its own HIP stream with its own buffers: FPS is a serial
    # chain that occupies only B of the 256 CUs, so the next batch's FPS runs under this batch's gathers / queries.
    nstream = max(1, args.streams)
    streams = [torch.cuda.Stream(device=device) for _ in range(nstream)]
    steps = []
    for s in streams:
        with torch.cuda.stream(s):
            steps.append(Step(inp, device, fused_msg=args.fused_msg))
    torch.cuda.synchronize()

    timers, dom_timers = [], []

    def run_steps(n, timed=False, nstr=nstream):
        # Per-operator HIP events go around every launch of every EVENT_EVERY-th timed step (alternating streams): an
        # event pair is a barrier packet either side of the kernel, and around all 14 launches of every step they cost
        # 5-8 % of the step time (tools/host_overhead.py).  The roofline kernel alone carries a pair on every timed step.
        for i in range(n):
            st = steps[i % nstr]
            st.timers = timers if (timed and i % EVENT_EVERY == 0) else None
            st.dom_timers = dom_timers if timed else None
            with torch.cuda.stream(streams[i % nstr]):
                st.run()

    run_steps(args.warmup)
    bracket()
    t0 = time.perf_counter()
    run_steps(args.steps, timed=True)
    bracket()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    _trace("c2 timed region done")
    # spread: the same K-step region REPEATS - 1 more times (no events except the dominant kernel's); `value` stays the
    # first region's, the one the contract describes
    region_ms = [elapsed / args.steps * 1e3]
    for _ in range(max(0, args.repeats - 1)):
        bracket()
        t0r = time.perf_counter()
        run_steps(args.steps, timed=True)
        bracket()
        region_ms.append(max_over_ranks(time.perf_counter() - t0r) / args.steps * 1e3)

    # the same step with ONE batch in flight (what a single training step sees): not the headline, printed beside it
    run_steps(min(args.warmup, 3), nstr=1)
    bracket()
    t1 = time.perf_counter()
    run_steps(args.steps, nstr=1)
    bracket()
    elapsed_single = max_over_ranks(time.perf_counter() - t1)

    # ... and one batch in flight with the sampling cut into pieces (PipelinedStep.cuts), its consumers overlapped
    elapsed_pipe = None
    pipe_cuts = None
    if world == 1:
        pstep = PipelinedStep(inp, device)
        for _ in range(min(args.warmup, 3)):
            pstep.run()
        bracket()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            pstep.run()
        bracket()
        elapsed_pipe = time.perf_counter() - t1
        pipe_equal = bool(torch.equal(pstep.fps_idx, steps[0].fps_idx) and torch.equal(pstep.sgrad, steps[0].sgrad)
                          and all(torch.equal(torch.cat([pstep.g_feat[p][i] for p in range(pstep.pieces)], dim=2),
                                              steps[0].g_feat[i]) for i in range(len(SCALES))))
        pipe_cuts = pstep.cuts
        del pstep

    c5 = None
    if dist is not None and not args.no_per_config:
        job = C5Sharded(torch, dist, device, world, rank, k=3)
        _trace("c5 inputs ready")
        ok = job.verify()
        _trace(f"c5 verified: {ok}")
        bracket()
        t2 = time.perf_counter()
        for _ in range(3):
            job.step()
        bracket()
        c5_ms = max_over_ranks(time.perf_counter() - t2) / 3 * 1e3
        ex_ms = max_over_ranks(job.exchange_ms())
        c5 = dict(job.report(c5_ms), equals_single_gpu_result=ok, allgather_ms=ex_ms)
        del job
    c4 = None
    if dist is not None and not args.no_per_config:
        torch.cuda.empty_cache()
        job, _, _, c4 = run_c4(torch, dist, device, world, rank, 5, max_over_ranks, bracket)
        _trace(f"c4 sharded pyramid: {c4['equals_single_gpu_result']}")
        del job

    if rank == 0:
        per_op = {}
        for name, s, e in timers + dom_timers:
            per_op.setdefault(name, []).append(s.elapsed_time(e))
        per_op_ms = {k: float(np.mean(v)) for k, v in per_op.items()}
        alg = algorithmic_bytes()
        dom = DOMINANT
        achieved = alg[dom] / (per_op_ms[dom] * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dom)
                traffic_source = ("profiles/traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "command, recorded " + str(tj.get("_recorded", "r01")) + " -- not measured by this run")
            except Exception:
                traffic = None
        ms_per_step = elapsed / args.steps * 1e3
        line.update({
            "value": QUERIES_PER_STEP * args.steps * world / elapsed,
            "ms_per_step": ms_per_step,
            "ms_per_step_min": float(np.min(region_ms)), "ms_per_step_median": float(np.median(region_ms)),
            "ms_per_step_max": float(np.max(region_ms)), "timed_region_repeats": len(region_ms),
            "ms_per_step_single_stream": elapsed_single / args.steps * 1e3,
            "value_single_stream": QUERIES_PER_STEP * args.steps * world / elapsed_single,
            "ms_per_step_single_batch_pipelined": None if elapsed_pipe is None else elapsed_pipe / args.steps * 1e3,
            "single_batch_pipelined": None if elapsed_pipe is None else {
                "what": f"one batch in flight, FPS in {len(pipe_cuts) - 1} launches (v3d_fps_resume_f32: rounds "
                        + ", ".join(f"{a}-{b - 1}" for a, b in zip(pipe_cuts[:-1], pipe_cuts[1:]))
                        + "), ball queries / gathers of each piece on a second stream under the sampling of the next; "
                          "not the reference's call pattern",
                "equals_plain_step": pipe_equal},
            "scaling": "weak",
            "config": {"workload": "pointnet2_msg_fp_s3dis", "B_per_gpu": B, "N": N, "M": M,
                       "scales": [list(s) for s in SCALES], "C_group": C_FEAT, "C_interp": C_INTERP,
                       "sharding": f"batch x{world} (independent clouds, no data-path collective)",
                       "batches_in_flight": nstream,
                       "ball_query": "one launch for the 3 MSG scales" if args.fused_msg else "one launch per scale"},
            "fps_points_per_sec": B * N * world / (per_op_ms["fps"] * 1e-3),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes": alg[dom], "avg_ms": per_op_ms[dom],
                         "launches_timed": len(per_op[dom])},
            "per_op_ms": per_op_ms,
            "per_op_alg_GBs": {k: alg[k] / (per_op_ms[k] * 1e-3) / 1e9 for k in alg if k in per_op_ms},
        })
        per_config = {}
        if c5 is not None:
            per_config["c5_matching_refset_sharded"] = c5
        if c4 is not None:
            per_config["c4_kpconv_pyramid_sharded"] = c4
        if world == 1 and not args.no_per_config:
            del steps[:]
            torch.cuda.empty_cache()
            per_config.update(per_config_single(torch, device))
        line["per_config"] = per_config
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"], q = cpu_baseline(1234)
            line["cpu_baseline_reference"] = cpu_baseline_reference(1234, q)
        elif world == 1:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=("c2", "c5_refset", "c4_pyramid"), default="c2")
    ap.add_argument("--c5-k", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region is run this many times in all; `value` is the first region's, the "
                         "min / median / max over the repeats ride along")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-config", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="batches in flight (one HIP stream each)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL (default); gloo only to rehearse the rank plumbing where ranks must share a GPU")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (one-GPU rehearsal, with gloo)")
    ap.add_argument("--fused-msg", action="store_true",
                    help="one ext.ball_query_multi launch for the three MSG scales instead of the reference's three "
                         "ball_query calls (same outputs; off by default: the default step is the reference's call pattern)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not args.share_gpu:
            import torch  # device_count() does not create a context

            have = torch.cuda.device_count()
            if have < args.gpus:
                sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible")
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank_main(args)


if __name__ == "__main__":
    main()
