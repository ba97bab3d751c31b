#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json config[1] on N GPUs of one node.

Workload "pointnet2_msg_fp_s3dis" (SURVEY.md 8d, C2), one STEP = one pass over one batch of B=32 synthetic
S3DIS-sized blocks of N=4096 points, inputs already resident in HBM, outputs pre-allocated:
    FPS 4096 -> 1024 | gather centroids | for (r,k) in (0.1,16),(0.2,32),(0.4,128): ball_query,
    group_gather(xyz, 3 ch), group_gather(features, 64 ch) | three_nn 4096 <- 1024 | interpolation weights
    (torch elementwise, as the reference's FeaturePropagate does) | three_interpolate fwd + bwd (128 ch).
Every operator call goes through vision3d_amd.ext -> the C ABI -> hand-written gfx950 kernels.

Metric (BASELINE.json): neighbor queries/sec = (3*B*M ball queries + B*Q three_nn queries) / step time, whole job
over all ranks; FPS points/sec and per-operator timings ride along in the same JSON line.  Multi-GPU: independent
clouds shard by batch (one process per GPU, no data-path collective), so scaling is "weak": every rank runs its
own B=32 batch; the only collectives are the timing barrier and a MAX all-reduce of the elapsed time.

`roofline` describes the dominant HBM kernel (group_gather forward, k=128, C=64): algorithmic bytes per launch /
its average duration measured with HIP events inside the timed steps (every EVENT_EVERY-th step carries the events).  `cpu_baseline` is the CPU oracle (a port of
the reference kernels' semantics, OpenMP over clouds/queries) timed on this host on the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable by a float4 copy)

B, N, M, C_FEAT, C_INTERP = 32, 4096, 1024, 64, 128
SCALES = ((0.1, 16), (0.2, 32), (0.4, 128))
QUERIES_PER_STEP = len(SCALES) * B * M + B * N  # ball queries + three_nn queries
EVENT_EVERY = 5  # timed steps whose launches carry HIP events (per-operator durations, roofline)


def make_inputs(seed, device):
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand((B, N, 3), generator=g)  # ~U[0,1)^3: a 1 m block
    feats = torch.randn((B, C_FEAT, N), generator=g)
    sfeats = torch.randn((B, C_INTERP, M), generator=g)
    grad = torch.randn((B, C_INTERP, N), generator=g)
    return {k: v.to(device) for k, v in dict(pts=pts, feats=feats, sfeats=sfeats, grad=grad).items()}


class Step:
    """Pre-allocated buffers + the operator sequence; `timers` collects (name, start_event, end_event)."""

    def __init__(self, inp, device, fused_msg=False):
        from vision3d_amd import ext

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
        self.timers = None

    def _timed(self, name, fn):
        if self.timers is None:
            fn()
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.timers.append((name, s, e))

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


def cpu_baseline(seed, budget_s=12.0):
    """The same step on the CPU oracle (port of the reference kernel semantics), all host cores."""
    import oracle

    cores = min(oracle.max_threads(), os.cpu_count() or 1)
    oracle.set_threads(cores)
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand((B, N, 3), generator=g).numpy()
    feats = torch.randn((B, C_FEAT, N), generator=g).numpy()
    sfeats = torch.randn((B, C_INTERP, M), generator=g).numpy()
    grad = torch.randn((B, C_INTERP, N), generator=g).numpy()
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
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="batches in flight (one HIP stream each)")
    ap.add_argument("--fused-msg", action="store_true",
                    help="one ext.ball_query_multi launch for the three MSG scales instead of the reference's three "
                         "ball_query calls (same outputs; off by default: the default step is the reference's call pattern)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the operators have no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=device)  # RCCL

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

    timers = []

    def run_steps(n, timed=False):
        # Per-operator HIP events go around every launch of every EVENT_EVERY-th timed step (alternating streams): an
        # event pair is a barrier packet either side of the kernel, and on every step they cost 5-8 % of the step time
        # (tools/host_overhead.py), which is not the operators' time.
        for i in range(n):
            st = steps[i % nstream]
            st.timers = timers if (timed and i % EVENT_EVERY == 0) else None
            with torch.cuda.stream(streams[i % nstream]):
                st.run()

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, timed=True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    if rank == 0:
        per_op = {}
        for name, s, e in timers:
            per_op.setdefault(name, []).append(s.elapsed_time(e))
        per_op_ms = {k: float(np.mean(v)) for k, v in per_op.items()}
        alg = algorithmic_bytes()
        dom = "group_gather_feat_k128"
        achieved = alg[dom] / (per_op_ms[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "neighbor_queries_per_sec",
            "value": QUERIES_PER_STEP * args.steps * world / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "pointnet2_msg_fp_s3dis", "B_per_gpu": B, "N": N, "M": M,
                       "scales": [list(s) for s in SCALES], "C_group": C_FEAT, "C_interp": C_INTERP,
                       "sharding": f"batch x{world} (independent clouds, no data-path collective)",
                       "batches_in_flight": nstream,
                       "ball_query": "one launch for the 3 MSG scales" if args.fused_msg else "one launch per scale"},
            "fps_points_per_sec": B * N * world / (per_op_ms["fps"] * 1e-3),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes": alg[dom], "avg_ms": per_op_ms[dom],
                         "launches_timed": len(per_op[dom])},
            "per_op_ms": per_op_ms,
            "per_op_alg_GBs": {k: alg[k] / (per_op_ms[k] * 1e-3) / 1e9 for k in alg if k in per_op_ms},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(1234)
        elif world == 1:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
