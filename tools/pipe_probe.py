#!/usr/bin/env python3
"""(GPU box) Where the pipelined single-batch step of bench.py spends its time: FPS in one launch and in pieces, the
consumers alone, the tail alone, the whole schedule.   python tools/pipe_probe.py [cut cut ...]   (default 512)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vision3d_amd import ext  # noqa: E402

inner = [int(a) for a in sys.argv[1:]] or [512]
dev = torch.device("cuda", 0)
inp = bench.make_inputs(1234, dev)
ps = bench.PipelinedStep(inp, dev, cuts=[0] + inner + [bench.M])
pieces, cuts = ps.pieces, ps.cuts
B, N, M = bench.B, bench.N, bench.M


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def fps_one():
    ps.fps_dist.fill_(1e10)
    ext.furthest_point_sample(ps.pts, ps.fps_dist, ps.fps_idx, M)


def fps_pieces():
    ps.fps_dist.fill_(1e10)
    for p in range(pieces):
        ext.furthest_point_sample_rounds(ps.pts, ps.fps_dist, ps.fps_idx, M, cuts[p], cuts[p + 1])


def consumers():
    for x in ps.ball_flat:
        x.zero_()
    for p in range(pieces):
        idx = ps.fps_idx[:, cuts[p]:cuts[p + 1]]
        q = torch.gather(ps.pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
        ext.ball_query_multi(q, ps.pts, [x[p] for x in ps.ball_idx], ps.radii, ps.ks)
        for i in range(len(bench.SCALES)):
            ext.group_gather_forward(ps.pts_cf, ps.ball_idx[i][p], ps.g_xyz[p][i])
            ext.group_gather_forward(inp["feats"], ps.ball_idx[i][p], ps.g_feat[p][i])


def tail():
    q_all = torch.gather(ps.pts, 1, ps.fps_idx.unsqueeze(-1).expand(-1, -1, 3))
    ext.three_nn(ps.pts, q_all, ps.tnn_d, ps.tnn_i)
    w = 1.0 / (ps.tnn_d + 1e-5)
    w = w / w.sum(dim=2, keepdim=True)
    ext.three_interpolate_forward(inp["sfeats"], ps.tnn_i, w, ps.interp)
    ext.three_interpolate_backward(inp["grad"], ps.tnn_i, w, ps.sgrad, M)


print(f"cuts {cuts}")
print(f"fps, one launch      {timeit(fps_one):.3f} ms")
print(f"fps, {pieces} launches      {timeit(fps_pieces):.3f} ms")
print(f"consumers alone      {timeit(consumers):.3f} ms")
print(f"tail alone           {timeit(tail):.3f} ms")
print(f"pipelined step       {timeit(ps.run):.3f} ms")
if os.environ.get("V3D_PIPE_GRAPH"):  # the same schedule captured in a HIP graph: measured slower (1.45 against 1.02 ms)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ps.run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ps.run()
    print(f"... as a HIP graph   {timeit(g.replay):.3f} ms")

# timeline of one step: when each piece's sampling / grouping ends (ms after the start of the step)
if os.environ.get("V3D_PIPE_TIMELINE", "1") == "1":
    E = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    main = torch.cuda.current_stream()
    t_start, t_fps, t_grp, t_tail = E(), [E() for _ in range(pieces)], [E() for _ in range(pieces)], E()
    torch.cuda.synchronize()
    t_start.record(main)
    ps.start.record(main)
    ps.fps_dist.fill_(1e10)
    for p in range(pieces):
        ext.furthest_point_sample_rounds(ps.pts, ps.fps_dist, ps.fps_idx, M, cuts[p], cuts[p + 1])
        ps.sampled[p].record(main)
        t_fps[p].record(main)
    with torch.cuda.stream(ps.side):
        ps.side.wait_event(ps.start)
        for x in ps.ball_flat:
            x.zero_()
        for p in range(pieces):
            ps.side.wait_event(ps.sampled[p])
            idx = ps.fps_idx[:, cuts[p]:cuts[p + 1]]
            q = torch.gather(ps.pts, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
            ext.ball_query_multi(q, ps.pts, [x[p] for x in ps.ball_idx], ps.radii, ps.ks)
            for i in range(len(bench.SCALES)):
                ext.group_gather_forward(ps.pts_cf, ps.ball_idx[i][p], ps.g_xyz[p][i])
                ext.group_gather_forward(inp["feats"], ps.ball_idx[i][p], ps.g_feat[p][i])
            t_grp[p].record(ps.side)
    tail()
    t_tail.record(main)
    torch.cuda.synchronize()
    print("sampling ends   ", " ".join(f"{t_start.elapsed_time(e):.3f}" for e in t_fps))
    print("grouping ends   ", " ".join(f"{t_start.elapsed_time(e):.3f}" for e in t_grp))
    print(f"tail ends        {t_start.elapsed_time(t_tail):.3f}")
