"""Where the bench step's wall time goes: host issue time per step (eager, with / without per-op events) against the
GPU time, and the same step replayed from a captured HIP graph, 1-3 streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
inp = bench.make_inputs(1234, dev)
for nstream in (1, 2, 3):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
    steps = []
    for s in streams:
        with torch.cuda.stream(s):
            steps.append(bench.Step(inp, dev))
    torch.cuda.synchronize()

    def run(n, fn):
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % nstream]):
                fn(i % nstream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3

    run(10, lambda j: steps[j].run())
    torch.cuda.synchronize()
    issue, total = run(100, lambda j: steps[j].run())
    print(f"streams {nstream} eager no events : host issue {issue:.3f} ms/step, total {total:.3f} ms/step", flush=True)
    timers = []
    for st in steps:
        st.timers = timers
    issue, total = run(100, lambda j: steps[j].run())
    print(f"streams {nstream} eager + events  : host issue {issue:.3f} ms/step, total {total:.3f} ms/step", flush=True)
    for st in steps:
        st.timers = None
    graphs = []
    for s, st in zip(streams, steps):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            st.run()
        graphs.append(g)
    torch.cuda.synchronize()
    run(10, lambda j: graphs[j].replay())
    issue, total = run(100, lambda j: graphs[j].replay())
    print(f"streams {nstream} graph replay    : host issue {issue:.3f} ms/step, total {total:.3f} ms/step", flush=True)
