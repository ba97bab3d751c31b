"""Times the threshold / global top-k selectors against the torch formulation the reference uses (matching.py:64-121)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision3d_amd import ext

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

dev = "cuda:0"
CASES = ((512, 256), (2048, 256), (8192, 256), (8192, 5000), (16384, 256))
if len(sys.argv) > 2:
    CASES = ((int(sys.argv[1]), int(sys.argv[2])),)
for n, k in CASES:
    s = torch.randn((n, n), device=dev)
    idx = torch.empty((2, max(k, 1 << 20)), dtype=torch.long, device=dev)
    sc = torch.empty((max(k, 1 << 20),), device=dev)
    cnt = torch.zeros((), dtype=torch.long, device=dev)
    t_nat = timeit(lambda: ext.global_topk_select(s, k, out_row=idx[0], out_col=idx[1], out_score=sc))
    def ref_topk():
        v, i = s.view(-1).topk(k)
        return i // n, i % n, v
    t_ref = timeit(ref_topk)
    thr = 3.0
    t_thr = timeit(lambda: ext.threshold_select(s, thr, out_row=idx[0], out_col=idx[1], out_score=sc, num_out=cnt))
    def ref_thr():
        r, c = torch.nonzero(s > thr, as_tuple=True)
        return r, c, s[r, c]
    t_rthr = timeit(ref_thr)
    gb = 4 * n * n / 1e9
    print(f"{n}x{n} k={k}: global_topk {t_nat:8.1f} us (torch {t_ref:8.1f}) = {4*gb/t_nat*1e6/1e3:6.2f} TB/s over 4 reads | "
          f"threshold {t_thr:8.1f} us (torch {t_rthr:8.1f}) = {2*gb/t_thr*1e6/1e3:6.2f} TB/s over 2 reads", flush=True)
