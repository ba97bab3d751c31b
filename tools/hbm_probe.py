"""Write-only / copy HBM probe (torch fill_ and copy_) to calibrate what the gather's store stream can reach."""
import torch

def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[len(ts) // 2] * 1e-3

for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device="cuda", dtype=torch.float32); y = torch.empty_like(x)
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x))
    print(f"{mb} MiB fill {n*4/tf/1e9:.0f} GB/s  copy(r+w) {2*n*4/tc/1e9:.0f} GB/s", flush=True)
