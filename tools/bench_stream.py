"""HBM calibration on this box: device copy (read+write) and read-only reduction at several sizes."""
import torch
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for mb in (84, 252, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device="cuda", dtype=torch.float16); y = torch.empty_like(x)
    ms = timeit(lambda: y.copy_(x))
    ms2 = timeit(lambda: x.sum())
    print(f"{mb:5d} MB: copy {ms*1e3:8.1f} us = {2*mb*1.048576/ms:6.2f} GB/ms (r+w)   sum {ms2*1e3:8.1f} us = {mb*1.048576/ms2:6.2f} GB/ms (read)")
