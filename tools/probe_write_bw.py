#!/usr/bin/env python3
"""What a pure streaming WRITE reaches on this GPU: the practical ceiling for the
board Jacobian kernel, which writes 343 MB and reads 28 MB per launch (dev tool).
torch fill_ and copy_ kernels over the same footprint, timed with events"""
import torch
n = 343_200_000 // 8
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e-3
t = timeit(lambda: a.fill_(1.0));  print(f"fill  {n*8/1e6:.0f} MB: {t*1e6:.1f} us  {n*8/t/1e12:.2f} TB/s written")
t = timeit(lambda: a.zero_());     print(f"zero  {n*8/1e6:.0f} MB: {t*1e6:.1f} us  {n*8/t/1e12:.2f} TB/s written")
t = timeit(lambda: b.copy_(a));    print(f"copy  {n*8/1e6:.0f} MB: {t*1e6:.1f} us  {2*n*8/t/1e12:.2f} TB/s read+written")
