"""Host cost of a 4 KB host-to-device copy, by way of issuing it (the AT per-sample loop copies two 512-vectors per sample)."""
import time

import torch

dev = torch.device("cuda:0")
ring = torch.empty((2, 2, 512)).pin_memory()
whole = torch.empty((2, 512)).pin_memory()
pageable = torch.empty((2, 512))
dbuf = torch.empty((2, 512), device=dev)
torch.cuda.synchronize()


def bench(name, fn, n=2000):
    for _ in range(20):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:55s} {(t1 - t0) / n * 1e6:7.1f} us issue")


bench("pageable.to(dev)", lambda i: pageable.to(dev))
bench("pinned view .to(dev, non_blocking)", lambda i: ring[i & 1].to(dev, non_blocking=True))
bench("pinned whole .to(dev, non_blocking)", lambda i: whole.to(dev, non_blocking=True))
bench("dbuf.copy_(pinned whole, non_blocking)", lambda i: dbuf.copy_(whole, non_blocking=True))
bench("dbuf.copy_(pinned view, non_blocking)", lambda i: dbuf.copy_(ring[i & 1], non_blocking=True))
bench("dbuf.copy_(pageable)", lambda i: dbuf.copy_(pageable))
bench("torch.empty((2,512), device)", lambda i: torch.empty((2, 512), device=dev))
bench("whole.is_pinned()", lambda i: whole.is_pinned())
