"""How long does a large hipMalloc take behind the release of large blocks?  (the C5 PCG leg's image allocations
take ~4 s inside bench.py, 0.3 s in a fresh process.)  Each variant releases three touched 40 GB blocks and then times two 40 GB
allocations: at once, behind a sleep, behind a small throw-away allocation, behind a throw-away allocation of the same size."""
import sys
import time

import torch

dev = torch.device("cuda:0")
GB = 1 << 30


def alloc2(tag):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = torch.empty(40 * GB, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    b = torch.empty(40 * GB, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    a.fill_(1)
    b.fill_(1)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"{tag:52s} alloc A {1e3 * (t1 - t0):8.1f} ms  alloc B {1e3 * (t2 - t1):8.1f} ms  first touch {1e3 * (t3 - t2):8.1f} ms", flush=True)
    del a, b
    torch.cuda.empty_cache()
    torch.cuda.synchronize()


def dirty(nblk=3):
    big = [torch.empty(40 * GB, dtype=torch.uint8, device=dev) for _ in range(nblk)]
    for t in big:
        t.fill_(3)
    torch.cuda.synchronize()
    del big, t
    t0 = time.perf_counter()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


torch.empty(1, device=dev)
alloc2("fresh process")
dirty()
alloc2("behind 120 GB released: at once")
time.sleep(10)
dirty()
time.sleep(10)
alloc2("behind 120 GB released: after 10 s of sleep")
time.sleep(10)
dirty()
s = torch.empty(GB, dtype=torch.uint8, device=dev)
del s
torch.cuda.empty_cache()
alloc2("behind 120 GB released: after a 1 GB throw-away")
time.sleep(10)
dirty()
t0 = time.perf_counter()
s = [torch.empty(40 * GB, dtype=torch.uint8, device=dev) for _ in range(3)]
torch.cuda.synchronize()
t1 = time.perf_counter()
del s
torch.cuda.empty_cache()
torch.cuda.synchronize()
print(f"  (throw-away 3 x 40 GB took {1e3 * (t1 - t0):.1f} ms)")
alloc2("behind 120 GB released: after a 120 GB throw-away")
time.sleep(10)
dirty(2)
alloc2("behind 80 GB released: at once")
time.sleep(10)
dirty(4)
alloc2("behind 160 GB released: at once")
