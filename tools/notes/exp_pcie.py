#!/usr/bin/env python3
"""PCIe rates on the box: device -> pageable host, device -> pinned host, pinned -> pageable memcpy (what a chunked
pinned ring would have to beat for the host ABI's 537 MB of u64 labels at 8192^2)."""
import time
import numpy as np
import torch
n = 512 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda"); d.fill_(3)
page = torch.empty(n, dtype=torch.uint8); page.fill_(1)
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True); pin.fill_(1)
def t(f, k=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
a = t(lambda: page.copy_(d)); print(f"D2H pageable {n/a/1e9:.1f} GB/s")
b = t(lambda: pin.copy_(d, non_blocking=True)); print(f"D2H pinned   {n/b/1e9:.1f} GB/s")
c = t(lambda: page.copy_(pin)); print(f"pinned->pageable memcpy (torch, threads={torch.get_num_threads()}) {n/c/1e9:.1f} GB/s")
pn, gn = pin.numpy(), page.numpy()
e = t(lambda: np.copyto(gn, pn)); print(f"pinned->pageable memcpy (1 thread) {n/e/1e9:.1f} GB/s")
h = t(lambda: d.copy_(page)); print(f"H2D pageable {n/h/1e9:.1f} GB/s")
g = t(lambda: d.copy_(pin, non_blocking=True)); print(f"H2D pinned   {n/g/1e9:.1f} GB/s")
