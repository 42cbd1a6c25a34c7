#!/usr/bin/env python3
"""GPU-box probe: host-to-device copy rate of one feeder batch (735 MB) from (a) torch pinned memory, (b) a shared-memory tensor
registered with cudaHostRegister (what the feeder's ring is), (c) pageable memory."""
import time, torch
n = 735 << 20
dev = torch.device("cuda:0")
dst = torch.empty(n, dtype=torch.uint8, device=dev)
def rate(src, reps=8):
    torch.cuda.synchronize()
    for _ in range(2): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return n / dt / 1e9
a = torch.empty(n, dtype=torch.uint8).pin_memory(); a.fill_(1)
print("pinned (torch):            %.1f GB/s" % rate(a))
b = torch.empty(n, dtype=torch.uint8).share_memory_(); b.fill_(1)
rt = torch.cuda.cudart()
rc = int(rt.cudaHostRegister(b.data_ptr(), b.numel(), 0))
print("shared memory, registered: %.1f GB/s (register rc %d)" % (rate(b), rc))
for flag, nm in ((1, "portable"), (2, "mapped")):
    c = torch.empty(n, dtype=torch.uint8).share_memory_(); c.fill_(1)
    rc = int(rt.cudaHostRegister(c.data_ptr(), c.numel(), flag))
    print("shared memory, registered (%s): %.1f GB/s (rc %d)" % (nm, rate(c), rc))
p = torch.empty(n, dtype=torch.uint8); p.fill_(1)
print("pageable:                  %.1f GB/s" % rate(p, 3))
