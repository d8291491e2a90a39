"""PCIe probe: pinned host -> device DMA rate (hipMemcpyAsync through torch) for 32 MB in 1 / 8 / 40 pieces, and the same while a
compute kernel keeps every CU busy.  Decides whether DMA-streamed inputs can beat the kernel's own zero-copy reads (~24 GB/s)."""
import time
import torch
torch.cuda.init()
dev = torch.device("cuda", 0)
n = 32 * 1024 * 1024 // 8
h = torch.empty(n, dtype=torch.float64).pin_memory()
h.uniform_()
d = torch.empty(n, dtype=torch.float64, device=dev)
s2 = torch.cuda.Stream(dev)
def timed(pieces, busy=False, reps=10):
    xs = torch.randn(8192, 8192, device=dev)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        if busy:
            for _ in range(3): xs @ xs
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s2):
            e0.record()
            step = n // pieces
            for k in range(pieces):
                d[k * step:(k + 1) * step].copy_(h[k * step:(k + 1) * step], non_blocking=True)
            e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return 32.0 / 1024 / (best * 1e-3)
for p in (1, 8, 40):
    print("pieces", p, "idle GB/s %.1f" % timed(p), "busy GB/s %.1f" % timed(p, True))
t = time.perf_counter()
for k in range(40): d[:1024].copy_(h[:1024], non_blocking=True)
print("enqueue cost per async copy us %.1f" % ((time.perf_counter() - t) / 40 * 1e6))
torch.cuda.synchronize()
