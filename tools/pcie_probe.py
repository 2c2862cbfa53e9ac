"""Raw PCIe ceiling on the GPU box (torch only as a HIP memcpy driver): pinned H2D alone, D2H alone, both at once."""
import time
import torch
n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(h2d, d2h, reps=4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1):
                d_a.copy_(h_in, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2):
                h_out.copy_(d_b, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for name, a, b in (("H2D alone", 1, 0), ("D2H alone", 0, 1), ("H2D + D2H concurrently", 1, 1)):
    run(a, b, 1)
    t = min(run(a, b) for _ in range(3))
    print(f"{name:26s}: {t*1e3:7.1f} ms per GiB-each  -> {(a + b) * n / t / 1e9:6.1f} GB/s total")
