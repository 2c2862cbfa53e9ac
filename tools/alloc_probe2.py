"""Which buffers are slow to write?  Six hipMalloc'ed 4 GiB buffers (allocated after a torch 4 GiB
block, like bench.py), every one used as output (input = the previous one) and as input."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n, nffts = 1024, 524288
nbytes = n * nffts * 8
dev = torch.device("cuda", 0)
t_in = torch.rand((nffts, n, 2), dtype=torch.float32, device=dev)
bufs = [sm.DeviceBuffer(nbytes) for _ in range(6)]
for b in bufs:
    sm.lib.smfft_memcpy_d2d(b.ptr, t_in.data_ptr(), nbytes)
if os.environ.get("FREE_TORCH"):
    del t_in
    torch.cuda.empty_cache()
print(" ".join(f"{b.ptr:#x}" for b in bufs))
def run(i, o):
    for _ in range(3):
        sm.launch("ct", "external", i, o, n, nffts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        sm.launch("ct", "external", i, o, n, nffts)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e3
for rep in range(2):
    print("out=k, in=k-1 :", "  ".join(f"{k}:{run(bufs[k-1].ptr, bufs[k].ptr):.4f}" for k in range(6)))
    print("in=k, out=k-1 :", "  ".join(f"{k}:{run(bufs[k].ptr, bufs[k-1].ptr):.4f}" for k in range(6)))
