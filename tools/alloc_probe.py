"""Does the allocator matter?  N=1024 external kernel on torch-allocated vs hipMalloc'ed 4 GiB buffers,
same process, same data, back-to-back launches between one event pair."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm
n, nffts = 1024, 524288
dev = torch.device("cuda", 0)
t_in = torch.rand((nffts, n, 2), dtype=torch.float32, device=dev)
t_out = torch.empty_like(t_in)
h_in, h_out = sm.DeviceBuffer(t_in.numel() * 4), sm.DeviceBuffer(t_in.numel() * 4)
sm.lib.smfft_memcpy_d2d(h_in.ptr, t_in.data_ptr(), t_in.numel() * 4)
print(f"torch in {t_in.data_ptr():#x} out {t_out.data_ptr():#x} | hipMalloc in {h_in.ptr:#x} out {h_out.ptr:#x}")
print("allocator settings:", os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("PYTORCH_CUDA_ALLOC_CONF"))
def run(i, o, label):
    for _ in range(5):
        sm.launch("ct", "external", i, o, n, nffts)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40):
            sm.launch("ct", "external", i, o, n, nffts)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 40 * 1e3)
    print(f"{label:28s} " + "  ".join(f"{r:.4f} ms ({2*n*nffts*8/r/1e6:.0f} GB/s)" for r in res), flush=True)
for _ in range(2):
    run(t_in.data_ptr(), t_out.data_ptr(), "torch -> torch")
    run(h_in.ptr, h_out.ptr, "hipMalloc -> hipMalloc")
    run(t_in.data_ptr(), h_out.ptr, "torch -> hipMalloc")
    run(h_in.ptr, t_out.data_ptr(), "hipMalloc -> torch")
