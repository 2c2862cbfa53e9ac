"""How strict is the reference's Compare_data metric (max_error = 1e-4, CT/FFT.c:23-77) at large N?
Counts its 'errors' for ours vs fp64, rocFFT fp32 (torch.fft) vs fp64 and ours vs rocFFT, U[0,1) data."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import smfft_amd as sm

def ref_metric(a, b):
    a, b = np.abs(a), np.abs(b)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    d = hi - lo
    big = lo > 10
    d[big] = d[big] / 10.0 ** np.floor(np.log10(lo[big]))
    return np.minimum(d, 10000.0)

def count(x, y):
    e = np.maximum(ref_metric(x.real, y.real), ref_metric(x.imag, y.imag))
    return int((e > 1e-4).sum()), float(e.max())

rng = np.random.default_rng(7)
for n in (1024, 2048, 4096):
    nffts = 4096
    x = (rng.random((nffts, n), dtype=np.float32) + 1j * rng.random((nffts, n), dtype=np.float32)).astype(np.complex64)
    for inv in (0, 1):
        ours = sm.c2c(x, bool(inv), True)
        xt = torch.from_numpy(x).cuda()
        f = torch.fft.ifft if inv else torch.fft.fft
        roc = f(xt, dim=-1, norm="forward" if inv else "backward").cpu().numpy()
        ref = f(xt.to(torch.complex128), dim=-1, norm="forward" if inv else "backward").cpu().numpy()
        print(f"N={n} inv={inv}: ours vs fp64 {count(ours, ref)}  rocFFT vs fp64 {count(roc, ref)}  ours vs rocFFT {count(ours, roc)}  "
              f"relL2 ours {np.linalg.norm(ours - ref) / np.linalg.norm(ref):.2e} rocFFT {np.linalg.norm(roc - ref) / np.linalg.norm(ref):.2e}", flush=True)
