"""smfft_malloc_pair with SMFFT_PAIR_DEBUG=1: the scan's per-chunk write times and class probes, what it chose, how long it
took, and the config-2 kernel time on the pair -- for the default policy, with only interleaving allowed
(SMFFT_PAIR_NO_MIXED=1), with only mixed chunks allowed (SMFFT_PAIR_NO_INTERLEAVE=1), and on two plain allocations."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["SMFFT_PAIR_DEBUG"] = "1"
import smfft_amd as sm  # noqa: E402


def kernel_ms(a, b, n=1024, nffts=524288):
    ts = []
    for _ in range(9):
        t = ctypes.c_double(0)
        sm.lib.smfft_ct_external_benchmark(a, b, n, nffts, 0, 1, ctypes.byref(t))
        ts.append(t.value)
    return sorted(ts[2:])[len(ts[2:]) // 2]


for label, env in (("default", {}), ("interleave only", {"SMFFT_PAIR_NO_MIXED": "1"}), ("mixed only", {"SMFFT_PAIR_NO_INTERLEAVE": "1"}),
                   ("plain", {"SMFFT_PAIR_POLICY": "plain"})):
    for k in ("SMFFT_PAIR_NO_MIXED", "SMFFT_PAIR_NO_INTERLEAVE", "SMFFT_PAIR_POLICY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for nbytes in (4 << 30, 1 << 30):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        t0 = time.time()
        assert sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(a), ctypes.byref(b)) == 0
        took = time.time() - t0
        info = sm.last_pair_info()
        ms = kernel_ms(a.value, b.value, 1024, nbytes // 8192)
        print(f"[{label}] {nbytes >> 30} GiB: {took * 1e3:.0f} ms, N=1024 external {ms:.4f} ms = {2 * nbytes / ms / 8e9:.3f} of peak, {info}", flush=True)
        sm.lib.smfft_free_pair(a.value)
