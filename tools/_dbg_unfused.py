import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import smfft_amd as sm
for n in (64, 128, 256, 512, 1024, 2048, 4096):
    nffts = 100 * 40 + 7
    rng = np.random.default_rng(n + 77)
    x = ((rng.random((nffts, n), dtype=np.float32) - 0.5) + 1j * (rng.random((nffts, n), dtype=np.float32) - 0.5)).astype(np.complex64)
    for k in (1, 2, 3):
        sm.lib.smfft_set_nreuses(k)
        fused = sm.c2c(x, False, True, path="multiple")
        unfused = sm.c2c(x, False, True, path="multiple_unfused")
        d = fused.view(np.uint32) != unfused.view(np.uint32)
        print(n, k, "differing words", int(d.sum()), "rows", np.unique(np.nonzero(d)[0])[:10], "cols", np.unique(np.nonzero(d)[1])[:16])
sm.lib.smfft_set_nreuses(0)
