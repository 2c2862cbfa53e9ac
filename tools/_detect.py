import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import smfft_amd as sm
pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
assert sm.lib.smfft_malloc_pair(1 << 32, ctypes.byref(pa), ctypes.byref(pb)) == 0
i = sm.last_pair_info()
print("DETECT", "uniform" if (i["classification"] == 0 and i["mixed_bytes"] == 0 and i["interleaved_bytes"] == 0) else "ordinary", i["candidates"], i["search_ms"])
