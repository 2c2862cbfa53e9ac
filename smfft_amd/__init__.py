"""smfft_amd -- MI355X-native shared-memory FFT (drop-in for the hot path of KAdamek/SMFFT).

The product is smfft_amd/libsmfft_amd.so (hand-written HIP for gfx950 + the C ABI of
include/smfft.h).  This package is only the thin Python host-side mirror of that ABI; it never
falls back to a CPU implementation: importing `smfft_amd.api` without the built library raises.
"""
from .api import (  # noqa: F401
    DeviceBuffer,
    FFT_external_benchmark,
    FFT_init,
    FFT_multiple_benchmark,
    LIB_PATH,
    NREUSES,
    c2c,
    c2r,
    host_transform,
    last_pair_info,
    launch,
    lib,
    pinned_empty,
    r2c,
    stockham_c2c,
)
