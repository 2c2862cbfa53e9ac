"""ctypes mirror of include/smfft.h.

Host-side mirror of the reference's launch API (same names, argument meaning and error
behaviour as SMFFT_CooleyTukey_C2C/FFT-GPU-32bit.cu:576-752 and its Stockham / R2C siblings):
device pointers are plain integers, transforms are out of place, timings are ADDED to a running
total exactly like `*FFT_time += timer.Elapsed()` upstream.

There is deliberately no CPU fallback here: if the HIP library is missing the import fails.
"""
import ctypes
import os

import numpy as np

NREUSES = 100
LIB_PATH = os.environ.get("SMFFT_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsmfft_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C smfft_amd/csrc`).  smfft_amd has no CPU fallback."
    )

lib = ctypes.CDLL(LIB_PATH)

_vp, _i, _dp = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
_ull = ctypes.c_ulonglong
_SIGS = {
    "smfft_init": (None, []),
    "smfft_ct_external_benchmark": (_i, [_vp, _vp, _i, _i, _i, _i, _dp]),
    "smfft_ct_multiple_benchmark": (_i, [_vp, _vp, _i, _i, _i, _i, _dp]),
    "smfft_st_external_benchmark": (_i, [_vp, _vp, _i, _i, _dp]),
    "smfft_st_multiple_benchmark": (_i, [_vp, _vp, _i, _i, _dp]),
    "smfft_rc_external_benchmark": (_i, [_vp, _vp, _i, _i, _i, _dp]),
    "smfft_rc_multiple_benchmark": (_i, [_vp, _vp, _i, _i, _dp]),
    "smfft_launch": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "smfft_graph_create": (_vp, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "smfft_graph_launch": (_i, [_vp, _vp]),
    "smfft_graph_destroy": (_i, [_vp]),
    "smfft_copy_launch": (_i, [_vp, _vp, ctypes.c_longlong, _vp]),
    "smfft_gpu_ct": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _dp, _dp]),
    "smfft_gpu_st": (_i, [_vp, _vp, _i, _i, _i, _dp, _dp]),
    "smfft_gpu_r2c": (_i, [_vp, _vp, _i, _i, _i]),
    "smfft_gpu_c2r": (_i, [_vp, _vp, _i, _i, _i]),
    "smfft_set_grid_cap": (None, [_i]),
    "smfft_get_grid_cap": (_i, []),
    "smfft_set_nreuses": (None, [_i]),
    "smfft_get_nreuses": (_i, []),
    "smfft_device_count": (_i, []),
    "smfft_set_device": (_i, [_i]),
    "smfft_version": (ctypes.c_char_p, []),
    "smfft_malloc": (_vp, [_ull]),
    "smfft_malloc_pair": (_i, [_ull, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "smfft_free_pair": (_i, [_vp]),
    "smfft_free": (_i, [_vp]),
    "smfft_memcpy_h2d": (_i, [_vp, _vp, _ull]),
    "smfft_memcpy_d2h": (_i, [_vp, _vp, _ull]),
    "smfft_memcpy_d2d": (_i, [_vp, _vp, _ull]),
    "smfft_memset": (_i, [_vp, _i, _ull]),
    "smfft_synchronize": (_i, []),
}
for _name, (_res, _args) in _SIGS.items():
    _f = getattr(lib, _name)  # AttributeError here = the library does not export what smfft.h declares
    _f.restype = _res
    _f.argtypes = _args

EXPORTED_C_SYMBOLS = tuple(_SIGS)
# the reference's own C++-linkage symbols (include/smfft_reference_api.h), Itanium-mangled
EXPORTED_CXX_SYMBOLS = (
    "_Z8FFT_initv",
    "_Z22FFT_external_benchmarkP15HIP_vector_typeIfLj2EES1_iibbPd",
    "_Z22FFT_multiple_benchmarkP15HIP_vector_typeIfLj2EES1_iibbPd",
    "_Z22FFT_external_benchmarkP15HIP_vector_typeIfLj2EES1_iiPd",
    "_Z22FFT_multiple_benchmarkP15HIP_vector_typeIfLj2EES1_iiPd",
    "_Z22FFT_external_benchmarkPfS_iiiPd",
    "_Z22FFT_multiple_benchmarkPfS_iiPd",
    "_Z19GPU_smFFT_4elementsP15HIP_vector_typeIfLj2EES1_iibbiPdS2_",
    "_Z20GPU_FFT_C2C_StockhamP15HIP_vector_typeIfLj2EES1_iiiPdS2_",
    "_Z13GPU_smFFT_R2CP15HIP_vector_typeIfLj2EEPfiii",
    "_Z13GPU_smFFT_C2RPfP15HIP_vector_typeIfLj2EEiii",
)


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with HIP error {rc}")


class DeviceBuffer:
    """Owning handle on device memory obtained through the library's own allocator."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib.smfft_malloc(self.nbytes) if self.nbytes else None
        if self.nbytes and not self.ptr:
            raise MemoryError(f"smfft_malloc({self.nbytes}) failed")

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(max(arr.nbytes, 8))
        if arr.nbytes:
            _ck(lib.smfft_memcpy_h2d(buf.ptr, arr.ctypes.data, arr.nbytes), "memcpy_h2d")
        return buf

    def to_host(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            _ck(lib.smfft_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes), "memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib.smfft_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- the reference's L2 API, mirrored (device pointers = ints) -------------------------------------
def FFT_init():
    lib.smfft_init()


def FFT_external_benchmark(d_input, d_output, FFT_size, nFFTs, inverse=False, reorder=True, family="ct"):
    """One timed launch; returns (status, elapsed_ms).  family: 'ct' | 'st' | 'rc'."""
    t = ctypes.c_double(0.0)
    if family == "ct":
        rc = lib.smfft_ct_external_benchmark(d_input, d_output, FFT_size, nFFTs, int(inverse), int(reorder), ctypes.byref(t))
    elif family == "st":
        rc = lib.smfft_st_external_benchmark(d_input, d_output, FFT_size, nFFTs, ctypes.byref(t))
    else:
        rc = lib.smfft_rc_external_benchmark(d_input, d_output, FFT_size, nFFTs, int(inverse), ctypes.byref(t))
    return rc, t.value


def FFT_multiple_benchmark(d_input, d_output, FFT_size, nFFTs, inverse=False, reorder=True, family="ct"):
    t = ctypes.c_double(0.0)
    if family == "ct":
        rc = lib.smfft_ct_multiple_benchmark(d_input, d_output, FFT_size, nFFTs, int(inverse), int(reorder), ctypes.byref(t))
    elif family == "st":
        rc = lib.smfft_st_multiple_benchmark(d_input, d_output, FFT_size, nFFTs, ctypes.byref(t))
    else:
        rc = lib.smfft_rc_multiple_benchmark(d_input, d_output, FFT_size, nFFTs, ctypes.byref(t))
    return rc, t.value


_FAMILY = {"ct": 0, "st": 1, "rc": 2}


def launch(family, path, d_input, d_output, FFT_size, nFFTs, inverse=False, reorder=True, stream=0):
    """Launch-only form (no events, no sync) on a hipStream_t handle (int; 0 = null stream)."""
    rc = lib.smfft_launch(_FAMILY[family], 0 if path == "external" else 1, d_input, d_output, FFT_size, nFFTs, int(inverse), int(reorder), stream)
    if rc != 0:
        raise RuntimeError(f"smfft_launch({family},{path},N={FFT_size}) -> {rc}")


# ---- NumPy-level conveniences used by the tests (host arrays in/out, still the HIP path) -----------
def _run(x, out_dtype, out_shape, fn):
    din = DeviceBuffer.from_host(x)
    dout = DeviceBuffer(max(int(np.prod(out_shape)) * np.dtype(out_dtype).itemsize, 8))
    lib.smfft_memset(dout.ptr, 0xFF, dout.nbytes)   # NaN pattern: untouched outputs are caught
    rc, ms = fn(din.ptr, dout.ptr)
    if rc != 0:
        raise RuntimeError(f"benchmark call returned {rc}")
    return dout.to_host(out_dtype, out_shape)


def c2c(x, inverse=False, reorder=True, path="external"):
    """x: (nFFTs, N) complex64 host array -> CT-family result through the HIP library."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    nffts, n = x.shape
    f = FFT_external_benchmark if path == "external" else FFT_multiple_benchmark
    return _run(x, np.complex64, x.shape, lambda i, o: f(i, o, n, nffts, inverse, reorder, "ct"))


def stockham_c2c(x):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    nffts, n = x.shape
    return _run(x, np.complex64, x.shape, lambda i, o: FFT_external_benchmark(i, o, n, nffts, family="st"))


def r2c(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    nffts, n = x.shape
    return _run(x, np.complex64, (nffts, n // 2), lambda i, o: FFT_external_benchmark(i, o, n, nffts, inverse=False, family="rc"))


def c2r(xp):
    xp = np.ascontiguousarray(xp, dtype=np.complex64)
    nffts, half = xp.shape
    return _run(xp, np.float32, (nffts, 2 * half), lambda i, o: FFT_external_benchmark(i, o, 2 * half, nffts, inverse=True, family="rc"))
