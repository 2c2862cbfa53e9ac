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
    "smfft_ct_multiple_unfused_benchmark": (_i, [_vp, _vp, _i, _i, _i, _dp]),
    "smfft_ct_multiple_percall_benchmark": (_i, [_vp, _vp, _i, _i, _i, _i, _dp]),
    "smfft_st_external_benchmark": (_i, [_vp, _vp, _i, _i, _dp]),
    "smfft_st_external_benchmark_dir": (_i, [_vp, _vp, _i, _i, _i, _dp]),
    "smfft_st_multiple_benchmark": (_i, [_vp, _vp, _i, _i, _dp]),
    "smfft_host_transform": (_i, [_i, _vp, _vp, _i, ctypes.c_longlong, _i, _i, ctypes.c_longlong, _i, _dp]),
    "smfft_host_malloc": (_vp, [ctypes.c_ulonglong]),
    "smfft_host_free": (_i, [_vp]),
    "smfft_host_pipeline_release": (None, []),
    "smfft_rc_external_benchmark": (_i, [_vp, _vp, _i, _i, _i, _dp]),
    "smfft_rc_multiple_benchmark": (_i, [_vp, _vp, _i, _i, _dp]),
    "smfft_launch": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "smfft_copy_launch": (_i, [_vp, _vp, ctypes.c_longlong, _vp]),
    "smfft_gpu_ct": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _dp, _dp]),
    "smfft_gpu_st": (_i, [_vp, _vp, _i, _i, _i, _dp, _dp]),
    "smfft_gpu_r2c": (_i, [_vp, _vp, _i, _i, _i]),
    "smfft_gpu_c2r": (_i, [_vp, _vp, _i, _i, _i]),
    "smfft_set_grid_cap": (None, [_i]),
    "smfft_get_grid_cap": (_i, []),
    "smfft_set_nreuses": (None, [_i]),
    "smfft_set_pacing": (None, [_i]),
    "smfft_pacing_for_output": (_i, [_vp, _i, _i]),
    "smfft_measure_multiple_residency": (_i, [_i, _i, _i, _i, _i, ctypes.POINTER(_i)]),
    "smfft_set_multiple_balance": (None, [_i]),
    "smfft_get_multiple_balance": (_i, []),
    "smfft_set_multiple_rotation": (None, [_i]),
    "smfft_get_multiple_rotation": (_i, []),
    "smfft_set_handoff_wait_us": (None, [_i]),
    "smfft_debug_delay_parking": (None, [_i, _i, _i]),
    "smfft_schedule_buffers": (_i, [ctypes.POINTER(ctypes.c_int)]),
    "smfft_va_window": (_i, [ctypes.POINTER(_ull), ctypes.POINTER(_ull)]),
    "smfft_get_nreuses": (_i, []),
    "smfft_device_count": (_i, []),
    "smfft_set_device": (_i, [_i]),
    "smfft_version": (ctypes.c_char_p, []),
    "smfft_malloc": (_vp, [_ull]),
    "smfft_malloc_pair": (_i, [_ull, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "smfft_malloc_pair_for_wrapper": (_i, [_ull, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "smfft_malloc_pair_budget": (_i, [_ull, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.c_double, ctypes.c_double]),
    "smfft_free_pair": (_i, [_vp]),
    "smfft_malloc_written": (_i, [_ull, ctypes.POINTER(_vp)]),
    "smfft_malloc_written_for": (_i, [_vp, _ull, ctypes.POINTER(_vp)]),
    "smfft_free_written": (_i, [_vp]),
    "smfft_pair_cache_release": (_i, []),
    "smfft_last_pair_info": (_i, [_vp]),
    "smfft_free": (_i, [_vp]),
    "smfft_memcpy_h2d": (_i, [_vp, _vp, _ull]),
    "smfft_memcpy_d2h": (_i, [_vp, _vp, _ull]),
    "smfft_memcpy_d2d": (_i, [_vp, _vp, _ull]),
    "smfft_memset": (_i, [_vp, _i, _ull]),
    "smfft_synchronize": (_i, []),
    "smfft_mem_info": (_i, [ctypes.POINTER(_ull), ctypes.POINTER(_ull)]),
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


class SmfftPairInfo(ctypes.Structure):
    """mirror of include/smfft.h SmfftPairInfo"""
    _fields_ = [("bytes", ctypes.c_ulonglong), ("candidate_bytes", ctypes.c_ulonglong), ("candidates", ctypes.c_int), ("chosen", ctypes.c_int),
                ("good_enough", ctypes.c_int), ("read_ms", ctypes.c_float), ("copy_ms", ctypes.c_float), ("first_copy_ms", ctypes.c_float),
                ("search_ms", ctypes.c_double), ("mixed_bytes", ctypes.c_ulonglong), ("interleaved_bytes", ctypes.c_ulonglong),
                ("first_ordinary_copy_ms", ctypes.c_float), ("classification", ctypes.c_int)]


def last_pair_info():
    info = SmfftPairInfo()
    lib.smfft_last_pair_info(ctypes.byref(info))
    return {name: getattr(info, name) for name, _ in SmfftPairInfo._fields_}


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


def launch(family, path, d_input, d_output, FFT_size, nFFTs, inverse=None, reorder=True, stream=0):
    """Launch-only form (no events, no sync) on a hipStream_t handle (int; 0 = null stream).
    inverse=None: the program's own direction (Stockham: inverse, ST:76; otherwise forward)."""
    if inverse is None:
        inverse = (family == "st")
    rc = lib.smfft_launch(_FAMILY[family], {"external": 0, "multiple": 1, "multiple_unfused": 2}[path], d_input, d_output, FFT_size, nFFTs, int(inverse), int(reorder), stream)
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
    if path == "multiple_unfused":      # launch-only entry point (no *_benchmark call exists for it)
        def fn(i, o):
            launch("ct", path, i, o, n, nffts, inverse, reorder)
            return lib.smfft_synchronize(), 0.0
        return _run(x, np.complex64, x.shape, fn)
    f = FFT_external_benchmark if path == "external" else FFT_multiple_benchmark
    return _run(x, np.complex64, x.shape, lambda i, o: f(i, o, n, nffts, inverse, reorder, "ct"))


def stockham_c2c(x, inverse=True):
    """Stockham program: un-normalised inverse (+i) transform as upstream (ST:76); inverse=False is the forward
    extension (smfft_st_external_benchmark_dir)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    nffts, n = x.shape

    def fn(i, o):
        t = ctypes.c_double(0.0)
        return lib.smfft_st_external_benchmark_dir(i, o, n, nffts, int(inverse), ctypes.byref(t)), t.value
    return _run(x, np.complex64, x.shape, fn)


def r2c(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    nffts, n = x.shape
    return _run(x, np.complex64, (nffts, n // 2), lambda i, o: FFT_external_benchmark(i, o, n, nffts, inverse=False, family="rc"))


def c2r(xp):
    xp = np.ascontiguousarray(xp, dtype=np.complex64)
    nffts, half = xp.shape
    return _run(xp, np.float32, (nffts, 2 * half), lambda i, o: FFT_external_benchmark(i, o, 2 * half, nffts, inverse=True, family="rc"))


# ---- host-resident batches (smfft_host_transform) ----------------------------------------------------
def pinned_empty(shape, dtype):
    """NumPy array over pinned host memory from smfft_host_malloc (kept alive by the array's base object)."""
    dtype = np.dtype(dtype)
    nbytes = max(int(np.prod(shape)) * dtype.itemsize, 8)
    ptr = lib.smfft_host_malloc(nbytes)
    if not ptr:
        raise MemoryError(f"smfft_host_malloc({nbytes}) failed")

    class _Owner:
        def __init__(self, p):
            self.p = p

        def __del__(self):
            try:
                lib.smfft_host_free(self.p)
            except Exception:
                pass
    owner = _Owner(ptr)
    buf = (ctypes.c_char * nbytes).from_address(ptr)
    buf._owner = owner
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def host_transform(x, out=None, family="ct", inverse=None, reorder=True, slab_ffts=0, lanes=0):
    """x: (nFFTs, N) host array (complex64; float32 for R2C) -> (result, elapsed_ms) streamed through the GPU in
    slabs with H2D / FFT / D2H overlapped; x and out may be pageable or pinned (pinned_empty).
    inverse=None: the program's own direction -- the Stockham C2C program is the + sign (inverse) transform
    (ST:76), everything else defaults to forward; family="st" with inverse=False is the forward extension."""
    fam = _FAMILY[family]
    if inverse is None:
        inverse = (family == "st")
    nffts, width = x.shape
    if fam == 2:
        n = width if not inverse else 2 * width          # R2C: reals in; C2R: N/2 packed complex in
        in_dtype, out_dtype = (np.float32, np.complex64) if not inverse else (np.complex64, np.float32)
        out_shape = (nffts, n // 2) if not inverse else (nffts, n)
    else:
        n, in_dtype, out_dtype, out_shape = width, np.complex64, np.complex64, (nffts, width)
    if x.dtype != in_dtype or not x.flags.c_contiguous:
        x = np.ascontiguousarray(x, dtype=in_dtype)
    if out is None:
        out = np.empty(out_shape, dtype=out_dtype)
    assert out.dtype == out_dtype and out.shape == out_shape and out.flags.c_contiguous
    t = ctypes.c_double(0.0)
    rc = lib.smfft_host_transform(fam, x.ctypes.data, out.ctypes.data, n, nffts, int(inverse), int(reorder), int(slab_ffts), int(lanes), ctypes.byref(t))
    if rc != 0:
        raise RuntimeError(f"smfft_host_transform({family}, N={n}) -> {rc}")
    return out, t.value
