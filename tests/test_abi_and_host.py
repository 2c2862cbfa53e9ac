"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol
include/smfft.h declares plus the reference's own C++-linkage names; host-side logic."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def built(built_product):
    return built_product


def _declared_c_functions():
    text = open(os.path.join(ROOT, "include", "smfft.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smfft_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(built):
    lib = ctypes.CDLL(built)
    names = _declared_c_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/smfft.h but not exported"


def test_python_mirror_matches_header(built):
    import smfft_amd.api as api
    assert sorted(api.EXPORTED_C_SYMBOLS) == _declared_c_functions()
    assert api.lib.smfft_version().startswith(b"smfft_amd")
    assert api.lib.smfft_get_nreuses() == 100 == api.NREUSES


def test_reference_cxx_symbols_are_exported(built):
    """The reference's FFT.c binds these by C++ linkage (CT/FFT.c:80-81, ST/FFT.c:79-81,
    RC/FFT.c:188-191 + the L2 functions of the .cu files)."""
    import smfft_amd.api as api
    out = subprocess.check_output(["nm", "-D", "--defined-only", built], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for sym in api.EXPORTED_CXX_SYMBOLS:
        assert sym in exported, sym
    demangled = subprocess.check_output(["c++filt"] + list(api.EXPORTED_CXX_SYMBOLS), text=True).splitlines()
    assert "FFT_init()" in demangled
    assert any(d.startswith("FFT_external_benchmark(HIP_vector_type<float, 2u>*, HIP_vector_type<float, 2u>*, int, int, bool, bool, double*)") for d in demangled)
    assert any(d.startswith("GPU_smFFT_4elements(") for d in demangled)


def test_no_oracle_in_product(built):
    """The product library must not link or load anything from oracle/."""
    needed = subprocess.check_output(["readelf", "-d", built], text=True)
    assert "oracle" not in needed
    for root, _, files in os.walk(os.path.join(ROOT, "smfft_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "liboracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_missing_library_fails_loudly(tmp_path):
    code = "import os; os.environ['SMFFT_AMD_LIB']='/nonexistent/libsmfft_amd.so'; import smfft_amd"
    p = subprocess.run(["python", "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert p.returncode != 0 and "no CPU fallback" in p.stderr


def test_parameter_classes_match_reference_names():
    hdr = open(os.path.join(ROOT, "smfft_amd", "csrc", "SM_FFT_parameters.hpp")).read()
    for n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        for suf in ("forward", "forward_noreorder", "inverse", "inverse_noreorder"):
            assert f"class FFT_{n}_{suf} " in hdr
    for member in ("fft_exp", "fft_sm_required", "fft_length", "fft_length_quarter", "fft_length_half",
                   "fft_length_three_quarters", "fft_direction", "fft_reorder", "warp"):
        assert member in hdr


def test_shard_range_tiles_batch():
    from smfft_amd.sharding import shard_range
    for nffts in (0, 1, 7, 8, 524288, 524289, 4194304 + 5):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            for r in range(world):
                first, count = shard_range(nffts, r, world)
                assert first == pos and count in (nffts // world, nffts // world + 1)
                pos += count
            assert pos == nffts
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)
