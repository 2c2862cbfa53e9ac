"""Compile-time budgets of the hot kernels, read from hipcc's own reports (CPU only: hipcc cross-compiles gfx950).  What they pin
regresses silently otherwise: the in-LDS kernels live on four waves per SIMD (<= 128 VGPRs, no scratch in the application loop's
kernels measured here), the external kernel on its register count, and the reference-shaped N = 4096 `multiple` loop on its number
of workgroup barriers (DESIGN.md sections 2.4, 5.2-5.4)."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, os.path.join(ROOT, "tools"))
from inst_flags import part_flags  # noqa: E402  (the product's two objects per length and their flags: smfft_amd/csrc/Makefile)
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include")]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void ", "") for o in out]


def _resources(n):
    src = os.path.join(ROOT, "smfft_amd", "csrc", "smfft_inst.hip")
    report = ""
    for part in (1, 2):      # what ships: the external kernels' object and the in-LDS kernels' object, each with its flags
        p = subprocess.run([HIPCC] + FLAGS + part_flags(n, part) + [f"-DSMFFT_N={n}", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        report += p.stderr
    rows, cur = [], None
    for line in report.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    for r, name in zip(rows, _demangle([r["mangled"] for r in rows])):
        r["name"] = name
    return {r["name"]: r for r in rows}


def _example_isa():
    out = f"/tmp/smfft_test_isa_{os.getpid()}.s"
    p = subprocess.run([HIPCC] + FLAGS + ["-S", "--cuda-device-only", os.path.join(ROOT, "examples", "reference_shape_kernel.hip"), "-o", out], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    text = open(out).read()
    os.remove(out)
    return text


def _inst_isa(n):
    out = f"/tmp/smfft_test_isa_{os.getpid()}_{n}.s"
    p = subprocess.run([HIPCC] + FLAGS + part_flags(n, 2) + [f"-DSMFFT_N={n}", "-S", "--cuda-device-only", os.path.join(ROOT, "smfft_amd", "csrc", "smfft_inst.hip"), "-o", out], capture_output=True, text=True)      # the in-LDS kernels' object
    assert p.returncode == 0, p.stderr[-2000:]
    text = open(out).read()
    os.remove(out)
    return text


def _scratch_by_loop_depth(isa, mangled):
    """scratch instructions of one kernel by the nesting depth of the block they sit in (hipcc annotates every block label with its loop)"""
    m = re.search(r"^%s:[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(mangled), isa, re.S | re.M)
    assert m, mangled
    depth, found = 0, {}
    for line in m.group(1).split("\n"):
        if line.startswith(".LBB"):
            d = re.findall(r"Depth=(\d+)", line)
            depth = max(int(x) for x in d) if d else 0
        elif re.search(r"\bscratch_(load|store)", line):
            found[depth] = found.get(depth, 0) + 1
    return found


@pytest.fixture(scope="module")
def built():
    with ThreadPoolExecutor(max_workers=4) as ex:
        futs = {n: ex.submit(_resources, n) for n in (256, 1024, 2048)}
        isa = ex.submit(_example_isa)
        inst = {n: ex.submit(_inst_isa, n) for n in (32, 64, 2048, 4096)}
        return {"res": {n: f.result() for n, f in futs.items()}, "isa": isa.result(), "inst": {n: f.result() for n, f in inst.items()}}


def test_in_lds_kernels_keep_four_waves_per_simd(built):
    """the compact `multiple` kernels of the single-wave lengths and every R2C / C2R one: <= 128 VGPRs, nothing spilled"""
    seen = 0
    for n, res in built["res"].items():
        for name, r in res.items():
            single_wave_ct = n <= 1024 and name.startswith("SMFFT_DIT_multiple<") and "unfused" not in name
            if single_wave_ct or name.startswith("FFT_GPU_R2C_C2R_multiple<"):
                seen += 1
                assert r["vgpr"] <= 128 and r["occ"] >= 4, (name, r)
                assert r["scratch"] == 0, (name, r)
    assert seen >= 8 + 6, seen          # 4 CT variants x 2 lengths, 2 RC directions x 3 lengths


def test_no_in_lds_kernel_has_a_private_segment(built):
    """Round 6: no shipped `multiple` kernel -- C2C fused and per-call, Stockham, R2C / C2R; N = 32, 64, 2048, 4096 compiled
    here, 256 / 1024 / 2048 through the resource report above -- holds a scratch instruction anywhere or a private segment: round 5's
    N = 4096 kernels sat at the 128-register cap with 40-52 bytes per lane spilled around their tile copies (sixteen values in flight
    while 60 registers of twiddles were alive: the copies now move eight at a time), N = 32's carried addresses of the piece loop
    across the applications."""
    checked = 0
    for n, isa in built["inst"].items():
        for mangled in re.findall(r"^(_Z\w*(?:SMFFT_DIT_multiple|FFT_GPU_multiple|FFT_GPU_R2C_C2R_multiple)\w*):", isa, re.M):
            assert not _scratch_by_loop_depth(isa, mangled), (n, mangled, _scratch_by_loop_depth(isa, mangled))
            m = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(mangled), isa, re.S)
            assert m, mangled
            seg = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(1))
            assert seg and int(seg.group(1)) == 0, (n, mangled, seg and seg.group(1))
            checked += 1
    assert checked >= 9 + 9 + 7 + 7, checked
    for n, res in built["res"].items():
        for name, r in res.items():
            if "multiple" in name:
                assert r["scratch"] == 0, (n, name, r)


def test_external_kernel_budget(built):
    """the headline kernel (config 2): one 256-thread workgroup per 4096 elements, 34 KiB of LDS, no scratch, at least 3 waves per SIMD"""
    r = built["res"][1024]["SMFFT_DIT_external<FFT_1024_forward>"]
    assert r["scratch"] == 0 and r["occ"] >= 3 and r["lds"] <= 40 * 1024, r


def _loop_barriers(isa, mangled_fragment, demangled_fragment):
    """workgroup barriers inside the (only) loop of one kernel of the example file"""
    m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % mangled_fragment, isa, re.S | re.M)
    assert m, mangled_fragment
    name = _demangle([m.group(1)])[0]
    assert demangled_fragment in name, name
    body = m.group(2).split("\n")
    head = next(i for i, l in enumerate(body) if "Loop Header" in l)
    label = body[head].split(":")[0]
    tail = max(i for i, l in enumerate(body) if re.search(r"s_cbranch\w+\s+%s\b" % re.escape(label), l))
    return sum("s_barrier" in l for l in body[head:tail + 1])


def test_reference_shaped_multiple_loop_barriers_at_4096(built):
    """SMFFT_DIT_multiple<FFT_4096_*> in the reference's shape (1024 threads, NREUSES calls of do_SMFFT_CT_DIT with the loop's own
    barrier): natural order 4 workgroup barriers per application (round 3: 8, round 5: 5), no reorder 2 (round 5: 3) -- both cross-wave
    passes behind ONE barrier (quarter_fft's last phase; CT:553-572); N = 2048 the same"""
    isa = built["isa"]
    assert _loop_barriers(isa, "SMFFT_DIT_multipleI16FFT_4096_forwardE", "SMFFT_DIT_multiple<FFT_4096_forward>") <= 4
    assert _loop_barriers(isa, "SMFFT_DIT_multipleI26FFT_4096_forward_noreorderE", "SMFFT_DIT_multiple<FFT_4096_forward_noreorder>") <= 2
    assert _loop_barriers(isa, "SMFFT_DIT_multipleI16FFT_2048_forwardE", "SMFFT_DIT_multiple<FFT_2048_forward>") <= 4
    assert _loop_barriers(isa, "SMFFT_DIT_multipleI26FFT_2048_forward_noreorderE", "SMFFT_DIT_multiple<FFT_2048_forward_noreorder>") <= 2


def _loops(body):
    """(first, last) line of every loop of a kernel body: a label that a later conditional branch jumps back to"""
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    out = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def test_lane_engine_applications_stay_in_registers(built):
    """N = 32 and N = 64 without reorder in the in-LDS path (PairEngine32 / QuadEngine64): the loop over two applications
    holds the cross stages -- DPP-fed v_fmac_f32 and, for part of them, ds_swizzle_b32 (the LDS crossbar, no memory) + v_fmac_f32 -- and
    no LDS memory instruction, no barrier, no scratch access: the image in LDS is touched where a piece starts and ends only"""
    # (fused, swizzled) per iteration of two applications: N = 32 natural order dit 32 + dif 16 fused, dif 16 swizzled; no reorder 2 x (16 + 16);
    # N = 64 no reorder: stage 1 swizzled (32 per application), stage 2 fused (32)
    for n, frag, fused, swizzled in ((32, "SMFFT_DIT_multipleI14FFT_32_forwardE", 48, 16), (32, "SMFFT_DIT_multipleI24FFT_32_forward_noreorderE", 32, 32), (32, "SMFFT_DIT_multipleI14FFT_32_inverseE", 48, 16),
                                     (64, "SMFFT_DIT_multipleI24FFT_64_forward_noreorderE", 64, 64), (64, "SMFFT_DIT_multipleI24FFT_64_inverse_noreorderE", 64, 64)):
        isa = built["inst"][n]
        m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % frag, isa, re.S | re.M)
        assert m, frag
        body = m.group(2).split("\n")
        pair = [(a, b) for a, b in _loops(body) if sum("v_fmac_f32_dpp" in l for l in body[a:b + 1]) == fused]
        assert pair, frag
        a, b = min(pair, key=lambda ab: ab[1] - ab[0])
        loop = [l.strip() for l in body[a:b + 1] if l.strip() and not l.strip().startswith((";", "."))]
        memory = [l for l in loop if l.startswith(("ds_", "s_barrier", "scratch_", "global_", "buffer_")) and not l.startswith("ds_swizzle_b32")]
        assert not memory, memory[:4]
        assert sum(l.startswith("ds_swizzle_b32") for l in loop) == swizzled, (frag, sum(l.startswith("ds_swizzle_b32") for l in loop))
        valu = sum(l.startswith("v_") and "dpp" not in l for l in loop)
        assert valu <= 2 * (245 if n == 32 else 285), (frag, valu)          # per application: N = 32 216 (natural order) / 242 (no reorder); N = 64: 273 (32 of them selects, 32 the plain v_fmac of the swizzled stage)


def test_reference_shaped_multiple_loop_keeps_its_twiddles(built):
    """SMFFT_DIT_multiple<P> in the reference's shape: the twiddles of do_SMFFT_CT_DIT are fetched in front of its first
    synchronisation, so the loop of NREUSES calls holds no global load at any length (round 6: N = 2048 / 4096 keep theirs too --
    56 / 64 registers, still eight waves per SIMD) -- DESIGN.md 2.3, CT:553-572"""
    isa = built["isa"]
    for frag, allowed in (("SMFFT_DIT_multipleI15FFT_256_forwardE", 0), ("SMFFT_DIT_multipleI16FFT_1024_forwardE", 0), ("SMFFT_DIT_multipleI26FFT_1024_forward_noreorderE", 0),
                          ("SMFFT_DIT_multipleI16FFT_2048_forwardE", 0), ("SMFFT_DIT_multipleI16FFT_4096_forwardE", 0), ("SMFFT_DIT_multipleI26FFT_4096_forward_noreorderE", 0)):
        m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % frag, isa, re.S | re.M)
        assert m, frag
        body = m.group(2).split("\n")
        a, b = max(_loops(body), key=lambda ab: ab[1] - ab[0])
        loads = sum("global_load" in l for l in body[a:b + 1])
        assert loads <= allowed, (frag, loads)


def test_reference_shaped_loops_hold_no_packed_fp32(built):
    """Packed fp32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) runs at half rate on gfx950 and costs moves to line register pairs up; the
    library is built with -fno-slp-vectorize, but a float2 that reaches arithmetic as a <2 x float> value gets packed all the same -- round 6 lost
    11 % at N = 4096 natural order to ten v_pk_add_f32 that a ternary between two float2 loads brought in.  No reference-shaped `multiple` loop
    (the in-LDS benchmark shape, CT:553-572) holds one."""
    isa = built["isa"]
    seen = 0
    for m in re.finditer(r"^(_Z18SMFFT_DIT_multipleI\w+):[^\n]*\n(.*?)\n\s*s_endpgm", isa, re.S | re.M):
        packed = [l.strip() for l in m.group(2).split("\n") if re.match(r"\s*v_pk_(add|mul|fma)_f32", l)]
        assert not packed, (m.group(1), packed[:3])
        seen += 1
    assert seen >= 16, seen


def test_convolution_example_fetches_its_filter_with_the_series(built):
    """The user's convolution kernels on the reference's contract (examples/reference_shape_kernel.hip, README.md:10-18): the filter's
    values are fetched with the series.  Fetched between the two transforms they sat behind a barrier, one exposed L2 latency per
    block (round 6: 2.05 -> 1.88 ms on the config-2 batch; profiles/r06_convolution_user_kernel.txt).  In the ISA: no global load
    behind the forward transform's first LDS write-back, i.e. every load (series, filter, the device function's twiddles, fetched at
    its top) is in front of the second barrier of the kernel."""
    isa = built["isa"]
    for frag in ("user_convolution_kernelI16FFT_1024_forward16FFT_1024_inverseE", "user_convolution_kernel_registersI16FFT_1024_forward16FFT_1024_inverseE"):
        m = re.search(r"^(_Z\d+%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % frag, isa, re.S | re.M)
        assert m, frag
        body = [l.strip() for l in m.group(2).split("\n")]
        barriers = [i for i, l in enumerate(body) if l.startswith("s_barrier")]
        loads = [i for i, l in enumerate(body) if l.startswith("global_load")]
        assert len(loads) >= 8 and len(barriers) >= 4, (frag, len(loads), len(barriers))
        assert max(loads) < barriers[1], (frag, max(loads), barriers[:3])
