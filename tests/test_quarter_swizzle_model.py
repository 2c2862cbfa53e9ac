"""The four-elements-per-thread engine's LDS swizzle (quarter_swizzle, include/smfft/smfft_device_functions.hpp): the header
computes what tools/quarter_swizzle.py models, the function is a GF(2)-linear permutation of every aligned group of 32
elements (so it stays inside the caller's N float2 and a pass's four addresses are one swizzled base XOR three constants),
and in the model (gfx950 lane groups and banks, MI355X_MICROARCH.md) it removes the bank conflicts of the natural layout.
CPU only: the header's constexpr function is evaluated by a host-side hipcc compile (no GPU code is run)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import quarter_swizzle as qs  # noqa: E402


def test_swizzle_is_a_linear_permutation_of_aligned_groups():
    f = qs.product_swizzle
    for block in range(0, 4096, 32):
        assert sorted(f(i) for i in range(block, block + 32)) == list(range(block, block + 32))
    for a in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 37, 1234, 4095):
        for b in (3, 5, 48, 1000, 2048, 4094):
            assert f(a ^ b) == f(a) ^ f(b)
    assert all(f(m) == m for m in range(4))      # a thread's four pass-0 results stay one aligned group of four


@pytest.mark.parametrize("n", [32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_swizzle_removes_the_conflicts_in_the_model(n):
    natural, swizzled, ideal = qs.lds_cycles(n, lambda i: i), qs.lds_cycles(n, qs.product_swizzle), qs.ideal_cycles(n)
    assert swizzled <= 1.13 * ideal, (n, swizzled, ideal)      # what is left: 2-way stores (8 instead of 6 cycles) in one pass
    assert swizzled < natural
    if n >= 256:
        assert natural >= 2 * swizzled - 150, (n, natural, swizzled)


def test_header_computes_the_models_function(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = tmp_path / "swz.hip"
    src.write_text(r'''
#include <cstdio>
#include "smfft_device.hpp"
int main() {
    for (int i = 0; i < 4096; ++i) printf("%d\n", smfft::quarter_swizzle(i));
    return 0;
}
''')
    exe = tmp_path / "swz"
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                          stderr=subprocess.DEVNULL)
    got = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    assert got == [qs.product_swizzle(i) for i in range(4096)]


def test_lane_and_register_form_of_the_ladder_is_the_dft():
    """tools/quarter_lanes_model.py: the N <= 256 form of the reference-contract engine (exchanges between passes as lane <-> slot
    bit transposes, no LDS) computes the DFT (natural order) / the DFT of the bit-reversed input (no reorder) for N = 32 ... 256,
    and its results end where QuarterLanes::out_offset says; for N = 512 ... 4096 the lane passes inside every aligned block of 256
    elements (all four without reorder: kLanesHead; passes 1 ... 3 after the scattered first pass in natural order: kLanesMiddle)
    followed by the passes through LDS, and that a block's thread ends with the elements lane + 64 i the next pass expects."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import quarter_lanes_model
    assert quarter_lanes_model.check() < 1e-14


def test_phased_form_of_the_ladder_is_the_dft_and_conflict_free():
    """tools/quarter_phases_model.py (round 6): the reference-contract engine of N >= 256 cut into phases in which a wave owns eight
    index bits -- natural order read back with slots = index bits (2, 3), the no-reorder ladder's middle exchange through the wave's
    own block of LDS, the two cross-wave passes of N = 2048 / 4096 between one pair of barriers -- computes the DFT (natural order) /
    the DFT of the bit-reversed input (no reorder), both directions, and every LDS access of it is conflict free under the gfx950
    lane-group rules except the four natural-layout reads of N = 4096's last phase (2-way: the price of storing without a barrier)."""
    import quarter_phases_model as qp
    assert qp.check() < 1e-14
    assert qp.M256 in qp.search_256()
    assert all(qp.apply_rows(qp.M256, p) == qp.image256(p) for p in range(256))
    for n in (256, 512, 1024, 2048):
        for reorder in (1, 0):
            total, ideal, _ = qp.lds_report(n, reorder)
            assert total == ideal, (n, reorder, total, ideal)
    for reorder in (1, 0):
        total, ideal, _ = qp.lds_report(4096, reorder)
        assert total == ideal + 16 * 4 * 2, (reorder, total, ideal)      # sixteen waves x four reads x two extra cycles


@pytest.mark.parametrize("n", [512, 1024, 2048])
def test_natural_order_in_pairs_of_passes_is_the_dft_and_conflict_free(n):
    """quarter_fft's natural-order form of N = 512 / 1024 / 2048 in PAIRS of passes (round 6, last day; SMFFT_QUARTER_PAIRS): every phase two passes
    with the exchange of the lane bits 4, 5 between them; tools/quarter_phases_model.py, transform_pairs replays lanes, slots, swaps and the
    swizzled image against numpy.fft, both directions; every LDS access is conflict free under the gfx950 lane-group rules, as many accesses as
    in the three-pass form; N = 512 stores into the words it read (asserted inside the model).  N = 2048 runs pass 1 in front of the scattered
    store: its four loads of the caller's natural layout are 2-way conflicted (eight waves x four reads x two cycles), nothing else."""
    import numpy as np
    import quarter_phases_model as qp
    rng = np.random.default_rng(n)
    for direction in (0, 1):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        got = qp.transform_pairs(n, direction, x)
        want = np.fft.ifft(x) * n if direction else np.fft.fft(x)
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-14
    total, ideal, count = qp.lds_report_pairs(n)
    assert total == ideal + (64 if n == 2048 else 0) and count == qp.lds_report(n, 1)[2], (total, ideal, count)


@pytest.mark.parametrize("n", [64, 128])
@pytest.mark.parametrize("lanes", [32, 64])
def test_small_lengths_take_one_trip_through_the_blocks_image(n, lanes):
    """N = 64 / 128 on the reference's contract (quarter_small_natural / quarter_small_noreorder; upstream's 32-thread blocks and the 64-thread
    _wave64 classes): replayed in NumPy against numpy.fft inside check(); here: the scattered stores of pass 0 and the slot-(2,3) read-back of the
    natural-order form are conflict free under the gfx950 lane-group rules in both block shapes (the no-reorder form's exchange is checked
    inside check()); what conflicts is the contract's own natural layout of several transforms per wave."""
    import numpy as np
    import quarter_phases_model as qp
    log = []
    qp.transform_small(n, 0, np.zeros((lanes // (n // 4), n), complex), log)
    assert len(log) == 16
    for kind, addr in log[4:12]:                     # four scattered stores, four reads
        group, banks = (32, 32) if kind == "r" else (16, 16)
        for g in range(0, lanes, group):
            assert len({a % banks for a in addr[g:g + group]}) == group, (n, lanes, kind)


def test_header_computes_the_phase_images(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    import quarter_phases_model as qp
    src = tmp_path / "img.hip"
    src.write_text(r'''
#include <cstdio>
#include "smfft_device.hpp"
int main() {
    for (int i = 0; i < 256; ++i) printf("%d %d %d %d %d\n", smfft::quarter_image256(i), smfft::QuarterLanes<256, 0, 0>::exchange_image(i),
                                         smfft::quarter_small_image<32>(i), smfft::quarter_small_image<64>(i), smfft::quarter_small_image<128>(i));
    return 0;
}
''')
    exe = tmp_path / "img"
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                          stderr=subprocess.DEVNULL)
    got = [tuple(int(v) for v in line.split()) for line in subprocess.check_output([str(exe)], text=True).splitlines()]
    assert got == [(qp.image256(i), qp.exchange_image(i), qp.small_image(32, i), qp.small_image(64, i), qp.small_image(128, i)) for i in range(256)]
