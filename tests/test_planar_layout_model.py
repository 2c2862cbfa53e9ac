"""The planar in-LDS engine's row layout (include/smfft/smfft_planar.hpp): the residues the header computes are the ones
tools/soa_model.py holds, and in the model (gfx950 lane groups and banks per instruction, MI355X_MICROARCH.md) every read set
of every length is bank-conflict free with them.  CPU only: the header's constexpr functions are evaluated by a host-side
hipcc compile (no GPU code is run)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import soa_model  # noqa: E402


@pytest.mark.parametrize("n", sorted(soa_model.HEADER_RESIDUES))
def test_header_residues_are_conflict_free_in_the_model(n):
    got = soa_model.header_conflicts(n)
    assert got and all(v == 0 for v in got.values()), got


def test_header_computes_the_models_residues(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = tmp_path / "residues.hip"
    src.write_text(r'''
#include <cstdio>
#include "smfft/smfft_planar.hpp"
template <int N, int REORDER>
void dump() {
    const char* names[3] = {"image", "x1", "x2"};
    const smfft::RowKind kinds[3] = {smfft::RowKind::image, smfft::RowKind::x1, smfft::RowKind::x2};
    for (int k = 0; k < 3; ++k) {
        printf("%d %d %s", N, REORDER, names[k]);
        for (int j = 0; j < 16; ++j) printf(" %d", smfft::row_residue<N, REORDER>(kinds[k], j));
        printf("\n");
    }
}
int main() {
    dump<64, 0>(); dump<128, 0>(); dump<256, 0>(); dump<512, 0>(); dump<512, 1>(); dump<1024, 0>(); dump<1024, 1>();
    dump<2048, 0>(); dump<2048, 1>(); dump<4096, 0>(); dump<4096, 1>();
    return 0;
}
''')
    exe = tmp_path / "residues"
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                          stderr=subprocess.DEVNULL)
    rows = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        n, reo, kind, *vals = line.split()
        rows[(int(n), int(reo), kind)] = [int(v) for v in vals]
    H = soa_model.HEADER_RESIDUES
    for n, tables in H.items():
        assert rows[(n, 0, "image")] == tables["image"], (n, "image")
        if "x1" in tables:
            assert rows[(n, 0, "x1")] == tables["x1"] and rows.get((n, 1, "x1"), tables["x1"]) == tables["x1"], (n, "x1")
        if "last" in tables:
            assert rows[(n, 0, "x2")] == tables["last"], (n, "last")
        if "x2" in tables:
            assert rows[(n, 0, "x2")] == tables["x2"], (n, "x2")
        if "x2_reorder" in tables:
            assert rows[(n, 1, "x2")] == tables["x2_reorder"], (n, "x2 reorder")
    assert rows[(4096, 0, "x2")] == H[4096]["x2_reorder"]       # N = 4096 computes klow = role in both orderings
    assert rows[(512, 1, "x2")] == H[512]["x2"]                 # N = 512 keeps roles = positions in both


def test_pairwise_hermitian_split_and_merge_bookkeeping():
    """tools/hermitian_pairs_model.py: PlanarEngine::hermitian_apply_pairs replayed thread by thread -- who holds which element, whose
    register the partner is, which rows the second results travel through, role 0's own pairs and the packed element 0 -- gives the
    packed rfft (RC:269-344, S5) and, as a merge in front of the inverse transform, (N/2) x (S6), for complex lengths 256 ... 2048."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hermitian_pairs_model
    assert hermitian_pairs_model.check() < 1e-13
