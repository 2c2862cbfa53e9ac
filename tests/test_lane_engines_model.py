"""The book-keeping of the lane engines (include/smfft/smfft_engine.hpp: PairEngine32, QuadEngine64) replayed in NumPy against numpy.fft:
layouts A / B and their alternation, the renaming of the no-reorder variants, the sign vectors by the application's number in the
chain, lane j <-> stored block rev2(j) at N = 64, the turn of lane 3, and chains cut anywhere with the sign flip of odd pieces
(tools/lane_engines_model.py).  CPU only: that the kernels implement this is what the GPU parity and bit-identity tests check."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_engines_model_matches_numpy_fft():
    spec = importlib.util.spec_from_file_location("lane_engines_model", os.path.join(ROOT, "tools", "lane_engines_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check() < 1e-12
