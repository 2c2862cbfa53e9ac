#!/usr/bin/env python3
"""bench.py -- headline benchmark of smfft_amd on MI355X.

Metric (BASELINE.json): batched FFTs/s + achieved HBM GB/s, N=1024 C2C forward, 4 GB input,
1/2/4/8 GPU.  A "step" is one pass of the hot path -- one FFT_external_benchmark-equivalent launch
(SMFFT_DIT_external<FFT_1024_forward>) over a batch of 524288 FFTs (4 GiB in, 4 GiB out) that is
already resident in HBM (config 2 of BASELINE.json).  With N GPUs every rank owns its own 4 GiB
batch (config 5: weak scaling, no data-path collective; RCCL only reduces the timings).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself: the parent process never touches
the GPU, it only spawns one child per rank and returns the worst exit code.  `--gpus N` under a launcher whose
WORLD_SIZE is not N is an error.

Prints ONE JSON line on rank 0 (see the driver contract).  Objects beyond the contract:
  roofline        the dominant kernel against the HBM roofline (8.0 TB/s datasheet peak) on the buffers the
                  library's allocator hands out (smfft_malloc_pair: budget-bounded placement);
                  `achieved` = algorithmic bytes per launch (2 * N * nFFTs * 8 B) / average launch duration measured
                  with events on the launch stream over the timed region.
  roofline_plain  the same kernel, same pre-warm, same number of timed launches, on two PLAIN hipMalloc buffers
                  (what a caller of FFT_external_benchmark brings along, CT:850-853).
  roofline_own_input  the same on a plain hipMalloc INPUT with only the output taken from smfft_malloc_written_for
  configs         N = 1 only, same buffers: config 2 at every length (forward / inverse x reorder / no reorder); config 3
                  (N = 32..4096, FFT_multiple_benchmark: ms, FFT/s, fraction of the fp32 peak on the README batch and on a
                  saturating batch of 8 x its slots); config 4 (real N = 512..4096, R2C and C2R); the Stockham program; and
                  `reference_contract`: the two-argument kernels in the reference's own launch shape (blockDim = N/4).
  cpu_baseline    FFTW-API batched C2C (MKL's FFTW3 interface; real FFTW is not in the image) on the host cores of
                  this box over the SAME input as the GPU run; falls back to the oracle's C restatement.
PyTorch is plumbing only (device memory, stream, torch.distributed); the transform is the HIP
library behind the C ABI (include/smfft.h).  There is no CPU fallback in the timed path.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FFT_SIZE = 1024
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak (spec)


def cpu_baseline(x, threads):
    """FFTW-API batched plan on the host (oracle/fftw_baseline.so) over x = (nFFTs, N) complex64 -- the GPU run's own
    input; fallback: the oracle's restatement on a slice of it."""
    import numpy as np

    sample_ffts, n = x.shape
    out = np.empty_like(x)
    fp = ctypes.POINTER(ctypes.c_float)
    res = None
    try:
        fb = ctypes.CDLL(os.path.join(ROOT, "oracle", "fftw_baseline.so"))
        fb.fftw_baseline_init.argtypes = [ctypes.c_int]
        fb.fftw_baseline_backend.restype = ctypes.c_char_p
        fb.fftw_baseline_c2c_sliced.restype = ctypes.c_double
        fb.fftw_baseline_c2c_sliced.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        # one single-threaded plan per batch slice, slices run concurrently (oracle/fftw_baseline.c);
        # a few thread counts are tried for a bounded time, `cores` = the count that won
        # (largest counts first: they are the ones that win on the GPU boxes' 128-256 hardware threads, and the budget may
        # not reach the small ones on a 4 GiB input)
        cands = sorted({c for c in (8, 32, 64, 128, threads // 2, threads) if 1 <= c <= threads}, reverse=True)
        best_all, best_thr = 1e30, 0
        out.fill(0)                       # first touch of the output outside the timing
        t_start = time.time()
        if fb.fftw_baseline_init(1):
            for thr in cands:
                if time.time() - t_start > 30.0:
                    break
                t = fb.fftw_baseline_c2c_sliced(x.ctypes.data_as(fp), out.ctypes.data_as(fp), n, sample_ffts, 0, 5, thr)
                if 0 < t < best_all:
                    best_all, best_thr = t, thr
        if best_all < 1e29:
            res = {"value": sample_ffts / best_all, "unit": "FFT/s", "cores": best_thr, "kind": "port",
                   "impl": fb.fftw_baseline_backend().decode() + " fftwf_plan_many_dft (FFTW_ESTIMATE) per batch slice, out of place, slices on pthreads; thread counts tried " + str(cands),
                   "sample": f"N={n} C2C forward, {sample_ffts} FFTs ({x.nbytes >> 20} MiB): the GPU run's own input copied to the host, best of 5 passes per thread count",
                   "host_cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    except OSError:
        pass
    if res is None:
        olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        olib.oracle_ct_c2c_f32.argtypes = [fp, fp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int]
        part = min(sample_ffts, 131072)
        best = 1e30
        for _ in range(5):
            t0 = time.time()
            olib.oracle_ct_c2c_f32(x.ctypes.data_as(fp), out.ctypes.data_as(fp), n, part, 0, 1)
            best = min(best, time.time() - t0)
        res = {"value": part / best, "unit": "FFT/s", "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)),
               "kind": "port", "impl": "oracle/smfft_oracle.c radix-2 restatement, OpenMP over FFTs",
               "sample": f"N={n} C2C forward, the first {part} FFTs of the GPU run's input, best of 5"}
    res["GB/s"] = res["value"] * 2 * n * 8 / 1e9
    # `kind` has two values in the bench contract: "reference" (the reference's own code: CUDA, unbuildable here) and "port".  The headline
    # figure above is a LIBRARY's transform (the FFTW3 API, served by MKL on the GPU boxes: what BASELINE.json's north_star names as the CPU
    # baseline), not the oracle; the oracle's own restatement of the reference's radix-2 ladder (oracle/smfft_oracle.c, OpenMP over the
    # FFTs) is timed beside it on a bounded slice so that both readings of "port" are in the line.
    if "oracle" not in res.get("impl", ""):
        res["what"] = "FFTW3-API library transform (north_star's CPU baseline), not the oracle; the oracle's port: oracle_port"
        try:
            olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
            olib.oracle_ct_c2c_f32.argtypes = [fp, fp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int]
            part = min(sample_ffts, 65536)
            best, t_begin = 1e30, time.time()
            for _ in range(3):
                t0 = time.time()
                olib.oracle_ct_c2c_f32(x.ctypes.data_as(fp), out.ctypes.data_as(fp), n, part, 0, 1)
                best = min(best, time.time() - t0)
                if time.time() - t_begin > 10.0:
                    break
            res["oracle_port"] = {"value": part / best, "unit": "FFT/s", "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)),
                                  "sample": f"the first {part} FFTs of the same input, best of 3", "impl": "oracle/smfft_oracle.c (the reference's radix-2 DIT ladder restated), OpenMP over FFTs"}
        except OSError:
            pass
    return res


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/*_pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE x2 per the
    gfx950 correction, calibrated on the same access shape) and where the figure comes from: the counters cannot be
    collected inside this run (rocprofv3 wraps a whole process), so the line names the profile it quotes.
    (None, reason) if no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, "no profiles/r*_pmc_traffic.json committed"
    try:
        k = json.load(open(files[-1]))["kernels"]["void SMFFT_DIT_external<FFT_1024_forward>"]
        commit = k.get("commit") or json.load(open(files[-1])).get("commit")     # written by tools/collect_profiles.py
        try:
            commit = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", files[-1]], capture_output=True, text=True, timeout=10).stdout.strip() or commit
        except Exception:
            pass
        return k["hbm_bytes_per_launch"], {"file": os.path.relpath(files[-1], ROOT), "commit": commit,
                                           "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not collected in this run"}
    except (KeyError, ValueError):
        return None, "profile unreadable"


def hipfft_ms(torch, dev, stream, i_ptr, o_ptr, n, nffts, reps=10):
    """hipFFT (rocFFT) on the same device buffers, same events: the vendor library as a same-run yardstick (the harness's
    GPU_cuFFT analogue, smfft_vendor.hip).  None if the library cannot be loaded."""
    try:
        lib = ctypes.CDLL("libhipfft.so")
    except OSError:
        try:
            lib = ctypes.CDLL("/opt/rocm/lib/libhipfft.so")
        except OSError:
            return None
    plan = ctypes.c_void_p()
    HIPFFT_C2C, HIPFFT_FORWARD = 0x29, -1
    lib.hipfftPlan1d.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.hipfftSetStream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.hipfftExecC2C.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.hipfftDestroy.argtypes = [ctypes.c_void_p]
    if lib.hipfftPlan1d(ctypes.byref(plan), n, HIPFFT_C2C, nffts) != 0:
        return None
    lib.hipfftSetStream(plan, ctypes.c_void_p(stream.cuda_stream))
    for _ in range(3):
        lib.hipfftExecC2C(plan, i_ptr, o_ptr, HIPFFT_FORWARD)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        lib.hipfftExecC2C(plan, i_ptr, o_ptr, HIPFFT_FORWARD)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    lib.hipfftDestroy(plan)
    return e0.elapsed_time(e1) / reps


def vram_used_bytes(torch, dev):
    """mem_info_vram_used of the card that runs the bench (None if its sysfs node cannot be identified)."""
    try:
        import glob
        p = torch.cuda.get_device_properties(dev)
        want = "%04x:%02x:%02x." % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        for d in glob.glob("/sys/class/drm/card*/device"):
            if os.path.basename(os.path.realpath(d)).startswith(want):
                return int(open(os.path.join(d, "mem_info_vram_used")).read())
    except Exception:
        pass
    return None


def device_info(torch, dev):
    """Which part ran the numbers (runs on different boxes of the pool differ by up to 6 %)."""
    info = {}
    try:
        p = torch.cuda.get_device_properties(dev)
        for k in ("name", "gcnArchName", "total_memory", "multi_processor_count", "clock_rate", "memory_clock_rate", "memory_bus_width", "L2_cache_size",
                  "pci_domain_id", "pci_bus_id", "pci_device_id"):
            if hasattr(p, k):
                info[k] = getattr(p, k)
    except Exception as e:   # informational only
        info["error"] = repr(e)
    # clocks / partition modes of the amdgpu device (best effort, read-only sysfs)
    try:
        import glob
        want = None
        if all(k in info for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            want = "%04x:%02x:%02x." % (info["pci_domain_id"], info["pci_bus_id"], info["pci_device_id"])
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            bdf = os.path.basename(os.path.realpath(d))
            if want is None or not bdf.startswith(want):
                continue                  # only the card that ran the bench (other tenants' cards share the sysfs tree)
            sysfs = {}
            for name in ("current_memory_partition", "current_compute_partition", "pp_dpm_mclk", "pp_dpm_sclk", "pp_dpm_fclk",
                         "mem_info_vram_total", "mem_info_vram_used", "power_dpm_force_performance_level"):
                try:
                    sysfs[name] = open(os.path.join(d, name)).read().strip().replace("\n", " | ")
                except OSError:
                    pass
            if sysfs:
                sysfs["pci"] = bdf
                info["sysfs"] = sysfs
                break
    except Exception as e:
        info["sysfs_error"] = repr(e)
    return info


def _sig(x, digits=6):
    """floats to `digits` significant digits (the compact line is bounded), NaN / inf -> None (strict JSON)"""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"%.{digits}g" % x)
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def _minmax(values):
    vals = [v for v in values if isinstance(v, (int, float)) and v == v]
    return [min(vals), max(vals)] if vals else None


def _r3(v):
    return round(v, 3) if isinstance(v, float) and v == v and abs(v) != float("inf") else (v if isinstance(v, int) else None)


LENGTHS = ("32", "64", "128", "256", "512", "1024", "2048", "4096")
COMPACT_LIMIT = 4096      # bytes: the driver's parser lost round 3's 22.6 KB line
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline", "pair_search", "comm_backend", "ranks_seen", "spot_check_relL2")


def compact_line(detail):
    """The ONE line of the driver contract, built from the full record (`detail`, written to bench_detail.json): the contract's
    keys, `roofline` and `cpu_baseline`, the outcome of every allocation attempt, the per-rank list, and a dozen scalars that
    summarise configs 3 / 4 and the reference-contract path.  Strict JSON (no NaN), under COMPACT_LIMIT bytes."""
    roof = dict(detail["roofline"])
    src = roof.get("traffic_source")
    if isinstance(src, dict):
        roof["traffic_source"] = f"{src.get('file')}@{src.get('commit')} (separate --pmc passes of this command)"
    plain, own = detail.get("roofline_plain"), detail.get("roofline_own_input")
    cpu = detail.get("cpu_baseline")
    if cpu:
        cpu = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "sample", "GB/s")}
        cpu["sample"] = (cpu["sample"] or "")[:120]
        cpu["impl"] = (detail["cpu_baseline"].get("impl") or "")[:60]
        port = detail["cpu_baseline"].get("oracle_port")
        if port:
            cpu["oracle_port"] = {"value": port.get("value"), "cores": port.get("cores")}
    attempts = []
    for a in detail.get("pair_attempts") or []:
        attempts.append({"budget": (a.get("budget") or "")[:7], "good_enough": a.get("good_enough"), "classification": a.get("classification"),
                         "copy_ms": a.get("copy_ms"), "read_ms": a.get("read_ms"), "chunks": a.get("candidates"),
                         "seconds": (a.get("search_ms") or 0.0) / 1e3, "kept": bool(a.get("kept"))})
    cfg = detail.get("configs") or {}
    c3, c4, c2 = cfg.get("config3_multiple") or {}, cfg.get("config4_r2c_c2r_external") or {}, cfg.get("config2_external_by_length") or {}
    cref = cfg.get("reference_contract") or {}
    by_len = cref.get("by_length") or {}
    summary = {
        # the in-LDS path at N = 1024, FFT/s: what ONE call of the device function costs (contract path; the compact kernel
        # without cross-application fusion) next to the fused compact kernel
        "in_lds_1024_contract_FFTps": _minmax([_get(cref, "ct_multiple_reorder", "FFT/s"), _get(cref, "ct_multiple_noreorder", "FFT/s")]),
        "in_lds_1024_unfused_FFTps": _get(c3, "1024", "reorder", "unfused", "FFT/s"),
        "config3_unfused_frac": _minmax([_get(c3, k, o, "unfused", "frac_fp32_peak") for k in c3 for o in ("reorder", "noreorder")]),
        # one row per length 32, 64, ..., 4096: [fused loop natural order, no reorder, one image trip per application natural order, no reorder]
        # (fraction of the fp32 peak) and [natural order, no reorder] of the reference's contract over the compact kernel
        "config3_by_length_frac": [[_r3(_get(c3, k, "reorder", "frac_fp32_peak")), _r3(_get(c3, k, "noreorder", "frac_fp32_peak")),
                                    _r3(_get(c3, k, "reorder", "unfused", "frac_fp32_peak")), _r3(_get(c3, k, "noreorder", "unfused", "frac_fp32_peak"))] for k in LENGTHS if k in c3],
        "contract_in_lds_ratio_by_length": [[_r3(_get(by_len, k, "reorder", "in_lds_ratio_to_compact")), _r3(_get(by_len, k, "noreorder", "in_lds_ratio_to_compact"))] for k in LENGTHS if k in by_len],
        # FFT -> .H -> IFFT of the config-2 batch inside one user kernel: [reference contract, its register form, register-level engine]
        "convolution_ms": [_get(cref, "convolution_1024", k, "ms") for k in ("contract", "contract_registers", "register_engine")],
        "convolution_frac_hbm": [_get(cref, "convolution_1024", k, "frac") for k in ("contract", "contract_registers", "register_engine")],
        "config3_old_schedule_frac": _minmax([_get(c3, k, "reorder", "one_chain_per_workgroup_oldest_first", "frac_fp32_peak") for k in c3]),
        "in_lds_1024_fused_FFTps": _minmax([_get(c3, "1024", "reorder", "FFT/s"), _get(c3, "1024", "noreorder", "FFT/s")]),
        "config3_frac_fp32_peak": _minmax([_get(c3, k, o, "frac_fp32_peak") for k in c3 for o in ("reorder", "noreorder")]),
        "config3_saturating_frac": _minmax([_get(c3, k, o, "saturating_batch", "frac_fp32_peak") for k in c3 for o in ("reorder", "noreorder")]),
        "config3_frac_2048_4096": [_get(c3, k, o, "frac_fp32_peak") for k in ("2048", "4096") for o in ("reorder", "noreorder")],
        "config2_by_length_frac": _minmax([_get(c2, k, o, "frac") for k in c2 for o in ("forward", "inverse", "forward_noreorder", "inverse_noreorder")]),
        "config4_r2c_frac": _get(c4, "r2c", "frac"), "config4_c2r_frac": _get(c4, "c2r", "frac"),
        # in-LDS R2C / C2R (real N = 512 ... 4096): time over the C2C of the same complex length, same slots (section 5.3)
        "rc_in_lds_r2c_over_c2c": [_get(c4, k, "in_lds", "r2c_over_c2c") for k in ("512", "1024", "2048", "4096")],
        "rc_in_lds_c2r_over_c2c": [_get(c4, k, "in_lds", "c2r_over_c2c") for k in ("512", "1024", "2048", "4096")],
        "contract_external_ratio_to_tiled": _minmax([_get(by_len, k, o, "external_ratio_to_tiled") for k in by_len for o in ("reorder", "noreorder")]),
        "contract_in_lds_ratio_to_compact": _minmax([_get(by_len, k, o, "in_lds_ratio_to_compact") for k in by_len for o in ("reorder", "noreorder")]),
        "contract_small_N_external_ratio": [_get(by_len, k, "reorder", "external_ratio_to_tiled") for k in ("32", "64", "128")],
        "contract_small_N_in_lds_ratio": [_get(by_len, k, "reorder", "in_lds_ratio_to_compact") for k in ("32", "64", "128")],
        "contract_wave64_small_N_external_ratio": [_get(by_len, k, "reorder", "wave64", "external_ratio_to_tiled") for k in ("32", "64", "128")],
        "contract_wave64_small_N_in_lds_ratio": [_get(by_len, k, "reorder", "wave64", "in_lds_ratio_to_compact") for k in ("32", "64", "128")],
        "hipfft_ms_on_pair": _get(detail, "vendor_hipfft", "ms_on_pair"),
    }
    line = {k: detail.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                       "vs_baseline", "dtype", "data")}
    conf = dict(detail["config"])
    conf["buffers"] = (conf.get("buffers") or "")[:60]
    line.update({
        "config": conf,
        "hbm_GBps_per_gpu": detail.get("hbm_GBps_per_gpu"),
        "roofline": roof,
        "roofline_plain": {k: plain.get(k) for k in ("frac", "kernel_ms", "frac_of_copy")} if plain else None,
        "roofline_own_input": {k: own.get(k) for k in ("frac", "kernel_ms")} if own else None,
        "cpu_baseline": cpu,
        "pair_search": attempts,
        "per_rank": detail.get("per_rank"),
        "value_sum_of_rates": detail.get("value_sum_of_rates"),
        "multiple_path_FFTps": {k: _get(detail, "multiple_path", k, "FFT/s") for k in ("noreorder", "reorder")},
        "summary": summary if cfg else None,
        "comm_backend": detail.get("comm_backend"),
        "ranks_seen": detail.get("ranks_seen"),
        "spot_check_relL2": detail.get("spot_check_relL2"),
        "detail": detail.get("detail_file"),
    })
    line = _sig(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= COMPACT_LIMIT:          # never again an unparseable line: drop the optional parts, largest first
        for k in ("roofline_own_input", "pair_search_detail", "per_rank", "summary", "multiple_path_FFTps"):
            if k == "pair_search_detail":       # keep the verdicts, drop the figures of every allocation attempt
                line["pair_search"] = [{"good_enough": a.get("good_enough"), "classification": a.get("classification"), "copy_ms": a.get("copy_ms"),
                                        "chunks": a.get("chunks"), "seconds": a.get("seconds"), "kept": a.get("kept")} for a in (line.get("pair_search") or [])]
                text = json.dumps(line, allow_nan=False, separators=(",", ":"))
                if len(text) < COMPACT_LIMIT:
                    break
                continue
            line[k] = None
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(text) < COMPACT_LIMIT:
                break
    return text


def spawn_ranks(args):
    """--gpus N without a launcher: one child process per rank.  This parent has not initialised HIP (and never
    will): it imports neither torch.cuda state nor the library, so nothing is re-executed from a GPU process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        code = p.wait()
        rc = rc or code
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nffts", type=int, default=524288, help="FFTs per GPU per step (default: 4 GiB of N=1024 float2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 3 / config 4 measurements (N = 1 extras)")
    ap.add_argument("--no-plain", action="store_true",
                    help="skip the plain-hipMalloc measurement (for rocprofv3 --stats runs: every launch of the roofline kernel is then on the same pair, so the profiler's average duration is comparable with roofline.kernel_ms)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            raise SystemExit(spawn_ranks(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")

    import torch  # first: the HIP runtime torch bundles must be the one the library binds to

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # test hooks (a 1-GPU box cannot host two RCCL ranks): SMFFT_BENCH_DEVICE pins every rank to one
    # device, SMFFT_BENCH_BACKEND=gloo reduces the timings on the CPU instead of over RCCL
    if "SMFFT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SMFFT_BENCH_DEVICE"])
    backend = os.environ.get("SMFFT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    group = None            # the process group of the barriers and reductions: RCCL's, or None = the default (gloo) group
    if world > 1:
        import datetime

        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The default group is gloo on every rank (CPU, TCP to the launcher's store: nothing of the GPUs can keep it from coming up);
        # RCCL is a second group on top, and whether it is used is decided ONCE for all ranks over the first
        # (smfft_amd/sharding.py, agree_on_fast_group) -- never per rank.  Both groups carry timeouts: a rank that never arrives
        # ends the job with an error.  SMFFT_BENCH_REQUIRE_RCCL=1: no gloo fallback at all (exit code 3).
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=600))
        if backend == "nccl":
            from smfft_amd.sharding import agree_on_fast_group

            def rccl_group():
                g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=300))
                probe = torch.ones(1, device=torch.device("cuda", local_rank))
                dist.all_reduce(probe, group=g)          # first collective: RCCL over xGMI is really up
                torch.cuda.synchronize(local_rank)
                if int(probe.item()) != world:
                    raise RuntimeError(f"RCCL all-reduce returned {probe.item()} for {world} ranks")
                return g
            group, rccl_error = agree_on_fast_group(dist, rccl_group)
            if group is None:
                print(f"[bench] rank {rank}: RCCL not used by any rank ({rccl_error or 'another rank could not bring it up'}); timings reduced over gloo", file=sys.stderr)
                if os.environ.get("SMFFT_BENCH_REQUIRE_RCCL", "0") not in ("", "0"):
                    raise SystemExit(3)
                backend = "gloo"

    import smfft_amd as sm  # raises if libsmfft_amd.so is missing

    sm.lib.smfft_set_device(local_rank)
    sm.FFT_init()

    n, nffts = FFT_SIZE, args.nffts
    dev = torch.device("cuda", local_rank)
    stats_dev = dev if backend == "nccl" else torch.device("cpu")
    vram = {"before_any_allocation": vram_used_bytes(torch, dev)}
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    # U[0,1) re/im like the reference harness (SMFFT_CooleyTukey_C2C/FFT.c:141-142); float2 = 2 floats
    t_in = torch.rand((nffts, n, 2), dtype=torch.float32, device=dev, generator=gen)
    nbytes = nffts * n * 8
    # Two buffer pairs, filled with the same batch:
    #  * `pair`:  from the library's allocator smfft_malloc_pair (the buffers of `value` and `roofline`).  On MI355X the
    #    rate of a kernel that reads one buffer and writes another depends on which physical memory the two are
    #    (DESIGN.md section 5); the allocator looks for a good pair within a budget (a quarter of the free memory, 2 s).
    #  * `plain`: two ordinary hipMalloc calls, as in the reference's wrapper (CT:850-853) -> `roofline_plain`.
    pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
    t_alloc = time.perf_counter()
    if sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) != 0:
        raise SystemExit("smfft_malloc_pair failed")
    alloc_s = time.perf_counter() - t_alloc
    pair_info = sm.last_pair_info()
    pair_info["budget"] = "default: a quarter of the free memory, 2 s"
    pair_attempts = [pair_info]
    # The default budget does not reach mixed memory on every box (on about one box in four the first quarter of the
    # memory holds none).  This process owns its device, so it may ask once more with a patient budget; both attempts
    # are reported, `value` / `roofline` are measured on the pair that was kept.
    # (not when several ranks were pinned to ONE device by the test hook: a patient scan would starve the other rank)
    patient_default = "0" if "SMFFT_BENCH_DEVICE" in os.environ else "1"
    # (when the first attempt's output is not `good_enough`: not clearly better, as the target of the whole-pair copy from the
    #  real input, than ordinary memory of one class measured in the same scan -- a device whose free memory starts with a long
    #  run of ONE class, profiles/r03_uniform_box.txt.  The first pair is kept until the second one is there: the better target
    #  of the two -- by the allocator's own timed copy -- is used.)
    if not pair_info["good_enough"] and os.environ.get("SMFFT_BENCH_PATIENT", patient_default) != "0":
        pa2, pb2 = ctypes.c_void_p(), ctypes.c_void_p()
        t_alloc = time.perf_counter()
        if sm.lib.smfft_malloc_pair_budget(nbytes, ctypes.byref(pa2), ctypes.byref(pb2), 0.9, 20000.0) != 0:
            raise SystemExit("smfft_malloc_pair_budget failed")
        alloc_s += time.perf_counter() - t_alloc
        second = sm.last_pair_info()
        second["budget"] = "patient: 90 % of the free memory, 20 s (second attempt: the default budget's output was not clearly better than ordinary memory)"
        pair_attempts.append(second)
        if second["copy_ms"] <= pair_info["copy_ms"]:
            sm.lib.smfft_free_pair(pa.value)
            pa, pb, pair_info = pa2, pb2, second
        else:
            sm.lib.smfft_free_pair(pa2.value)
            pair_info["budget"] += " (kept: the patient attempt's output measured no better)"
    pair_info["kept"] = True             # `value` / `roofline` are measured on this one
    vram["after_smfft_malloc_pair"] = vram_used_bytes(torch, dev)
    p_in = p_out = None
    if not args.no_plain:
        p_in, p_out = sm.DeviceBuffer(nbytes), sm.DeviceBuffer(nbytes)
        sm.lib.smfft_memcpy_d2d(p_in.ptr, t_in.data_ptr(), nbytes)
    sm.lib.smfft_memcpy_d2d(pa.value, t_in.data_ptr(), nbytes)
    xs = torch.view_as_complex(t_in[:4].contiguous()).to(torch.complex128)   # kept for the spot check
    host_in = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        host_in = torch.view_as_complex(t_in).cpu().numpy()                  # the CPU baseline's input = the GPU's
    del t_in
    torch.cuda.empty_cache()
    stream = torch.cuda.current_stream(dev)
    sh = stream.cuda_stream

    def barrier():
        if dist is not None:
            if group is not None:
                dist.barrier(group=group, device_ids=[local_rank])
            else:
                dist.barrier()

    def run_timed(i_ptr, o_ptr, contract):
        """pre-warm, W warm-up steps, K timed steps; returns (wall seconds, average kernel ms from stream events).
        contract = True: the barrier-bracketed region of the driver contract."""
        def step():
            sm.launch("ct", "external", i_ptr, o_ptr, n, nffts, inverse=False, reorder=True, stream=sh)
        prewarm_s = float(os.environ.get("SMFFT_BENCH_PREWARM_S", "1.0"))   # clocks / power state settle (not part of W)
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < prewarm_s:
            for _ in range(20):
                step()
            torch.cuda.synchronize(dev)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize(dev)
        if contract:
            barrier()
            torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)
        torch.cuda.synchronize(dev)
        if contract:
            barrier()
            torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) / args.steps

    # the plain pair first (so that the contract's timed region is the last thing before the reductions)
    plain_wall, plain_kernel_ms = run_timed(p_in.ptr, p_out.ptr, False) if p_in else (0.0, 0.0)
    wall, kernel_ms = run_timed(pa.value, pb.value, True)

    # sanity on the timed output (cheap, outside the timed region): spot-check 4 FFTs of BOTH outputs against torch fp64
    want = torch.fft.fft(xs, dim=-1)
    err = 0.0
    for o_ptr in ([pb.value, p_out.ptr] if p_out else [pb.value]):
        y4 = torch.empty((4, n, 2), dtype=torch.float32, device=dev)
        sm.lib.smfft_memcpy_d2d(y4.data_ptr(), o_ptr, 4 * n * 8)
        ys = torch.view_as_complex(y4).to(torch.complex128)
        err = max(err, (torch.linalg.vector_norm(ys - want) / torch.linalg.vector_norm(want)).item())
    assert err < 5e-7, f"timed output failed the spot check: relL2={err}"

    from smfft_amd.sharding import gather_stats, reduce_stats
    wall_max, kernel_ms_max, ranks_seen = reduce_stats(dist, stats_dev, wall, kernel_ms, 1, group=group)
    plain_wall_max, plain_kernel_ms_max, _ = reduce_stats(dist, stats_dev, plain_wall, plain_kernel_ms, group=group)

    # same-run copy ceiling: the kernel's own access shape without the FFT (outside the timed region)
    def copy_ms(i_ptr, o_ptr):
        for _ in range(3):
            sm.lib.smfft_copy_launch(i_ptr, o_ptr, nffts * n, sh)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(20):
            sm.lib.smfft_copy_launch(i_ptr, o_ptr, nffts * n, sh)
        c1.record(stream)
        torch.cuda.synchronize(dev)
        return c0.elapsed_time(c1) / 20
    pair_copy_ms = copy_ms(pa.value, pb.value)
    # every rank's own outcome, so that a straggler (a rank whose allocator scan found no good output) shows in the line:
    # `value` = world * nffts / max_g(t_g) is hostage to the slowest rank, sum_g(nffts / t_g) is what the ranks did separately
    rows = gather_stats(dist, stats_dev, [wall, kernel_ms, float(pair_info["good_enough"]), pair_copy_ms, float(len(pair_attempts))], group=group)
    per_rank = {"wall_ms_per_step": [r[0] / args.steps * 1e3 for r in rows], "kernel_ms": [r[1] for r in rows],
                "good_enough": [int(r[2]) for r in rows], "copy_ms": [r[3] for r in rows], "attempts": [int(r[4]) for r in rows]}
    value_sum_of_rates = sum(nffts / (r[0] / args.steps) for r in rows)
    plain_copy_ms = copy_ms(p_in.ptr, p_out.ptr) if p_in else None

    # a caller's own plain input with only the OUTPUT taken from the library (smfft_malloc_written_for): the one-line change
    # for code that allocates its own buffers (N = 1 only; outside the contract's timed region, same pre-warm and step count)
    own = None
    if p_in and world == 1:
        w = ctypes.c_void_p()
        if sm.lib.smfft_malloc_written_for(p_in.ptr, nbytes, ctypes.byref(w)) == 0 and w.value:
            w_info = sm.last_pair_info()
            _, w_kernel_ms = run_timed(p_in.ptr, w.value, False)
            own = {"kernel_ms": w_kernel_ms, "copy_ms": copy_ms(p_in.ptr, w.value), "search": w_info}
            sm.lib.smfft_free_written(w.value)

    # the vendor library on the same buffers (N = 1 only; informational)
    vendor = None
    if world == 1 and not args.no_configs:
        v_pair = hipfft_ms(torch, dev, stream, pa.value, pb.value, n, nffts)
        v_plain = hipfft_ms(torch, dev, stream, p_in.ptr, p_out.ptr, n, nffts) if p_in else None
        if v_pair:
            vendor = {"library": "hipFFT (rocFFT) hipfftExecC2C, batched plan, same device buffers, 10 launches after 3 warm-ups",
                      "ms_on_pair": v_pair, "ms_on_plain": v_plain}

    def median_ms(fn, reps=11, warm=3, settle_ms=0.0):
        # settle_ms: the in-LDS kernels are compute-bound and the device's clocks follow the load -- right after memory-bound
        # launches they run 6-7 % slower for the first ~25 ms (profiles/r03_warm_ramp.txt); their figures are the settled ones
        for _ in range(warm):
            fn(None)
        spent = ctypes.c_double(0.0)
        while spent.value < settle_ms:
            fn(ctypes.byref(spent))
        ts = []
        for _ in range(reps):
            t = ctypes.c_double(0.0)
            fn(ctypes.byref(t))
            ts.append(t.value)
        return sorted(ts)[len(ts) // 2]

    SETTLE = 40.0                       # ms of untimed in-LDS launches before the timed ones (see median_ms)
    SETTLE_HBM = 15.0                   # ... and of untimed HBM-bound launches where those follow compute-bound ones (the reverse effect; box dependent)

    def launch_event_ms(fn, reps=7, settle_ms=SETTLE):
        # the same for launch-only entry points (smfft_launch on `stream`): HIP events around each launch
        def one():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1)
        spent = 0.0
        while spent < settle_ms:
            spent += one()
        return sorted(one() for _ in range(reps))[reps // 2]
    # in-LDS `multiple` path on the same buffers (config 3's N=1024 point) on every rank: whole-job figure
    mult = {}
    for reo in (0, 1):
        mult[reo] = median_ms(lambda t, reo=reo: sm.lib.smfft_ct_multiple_benchmark(pa.value, pb.value, n, nffts, 0, reo, t), settle_ms=SETTLE)
    nr_max, re_max, _ = reduce_stats(dist, stats_dev, mult[0], mult[1], group=group)
    mult = {k: {"ms": ms, "FFT/s": world * (nffts // 100) * 100 / (ms * 1e-3), "ms_is": "median of 11 launches after 40 ms of untimed ones (settled clocks), max over ranks", "n_gpus": world}
            for k, ms in (("noreorder", nr_max), ("reorder", re_max))}

    # configs 3 and 4 of BASELINE.json (N = 1 only; FFT_*_benchmark calls = one event-timed launch each, median of 11)
    configs = None
    if world == 1 and not args.no_configs:
        import math
        total = 1 << 29                  # README batches: 4 GiB of float2
        SAT = 8                          # "saturating" batch of the in-LDS path: 8 x the README batch's slots (several rounds of resident waves)
        c3 = {}
        for fn_n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
            bn = min(total // fn_n, nffts * n // fn_n)
            done = (bn // 400 * 400) if fn_n == 32 else (bn // 200 * 200) if fn_n == 64 else (bn // 100 * 100)
            row = {"nFFTs": bn, "FFTs_executed": done}
            for name, reo in (("noreorder", 0), ("reorder", 1)):
                ms = median_ms(lambda t, reo=reo: sm.lib.smfft_ct_multiple_benchmark(pa.value, pb.value, fn_n, bn, 0, reo, t), settle_ms=SETTLE)
                tf = done * 5 * fn_n * math.log2(fn_n) / (ms * 1e-3) / 1e12
                # the multiple path touches only the first nFFTs/100 slots, so a "batch" of SAT x nFFTs stays inside the buffers
                ms_sat = median_ms(lambda t, reo=reo: sm.lib.smfft_ct_multiple_benchmark(pa.value, pb.value, fn_n, SAT * bn, 0, reo, t), reps=7, settle_ms=SETTLE)
                row[name] = {"ms": ms, "FFT/s": done / (ms * 1e-3), "TFLOP/s": tf, "frac_fp32_peak": tf / FP32_PEAK_TFLOPS,
                             "saturating_batch": {"slots_x": SAT, "ms": ms_sat, "FFT/s": SAT * done / (ms_sat * 1e-3),
                                                  "frac_fp32_peak": SAT * done * 5 * fn_n * math.log2(fn_n) / (ms_sat * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}}
            # the same launch the way rounds 1-3 scheduled it (one chain per workgroup, the arbiter's oldest-first order), and the
            # natural-order kernel WITHOUT cross-application fusion (what one call of the device function costs, N >= 64)
            sm.lib.smfft_set_multiple_balance(0)
            sm.lib.smfft_set_multiple_rotation(0)
            ms_old = median_ms(lambda t: sm.lib.smfft_ct_multiple_benchmark(pa.value, pb.value, fn_n, bn, 0, 1, t), reps=7, settle_ms=SETTLE)
            sm.lib.smfft_set_multiple_balance(-1)
            sm.lib.smfft_set_multiple_rotation(-1)
            row["reorder"]["one_chain_per_workgroup_oldest_first"] = {"ms": ms_old, "frac_fp32_peak": done * 5 * fn_n * math.log2(fn_n) / (ms_old * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
            # `unfused` = one image load + one image store per application (smfft_ct_multiple_percall_benchmark): the natural-order planar
            # kernels, and the lane engines of N = 32 (both orderings) and N = 64 without reorder, whose fused loop touches no LDS memory
            # between a chain's first and last application; the planar no-reorder kernels re-read the image as they are (unfused = fused)
            for name, reo in (("reorder", 1), ("noreorder", 0)):
                if reo == 0 and fn_n >= 128:
                    row[name]["unfused"] = {k: row[name][k] for k in ("ms", "FFT/s", "frac_fp32_peak")}
                    continue
                ms_unf = median_ms(lambda t, reo=reo: sm.lib.smfft_ct_multiple_percall_benchmark(pa.value, pb.value, fn_n, bn, 0, reo, t), reps=7, settle_ms=SETTLE)
                row[name]["unfused"] = {"ms": ms_unf, "FFT/s": done / (ms_unf * 1e-3), "frac_fp32_peak": done * 5 * fn_n * math.log2(fn_n) / (ms_unf * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
            c3[str(fn_n)] = row
        # config 4: real N = 2048, 262144 FFTs (2 GiB of reals <-> 2 GiB packed spectrum) -- and the other three real lengths
        # at the same byte count; first halves of the pair for R2C, second halves for C2R
        half = nbytes // 2
        c4 = {}
        for rn in (512, 1024, 2048, 4096):
            rnffts = min((1 << 29) // rn, nffts * n * 2 // rn // 2)      # 2 GiB of reals
            rbytes = rn * rnffts * 4
            row = {"nFFTs": rnffts, "algorithmic_bytes_per_launch": 2 * rbytes}
            # R2C: first half of the read buffer -> first half of the written buffer; the packed spectra are then copied into the
            # read buffer's second half so that C2R, too, READS the read buffer and WRITES the written one (second halves)
            for name, inv, src, dst in (("r2c", 0, pa.value, pb.value), ("c2r", 1, pa.value + half, pb.value + half)):
                if inv:
                    sm.lib.smfft_memcpy_d2d(src, pb.value, rbytes)
                ms = median_ms(lambda t, inv=inv, src=src, dst=dst, rn=rn, rnffts=rnffts: sm.lib.smfft_rc_external_benchmark(src, dst, rn, rnffts, inv, t), settle_ms=SETTLE_HBM)
                gbps = 2 * rbytes / (ms * 1e-3) / 1e9
                row[name] = {"ms": ms, "TB/s": gbps / 1e3, "frac": gbps / HBM_PEAK_GBPS}
            c4[str(rn)] = row
        # in-LDS path (FFT_GPU_R2C_C2R_multiple, RC:367-384: 100 applications per load / store) next to the C2C of the same complex
        # length on the same number of slots (the natural-order Stockham program's `multiple` kernel) -- a loop of its own, behind
        # the HBM-bound rows: on some boxes the memory-bound kernels run 10 % slower for a while after compute-bound launches
        for rn in (512, 1024, 2048, 4096):
            lds_ffts = min((1 << 30) // rn, nffts * n * 2 // rn)            # 4 GiB of reals: as many slots of complex length rn / 2 as config 3 has

            def launch_ms(family, size, inv, rnffts=lds_ffts):
                rc = []
                ms = launch_event_ms(lambda: rc.append(sm.lib.smfft_launch(family, 1, pa.value, pb.value, size, rnffts, inv, 1, sh)))
                if any(rc):
                    raise RuntimeError(f"smfft_launch(family {family}, multiple, {size}) -> {set(rc)}")
                return ms
            lds = {"nFFTs": lds_ffts, "r2c_ms": launch_ms(2, rn, 0), "c2r_ms": launch_ms(2, rn, 1), "c2c_same_complex_length_ms": launch_ms(1, rn // 2, 1)}
            lds["r2c_over_c2c"] = lds["r2c_ms"] / lds["c2c_same_complex_length_ms"] - 1.0
            lds["c2r_over_c2c"] = lds["c2r_ms"] / lds["c2c_same_complex_length_ms"] - 1.0
            c4[str(rn)]["in_lds"] = lds
        c4.update({"real_N": 2048, "nFFTs": c4["2048"]["nFFTs"], "algorithmic_bytes_per_launch": c4["2048"]["algorithmic_bytes_per_launch"],
                   "r2c": c4["2048"]["r2c"], "c2r": c4["2048"]["c2r"]})      # BASELINE's config 4 itself, as in round 2's line
        # config 2 at every length: forward / inverse x reorder / no reorder, whole 4 GiB batch
        c2 = {}
        for fn_n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
            bn = nffts * n // fn_n
            row = {"nFFTs": bn}
            for name, inv, reo in (("forward", 0, 1), ("inverse", 1, 1), ("forward_noreorder", 0, 0), ("inverse_noreorder", 1, 0)):
                ms = median_ms(lambda t, inv=inv, reo=reo: sm.lib.smfft_ct_external_benchmark(pa.value, pb.value, fn_n, bn, inv, reo, t), reps=7, settle_ms=SETTLE_HBM if name == "forward" else 0.0)
                gbps = 2 * fn_n * bn * 8 / (ms * 1e-3) / 1e9
                row[name] = {"ms": ms, "TB/s": gbps / 1e3, "frac": gbps / HBM_PEAK_GBPS}
            c2[str(fn_n)] = row
        # the Stockham program (ST:299-384: + sign, natural order): external and in-LDS path by length
        cst = {}
        for fn_n in (256, 512, 1024, 2048, 4096):
            bn = nffts * n // fn_n
            ms = median_ms(lambda t: sm.lib.smfft_st_external_benchmark(pa.value, pb.value, fn_n, bn, t), reps=7, settle_ms=SETTLE_HBM)
            gbps = 2 * fn_n * bn * 8 / (ms * 1e-3) / 1e9
            msm = median_ms(lambda t: sm.lib.smfft_st_multiple_benchmark(pa.value, pb.value, fn_n, bn, t), reps=7, settle_ms=SETTLE)
            done = bn // 100 * 100
            cst[str(fn_n)] = {"nFFTs": bn, "external": {"ms": ms, "TB/s": gbps / 1e3, "frac": gbps / HBM_PEAK_GBPS},
                              "multiple": {"ms": msm, "FFT/s": done / (msm * 1e-3), "frac_fp32_peak": done * 5 * fn_n * math.log2(fn_n) / (msm * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}}
        # the device functions in the REFERENCE'S OWN contract (blockDim.x = fft_length / 4, two-argument kernels launched in the
        # reference's shape; examples/reference_shape_kernel.hip) next to the library's tiled / compact kernels, same buffers
        cref = None
        try:
            ex = ctypes.CDLL(os.path.join(os.path.dirname(sm.LIB_PATH), "libsmfft_examples.so"))
            vp, ci = ctypes.c_void_p, ctypes.c_int
            ex.smfft_example_reference_shape_ct.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
            ex.smfft_example_reference_shape_st.argtypes = [vp, vp, ci, ci, vp]
            ex.smfft_example_reference_shape_multiple_one.argtypes = [vp, vp, ci, ci, vp]

            def event_ms(fn, reps=7, warm=2, settle_launches=0):
                for _ in range(warm + settle_launches):
                    fn()
                ts = []
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    fn()
                    e1.record(stream)
                    torch.cuda.synchronize(dev)
                    ts.append(e0.elapsed_time(e1))
                return sorted(ts)[len(ts) // 2]
            slots = (nffts // 100)
            gb = 2 * n * nffts * 8 / 1e9
            cref = {"kernels": "SMFFT_DIT_external<P>(in, out) <<<nFFTs, N/4>>> etc. of include/smfft/smfft_device_functions.hpp", "N": n}
            # ct_external_*: the library's two-argument kernel (N >= 256: the block's transform on registers, do_SMFFT_CT_DIT_registers);
            # user_kernel_external_*: a user's kernel around do_SMFFT_CT_DIT (fill LDS, call, drain), examples/reference_shape_kernel.hip
            for key, which_reo, which_kernel in (("ct_external_reorder", 1, 1), ("ct_external_noreorder", 0, 1),
                                                 ("user_kernel_external_reorder", 1, 0), ("user_kernel_external_noreorder", 0, 0)):
                ms = event_ms(lambda r=which_reo, w=which_kernel: ex.smfft_example_reference_shape_ct(pa.value, pb.value, n, nffts, 0, r, w, sh), settle_launches=8)
                tiled = c2[str(n)]["forward" if which_reo else "forward_noreorder"]["ms"]
                cref[key] = {"ms": ms, "TB/s": gb / ms, "frac": gb / ms * 1e3 / HBM_PEAK_GBPS, "ratio_to_tiled": tiled / ms}
            ms = event_ms(lambda: ex.smfft_example_reference_shape_st(pa.value, pb.value, n, nffts, sh))
            cref["stockham_external"] = {"ms": ms, "TB/s": gb / ms, "frac": gb / ms * 1e3 / HBM_PEAK_GBPS, "ratio_to_tiled": cst[str(n)]["external"]["ms"] / ms}
            for key, which, compact in (("ct_multiple_reorder", 0, c3[str(n)]["reorder"]["ms"]), ("ct_multiple_noreorder", 1, c3[str(n)]["noreorder"]["ms"]),
                                        ("stockham_multiple", 2, cst[str(n)]["multiple"]["ms"])):
                ms = event_ms(lambda w=which: ex.smfft_example_reference_shape_multiple_one(pa.value, pb.value, slots, w, sh), settle_launches=30)
                cref[key] = {"ms": ms, "FFT/s": slots * 100 / (ms * 1e-3), "ratio_to_compact": compact / ms}
            # every length: the CT kernels in the reference's shape, external (reorder / no reorder) and in-LDS (README batch)
            ex.smfft_example_reference_shape_ct_multiple.argtypes = [vp, vp, ci, ci, ci, vp]
            ex.smfft_example_reference_shape_ct_multiple_wave64.argtypes = [vp, vp, ci, ci, ci, vp]
            by_len = {}
            for fn_n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
                bn = min(total // fn_n, nffts * n // fn_n)
                per_block = max(1, 128 // fn_n)
                blocks = (bn // 100) // per_block
                row = {}
                for name, reo in (("reorder", 1), ("noreorder", 0)):
                    ms = event_ms(lambda r=reo, fn_n=fn_n, bn=bn: ex.smfft_example_reference_shape_ct(pa.value, pb.value, fn_n, bn, 0, r, 1, sh), reps=5, settle_launches=8)
                    ms_user = event_ms(lambda r=reo, fn_n=fn_n, bn=bn: ex.smfft_example_reference_shape_ct(pa.value, pb.value, fn_n, bn, 0, r, 0, sh), reps=5)
                    tiled = c2[str(fn_n)]["forward" if reo else "forward_noreorder"]["ms"]
                    msm = event_ms(lambda r=reo, fn_n=fn_n, blocks=blocks: ex.smfft_example_reference_shape_ct_multiple(pa.value, pb.value, fn_n, blocks, r, sh), reps=5, settle_launches=20)
                    compact = c3[str(fn_n)][name]["ms"] * (blocks * per_block * 100) / c3[str(fn_n)]["FFTs_executed"]
                    row[name] = {"external_ms": ms, "external_ratio_to_tiled": tiled / ms, "user_kernel_external_ms": ms_user,
                                 "user_kernel_external_ratio_to_tiled": tiled / ms_user, "in_lds_ms": msm,
                                 "in_lds_FFT/s": blocks * per_block * 100 / (msm * 1e-3), "in_lds_ratio_to_compact": compact / msm}
                    if fn_n <= 128:
                        # the wave64-full classes FFT_<N>_..._wave64 (blockDim.x = 64: a whole wavefront per block; upstream's 32-thread
                        # block, CT:586-595, is half of one): same contract, same kernels
                        per64 = 256 // fn_n
                        blocks64 = (bn // 100) // per64
                        ms64 = event_ms(lambda r=reo, fn_n=fn_n, bn=bn: ex.smfft_example_reference_shape_ct(pa.value, pb.value, fn_n, bn, 0, r, 3, sh), reps=5, settle_launches=8)
                        ms64_user = event_ms(lambda r=reo, fn_n=fn_n, bn=bn: ex.smfft_example_reference_shape_ct(pa.value, pb.value, fn_n, bn, 0, r, 2, sh), reps=5)
                        msm64 = event_ms(lambda r=reo, fn_n=fn_n, blocks64=blocks64: ex.smfft_example_reference_shape_ct_multiple_wave64(pa.value, pb.value, fn_n, blocks64, r, sh), reps=5, settle_launches=20)
                        compact64 = c3[str(fn_n)][name]["ms"] * (blocks64 * per64 * 100) / c3[str(fn_n)]["FFTs_executed"]
                        row[name]["wave64"] = {"external_ms": ms64, "external_ratio_to_tiled": tiled / ms64, "user_kernel_external_ms": ms64_user,
                                               "user_kernel_external_ratio_to_tiled": tiled / ms64_user, "in_lds_ms": msm64,
                                               "in_lds_FFT/s": blocks64 * per64 * 100 / (msm64 * 1e-3), "in_lds_ratio_to_compact": compact64 / msm64}
                by_len[str(fn_n)] = row
            cref["by_length"] = by_len
            # The one APPLICATION of the product (reference README.md:10-18: "expected to be called within a GPU kernel"): batched circular
            # convolution y = IFFT(FFT(x) . H) / N of the config-2 batch inside ONE user kernel, three ways -- a user's kernel on the
            # reference's contract (blockDim = N/4, do_SMFFT_CT_DIT forward, a product in shared memory, do_SMFFT_CT_DIT inverse), the
            # same thread shape on the register form of those functions, and the library's register-level engine -- on the pair of
            # `roofline`; fraction of the HBM peak of input + output (examples/reference_shape_kernel.hip, examples/fft_convolution.hip)
            import numpy as np
            hspec = np.zeros(n, np.complex128)
            hspec[:5] = [0.4, 0.3, 0.2, 0.1, -0.05j]
            H = sm.DeviceBuffer.from_host(np.fft.fft(hspec).astype(np.complex64))
            conv = {}
            for key, sym in (("contract", "smfft_example_reference_shape_convolve_1024"), ("contract_registers", "smfft_example_reference_shape_convolve_1024_registers"),
                             ("register_engine", "smfft_example_convolve_1024_registers")):
                fn = getattr(ex, sym)
                fn.argtypes = [vp, vp, vp, ci, vp]
                ms = event_ms(lambda fn=fn: fn(pa.value, H.ptr, pb.value, nffts, sh), reps=9, settle_launches=8)
                conv[key] = {"ms": ms, "TB/s": gb / ms, "frac": gb / ms * 1e3 / HBM_PEAK_GBPS}
            H.free()
            cref["convolution_1024"] = conv
        except (OSError, AttributeError) as e:
            cref = {"error": repr(e)}
        configs = {"timing": "median of 11 (7 where many cases) event-timed launches after 3 warm-ups, buffers of `roofline`; the in-LDS (multiple) figures after a further 40 ms of untimed launches (clocks settled, profiles/r03_warm_ramp.txt), HBM-bound figures that follow in-LDS ones after 15 ms of untimed launches",
                   "config3_schedule": "multiple path: persistent grid of the co-resident workgroups sharing the launch's applications evenly, wave priorities rotating every 2^15 clocks (DESIGN.md section 5.2); `one_chain_per_workgroup_oldest_first` = the schedule of rounds 1-3 on the same kernel",
                   "config2_external_by_length": c2, "config3_multiple": c3, "config4_r2c_c2r_external": c4,
                   "stockham_program": cst, "reference_contract": cref}

    # release everything, then look at the driver's accounting once more: freed VRAM is returned asynchronously
    sm.lib.smfft_free_pair(pa.value)
    if p_in:
        p_in.free()
        p_out.free()
    vram["after_freeing_all_buffers"] = vram_used_bytes(torch, dev)
    time.sleep(1.0)
    vram["one_second_later"] = vram_used_bytes(torch, dev)
    if rank == 0:
        ms_per_step = wall_max / args.steps * 1e3
        total_ffts = nffts * world
        alg_bytes = 2 * n * nffts * 8

        def roof(kms, cms):
            achieved = alg_bytes / (kms * 1e-3) / 1e9
            traffic, traffic_source = measured_traffic()
            return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                    "traffic": traffic, "traffic_source": traffic_source, "kernel": "SMFFT_DIT_external<FFT_1024_forward>", "kernel_ms": kms,
                    "algorithmic_bytes_per_launch": alg_bytes, "copy_ceiling": alg_bytes / (cms * 1e-3) / 1e9, "frac_of_copy": cms / kms}
        out = {
            "metric": "batched_ffts_per_sec_N1024_c2c_fwd_4GiB_external",
            "value": total_ffts / (wall_max / args.steps),
            "unit": "FFT/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"config 2: N={n} C2C forward, reorder, {nffts} FFTs per GPU ({alg_bytes // 2 >> 20} MiB in + out), external path",
                       "fft_size": n, "nffts_per_gpu": nffts, "parallelism": f"batch-split x{world}",
                       "buffers": "smfft_malloc_pair (output built from mixed memory; budget: " + pair_info["budget"] + "); plain hipMalloc figures in roofline_plain / value_plain"},
            "hbm_GBps_per_gpu": alg_bytes / (ms_per_step * 1e-3) / 1e9,
            "roofline": roof(kernel_ms_max, pair_copy_ms),
            "roofline_plain": roof(plain_kernel_ms_max, plain_copy_ms) if p_in else None,
            "roofline_own_input": dict(roof(own["kernel_ms"], own["copy_ms"]), buffers="plain hipMalloc input + smfft_malloc_written_for output", search=own["search"]) if own else None,
            "value_plain": total_ffts / (plain_wall_max / args.steps) if p_in else None,
            "pair_alloc_s": alloc_s,
            "pair_search": pair_info,
            "pair_attempts": pair_attempts,
            "per_rank": per_rank,
            "value_sum_of_rates": value_sum_of_rates,
            "vram_used_bytes": vram,
            "vendor_hipfft": vendor,
            "multiple_path": mult,
            "configs": configs,
            "comm_backend": (backend if world > 1 else None),
            "ranks_seen": ranks_seen,
            "spot_check_relL2": err,
            "device": device_info(torch, dev),
        }
        if host_in is not None:
            out["cpu_baseline"] = cpu_baseline(host_in, os.cpu_count() or 1)
        else:
            out["cpu_baseline"] = None
        # the full record goes to a file; stdout ends with ONE compact strict-JSON line (< 4 KB)
        detail_path = os.environ.get("SMFFT_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        out["detail_file"] = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
        out = _sig(out, 9)
        try:
            with open(detail_path, "w") as f:
                json.dump(out, f, allow_nan=False, indent=1)
            side = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(side):
                with open(os.path.join(side, "bench_detail.json"), "w") as f:
                    json.dump(out, f, allow_nan=False, indent=1)
        except OSError as e:
            print(f"[bench] could not write {detail_path}: {e}", file=sys.stderr)
            out["detail_file"] = None
        print(compact_line(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
