#!/usr/bin/env python3
"""bench.py -- headline benchmark of smfft_amd on MI355X.

Metric (BASELINE.json): batched FFTs/s + achieved HBM GB/s, N=1024 C2C forward, 4 GB input,
1/2/4/8 GPU.  A "step" is one pass of the hot path -- one FFT_external_benchmark-equivalent launch
(SMFFT_DIT_external<FFT_1024_forward>) over a batch of 524288 FFTs (4 GiB in, 4 GiB out) that is
already resident in HBM (config 2 of BASELINE.json).  With N GPUs every rank owns its own 4 GiB
batch (config 5: weak scaling, no data-path collective; RCCL only reduces the timings).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline     : the dominant kernel against the HBM roofline (8.0 TB/s datasheet peak); `achieved`
                 = algorithmic bytes per launch (2 * N * nFFTs * 8 B) / average launch duration
                 measured with events on the launch stream over the timed region.
  cpu_baseline : FFTW-API batched C2C (MKL's FFTW3 interface; real FFTW is not in the image) on the
                 host cores of this box, bounded sample; falls back to the oracle's C restatement.
PyTorch is plumbing only (device memory, stream, torch.distributed); the transform is the HIP
library behind the C ABI (include/smfft.h).  There is no CPU fallback in the timed path.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FFT_SIZE = 1024
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def cpu_baseline(n, sample_ffts, threads):
    """FFTW-API batched plan on the host (oracle/fftw_baseline.so); fallback: oracle restatement."""
    import numpy as np

    rng = np.random.default_rng(1)
    x = (rng.random((sample_ffts, n), dtype=np.float32) + 1j * rng.random((sample_ffts, n), dtype=np.float32)).astype(np.complex64)
    out = np.empty_like(x)
    fp = ctypes.POINTER(ctypes.c_float)
    res = None
    try:
        fb = ctypes.CDLL(os.path.join(ROOT, "oracle", "fftw_baseline.so"))
        fb.fftw_baseline_init.argtypes = [ctypes.c_int]
        fb.fftw_baseline_backend.restype = ctypes.c_char_p
        fb.fftw_baseline_c2c.restype = ctypes.c_double
        fb.fftw_baseline_c2c.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        fb.fftw_baseline_c2c_sliced.restype = ctypes.c_double
        fb.fftw_baseline_c2c_sliced.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        # one single-threaded plan per batch slice, slices run concurrently (oracle/fftw_baseline.c);
        # a few thread counts are tried for a bounded time, `cores` = the count that won
        cands = sorted({c for c in (1, 8, 32, 64, 128, threads // 2, threads) if 1 <= c <= threads})
        best_all, best_thr = 1e30, 0
        t_start = time.time()
        if fb.fftw_baseline_init(1):
            for thr in cands:
                if time.time() - t_start > 25.0:
                    break
                t = fb.fftw_baseline_c2c_sliced(x.ctypes.data_as(fp), out.ctypes.data_as(fp), n, sample_ffts, 0, 5, thr)
                if 0 < t < best_all:
                    best_all, best_thr = t, thr
        if best_all < 1e29:
            res = {"value": sample_ffts / best_all, "unit": "FFT/s", "cores": best_thr, "kind": "port",
                   "impl": fb.fftw_baseline_backend().decode() + " fftwf_plan_many_dft (FFTW_ESTIMATE) per batch slice, out of place, slices on pthreads; thread counts tried " + str(cands),
                   "sample": f"N={n} C2C forward, {sample_ffts} FFTs ({x.nbytes >> 20} MiB in), best of 5 rounds",
                   "host_cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    except OSError:
        pass
    if res is None:
        olib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        olib.oracle_ct_c2c_f32.argtypes = [fp, fp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int]
        best = 1e30
        for _ in range(3):
            t0 = time.time()
            olib.oracle_ct_c2c_f32(x.ctypes.data_as(fp), out.ctypes.data_as(fp), n, sample_ffts, 0, 1)
            best = min(best, time.time() - t0)
        res = {"value": sample_ffts / best, "unit": "FFT/s", "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)),
               "kind": "port", "impl": "oracle/smfft_oracle.c radix-2 restatement, OpenMP over FFTs",
               "sample": f"N={n} C2C forward, {sample_ffts} FFTs, best of 3"}
    res["GB/s"] = res["value"] * 2 * n * 8 / 1e9
    return res


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/*_pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE x2 per the
    gfx950 correction, calibrated on the same access shape).  None if no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"]["void SMFFT_DIT_external<FFT_1024_forward>"]
        return k["hbm_bytes_per_launch"]
    except (KeyError, ValueError):
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
    # clocks / partition modes of the first amdgpu device the driver exposes (best effort, read-only sysfs)
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
            if want is not None and not bdf.startswith(want):
                continue
            sysfs = {}
            for name in ("current_memory_partition", "current_compute_partition", "pp_dpm_mclk", "pp_dpm_sclk", "pp_dpm_fclk",
                         "mem_info_vram_total", "mem_info_vram_used", "power_dpm_force_performance_level"):
                try:
                    sysfs[name] = open(os.path.join(d, name)).read().strip().replace("\n", " | ")
                except OSError:
                    pass
            for cap in glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_cap")):
                try:
                    sysfs["power1_cap_uW"] = open(cap).read().strip()
                except OSError:
                    pass
            if sysfs:
                sysfs["pci"] = bdf
                info["sysfs"] = sysfs
                break
    except Exception as e:
        info["sysfs_error"] = repr(e)
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nffts", type=int, default=524288, help="FFTs per GPU per step (default: 4 GiB of N=1024 float2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()          # first collective: RCCL over xGMI is really up
            except Exception as e:      # the timings can still be reduced on the CPU
                print(f"[bench] RCCL unavailable ({type(e).__name__}: {e}); reducing timings over gloo", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
                dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend=backend)

    import smfft_amd as sm  # raises if libsmfft_amd.so is missing

    sm.lib.smfft_set_device(local_rank)
    sm.FFT_init()

    n, nffts = FFT_SIZE, args.nffts
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    # U[0,1) re/im like the reference harness (SMFFT_CooleyTukey_C2C/FFT.c:141-142); float2 = 2 floats
    t_in = torch.rand((nffts, n, 2), dtype=torch.float32, device=dev, generator=gen)
    # The batch lives in plain hipMalloc'ed buffers, as in the reference's wrapper (cudaMalloc,
    # CT:850-853).  Measured (tools/alloc_probe.py): writing into a 4 GiB block of torch's caching
    # allocator is 6-7 % slower than into a hipMalloc'ed one (1.53 vs 1.43 ms per launch).
    nbytes = nffts * n * 8
    # Buffer placement (profiles/r01_chunk_map.txt): on MI355X the rate of a kernel that reads
    # one buffer and writes another depends on which physical memory the two are (1.31 ... 1.55 ms for this batch).
    # smfft_malloc_pair() -- the allocator the library's own L3 wrappers use -- allocates buffer-sized chunks over
    # the free memory, times candidate (input, output) pairs with a stream copy and keeps the fastest.  For
    # transparency the same launches are also timed on two plain allocations.
    pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
    if sm.lib.smfft_malloc_pair(nbytes, ctypes.byref(pa), ctypes.byref(pb)) != 0:
        raise SystemExit("smfft_malloc_pair failed")

    class _Raw:
        def __init__(self, ptr):
            self.ptr = ptr
    b_in, b_out = _Raw(pa.value), _Raw(pb.value)
    sm.lib.smfft_memcpy_d2d(b_in.ptr, t_in.data_ptr(), nbytes)
    xs = torch.view_as_complex(t_in[:4].contiguous()).to(torch.complex128)   # kept for the spot check
    del t_in
    torch.cuda.empty_cache()

    def _probe(i_ptr, o_ptr):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0 = torch.cuda.current_stream(dev)
        for _ in range(2):
            sm.launch("ct", "external", i_ptr, o_ptr, n, nffts, stream=s0.cuda_stream)
        e0.record(s0)
        for _ in range(8):
            sm.launch("ct", "external", i_ptr, o_ptr, n, nffts, stream=s0.cuda_stream)
        e1.record(s0)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / 8

    # placement telemetry: the same launches on two plain allocations (what two hipMalloc calls give a caller who does
    # not use smfft_malloc_pair), made after the pair and released again
    plain_ms = None
    try:
        p_in, p_out = sm.DeviceBuffer(nbytes), sm.DeviceBuffer(nbytes)
        sm.lib.smfft_memset(p_in.ptr, 0, nbytes)
        plain_ms = round(_probe(p_in.ptr, p_out.ptr), 4)
        p_in.free()
        p_out.free()
    except MemoryError:
        pass
    placement = {"paired_ms": round(_probe(b_in.ptr, b_out.ptr), 4), "plain_hipmalloc_ms": plain_ms}

    class _Ptr:                      # tiny adaptor so the rest of the script reads like tensor code
        def __init__(self, buf):
            self.buf = buf

        def data_ptr(self):
            return self.buf.ptr
    d_in, d_out = _Ptr(b_in), _Ptr(b_out)
    stream = torch.cuda.current_stream(dev)
    sh = stream.cuda_stream

    def step():
        sm.launch("ct", "external", d_in.data_ptr(), d_out.data_ptr(), n, nffts, inverse=False, reorder=True, stream=sh)

    def barrier():
        if dist is not None:
            dist.barrier()

    # device pre-warm (not part of the W warm-up steps of the contract): clocks / power state settle
    prewarm_s = float(os.environ.get("SMFFT_BENCH_PREWARM_S", "1.0"))
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < prewarm_s:
        for _ in range(20):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch duration on the launch stream

    # sanity on the timed output (cheap, outside the timed region): spot-check 4 FFTs against torch fp64
    y4 = torch.empty((4, n, 2), dtype=torch.float32, device=dev)
    sm.lib.smfft_memcpy_d2d(y4.data_ptr(), b_out.ptr, 4 * n * 8)
    ys = torch.view_as_complex(y4).to(torch.complex128)
    err = (torch.linalg.vector_norm(ys - torch.fft.fft(xs, dim=-1)) / torch.linalg.vector_norm(torch.fft.fft(xs, dim=-1))).item()
    assert err < 5e-7, f"timed output failed the spot check: relL2={err}"

    from smfft_amd.sharding import reduce_stats
    wall_max, kernel_ms_max, _ = reduce_stats(dist, dev if backend == "nccl" else torch.device("cpu"), wall, kernel_ms)

    # same-run copy ceiling: the kernel's own access shape without the FFT (outside the timed region)
    for _ in range(3):
        sm.lib.smfft_copy_launch(d_in.data_ptr(), d_out.data_ptr(), nffts * n, sh)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record(stream)
    for _ in range(20):
        sm.lib.smfft_copy_launch(d_in.data_ptr(), d_out.data_ptr(), nffts * n, sh)
    c1.record(stream)
    torch.cuda.synchronize(dev)
    copy_ms = c0.elapsed_time(c1) / 20

    # in-LDS `multiple` path on the same buffers (config 3's N=1024 point), informational
    mult = {}
    for reo in (0, 1):
        tm = ctypes.c_double(0.0)
        for _ in range(3):
            sm.lib.smfft_ct_multiple_benchmark(d_in.data_ptr(), d_out.data_ptr(), n, nffts, 0, reo, None)
        reps = 10
        for _ in range(reps):
            sm.lib.smfft_ct_multiple_benchmark(d_in.data_ptr(), d_out.data_ptr(), n, nffts, 0, reo, ctypes.byref(tm))
        mult["reorder" if reo else "noreorder"] = tm.value / reps
    # whole-job figure for the multiple path too: every rank ran it on its own shard at the same time
    nr_max, re_max, _ = reduce_stats(dist, dev if backend == "nccl" else torch.device("cpu"), mult["noreorder"], mult["reorder"])
    mult = {k: {"ms": ms, "FFT/s": world * (nffts // 100) * 100 / (ms * 1e-3), "ms_is": "max over ranks", "n_gpus": world}
            for k, ms in (("noreorder", nr_max), ("reorder", re_max))}

    if rank == 0:
        ms_per_step = wall_max / args.steps * 1e3
        total_ffts = nffts * world
        alg_bytes = 2 * n * nffts * 8
        achieved = alg_bytes / (kernel_ms_max * 1e-3) / 1e9
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
                       "fft_size": n, "nffts_per_gpu": nffts, "parallelism": f"batch-split x{world}"},
            "hbm_GBps_per_gpu": alg_bytes / (ms_per_step * 1e-3) / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": measured_traffic(), "kernel": "SMFFT_DIT_external<FFT_1024_forward>", "kernel_ms": kernel_ms_max,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "copy_ceiling": alg_bytes / (copy_ms * 1e-3) / 1e9, "frac_of_copy": copy_ms / kernel_ms_max},
            "multiple_path": mult,
            "comm_backend": (backend if world > 1 else None),
            "buffer_placement_probe": placement,
            "spot_check_relL2": err,
            "device": device_info(torch, dev),
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(n, 131072, threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
