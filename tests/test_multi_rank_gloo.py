"""world_size-2 test of the N>1 path on CPU (gloo): contiguous batch split, identical work per
rank, no data-path collective, MAX-reduced timings / SUM-reduced error counts.
The per-rank transform is played by the CPU oracle here (no GPU in this container); on GPUs
bench.py runs the same logic with the HIP kernel and RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nffts, n, outdir):
    sys.path.insert(0, ROOT)
    import ctypes

    from smfft_amd.sharding import reduce_stats, shard_range
    from tests import oracle_api as oa

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    dp = ctypes.POINTER(ctypes.c_double)
    lib.oracle_ct_c2c_f64.argtypes = [dp, dp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(99)                       # same batch on every rank
    x = rng.standard_normal((nffts, n)) + 1j * rng.standard_normal((nffts, n))
    first, count = shard_range(nffts, rank, world)
    y = oa.ct_c2c(lib, x[first:first + count], 0, 1, "f64")
    np.save(os.path.join(outdir, f"shard{rank}.npy"), y)
    wall, kernel, errs = reduce_stats(dist, torch.device("cpu"), wall_s=1.0 + rank, kernel_ms=10.0 - rank, errors=rank + 1)
    assert wall == float(world) and kernel == 10.0 and errs == world * (world + 1) // 2
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_batch_split(tmp_path):
    world, nffts, n = 2, 11, 256
    mp.spawn(_worker, args=(world, _free_port(), nffts, n, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"shard{r}.npy") for r in range(world)])
    rng = np.random.default_rng(99)
    x = rng.standard_normal((nffts, n)) + 1j * rng.standard_normal((nffts, n))
    np.testing.assert_allclose(got, np.fft.fft(x, axis=-1), atol=1e-9)


def _bench(args, env_extra=None, timeout=600):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_flag_never_silently_runs_one_rank():
    """`bench.py --gpus N` either runs N ranks or fails: under a launcher with a different WORLD_SIZE it exits non-zero
    before touching anything; without a launcher it spawns the N ranks itself from a parent that never initialises the
    GPU (here, without a GPU, every child refuses to run -- twice for --gpus 2 -- and the parent reports the failure)."""
    p = _bench(["--gpus", "8"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode != 0 and "--gpus 8" in p.stderr and "WORLD_SIZE=2" in p.stderr and p.stdout.strip() == ""
    if torch.cuda.is_available():
        return   # the spawn path itself is exercised on the GPU box (tests/test_gpu_parity.py::test_bench_two_ranks_on_one_device)
    p = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert p.stderr.count("bench.py needs a GPU") == 2


def _agree_worker(rank, world, port, fail_on, outdir):
    sys.path.insert(0, ROOT)
    from smfft_amd.sharding import agree_on_fast_group, gather_stats, reduce_stats

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def try_fast():
        # (a second gloo group plays RCCL's part on the CPU: what is tested is that the ranks end up on the SAME group)
        g = dist.new_group(backend="gloo")
        if rank in fail_on:
            raise RuntimeError("this rank could not bring the fast communicator up")
        return g
    group, error = agree_on_fast_group(dist, try_fast)
    assert (error is not None) == (rank in fail_on)
    # whatever was agreed, the reductions of bench.py run on it and see every rank
    wall, kernel, errs = reduce_stats(dist, torch.device("cpu"), 1.0 + rank, 5.0, 1, group=group)
    rows = gather_stats(dist, torch.device("cpu"), [float(rank)], group=group)
    assert wall == float(world) and errs == world and [r[0] for r in rows] == [float(r) for r in range(world)]
    with open(os.path.join(outdir, f"agreed{rank}.txt"), "w") as f:
        f.write("fast" if group is not None else "default")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_on", [(), (1,), (0, 1)])
def test_ranks_agree_on_the_communicator(tmp_path, fail_on):
    """bench.py decides RCCL-or-gloo ONCE for all ranks (round 5 let every rank fall back on its own: RCCL up on some ranks only would
    have hung the job): if any rank cannot bring the fast group up, no rank uses it."""
    world = 2
    mp.spawn(_agree_worker, args=(world, _free_port(), tuple(fail_on), str(tmp_path)), nprocs=world, join=True)
    verdicts = {open(tmp_path / f"agreed{r}.txt").read() for r in range(world)}
    assert verdicts == ({"fast"} if not fail_on else {"default"})
