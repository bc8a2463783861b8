"""The N > 1 path: spp split across ranks + one all-reduce of the accumulation buffer, rehearsed
with world_size 2 over gloo on CPU (the CPU checker stands in for the renderer here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ray_tracing_amd import scenarios
from ray_tracing_amd.distributed import render_sharded, sample_range


def test_sample_ranges_partition():
    for total in (0, 1, 7, 1024, 16384, 1027):
        for n in (1, 2, 3, 4, 8):
            ranges = [sample_range(total, r, n) for r in range(n)]
            assert sum(c for _, c in ranges) == total
            pos = 0
            for first, count in ranges:
                assert first == pos
                pos += count
            assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1
    with pytest.raises(ValueError):
        sample_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, total_spp, out_path):
    from tests.oracle_ffi import OracleWorld
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    W = H = 20
    ow = OracleWorld(scenarios.cornell_smoke())
    accum = torch.zeros((H, W, 3), dtype=torch.float32)
    buf = accum.numpy()

    def render(first, count):
        ow.render(W, H, count, seed=4, first_sample=first, threads=2, accum=buf)

    first, count = render_sharded(render, lambda: dist.all_reduce(accum), total_spp, rank, world_size)
    assert (first, count) == sample_range(total_spp, rank, world_size)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_spp_split_equals_single_render(tmp_path):
    from tests.oracle_ffi import OracleWorld
    total_spp = 7
    out = str(tmp_path / "sharded.npy")
    mp.spawn(_worker, args=(2, _free_port(), total_spp, out), nprocs=2, join=True)
    sharded = np.load(out)
    single, _ = OracleWorld(scenarios.cornell_smoke()).render(20, 20, total_spp, seed=4, threads=2)
    # same pixel-samples, different f32 summation order
    np.testing.assert_allclose(sharded, single, rtol=1e-5, atol=1e-6)


def _run_bench(*flags):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_weak_and_strong():
    """`bench.py --gpus 2` without a launcher forks two fresh ranks, joins them over the process group and prints
    ONE line with n_gpus 2 (rehearsed on CPUs: gloo, no rendering). Every rank adds 1 per sample it owns, so the
    all-reduced image divided by the total sample count is exactly 1 when the shares cover the range once."""
    weak = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--warmup", "1", "--width", "32",
                      "--height", "16", "--spp", "4")
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["dry"] is True
    # what the collective saw, reported by the ranks themselves (n_gpus alone is copied from the command line)
    col = weak["collective"]
    assert col["world_size"] == 2 and col["ranks_seen"] == [0, 1] and col["backend"] == "gloo"
    assert col["bytes"] == 32 * 16 * 3 * 4 and col["allreduce_ms"] > 0 and col["spp_per_rank"] == [4, 4]
    assert len(weak["per_rank_mpaths"]) == 2 and all(v > 0 for v in weak["per_rank_mpaths"])
    assert weak["mean_radiance"] == 1.0
    assert weak["value"] > 0 and abs(weak["value"] * weak["ms_per_step"] * 1e3 - 32 * 16 * 4 * 2) < 1e-3 * 32 * 16 * 8
    strong = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--strong", "--steps", "1", "--warmup", "0",
                        "--width", "32", "--height", "16", "--spp", "5")
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong" and strong["mean_radiance"] == 1.0
    assert strong["collective"]["spp_per_rank"] == [3, 2]
    assert abs(strong["value"] * strong["ms_per_step"] * 1e3 - 32 * 16 * 5) < 1e-3 * 32 * 16 * 5


def test_bench_refuses_a_launcher_with_another_world_size():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry", "--backend", "gloo"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_bench_ends_quickly_when_one_rank_dies():
    """A rank that ends before the rendezvous (no such GPU, a scene error, a failed RCCL start) must not leave the
    others waiting in `init_process_group` until its timeout: the launcher stops them and returns the error."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MRT_BENCH_FAIL_RANK"] = "1"
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry",
                        "--steps", "1", "--warmup", "0", "--width", "32", "--height", "16", "--spp", "4"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert time.monotonic() - t0 < 20.0
    assert "rank 1" in p.stderr and "MRT_BENCH_FAIL_RANK" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]     # no result line from a failed run
    with open(os.path.join(root, "gpurun_out", "rank1.err")) as f:
        assert "MRT_BENCH_FAIL_RANK" in f.read()


def test_bench_as_ranks_of_torch_distributed_run():
    """The other way the benchmark is started for N > 1: `python -m torch.distributed.run ... bench.py --gpus N`, one
    process per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment (rehearsed with gloo, no GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--backend", "gloo", "--dry", "--width", "32", "--height", "16", "--spp", "4"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                 # rank 0 alone prints the line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["collective"]["world_size"] == 2 and out["collective"]["ranks_seen"] == [0, 1]
    assert out["mean_radiance"] == 1.0
