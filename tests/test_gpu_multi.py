"""Rank-partitioned V-cycle on SEVERAL GPUs: one process per GPU, RCCL over xGMI through the C ABI (amgx_comm_create /
amgx_dist_apply), against the serial CPU oracle on the assembled global hierarchy.  Skipped on boxes with fewer GPUs than
ranks (counting devices does not initialise the GPU, so the child ranks start from a clean process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    import torch
    return torch.cuda.device_count()


def _run(*argv, timeout=900):
    env = dict(os.environ, NGSAMG_CHECK_TIMEOUT=str(max(30, timeout - 90)))     # the launcher's watchdog kills its ranks first
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_rccl_check.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0 and "RCCL CHECK PASSED" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("sm,extra", [("jacobi", []), ("jacobi", ["--no-fold"]), ("gs", []), ("bgs", [])])
def test_rccl_ranks_match_serial_oracle(world, sm, extra):
    """cfg 4's arrangement (2 x 2 x 2 boxes at world 8; 2 x 1 x 1, 2 x 2 x 1 below) at a size the oracle finishes in seconds"""
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs")
    _run("--world", str(world), "--box", "28", "--sm", sm, *extra)


@pytest.mark.parametrize("world", [2, 8])
def test_rccl_slab_partition_as_in_bench(world):
    """bench.py --gpus N stacks the ranks as slabs (N, 1, 1)"""
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs")
    _run("--world", str(world), "--box", "24", "--pgrid", "slab")


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("elast", ["3", "6"])
def test_rccl_elasticity_ranks_match_serial_oracle(world, elast):
    """cfg 5's shape on several GPUs: rank-partitioned elasticity (block size 3 or 6), halo messages of block vectors"""
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs")
    _run("--world", str(world), "--box", "12", "--elast", elast, "--dmin", "50")


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("sm", ["hgs"])
def test_rccl_block_hybrid_gauss_seidel_ranks_match_serial_oracle(world, sm):
    """sm_type = hgs over RCCL: scalar levels (gsb_sweep_kernel) and 6 x 6 elasticity levels (bgsb_sweep_kernel) swept as boundary
    blocks, exchange, interior blocks; the whole cycle replayed from its graph"""
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs")
    _run("--world", str(world), "--box", "28", "--sm", sm)
    _run("--world", str(world), "--box", "14", "--elast", "6", "--sm", sm, "--dmin", "200")
