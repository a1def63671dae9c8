"""Device-side image builders (csrc/device/devbuild.hpp): SELL images of A and A', the sparse product Q = (I - omega Dinv A) P
and its windowed image are written by kernels from the CSR arrays.  AMGX_VERIFY_IMAGES=1 makes amgx_create build every such image
with the host builders too and compare all arrays bit by bit (it raises on the first difference); the cycle's result must be
bitwise the same as with AMGX_HOST_IMAGES=1."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

from tests.problems import poisson_case, elasticity_case, rhs

pytestmark = pytest.mark.gpu


@contextmanager
def env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    try:
        for k, v in kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _formats(d):
    return [{w: d.matrix_info(l, w) for w in ("A", "Apre", "Q")} for l in range(d.GetNLevels() - 1)]


def _apply(H, p, sm, **e):
    from ngsamg_amd.device import DeviceAMGMatrix
    with env(**e):
        d = DeviceAMGMatrix(H, device=0, sm_type=sm)
    b = rhs(p, 3)
    x = np.full(b.size, np.nan)
    d.Mult(b, x)
    return x, d


# small levels take the one-thread-per-row form only with the lane cap (AMGX_SELL_MAX_LANES = 1, a test hook of upload_matrix)
SMALL = dict(AMGX_DEV_IMAGES_MIN_ROWS=0, AMGX_SELL_MAX_LANES=1)


@pytest.mark.parametrize("shape,diri,mcs", [((33, 33), "left|top", 5), ((224, 224), "left|top", 5), ((17, 17, 17), "right|top", 20),
                                            ((9, 30, 13), ".*", 20), ((41, 37, 29), "right|top", 10), ((7, 5, 3), "", 4)])
@pytest.mark.parametrize("sm", ["jacobi", "gs", "hgs"])
def test_device_built_images_equal_host_built_images(shape, diri, mcs, sm):
    p, H = poisson_case(shape, diri, mcs)
    xv, dv = _apply(H, p, sm, AMGX_VERIFY_IMAGES=1, **SMALL)        # raises if any array of any image differs
    xh, dh = _apply(H, p, sm, AMGX_HOST_IMAGES=1, **SMALL)
    xd, dd = _apply(H, p, sm, **SMALL)
    assert np.array_equal(xd, xh) and np.array_equal(xv, xh)
    assert _formats(dd) == _formats(dh)


@pytest.mark.parametrize("sm", ["jacobi", "hgs"])
def test_device_built_images_million_rows_default_settings(sm):
    """the path as cfg 2 takes it (no hooks): 102^3 = 1.06 M rows, level 0 built on the device and verified against the host
    (hgs: rows of up to 27 entries, i.e. two lanes per row in the colour-sorted images)"""
    p, H = poisson_case((102, 102, 102), "right|top", 50)
    xv, dv = _apply(H, p, sm, AMGX_VERIFY_IMAGES=1)
    xh, dh = _apply(H, p, sm, AMGX_HOST_IMAGES=1)
    assert np.array_equal(xv, xh)
    assert _formats(dv) == _formats(dh)
    fi = _formats(dv)
    assert fi[0]["A"]["fmt"] == "sell"
    if sm == "jacobi":
        assert fi[0]["Apre"]["fmt"] == "sell" and fi[0]["Q"]["fmt"] == "sellwin"


@pytest.mark.parametrize("shape,rot", [((9, 8, 7), False), ((10, 9, 8), True), ((24, 23), False), ((33, 18, 18), True)])
@pytest.mark.parametrize("sm", ["jacobi", "hgs"])
def test_device_built_block_images_equal_host_built_images(shape, rot, sm):
    """square-block levels (2 x 2 ... 6 x 6): the BSELL image of A and the four images of the block-hybrid Gauss-Seidel smoother
    (off / lower-in / upper-in / rest) gathered on the device from one block-CSR upload"""
    p, H = elasticity_case(shape, rotations=rot, max_coarse_size=10)
    xv, dv = _apply(H, p, sm, AMGX_VERIFY_IMAGES=1, AMGX_DEV_IMAGES_MIN_ROWS=0)
    xh, dh = _apply(H, p, sm, AMGX_HOST_IMAGES=1)
    xd, dd = _apply(H, p, sm, AMGX_DEV_IMAGES_MIN_ROWS=0)
    assert np.array_equal(xd, xh) and np.array_equal(xv, xh)
    assert [dd.matrix_info(l, "A") for l in range(dd.GetNLevels() - 1)] == [dh.matrix_info(l, "A") for l in range(dh.GetNLevels() - 1)]


def test_device_built_block_images_default_settings():
    """the path as cfg 3 / cfg 5 take it (no hooks): 30^3 nodes with rotations = 162 k scalar rows"""
    p, H = elasticity_case((30, 30, 30), rotations=True, max_coarse_size=20)
    xv, dv = _apply(H, p, "hgs", AMGX_VERIFY_IMAGES=1)
    xh, dh = _apply(H, p, "hgs", AMGX_HOST_IMAGES=1)
    assert np.array_equal(xv, xh)
    assert dv.matrix_info(0, "A") == dh.matrix_info(0, "A") and dv.matrix_info(0, "A")["fmt"] == "bsell"


@pytest.mark.parametrize("sm", ["jacobi", "hgs"])
def test_device_built_images_on_rank_partitioned_levels(sm):
    """levels with ghost columns (n_cols > n_rows: two virtual ranks): the device builders must reproduce the host images there too
    (rest / off images carry [owned | ghost] column ids), and the collective cycle must equal the oracle"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    R = 2
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_poisson_owned(r, pg, (18, 16, 16)) for r in range(R)]
    with env(AMGX_VERIFY_IMAGES=1, **SMALL):
        amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=200, device=0, max_coarse_size=10, sm_type=sm,
                               **({"hgs_block_rows": 256} if sm == "hgs" else {}))
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    ref = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv)).apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("shape,cap", [((26, 25, 24), None), ((40, 38, 30), None), ((40, 38, 30), 1500), ((40, 38, 30), 600)])
def test_device_built_local_window_images_equal_host_built_ones(shape, cap):
    """dev_build_lw (window lists by a bitmap in LDS) against build_sell_lw / build_sell_lw_windowed: SELL arrays, unit offsets, column
    lists and the row order of the windows, with chunks beyond the capacity (AMGX_LW_TEST_CAP) keeping their global columns"""
    p, H = poisson_case(shape, "right|top", 20)
    e = dict(AMGX_DEV_IMAGES_MIN_ROWS=0, AMGX_LW_MIN_ROWS=300, AMGX_NO_DENSE_TAIL=1)
    if cap:
        e["AMGX_LW_TEST_CAP"] = cap
    xv, dv = _apply(H, p, "jacobi", AMGX_VERIFY_IMAGES=1, **e)       # raises if any array of any image differs
    used = [l for l in range(1, H.n_levels - 1) if dv.matrix_info(l, "ApreLW")["fmt"] == "sell-lw"]
    assert used and dv.matrix_info(0, "QLW")["fmt"] == "sell-lw"
    xh, dh = _apply(H, p, "jacobi", AMGX_HOST_LW=1, **e)
    assert np.array_equal(xv, xh)
    assert _formats(dv) == _formats(dh)
    for l in range(H.n_levels):
        for w in ("ApreLW", "QLW"):
            assert dv.matrix_info(l, w) == dh.matrix_info(l, w)
