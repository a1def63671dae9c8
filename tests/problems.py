"""Shared small problems for the tests (seeded, sized so the oracle finishes in seconds)."""
import functools

import numpy as np

from ngsamg_amd import fem
from ngsamg_amd._lib import Matrix
from ngsamg_amd.hierarchy import Hierarchy


def to_matrix(p):
    return Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)


@functools.lru_cache(maxsize=None)
def poisson_case(shape, dirichlet="right|top", max_coarse_size=20):
    p = fem.poisson_fast(shape, dirichlet=dirichlet)
    H = Hierarchy(to_matrix(p), p.free, p.coords, dim=len(shape), energy=0, max_coarse_size=max_coarse_size)
    return p, H


@functools.lru_cache(maxsize=None)
def elasticity_case(shape, rotations=False, max_coarse_size=20, first_aaf=None):
    p = fem.elasticity_fast(shape, dirichlet="left", mu=1.0, lam=0.5, rotations=rotations)
    H = Hierarchy(to_matrix(p), p.free, p.coords, dim=len(shape), energy=1, max_coarse_size=max_coarse_size,
                  regularize_cmats=0 if rotations else 1, first_aaf=first_aaf)
    return p, H


def rhs(p, seed=0):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
