"""ngsamg_amd -- MI355X-native V-cycle apply path behind the NgsAMG interface.

    ngsamg_amd.NgsAMG / ngsamg_amd.ngs_amg   Python surface of the reference's pybind module
    ngsamg_amd.device.DeviceAMGMatrix        thin handle over the C ABI (include/amgx.h, HIP kernels)
    ngsamg_amd.hierarchy.Hierarchy           host setup (include/amgh.h)
    ngsamg_amd.krylov.CGSolver               device-resident PCG (NGSolve CGSolver stand-in)
    ngsamg_amd.fem                           synthetic P1 problems (stand-in for the calling FEM package)
"""
from ._lib import Matrix, NgsAMGError  # noqa: F401

__all__ = ["Matrix", "NgsAMGError"]
