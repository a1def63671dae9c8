"""Legacy module name of the reference's Python package (its tests still ``import ngs_amg`` and register
``"ngs_amg.h1_scal"``, reference tests/h1/simple/test_2d_lo.py:1,10).  Same objects as ngsamg_amd.NgsAMG."""
from .NgsAMG import *          # noqa: F401,F403
from .NgsAMG import (h1_scal, h1_2d, h1_3d, elast_2d, elast_3d, AMGMatrix, DOFMap, BaseDOFMapStep, BaseSmoother,  # noqa: F401
                     ProxySmoother, CreateJacobiSmoother, CreateHybridGSS, Preconditioner)
