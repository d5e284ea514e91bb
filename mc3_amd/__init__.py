"""mc3_amd -- mc3's per-iteration chain update on AMD MI355X (gfx950).

A from-scratch, MI355X-native implementation of ONE hot path of pcubillos/mc3
(v3.2.1): DEMC / snooker / MRW proposal -> model -> Gaussian or wavelet chisq ->
Metropolis accept -> Z append (mc3/chain.py:183-299, src_c/_chisq.c,
src_c/_dwt.c), behind mc3's own ``sample()`` / ``mcmc()`` API.  Host code is
plain Python + NumPy over a ctypes C-ABI (include/mc3hip.h); all numerics run
in hand-written HIP kernels (mc3_amd/csrc).  There is no CPU fallback.
"""
__version__ = '0.1.0'

from .engine import DeviceChains, dwt_daub4, comm_unique_id  # noqa: E402,F401
from .sampler_driver import sample  # noqa: E402,F401
from .mcmc_driver import mcmc  # noqa: E402,F401
from .fit_driver import fit  # noqa: E402,F401
from . import models, stats, utils, parallel  # noqa: E402,F401
