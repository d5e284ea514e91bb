"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the mc3 (v3.2.1) chain-update hot path, used only as a
checker: by tests/, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``mc3_amd/`` imports, links or executes
anything from here; the product path fails loudly when its HIP library is
missing.

Contents
--------
oracle_kernels.c      reference-faithful chisq / residuals / priors / D4 DWT /
                      wavelet chisq (src_c/_chisq.c, src_c/_dwt.c,
                      src_c/include/{stats,wavelet}.h)
kernels.py            ctypes wrappers with mc3.stats' call signatures
chain_sequential.py   the reference's own SEQUENTIAL chain loop and hub
                      initialisation (mc3/chain.py, mc3/mcmc_driver.py) on
                      NumPy's legacy MT19937 stream -- pinned bit-for-bit
                      against the reference (tests/golden/*.npz)
oracle_lockstep.c     the same step rules in the LOCKSTEP / Philox form the HIP
lockstep.py           kernels implement -- the bit-exact checker for the GPU
_ref/                 (git-ignored) the reference's own C kernels compiled from
                      /root/reference/src_c by ``make -C oracle ref``

Parity status: PINNED (see tests/test_oracle_golden.py).
"""
