"""Pin the shared deterministic math (include/mc3hip_detmath.h) to libm and the
Philox4x32-10 restatement to the Random123 known-answer vectors."""
import math

import numpy as np

from oracle import kernels as ok
from oracle import lockstep as ls


def ulp_err(got, want):
    got = np.asarray(got, float)
    want = np.asarray(want, float)
    return np.abs(got - want)/np.spacing(np.abs(want))


def test_det_exp_vs_libm():
    L = ok.lib()
    rs = np.random.RandomState(1)
    xs = np.concatenate([rs.uniform(-700, 700, 20000), rs.uniform(-2, 2, 20000),
                         rs.normal(0, 1e-3, 1000), [0.0, -0.0, 1.0, -1.0]])
    got = np.array([L.ls_det_exp(float(x)) for x in xs])
    want = np.exp(xs)
    assert ulp_err(got, want).max() <= 2.0
    assert L.ls_det_exp(0.0) == 1.0
    assert L.ls_det_exp(710.0) == math.inf
    assert L.ls_det_exp(-800.0) == 0.0
    assert math.isnan(L.ls_det_exp(math.nan))
    assert L.ls_det_exp(math.inf) == math.inf
    assert L.ls_det_exp(-math.inf) == 0.0
    # subnormal results round once
    sub = np.array([L.ls_det_exp(float(x)) for x in (-709.0, -720.0, -740.0, -745.0)])
    np.testing.assert_allclose(sub, np.exp([-709.0, -720.0, -740.0, -745.0]),
                               rtol=1e-12, atol=5e-324)


def test_det_exp2_exact_on_integers():
    L = ok.lib()
    for k in range(-60, 61):
        assert L.ls_det_exp2(float(k)) == 2.0**k
    rs = np.random.RandomState(2)
    ys = rs.uniform(-50, 50, 5000)
    got = np.array([L.ls_det_exp2(float(y)) for y in ys])
    assert ulp_err(got, np.exp2(ys)).max() <= 2.5


def test_det_log_vs_libm():
    L = ok.lib()
    rs = np.random.RandomState(3)
    xs = np.concatenate([np.exp(rs.uniform(-700, 700, 20000)),
                         rs.uniform(0.5, 2.0, 20000), [1.0, 5e-324, 2.2e-308]])
    got = np.array([L.ls_det_log(float(x)) for x in xs])
    want = np.log(xs)
    ok_ = (want == 0) & (got == 0)
    assert ulp_err(got[~ok_], want[~ok_]).max() <= 2.0
    assert L.ls_det_log(0.0) == -math.inf
    assert math.isnan(L.ls_det_log(-1.0))
    assert L.ls_det_log(math.inf) == math.inf


TWO_PI_LD = np.longdouble(2)*np.longdouble('3.14159265358979323846264338327950288')


def test_det_sincos_vs_libm():
    import ctypes
    L = ok.lib()
    rs = np.random.RandomState(4)
    us = np.concatenate([rs.uniform(0, 1, 20000), [0.0, 0.25, 0.5, 0.75, 0.125]])
    s = ctypes.c_double()
    c = ctypes.c_double()
    err = 0.0
    for u in us:
        L.ls_det_sincos2pi(float(u), ctypes.byref(s), ctypes.byref(c))
        a = TWO_PI_LD*np.longdouble(u)       # 80-bit reference angle
        err = max(err, abs(float(np.longdouble(s.value) - np.sin(a))),
                  abs(float(np.longdouble(c.value) - np.cos(a))))
        assert abs(s.value**2 + c.value**2 - 1.0) < 5e-16
    assert err < 2.5e-16
    L.ls_det_sincos2pi(0.25, ctypes.byref(s), ctypes.byref(c))
    assert (s.value, c.value) == (1.0, -0.0) or (s.value, abs(c.value)) == (1.0, 0.0)


def test_det_pow_half():
    L = ok.lib()
    assert L.ls_det_pow_half(2.5, 0) == 1.0
    assert L.ls_det_pow_half(2.5, 2) == 2.5
    assert L.ls_det_pow_half(2.5, 4) == 2.5*2.5
    assert L.ls_det_pow_half(4.0, 1) == 2.0
    assert L.ls_det_pow_half(4.0, 3) == 8.0
    np.testing.assert_allclose(L.ls_det_pow_half(1.7, 9), 1.7**4.5, rtol=1e-14)
    assert L.ls_det_pow_half(math.inf, 0) == 1.0
    assert L.ls_det_pow_half(math.inf, 2) == math.inf


def philox_numpy(ctr, key):
    """Independent NumPy statement of Philox4x32-10 (Salmon et al., SC'11)."""
    c = [np.uint64(v) for v in ctr]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0*c[0]
        p1 = M1*c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask, p1 & mask,
             ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask, p0 & mask]
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return tuple(int(v) for v in c)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [
        ((0, 0, 0, 0), (0, 0),
         (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,)*4, (0xffffffff,)*2,
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
         (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        assert ls.philox4x32_10(ctr, key) == want
        assert philox_numpy(ctr, key) == want
    rs = np.random.RandomState(5)
    for _ in range(50):
        ctr = tuple(int(v) for v in rs.randint(0, 2**32, 4, dtype=np.uint64))
        key = tuple(int(v) for v in rs.randint(0, 2**32, 2, dtype=np.uint64))
        assert ls.philox4x32_10(ctr, key) == philox_numpy(ctr, key)
