"""ctypes binding of libmc3hip.so (include/mc3hip.h).

The product path has no CPU fallback: if the library is missing or fails to
load, importing the engine raises.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (MC3HIP_LIBRARY: a diagnostic build of the same library, e.g. tools/stamps.py's)
LIB_PATH = os.environ.get('MC3HIP_LIBRARY') or os.path.join(HERE, 'libmc3hip.so')

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_lp = ctypes.POINTER(ctypes.c_int64)
_up = ctypes.POINTER(ctypes.c_uint32)

OK = 0
ERR_ARG, ERR_HIP, ERR_STATE, ERR_FULL, ERR_COMM, ERR_ABORTED = -1, -2, -3, -4, -5, -6
MRW, DEMC, SNOOKER = 0, 1, 2
MODEL_POLY, MODEL_LINEAR = 0, 1
FLAG_NO_GRAPH = 1
FLAG_NO_PERSIST = 2
FLAG_P2P = 4
FLAG_NO_COURIER = 8
FLAG_NO_JIT = 16
FLAG_JIT = 32
MAX_FREE, MAX_PARAMS = 64, 128

SAMPLERS = {'mrw': MRW, 'demc': DEMC, 'snooker': SNOOKER}


class Problem(ctypes.Structure):
    """mc3hip_problem"""
    _fields_ = [
        ('struct_bytes', ctypes.c_int32), ('device', ctypes.c_int32),
        ('nchains', ctypes.c_int32), ('chain0', ctypes.c_int32),
        ('nlocal', ctypes.c_int32), ('nparams', ctypes.c_int32),
        ('ndata', ctypes.c_int32), ('sampler', ctypes.c_int32),
        ('wlike', ctypes.c_int32), ('model_kind', ctypes.c_int32),
        ('nmodel', ctypes.c_int32), ('thinning', ctypes.c_int32),
        ('flags', ctypes.c_uint32), ('trace_iters', ctypes.c_int32),
        ('M0', ctypes.c_int64), ('zlen', ctypes.c_int64),
        ('seed', ctypes.c_uint64),
        ('gamma', ctypes.c_double), ('fepsilon', ctypes.c_double),
        ('params', _dp), ('pstep', _dp), ('pmin', _dp), ('pmax', _dp),
        ('prior', _dp), ('priorlow', _dp), ('priorup', _dp),
        ('indep', _dp), ('data', _dp), ('uncert', _dp),
        ('zlag', ctypes.c_int32), ('wl_nreal', ctypes.c_int32),
    ]


class Plan(ctypes.Structure):
    """mc3hip_plan"""
    _fields_ = [('vlanes', ctypes.c_int32), ('tile', ctypes.c_int32),
                ('waves_per_chain', ctypes.c_int32),
                ('graph_nodes', ctypes.c_int32),
                ('persistent', ctypes.c_int32), ('fast', ctypes.c_int32),
                ('comm_graph', ctypes.c_int32), ('recoveries', ctypes.c_int32),
                ('tile_tree', ctypes.c_int32), ('resident', ctypes.c_int32),
                ('chains_per_workgroup', ctypes.c_int32),
                ('jit', ctypes.c_int32), ('capacity', ctypes.c_int32),
                ('zlag', ctypes.c_int32), ('remote_delay_ns', ctypes.c_int32),
                ('owner_waves', ctypes.c_int32)]


class Mc3HipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'libmc3hip error {code}: {message}')
        self.code = code


# every symbol include/mc3hip.h declares: (restype, argtypes)
_vp = ctypes.c_void_p
SYMBOLS = {
    'mc3hip_create': (ctypes.c_int, [ctypes.POINTER(Problem), ctypes.POINTER(_vp)]),
    'mc3hip_destroy': (None, [_vp]),
    'mc3hip_last_error': (ctypes.c_char_p, []),
    'mc3hip_get_plan': (ctypes.c_int, [_vp, ctypes.POINTER(Plan)]),
    'mc3hip_jit_request': (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int32]),
    'mc3hip_jit_install': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_char_p]),
    'mc3hip_init_population': (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_uint64, _dp, _dp,
                                              _lp, _lp]),
    'mc3hip_set_population': (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp]),
    'mc3hip_run': (ctypes.c_int, [_vp, ctypes.c_int64]),
    'mc3hip_sync': (ctypes.c_int, [_vp]),
    'mc3hip_recycle': (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_uint64]),
    'mc3hip_last_run_ms': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    'mc3hip_get_rows': (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, _dp, _dp, _ip]),
    'mc3hip_get_rows_settled': (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, _dp, _dp,
                                               _ip]),
    'mc3hip_put_rows': (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, _dp, _dp]),
    'mc3hip_get_state': (ctypes.c_int, [_vp, _dp, _dp, _lp, _lp, _dp, _lp, _dp, _lp]),
    'mc3hip_get_trace': (ctypes.c_int, [_vp, ctypes.c_int64, _dp, _ip]),
    'mc3hip_chain_moments': (ctypes.c_int, [_vp, ctypes.c_int64, _dp, _dp]),
    'mc3hip_chain_moments_rows': (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, _dp, _dp]),
    'mc3hip_posterior_summary': (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64,
                                                ctypes.c_int32, _lp, _dp,
                                                _dp, _dp, _dp, _dp]),
    'mc3hip_get_chisq': (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, _dp]),
    'mc3hip_kde_pdf': (ctypes.c_int, [ctypes.c_int32, _dp, ctypes.c_int32, _dp, ctypes.c_int32,
                                      ctypes.c_double, _dp]),
    'mc3hip_eval_chisq': (ctypes.c_int, [_vp, _dp, ctypes.c_int32, _dp]),
    'mc3hip_eval_model': (ctypes.c_int, [_vp, _dp, _dp]),
    'mc3hip_residuals': (ctypes.c_int, [ctypes.c_int32, _dp, _dp, _dp, ctypes.c_int32, _dp, _dp,
                                        _dp, ctypes.c_int32, _dp]),
    'mc3hip_chisq': (ctypes.c_int, [ctypes.c_int32, _dp, _dp, _dp, ctypes.c_int32, _dp, _dp, _dp,
                                    ctypes.c_int32, _dp]),
    'mc3hip_dwt_chisq': (ctypes.c_int, [ctypes.c_int32, _dp, ctypes.c_int32, _dp, _dp,
                                        ctypes.c_int32, _dp, _dp, _dp, ctypes.c_int32, _dp]),
    'mc3hip_dwt_daub4': (ctypes.c_int, [ctypes.c_int32, _dp, ctypes.c_int32, ctypes.c_int32, _dp]),
    'mc3hip_comm_unique_id': (ctypes.c_int, [_vp]),
    'mc3hip_comm_init': (ctypes.c_int, [_vp, _vp, ctypes.c_int32, ctypes.c_int32]),
    'mc3hip_comm_gather_log_post': (ctypes.c_int, [_vp]),
    'mc3hip_comm_info': (ctypes.c_int, [_vp, _ip, _ip]),
    'mc3hip_device_memory': (ctypes.c_int, [ctypes.c_int32, _lp, _lp]),
    'mc3hip_p2p_export': (ctypes.c_int, [_vp, _vp]),
    'mc3hip_p2p_attach': (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, _vp]),
    'mc3hip_p2p_detach': (ctypes.c_int, [_vp]),
    'mc3hip_p2p_rollback': (ctypes.c_int, [_vp, _lp, _lp]),
    'mc3hip_comm_destroy': (ctypes.c_int, [_vp]),
    'mc3hip_debug_stamps': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    'mc3hip_debug_launch_counts': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int64, _lp]),
    'mc3hip_debug_last_kernel': (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int64]),
    'mc3hip_selftest_math': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, _dp, ctypes.c_int32, _dp]),
    'mc3hip_selftest_philox': (ctypes.c_int, [ctypes.c_int32, _up, ctypes.c_int32, _up, _up]),
}

_LIB = None


def load():
    """Load libmc3hip.so and bind every declared entry point; raises if the
    library is absent (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} is missing: build it with `python -m mc3_amd.build` '
                '(hipcc, gfx950).  mc3_amd has no CPU fallback.')
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            if not hasattr(lib, name) and os.environ.get('MC3HIP_LIBRARY'):
                continue            # (a diagnostic build of another vintage)
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = lib
    return _LIB


def check(rc):
    if rc != OK:
        msg = load().mc3hip_last_error()
        raise Mc3HipError(rc, msg.decode() if msg else '')


def launch_counts():
    """The library's launch ledger of THIS process (include/mc3hip.h
    mc3hip_debug_launch_counts): [(launches, launches with no MC3HIP_* switch in the
    environment, stub offset in libmc3hip.so, device kernel name, [switches seen])]."""
    lib = load()
    need = ctypes.c_int64()
    check(lib.mc3hip_debug_launch_counts(None, 0, ctypes.byref(need)))
    buf = ctypes.create_string_buffer(int(need.value) + 4096)
    check(lib.mc3hip_debug_launch_counts(buf, len(buf), ctypes.byref(need)))
    rows = []
    for line in buf.value.decode().splitlines():
        n, plain, off, name, *switches = line.split(' ')
        rows.append((int(n), int(plain), int(off, 16), name, switches))
    return rows


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def lptr(a):
    return a.ctypes.data_as(_lp) if a is not None else None
