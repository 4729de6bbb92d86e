"""ctypes binding of libmxf_gp.so (include/mxf_gp.h).  There is NO CPU fallback: if the HIP library is
missing or no GPU is present the product path raises."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MXF_GP_LIB', os.path.join(_HERE, 'libmxf_gp.so'))   # env override: A/B builds of the kernels

F32, F64 = 0, 1
K_RBF, K_MATERN12, K_MATERN32, K_MATERN52, K_LINEAR, K_BIAS, K_WHITE = range(7)
WRITE, ACC_ADD, ACC_MUL = 0, 1, 2

_c = ctypes
_vp, _i, _i64, _d = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_double

# name -> argtypes (after the handle); every entry point include/mxf_gp.h declares is listed here
SIGNATURES = {
    'mxf_gram': [_i, _i, _i, _i64, _i64, _i, _vp, _i64, _vp, _i64, _vp, _i, _i64, _vp, _i64, _vp, _i64, _d, _i,
                 _vp, _i64, _i64, _vp],
    'mxf_gram2': [_i, _i, _i, _i, _i, _i64, _i64, _i, _vp, _i64, _vp, _i64, _vp, _i, _i64, _vp, _i64, _vp, _i, _i64, _vp, _i64, _vp, _i64, _d,
                  _vp, _i64, _i64, _vp],
    'mxf_gram_bwd': [_i, _i, _i, _i64, _i64, _i, _vp, _i64, _vp, _i64, _vp, _i, _i64, _vp, _i64, _vp, _i64, _i64,
                     _vp, _vp, _vp, _vp, _vp],
    'mxf_gemm': [_i, _i, _i, _i64, _i64, _i64, _d, _vp, _i64, _i64, _vp, _i64, _i64, _d, _vp, _i64, _i64, _i, _vp],
    'mxf_potrf': [_i, _i, _i64, _vp, _i64, _i64, _vp, _vp],
    'mxf_trsm': [_i, _i, _i, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp],
    'mxf_trtri': [_i, _i, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp],
    'mxf_sumlogdiag': [_i, _i, _i64, _vp, _i64, _i64, _vp, _vp],
    'mxf_coldot': [_i, _i, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp],
    'mxf_kdiag': [_i, _i, _i, _i64, _i, _vp, _i64, _vp, _i, _i64, _vp, _i64, _vp, _vp],
    'mxf_gp_predict': [_i, _i, _i, _i64, _i64, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i64, _vp, _vp, _i, _i, _vp, _vp, _vp],
    'mxf_svgp_predict': [_i, _i, _i, _i64, _i64, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _vp, _vp, _vp, _vp],
    'mxf_softplus_fwd': [_i, _i64, _vp, _vp, _vp],
    'mxf_softplus_bwd': [_i, _i64, _vp, _vp, _vp, _vp],
    'mxf_normal_reparam': [_i, _i, _i64, _vp, _vp, _vp, _vp, _vp],
    'mxf_normal_logpdf': [_i, _i, _i64, _vp, _vp, _i64, _vp, _i64, _d, _vp, _vp, _vp, _vp, _vp],
    'mxf_normal_reparam_bwd': [_i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    'mxf_adam_step': [_i, _i64, _vp, _vp, _vp, _vp, _d, _d, _d, _d, _d, _i, _vp],
    'mxf_sgd_step': [_i, _i64, _vp, _vp, _vp, _d, _d, _d, _d, _vp],
    'mxf_opt_step': [_i, _i, _i64, _vp, _vp, _vp, _vp, _d, _d, _d, _d, _d, _vp],
    'mxf_uniform_sum': [_i, _i64, _vp, _vp, _vp],
    'mxf_gp_logpdf': [_i, _i, _i, _i64, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _i64, _vp, _i64, _d,
                      _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    'mxf_svgp_logpdf': [_i, _i, _i, _i64, _i64, _i, _i, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp,
                        _d, _d, _d, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'mxf_svgp_logpdf_sampled': [_i, _i, _i, _i64, _i64, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _i64,
                                _vp, _i64, _d, _d, _d, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'mxf_f32x3_split': [_i64, _i64, _vp, _i64, _vp, _vp],
    'mxf_gemm_f32x3_planes': [_i64, _i64, _i64, _d, _vp, _vp, _d, _vp, _i64, _i, _vp],
    'mxf_gemm_f32x3': [_i64, _i64, _i64, _d, _vp, _i64, _vp, _i64, _d, _vp, _i64, _i, _vp],
    'mxf_make_diagonal': [_i, _i64, _i64, _vp, _vp, _vp],
    'mxf_diag_of': [_i, _i64, _i64, _vp, _vp, _vp],
    'mxf_gemm_f16x2': [_i64, _i64, _i64, _d, _vp, _i64, _vp, _i64, _d, _vp, _i64, _i, _vp],
    'mxf_f16x2_split': [_i64, _i64, _vp, _i64, _vp, _vp, _vp],
    'mxf_gemm_f16x2_planes': [_i64, _i64, _i64, _d, _vp, _vp, _vp, _vp, _d, _vp, _i64, _i, _vp],
    'mxf_gemm_f16x2_planes_kmajor': [_i64, _i64, _i64, _d, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    'mxf_svgp_logpdf_het': [_i, _i, _i, _i64, _i64, _i, _i, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp,
                            _d, _d, _d, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'mxf_svgp_logpdf_mat': [_i, _i, _i64, _i64, _i, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i, _vp, _vp, _vp, _d, _d, _d, _vp, _vp, _i,
                            _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'mxf_sgp_logpdf': [_i, _i, _i64, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _d, _d, _vp, _vp, _vp, _vp, _vp, _i,
                       _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'mxf_svgp_last_cond': [_c.POINTER(_d)],
    'mxf_svgp_cond_nowait': [_c.POINTER(_d), _i],
    'mxf_svgp_configure': [_i, _i],
    'mxf_svgp_timing': [_i],
    'mxf_svgp_timing_read': [_c.POINTER(_d)],
    'mxf_svgp_cond_slot': [_i, _c.POINTER(_d), _c.POINTER(_d), _i],
    'mxf_gemm_f16x2_planes_out': [_i64, _i64, _i64, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    'mxf_f16x2_planes_transpose': [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    'mxf_comm_unique_id': [_vp],
    'mxf_comm_init': [_i, _i, _vp],
    'mxf_allreduce_sum': [_i, _vp, _i64, _vp],
    'mxf_bcast': [_i, _vp, _i64, _i, _vp],
    'mxf_comm_destroy': [],
}
PLAIN = {  # entry points without the (handle, ...) -> int shape
    'mxf_version': ([], _i),
    'mxf_create': ([_i, _c.POINTER(_vp)], _i),
    'mxf_destroy': ([_vp], _i),
    'mxf_last_error': ([_vp], _c.c_char_p),
    'mxf_workspace_bytes': ([_vp], _i64),
    'mxf_workspace_generation': ([_vp], _i64),
    'mxf_f32x3_plane_elems': ([_i64, _i64], _i64),
    'mxf_svgp_whitened_ok': ([_i, _i, _i64, _i64, _i, _i, _i64], _i),
}
ALL_SYMBOLS = sorted(list(SIGNATURES) + list(PLAIN))

_lib = None
_lock = threading.Lock()
_handles = {}


class MXFError(RuntimeError):
    pass


def load():
    """dlopen libmxf_gp.so and declare prototypes.  Raises (loudly) when the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MXFError('libmxf_gp.so not found at %s -- build it with `python -c "import __graft_entry__ as g; '
                           'g.build()"` (or make -C mxfusion_amd/csrc). There is no CPU fallback.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (args, res) in PLAIN.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                raise MXFError('libmxf_gp.so does not export %s' % name)
            fn.argtypes, fn.restype = [_vp] + args, _i
        _lib = lib
    return _lib


def handle(device_index):
    """One library handle per (thread, device): handle calls are not re-entrant (include/mxf_gp.h)."""
    key = (threading.get_ident(), int(device_index))
    h = _handles.get(key)
    if h is None:
        lib = load()
        out = _vp()
        rc = lib.mxf_create(int(device_index), ctypes.byref(out))
        if rc != 0:
            raise MXFError('mxf_create(device=%d) failed with %d: no usable MI355X visible. There is no CPU '
                           'fallback.' % (device_index, rc))
        h = out
        _handles[key] = h
    return h


def workspace_generation(device_index):
    """Re-allocation count of the (thread, device) handle's scratch (mxf_workspace_generation): hipGraph holders compare it before a
    replay -- a captured launch carries the scratch addresses of capture time."""
    return int(load().mxf_workspace_generation(handle(device_index)))


def svgp_cond_nowait(device_index, reset=False):
    """Running maximum of cond_1(Kuu + jitter I) over the SVGP training calls finished so far on this (thread, device) handle
    (mxf_svgp_cond_nowait: a read of pinned host memory, no synchronisation)."""
    out = _d(0.0)
    call('mxf_svgp_cond_nowait', handle(device_index), ctypes.byref(out), int(bool(reset)))
    return float(out.value)


FORM_EXPLICIT, FORM_WHITENED = 0, 1
COND_SLOTS = 64


def svgp_configure(device_index, form, slot):
    """Float32 streaming form (FORM_EXPLICIT / FORM_WHITENED) and condition slot of the next SVGP training calls of this (thread, device)
    handle (mxf_svgp_configure)."""
    call('mxf_svgp_configure', handle(device_index), int(form), int(slot))


def svgp_cond_slot(device_index, slot, reset=False):
    """(last, running max) of the condition numbers the finished training calls published into `slot` -- no synchronisation
    (mxf_svgp_cond_slot)."""
    last, mx = _d(0.0), _d(0.0)
    call('mxf_svgp_cond_slot', handle(device_index), int(slot), ctypes.byref(last), ctypes.byref(mx), int(bool(reset)))
    return float(last.value), float(mx.value)


TIMING_KEYS = ('planes_a', 'psi2', 'planes_b', 't_gemm', 'reverse_pass', 'core_chain', 'v_gemm', 'call')


def svgp_timing(device_index, enable):
    call('mxf_svgp_timing', handle(device_index), int(bool(enable)))


def svgp_timing_read(device_index):
    """In-step durations (ms) of the last SVGP training call's bulk kernels (mxf_svgp_timing_read; synchronises); absent stages are dropped."""
    buf = (_d * 8)()
    call('mxf_svgp_timing_read', handle(device_index), buf)
    return {k: float(buf[i]) for i, k in enumerate(TIMING_KEYS) if buf[i] >= 0}


def svgp_whitened_ok(dtype, S, B, M, Q, P, stride_x):
    return bool(load().mxf_svgp_whitened_ok(int(dtype), int(S), int(B), int(M), int(Q), int(P), int(stride_x)))


def svgp_last_cond(device_index):
    """1-norm condition number of Kuu + jitter I of the last SVGP training call on this (thread, device) handle (mxf_svgp_last_cond;
    synchronises)."""
    out = _d(0.0)
    call('mxf_svgp_last_cond', handle(device_index), ctypes.byref(out))
    return float(out.value)


def call(name, h, *args):
    lib = load()
    rc = getattr(lib, name)(h, *args)
    if rc != 0:
        raise MXFError('%s failed (%d): %s' % (name, rc, lib.mxf_last_error(h).decode()))
