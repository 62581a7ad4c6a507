"""
ctypes binding of libnjode_hip.so (C ABI: include/njode_hip.h).

There is no CPU fallback: if the shared library has not been built
(``python -m njode_amd.build`` / ``__graft_entry__.build()``) importing this
module's ``lib()`` raises, and every product entry point goes through it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NJODE_LIB: another build of the same library, e.g. the diagnostic build of tools/ubench)
LIB_PATH = os.environ.get('NJODE_LIB') or os.path.join(_HERE, 'libnjode_hip.so')

# ---- constants mirrored from include/njode_hip.h -----------------------------------
NJODE_OK = 0
E_UNSUPPORTED, E_BADARG, E_WORKSPACE, E_HIP = 1, 2, 3, 4
ACT_TANH, ACT_RELU = 0, 1
F_MASKED, F_INPUT_CURRENT_T, F_RESIDUAL, F_LOSS_EASY, F_USE_RNN = 0x1, 0x2, 0x4, 0x8, 0x10
C_TRAIN, C_GET_LOSS, C_RETURN_PATH, C_SAVE_BWD, C_LOSS_IN_BWD = 0x1, 0x2, 0x4, 0x8, 0x10
C_SCHED_KNOWN, C_SCHED_TAIL = 0x20, 0x40
C_PLAN_READY, C_NEED_HT = 0x80, 0x100
C_ROWS_IN_FWD = 0x400       # saving forward also runs the backward's row pass (autograd bridge)
C_GEN_LOCKSTEP = 0x200      # shape-generic kernels: unmasked loss calls stay on the lockstep plan
C_PLAN_DEFER = 0x800        # njode_plan_f32: the plan rides in front of the next forward call's ODE-forward launch

EXPORTS = ('njode_supported', 'njode_param_count', 'njode_workspace_bytes',
           'njode_plan_bytes', 'njode_plan_f32', 'njode_plan_flush', 'njode_plan_barrier_failures',
           'njode_forward_f32', 'njode_backward_f32', 'njode_backward_loss_f32',
           'njode_adam_step_f32',
           'njode_last_error', 'njode_build_info', 'njode_profile_enable',
           'njode_profile_read',
           # include/njode_producer.h
           'njode_philox4x32_10', 'njode_generate_paths', 'njode_sample_observations',
           'njode_collate_count', 'njode_collate_fill',
           # include/njode_selftest.h
           'njode_selftest_dropout_words', 'njode_debug_plan_stamps')
SDE_MODELS = {'BlackScholes': 0, 'OrnsteinUhlenbeck': 1, 'Heston': 2}


MAX_HIDDEN = 8      # NJODE_MAX_HIDDEN (include/njode_hip.h)


class NjodeNet(C.Structure):
    _fields_ = [('n_hidden', C.c_int32), ('width', C.c_int32 * MAX_HIDDEN),
                ('act', C.c_int32 * MAX_HIDDEN)]


class NjodeDims(C.Structure):
    _fields_ = [('input_size', C.c_int32), ('hidden_size', C.c_int32),
                ('output_size', C.c_int32), ('n_hidden', C.c_int32),
                ('width', C.c_int32), ('act', C.c_int32), ('flags', C.c_int32),
                ('per_net', C.c_int32), ('nets', NjodeNet * 3)]


class NjodeSchedule(C.Structure):
    _fields_ = [('n_steps', C.c_int32), ('n_times', C.c_int32),
                ('step_dt', C.c_void_p), ('step_t', C.c_void_p),
                ('k_jump', C.c_void_p), ('time_f32', C.c_void_p),
                ('time_ptr', C.c_void_p)]


class NjodeBatch(C.Structure):
    _fields_ = [('batch_size', C.c_int32), ('n_obs', C.c_int32),
                ('start_X', C.c_void_p), ('X', C.c_void_p), ('M', C.c_void_p),
                ('obs_idx', C.c_void_p), ('n_obs_ot', C.c_void_p),
                ('loss_batch_size', C.c_float), ('path_id_offset', C.c_int64),
                ('plan', C.c_void_p), ('grad_hT', C.c_void_p)]


class NjodeSde(C.Structure):
    _fields_ = [('model', C.c_int32), ('n_paths', C.c_int32), ('dim', C.c_int32),
                ('n_steps', C.c_int32), ('has_sine', C.c_int32), ('reserved', C.c_int32),
                ('drift', C.c_double), ('volatility', C.c_double), ('mean', C.c_double),
                ('speed', C.c_double), ('correlation', C.c_double), ('S0', C.c_double),
                ('maturity', C.c_double), ('sine_coeff', C.c_double)]


class NjodeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libnjode_hip error {}: {}'.format(code, msg))
        self.code = code


class NjodeUnsupported(NjodeError, NotImplementedError):
    pass


_lib = None


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'njode_amd: {} not found. The NJ-ODE path has no CPU fallback; build '
            'the gfx950 library first: `python -m njode_amd.build` (needs hipcc).'
            .format(LIB_PATH))
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, u64, sz = C.c_void_p, C.c_int32, C.c_float, C.c_uint64, C.c_size_t
    L.njode_supported.argtypes = [C.POINTER(NjodeDims)]
    L.njode_supported.restype = C.c_int
    L.njode_param_count.argtypes = [C.POINTER(NjodeDims)]
    L.njode_param_count.restype = sz
    L.njode_workspace_bytes.argtypes = [C.POINTER(NjodeDims), i32, i32, i32, i32, i32,
                                        C.POINTER(sz)]
    L.njode_workspace_bytes.restype = C.c_int
    L.njode_plan_bytes.argtypes = [C.POINTER(NjodeDims), i32, i32, i32, i32, i32, C.POINTER(sz)]
    L.njode_plan_bytes.restype = C.c_int
    L.njode_plan_f32.argtypes = [C.POINTER(NjodeDims), C.POINTER(NjodeBatch),
                                 C.POINTER(NjodeSchedule), i32, vp, sz, vp]
    L.njode_plan_f32.restype = C.c_int
    L.njode_plan_flush.argtypes = []
    L.njode_plan_flush.restype = C.c_int
    L.njode_plan_barrier_failures.argtypes = []
    L.njode_plan_barrier_failures.restype = C.c_int
    L.njode_forward_f32.argtypes = [C.POINTER(NjodeDims), vp, C.POINTER(NjodeBatch),
                                    C.POINTER(NjodeSchedule), i32, f32, f32, u64,
                                    vp, vp, vp, vp, vp, sz, vp]
    L.njode_forward_f32.restype = C.c_int
    L.njode_backward_f32.argtypes = [C.POINTER(NjodeDims), vp, C.POINTER(NjodeBatch),
                                     C.POINTER(NjodeSchedule), i32, f32, f32, u64,
                                     vp, vp, vp, sz, vp]
    L.njode_backward_f32.restype = C.c_int
    L.njode_backward_loss_f32.argtypes = [C.POINTER(NjodeDims), vp, C.POINTER(NjodeBatch),
                                          C.POINTER(NjodeSchedule), i32, f32, f32, u64,
                                          vp, vp, vp, vp, sz, vp]
    L.njode_backward_loss_f32.restype = C.c_int
    L.njode_adam_step_f32.argtypes = [vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, i32,
                                      f32, vp]
    L.njode_adam_step_f32.restype = C.c_int
    L.njode_last_error.restype = C.c_char_p
    L.njode_build_info.restype = C.c_char_p
    L.njode_profile_enable.argtypes = [C.c_int]
    L.njode_profile_enable.restype = C.c_int
    L.njode_profile_read.argtypes = [C.c_char_p, sz]
    L.njode_profile_read.restype = C.c_int
    f64 = C.c_double
    L.njode_philox4x32_10.argtypes = [i32, vp, vp, vp, vp]
    L.njode_generate_paths.argtypes = [C.POINTER(NjodeSde), u64, vp, vp, vp]
    L.njode_sample_observations.argtypes = [i32, i32, f64, u64, vp, vp, vp, vp]
    L.njode_collate_count.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp, vp]
    L.njode_collate_fill.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp,
                                     C.POINTER(C.c_int32), i32, vp, vp, vp, vp]
    for name in ('njode_philox4x32_10', 'njode_generate_paths', 'njode_sample_observations',
                 'njode_collate_count', 'njode_collate_fill',
           # include/njode_selftest.h
           'njode_selftest_dropout_words', 'njode_debug_plan_stamps'):
        getattr(L, name).restype = C.c_int
    L.njode_selftest_dropout_words.argtypes = [u64, u64, C.c_uint32, C.c_uint32, i32, vp, vp]
    L.njode_selftest_dropout_words.restype = C.c_int
    L.njode_debug_plan_stamps.argtypes = [vp, C.c_int]
    _lib = L
    return L


def check(rc):
    if rc == NJODE_OK:
        return
    msg = lib().njode_last_error().decode('utf-8', 'replace')
    if rc == E_UNSUPPORTED:
        raise NjodeUnsupported(rc, msg)
    raise NjodeError(rc, msg)


def build_info():
    return lib().njode_build_info().decode()


def profile_enable(on=True):
    """True / 1: every kernel; 2: the ODE backward kernel only (cheap enough for a timed region)."""
    check(lib().njode_profile_enable(int(on)))


def profile_read():
    """{kernel name: (launches, total_ms)} since the last read (synchronises)."""
    buf = C.create_string_buffer(1 << 16)
    check(lib().njode_profile_read(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms = line.split()
        out[name] = (int(n), float(ms))
    return out
