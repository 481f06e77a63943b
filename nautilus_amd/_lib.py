"""ctypes binding of the C ABI in ``include/nautilus_hip.h``.

The HIP library is the product path; there is no CPU fallback.  If the shared
object is missing or a call fails, a ``RuntimeError`` is raised.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NAUTILUS_HIP_LIB') or os.path.join(
    _HERE, 'lib', 'libnautilus_hip.so')

# NB_ABI_VERSION of include/nautilus_hip.h this binding was written against
ABI_VERSION = 6

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class MemberDesc(C.Structure):
    _fields_ = [('n_ell', C.c_int32), ('idx_ell', c_int32_p),
                ('c', c_double_p), ('B', c_double_p), ('B_inv', c_double_p),
                ('free_dims', C.c_int32)]


class MlpDesc(C.Structure):
    _fields_ = [('n_networks', C.c_int32), ('mean', c_double_p),
                ('scale', c_double_p), ('coefs', C.POINTER(c_double_p)),
                ('intercepts', C.POINTER(c_double_p))]


class NeuralDesc(C.Structure):
    _fields_ = [('ellipsoid', MemberDesc), ('mlp', C.POINTER(MlpDesc)),
                ('score_predict_min', C.c_double), ('radius2', C.c_double)]


class BoundDesc(C.Structure):
    _fields_ = [('n_dim', C.c_int32), ('n_members', C.c_int32),
                ('members', C.POINTER(MemberDesc)), ('log_v_all', c_double_p),
                ('unit_cube', C.c_int32), ('n_neural', C.c_int32),
                ('neural', C.POINTER(NeuralDesc)),
                ('n_periodic', C.c_int32), ('periodic', c_int32_p),
                ('centers', c_double_p)]


_SIGNATURES = {
    'nb_abi_version': (C.c_int, []),
    'nb_last_error': (C.c_char_p, []),
    'nb_bound_create': (C.c_int, [C.POINTER(BoundDesc),
                                  C.POINTER(C.c_void_p)]),
    'nb_bound_destroy': (C.c_int, [C.c_void_p]),
    'nb_bound_nbytes': (C.c_int64, [C.c_void_p]),
    'nb_boundlist_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int32,
                                      C.POINTER(C.c_void_p)]),
    'nb_boundlist_destroy': (C.c_int, [C.c_void_p]),
    'nb_contains': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                              C.c_void_p]),
    'nb_contains_any': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_void_p]),
    'nb_first_containing': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p]),
    'nb_member_count': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_void_p]),
    'nb_neural_score': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_void_p]),
    'nb_propose': (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64,
                             C.c_void_p, C.c_void_p]),
    'nb_accept': (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                            C.c_int64, C.c_void_p, C.c_void_p]),
    'nb_compact_scratch_bytes': (C.c_int64, [C.c_int64]),
    'nb_compact_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint8,
                                  C.c_uint8, C.c_int64, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    'nb_shell_stats': (C.c_int, [C.c_void_p, C.c_int64, C.c_double,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    'nb_shell_stats_scratch_bytes': (C.c_int64, [C.c_int64]),
    'nb_trainer_create': (C.c_int, [C.c_int32, C.c_int32, C.c_int64,
                                    C.c_void_p, C.c_void_p,
                                    C.POINTER(c_double_p),
                                    C.POINTER(c_double_p),
                                    C.POINTER(C.c_void_p)]),
    'nb_trainer_create_fleet': (C.c_int, [C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    'nb_trainer_run_fleet': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    'nb_trainer_set_hparams': (C.c_int, [C.c_void_p, C.c_double, C.c_double,
                                         C.c_double, C.c_double, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_double]),
    'nb_trainer_run': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32,
                                 c_int32_p, C.c_void_p]),
    'nb_trainer_status': (C.c_int, [C.c_void_p, c_int32_p, C.c_void_p]),
    'nb_trainer_run_async': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.POINTER(C.c_int64)]),
    'nb_trainer_wait': (C.c_int, [C.c_void_p, C.c_int64, c_int32_p]),
    'nb_host_shuffle_epochs': (C.c_int, [C.c_int32, C.c_void_p, c_int32_p,
                                         C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_void_p]),
    'nb_trainer_loss_curve': (C.c_int, [C.c_void_p, C.c_int32, c_double_p,
                                        C.c_int32]),
    'nb_trainer_weights': (C.c_int, [C.c_void_p, C.c_int32,
                                     C.POINTER(c_double_p),
                                     C.POINTER(c_double_p)]),
    'nb_trainer_destroy': (C.c_int, [C.c_void_p]),
    'nb_philox_uniform': (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32,
                                    C.c_uint32, C.c_int64, C.c_void_p,
                                    C.c_void_p]),
    'nb_set_eval_counters': (C.c_int, [C.c_void_p]),
    'nb_mfma_f64_peak': (C.c_int, [C.c_int32, c_double_p]),
    'nb_mvee_weights': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    'nb_mvee_weights_work_doubles': (C.c_int64, [C.c_int64, C.c_int32,
                                                 C.c_int32]),
    'nb_mvee_work_doubles': (C.c_int64, [C.c_int32, C.c_int64, C.c_int32,
                                         C.c_int32]),
    'nb_mvee_khachiyan': (C.c_int, [C.c_int32, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_int64), C.c_int32,
                                    C.c_int32, C.c_int32,
                                    C.POINTER(C.c_void_p), C.c_void_p,
                                    C.c_void_p]),
    'nb_whiten_work_doubles': (C.c_int64, [C.c_int64, C.c_int32]),
    'nb_whiten': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p]),
    'nb_moments_work_doubles': (C.c_int64, [C.c_int64, C.c_int32]),
    'nb_weighted_moments': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int32, C.c_double, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    'nb_quadform_work_doubles': (C.c_int64, []),
    'nb_quadform_max': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    'nb_ellipsoid_transform': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_void_p]),
    'nb_standardize': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    'nb_prior_transform': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32,
                                     C.c_void_p, c_double_p, c_double_p,
                                     C.c_void_p, C.c_void_p]),
    'nb_loglike_rosenbrock': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32,
                                        C.c_double, C.c_double, C.c_double,
                                        C.c_void_p, C.c_void_p]),
    'nb_loglike_funnel': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32,
                                    C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_void_p, C.c_void_p]),
    'nb_live_append': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int32,
                                 C.c_void_p, C.c_void_p]),
    'nb_live_select': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    'nb_live_stats': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    'nb_comm_unique_id': (C.c_int, [C.c_void_p]),
    'nb_comm_init': (C.c_int, [C.c_int32, C.c_int32, C.c_void_p,
                               C.POINTER(C.c_void_p)]),
    'nb_comm_destroy': (C.c_int, [C.c_void_p]),
    'nb_comm_rank_key': (C.c_uint64, [C.c_uint64, C.c_int32]),
    'nb_comm_allgather_f64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p]),
    'nb_comm_allreduce_i64': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p]),
    'nb_list_eval_work_bytes': (C.c_int64, [C.c_void_p, C.c_int64]),
    'nb_list_eval': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p]),
    'nb_accept_staged_work_bytes': (C.c_int64, [C.c_void_p, C.c_int64]),
    'nb_accept_staged': (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64,
                                   C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.c_int64,
                                   C.POINTER(C.c_int64), C.c_void_p]),
    'nb_neural_score_rows': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p]),
    'nb_gmm_out_doubles': (C.c_int64, [C.c_int32]),
    'nb_gmm_scratch_doubles': (C.c_int64, [C.c_int64, C.c_int32]),
    'nb_gmm_work_doubles': (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    'nb_gmm_logp_offset': (C.c_int64, [C.c_int32]),
    'nb_gmm_fit': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                             C.c_uint64, C.c_double, C.c_double, C.c_int32,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nb_gmm_set_max_wgs': (C.c_int, [C.c_int32]),
    'nb_phase_shift': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                 c_int32_p, c_double_p, C.c_int32,
                                 C.c_void_p]),
    'nb_ellipsoid_contains_stream': (C.c_int, [C.c_void_p, C.c_void_p,
                                               C.c_int64, C.c_void_p,
                                               C.c_void_p]),
}

_lib = None


def load():
    """Load ``libnautilus_hip.so`` (built by ``make`` / ``__graft_entry__.build``).

    Raises RuntimeError if it is missing -- the product has no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: the library resolves libamdhip64 to whatever the process
    # has loaded, and the HIP runtime torch ships has to be that one (loaded
    # the other way round -- build() and smoke() in one process -- the two
    # runtimes each see their own device state: "no ROCm-capable device")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'nautilus_amd: HIP library %s not found; run `make` (hipcc, '
            'gfx950).  There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.nb_abi_version() != ABI_VERSION:
        raise RuntimeError(
            'nautilus_amd: %s has ABI version %d, this package binds version '
            '%d (rebuild with `make`)' % (LIB_PATH, lib.nb_abi_version(),
                                          ABI_VERSION))
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


ERR_ARG, ERR_HIP, ERR_UNSUPPORTED = 1, 2, 3     # nb_common.h


class NativeError(RuntimeError):
    """A nonzero status of the C ABI; ``code`` is the status."""

    def __init__(self, code, message):
        super().__init__('nautilus_hip: ' + message)
        self.code = code


def check(status):
    if status != 0:
        raise NativeError(status, load().nb_last_error().decode(
            'utf-8', 'replace'))
