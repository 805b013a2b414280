"""ctypes binding of libgapflow_hip.so (include/gapflow_hip.h).

There is deliberately no fallback: if the HIP library is missing or no MI355X is visible,
every compute entry point raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GPF_LIB_PATH', os.path.join(HERE, 'lib', 'libgapflow_hip.so'))

EOS_IDS = {'DH': 0, 'PL': 1, 'vdW': 2, 'MT': 3, 'cubic': 4, 'BWR': 5, 'Bayada': 6}
EOS_KEYS = {'DH': ['rho0', 'P0', 'C1', 'C2'], 'PL': ['rho0', 'P0', 'alpha'], 'vdW': ['M', 'T', 'a', 'b'],
            'MT': ['rho0', 'P0', 'K', 'n'], 'cubic': ['a', 'b', 'c', 'd'], 'BWR': ['T', 'gamma'],
            'Bayada': ['rho_l', 'rho_v', 'c_l', 'c_v']}
# defaults of the reference's EOS functions (pressure.py:79, 112, 140, 176, 208, 233): eos_pressure passes only the keys
# present in `properties` (pressure.py:73-76), so absent ones take these
EOS_DEFAULTS = {'DH': {'rho0': 877.7007, 'P0': 101325., 'C1': 3.5e8, 'C2': 1.23}, 'PL': {'rho0': 1.1853, 'P0': 101325., 'alpha': 0.},
                'vdW': {'M': 39.948, 'T': 100., 'a': 1.355, 'b': 0.03201}, 'MT': {'rho0': 700., 'P0': 0.101e6, 'K': 0.557e9, 'n': 7.33},
                'cubic': {'a': 15.2, 'b': -9.6, 'c': 3.35, 'd': -0.07}, 'BWR': {'gamma': 3.}, 'Bayada': {}}
PIEZO_IDS = {'Barus': 1, 'Roelands': 2, 'Dukler': 3, 'McAdams': 4}
PIEZO_KEYS = {'Barus': ['aB'], 'Roelands': ['mu_inf', 'p_ref', 'z'], 'Dukler': ['eta_v', 'rho_l', 'rho_v'],
              'McAdams': ['eta_v', 'rho_l', 'rho_v']}
# defaults of the reference's viscosity laws (viscosity.py:144, 168, 200, 231, 262, 288)
PIEZO_DEFAULTS = {'Barus': {'aB': 2.e-8}, 'Roelands': {'mu_inf': 1.e-3, 'p_ref': 1.96e8, 'z': 0.68},
                  'Dukler': {'eta_v': 3.9e-5, 'rho_l': 850., 'rho_v': 0.019}, 'McAdams': {'eta_v': 3.9e-5, 'rho_l': 850., 'rho_v': 0.019}}
THINNING_DEFAULTS = {'Eyring': {'tauE': 5.e5}, 'Carreau': {'mu_inf': 1.e-3, 'lam': 0.02, 'a': 2., 'N': 0.8}}
THINNING_IDS = {'Eyring': 1, 'Carreau': 2}
THINNING_KEYS = {'Eyring': ['tauE'], 'Carreau': ['mu_inf', 'lam', 'a', 'N']}
BC_P, BC_D, BC_N = 0, 1, 2
FIELD_Q, FIELD_TOPO, FIELD_EXTRA, FIELD_PRESSURE, FIELD_TAU_AVG, FIELD_WALL_LOWER, FIELD_WALL_UPPER = range(7)
FIELD_PRESSURE_VAR, FIELD_WALL_XZ_VAR, FIELD_WALL_YZ_VAR = 7, 8, 9
FIELD_DEFORMATION = 10
FIELD_NCOMP = {FIELD_Q: 3, FIELD_TOPO: 3, FIELD_EXTRA: 1, FIELD_PRESSURE: 1, FIELD_TAU_AVG: 3,
               FIELD_WALL_LOWER: 6, FIELD_WALL_UPPER: 6, 7: 1, 8: 1, 9: 1, 10: 1}


class GpfConfig(C.Structure):
    _fields_ = [('Nx', C.c_int32), ('Ny', C.c_int32), ('dx', C.c_double), ('dy', C.c_double),
                ('U', C.c_double), ('V', C.c_double), ('eta', C.c_double), ('zeta', C.c_double),
                ('eos', C.c_int32), ('eos_par', C.c_double * 8),
                ('piezo', C.c_int32), ('piezo_par', C.c_double * 4),
                ('bc_rule', (C.c_int32 * 3) * 4), ('bc_value', C.c_double * 4),
                ('halo_lo', C.c_int32), ('halo_hi', C.c_int32),
                ('adaptive', C.c_int32), ('CFL', C.c_double), ('dt_fixed', C.c_double), ('tol', C.c_double),
                ('max_it', C.c_int64), ('mc_order', C.c_int32), ('device', C.c_int32),
                ('thinning', C.c_int32), ('thinning_par', C.c_double * 4)]


class GpfScalars(C.Structure):
    _fields_ = [('step', C.c_int64), ('simtime', C.c_double), ('dt', C.c_double), ('ekin', C.c_double),
                ('ekin_old', C.c_double), ('residual', C.c_double), ('v_max', C.c_double), ('v_sound', C.c_double),
                ('mass', C.c_double), ('invalid', C.c_int32), ('converged', C.c_int32)]


# name -> (restype, argtypes); every symbol declared in include/gapflow_hip.h
_DP = C.POINTER(C.c_double)
_VPP = C.POINTER(C.c_void_p)
SIGNATURES = {
    'gpf_create': (C.c_int, [C.POINTER(GpfConfig), _VPP]),
    'gpf_destroy': (C.c_int, [C.c_void_p]),
    'gpf_set_stream': (C.c_int, [C.c_void_p, C.c_void_p]),
    'gpf_last_error': (C.c_char_p, []),
    'gpf_device_count': (C.c_int, []),
    'gpf_upload': (C.c_int, [C.c_void_p, C.c_int, _DP, C.c_size_t]),
    'gpf_download': (C.c_int, [C.c_void_p, C.c_int, _DP, C.c_size_t]),
    'gpf_update_closures': (C.c_int, [C.c_void_p]),
    'gpf_pre_run': (C.c_int, [C.c_void_p]),
    'gpf_step': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(GpfScalars), C.c_int64, C.POINTER(C.c_int64)]),
    'gpf_step_unfused': (C.c_int, [C.c_void_p]),
    'gpf_step_timed': (C.c_int, [C.c_void_p, C.c_int64, _DP, _DP]),
    'gpf_scalars': (C.c_int, [C.c_void_p, C.POINTER(GpfScalars)]),
    'gpf_set_ekin_old': (C.c_int, [C.c_void_p, C.c_double]),
    'gpf_set_dt': (C.c_int, [C.c_void_p, C.c_double]),
    'gpf_slab_message': (C.c_int, [C.c_void_p, _VPP, C.POINTER(C.c_size_t)]),
    'gpf_step_local': (C.c_int, [C.c_void_p, C.c_int]),
    'gpf_step_commit': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    'gpf_state': (C.c_int, [C.c_void_p, C.POINTER(GpfScalars)]),
    'gpf_viscous_stress': (C.c_int, [C.c_int64] + [C.c_void_p] * 6 + [C.c_double] * 3 + [C.c_int] + [C.c_void_p] * 3),
    'gpf_eos': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'gpf_viscosity': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_double, C.c_double, C.c_void_p]),
    'gpf_elastic_setup': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_int]),
    'gpf_elastic_update': (C.c_int, [C.c_void_p]),
    'gpf_p2p_export': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'gpf_p2p_connect': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    'gpf_p2p_set_timeout': (C.c_int, [C.c_void_p, C.c_double]),
    'gpf_upload_beyond': (C.c_int, [C.c_void_p, C.c_int, _DP, C.c_size_t]),
    'gpf_stream_probe': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, _DP]),
    'gpf_step_p2p': (C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    'gpf_stage_message': (C.c_int, [C.c_void_p]),
    'gpf_stage_absorb': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    'gpf_close_step_local': (C.c_int, [C.c_void_p]),
    'gpf_close_step_commit': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(GpfScalars)]),
    'gpf_set_seam_topo': (C.c_int, [C.c_void_p, C.c_int, _DP, C.c_size_t]),
    'gpf_predictor_corrector': (C.c_int, [C.c_int, C.c_int, _DP, _DP, _DP, C.c_int, _DP, _DP]),
    'gpf_source': (C.c_int, [C.c_int, C.c_int, _DP, _DP, _DP, _DP, _DP, _DP]),
    'gpf_open_step': (C.c_int, [C.c_void_p]),
    'gpf_stage_closures': (C.c_int, [C.c_void_p]),
    'gpf_stage_advance': (C.c_int, [C.c_void_p, C.c_int]),
    'gpf_close_step': (C.c_int, [C.c_void_p, C.POINTER(GpfScalars)]),
    'gpf_gp_fit': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _DP, _DP, C.c_double, _DP, C.c_double, _DP, _DP, _DP]),
    'gpf_gp_set_model': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), _DP, _DP, _DP,
                                   C.c_double, _DP, C.c_double, C.c_double]),
    'gpf_gp_clear_model': (C.c_int, [C.c_void_p, C.c_int]),
    'gpf_gp_factorisation': (C.c_char_p, []),
    'gpf_plan_note': (C.c_char_p, [C.c_void_p]),
    'gpf_gp_set_scales': (C.c_int, [C.c_void_p, C.c_int, _DP, C.c_double]),
    'gpf_gp_variance': (C.c_int, [C.c_void_p, C.c_int, C.c_int, _DP]),
    'gpf_gp_pass_counts': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'gpf_gp_nll_open': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _DP, _DP, C.c_double, C.POINTER(C.c_void_p)]),
    'gpf_gp_nll_eval': (C.c_int, [C.c_void_p, _DP, _DP, _DP, C.POINTER(C.c_int)]),
    'gpf_gp_nll_close': (C.c_int, [C.c_void_p]),
}

_lib = None


class GapflowHipError(RuntimeError):
    pass


def _pin_hip_runtime():
    """Make sure ONE HIP runtime serves this process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Whichever copy is
    mapped first wins, and a process that maps /opt/rocm's first cannot initialise torch.cuda afterwards
    ("No HIP GPUs are available").  The slab path needs torch.distributed next to this library, so when
    torch is installed its bundled runtime is mapped first -- without importing torch."""
    import sys
    if os.environ.get('GPF_SYSTEM_HIP') == '1':
        return
    if 'torch' in sys.modules:
        libdir = os.path.join(os.path.dirname(sys.modules['torch'].__file__), 'lib')
        for var, name in (('GPF_ROCBLAS_PATH', 'librocblas.so'), ('GPF_HIPFFT_PATH', 'libhipfft.so')):
            if os.path.exists(os.path.join(libdir, name)):
                os.environ.setdefault(var, os.path.join(libdir, name))
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], 'lib')
    cand = os.path.join(libdir, 'libamdhip64.so')
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        # rocBLAS (GP variance solve) must come from the same bundle, or a later `import torch` crashes
        for var, name in (('GPF_ROCBLAS_PATH', 'librocblas.so'), ('GPF_HIPFFT_PATH', 'libhipfft.so')):
            if os.path.exists(os.path.join(libdir, name)):
                os.environ.setdefault(var, os.path.join(libdir, name))


def _preload_rocsolver():
    """Unless GPF_USE_ROCSOLVER=0: map rocBLAS and the rocSOLVER that lies NEXT TO it now, before this process makes its first HIP
    call (rocSOLVER's dpotrf / dpotrs are the default factorisation of the GP kernel matrices).  Round 2 recorded a dlopen of rocSOLVER that did not return within 500 s when issued after the HIP runtime was up,
    and an abort when two rocBLAS images met (profiles/r03_rocsolver/README.md): the library itself only ever binds to
    images already mapped or to the copy beside its rocBLAS (csrc/gp_kernels.hip: roclibs), and this puts them there early."""
    if os.environ.get('GPF_USE_ROCSOLVER') == '0':
        return
    path = os.environ.get('GPF_ROCBLAS_PATH')
    try:
        blas = C.CDLL(path if path and os.path.exists(path) else 'librocblas.so.5', mode=C.RTLD_GLOBAL)
    except OSError:
        return                          # no rocBLAS here: the GP entry points will say so when they are called
    info = _DlInfo()
    libdl = C.CDLL(None)
    libdl.dladdr.argtypes, libdl.dladdr.restype = [C.c_void_p, C.POINTER(_DlInfo)], C.c_int
    if not libdl.dladdr(C.cast(blas.rocblas_create_handle, C.c_void_p), C.byref(info)) or not info.dli_fname:
        return
    here = os.path.dirname(info.dli_fname.decode())
    for name in ('librocsolver.so.0', 'librocsolver.so'):
        if os.path.exists(os.path.join(here, name)):
            try:
                C.CDLL(os.path.join(here, name), mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


class _DlInfo(C.Structure):
    _fields_ = [('dli_fname', C.c_char_p), ('dli_fbase', C.c_void_p), ('dli_sname', C.c_char_p), ('dli_saddr', C.c_void_p)]


def load():
    """Load libgapflow_hip.so once; raises if it has not been built (python -m gapflow_amd.build)."""
    global _lib
    if _lib is None:
        _pin_hip_runtime()
        _preload_rocsolver()
        if not os.path.exists(LIB_PATH):
            raise GapflowHipError(
                f"{LIB_PATH} not found: build it with `python -m gapflow_amd.build` (needs hipcc). "
                "gapflow_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(code):
    if code != 0:
        raise GapflowHipError(f"libgapflow_hip error {code}: {load().gpf_last_error().decode()}")


def require_device():
    lib = load()
    if lib.gpf_device_count() < 1:
        raise GapflowHipError("no HIP device visible: gapflow_amd runs its hot path on an MI355X only "
                              "(there is no CPU fallback)")
    return lib


def as_dp(a):
    return a.ctypes.data_as(_DP)


def f64c(a):
    return np.ascontiguousarray(a, dtype=np.float64)
