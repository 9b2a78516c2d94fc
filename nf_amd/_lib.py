"""ctypes binding of libnis_hip.so (C ABI in include/nis_hip.h).

The product path has NO CPU fallback: `lib()` raises if the shared library is missing and
`require_gpu()` raises if no ROCm device is visible.
"""
import ctypes as C
import os

NIS_MAX_DIM = 32
NIS_MAX_HIDDEN = 8
NIS_MAX_WIDTH = 64
NIS_MAX_BINS = 128
NIS_MAX_CELLS = 64
NIS_MAX_LEGS = 10

NIS_PWLIN, NIS_PWQUAD, NIS_AFFINE = 0, 1, 2
NIS_BN_EVAL, NIS_BN_TRAIN = 0, 1
NIS_STATUS_PWLIN_RANGE, NIS_STATUS_BIN_AMBIGUOUS = 1, 2      # device status words of a flow workspace (include/nis_hip.h)
NIS_BIN_AMBIGUOUS_FLAG = 0x40000000                           # bins_out entries: bin | flag
NIS_F64, NIS_F32 = 0, 1
NIS_F_CAMEL, NIS_F_NORMAL, NIS_F_ONE = 0, 1, 2
NIS_F_NONE = -1        # integrate_steps*: no integrand, X / J kept in the planes for a caller-side f
NIS_RAMBO_PDF_ACTIVE, NIS_RAMBO_TAU, NIS_RAMBO_LAB_BOOST = 1, 2, 4

ERRORS = {-1: "invalid argument", -2: "configuration outside the compiled kernel set",
          -3: "workspace too small", -4: "HIP runtime error", -5: "no device"}

LIB_PATH = os.environ.get("NIS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libnis_hip.so")   # NIS_LIB: A/B builds (tools/)

EXPORTS = ["nis_cell_param_count", "nis_cell_running_count", "nis_flow_create", "nis_flow_destroy",
           "nis_flow_workspace_bytes", "nis_flow_forward", "nis_flow_forward_sync", "nis_integrate_steps", "nis_integrate_steps_rng",
           "nis_integrate_rng_mode", "nis_philox_rows", "nis_integrate_steps_bins", "nis_flow_fusable", "nis_flow_backward_supported",
           "nis_flow_bnsave_count", "nis_flow_forward_train", "nis_flow_forward_hist", "nis_flow_backward_eval", "nis_flow_backward_workspace_bytes", "nis_flow_backward", "nis_flow_backward_loss", "nis_loss_terms", "nis_loss_workspace_bytes",
           "nis_flow_forward_hist_batched", "nis_loss_workspace_bytes_batched", "nis_loss_terms_batched",
           "nis_flow_backward_workspace_bytes_batched", "nis_flow_backward_loss_batched", "nis_integrand_moments", "nis_integrand_eval",
           "nis_moments_workspace_bytes",
           "nis_moments", "nis_merge_moments", "nis_rambo_forward", "nis_rambo_forward_pdf", "nis_rambo_moments", "nis_rambo_moments_workspace_bytes", "nis_pdf_xfx", "nis_prof_enable", "nis_prof_read", "nis_prof_disable", "nis_last_error", "nis_version", "nis_device_count"]


SYNC_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)       # nis_sync_fn


class CellDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d", C.c_int32), ("P", C.c_int32), ("K", C.c_int32),
                ("n_hidden", C.c_int32), ("width", C.c_int32 * NIS_MAX_HIDDEN), ("in_col", C.c_int32 * NIS_MAX_DIM)]


class RamboDesc(C.Structure):
    _fields_ = [("n_final", C.c_int32), ("flags", C.c_int32), ("initial_masses", C.c_double * 2),
                ("final_masses", C.c_double * NIS_MAX_LEGS), ("E_cm", C.c_double), ("pT_mincut", C.c_double),
                ("delR_mincut", C.c_double), ("rap_maxcut", C.c_double), ("bisect_levels", C.c_int32),
                ("reserved", C.c_int32)]


class NisError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libnis_hip.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NisError(f"{LIB_PATH} not found: build it with `python -m nf_amd.build` "
                       "(nf_amd has no CPU / PyTorch fallback for its kernels)")
    # torch FIRST: its wheel bundles its own libamdhip64 / libhsa-runtime64, and the library must bind to the SAME runtime the tensors,
    # streams and events it is handed live in.  Loaded before torch it pulls in /opt/rocm's copy, the process then holds two HIP runtimes
    # and the first launch fails with "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    L.nis_cell_param_count.restype = sz
    L.nis_cell_param_count.argtypes = [C.POINTER(CellDesc)]
    L.nis_cell_running_count.restype = sz
    L.nis_cell_running_count.argtypes = [C.POINTER(CellDesc)]
    L.nis_flow_create.restype = i32
    L.nis_flow_create.argtypes = [C.POINTER(CellDesc), i32, i32, C.POINTER(i32), C.POINTER(vp)]
    L.nis_flow_destroy.restype = None
    L.nis_flow_destroy.argtypes = [vp]
    L.nis_flow_workspace_bytes.restype = sz
    L.nis_flow_workspace_bytes.argtypes = [vp, i64]
    L.nis_flow_forward.restype = i32
    L.nis_flow_forward.argtypes = [vp, vp, i32, i32, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, vp, i64, vp, vp,
                                   vp, sz, vp]
    L.nis_integrate_steps.restype = i32
    L.nis_integrate_steps.argtypes = [vp, vp, i32, i32, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, i32,
                                      C.POINTER(C.c_double), vp, i64, i32, vp, vp, vp, sz, vp]
    u64 = C.c_uint64
    L.nis_integrate_steps_rng.restype = i32
    L.nis_integrate_steps_rng.argtypes = [vp, i32, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, i32,
                                          C.POINTER(C.c_double), u64, u64, vp, vp, vp, i64, i32, vp, vp, sz, vp]
    L.nis_integrate_rng_mode.restype = i32
    L.nis_integrate_rng_mode.argtypes = [vp, i32, i64, i32, i32, i32]
    L.nis_integrate_steps_bins.restype = i32
    L.nis_integrate_steps_bins.argtypes = [vp, vp, i32, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i64, i32,
                                           C.POINTER(C.c_double), vp, i64, vp, vp, vp, sz, vp]
    L.nis_flow_fusable.restype = i32
    L.nis_flow_fusable.argtypes = [vp]
    L.nis_philox_rows.restype = i32
    L.nis_philox_rows.argtypes = [u64, u64, i32, i64, i32, vp, vp]
    L.nis_flow_backward_supported.restype = i32
    L.nis_flow_backward_supported.argtypes = [vp]
    L.nis_flow_bnsave_count.restype = sz
    L.nis_flow_bnsave_count.argtypes = [vp]
    L.nis_flow_forward_train.restype = i32
    L.nis_flow_forward_train.argtypes = [vp, vp, i32, i32, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i64, vp, i64, vp, vp,
                                         vp, sz, vp]
    L.nis_flow_forward_hist.restype = i32
    L.nis_flow_forward_hist.argtypes = [vp, vp, i32, i32, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, vp, i64, vp, vp,
                                        vp, sz, vp]
    L.nis_flow_backward_eval.restype = i32
    L.nis_flow_backward_eval.argtypes = [vp, vp, i64, i64, i32, C.POINTER(vp), vp, vp, C.POINTER(vp), vp, vp, sz, vp]
    L.nis_flow_backward_workspace_bytes.restype = sz
    L.nis_flow_backward_workspace_bytes.argtypes = [vp, i64]
    L.nis_flow_backward.restype = i32
    L.nis_flow_backward.argtypes = [vp, vp, i64, i64, i32, C.POINTER(vp), vp, vp, C.POINTER(vp), vp, vp, sz, vp]
    L.nis_flow_backward_loss.restype = i32
    L.nis_flow_backward_loss.argtypes = [vp, vp, i64, i64, i32, C.POINTER(vp), vp, vp, vp, vp, vp, i32, C.POINTER(vp), vp, sz, vp]
    L.nis_loss_workspace_bytes.restype = sz
    L.nis_loss_workspace_bytes.argtypes = []
    L.nis_loss_terms.restype = i32
    L.nis_loss_terms.argtypes = [vp, vp, i64, i32, i32, C.POINTER(C.c_double), vp, vp, i64, i64, vp, i32, i32, vp, vp, vp, sz, vp]
    L.nis_flow_forward_hist_batched.restype = i32
    L.nis_flow_forward_hist_batched.argtypes = [vp, vp, i32, i32, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, i64, vp, i64, vp, vp,
                                                vp, sz, vp]
    L.nis_loss_workspace_bytes_batched.restype = sz
    L.nis_loss_workspace_bytes_batched.argtypes = [i32, i64]
    L.nis_loss_terms_batched.restype = i32
    L.nis_loss_terms_batched.argtypes = [vp, vp, i64, i64, i64, C.POINTER(i32), i32, i32, C.POINTER(C.c_double), vp, vp, i64, i64, i32, i64,
                                         vp, i32, i32, vp, vp, vp, sz, vp]
    L.nis_flow_backward_workspace_bytes_batched.restype = sz
    L.nis_flow_backward_workspace_bytes_batched.argtypes = [vp, i32, i64]
    L.nis_flow_backward_loss_batched.restype = i32
    L.nis_flow_backward_loss_batched.argtypes = [vp, vp, i64, i32, i64, C.POINTER(vp), vp, i32, vp, vp, vp, vp, i32, C.POINTER(vp), vp, sz, vp]
    L.nis_flow_forward_sync.restype = i32
    L.nis_flow_forward_sync.argtypes = [vp, vp, i32, i32, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i64, vp, i64, vp, vp, sz, vp,
                                        SYNC_FN, vp, vp, i32, i32]
    L.nis_integrand_eval.restype = i32
    L.nis_integrand_eval.argtypes = [i32, C.POINTER(C.c_double), vp, i64, i64, i32, vp, vp]
    L.nis_integrand_moments.restype = i32
    L.nis_integrand_moments.argtypes = [vp, i64, i64, i32, C.POINTER(i32), i32, C.POINTER(C.c_double), vp, vp, vp, sz, vp]
    L.nis_moments_workspace_bytes.restype = sz
    L.nis_moments_workspace_bytes.argtypes = []
    L.nis_moments.restype = i32
    L.nis_moments.argtypes = [vp, vp, i64, vp, vp, sz, vp]
    L.nis_merge_moments.restype = i32
    L.nis_merge_moments.argtypes = [vp, i32, i32, vp, vp]
    L.nis_rambo_forward.restype = i32
    L.nis_rambo_forward.argtypes = [C.POINTER(RamboDesc), vp, i64, i32, C.POINTER(i32), i64, vp, vp, vp, vp]
    L.nis_rambo_forward_pdf.restype = i32
    L.nis_rambo_forward_pdf.argtypes = [C.POINTER(RamboDesc), vp, vp, i64, i32, C.POINTER(i32), i64, vp, vp, vp, vp]
    L.nis_rambo_moments_workspace_bytes.restype = sz
    L.nis_rambo_moments_workspace_bytes.argtypes = [i32]
    L.nis_rambo_moments.restype = i32
    L.nis_rambo_moments.argtypes = [C.POINTER(RamboDesc), vp, vp, i64, i64, C.POINTER(i32), i32, i64, i32, vp, vp, vp, vp, sz, vp]
    L.nis_pdf_xfx.restype = i32
    L.nis_pdf_xfx.argtypes = [vp, vp, i64, vp, vp]
    L.nis_prof_enable.restype = i32
    L.nis_prof_enable.argtypes = [i32, i32]
    L.nis_prof_read.restype = i32
    L.nis_prof_read.argtypes = [C.POINTER(C.c_double), C.POINTER(i32)]
    L.nis_prof_disable.restype = i32
    L.nis_prof_disable.argtypes = []
    L.nis_last_error.restype = C.c_char_p
    L.nis_last_error.argtypes = []
    L.nis_version.restype = i32
    L.nis_device_count.restype = i32
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        detail = ""
        if rc == -4:
            detail = ": " + (lib().nis_last_error() or b"").decode()
        raise NisError(f"{what} failed: {ERRORS.get(rc, rc)}{detail}")


def require_gpu():
    """The kernels only exist for gfx950: fail loudly instead of computing anywhere else."""
    import torch
    if not torch.cuda.is_available():
        raise NisError("nf_amd needs a ROCm GPU (MI355X / gfx950): torch.cuda.is_available() is False "
                       "and there is no CPU fallback")
    lib()


def stream_ptr(device=None):
    """Current HIP stream of `device` (default: the current device) as the void* the C ABI takes."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device(device):
    """Context for every libnis_hip call: the kernels launch on the device that HOLDS the tensors (the reference selects
    the GPU by argument: create_model(dev=), integrate(dev=)), not on whatever device happens to be current."""
    import torch
    return torch.cuda.device(device)
