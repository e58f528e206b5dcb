"""ctypes binding of libnmpc_amd.so (the C ABI declared in include/nmpc_amd.h).

There is no CPU fallback: if the shared library is missing or a call is made without a GPU the
package raises instead of computing anything on the host.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NMPC_AMD_LIB names another build of the library (A/B measurements of kernel variants); default: the in-tree one
LIB_PATH = os.environ.get("NMPC_AMD_LIB") or os.path.join(_HERE, "libnmpc_amd.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_void_p = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/nmpc_amd.h declares
SIGNATURES = {
    "nmpc_last_error": (ctypes.c_char_p, []),
    "nmpc_version": (ctypes.c_int, []),
    "nmpc_model_create": (ctypes.c_int, [ctypes.POINTER(c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_float_p),
                                         ctypes.POINTER(c_float_p), ctypes.POINTER(c_float_p)]),
    "nmpc_model_create_layers": (ctypes.c_int, [ctypes.POINTER(c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(c_float_p), ctypes.POINTER(c_float_p),
                                                ctypes.POINTER(c_float_p)]),
    "nmpc_model_destroy": (None, [c_void_p]),
    "nmpc_lpv_embed": (ctypes.c_int, [c_void_p, ctypes.c_int, c_void_p, c_void_p, c_void_p, c_float_p, c_float_p,
                                      ctypes.c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nmpc_quad_nodes": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_double, c_float_p, c_float_p]),
    "nmpc_lpv_rk4": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    ctypes.c_double, c_void_p, c_void_p, c_void_p]),
    "nmpc_condense": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nmpc_qp_solve": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "nmpc_mpc_qp": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int] + [c_void_p] * 7 + [c_double_p,
                                   ctypes.c_double, c_double_p] + [ctypes.c_double] * 6 + [c_void_p] * 9),
    "nmpc_ctrl_create": (ctypes.c_int, [ctypes.POINTER(c_void_p), c_void_p, c_void_p]),
    "nmpc_ctrl_destroy": (None, [c_void_p]),
    "nmpc_ctrl_reset": (ctypes.c_int, [c_void_p, c_double_p, ctypes.c_double, c_double_p, c_void_p]),
    "nmpc_ctrl_run": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int] + [c_void_p] * 11),
    "nmpc_target_select": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, c_void_p] + [ctypes.c_double] * 4 +
                           [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_double] + [c_void_p] * 6),
    "nmpc_target_select_static": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, c_void_p] + [ctypes.c_double] * 4 +
                                  [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_double] + [c_void_p] * 6),
    "nmpc_target_select_state": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, c_void_p, ctypes.c_double,
                                                ctypes.c_double, c_double_p, c_double_p, c_double_p, c_double_p,
                                                ctypes.c_double, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_double] + [c_void_p] * 6),
    "nmpc_target_workspace_bytes": (ctypes.c_ulonglong, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "nmpc_target_select_ws": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, c_void_p] + [ctypes.c_double] * 4 +
                              [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_double] + [c_void_p] * 6 +
                              [ctypes.c_ulonglong, c_void_p]),
    "nmpc_target_select_state_ws": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, c_void_p, ctypes.c_double,
                                                   ctypes.c_double, c_double_p, c_double_p, c_double_p, c_double_p,
                                                   ctypes.c_double, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_double] + [c_void_p] * 6 + [ctypes.c_ulonglong, c_void_p]),
    "nmpc_ctrl_profile": (ctypes.c_int, [c_void_p, ctypes.c_int]),
    "nmpc_ctrl_profile_read": (ctypes.c_int, [c_void_p, c_int_p, c_double_p, c_void_p]),
    "nmpc_ctrl_profile_read_steps": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, c_double_p, c_void_p]),
    "nmpc_ctrl_tail_from": (ctypes.c_int, [c_void_p]),
    "nmpc_ctrl_stats": (ctypes.c_int, [c_void_p, ctypes.POINTER(ctypes.c_longlong), c_void_p]),
    "nmpc_ctrl_set_status_log": (ctypes.c_int, [c_void_p, c_void_p]),
    "nmpc_ctrl_set_embed_stamps": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int]),
}

MAX_NODES = 256


class CtrlConfig(ctypes.Structure):
    """struct nmpc_ctrl_config of include/nmpc_amd.h"""
    _fields_ = [("struct_size", ctypes.c_int), ("batch", ctypes.c_int), ("Nc", ctypes.c_int), ("max_iterations", ctypes.c_int),
                ("n_sim", ctypes.c_int), ("tol", ctypes.c_double), ("rho", ctypes.c_double),
                ("Q", ctypes.c_double * 64), ("R", ctypes.c_double),
                ("u_min", ctypes.c_double), ("u_max", ctypes.c_double), ("y_min", ctypes.c_double),
                ("y_max", ctypes.c_double), ("u0", ctypes.c_double), ("ustd", ctypes.c_double),
                ("y0", ctypes.c_double), ("ystd", ctypes.c_double), ("dt", ctypes.c_double),
                ("plant", ctypes.c_double * 6), ("n_nodes", ctypes.c_int),
                ("lam", ctypes.c_float * MAX_NODES), ("w", ctypes.c_float * MAX_NODES),
                ("mode", ctypes.c_int), ("hard", ctypes.c_int),
                ("x_min", ctypes.c_double * 8), ("x_max", ctypes.c_double * 8),
                ("x_norm0", ctypes.c_double * 8), ("x_normstd", ctypes.c_double * 8),
                ("as_written", ctypes.c_int),
                ("n_uhist", ctypes.c_int), ("n_yhist", ctypes.c_int)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = ctypes.sizeof(CtrlConfig)


_lib = None


class NmpcError(RuntimeError):
    pass


def lib():
    """Load the HIP library once; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NmpcError("%s not found: build it with `make -C nmpc_amd/csrc` (hipcc, gfx950); "
                            "nmpc_amd has no CPU fallback" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        missing = [name for name in SIGNATURES if not hasattr(handle, name)]
        if missing:
            raise NmpcError("%s is stale: it does not export %s; rebuild with `make -C nmpc_amd/csrc`"
                            % (LIB_PATH, ", ".join(missing)))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise NmpcError("%s failed: %s" % (what, lib().nmpc_last_error().decode()))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise NmpcError("nmpc_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU fallback")
