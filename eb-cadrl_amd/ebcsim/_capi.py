"""Loader of libebcsim.so (the HIP kernels + C ABI).  No fallback: a missing library or a
missing device is an error, never a silent CPU path."""
import ctypes as C
import os

from . import _abi

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# EBCSIM_LIB: another build of the same library (tools/wave_timeline.py uses the trace build)
LIB_PATH = os.environ.get("EBCSIM_LIB") or os.path.join(_PKG, "lib", "libebcsim.so")

_lib = None

# name -> (restype, argtypes): every symbol include/ebcsim.h declares
SYMBOLS = {
    "ebc_abi_version": (C.c_int, []),
    "ebc_last_error": (C.c_char_p, []),
    "ebc_params_default": (C.c_int, [C.c_void_p]),
    "ebc_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ebc_destroy": (C.c_int, [C.c_void_p]),
    "ebc_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ebc_synchronize": (C.c_int, [C.c_void_p]),
    "ebc_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ebc_set_scene_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ebc_set_human_actions": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ebc_observe": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ebc_robot_orca": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_void_p]),
    "ebc_robot_orca_sim": (C.c_int, [C.c_void_p, C.c_int]),
    "ebc_robot_orca_sim_state": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ebc_step": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ebc_lookahead": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ebc_step_k": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ebc_get_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ebc_row_counts": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ebc_generate_scenes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]),
    "ebc_generate_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int]),
    "ebc_generate_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int]),
    "ebc_il_targets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                 C.c_void_p, C.c_void_p]),
    "ebc_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "ebc_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ebc_timing_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ebc_mlp2_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ebc_mlp2_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p]),
    "ebc_mlp2_create_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ebc_mlp2_forward_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ebc_mlp2_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "ebc_mlp2_forward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p]),
    "ebc_mlp2_destroy": (C.c_int, [C.c_void_p]),
    "ebc_mlp2_forward_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ebc_pair_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ebc_decision_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "ebc_decision_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p]),
    "ebc_pair_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ebc_pair_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ebc_pair_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ebc_pair_attend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}


class EbcError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libebcsim error %d: %s" % (code, message))
        self.code = code


def lib():
    """dlopen libebcsim.so once; raises if it has not been built (`make -C eb-cadrl_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libebcsim.so is not built (%s). Run __graft_entry__.build() or "
                "`make -C eb-cadrl_amd/csrc`; there is no CPU fallback." % LIB_PATH)
        try:
            # torch bundles its own libamdhip64.so.7; loading it first makes this process use ONE
            # HIP runtime whatever the import order (a second runtime sees no device)
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("EBCSIM_LIB") and not hasattr(L, name):
                continue  # another build given for measurement (tools/ab_bench.sh: an older round's library)
            fn = getattr(L, name)  # AttributeError if the product library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        if L.ebc_abi_version() != _abi.ABI_VERSION:
            raise ImportError("libebcsim.so ABI %d != bindings %d" % (L.ebc_abi_version(), _abi.ABI_VERSION))
        _lib = L
    return _lib


def csrc_sha256():
    """SHA-256 over the kernel sources the library is built from (csrc/*.h, *.hip, Makefile, the public header):
    committed PMC measurements (profiles/traffic_*.json) name the sources they were taken on, and bench.py
    refuses them once the kernels have changed."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_PKG, "csrc")
    files = sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith((".h", ".hip")) or f == "Makefile")
    files.append(os.path.join(os.path.dirname(_PKG), "include", "ebcsim.h"))
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def check(rc):
    if rc != 0:
        raise EbcError(rc, lib().ebc_last_error().decode("utf-8", "replace"))
