"""ctypes loader for oracle/_build/libme_oracle.so (TEST INFRASTRUCTURE, see me_oracle.c)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.dirname(_HERE)
_SO = os.path.join(_ORACLE_DIR, "_build", "libme_oracle.so")

_lib = None


def build(force=False):
    src = os.path.join(_ORACLE_DIR, "me_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64, i32, p = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
        L.orc_unique_first.restype = i64
        L.orc_unique_first.argtypes = [p, i64, p, p]
        L.orc_stride.restype = i64
        L.orc_stride.argtypes = [p, i64, i32, p, p]
        L.orc_kernel_map.restype = i64
        L.orc_kernel_map.argtypes = [p, i64, p, i64, p, i32, p]
        L.orc_pairs_from_nbr.restype = None
        L.orc_pairs_from_nbr.argtypes = [p, i64, i32, p, p, p]
        for name in ("orc_conv_fwd", "orc_conv_bwd_data", "orc_conv_bwd_weight"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [p, p, p, p, p, i32, i32, i32, p]
        for name in ("orc_gather_rows", "orc_scatter_add_rows"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [p, p, i64, i32, p]
        L.orc_set_threads.restype = None
        L.orc_set_threads.argtypes = [i32]
        L.orc_sparse_quantize.restype = i64
        L.orc_sparse_quantize.argtypes = [p, i64, p, i32, p, p, p]
        _lib = L
    return _lib
