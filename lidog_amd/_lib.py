"""ctypes binding of liblidog_amd.so (the C ABI declared in include/lidog_amd.h).

There is NO fallback: if the HIP library is missing or a call fails, this raises."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LIDOG_SO: another build of the same library (kernel A/B runs); there is still no fallback behind it
SO_PATH = os.environ.get("LIDOG_SO") or os.path.join(_HERE, "_C", "liblidog_amd.so")

_i32, _i64, _p, _f, _d = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_float, ctypes.c_double

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "lidog_abi_version": [],
    "lidog_hash_capacity": [_i64],
    "lidog_coords_insert": [_p, _i64, _p, _p, _i64, _p, _p, _p, _p],
    "lidog_coords_insert_info": [_p, _i64, _p, _p, _i64, _p, _p, _p, _p],
    "lidog_coords_compact": [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p],
    "lidog_coords_stride": [_p, _i64, _i32, _p, _p, _i64, _p, _p, _p, _p, _p, _p],
    "lidog_kernel_map": [_p, _i64, _p, _p, _i64, _p, _i32, _p, _p],
    "lidog_bitmap_words": [_i32, _i32, _i32, _i32, _i64],
    "lidog_bitmap_set": [_p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p],
    "lidog_kernel_map_bits": [_p, _i64, _p, _p, _i64, _p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p],
    "lidog_kernel_map_pairs": [_p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p],
    "lidog_sconv_gemm_units": [_i32, _i32],
    "lidog_sconv_gemm": [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _i64, _p],
    "lidog_sconv_gemm_addend": [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p],
    "lidog_sconv_reduce": [_p, _p, _i64, _i32, _i32, _p, _p, _p, _p],
    "lidog_sconv_reduce_stats_ws": [_i64, _i32],
    "lidog_sconv_reduce_stats": [_p, _p, _i64, _i32, _i32, _p, _p, _p, _p, _d, _f, _f, _p, _p, _p, _p, _p],
    "lidog_kernel_map_rows": [_p, _i64, _i32, _i32, _p, _p, _p, _p],
    "lidog_sconv_reduce_rows": [_p, _p, _p, _i64, _i32, _p, _p, _p, _p],
    "lidog_sconv_reduce_rows_bn": [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _i32, _p, _p],
    "lidog_sconv_reduce_rows_stats": [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _d, _f, _f, _p, _p, _p, _p, _p],
    "lidog_sconv_reduce_rows_bwdstats": [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _d, _p, _p, _p],
    "lidog_bn_bwd_reduce_blocks": [_i64, _i32],
    "lidog_stats_max_blocks": [],
    "lidog_sconv_cin1": [_p, _p, _p, _p, _i64, _i32, _i32, _p, _p],
    "lidog_kernel_map_subset": [_p, _i64, _i32, _p, _i32, _p, _p],
    "lidog_kernel_map_sorted_ws": [_i64],
    "lidog_kernel_map_sorted": [_p, _i64, _i32, _p, _p, _p, _p, _p, _i64, _p],
    "lidog_sconv_os": [_p, _p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _p, _i32, _i32, _p, _p],
    "lidog_sconv_os_stats_ws": [_i64, _i32],
    "lidog_sconv_os_bn": [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p, _i32, _p, _p],
    "lidog_sconv_os_stats": [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _i32, _p, _p, _p, _d, _f, _f, _p, _p, _p, _p, _p],
    "lidog_sconv_wgrad": [_p, _p, _p, _p, _p, _i32, _p, _i32, _i32, _i32, _p, _p, _p],
    "lidog_sconv_gemm_in_bn": [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _i64, _p],
    "lidog_sconv_os_stats_in_bn": [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _i32, _p, _p, _p, _d, _f, _f, _p, _p, _p, _p,
                                   _p, _p, _p, _p, _i32, _p],
    "lidog_sconv_wgrad_in_bn": [_p, _p, _p, _p, _p, _i32, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p],
    "lidog_sconv_wgrad_slabs": [_i32, _i32, _i32],
    "lidog_sconv_wgrad_slots": [_i32, _i32, _i32],
    "lidog_set_sparse_core": [_i32],
    "lidog_get_sparse_core": [],
    "lidog_transpose_kernel": [_p, _i32, _i32, _i32, _p, _p],
    "lidog_transpose_batched": [_p, _p, _p, _i32, _i64, _p],
    "lidog_bn_reduce_ws": [_i32, _i64],
    "lidog_bn_stats": [_p, _i64, _i32, _i64, _p, _p, _d, _f, _f, _p, _p, _p, _p, _p],
    "lidog_bn_finalize": [_p, _d, _i32, _f, _f, _p, _p, _p, _p, _p],
    "lidog_bn_apply": [_p, _i64, _i32, _i64, _p, _p, _p, _p, _p, _i32, _p, _p],
    "lidog_bn_bwd_reduce": [_p, _p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _d, _p, _p, _p, _p, _p],
    "lidog_bn_bwd_apply": [_p, _p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _d, _p, _p, _p, _p, _p, _p],
    "lidog_relu_bits_words": [_i64, _i32],
    "lidog_bn_apply_bits": [_p, _i64, _i32, _i64, _p, _p, _p, _p, _p, _i32, _p, _p, _p],
    "lidog_bn_bwd_reduce_bits": [_p, _p, _p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _d, _p, _p, _p, _p, _p],
    "lidog_bn_bwd_apply_bits": [_p, _p, _p, _p, _i64, _i32, _i64, _p, _p, _p, _p, _d, _p, _p, _p, _p, _p, _p],
    "lidog_colsum": [_p, _i64, _i32, _p, _p, _p],
    "lidog_colsum_ws": [_i32],
    "lidog_bn_eval_invstd": [_p, _f, _i32, _p, _p],
    "lidog_cat2": [_p, _i32, _p, _i32, _i64, _p, _p],
    "lidog_split2": [_p, _i32, _i32, _i64, _p, _p, _p],
    "lidog_relu_fwd": [_p, _i64, _p, _p],
    "lidog_relu_bwd": [_p, _p, _i64, _p, _p],
    "lidog_add": [_p, _p, _i64, _p, _p],
    "lidog_bev_winner": [_p, _i64, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p],
    "lidog_bev_pool_fwd": [_p, _i32, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p],
    "lidog_bev_pool_bwd": [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p],
    "lidog_conv2d_fwd": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p],
    "lidog_conv2d_dgrad": [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p],
    "lidog_conv2d_wgrad": [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _i64, _p],
    "lidog_conv2d_support_ws": [_i32, _i32, _i32, _i32],
    "lidog_conv2d_support": [_p, _i32, _i32, _i32, _i32, _p, _p],
    "lidog_conv2d_fwd_sparse": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p],
    "lidog_conv2d_dgrad_sparse": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p],
    "lidog_conv2d_wgrad_sparse_ws": [_i32, _i32, _i32, _i32, _i32],
    "lidog_conv2d_wgrad_sparse": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _i64, _p],
    "lidog_voxel_floor": [_p, _i64, _f, _f, _f, _i32, _p, _p],
    "lidog_label_vote": [_p, _p, _p, _i64, _i64, _i32, _p, _p],
    "lidog_bev_label_raster": [_p, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p],
    "lidog_dice_ws": [_i32],
    "lidog_dice_fwd": [_p, _p, _i64, _i32, _i64, _i32, _f, _i32, _i32, _i32, _f, _p, _p, _p, _p],
    "lidog_dice_bwd": [_p, _p, _i64, _i32, _i64, _i32, _f, _i32, _i32, _p, _p, _p, _p],
    "lidog_adam_step": [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i32, _f, _p],
    "lidog_sgd_step": [_p, _p, _p, _i64, _f, _f, _f, _i32, _f, _p],
    "lidog_comm_unique_id_bytes": [],
    "lidog_comm_unique_id": [_p],
    "lidog_comm_init_rank": [_p, _i32, _i32, ctypes.POINTER(ctypes.c_void_p)],
    "lidog_comm_destroy": [_p],
    "lidog_allreduce_f32": [_p, _i64, _p, _p],
    "lidog_allreduce_f64": [_p, _i64, _p, _p],
    "lidog_comm_count": [_p],
    "lidog_peer_handle_bytes": [],
    "lidog_peer_mailbox_bytes": [_i32, _i32],
    "lidog_peer_mailbox_alloc": [_i64, ctypes.POINTER(ctypes.c_void_p), _p],
    "lidog_peer_mailbox_open": [_p, ctypes.POINTER(ctypes.c_void_p)],
    "lidog_peer_comm_create": [_i32, _i32, _i32, _p, _p, ctypes.POINTER(ctypes.c_void_p)],
    "lidog_peer_max_doubles": [_p],
    "lidog_peer_set_spin_limit": [_p, _i64],
    "lidog_peer_allreduce_f64": [_p, _p, _i64, _p],
    "lidog_peer_status": [_p],
    "lidog_peer_comm_destroy": [_p, _i32],
    "lidog_peer_rebind_stream": [_p, _p],
    "lidog_peer_calls": [_p],
    "lidog_peer_inject_skip_flag": [_p, _i64],
    "lidog_peer_mailbox_free": [_p],
    "lidog_peer_mailbox_close": [_p],
    "lidog_tiles_host": [_p, _i32, _i32, _i32, _p, _i64],
    "lidog_wgrad_items_host": [_p, _i32, _i64, _i32, _i32, _p, _p, _i64],
    "lidog_conv2d_support_work": [_p, _i32, _i32, _i32, _i32, _i32, _p, _p],
    "lidog_bn_apply_sync": [_p, _i64, _i32, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p],
    "lidog_trunk_fusions": [_i32],
    "lidog_trunk_in_bn_readers": [_p, _i32, _p, _i32, _p, _i32, _p, _i32, _p],
    "lidog_trunk_gemm_timing": [_i32],
    "lidog_trunk_gemm_timing_read": [_p],
    "lidog_trunk_work_read": [_p],
    "lidog_trunk_forward": [_p, _p, _i32, _p, _i32, _p, _i32, _p, _i32, _p, _p, _p, _i64, _p, _i64, _p, _p, _i32, _p, _p, _p],
    "lidog_trunk_backward": [_p, _p, _i32, _p, _i32, _p, _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _p,
                             _i64, _p, _p, _i32, _i32, _p, _p, _p],
}
_RESTYPES = {"lidog_hash_capacity": _i64, "lidog_sconv_reduce_stats_ws": _i64, "lidog_bn_reduce_ws": _i64,
             "lidog_dice_ws": _i64, "lidog_colsum_ws": _i64, "lidog_conv2d_support_ws": _i64, "lidog_conv2d_wgrad_sparse_ws": _i64,
             "lidog_tiles_host": _i64, "lidog_wgrad_items_host": _i64, "lidog_bitmap_words": _i64,
             "lidog_bn_bwd_reduce_blocks": _i64, "lidog_relu_bits_words": _i64,
             "lidog_peer_mailbox_bytes": _i64, "lidog_peer_calls": _i64,
             "lidog_kernel_map_sorted_ws": _i64, "lidog_sconv_os_stats_ws": _i64}

# lidog_abi_version() of the library these signatures were written against: a stale .so (or a header an external caller
# compiled against long ago) would take mis-sized arguments without any diagnostic
ABI_VERSION = 8

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -m lidog_amd.build` (hipcc, gfx950). "
                "lidog_amd has no CPU or torch fallback.")
        L = ctypes.CDLL(SO_PATH)
        L.lidog_last_error.restype = ctypes.c_char_p
        L.lidog_last_error.argtypes = []
        have = L.lidog_abi_version()
        if have != ABI_VERSION:
            raise RuntimeError(f"{SO_PATH} is ABI version {have}, this package expects {ABI_VERSION}: rebuild it with "
                               "`python -m lidog_amd.build --force`")
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        _lib = L
    return _lib


def ptr(t):
    """device pointer of a contiguous tensor (None -> NULL)"""
    if t is None:
        return None
    assert t.is_contiguous(), "lidog_amd kernels need contiguous tensors"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (the raw C query: the python Stream object costs
    ~9 us per call, 2 ms per training step over ~230 launches per pass)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_fns = {}


def call(name, *args):
    """call a status-returning entry point on the current torch stream; raise on failure"""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args, stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {load().lidog_last_error().decode()}")


def call_on(raw_stream, name, *args):
    """the same on an explicit hipStream_t (a stream's .cuda_stream): no switch of torch's current stream"""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    rc = fn(*args, raw_stream)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {load().lidog_last_error().decode()}")


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"lidog_amd: {what} must live on the GPU (got {t.device}); there is no CPU path")
