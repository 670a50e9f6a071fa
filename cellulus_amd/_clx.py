"""ctypes binding of libclx.so — the C ABI declared in ``include/clx.h``.

There is deliberately NO fallback: if the library is missing, or a kernel entry
point is called without a HIP device, an exception is raised.  The product
path never computes on the CPU.
"""

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_int, c_longlong, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libclx.so")


class ClxError(RuntimeError):
    """A libclx entry point returned a negative status."""


class ClxSrc(Structure):
    """``clx_src`` (include/clx.h)."""

    _fields_ = [
        ("ptr", c_void_p),
        ("C", c_int),
        ("ld", c_int),
        ("D", c_int),
        ("H", c_int),
        ("W", c_int),
        ("oz", c_int),
        ("oy", c_int),
        ("ox", c_int),
        ("fz", c_int),
        ("fy", c_int),
        ("fx", c_int),
    ]


class ClxPackJob(Structure):
    """``clx_pack_job`` (include/clx.h)."""

    _fields_ = [("w", c_void_p), ("wp", c_void_p), ("cout", c_int), ("cin", c_int), ("taps", c_int),
                ("cin_pad", c_int), ("cout_pad", c_int), ("mode", c_int)]


class ClxConvDesc(Structure):
    """``clx_conv_desc`` (include/clx.h)."""

    _fields_ = [
        ("nsrc", c_int),
        ("src", ClxSrc * 2),
        ("B", c_int),
        ("ID", c_int),
        ("IH", c_int),
        ("IW", c_int),
        ("KD", c_int),
        ("KH", c_int),
        ("KW", c_int),
        ("PD", c_int),
        ("PH", c_int),
        ("PW", c_int),
        ("N", c_int),
        ("wpack", c_void_p),
        ("bias", c_void_p),
        ("relu", c_int),
        ("mask", c_void_p),
        ("ld_mask", c_int),
        ("out", c_void_p),
        ("ld_out", c_int),
        ("accumulate", c_int),
        ("algo", c_int),
        ("workspace", c_void_p),
        ("workspace_bytes", c_size_t),
        ("vcache", c_void_p),
        ("vcache_valid", c_int),
        ("c_real", c_int),
        ("dy_vcache", c_void_p),
        ("gate_out", c_void_p),
        ("ld_gate", c_int),
        ("mask_bits", c_void_p),
        ("ld_mask_bits", c_int),
        ("det_turns", c_void_p),
        ("adjoint", c_int),
        ("pool_out", c_void_p),
        ("ld_pool", c_int),
        ("tile_list", c_void_p),
        ("tile_count", c_int),
        ("precision", c_int),
        ("wplanes", c_void_p),
        ("aplanes", c_void_p),
        ("aplanes_valid", c_int),
        ("dyplanes", c_void_p),
        ("dyplanes_valid", c_int),
        ("out_planes", c_void_p),
        ("out_colsum", c_void_p),
    ]


_P = c_void_p
_I = c_int
_LL = c_longlong
_D = c_double

# name -> (restype, argtypes); mirrors include/clx.h one to one
PROTOTYPES = {
    "clx_last_error": (c_char_p, []),
    "clx_abi_version": (_I, []),
    "clx_device_count": (_I, []),
    "clx_profile_enable": (_I, [_I]),
    "clx_profile_read": (_I, [_I, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "clx_profile_clock": (_I, [POINTER(c_double), POINTER(c_double), _I]),
    "clx_conv_fwd": (_I, [POINTER(ClxConvDesc), _P]),
    "clx_conv_wgrad": (_I, [POINTER(ClxConvDesc), _P, _I, _P, _P, _P]),
    "clx_chain64_fwd": (_I, [_P, _I, _LL, _P, _P, _P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _P, _I, _P]),
    "clx_chain64_bwd": (_I, [_P, _I, _I, _P, _I, _P, _I, _I, _LL, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "clx_pack_weights": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "clx_pack_weights_batch": (_I, [_P, _I, _LL, _P]),
    "clx_unpack_wgrad": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "clx_unpack_wgrad_wino": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "clx_conv_workspace_bytes": (c_size_t, [POINTER(ClxConvDesc), _I]),
    "clx_conv_vcache_bytes": (c_size_t, [POINTER(ClxConvDesc), _I]),
    "clx_conv_sp_covers": (_I, [POINTER(ClxConvDesc)]),
    "clx_planes_bytes": (c_size_t, [_LL, _I]),
    "clx_split_planes": (_I, [_P, _LL, _LL, _I, _P, _P]),
    "clx_join_planes": (_I, [_P, _LL, _I, _P, _LL, _P]),
    "clx_gemm_planes": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _I, _P]),
    "clx_wgrad_planes": (_I, [_P, _P, _LL, _I, _I, _P, _I, _P]),
    "clx_conv_fused_applicable": (_I, [POINTER(ClxConvDesc)]),
    "clx_conv_fused_workspace_bytes": (c_size_t, [POINTER(ClxConvDesc)]),
    "clx_planar_to_pixel": (_I, [_P, _P, _I, _I, _LL, _I, _P]),
    "clx_pixel_to_planar": (_I, [_P, _P, _I, _I, _LL, _I, _P]),
    "clx_depth_to_space": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "clx_space_to_depth": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "clx_subpixel_split_weights": (_I, [_P, _P, _P] + [_I] * 10 + [_P]),
    "clx_subpixel_fold_grads": (_I, [_P, _P, _P] + [_I] * 10 + [_P]),
    "clx_maxpool_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "clx_changed_rows_workspace": (c_size_t, [_I, _I, _I, _I]),
    "clx_changed_rows": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _LL, _P, _P]),
    "clx_changed_tiles": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _LL, _P]),
    "clx_grey_rows": (_I, [_P, _I, _I, _I, _I, _I, _P, _LL, _P, _P, _I, _I, _P, _I, _P]),
    "clx_gather_rows": (_I, [_P, _I, _P, _LL, _I, _P, _I, _P]),
    "clx_scatter_rows": (_I, [_P, _I, _P, _LL, _I, _P, _I, _P]),
    "clx_broadcast_rows": (_I, [_P, _LL, _P, _I, _P]),
    "clx_maxpool_bwd": (_I, [_P, _P, _P, _P] + [_I] * 7 + [_P] + [_I] * 8 + [_P]),
    "clx_upsample_bwd": (_I, [_P] + [_I] * 8 + [_P, _P] + [_I] * 8 + [_P]),
    "clx_gather_add_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "clx_gather_add_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "clx_oce_loss_fwd_bwd": (_I, [_P, _P, _P, _P, _LL, _I, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P]),
    "clx_conv_wgrad_turns_bytes": (c_size_t, [POINTER(ClxConvDesc)]),
    "clx_colsum_scratch_bytes": (c_size_t, [_I]),
    "clx_colsum_ordered": (_I, [_P, _I, _LL, _I, _P, _P, _P]),
    "clx_oce_pairs_det_scratch_bytes": (c_size_t, [_I, _I, _LL]),
    "clx_oce_pairs_fused_det": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float, _P, _P]),
    "clx_oce_pairs_fused": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, ctypes.c_float, ctypes.c_float, _P]),
    "clx_sample_pairs": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, POINTER(c_int), ctypes.c_ulonglong,
                              ctypes.c_ulonglong, _P]),
    "clx_adam_step": (_I, [_P, _P, _P, _P, _LL, _D, _D, _D, _D, _D, _I, _P]),
    "clx_adam_step_guarded": (_I, [_P, _P, _P, _P, _LL, _D, _D, _D, _D, _D, _I, _P, _P]),
    "clx_noise_stats": (_I, [_P, _P, _I, _I, _LL, _P]),
    "clx_noise_inject": (_I, [_P, _P, _P, _I, _I, _LL, ctypes.c_float, _P]),
    "clx_zero_many": (_I, [POINTER(_P), POINTER(_LL), _I, _P]),
    "clx_gather_rows_f64": (_I, [_P, _P, _LL, _I, _P, _P]),
    "clx_rows_extent_f64": (_I, [_P, _LL, _I, _P, _P]),
    "clx_noise_stats_minmax": (_I, [_P, _P, _I, _I, _LL, _P, _I, _P]),
    "clx_ms_prepare_workspace": (c_size_t, [_LL]),
    "clx_ms_prepare": (_I, [_P, _P, _D, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "clx_ms_prepare_f32": (_I, [_P, _P, _D, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "clx_ms_iterate": (_I, [_P, _I, _P, _I, _I, _D, _I, _P, _P, _P, _P]),
    "clx_ms_iterate_grid": (_I, [_P, _I, _P, POINTER(c_double), _D, _I, _I, _I, _P, _I, _I, _D, _I,
                                 _P, _P, _P, _P]),
    "clx_ms_assign": (_I, [_P, _P, _I, _P, _I, _I, _P, _P]),
    "clx_ms_assign_grid": (_I, [_P, _P, _I, _P, _I, _I, _P, _P, POINTER(c_double), _D, _I, _I, _I, _P, _P]),
    "clx_ms_assign_cells": (_I, [_P, _P, _I, _P, _I, _I, _P, _P, POINTER(c_double), _D, _I, _I, _I, _P, _P]),
    "clx_ms_assign_dense": (_I, [_P, _P, _I, _I, _P, _P, POINTER(c_double), _D, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P]),
    "clx_ms_dedup_centers": (_I, [_P, _P, _I, _I, _D, _P, POINTER(_I)]),
    "clx_sample_offsets_mt19937": (_I, [_P, POINTER(_I), _I, _I, _LL, _P, POINTER(_I)]),
    "clx_ms_bucket_workspace": (c_size_t, [_I, _LL]),
    "clx_ms_bucket": (_I, [_P, _I, _I, POINTER(c_double), _D, _I, _I, _I, _P, _P, _P, _P]),
    "clx_offset_magnitude": (_I, [_P, _P, _I, _LL, _P]),
    "clx_gaussian_filter_f64": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "clx_negate_f64": (_I, [_P, _P, _LL, _P]),
    "clx_peak_local_max": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P]),
    "clx_greedy_cluster": (_I, [_P, _P, _I, _I, _I, _D, _I, _D, _I, _P, _P, _P, _P]),
    "clx_cc_workspace": (c_size_t, [_LL]),
    "clx_cc_label_filter": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "clx_edt_workspace": (c_size_t, [_LL]),
    "clx_edt_sq": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "clx_grow_shrink": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "clx_minmax_f64": (_I, [_P, _LL, _P, _P]),
    "clx_histogram_f64": (_I, [_P, _LL, _P, _I, _P, _P]),
    "clx_minmax_f32": (_I, [_P, _LL, _P, _P]),
    "clx_histogram_f32": (_I, [_P, _LL, _P, _I, _P, _P]),
    "clx_inst_stats": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "clx_inst_histogram": (_I, [_P, _P, _I, _LL, _P, _I, _P, _I, _P, _P]),
    "clx_inst_refine": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "clx_lz4_decompress": (_LL, [_P, _LL, _P, _LL]),
    "clx_blosclz_decompress": (_LL, [_P, _LL, _P, _LL]),
    "clx_blosc_compress_bound": (_LL, [_LL]),
    "clx_blosc_compress_lz4": (_LL, [_P, _LL, _I, _I, _P, _LL]),
    "clx_unshuffle_bytes": (_I, [_P, _P, _LL, _I]),
    "clx_label_presence": (_I, [_P, _LL, _I, _P, _P, _P]),
    "clx_joint_histogram": (_I, [_P, _P, _LL, _P, _P, _I, _P, _P]),
}

MINMAX_DOUBLES = 2 + 2 * 512            # CLX_MINMAX_DOUBLES (include/clx.h): results + per-block partials
NOISE_MINMAX_FLOATS = 2 + 2 * 1024      # CLX_NOISE_MINMAX_FLOATS
ROWS_EXTENT_DOUBLES = 6 * (1 + 256)     # CLX_ROWS_EXTENT_DOUBLES
NOISE_MINMAX_MAX_T = 64                 # clx_noise_stats_minmax: predictions per pixel it keeps in registers

_lib = None


def load():
    """Load libclx.so (once) and attach the prototypes. Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ClxError(
            f"{LIB_PATH} is missing: build it with `python -m cellulus_amd._build` "
            "(hipcc --offload-arch=gfx950). cellulus_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().clx_last_error().decode("utf-8", "replace")
        raise ClxError(f"{what or 'libclx'} failed ({status}): {msg}")


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream on `device`."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def zero_many(*tensors):
    """Zero fills of contiguous device tensors on the current stream of their device, eight buffers per launch
    (clx_zero_many): the accumulators a step adds into, without one torch fill kernel each."""
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    if not ts:
        return
    dev = ts[0].device
    fast = []
    for t in ts:
        if not t.is_cuda:
            raise ClxError(f"zero_many: tensor on {t.device}; there is no CPU path")
        if t.device != dev:
            raise ClxError(f"zero_many: tensors on different devices ({dev} and {t.device}) in one call")
        nbytes = t.numel() * t.element_size()
        # (the kernel wants 16-byte-aligned buffers of whole 4-byte words: a sliced view or an odd size takes torch's fill)
        if t.is_contiguous() and t.data_ptr() % 16 == 0 and nbytes % 4 == 0:
            fast.append(t)
        else:
            t.zero_()
    if not fast:
        return
    ptrs = (_P * len(fast))(*[t.data_ptr() for t in fast])
    sizes = (_LL * len(fast))(*[t.numel() * t.element_size() for t in fast])
    call("clx_zero_many", ptrs, sizes, len(fast), stream_ptr(dev))


def zeros(shape, dtype, device):
    """torch.zeros on a HIP device without a torch fill kernel: an uninitialised tensor + clx_zero_many."""
    t = torch.empty(shape, dtype=dtype, device=device)
    if t.is_cuda and t.numel() > 0 and t.element_size() % 4 == 0:
        with torch.cuda.device(t.device):
            zero_many(t)
    else:
        t.zero_()
    return t


def require_device(t, name="tensor"):
    if not t.is_cuda:
        raise ClxError(
            f"{name} lives on {t.device}: cellulus_amd kernels run on HIP devices only "
            "(device strings 'cuda:N' denote HIP device N on ROCm); there is no CPU path."
        )


def call(name, *args):
    fn = getattr(load(), name)
    check(fn(*args), name)
