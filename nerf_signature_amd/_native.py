"""ctypes loader of libnerfsig.so (the C ABI declared in include/nerfsig.h).

The library is the product path; there is no fallback.  If it is missing the import-time error says
how to build it, and every call checks its return code and raises with nsig_last_error().
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libnerfsig.so")

_c = ctypes
_vp, _u32, _fl, _int, _sz = _c.c_void_p, _c.c_uint32, _c.c_float, _c.c_int, _c.c_size_t

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/nerfsig.h one to one.
SIGNATURES = {
    "nsig_abi_version": [],
    "nsig_last_error": [],
    "nsig_host_device_pointer": [_vp],
    "rg_sample_rays": [_vp, _u32, _vp, _fl, _fl, _fl, _fl, _u32, _u32, _u32, _vp, _u32, _u32, _c.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp],
    "rg_get_rays": [_vp, _fl, _fl, _fl, _fl, _u32, _u32, _vp, _u32, _u32, _vp, _vp, _vp],
    "rm_near_far_from_aabb": [_vp, _vp, _vp, _u32, _fl, _vp, _vp, _vp],
    "rm_sph_from_ray": [_vp, _vp, _fl, _u32, _vp, _vp],
    "rm_morton3D": [_vp, _u32, _vp, _vp],
    "rm_morton3D_invert": [_vp, _u32, _vp, _vp],
    "rm_packbits": [_vp, _u32, _fl, _vp, _vp],
    "rg_refresh_draw_scratch_bytes": [_u32, _u32],
    "rg_refresh_begin": [_vp, _u32, _vp],
    "rg_refresh_draw": [_vp, _vp, _u32, _u32, _vp, _vp, _c.c_uint64, _vp, _u32, _vp],
    "rg_refresh_points": [_vp, _vp, _u32, _u32, _fl, _fl, _c.c_uint64, _vp, _u32, _vp, _vp, _vp],
    "rg_refresh_scatter": [_vp, _vp, _u32, _fl, _vp, _vp],
    "rg_refresh_partials_bytes": [_u32],
    "rg_refresh_finish": [_vp, _vp, _u32, _fl, _vp, _fl, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp],
    "rm_march_train_scratch_bytes": [_u32, _u32],
    "rm_march_train_count": [_vp, _vp, _vp, _fl, _fl, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "rm_march_train_scan": [_vp, _u32, _vp, _vp, _vp],
    "rm_march_train_scan_blocks": [_u32],
    "rm_march_train_scan_wide": [_vp, _u32, _vp, _vp, _vp, _vp],
    "rm_march_train_write": [_vp, _vp, _fl, _fl, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rm_march_train_count_nf": [_vp, _vp, _vp, _fl, _vp, _fl, _fl, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "rm_march_train_scan_write_max_rays": [],
    "rm_march_train_scan_write": [_vp, _vp, _fl, _fl, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rm_composite_train_fwd": [_vp, _vp, _vp, _vp, _u32, _u32, _fl, _vp, _vp, _vp, _vp],
    "rm_composite_train_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _fl, _vp, _vp, _vp],
    "rm_march": [_u32, _u32, _vp, _vp, _vp, _vp, _fl, _fl, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "rm_composite": [_u32, _u32, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rm_compact_alive": [_vp, _u32, _vp, _vp, _vp],
    "rm_eval_begin": [_u32, _vp, _vp, _vp],
    "rm_eval_march": [_vp, _u32, _vp, _vp, _vp, _vp, _fl, _fl, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rm_eval_composite": [_vp, _u32, _fl, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rm_eval_compact": [_vp, _u32, _u32, _vp, _vp, _vp],
    "hg_encode_planes_rows": [_vp, _u32, _vp, _fl, _vp, _vp, _vp, _vp],
    "hg_encode_planes_mixed": [_vp, _u32, _vp, _fl, _vp, _vp, _vp, _vp],
    "field_fwd_rows": [_vp, _vp, _u32, _vp, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp],
    "hg_codebook_presum": [_vp, _u32, _vp, _vp],
    "hg_codebook_presum_sel": [_vp, _vp, _u32, _vp, _vp],
    "hg_encode_fwd": [_vp, _u32, _vp, _vp, _vp, _vp],
    "hg_codebook_encode_fwd": [_vp, _u32, _vp, _u32, _vp, _vp],
    "hg_codebook_bwd": [_vp, _u32, _vp, _vp, _vp],
    "hg_scatter_sliced": [_vp, _u32, _vp, _vp],
    "hg_fanout_grad": [_vp, _vp, _u32, _int, _vp],
    "hg_level_lookup": [_vp, _u32, _fl, _vp, _vp, _vp],
    "opt_codebook_adam": [_vp, _vp, _vp, _vp, _u32, _fl, _fl, _fl, _vp, _vp, _fl, _vp],
    "opt_codebook_adam_sel": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _fl, _fl, _fl, _fl, _vp, _vp],
    "opt_codebook_adam_sel_next": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _fl, _fl, _fl, _fl, _vp, _vp, _vp, _vp],
    "mlp_packed_bytes": [],
    "mlp_get_precision": [],
    "mlp_set_precision": [_int],
    "mlp_get_pipelined": [],
    "mlp_set_pipelined": [_int],
    "mlp_pack_weights": [_vp, _vp, _vp, _vp],
    "hg_planes_bytes": [_u32],
    "hg_encode_planes": [_vp, _u32, _fl, _vp, _vp, _vp, _vp],
    "hg_warm_tables": [_vp, _vp, _vp, _vp],
    "hg_encode_codebook_plane": [_vp, _u32, _fl, _vp, _vp, _int, _vp, _vp],
    "field_fwd": [_vp, _vp, _u32, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp],
    "field_color_fwd": [_vp, _vp, _u32, _vp, _vp, _vp],
    "opt_adam_dense_host": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _fl, _fl, _fl, _fl, _vp],
    "opt_adam_dense": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _fl, _fl, _fl, _fl, _vp, _vp],
    "rm_composite_train_finish_fwd": [_vp, _vp, _vp, _vp, _u32, _u32, _fl, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "rm_composite_train_finish_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _fl, _u32, _vp, _vp, _vp],
    "rm_finish_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "rm_finish_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "loop_step_begin": [_vp, _u32, _vp, _u32, _u32, _vp, _vp, _vp],
    "wm_loss_fwd": [_vp, _vp, _u32, _vp, _vp, _u32, _fl, _fl, _fl, _vp, _vp, _vp, _vp],
    "wm_loss_bwd": [_vp, _vp, _vp, _fl, _fl, _vp, _u32, _vp, _u32, _vp, _vp, _vp],
    "dec_bn_gelu_fwd": [_vp, _vp, _vp, _u32, _u32, _u32, _fl, _vp, _vp, _vp],
    "dec_bn_gelu_bwd": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp],
    "dec_workspace_bytes": [_u32, _u32, _u32, _u32],
    "dec_forward": [_vp, _u32, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _fl, _vp, _vp, _vp, _vp],
    "dec_backward": [_vp, _vp, _u32, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp],
    "dec_forward_train": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _fl, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _fl, _fl, _vp, _vp],
    "dec_backward_train": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp],
    "wm_distort_draw": [_u32, _c.c_uint64, _vp, _u32, _vp, _vp, _vp],
    "wm_distort_fwd": [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp],
    "wm_distort_bwd": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp],
    "wm_distort_geom_fwd": [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _vp, _vp, _vp],
    "wm_distort_geom_bwd": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _vp, _vp],
    "field_fwd_trace": [_vp, _vp, _u32, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "field_bwd_trace": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "hg_scatter_level": [_vp, _fl, _vp, _u32, _u32, _vp, _vp],
    "hg_scatter_binned_scratch_bytes": [_u32],
    "hg_scatter_binned": [_vp, _u32, _vp, _vp, _vp],
    "hg_scatter_levels_scratch_bytes": [_u32],
    "hg_scatter_levels": [_vp, _fl, _vp, _u32, _u32, _vp, _vp, _vp],
    "field_bwd": [_vp, _u32, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "field_bwd_planned": [_vp, _u32, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "hg_scatter_plan_bytes": [_u32],
    "hg_scatter_plan": [_vp, _u32, _fl, _vp, _vp],
    "hg_scatter_planned": [_vp, _u32, _vp, _vp],
    "field_fwd_trace_rows": [_vp, _vp, _u32, _vp, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "field_fwd_trace_f16": [_vp, _vp, _u32, _vp, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "field_bwd_trace_rows": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "field_wgrad_scratch_bytes": [_u32],
    "field_wgrad": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "field_bwd_wgrad_scratch_bytes": [_u32],
    "field_bwd_wgrad": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "field_bwd_wgrad_f16": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rm_composite_train_mse": [_vp, _vp, _vp, _vp, _u32, _u32, _fl, _vp, _vp, _vp, _u32, _vp, _u32, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "opt_ema_update": [_u32, _vp, _vp, _vp, _vp, _c.c_double, _vp],
    "clean_loss": [_vp, _vp, _u32, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _u32, _c.c_uint64, _vp],
    "hg_levels_plan_bytes": [_u32],
    "hg_levels_plan": [_vp, _u32, _vp, _fl, _vp, _vp],
    "hg_levels_scatter": [_vp, _u32, _vp, _fl, _vp, _u32, _vp, _vp, _vp],
    "hg_levels_scatter_adam": [_vp, _u32, _vp, _fl, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _fl, _fl, _fl, _fl, _vp, _vp],
}
_RESTYPES = {"nsig_last_error": _c.c_char_p, "nsig_host_device_pointer": _c.c_void_p, "rm_march_train_scratch_bytes": _sz, "mlp_packed_bytes": _sz, "hg_planes_bytes": _sz, "dec_workspace_bytes": _sz, "hg_scatter_levels_scratch_bytes": _sz, "hg_scatter_binned_scratch_bytes": _sz, "hg_scatter_plan_bytes": _sz, "field_wgrad_scratch_bytes": _sz, "field_bwd_wgrad_scratch_bytes": _sz, "hg_levels_plan_bytes": _sz, "rg_refresh_partials_bytes": _sz, "rg_refresh_draw_scratch_bytes": _sz}

_lib = None


class NativeError(RuntimeError):
    pass


def load():
    """Load libnerfsig.so; raises if the library is missing (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("NERFSIG_LIB", LIB_PATH)   # override: instrumented builds of the same sources (tools/dec_timing.py)
    if not os.path.exists(path):
        raise NativeError(f"{path} not found: the HIP extension is required (no fallback). "
                          f"Build it with `python -m nerf_signature_amd.build`.")
    _lib = ctypes.CDLL(path)
    _lib.nsig_last_error.restype = _c.c_char_p
    return _lib


_bound = {}


def fn(name):
    """Bound entry point `name` with the argtypes of include/nerfsig.h; raises if the library lacks it."""
    f = _bound.get(name)
    if f is None:
        lib = load()
        try:
            f = getattr(lib, name)
        except AttributeError as e:
            raise NativeError(f"libnerfsig.so does not export {name}; rebuild with `python -m nerf_signature_amd.build --force`") from e
        f.argtypes = SIGNATURES[name]
        f.restype = _RESTYPES.get(name, _int)
        _bound[name] = f
    return f


def verify_exports():
    """Every symbol declared in the header must be exported (used by the CPU test-suite and build())."""
    for name in SIGNATURES:
        fn(name)
    return sorted(SIGNATURES)


def call(name, *args):
    """Invoke an int-returning entry point; non-zero return raises with the library's message."""
    rc = fn(name)(*args)
    if rc != 0:
        msg = load().nsig_last_error().decode("utf-8", "replace")
        raise (ValueError if rc == 1 else NativeError)(f"{name} failed (code {rc}): {msg}")


def mlp_precision_name():
    """Human-readable arithmetic of the MLP kernels as currently selected (mlp_get_precision)."""
    return ("fp16 MFMA with f32 accumulate (MLPs)" if fn("mlp_get_precision")() == 1 else "split-bf16 MFMA (3 per product) with f32 accumulate (MLPs)")


def mlp_mfma_per_wave():
    """(MFMAs the forward MLP kernel issues per 32 points, operand dtype key) for the selected arithmetic."""
    return (24, "f16") if fn("mlp_get_precision")() == 1 else (72, "bf16")


def set_mlp_precision(name):
    """'f16' or 'bf16x3' (mlp_set_precision)."""
    call("mlp_set_precision", {"bf16x3": 0, "f16": 1}[name])


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous and on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("libnerfsig entry points take device tensors; got a CPU tensor")
    if not t.is_contiguous():
        raise ValueError("libnerfsig entry points take contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The current torch stream as a hipStream_t, so launches compose with autograd/autocast/RCCL streams.
    (torch's raw-stream query: the eager loops call this ~30 times per step, and torch.cuda.current_stream() builds a Stream object each time.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr_array(tensors):
    """Host array of device pointers (for the `*_host` pointer-table arguments)."""
    arr = (ctypes.c_void_p * len(tensors))(*[ptr(t) for t in tensors])
    return arr


def check(t, dtype, name, shape_last=None):
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if shape_last is not None and t.shape[-1] != shape_last:
        raise ValueError(f"{name}: expected last dimension {shape_last}, got {tuple(t.shape)}")
    return t
