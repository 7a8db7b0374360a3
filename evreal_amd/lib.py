"""ctypes binding of libevreal_hip.so (the C ABI in include/evreal_hip.h).

PyTorch is used for device memory and streams only; every compute call goes through the
C ABI.  There is NO fallback: a missing library or a missing GPU raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libevreal_hip.so')

c_void_p, c_int, c_int64, c_size_t, c_uint, c_double, c_float, c_char_p = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_uint, ctypes.c_double,
    ctypes.c_float, ctypes.c_char_p)


class EvrError(RuntimeError):
    pass


class ModelDesc(ctypes.Structure):
    _fields_ = [('arch', c_int), ('num_bins', c_int), ('base_num_channels', c_int), ('num_encoders', c_int),
                ('num_residual_blocks', c_int), ('kernel_size', c_int), ('norm', c_int),
                ('use_upsample_conv', c_int), ('recurrent_block', c_int), ('final_activation', c_int),
                ('pad_multiple_log2', c_int), ('reserved', c_int * 5)]


class Tensor(ctypes.Structure):
    _fields_ = [('name', c_char_p), ('data_host', c_void_p), ('ndim', c_int), ('shape', c_int64 * 4)]


# every symbol include/evreal_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    'evr_last_error': (c_char_p, []),
    'evr_version': (c_int, []),
    'evr_device_info': (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_char_p, c_size_t]),
    'evr_stream_create_cu_masked': (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    'evr_stream_destroy': (c_int, [c_void_p]),
    'evr_event_create': (c_int, [ctypes.POINTER(c_void_p)]),
    'evr_event_destroy': (c_int, [c_void_p]),
    'evr_stream_wait_event': (c_int, [c_void_p, c_void_p]),
    'evr_model_set_gate': (c_int, [c_void_p, c_char_p, c_void_p]),
    'evr_voxelize_workspace_bytes': (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    'evr_voxelize': (c_int, [c_void_p] * 5 + [c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p,
                             c_void_p, c_size_t, c_void_p]),
    'evr_voxelize_dropped': (c_int, [c_void_p, ctypes.POINTER(c_int64), c_void_p]),
    'evr_voxelize_dropped_total': (c_int, [c_void_p, ctypes.POINTER(c_int64), c_void_p]),
    'evr_voxelize_raw': (c_int, [c_void_p] * 4 + [c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    'evr_voxelize_raw_windows': (c_int, [c_void_p] * 6 + [c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    'evr_event_tensor_normalize': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    'evr_model_create': (c_int, [ctypes.POINTER(ModelDesc), ctypes.POINTER(Tensor), c_int,
                                 ctypes.POINTER(c_void_p)]),
    'evr_model_destroy': (c_int, [c_void_p]),
    'evr_model_reset_states': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    'evr_model_release_shape': (c_int, [c_void_p]),
    'evr_model_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint, c_void_p]),
    'evr_model_read_tensor': (c_int, [c_void_p, c_char_p, c_void_p, c_int64, ctypes.POINTER(c_int64), c_void_p]),
    'evr_model_flops_per_step': (c_double, [c_void_p]),
    'evr_model_arith': (c_int, [c_void_p]),
    'evr_model_saturation': (c_int, [c_void_p, ctypes.POINTER(c_int64), c_char_p, c_size_t, c_int, c_void_p]),
    'evr_model_saturation_async': (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(c_int), c_void_p]),
    'evr_model_profile_enable': (c_int, [c_void_p, c_char_p]),
    'evr_model_profile_read': (c_int, [c_void_p, c_int, c_char_p, ctypes.POINTER(c_double), ctypes.POINTER(c_double),
                                       ctypes.POINTER(c_int64), ctypes.POINTER(c_int), c_void_p]),
    'evr_percentile_normalize_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'evr_percentile_normalize': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p,
                                         c_size_t, c_void_p]),
    'evr_hist_equalize_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'evr_hist_equalize': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'evr_metrics': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_uint, c_int, c_void_p, c_void_p, c_size_t,
                            c_void_p]),
    'evr_metrics_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'evr_lpips_create': (c_int, [ctypes.POINTER(Tensor), c_int, ctypes.POINTER(c_void_p)]),
    'evr_lpips_destroy': (c_int, [c_void_p]),
    'evr_lpips_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'evr_lpips_flops': (c_double, [c_void_p]),
    'evr_bayer_split': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'evr_color_merge': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'evr_split_pack': (c_int, [c_void_p, c_void_p, c_int64]),
    'evr_split_unpack': (c_int, [c_void_p, c_void_p, c_int64]),
    'evr_split_pack_device': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'evr_split_pack_weights': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'evr_h2_pack': (c_int, [c_void_p, c_void_p, c_int64]),
    'evr_h2_pack_weights': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'evr_h2_unpack': (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    'evr_h2_pack_device': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'evr_p6_pack': (c_int, [c_void_p, c_void_p, c_int64]),
    'evr_p6_pack_weights': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'evr_p6_unpack': (c_int, [c_void_p, c_void_p, c_int64]),
    'evr_p6_pack_device': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'evr_h2_act_exponent': (c_int, []),
    'evr_fastdiv_magic': (c_int, [ctypes.c_uint, c_void_p, c_void_p]),
    'evr_png_pool_create': (c_int, [c_int, c_int, ctypes.POINTER(c_void_p)]),
    'evr_png_pool_submit': (c_int, [c_void_p, c_char_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int64]),
    'evr_png_pool_wait': (c_int, [c_void_p, ctypes.POINTER(c_int64)]),
    'evr_png_pool_destroy': (c_int, [c_void_p]),
}

_lib = None


def load():
    """Load the shared library (once) and type every exported symbol.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel carries its own libamdhip64, and the process must end up with ONE HIP runtime -- loaded the other way
    # round (this library before torch, e.g. __graft_entry__.build() followed by smoke() in one process) the library's runtime sees
    # no device (hipGetDevice: error 100) while torch's does
    import torch  # noqa: F401
    path = os.environ.get('EVR_LIB') or LIB_PATH      # (EVR_LIB: another BUILD of this library -- timing-ablation variants, tools/ablate_wide.sh)
    if not os.path.exists(path):
        raise EvrError(f"{path} not found: build it with `python -m evreal_amd.build` "
                       "(there is no CPU fallback)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the ABI and the header drifted apart
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().evr_last_error().decode(errors='replace')
        raise EvrError(f"{what} failed ({rc}): {msg}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise EvrError("no HIP GPU visible: evreal_amd has no CPU path")
    return torch


def stream_ptr(stream=None):
    import torch
    if stream is not None:
        return c_void_p(stream.cuda_stream)
    # (the raw handle of the current stream without building a torch.cuda.Stream object: ~7 us -> <1 us per call, and a one-sequence
    # step makes six of these calls in ~410 us)
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
