"""ctypes binding of the HIP library (``vfa_amd/csrc/libvfa_hip.so``, C ABI in ``include/vfa_hip.h``).

There is deliberately NO fallback: if the library is missing or a launch fails the caller gets an
exception.  The product path never touches ``oracle/`` or any CPU implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VFA_AMD_LIB") or os.path.join(_HERE, "csrc", "libvfa_hip.so")  # (override: A/B runs of two builds)
ABI_VERSION = 9

CONV_KIND = {"MultiviewC": 0, "MultiviewX": 1, "Wildtrack": 2}
VOX_REFERENCE, VOX_LAYER_MAJOR = 0, 1
VOX_KERNEL_DIRECT, VOX_KERNEL_TAP_CACHE = 0x100, 0x200
BWD_ACCUMULATE = 1


FLAG_ROWS_ONLY, FLAG_SKIP_ROWS = 1 << 28, 1 << 29  # VFA_FLAG_ROWS_ONLY / VFA_FLAG_SKIP_ROWS (vfa_pool_collapse_relu_sum_f32)
FLAG_DUMP_VOX = 1 << 30  # VFA_FLAG_DUMP_VOX (both fused entry points; tests)


def collapse_flags(terms=0, reserved_cus=0):
    """`flags` of the MFMA collapse entry points: product arithmetic (0 / 2 = two fp16 pieces, the default of the fused frame
    kernels; 3 / 4 / 6 = the bf16 forms) | VFA_FLAG_RESERVED_CUS(n)."""
    return (int(terms) & 0xf) | ((int(reserved_cus) & 0xff) << 8)


_c_int, _c_float, _c_size_t, _vp, _c_longlong = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_longlong

# name -> argtypes; must list every symbol include/vfa_hip.h declares (tests/test_abi.py checks it)
SIGNATURES = {
    "vfa_abi_version": [],
    "vfa_integral_image_f32": [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp],
    "vfa_affine_relu_integral_image_f32": [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp],
    "vfa_box_params_f32": [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_int, _c_int,
                           _c_float, _c_float, _vp, _vp, _vp, _vp],
    "vfa_gather_f32": [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                       _vp],
    "vfa_project_gather_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                               _c_int, _c_int, _c_float, _c_float, _c_float, _c_float, _c_int, _vp],
    "vfa_project_collapse_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                 _c_int, _c_float, _c_float, _c_float, _c_float, _vp],
    "vfa_gather_workspace_bytes": [_c_int, _c_int, _c_int],
    "vfa_project_gather_ws_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_size_t, _c_int, _c_int, _c_int, _c_int, _c_int,
                                  _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float, _c_int, _vp],
    "vfa_project_gather_backward_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                        _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float, _c_int, _vp],
    "vfa_project_gather_backward_grid_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                             _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float, _c_int, _vp],
    "vfa_integral_image_backward_f32": [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp],
    "vfa_relu_mask_backward_f32": [_vp, _vp, _vp, _vp, _vp, _c_int, _c_size_t, _c_int, _vp],
    "vfa_bias_relu_accumulate_f32": [_vp, _vp, _vp, _c_int, _c_size_t, _c_int, _c_int, _vp],
    "vfa_scale_view_sum_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_size_t, _c_int, _c_int, _vp],
    "vfa_collapse_gemm_workspace_bytes": [_c_int, _c_int],
    "vfa_collapse_gemm_f32": [_vp, _vp, _vp, _vp, _c_size_t, _c_size_t, _c_int, _c_int, _c_int, _vp],
    "vfa_collapse_gemm_relu_backward_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_size_t, _c_int, _c_size_t, _c_int, _c_int, _c_int, _vp],
    "vfa_collapse_gemm_relu_backward_f16_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_size_t, _c_int, _c_size_t, _c_int, _c_int, _vp, _c_int, _vp,
                                                _vp, _c_int, _vp],
    "vfa_sliver_shifts_scratch_bytes": [_c_int, _c_int],
    "vfa_sliver_shifts_u8": [_vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float, _c_int, _c_int,
                             _c_int, _vp, _vp, _c_size_t, _vp],
    "vfa_collapse_relu_sum_f32": [_vp, _vp, _vp, _vp, _c_int, _c_size_t, _c_int, _c_int, _c_int, _c_int, _vp],
    "vfa_feature_stats_count": [_c_int, _c_int, _c_int],
    "vfa_integral_images_f32": [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _vp],
    "vfa_integral_images_hwc_f32": [_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _vp],
    "vfa_integral_absmax_f32": [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp],
    "vfa_lateral_conv_workspace_bytes": [_c_int, _c_int, _c_int],
    "vfa_lateral_conv_f32": [_vp, _vp, _vp, _vp, _vp, _c_float, _vp, _vp, _vp, _vp, _c_size_t, _c_int, _c_int, _c_int, _c_int, _vp],
    "vfa_grad_weight_workspace_bytes": [_c_longlong, _c_int],
    "vfa_grad_weight_f32": [_vp, _vp, _vp, _c_longlong, _c_int, _c_int, _vp, _c_size_t, _vp],
    "vfa_grad_input_workspace_bytes": [_c_int],
    "vfa_grad_input_f32": [_vp, _vp, _vp, _c_longlong, _c_int, _vp, _c_size_t, _vp],
    "vfa_lateral_convs_f32": [_c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp],
    "vfa_sort_vertices_f32": [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp],
    "vfa_bev_nms_f32": [_vp, _vp, _c_int, _c_int, _vp],
    "vfa_frame_workspace_bytes": [_c_int, _c_int, _c_int, _c_int],
    "vfa_frame_workspace_layout": [_c_int, _c_int, _c_int, _c_int, _vp, _vp],
    "vfa_frame_records_f32": [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float,
                              _c_int, _vp, _vp, _c_int, _vp, _c_size_t, _vp],
    "vfa_frame_boxes_f32": [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float,
                            _c_int, _vp, _vp, _c_size_t, _vp],
    "vfa_frame_cuts_f32": [_c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp, _c_size_t, _vp],
    "vfa_pool_windows_f32": [_vp, _vp, _c_size_t, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp],
    "vfa_pool_collapse_relu_sum_f32": [_vp, _vp, _vp, _vp, _c_size_t, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int,
                                       _vp],
    "vfa_pipe_workspace_bytes": [_c_int, _c_int, _c_int, _c_int, _c_int],
    "vfa_pipe_workspace_layout": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp],
    "vfa_pipe_boxes_f32": [_vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float,
                           _c_int, _vp, _c_int, _vp, _c_size_t, _vp],
    "vfa_pipe_cuts_f32": [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _vp, _c_size_t, _vp],
    "vfa_pipe_records_f32": [_vp, _vp, _vp, _c_int, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float,
                             _c_int, _vp, _vp, _c_int, _vp, _c_size_t, _vp],
    "vfa_pipe_collapse_relu_sum_f32": [_vp, _vp, _vp, _vp, _c_size_t, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int,
                                       _vp],
    "vfa_pipe_balance_f32": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_size_t, _vp],
}

_lib = None


class VFAHipError(RuntimeError):
    pass


def lib():
    """Load the HIP library once.  Raises if it has not been built (``python -m vfa_amd.build``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VFAHipError(
                f"{LIB_PATH} is missing: build it with `python -m vfa_amd.build` (hipcc --offload-arch=gfx950). "
                "vfa_amd has no CPU or PyTorch fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_size_t if name.endswith(("_bytes", "_count")) else ctypes.c_int
        got = handle.vfa_abi_version()
        if got != ABI_VERSION:
            raise VFAHipError(f"libvfa_hip.so has ABI version {got}, the Python side expects {ABI_VERSION}; rebuild")
        _lib = handle
    return _lib


def call(name, *args):
    """Call an entry point; a non-zero status becomes an exception (reference error behaviour = Python exceptions)."""
    status = getattr(lib(), name)(*args)
    if status != 0:
        raise VFAHipError(f"{name} failed with status {status}"
                          + (" (bad argument)" if status == 10001 else " (hipError_t)"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """HOST array of device pointers (None entries -> NULL), for the entry points that take one pointer per scale."""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def int_array(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def current_device_index():
    """Index of the current device, straight from the runtime binding: `torch.cuda.current_stream(device)` and friends go through
    `torch.cuda.is_available()` for anything but an int -- an `os.environ.get` each, 10 us per call, seven calls per frame: a third of
    the host time of a frame (round 5, cProfile of a one-camera frame)."""
    import torch
    return torch._C._cuda_getDevice()


def current_stream_handle():
    """hipStream_t of torch's current stream on the current device."""
    import torch
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:  # (an older torch)
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def current_stream(dev):
    """torch's current stream of `dev` as a Stream object, without the detour described above."""
    import torch
    return torch.cuda.current_stream(dev.index if dev.index is not None else current_device_index())


def require_device(*tensors):
    """The path runs on the GPU only; refuse CPU tensors loudly instead of falling back."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VFAHipError("vfa_amd runs on an AMD GPU (MI355X, gfx950) only; got a CPU tensor and there is no "
                              "CPU fallback")
