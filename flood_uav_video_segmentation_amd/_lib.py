"""ctypes binding of libfloodseg.so (the C ABI declared in include/floodseg.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load, importing
the ops raises.  torch is imported first on purpose -- torch ships its own libamdhip64 (same
soname), and device pointers / streams are only interchangeable inside ONE HIP runtime instance.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL so both share torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfloodseg.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_void = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float


class FsConfig(ctypes.Structure):
    _fields_ = [("arch", c_int), ("layers", c_int), ("classes", c_int), ("patch", c_int), ("d_model", c_int),
                ("n_layers", c_int), ("dec_layers", c_int), ("image_size", c_int), ("flags", c_int), ("winograd_tile", c_int)]


ARCH_PSPNET = 0
ARCH_DEEPLABV3 = 1
ARCH_SEGMENTER = 2
OPT_NO_WINOGRAD = 1      # FS_OPT_NO_WINOGRAD
OPT_NO_FUSED_HEAD = 2    # FS_OPT_NO_FUSED_HEAD
OPT_NO_FUSED_SHORTCUT = 4  # FS_OPT_NO_FUSED_SHORTCUT
OPT_NO_FUSED_WINOGRAD = 8  # FS_OPT_NO_FUSED_WINOGRAD
OPT_NO_SPLIT_BF16 = 16  # FS_OPT_NO_SPLIT_BF16
OPT_NO_RES_TOUCH = 128  # FS_OPT_NO_RES_TOUCH
OPT_NO_FUSED_POOL = 256  # FS_OPT_NO_FUSED_POOL
OPT_NO_FUSED_QKV = 1024  # FS_OPT_NO_FUSED_QKV
CONV_CHUNK_MAJOR = 0x400  # FS_CONV_CHUNK_MAJOR

# name -> (restype, argtypes); must list every symbol of include/floodseg.h (+ fs_test_hooks of include/floodseg_test.h)
_SIGNATURES = {
    "fs_version": (c_int, []),
    "fs_last_error": (ctypes.c_char_p, []),
    "fs_create": (c_int, [ctypes.POINTER(FsConfig), ctypes.POINTER(c_void)]),
    "fs_destroy": (c_int, [c_void]),
    "fs_load_weight": (c_int, [c_void, ctypes.c_char_p, c_void, ctypes.POINTER(c_i64), c_int, c_int, c_void]),
    "fs_finalize": (c_int, [c_void, c_void]),
    "fs_feature_shape": (c_int, [c_void, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "fs_workspace_bytes": (ctypes.c_size_t, [c_void, c_int, c_int, c_int]),
    "fs_reserve": (c_int, [c_void, c_int, c_int, c_int, c_void]),
    "fs_reserved_bytes": (ctypes.c_size_t, [c_void]),
    "fs_encoder_forward": (c_int, [c_void, c_void, c_int, c_int, c_int, c_void, c_void]),
    "fs_decoder_forward": (c_int, [c_void, c_void, c_int, c_int, c_int, c_void, c_void]),
    "fs_segment_forward": (c_int, [c_void, c_void, c_int, c_int, c_int, c_void, c_void]),
    "fs_encoder_forward2": (c_int, [c_void, c_void, c_int, c_void, c_int, c_int, c_int, c_void, c_void]),
    "fs_segment_forward2": (c_int, [c_void, c_void, c_int, c_void, c_int, c_int, c_int, c_void, c_void]),
    "fs_segment_crops": (c_int, [c_void, c_void, c_void, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_int,
                                 c_void, c_void]),
    "fs_profile_enable": (c_int, [c_void, c_int]),
    "fs_profile_dump": (c_int, [c_void, ctypes.c_char_p, ctypes.c_size_t]),
    "fs_grid_sample_nchw": (c_int, [c_void, c_int, c_int, c_int, c_int, c_void, c_int, c_int, c_void, c_int, c_void]),
    "fs_grid_sample_nhwc": (c_int, [c_void, c_int, c_int, c_int, c_int, c_int, c_void, c_int, c_int, c_void, c_int, c_int, c_void]),
    "fs_resize_bilinear_nchw": (c_int, [c_void, c_int, c_int, c_int, c_void, c_int, c_int, c_int, c_void]),
    "fs_resize_bilinear_nhwc": (c_int, [c_void, c_int, c_int, c_int, c_int, c_int, c_void, c_int, c_int, c_int, c_int, c_void]),
    "fs_blend": (c_int, [c_void, c_f32, c_void, c_f32, c_void, c_i64, c_void]),
    "fs_seg_tail": (c_int, [c_void, c_void, ctypes.POINTER(c_void), ctypes.POINTER(c_void), c_int, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_int, c_int, c_void, c_void, c_void, c_void]),
    "fs_feat_tail": (c_int, [c_void, c_void, c_int, c_int, c_int, ctypes.POINTER(c_void), ctypes.POINTER(c_void), c_int, c_int, c_void, c_int, c_int,
                             c_int, c_int, c_void, c_void, c_void]),
    "fs_seg_tail_accumulate": (c_int, [c_void, c_void, ctypes.POINTER(c_void), ctypes.POINTER(c_void), c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_void, c_void, c_int, c_int, c_int, c_int, c_void, c_void]),
    "fs_canvas_resize_argmax": (c_int, [c_void, c_int, c_int, c_int, c_int, c_void, c_int, c_int, c_void]),
    "fs_crop_grids": (c_int, [ctypes.POINTER(c_void), c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                              c_int, c_int, c_void, c_void]),
    "fs_crops_fuse": (c_int, [c_void, c_void, c_void, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)] + [c_int] * 9 + [c_void, c_void, c_int, c_int,
                              c_void, c_void]),
    "fs_argmax_u8": (c_int, [c_void, c_int, c_int, c_i64, c_void, c_void]),
    "fs_resize_crop": (c_int, [c_void, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void, c_void, c_int, c_int, c_void]),
    "fs_resize_argmax_u8": (c_int, [c_void, c_int, c_int, c_int, c_int, c_void, c_int, c_int, c_void]),
    "fs_iou_hist": (c_int, [c_void, c_void, c_i64, c_int, c_int, c_void, c_void]),
    "fs_colorize": (c_int, [c_void, c_void, c_int, c_void, c_i64, c_void]),
    "fs_mv_to_grids": (c_int, [c_void, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void, c_void, c_void, c_void]),
    "fs_softmax_accumulate": (c_int, [c_void, c_int, c_int, c_int, c_int, c_void, c_void, c_int, c_int, c_int, c_int, c_void]),
    "fs_canvas_finish": (c_int, [c_void, c_void, c_int, c_int, c_i64, c_void, c_void]),
    "fs_test_hooks": (c_void, []),
}

# members of fs_test_api (include/floodseg_test.h), in declaration order after `size`: name -> (restype, argtypes).  Reached as
# load().fs_<name>(...) like any exported function -- tests and tools do not care that they come from the table fs_test_hooks() returns.
_HOOKS = [
    ("pack_conv_weight", c_int, [c_void, c_void, c_int, c_int, c_int, c_int, c_void]),
    ("conv2d_nhwc", c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_int, c_void, c_int] + [c_int] * 12 + [c_void]),
    ("split_bf16x3", c_int, [c_void, c_i64, c_void, c_void]),
    ("conv2d_nhwc_split", c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_int, c_void, c_int] + [c_int] * 12 + [c_void]),
    ("attention_workspace_floats", ctypes.c_size_t, [c_int] * 4),
    ("attention", c_int, [c_void, c_void, c_int, c_int, c_int, c_f32, c_int, c_void, c_void]),
    ("winograd_workspace_floats", ctypes.c_size_t, [c_int] * 7),
    ("conv3x3_winograd_nhwc", c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_int] + [c_int] * 8 + [c_void, c_void]),
    ("winograd_fused_workspace_floats", ctypes.c_size_t, [c_int, c_int]),
    ("conv3x3_winograd_fused_nhwc", c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_int] + [c_int] * 7 + [c_void, c_void]),
    ("conv3x3_winograd_fused_pool_nhwc", c_int, [c_void, c_int, c_void, c_void, c_void, c_void] + [c_int] * 5 + [c_void, c_void]),
    ("stem_conv_nchw", c_int, [c_void, c_void, c_void, c_void, c_void] + [c_int] * 8 + [c_void]),
    ("stem_conv_nchw_split", c_int, [c_void, c_void, c_void, c_void, c_void] + [c_int] * 8 + [c_void]),
    ("maxpool3x3s2_nhwc", c_int, [c_void, c_void, c_int, c_int, c_int, c_int, c_void]),
    ("adaptive_avgpool_nhwc", c_int, [c_void, c_int, c_void, c_int, c_int, c_int, c_int, c_int, c_void]),
    ("nchw_to_nhwc", c_int, [c_void, c_void, c_int, c_int, c_int, c_void]),
    ("nhwc_to_nchw", c_int, [c_void, c_void, c_int, c_int, c_int, c_void]),
]


class FsTestApi(ctypes.Structure):
    _fields_ = [("size", ctypes.c_size_t)] + [(name, ctypes.CFUNCTYPE(res, *args)) for name, res, args in _HOOKS]


class _Library:
    """The loaded library: exported functions as attributes (ctypes), and the op-level test hooks of fs_test_hooks() under the names
    fs_<member> (so `lib.fs_conv2d_nhwc(...)` works whether a symbol is exported or lives in the table)."""

    def __init__(self, cdll):
        self._cdll = cdll
        self._hooks = None

    def __getattr__(self, name):
        if name.startswith("fs_") and name != "fs_test_hooks" and any(name == "fs_" + h[0] for h in _HOOKS):
            if self._hooks is None:
                table = ctypes.cast(self._cdll.fs_test_hooks(), ctypes.POINTER(FsTestApi)).contents
                if table.size < ctypes.sizeof(FsTestApi):
                    raise RuntimeError(f"floodseg: the library's test-hook table ({table.size} B) is older than this binding ({ctypes.sizeof(FsTestApi)} B)")
                self._hooks = table
            fn = getattr(self._hooks, name[3:])
            setattr(self, name, fn)
            return fn
        return getattr(self._cdll, name)


_lib = None
ALLOW_MISSING = False  # development A/B against an OLDER build only (bench.py --lib): symbols it lacks are skipped, not an error


def load():
    """Load libfloodseg.so once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback for this path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        if ALLOW_MISSING and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = _Library(lib)
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def hook_names():
    """Members of fs_test_api after `size`, in declaration order."""
    return [h[0] for h in _HOOKS]


def check(rc):
    if rc != 0:
        msg = load().fs_last_error()
        raise RuntimeError("floodseg: " + (msg.decode() if msg else f"error {rc}"))


def stream_ptr(device=None):
    """Current HIP stream of `device` (default: the current device) as the void* the C ABI takes."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def one_device(*tensors, handle_device=None, what="floodseg"):
    """All tensors (None entries skipped) must live on ONE HIP device -- and on the handle's, when given.  Returns it.
    The library launches on the calling thread's current device, so every wrapper runs its call under
    `with torch.cuda.device(dev)`: a cuda:1 tensor while cuda:0 is current must never reach a device-0 launch."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"{what}: tensors must live on the GPU (no CPU fallback exists)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"{what}: tensors on different devices ({dev} and {t.device})")
    if dev is None:
        raise RuntimeError(f"{what}: no device tensor given")
    if handle_device is not None and dev != handle_device:
        raise RuntimeError(f"{what}: tensor on {dev} but the network's weights live on {handle_device}")
    return dev


def ptr(t):
    """Device (or host) pointer of a tensor; None -> NULL."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())
