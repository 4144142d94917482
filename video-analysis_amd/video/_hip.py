"""ctypes binding of libvideoanalysis_hip.so (include/videoanalysis_hip.h).

This is the ONLY compute backend of the package: there is no NumPy/CPU fallback.  If the
shared library is missing, or no MI355X is visible, every filter/analysis call raises
:class:`HipUnavailableError` instead of silently computing something else.
"""
import ctypes as C
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libvideoanalysis_hip.so")

VA_U8, VA_F32, VA_F64, VA_I16 = 0, 1, 2, 3
BG_NONE, BG_MEAN, BG_EMA, BG_STATIC = 0, 1, 2, 3
MORPH_ERODE, MORPH_DILATE = 0, 1
SHAPE_RECT, SHAPE_CROSS, SHAPE_ELLIPSE = 0, 1, 2
TAPS_CV4, TAPS_CV3 = 0, 1
TAP_RULES = {None: TAPS_CV4, "cv4": TAPS_CV4, "cv3": TAPS_CV3}
MAX_MORPH_OPS = 4
STATS_STRIDE = 16
STAT_NAMES = ("area", "m10", "m01", "m20", "m11", "m02", "m30", "m21", "m12", "m03",
              "xmin", "ymin", "xmax", "ymax")

BG_MODES = {None: BG_NONE, "none": BG_NONE, "mean": BG_MEAN, "ema": BG_EMA, "static": BG_STATIC}
MORPH_OPS = {"erode": MORPH_ERODE, "dilate": MORPH_DILATE}
SHAPES = {"rect": SHAPE_RECT, "cross": SHAPE_CROSS, "ellipse": SHAPE_ELLIPSE}


class HipError(RuntimeError):
    """a call into libvideoanalysis_hip.so failed"""

    def __init__(self, code, message):
        super().__init__("libvideoanalysis_hip: %s (code %d)" % (message, code))
        self.code = code


class HipUnavailableError(HipError):
    """the HIP extension (or a GPU) is not available -- there is no CPU fallback"""

    def __init__(self, message):
        HipError.__init__(self, -19, message)


class va_config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("channels", C.c_int32), ("dtype", C.c_int32), ("max_batch", C.c_int32),
                ("bg_mode", C.c_int32), ("bg_rate", C.c_float), ("sigma", C.c_double),
                ("thresh", C.c_int32), ("maxval", C.c_int32), ("morph_count", C.c_int32),
                ("morph_op", C.c_int32 * MAX_MORPH_OPS),
                ("morph_shape", C.c_int32 * MAX_MORPH_OPS),
                ("morph_ksize", C.c_int32 * MAX_MORPH_OPS),
                ("connectivity", C.c_int32), ("max_labels", C.c_int32), ("tap_rule", C.c_int32)]


_vp, _i, _sz, _d, _i64 = C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_int64

# name -> (restype, argtypes); mirrors include/videoanalysis_hip.h one to one
SIGNATURES = {
    "va_init": (_i, [_i]),
    "va_device_count": (_i, []),
    "va_version": (C.c_char_p, []),
    "va_last_error": (C.c_char_p, []),
    "va_trim": (_i, [_sz]),
    "va_malloc": (_i, [C.POINTER(_vp), _sz]),
    "va_free": (_i, [_vp]),
    "va_host_alloc": (_i, [C.POINTER(_vp), _sz]),
    "va_host_free": (_i, [_vp]),
    "va_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "va_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "va_memcpy_d2d": (_i, [_vp, _vp, _sz, _vp]),
    "va_memset": (_i, [_vp, _i, _sz, _vp]),
    "va_stream_sync": (_i, [_vp]),
    "va_stream_create": (_i, [C.POINTER(_vp)]),
    "va_stream_destroy": (_i, [_vp]),
    "va_event_create": (_i, [C.POINTER(_vp)]),
    "va_event_destroy": (_i, [_vp]),
    "va_event_record": (_i, [_vp, _vp]),
    "va_stream_wait_event": (_i, [_vp, _vp]),
    "va_event_sync": (_i, [_vp]),
    "va_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "va_gaussian_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _vp]),
    "va_gaussian_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _vp]),
    "va_gaussian_u8_rule": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _i, _vp]),
    "va_gauss_taps_q8": (_i, [_d, C.POINTER(_i), _vp, _i]),
    "va_gauss_taps_q8_rule": (_i, [_d, _i, C.POINTER(_i), _vp, _i]),
    "va_gauss_taps_f32": (_i, [_d, C.POINTER(_i), _vp, _i]),
    "va_bg_update": (_i, [_i, _i, _vp, _vp, _vp, _i64, _d, _i, _sz, _vp]),
    "va_welford_u8": (_i, [_vp, _vp, _vp, _i64, _i, _sz, _vp]),
    "va_mean_any": (_i, [_vp, _i, _vp, _i64, _i, _sz, _vp]),
    "va_welford_any": (_i, [_vp, _i, _vp, _vp, _i64, _i, _sz, _vp]),
    "va_time_difference_u8": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "va_threshold_u8": (_i, [_vp, _vp, _sz, _i, _i, _vp]),
    "va_mono_mean_u8": (_i, [_vp, _vp, _sz, _vp]),
    "va_rot90": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "va_normalize_u8": (_i, [_vp, _vp, _sz, _d, _d, _d, _d, _vp]),
    "va_morph_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "va_label_workspace_bytes": (_sz, [_i, _i, _i]),
    "va_label_i32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "va_moments_i64": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "va_largest_region": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "va_detect_peaks_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "va_detect_peaks_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "va_image_statistics_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _i, _vp]),
    "va_mask_thinning_u8": (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(_i), _vp]),
    "va_image_statistics_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _i, _vp]),
    "va_contour_workspace_bytes": (_sz, [_i, _i, _i]),
    "va_largest_contour": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "va_normalize": (_i, [_vp, _i, _vp, _i, _sz, _d, _d, _d, _d, _vp]),
    "va_prepare_u8": (_i, [_vp, _vp] + [_i] * 10 + [_d, _d, _d, _d, _vp]),
    "va_gaussian_noise": (_i, [_vp, _i, _sz, _d, _d, C.c_uint64, C.c_uint64, _vp]),
    "va_resize_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "va_resize_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "va_contour_moments": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "va_pipeline_create": (_i, [C.POINTER(va_config), C.POINTER(_vp)]),
    "va_pipeline_destroy": (_i, [_vp]),
    "va_pipeline_run": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "va_pipeline_overlap": (_i, [_vp, _i]),
    "va_pipeline_fence": (_i, [_vp, _vp]),
    "va_bg_get_state": (_i, [_vp, _vp, _sz, C.POINTER(_i64)]),
    "va_bg_set_state": (_i, [_vp, _vp, _sz, _i64]),
    "va_bg_state_bytes": (_sz, [_vp]),
    "va_pipeline_describe": (C.c_char_p, [_vp]),
    "va_pipeline_profile": (_i, [_vp, _i]),
    "va_pipeline_stage_times": (_i, [_vp, _i, _vp, _vp, _vp, C.POINTER(_i)]),
    "va_gaussian_u8_generic": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _vp]),
    "va_gaussian_u8_valu": (_i, [_vp, _vp, _i, _i, _i, _i, _d, _vp]),
    "va_test_hook_labelling": (_i, [_i, _i]),
    "va_test_hook_gaussian_u8": (_i, [_i]),
    "va_test_hook_gaussian_f32": (_i, [_i]),
    "va_morph_bits_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "va_comm_unique_id": (_i, [_vp]),
    "va_comm_init": (_i, [C.POINTER(_vp), _i, _i, _vp]),
    "va_gather_counts": (_i, [_vp, _vp, _vp, _i, _vp]),
    "va_comm_destroy": (_i, [_vp]),
}

_lib = None
_ready_device = None


def load_library():
    """dlopen the shared library and declare the prototypes (no GPU needed for this)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipUnavailableError(
                "%s is missing -- build it with `make -C video-analysis_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise HipUnavailableError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(code):
    if code != 0:
        raise HipError(code, load_library().va_last_error().decode("utf-8", "replace"))


def lib(device=None):
    """library handle with an initialised GPU; raises HipUnavailableError without one.

    One device per process (one process per GPU, as bench.py / torch.distributed launch them): the
    first call selects `device` (default: $VA_DEVICE, else $LOCAL_RANK, else 0); asking for a
    device that is not visible, or for a second device later, raises instead of silently running
    somewhere else."""
    global _ready_device
    L = load_library()
    if _ready_device is None:
        if device is None:
            device = int(os.environ.get("VA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        count = L.va_device_count()
        if count <= 0:
            raise HipUnavailableError("no HIP device visible: the video filters/analysis ops "
                                      "run on MI355X only (no CPU fallback)")
        if not 0 <= device < count:
            raise HipUnavailableError("device %d requested but only %d GPU(s) are visible to this "
                                      "process (check LOCAL_RANK / VA_DEVICE / HIP_VISIBLE_DEVICES)"
                                      % (device, count))
        check(L.va_init(device))
        _ready_device = device
    elif device is not None and device != _ready_device:
        raise ValueError("this process already runs on GPU %d; device %d requested (the library "
                         "uses one device per process: start one process per GPU)"
                         % (_ready_device, device))
    return L


def gpu_available():
    try:
        return load_library().va_device_count() > 0
    except HipError:
        return False


class DeviceBuffer(object):
    """a hipMalloc'ed buffer with NumPy upload/download helpers"""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self._ptr = C.c_void_p()
        check(lib().va_malloc(C.byref(self._ptr), max(self.nbytes, 1)))

    @property
    def ptr(self):
        return self._ptr.value

    @classmethod
    def from_array(cls, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        buf.upload(arr, stream)
        return buf

    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes > self.nbytes:
            raise ValueError("array of %d bytes does not fit buffer of %d" % (arr.nbytes, self.nbytes))
        L = lib()
        check(L.va_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes, stream))
        check(L.va_stream_sync(stream))     # pageable source: make the copy complete

    def download(self, shape, dtype, stream=None):
        out = np.empty(shape, dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("download of %d bytes exceeds buffer of %d" % (out.nbytes, self.nbytes))
        L = lib()
        check(L.va_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, stream))
        check(L.va_stream_sync(stream))
        return out

    def free(self):
        if self._ptr is not None and self._ptr.value:
            load_library().va_free(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def gauss_taps_q8(sigma, tap_rule="cv4"):
    """the unsigned 8.8 fixed-point taps used by FilterBlur on uint8 frames (host only); tap_rule
    'cv4' (OpenCV >= 4: error diffusion, sum 256) or 'cv3' (OpenCV 2.4 / 3.x: tap-by-tap rounding)"""
    buf = np.zeros(256, np.uint16)
    ks = C.c_int()
    check(load_library().va_gauss_taps_q8_rule(float(sigma), TAP_RULES[tap_rule], C.byref(ks), buf.ctypes.data, 256))
    return buf[:ks.value].copy()


def gauss_taps_f32(sigma):
    buf = np.zeros(256, np.float32)
    ks = C.c_int()
    check(load_library().va_gauss_taps_f32(float(sigma), C.byref(ks), buf.ctypes.data, 256))
    return buf[:ks.value].copy()
