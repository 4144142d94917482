"""NumPy-in / NumPy-out wrappers over the C ABI (one upload, kernels, one download).

These are the per-call building blocks behind ``video.filters`` and ``video.analysis``; the
batched, device-resident path is :class:`video.engine.FrameEngine`.  Everything here runs on
the GPU through ``libvideoanalysis_hip.so`` -- nothing is computed with NumPy.
"""
import ctypes as C

import threading

import numpy as np

from . import _hip
from ._hip import DeviceBuffer, check

# ------------------------------------------------------------------------ device-buffer pool
# The per-call wrappers below run once per frame when filters are used one by one; a hipMalloc /
# hipFree pair per operand and call (both synchronise the device) used to dominate them.  Buffers
# are recycled by size class (powers of two) instead; the pool is bounded and trimmed on overflow.
_POOL = {}
_POOL_BYTES = 0
_POOL_LOCK = threading.Lock()       # VideoPreprocessor runs these ops from one worker thread per function
POOL_CAPACITY = 4 << 30


def _take(nbytes):
    """a device buffer of at least `nbytes` bytes (recycled when one of its size class is free)"""
    global _POOL_BYTES
    size = max(256, 1 << max(int(nbytes) - 1, 1).bit_length())
    with _POOL_LOCK:
        free = _POOL.get(size)
        if free:
            _POOL_BYTES -= size
            return free.pop()
    return DeviceBuffer(size)


def _upload(arr, stream=None):
    arr = np.ascontiguousarray(arr)
    buf = _take(arr.nbytes)
    buf.upload(arr, stream)
    return buf


def _give(*bufs):
    """hand buffers back (every call below has synchronised its stream by then)"""
    global _POOL_BYTES
    for b in bufs:
        if b is None:
            continue
        with _POOL_LOCK:
            keep = _POOL_BYTES + b.nbytes <= POOL_CAPACITY
            if keep:
                _POOL.setdefault(b.nbytes, []).append(b)
                _POOL_BYTES += b.nbytes
        if not keep:
            b.free()


def pool_clear():
    """free every pooled buffer (tests; before handing the GPU to another library) and let the
    library's own scratch pool go back to the device (va_trim)"""
    global _POOL_BYTES
    with _POOL_LOCK:
        bufs = [b for free in _POOL.values() for b in free]
        _POOL.clear()
        _POOL_BYTES = 0
    for b in bufs:
        b.free()
    try:
        _hip.load_library().va_trim(0)
    except Exception:
        pass



def _as_batch(arr, frame_ndim):
    """returns (contiguous array, n, frame_shape, was_single)"""
    arr = np.ascontiguousarray(arr)
    if arr.ndim == frame_ndim:
        return arr, 1, arr.shape, True
    if arr.ndim == frame_ndim + 1:
        return arr, arr.shape[0], arr.shape[1:], False
    raise ValueError("expected %d or %d dimensions, got shape %r"
                     % (frame_ndim, frame_ndim + 1, arr.shape))


def _hwc(frame_shape):
    if len(frame_shape) == 2:
        return frame_shape[0], frame_shape[1], 1
    if len(frame_shape) == 3:
        return frame_shape
    raise ValueError("frames must be (H,W) or (H,W,C), got %r" % (frame_shape,))


def gaussian_blur(frames, sigma, color=False, implementation=None, tap_rule="cv4"):
    """cv2.GaussianBlur(frame, (0,0), sigma) on uint8 or float32 frames
    (FilterBlur._process_frame, video/filters.py:388-392).

    frames: (H,W), (N,H,W); with color=True (H,W,C), (N,H,W,C).
    implementation: None (library's choice) or 'generic' (uint8 only, for cross-checks).
    tap_rule (uint8 only): 'cv4' | 'cv3', see FilterBlur.
    """
    frames = np.asarray(frames)
    if frames.dtype not in (np.uint8, np.float32):
        raise TypeError("gaussian_blur supports uint8 and float32, got %s" % frames.dtype)
    arr, n, fshape, single = _as_batch(frames, 3 if color else 2)
    h, w, c = _hwc(fshape)
    L = _hip.lib()
    src = _upload(arr)
    dst = _take(arr.nbytes)
    if arr.dtype == np.uint8:
        if tap_rule != "cv4":
            if implementation is not None:
                raise ValueError("the implementation hooks run the default tap rule only")
            fn = lambda *a: L.va_gaussian_u8_rule(*(a[:7] + (_hip.TAP_RULES[tap_rule],) + a[7:]))
        else:
            fn = {None: L.va_gaussian_u8, "generic": L.va_gaussian_u8_generic,
                  "valu": L.va_gaussian_u8_valu}[implementation]
    else:
        fn = L.va_gaussian_f32
    check(fn(src.ptr, dst.ptr, n, h, w, c, float(sigma), None))
    out = dst.download(arr.shape, arr.dtype)
    _give(src)
    _give(dst)
    return out


class BackgroundModel(object):
    """device-resident background state (BUILD-DEFINED FilterBackground; cumulative mean =
    measure_mean's arithmetic, video/analysis/video.py:33)."""

    def __init__(self, frame_shape, mode="mean", rate=0.02, dtype=np.uint8, background=None):
        self.mode = _hip.BG_MODES[mode]
        if self.mode == _hip.BG_NONE:
            raise ValueError("mode must be 'mean', 'ema' or 'static'")
        self.frame_shape = tuple(frame_shape)
        self.px = int(np.prod(self.frame_shape))
        self.rate = float(rate)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.uint8, np.float32):
            raise TypeError("background model supports uint8 and float32 frames")
        if self.dtype == np.float32 and self.mode != _hip.BG_EMA:
            raise ValueError("float32 frames support mode='ema' only")
        self.state_dtype = np.float32 if self.mode == _hip.BG_EMA else np.float64
        self.n_seen = 0
        init = np.zeros(self.frame_shape, self.state_dtype)
        if background is not None:
            init = np.ascontiguousarray(background, self.state_dtype).reshape(self.frame_shape)
        elif self.mode == _hip.BG_STATIC:
            raise ValueError("mode='static' needs a background image")
        self._state = _upload(init)

    def process(self, frames, want_diff=True):
        """fold `frames` (N, *frame_shape) in, return |frame - bg_prev| per frame"""
        arr = np.ascontiguousarray(frames, self.dtype)
        if arr.shape[1:] != self.frame_shape:
            raise ValueError("frames of shape %r do not match %r" % (arr.shape[1:], self.frame_shape))
        n = arr.shape[0]
        src = _upload(arr)
        dst = _take(arr.nbytes) if want_diff else None
        check(_hip.lib().va_bg_update(self.mode, _hip.VA_U8 if self.dtype == np.uint8 else _hip.VA_F32,
                                      src.ptr, dst.ptr if dst else None, self._state.ptr,
                                      self.n_seen, self.rate, n, self.px, None))
        if self.mode != _hip.BG_STATIC:
            self.n_seen += n
        out = dst.download(arr.shape, arr.dtype) if dst else None
        _give(src)
        if dst:
            _give(dst)
        return out

    @property
    def state(self):
        return self._state.download(self.frame_shape, self.state_dtype)

    def set_state(self, state, n_seen):
        self._state.upload(np.ascontiguousarray(state, self.state_dtype).reshape(self.frame_shape))
        self.n_seen = int(n_seen)


TEMPORAL_DTYPES = {np.dtype(np.uint8): _hip.VA_U8, np.dtype(np.int16): _hip.VA_I16,
                   np.dtype(np.float32): _hip.VA_F32}


def _temporal_frames(frames):
    arr = np.ascontiguousarray(frames)
    if arr.dtype not in TEMPORAL_DTYPES:
        raise TypeError("temporal statistics take uint8, int16 or float32 frames on the GPU path, got %s" % arr.dtype)
    return arr


def welford(frames, mean=None, m2=None, n_seen=0):
    """Welford update of measure_mean_std (video/analysis/video.py:48-50); returns (mean, M2).
    uint8, int16 (FilterTimeDifference) or float32 frames."""
    arr = _temporal_frames(frames)
    fshape = arr.shape[1:]
    px = int(np.prod(fshape))
    mean = np.zeros(fshape) if mean is None else np.ascontiguousarray(mean, np.float64)
    m2 = np.zeros(fshape) if m2 is None else np.ascontiguousarray(m2, np.float64)
    src = _upload(arr)
    dm = _upload(mean)
    dq = _upload(m2)
    if arr.dtype == np.uint8:
        check(_hip.lib().va_welford_u8(src.ptr, dm.ptr, dq.ptr, int(n_seen), arr.shape[0], px, None))
    else:
        check(_hip.lib().va_welford_any(src.ptr, TEMPORAL_DTYPES[arr.dtype], dm.ptr, dq.ptr, int(n_seen),
                                        arr.shape[0], px, None))
    out = dm.download(fshape, np.float64), dq.download(fshape, np.float64)
    for b in (src, dm, dq):
        _give(b)
    return out


def running_mean(frames, mean=None, n_seen=0):
    """measure_mean's update `mean*n/(n+1) + frame/(n+1)` (video/analysis/video.py:33) over a batch of
    uint8 / int16 / float32 frames, NumPy's promotions included; returns the float64 mean"""
    arr = _temporal_frames(frames)
    fshape = arr.shape[1:]
    px = int(np.prod(fshape))
    mean = np.zeros(fshape) if mean is None else np.ascontiguousarray(mean, np.float64)
    src = _upload(arr)
    dm = _upload(mean)
    check(_hip.lib().va_mean_any(src.ptr, TEMPORAL_DTYPES[arr.dtype], dm.ptr, int(n_seen), arr.shape[0], px, None))
    out = dm.download(fshape, np.float64)
    _give(src, dm)
    return out


def time_difference(this_frame, prev_frame):
    """this.astype(int16) - prev  (FilterTimeDifference, video/filters.py:564-568)"""
    a = np.ascontiguousarray(this_frame, np.uint8)
    b = np.ascontiguousarray(prev_frame, np.uint8)
    if a.shape != b.shape:
        raise ValueError("frame shapes differ")
    da, db = _upload(a), _upload(b)
    do = _take(a.size * 2)
    check(_hip.lib().va_time_difference_u8(da.ptr, db.ptr, do.ptr, a.size, None))
    out = do.download(a.shape, np.int16)
    for x in (da, db, do):
        _give(x)
    return out


def _pointwise_u8(fn, arr, out_shape, *args):
    src = _upload(arr)
    dst = _take(int(np.prod(out_shape)))
    check(fn(src.ptr, dst.ptr, *args))
    out = dst.download(out_shape, np.uint8)
    _give(src)
    _give(dst)
    return out


def threshold(frames, thresh, maxval=255):
    """BUILD-DEFINED FilterThreshold: frames > thresh ? maxval : 0"""
    a = np.ascontiguousarray(frames, np.uint8)
    return _pointwise_u8(_hip.lib().va_threshold_u8, a, a.shape, a.size, int(thresh), int(maxval), None)


def mono_mean(frames):
    """np.mean(frame, axis=2).astype(uint8)  (FilterMonochrome, video/filters.py:365-366)"""
    a = np.ascontiguousarray(frames, np.uint8)
    if a.shape[-1] != 3:
        raise ValueError("last dimension must be 3")
    return _pointwise_u8(_hip.lib().va_mono_mean_u8, a, a.shape[:-1], a.size // 3, None)


def rot90(frames, k=1, color=False):
    """np.rot90(frame, k) per frame (FilterRotate, video/filters.py:339-344).
    frames: (H,W) / (N,H,W), with color=True (H,W,C) / (N,H,W,C); any dtype whose pixel
    (channels x itemsize) is 1, 2, 3, 4, 6, 8 or 12 bytes."""
    arr, n, fshape, single = _as_batch(np.ascontiguousarray(frames), 3 if color else 2)
    h, w, c = _hwc(fshape)
    k = int(k) % 4
    out_shape = ((w, h) if k & 1 else (h, w)) + ((c,) if color else ())
    src = _upload(arr)
    dst = _take(arr.nbytes)
    check(_hip.lib().va_rot90(src.ptr, dst.ptr, n, h, w, c * arr.dtype.itemsize, k, None))
    out = dst.download((n,) + out_shape, arr.dtype)
    _give(src)
    _give(dst)
    return out[0] if single else out


def normalize(frames, fmin, fmax, alpha, tmin):
    """clip + affine + astype(uint8)  (FilterNormalize, video/filters.py:126-132)"""
    a = np.ascontiguousarray(frames, np.uint8)
    return _pointwise_u8(_hip.lib().va_normalize_u8, a, a.shape, a.size, float(fmin), float(fmax),
                         float(alpha), float(tmin), None)


INTERPOLATIONS = {"nearest": 0, "linear": 1, "cubic": 2, "area": 3, "lanczos": 4}


def resize(frames, size, interpolation="linear", color=False):
    """cv2.resize(frame, size, interpolation=...) per frame (FilterResize, video/filters.py:310-314);
    size = (width, height); uint8 or float32 frames (H,W) / (N,H,W), with color=True (H,W,C) / (N,H,W,C)"""
    frames = np.asarray(frames)
    if frames.dtype not in (np.uint8, np.float32):
        raise TypeError("resize supports uint8 and float32 frames on the GPU path, got %s" % frames.dtype)
    if interpolation not in INTERPOLATIONS:
        raise ValueError("Unknown interpolation method: %s" % (interpolation,))
    arr, n, fshape, single = _as_batch(frames, 3 if color else 2)
    h, w, c = _hwc(fshape)
    dw, dh = int(size[0]), int(size[1])
    if dw < 1 or dh < 1:
        raise ValueError("target size must be positive, got %r" % (size,))
    out_shape = (n, dh, dw) + ((c,) if color else ())
    src = _upload(arr)
    dst = _take(int(np.prod(out_shape)) * arr.dtype.itemsize)
    try:
        fn = _hip.lib().va_resize_u8 if arr.dtype == np.uint8 else _hip.lib().va_resize_f32
        check(fn(src.ptr, dst.ptr, n, h, w, c, dh, dw, INTERPOLATIONS[interpolation], None))
        out = dst.download(out_shape, arr.dtype)
    finally:
        _give(src, dst)
    return out[0] if single else out


_DTYPE_CODES = {np.dtype(np.uint8): _hip.VA_U8, np.dtype(np.float32): _hip.VA_F32,
                np.dtype(np.float64): _hip.VA_F64}


def normalize_any(frames, fmin, fmax, alpha, tmin, dtype):
    """FilterNormalize for uint8 / float32 frames and uint8 / float32 / float64 targets
    (video/filters.py:126-132): clip, (f - fmin)*alpha + tmin in float64, astype(dtype)"""
    a = np.ascontiguousarray(frames)
    dtype = np.dtype(dtype)
    if a.dtype not in (np.uint8, np.float32) or dtype not in _DTYPE_CODES:
        raise TypeError("normalize: %s -> %s is not supported on the GPU path" % (a.dtype, dtype))
    src = _upload(a)
    dst = _take(a.size * dtype.itemsize)
    try:
        check(_hip.lib().va_normalize(src.ptr, _DTYPE_CODES[a.dtype], dst.ptr, _DTYPE_CODES[dtype], a.size,
                                      float(fmin), float(fmax), float(alpha), float(tmin), None))
        return dst.download(a.shape, dtype)
    finally:
        _give(src, dst)


def gaussian_noise(shape, dtype=np.float64, mean=0.0, std=1.0, seed=0, first_index=0):
    """`mean + std*randn(*shape)` produced on the GPU (VideoGaussianNoise, video/io/computed.py:36-41):
    sample i of the seeded stream is a function of (seed, first_index + i) only"""
    dtype = np.dtype(dtype)
    if dtype not in _DTYPE_CODES:
        raise TypeError("gaussian_noise: dtype %s is not supported on the GPU path" % dtype)
    count = int(np.prod(shape))
    dst = _take(max(count, 1) * dtype.itemsize)
    try:
        check(_hip.lib().va_gaussian_noise(dst.ptr, _DTYPE_CODES[dtype], count, float(mean), float(std),
                                           int(seed) & (2 ** 64 - 1), int(first_index), None))
        return dst.download(tuple(shape), dtype)
    finally:
        _give(dst)


def morph(frames, op, shape="rect", ksize=3, implementation=None):
    """cv2.erode / cv2.dilate (video/analysis/image.py:248-251) on (H,W) or (N,H,W) uint8.
    implementation='bits' runs the bit-packed kernel of the pipeline (binary masks only)."""
    arr, n, fshape, _ = _as_batch(np.asarray(frames, np.uint8), 2)
    h, w = fshape
    L = _hip.lib()
    fn = L.va_morph_bits_u8 if implementation == "bits" else L.va_morph_u8
    return _pointwise_u8(fn, arr, arr.shape, n, h, w, _hip.MORPH_OPS.get(op, op),
                         _hip.SHAPES.get(shape, shape), int(ksize), None)


def label(masks, connectivity=4):
    """ndimage.measurements.label (video/analysis/regions.py:162) for (H,W) or (N,H,W) masks.
    returns (labels int32, counts): counts is an int for a single mask, else an int32 array"""
    m = np.asarray(masks)
    if m.dtype != np.uint8:
        m = (m != 0).astype(np.uint8)
    arr, n, fshape, single = _as_batch(m, 2)
    h, w = fshape
    L = _hip.lib()
    src = _upload(arr)
    lab = _take(arr.size * 4)
    cnt = _take(max(n, 1) * 4)
    ws_bytes = L.va_label_workspace_bytes(n, h, w)
    ws = _take(ws_bytes)
    check(L.va_label_i32(src.ptr, lab.ptr, cnt.ptr, n, h, w, int(connectivity), ws.ptr, ws_bytes, None))
    labels = lab.download(arr.shape, np.int32)
    counts = cnt.download((n,), np.int32)
    for b in (src, lab, cnt, ws):
        _give(b)
    if single:
        return labels, int(counts[0])
    return labels, counts


def region_stats(labels, max_labels):
    """per-label area / raw moments / bbox: (N?, max_labels, 16) int64, see _hip.STAT_NAMES"""
    arr, n, fshape, single = _as_batch(np.asarray(labels, np.int32), 2)
    h, w = fshape
    max_labels = max(int(max_labels), 1)
    src = _upload(arr)
    st = _take(n * max_labels * _hip.STATS_STRIDE * 8)
    check(_hip.lib().va_moments_i64(src.ptr, n, h, w, max_labels, st.ptr, None))
    out = st.download((n, max_labels, _hip.STATS_STRIDE), np.int64)
    _give(src)
    _give(st)
    return out[0] if single else out


def largest_region(mask, connectivity=4):
    """label + areas + first-max argmax + select, all on the GPU.
    returns (mask of the largest region as bool, its area, number of regions)"""
    m = np.asarray(mask)
    if m.ndim != 2:
        raise ValueError("mask must be 2-d")
    if m.dtype != np.uint8:
        m = (m != 0).astype(np.uint8)
    m = np.ascontiguousarray(m)
    h, w = m.shape
    L = _hip.lib()
    src = _upload(m)
    lab = _take(m.size * 4)
    cnt = _take(4)
    ws_bytes = L.va_label_workspace_bytes(1, h, w)
    ws = _take(ws_bytes)
    check(L.va_label_i32(src.ptr, lab.ptr, cnt.ptr, 1, h, w, int(connectivity), ws.ptr, ws_bytes, None))
    count = int(cnt.download((1,), np.int32)[0])
    bufs = [src, lab, cnt, ws]
    try:
        if count == 0:
            return np.zeros(m.shape, bool), 0, 0
        st = _take(count * _hip.STATS_STRIDE * 8)
        big = _take(4)
        area = _take(8)
        sel = _take(m.size)
        bufs += [st, big, area, sel]
        check(L.va_moments_i64(lab.ptr, 1, h, w, count, st.ptr, None))
        check(L.va_largest_region(lab.ptr, cnt.ptr, st.ptr, 1, h, w, count, big.ptr, area.ptr,
                                  sel.ptr, None))
        out = sel.download(m.shape, np.uint8).astype(bool)
        return out, int(area.download((1,), np.int64)[0]), count
    finally:
        for b in bufs:
            _give(b)


def largest_contour(mask, max_points=None, moments=False):
    """outer contour (cv2 RETR_EXTERNAL / CHAIN_APPROX_SIMPLE) of the component with the largest
    contour area.  returns (points (N,2) int32, area float, number of components); with
    moments=True also the ten spatial cv2.moments(contour) values, computed from the points while
    they are still on the device"""
    m = np.asarray(mask)
    if m.ndim != 2:
        raise ValueError("mask must be 2-d")
    if m.dtype != np.uint8:
        m = (m != 0).astype(np.uint8)
    m = np.ascontiguousarray(m)
    h, w = m.shape
    L = _hip.lib()
    cap = int(max_points) if max_points else 4096
    src = _upload(m)
    ws_bytes = L.va_contour_workspace_bytes(1, h, w)
    ws = _take(ws_bytes)
    npts, area, ncomp = _take(4), _take(8), _take(4)
    bufs = [src, ws, npts, area, ncomp]
    try:
        while True:
            pts = _take(cap * 8)
            bufs.append(pts)
            check(L.va_largest_contour(src.ptr, 1, h, w, pts.ptr, cap, npts.ptr, area.ptr, ncomp.ptr,
                                       ws.ptr, ws_bytes, None))
            n = int(npts.download((1,), np.int32)[0])
            if n <= cap or max_points:
                break
            cap = n                       # rare: a very long contour, run again with room for it
        count = int(ncomp.download((1,), np.int32)[0])
        points = pts.download((min(n, cap), 2), np.int32)
        res = (points, float(area.download((1,), np.float64)[0]), count)
        if moments:
            mom = _take(10 * 8)
            bufs.append(mom)
            check(L.va_contour_moments(pts.ptr, npts.ptr, 1, cap, 0, mom.ptr, None))
            res += (mom.download((10,), np.float64),)
        return res
    finally:
        for b in bufs:
            _give(b)


def contour_moments(contour):
    """the ten spatial moments of cv2.moments(contour) as a float64 array (m00 m10 m01 m20 m11 m02
    m30 m21 m12 m03) -- regionprops(contour=...), video/analysis/image.py:355; Polygon.moments,
    video/analysis/shapes.py:533.  Integer arrays are int32 points, everything else float32
    points (the two forms cv2.moments accepts; other dtypes it would read as an image)."""
    c = np.asarray(contour)
    if c.size == 0 or c.size % 2:
        raise ValueError("contour must hold (x, y) points")
    is_float = 0 if np.issubdtype(c.dtype, np.integer) else 1
    c = np.ascontiguousarray(c.reshape(-1, 2), np.float32 if is_float else np.int32)
    pts = _upload(c)
    out = _take(10 * 8)
    try:
        check(_hip.lib().va_contour_moments(pts.ptr, None, 1, len(c), is_float, out.ptr, None))
        return out.download((10,), np.float64)
    finally:
        _give(pts)
        _give(out)


def detect_peaks(img, include_plateaus=True):
    """boolean mask of the local maxima (video/analysis/image.py:267-306), uint8 or float32 images"""
    a = np.ascontiguousarray(img)
    if a.dtype not in (np.uint8, np.float32) or a.ndim != 2:
        raise TypeError("detect_peaks expects a 2-d uint8 or float32 image on the GPU path")
    if a.dtype == np.uint8:
        out = _pointwise_u8(_hip.lib().va_detect_peaks_u8, a, a.shape, 1, a.shape[0], a.shape[1],
                            1 if include_plateaus else 0, None)
        return out.astype(bool)
    src, dst = _upload(a), _take(a.size)
    try:
        check(_hip.lib().va_detect_peaks_f32(src.ptr, dst.ptr, 1, a.shape[0], a.shape[1],
                                             1 if include_plateaus else 0, None))
        return dst.download(a.shape, np.uint8).astype(bool)
    finally:
        _give(src, dst)


def mask_thinning(img):
    """skeleton by iterated 3x3-cross erosion/dilation (python method of mask_thinning,
    video/analysis/image.py:243-258); returns (skeleton uint8, iterations)"""
    a = np.ascontiguousarray(img, np.uint8)
    if a.ndim != 2:
        raise ValueError("mask must be 2-d")
    h, w = a.shape
    cur, tmp, skel = _upload(a), _take(a.size), _take(a.size)
    it = C.c_int()
    try:
        check(_hip.lib().va_mask_thinning_u8(cur.ptr, tmp.ptr, skel.ptr, h, w, C.byref(it), None))
        return skel.download(a.shape, np.uint8), it.value
    finally:
        for b in (cur, tmp, skel):
            _give(b)


def image_statistics(img, kernel="box", ksize=5, prior=0.0, exclude_center=False, ret_var=True):
    """local mean (and variance) in a window around every pixel
    (get_image_statistics, video/analysis/image.py:131-201), uint8 or float32 images (float values are
    truncated to integers first, as the reference's `img.astype(np.int)` does)"""
    a = np.ascontiguousarray(img)
    if a.dtype not in (np.uint8, np.float32) or a.ndim != 2:
        raise TypeError("image_statistics expects a 2-d uint8 or float32 image on the GPU path")
    h, w = a.shape
    src = _upload(a)
    dm = _take(a.size * 8)
    dv = _take(a.size * 8) if ret_var else None
    try:
        fn = _hip.lib().va_image_statistics_u8 if a.dtype == np.uint8 else _hip.lib().va_image_statistics_f32
        check(fn(src.ptr, dm.ptr, dv.ptr if dv else None, 1, h, w,
                                                {"box": 0, "ellipse": 1, "circle": 1}[kernel],
                                                int(ksize), float(prior), 1 if exclude_center else 0,
                                                None))
        mean = dm.download(a.shape, np.float64)
        return (mean, dv.download(a.shape, np.float64)) if ret_var else mean
    finally:
        for b in (src, dm, dv):
            if b:
                _give(b)
