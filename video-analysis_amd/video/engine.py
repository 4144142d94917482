"""FrameEngine -- the batched, device-resident form of the filter chain.

One engine = one `va_pipeline_t`:  FilterBackground -> FilterBlur -> FilterThreshold ->
FilterMorphology -> label/areas (get_largest_region's first half), executed for a whole batch
of frames that already sit in HBM.  `run()` takes NumPy frames (upload + download, for the
drop-in iterator API) and `run_device()` takes raw device pointers (e.g. torch tensors'
``data_ptr()``), which is what bench.py and a multi-GPU driver use.
"""
import ctypes as C

import numpy as np

from . import _hip
from ._hip import DeviceBuffer, check


class FrameEngine(object):

    def __init__(self, size, channels=1, dtype=np.uint8, max_batch=64, background=None,
                 bg_rate=0.02, sigma=0.0, thresh=None, maxval=255, morphology=(),
                 connectivity=0, max_labels=0, device=None, prepare=None, tap_rule="cv4"):
        """size = (width, height) as everywhere in the reference (video/io/base.py:40-55).
        background: None | 'mean' | 'ema' | 'static';  morphology: sequence of
        (op, shape, ksize) with op in {'erode','dilate'}, shape in {'rect','cross','ellipse'}.
        prepare: optional dict describing the pointwise pre-stages FilterCrop -> FilterMonochrome
        -> FilterNormalize, fused into one device pass in front of the chain (`run` then takes the
        RAW source frames): {'src_size': (W0, H0), 'src_channels': 1|3, 'rect': (left, top, width,
        height) | None, 'mono': None | 'mean' | 0..2, 'normalize': None | (fmin, fmax, alpha, tmin)};
        `size` is the size AFTER the crop."""
        self.width, self.height = int(size[0]), int(size[1])
        self.channels = int(channels)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.uint8, np.float32):
            raise TypeError("FrameEngine supports uint8 and float32 frames")
        self.max_batch = int(max_batch)
        self.connectivity = int(connectivity)
        self.max_labels = int(max_labels)
        self.has_mask = thresh is not None
        morphology = list(morphology)
        if len(morphology) > _hip.MAX_MORPH_OPS:
            raise ValueError("at most %d morphology steps" % _hip.MAX_MORPH_OPS)

        cfg = _hip.va_config()
        cfg.struct_size = C.sizeof(_hip.va_config)
        cfg.width, cfg.height, cfg.channels = self.width, self.height, self.channels
        cfg.dtype = _hip.VA_U8 if self.dtype == np.uint8 else _hip.VA_F32
        cfg.max_batch = self.max_batch
        cfg.bg_mode = _hip.BG_MODES[background]
        cfg.bg_rate = float(bg_rate)
        cfg.sigma = float(sigma) if sigma else 0.0
        cfg.thresh = -1 if thresh is None else int(thresh)
        cfg.maxval = int(maxval)
        cfg.morph_count = len(morphology)
        for i, (op, shape, ksize) in enumerate(morphology):
            cfg.morph_op[i] = _hip.MORPH_OPS.get(op, op)
            cfg.morph_shape[i] = _hip.SHAPES.get(shape, shape)
            cfg.morph_ksize[i] = int(ksize)
        cfg.connectivity = self.connectivity
        cfg.max_labels = self.max_labels
        cfg.tap_rule = _hip.TAP_RULES[tap_rule]        # 8-bit Gaussian taps: 'cv4' | 'cv3' (reference-era OpenCV)
        self._cfg = cfg
        self._lib = _hip.lib(device)
        self._handle = C.c_void_p()
        check(self._lib.va_pipeline_create(C.byref(cfg), C.byref(self._handle)))
        self.bg_mode = cfg.bg_mode
        self._dev = {}      # name -> DeviceBuffer cache for run()
        self.prepare = None
        if prepare is not None:
            if self.dtype != np.uint8:
                raise TypeError("prepare stages work on uint8 frames")
            w0, h0 = (int(v) for v in prepare["src_size"])
            c0 = int(prepare.get("src_channels", 1))
            rect = prepare.get("rect") or (0, 0, w0, h0)
            mono = prepare.get("mono")
            mono_code = -1 if mono is None else (3 if mono == "mean" else int(mono))
            out_c = c0 if mono_code < 0 else 1
            if (int(rect[2]), int(rect[3])) != (self.width, self.height) or out_c != self.channels:
                raise ValueError("prepare stages produce %dx%dx%d frames, the engine expects %dx%dx%d"
                                 % (rect[2], rect[3], out_c, self.width, self.height, self.channels))
            norm = prepare.get("normalize")
            self.prepare = dict(src_shape=(h0, w0) + ((c0,) if c0 > 1 else ()), h0=h0, w0=w0, c0=c0,
                                rect=tuple(int(v) for v in rect), mono=mono_code,
                                norm=None if norm is None else tuple(float(v) for v in norm))

    # ------------------------------------------------------------------ properties
    @property
    def frame_shape(self):
        return (self.height, self.width) + ((self.channels,) if self.channels > 1 else ())

    @property
    def description(self):
        return self._lib.va_pipeline_describe(self._handle).decode()

    # ------------------------------------------------------------------ device-pointer API
    def run_device(self, frames_ptr, n, filtered_ptr=None, mask_ptr=None, labels_ptr=None,
                   counts_ptr=None, stats_ptr=None, stream=None):
        """enqueue the chain for `n` frames at device address `frames_ptr` (asynchronous)"""
        check(self._lib.va_pipeline_run(self._handle, frames_ptr, int(n), filtered_ptr, mask_ptr,
                                        labels_ptr, counts_ptr, stats_ptr, stream))

    def overlap(self, enable=True):
        """run the label-image write of batch k on the engine's own stream, beside the stages of batch
        k + 1 (`run_device` only).  While on, labels / stats of a run are complete on a stream only
        after `fence(stream)`; give consecutive runs alternating label buffers for the full effect."""
        check(self._lib.va_pipeline_overlap(self._handle, 1 if enable else 0))
        self._overlap = bool(enable)

    def fence(self, stream=None):
        """make `stream` wait for every label-image write enqueued so far (asynchronous)"""
        check(self._lib.va_pipeline_fence(self._handle, stream))

    # ------------------------------------------------------------------ NumPy API
    def _buf(self, name, nbytes):
        b = self._dev.get(name)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = self._dev[name] = DeviceBuffer(nbytes)
        return b

    def run(self, frames, want=("mask", "labels", "counts")):
        """frames: (n, H, W[, C]) array; returns a dict with the requested outputs among
        'filtered', 'mask', 'labels', 'counts', 'stats'"""
        arr = np.ascontiguousarray(frames, self.dtype)
        in_shape = self.prepare["src_shape"] if self.prepare else self.frame_shape
        if arr.shape[1:] != in_shape:
            raise ValueError("frames of shape %r do not match %r" % (arr.shape[1:], in_shape))
        n = arr.shape[0]
        if n > self.max_batch:
            raise ValueError("batch of %d exceeds max_batch=%d" % (n, self.max_batch))
        want = set(want)
        unknown = want - {"filtered", "mask", "labels", "counts", "stats"}
        if unknown:
            raise ValueError("unknown outputs %r" % sorted(unknown))
        px = self.width * self.height
        if self.prepare:                    # crop / monochrome / normalize: one device pass, no host round trip
            pr = self.prepare
            raw = self._buf("raw", arr.nbytes)
            raw.upload(arr)
            src = self._buf("src", n * px * self.channels)
            nm = pr["norm"] or (0.0, 0.0, 0.0, 0.0)
            check(self._lib.va_prepare_u8(raw.ptr, src.ptr, n, pr["h0"], pr["w0"], pr["c0"], pr["rect"][0],
                                          pr["rect"][1], pr["rect"][2], pr["rect"][3], pr["mono"],
                                          1 if pr["norm"] else 0, nm[0], nm[1], nm[2], nm[3], None))
            arr = np.empty((n,) + self.frame_shape, self.dtype)     # (shape of the prepared frames)
        else:
            src = self._buf("src", arr.nbytes)
            src.upload(arr)
        ptr = {}
        if "filtered" in want:
            ptr["filtered"] = self._buf("filtered", arr.nbytes)
        if "mask" in want:
            ptr["mask"] = self._buf("mask", n * px)
        if "labels" in want:
            ptr["labels"] = self._buf("labels", n * px * 4)
        if "counts" in want or "stats" in want:
            ptr["counts"] = self._buf("counts", n * 4)
        if "stats" in want:
            ptr["stats"] = self._buf("stats", n * max(self.max_labels, 1) * _hip.STATS_STRIDE * 8)
        g = lambda k: ptr[k].ptr if k in ptr else None
        self.run_device(src.ptr, n, g("filtered"), g("mask"), g("labels"), g("counts"), g("stats"))
        if getattr(self, "_overlap", False):
            self.fence(None)                # the downloads below read what the side stream writes
        out = {}
        if "filtered" in want:
            out["filtered"] = ptr["filtered"].download(arr.shape, self.dtype)
        if "mask" in want:
            out["mask"] = ptr["mask"].download((n, self.height, self.width), np.uint8)
        if "labels" in want:
            out["labels"] = ptr["labels"].download((n, self.height, self.width), np.int32)
        if "counts" in ptr:
            out["counts"] = ptr["counts"].download((n,), np.int32)
        if "stats" in want:
            out["stats"] = ptr["stats"].download((n, self.max_labels, _hip.STATS_STRIDE), np.int64)
        return out

    # ------------------------------------------------------------------ per-stage timing
    def profile(self, enable=True, every=1):
        """start (and reset) / stop recording one HIP event per stage on the run's stream; `every` > 1
        records every `every`-th run only (the events themselves cost about 2 % of the chain)"""
        check(self._lib.va_pipeline_profile(self._handle, (max(int(every), 1) if enable else 0)))

    def stage_times(self):
        """{stage: (total_ms, launches)} accumulated since profile(True)"""
        cap = 32
        names = C.create_string_buffer(cap * 32)
        ms = (C.c_double * cap)()
        launches = (C.c_int32 * cap)()
        ns = C.c_int()
        check(self._lib.va_pipeline_stage_times(self._handle, cap, names, ms, launches, C.byref(ns)))
        out = {}
        for k in range(ns.value):
            nm = names.raw[k * 32:(k + 1) * 32].split(b"\0", 1)[0].decode()
            out[nm] = (ms[k], launches[k])
        return out

    # ------------------------------------------------------------------ background state
    def get_background(self):
        """(state array, n_seen) -- float64 for 'mean'/'static', float32 for 'ema'"""
        nbytes = self._lib.va_bg_state_bytes(self._handle)
        n_seen = C.c_int64()
        if nbytes == 0:
            return None, 0
        dt = np.float32 if self.bg_mode == _hip.BG_EMA else np.float64
        state = np.empty(self.frame_shape, dt)
        check(self._lib.va_bg_get_state(self._handle, state.ctypes.data, state.nbytes,
                                        C.byref(n_seen)))
        return state, n_seen.value

    def set_background(self, state, n_seen=0):
        dt = np.float32 if self.bg_mode == _hip.BG_EMA else np.float64
        if state is None:
            check(self._lib.va_bg_set_state(self._handle, None, 0, int(n_seen)))
            return
        st = np.ascontiguousarray(state, dt).reshape(self.frame_shape)
        check(self._lib.va_bg_set_state(self._handle, st.ctypes.data, st.nbytes, int(n_seen)))

    def close(self):
        if self._handle is not None and self._handle.value:
            self._lib.va_pipeline_destroy(self._handle)
            self._handle = C.c_void_p()
        for b in self._dev.values():
            b.free()
        self._dev = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
