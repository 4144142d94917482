"""video.filters -- lazy per-frame filters whose pixel work runs on the MI355X.

API surface of the reference's video/filters.py (FilterFunction :56, FilterNormalize :76,
FilterCrop :158, FilterResize :252, FilterMonochrome :348, FilterBlur :378, FilterDiffBase :492,
FilterTimeDifference :542) plus the BUILD-DEFINED classes the north star names but the
reference does not contain: FilterBackground, FilterThreshold, FilterMorphology,
and FilterAnalysisChain, the batched fused form of the whole chain.

Every filter is a `VideoFilterBase`: consumers iterate it or index it, `_process_frame` does
the work and notifies listeners.  The compute goes through `video.ops` / `video.engine`, i.e.
through the C ABI of libvideoanalysis_hip.so; there is no NumPy fallback.
"""
import logging

import weakref

import numpy as np

from . import ops
from .analysis.regions import rect_to_slices
from .engine import FrameEngine
from .io.base import VideoBase, VideoFilterBase

logger = logging.getLogger("video")

# translation dictionary for color channels (reference: video/filters.py:32-34)
COLOR_CHANNELS = {"blue": 0, "b": 0, 0: 0,
                  "green": 1, "g": 1, 1: 1,
                  "red": 2, "r": 2, 2: 2}


def get_color_range(dtype):
    """range a colour value of `dtype` can take: integer limits, or (0, 1) for floats
    (reference: video/filters.py:38-50)"""
    if np.issubdtype(dtype, np.integer):
        info = np.iinfo(dtype)
        return info.min, info.max
    if np.issubdtype(dtype, np.floating):
        return 0, 1
    raise ValueError("Unsupported data type `%r`" % dtype)


class FilterFunction(VideoFilterBase):
    """applies an arbitrary host callable to every frame (reference :56-72)"""

    def __init__(self, source, function):
        self._function = function
        super(FilterFunction, self).__init__(source)

    def _process_frame(self, frame):
        return super(FilterFunction, self)._process_frame(self._function(frame))


class FilterNormalize(VideoFilterBase):
    """maps the colour interval [vmin, vmax] onto the full range of `dtype`
    (reference :76-135): clip, (f - fmin)*alpha + tmin, astype.  uint8 and float32 frames,
    uint8 / float32 / float64 targets, on the GPU.  (The reference clips the SOURCE frame in
    place, `np.clip(..., out=frame)`; the source's frames are left untouched here.)"""

    def __init__(self, source, vmin=None, vmax=None, dtype=None):
        self._fmin, self._fmax = vmin, vmax
        self._dtype = None if dtype is None else np.dtype(dtype)
        self._tmin = self._alpha = None
        super(FilterNormalize, self).__init__(source)

    def _process_frame(self, frame):
        frame = np.asarray(frame)
        if frame.dtype not in (np.uint8, np.float32):
            raise TypeError("FilterNormalize: uint8 and float32 frames are supported on the GPU path")
        if self._dtype is None:                 # learnt lazily from the first frame (:104-109)
            self._dtype = frame.dtype
        if self._dtype not in (np.uint8, np.float32, np.float64):
            raise TypeError("FilterNormalize: targets are uint8, float32 or float64")
        if self._fmin is None:
            self._fmin = frame.min()
        if self._fmax is None:
            self._fmax = frame.max()
        if self._tmin is None:
            self._tmin, tmax = get_color_range(self._dtype)
            self._alpha = (tmax - self._tmin) / (float(self._fmax) - float(self._fmin))
        if frame.dtype == np.uint8 and self._dtype == np.uint8:
            out = ops.normalize(frame, self._fmin, self._fmax, self._alpha, self._tmin)
        else:
            out = ops.normalize_any(frame, self._fmin, self._fmax, self._alpha, self._tmin, self._dtype)
        return super(FilterNormalize, self)._process_frame(out)


def _check_coordinate(value, max_value):
    """fractions in (-1, 1) scale with max_value, negatives count from the far edge, result
    must lie in [0, max_value) (reference :139-154)"""
    if -1 < value < 1:
        value = int(value * max_value)
    if value < 0:
        value += max_value
    if not 0 <= value < max_value:
        raise IndexError("Coordinate %d is out of bounds [0, %d]." % (value, max_value))
    return value


class FilterCrop(VideoFilterBase):
    """crops to rect=(left, top, width, height) or to a named region ('lower', 'upper',
    'left', 'right' and combinations); optional colour-channel pick; consecutive crops
    contract into one (reference :158-248).  A view -- no pixel work."""

    def __init__(self, source, rect=None, region="", color_channel=None, size_alignment=1):
        source_width, source_height = source.size
        if rect is not None:
            left = _check_coordinate(rect[0], source_width)
            top = _check_coordinate(rect[1], source_height)
            width = _check_coordinate(rect[2], source_width)
            height = _check_coordinate(rect[3], source_height)
        else:
            region = region.lower()
            left, top, width, height = 0, 0, source_width, source_height
            if "left" in region:
                width //= 2
            elif "right" in region:
                width //= 2
                left = source_width - width
            if "upper" in region:
                height //= 2
            elif "lower" in region:
                height //= 2
                top = source_height - height
        while isinstance(source, FilterCrop):      # contract with parent crops
            left += source.rect[0]
            top += source.rect[1]
            if source.color_channel is not None:
                color_channel = source.color_channel
            source = source._source
        self.color_channel = COLOR_CHANNELS.get(color_channel, color_channel)
        is_color = None if color_channel is None else False
        if size_alignment != 1:
            width = int(round(width / size_alignment) * size_alignment)
            height = int(round(height / size_alignment) * size_alignment)
        self.rect = (left, top, width, height)
        self.slices = rect_to_slices(self.rect)
        super(FilterCrop, self).__init__(source, size=self.rect[2:], is_color=is_color)

    def _process_frame(self, frame):
        if self.color_channel is None:
            frame = frame[self.slices]
        else:
            frame = frame[self.slices[0], self.slices[1], self.color_channel]
        return super(FilterCrop, self)._process_frame(frame)


class FilterResize(VideoFilterBase):
    """resizes the video to `size` = (width, height), or by a factor if `size` is a number
    (reference :252-315: cv2.resize).  `interpolation`: 'auto' (area when the frame shrinks, cubic
    when it grows), 'nearest', 'linear', 'area', 'cubic', 'lanczos'.
    Consecutive resizes contract into one (:299-301).  uint8 and float32 frames."""

    def __init__(self, source, size=None, interpolation="auto", even_dimensions=False):
        if hasattr(size, "__iter__"):
            width, height = size
        else:
            width = int(source.size[0] * size)
            height = int(source.size[1] * size)
        if even_dimensions:
            width += width % 2
            height += height % 2
        if (width, height) == tuple(source.size):
            self.interpolation = None
        elif interpolation == "auto":
            shrinks = width * height < source.size[0] * source.size[1]
            self.interpolation = "area" if shrinks else "cubic"
        elif interpolation in ("nearest", "linear", "area", "cubic", "lanczos"):
            self.interpolation = interpolation
        else:
            raise ValueError("Unknown interpolation method: %s" % (interpolation,))
        while isinstance(source, FilterResize):        # contract with parent resize filters
            logger.debug("Combine this resize filter with the parent one.")
            source = source._source
        super(FilterResize, self).__init__(source, size=(width, height))
        logger.debug("Created filter for resizing to size %dx%d", width, height)

    def _process_frame(self, frame):
        if self.interpolation:
            frame = np.asarray(frame)
            if frame.dtype not in (np.uint8, np.float32):
                raise TypeError("FilterResize: uint8 and float32 frames are supported on the GPU path")
            frame = ops.resize(frame, self.size, self.interpolation, color=frame.ndim == 3)
        return super(FilterResize, self)._process_frame(frame)


class FilterMonochrome(VideoFilterBase):
    """colour -> monochrome: mode 'mean' (float64 mean of the channels, truncated) or a
    channel name/index (reference :348-374)"""

    def __init__(self, source, mode="mean"):
        mode = mode.lower() if isinstance(mode, str) else mode
        self.mode = COLOR_CHANNELS.get(mode, mode)
        if self.mode != "mean" and self.mode not in (0, 1, 2):
            raise ValueError("Unsupported conversion method to monochrome: %s" % (mode,))
        super(FilterMonochrome, self).__init__(source, is_color=False)

    def _process_frame(self, frame):
        if self.mode == "mean":
            if frame.dtype != np.uint8:
                raise TypeError("FilterMonochrome(mean): only uint8 frames on the GPU path")
            frame = ops.mono_mean(frame)
        else:
            frame = frame[:, :, self.mode]
        return super(FilterMonochrome, self)._process_frame(frame)


class _GpuStage(object):
    """Mixin of the filters whose pixel work is a stage of `FrameEngine`'s fused chain
    (FilterBackground -> FilterBlur -> FilterThreshold -> FilterMorphology).

    Consecutive stages CONTRACT: when such a filter is read and its source is another stage, the
    whole run of stages is executed by one engine on batches of frames of the first non-stage
    source, instead of one upload / kernel / download round trip per filter and frame -- the
    reference's own idiom for consecutive crops (video/filters.py:209-215 there), applied to the
    GPU filters.  Results are identical (same kernels, same arithmetic).  Listeners of every
    stage are still notified; a stage whose frames the engine does not materialise (the
    difference image under a blur, the mask under a morphology) and that has listeners keeps
    the chain uncontracted."""

    contract = True          # class-wide switch (tests compare both paths)
    chain_batch = 32         # frames per engine pass

    def _stage(self):
        """(kind, parameters) of this filter as a stage of the engine"""
        raise NotImplementedError

    def _chain_plan(self):
        """engine arguments for the run of stages ending here, or None when it does not contract"""
        stages, node = [], self
        while isinstance(node, _GpuStage):
            stages.append(node)
            node = node._source
        stages.reverse()
        # pointwise pre-stages below the first engine stage fold into one device pass in front of the
        # chain (FrameEngine(prepare=...)): [FilterCrop] -> [FilterMonochrome] -> [FilterNormalize with
        # explicit uint8 limits], in this order, none of them listened to
        prepare, pre_nodes = None, 0
        norm = mono = rect = None
        if isinstance(node, FilterNormalize) and not node._listeners and node._fmin is not None \
                and node._fmax is not None and node._dtype == np.uint8:
            tmin, tmax = get_color_range(np.uint8)
            norm = (node._fmin, node._fmax, (tmax - tmin) / (float(node._fmax) - float(node._fmin)), tmin)
            node, pre_nodes = node._source, pre_nodes + 1
        if isinstance(node, FilterMonochrome) and not node._listeners and node._source.is_color:
            mono = node.mode
            node, pre_nodes = node._source, pre_nodes + 1
        if isinstance(node, FilterCrop) and not node._listeners and \
                (node.color_channel is None or (mono is None and node._source.is_color)):
            rect = node.rect
            if node.color_channel is not None:
                mono = node.color_channel
            node, pre_nodes = node._source, pre_nodes + 1
        if pre_nodes:
            if (mono is None) == bool(node.is_color):      # colour frames must end up monochrome
                node, pre_nodes = stages[0]._source, 0
            else:
                prepare = dict(src_size=node.size, src_channels=3 if node.is_color else 1, rect=rect,
                               mono=mono, normalize=norm)
        root = node
        if not self.contract or len(stages) + pre_nodes < 2 or not root.seekable or \
                (root.is_color and prepare is None):
            return None         # (batched prefetch needs random access: fork clients, pipes stay per-frame)
        order = {"background": 0, "blur": 1, "threshold": 2, "morphology": 3}
        kinds = [st._stage()[0] for st in stages]
        ranks = [order[k] for k in kinds]
        if ranks != sorted(ranks) or any(kinds.count(k) > 1 for k in ("background", "blur", "threshold")):
            return None
        if "morphology" in kinds and "threshold" not in kinds:
            return None                     # the engine's morphology works on thresholded masks
        args = dict(background=None, rate=0.02, sigma=0.0, threshold=None, morphology=(), tap_rule="cv4",
                    connectivity=0, static_background=None, prepare=prepare, size=stages[0]._source.size)
        maxval = 255
        steps = []
        for st in stages:
            kind, par = st._stage()
            if kind == "background":
                args.update(background=par["mode"], rate=par["rate"], static_background=par["background"])
            elif kind == "blur":
                args["sigma"], args["tap_rule"] = par["sigma"], par["tap_rule"]
            elif kind == "threshold":
                args["threshold"], maxval = par["threshold"], par["maxval"]
            else:
                steps += par["steps"]
        if len(steps) > 4 or ("morphology" in kinds and maxval != 255) or (args["sigma"] and args["sigma"] <= 0):
            return None
        args["morphology"] = tuple(steps)
        # what the engine materialises: 'filtered' = frames before the threshold, 'mask' = final mask
        masks = "threshold" in kinds
        pre = [st for st, k in zip(stages, kinds) if k in ("background", "blur")]
        provides = {}
        if pre:
            provides[id(pre[-1])] = "filtered"
        if masks:
            provides[id(stages[-1])] = "mask"
        if id(self) not in provides:
            return None
        for st in stages[:-1]:
            if st._listeners and id(st) not in provides:
                return None                 # its frames never exist in the fused chain
        return dict(root=root, args=args, maxval=maxval, output=provides[id(self)], stages=stages,
                    notify=[(st, provides[id(st)]) for st in stages[:-1] if st._listeners])

    def _runner(self):
        plan = self._chain_plan()
        key = None if plan is None else (id(plan["root"]), repr(sorted(plan["args"].items(), key=str)),
                                         plan["maxval"], plan["output"],
                                         tuple(id(st) for st, _ in plan["notify"]))
        if key != getattr(self, "_runner_key", "unset"):
            old = getattr(self, "_runner_obj", None)
            if old is not None:
                old.close(propagate=False)
            self._runner_obj = None
            if plan is not None:
                a = plan["args"]
                extra = tuple(sorted({kind for _, kind in plan["notify"]} - {plan["output"]}))
                self._runner_obj = FilterAnalysisChain(
                    plan["root"], background=a["background"], rate=a["rate"], sigma=a["sigma"],
                    threshold=a["threshold"], morphology=a["morphology"], connectivity=0,
                    output=plan["output"], batch=self.chain_batch, extra_outputs=extra,
                    static_background=a["static_background"], prepare=a["prepare"], size=a["size"],
                    tap_rule=a["tap_rule"])
                self._runner_obj._engine_args["maxval"] = plan["maxval"]
                self._runner_notify = plan["notify"]
                for st in plan["stages"]:           # (a stage asked for its own state finds the reader)
                    st._contracted_reader = weakref.ref(self)
            self._runner_key = key
        return self._runner_obj

    def _contracted_frame(self, runner, index):
        try:
            frame = runner.get_frame(index)
        except TypeError:                   # frames the engine does not take (not uint8): per-filter path
            _GpuStage._disable(self)
            return None
        for stage, kind in self._runner_notify:
            VideoBase._process_frame(stage, runner.last_results[kind])
        self._contracted_pos = index + 1            # frames [0, index] have gone through the chain
        return VideoBase._process_frame(self, frame)

    def _disable(self):
        self.contract = False
        self._runner()

    def get_frame(self, index):
        runner = self._runner()
        if runner is not None:
            if index < 0:
                index += self.frame_count
            if not 0 <= index < self.frame_count:
                raise IndexError("frame %d is out of range" % index)
            out = self._contracted_frame(runner, index)
            if out is not None:
                self._frame_pos = index
                return out
        return super(_GpuStage, self).get_frame(index)

    def get_next_frame(self):
        runner = self._runner()
        if runner is not None:
            if self._frame_pos >= self.frame_count:
                raise StopIteration
            out = self._contracted_frame(runner, self._frame_pos)
            if out is not None:
                self._frame_pos += 1
                return out
            self._source.set_frame_pos(self._frame_pos)
        return super(_GpuStage, self).get_next_frame()

    def set_frame_pos(self, index):
        if self._runner() is not None:
            if index < 0:
                index += self.frame_count
            if not 0 <= index < self.frame_count:
                raise IndexError("Seeking to frame %d was not possible." % index)
            self._frame_pos = index
            return
        super(_GpuStage, self).set_frame_pos(index)

    def get_frame_pos(self):
        if self._runner() is not None:
            return self._frame_pos
        return super(_GpuStage, self).get_frame_pos()

    def close(self, propagate=True):
        old = getattr(self, "_runner_obj", None)
        if old is not None:
            old.close(propagate=False)
            self._runner_obj = None
            self._runner_key = None
        super(_GpuStage, self).close(propagate)


class FilterBlur(_GpuStage, VideoFilterBase):
    """Gaussian blur of standard deviation `sigma` (reference :378-392:
    cv2.GaussianBlur(frame.astype(np.uint8), (0, 0), sigma)).  Unlike the reference
     listeners ARE notified."""

    def __init__(self, source, sigma=3, tap_rule="cv4"):
        """tap_rule: which OpenCV's 8-bit Gaussian -- 'cv4' (>= 4.x: fixed-point taps with error
        diffusion, sum 256) or 'cv3' (2.4 / 3.x, the reference's era: every float32 tap rounded on its
        own).  Both are written definitions, unverifiable offline."""
        self.sigma = sigma
        if tap_rule not in ("cv4", "cv3"):
            raise ValueError("tap_rule must be 'cv4' or 'cv3'")
        self.tap_rule = tap_rule
        super(FilterBlur, self).__init__(source)

    def _stage(self):
        return "blur", {"sigma": float(self.sigma), "tap_rule": self.tap_rule}

    def _process_frame(self, frame):
        frame = np.asarray(frame).astype(np.uint8)          # C truncation/wrap like the reference
        out = ops.gaussian_blur(frame, self.sigma, color=frame.ndim == 3, tap_rule=self.tap_rule)
        return super(FilterBlur, self)._process_frame(out)


class FilterThreshold(_GpuStage, VideoFilterBase):
    """BUILD-DEFINED: binary threshold `frame > threshold ? maxval : 0` on uint8 frames
    (the reference builds such masks with NumPy comparisons, video/analysis/image.py:282-304)"""

    def __init__(self, source, threshold=127, maxval=255):
        self.threshold, self.maxval = int(threshold), int(maxval)
        if not 0 <= self.maxval <= 255:
            raise ValueError("maxval must be in [0, 255]")
        super(FilterThreshold, self).__init__(source)

    def _stage(self):
        return "threshold", {"threshold": self.threshold, "maxval": self.maxval}

    def _process_frame(self, frame):
        frame = np.asarray(frame)
        if frame.dtype != np.uint8:
            raise TypeError("FilterThreshold expects uint8 frames")
        return super(FilterThreshold, self)._process_frame(
            ops.threshold(frame, self.threshold, self.maxval))


class FilterMorphology(_GpuStage, VideoFilterBase):
    """BUILD-DEFINED: erode / dilate / open / close with a `ksize` x `ksize` structuring
    element of `shape` in {'rect','cross','ellipse'} (primitive: cv2.erode/cv2.dilate as used at
    video/analysis/image.py:248-251)"""

    _SEQUENCES = {"erode": ("erode",), "dilate": ("dilate",), "open": ("erode", "dilate"),
                  "close": ("dilate", "erode")}

    def __init__(self, source, operation="close", ksize=5, shape="rect"):
        if operation not in self._SEQUENCES:
            raise ValueError("operation must be one of %s" % sorted(self._SEQUENCES))
        if shape not in ("rect", "cross", "ellipse"):
            raise ValueError("shape must be 'rect', 'cross' or 'ellipse'")
        if int(ksize) < 1:
            raise ValueError("ksize must be >= 1")
        self.operation, self.ksize, self.element = operation, int(ksize), shape
        if source.is_color:
            raise ValueError("FilterMorphology expects a monochrome video")
        super(FilterMorphology, self).__init__(source)

    @property
    def steps(self):
        return [(op, self.element, self.ksize) for op in self._SEQUENCES[self.operation]]

    def _stage(self):
        return "morphology", {"steps": self.steps}

    def _process_frame(self, frame):
        frame = np.asarray(frame)
        if frame.dtype != np.uint8 or frame.ndim != 2:
            raise TypeError("FilterMorphology expects 2-d uint8 frames")
        for op, shape, ksize in self.steps:
            frame = ops.morph(frame, op, shape, ksize)
        return super(FilterMorphology, self)._process_frame(frame)


class _FrameIndexFilter(VideoFilterBase):
    """filters that only re-map frame indices (no pixel work): frame k of this video is frame
    `_source_index(k)` of the source"""

    def _source_index(self, index):
        raise NotImplementedError

    def _check_index(self, index):
        if index < 0:
            index += self.frame_count
        if not 0 <= index < self.frame_count:
            raise IndexError("Cannot access frame %d." % index)
        return index

    def set_frame_pos(self, index):
        self._frame_pos = self._check_index(index)

    def get_frame_pos(self):
        return self._frame_pos

    def get_frame(self, index):
        index = self._check_index(index)
        self._frame_pos = index + 1
        return self._process_frame(self._source.get_frame(self._source_index(index)))

    def get_next_frame(self):
        if self._frame_pos >= self.frame_count:
            raise StopIteration
        return self.get_frame(self._frame_pos)


class FilterRotate(VideoFilterBase):
    """returns the video rotated in counter-clockwise direction (reference :319-344):
    `np.rot90(frame, angle // 90)` on the GPU"""

    def __init__(self, source, angle=0):
        angle = angle % 360
        if angle in (0, 180):
            size = source.size
        elif angle in (90, 270):
            size = (source.size[1], source.size[0])
        else:
            raise ValueError('angle must be from [0, 90, 180, 270] but was %s' % angle)
        self.angle = angle
        super(FilterRotate, self).__init__(source, size=size)

    def _process_frame(self, frame):
        frame = np.asarray(frame)
        if self.angle:
            frame = ops.rot90(frame, self.angle // 90, color=frame.ndim == 3)
        return super(FilterRotate, self)._process_frame(frame)


class FilterReplicate(_FrameIndexFilter):
    """plays the source `count` times in a row (reference :396-430)"""

    def __init__(self, source, count=1):
        self.count = int(count)
        if self.count < 1:
            raise ValueError("count must be at least 1")
        super(FilterReplicate, self).__init__(source, frame_count=source.frame_count * self.count)

    def _source_index(self, index):
        return index % self._source.frame_count


class FilterDropFrames(_FrameIndexFilter):
    """keeps every `compression`-th frame (fractional factors allowed) and lowers the frame rate
    accordingly (reference :434-483)"""

    def __init__(self, source, compression=1):
        if compression < 1:
            raise ValueError("compression must be at least 1")
        self._compression = compression
        super(FilterDropFrames, self).__init__(
            source, frame_count=int((source.frame_count - 1) / compression) + 1,
            fps=source.fps / compression)

    def _source_index(self, index):
        return int(index * self._compression)


# ======================================================================================
# filters with temporal state
# ======================================================================================

class FilterDiffBase(VideoFilterBase):
    """compares consecutive frames: frame k of the filter = compare(source[k+1], source[k])
    (reference :492-538; iteration fixed to go through get_next_frame)"""

    def __init__(self, source):
        self._prev_frame = None
        super(FilterDiffBase, self).__init__(source, frame_count=source.frame_count - 1)

    def _compare_frames(self, this_frame, prev_frame):
        raise NotImplementedError

    def set_frame_pos(self, index):
        if index < 0:
            index += self.frame_count
        self._source.set_frame_pos(index)
        self._prev_frame = np.array(self._source.get_next_frame())
        self._frame_pos = index

    def get_frame_pos(self):
        return self._frame_pos

    def get_frame(self, index):
        if index < 0:
            index += self.frame_count
        if not 0 <= index < self.frame_count:
            raise IndexError("frame %d is out of range" % index)
        out = self._compare_frames(self._source.get_frame(index + 1), self._source.get_frame(index))
        return self._process_frame(out)

    def get_next_frame(self):
        if self._prev_frame is None:
            self.set_frame_pos(self._frame_pos)
        this_frame = np.array(self._source.get_next_frame())
        out = self._compare_frames(this_frame, self._prev_frame)
        self._prev_frame = this_frame
        self._frame_pos += 1
        return self._process_frame(out)


class FilterTimeDifference(FilterDiffBase):
    """this_frame.astype(int16) - prev_frame (reference :542-568)"""

    def __init__(self, source, dtype=np.int16):
        if dtype is not None and np.dtype(dtype) != np.int16:
            raise TypeError("FilterTimeDifference: the GPU path computes int16 differences")
        self._dtype = np.int16
        super(FilterTimeDifference, self).__init__(source)

    def _compare_frames(self, this_frame, prev_frame):
        if this_frame.dtype != np.uint8 or prev_frame.dtype != np.uint8:
            raise TypeError("FilterTimeDifference expects uint8 frames")
        return ops.time_difference(this_frame, prev_frame)


class _SequentialStateFilter(VideoFilterBase):
    """helper for filters whose output at frame k depends on frames < k: sequential iteration
    is incremental, a seek replays the history from frame 0 (state only)"""

    def _reset_state(self):
        raise NotImplementedError

    def _advance_state(self, frames):
        """fold frames in without producing output"""
        raise NotImplementedError

    def _emit(self, frame):
        raise NotImplementedError

    _needs_history = True       # False: the state does not depend on earlier frames

    def _seek_state(self, index):
        self._reset_state()
        chunk = 32
        for a in range(0, index if self._needs_history else 0, chunk):
            frames = np.stack([np.asarray(self._source.get_frame(k))
                               for k in range(a, min(index, a + chunk))])
            self._advance_state(frames)
        self._state_pos = index

    def set_frame_pos(self, index):
        if index < 0:
            index += self.frame_count
        if not 0 <= index < self.frame_count:
            raise IndexError("Seeking to frame %d was not possible." % index)
        if index != getattr(self, "_state_pos", None):
            self._seek_state(index)
        self._source.set_frame_pos(index)
        self._frame_pos = index

    def get_frame_pos(self):
        return self._frame_pos

    def get_frame(self, index):
        if index < 0:
            index += self.frame_count
        if not 0 <= index < self.frame_count:
            raise IndexError("frame %d is out of range" % index)
        if index != getattr(self, "_state_pos", None):
            self._seek_state(index)
        out = self._emit(np.asarray(self._source.get_frame(index)))
        self._state_pos = index + 1
        self._frame_pos = index
        return self._process_frame(out)

    def get_next_frame(self):
        if self._frame_pos != getattr(self, "_state_pos", None):
            self._seek_state(self._frame_pos)
            self._source.set_frame_pos(self._frame_pos)
        out = self._emit(np.asarray(self._source.get_next_frame()))
        self._frame_pos += 1
        self._state_pos = self._frame_pos
        return self._process_frame(out)


class FilterBackground(_GpuStage, _SequentialStateFilter):
    """BUILD-DEFINED running background subtraction.

    frame k -> sat_u8(trunc(|frame_k - bg_{k-1}|)), then the model is updated with frame k.
    mode 'mean'  : bg = cumulative mean with measure_mean's float64 arithmetic
                   `mean*n/(n+1) + frame/(n+1)` (video/analysis/video.py:33); bg_{-1} = 0
    mode 'ema'   : bg += rate*(frame - bg) in float32; the first frame initialises bg
    mode 'static': a fixed background image `background` (e.g. analysis.video.measure_mean())
    """

    def __init__(self, source, mode="mean", rate=0.02, background=None):
        if mode not in ("mean", "ema", "static"):
            raise ValueError("mode must be 'mean', 'ema' or 'static'")
        if mode == "static" and background is None:
            raise ValueError("mode='static' needs a background image")
        self.mode, self.rate = mode, float(rate)
        self._background = background
        self._model = None
        super(FilterBackground, self).__init__(source)

    @property
    def _needs_history(self):
        return self.mode != "static"

    def _frame_shape(self):
        return self.shape[1:]

    def _reset_state(self):
        self._model = ops.BackgroundModel(self._frame_shape(), self.mode, self.rate, np.uint8,
                                          self._background)

    def _advance_state(self, frames):
        if self.mode != "static":
            self._model.process(frames.astype(np.uint8, copy=False), want_diff=False)

    def _emit(self, frame):
        if frame.dtype != np.uint8:
            raise TypeError("FilterBackground expects uint8 frames")
        return self._model.process(frame[None])[0]

    def _stage(self):
        return "background", {"mode": self.mode, "rate": self.rate, "background": self._background}

    @property
    def background(self):
        """current background model (float64 / float32 array): the state after the frames read so far.
        When this filter runs as a stage of a contracted chain (its frames come out of a fused engine
        that works a batch ahead) the state is brought to the reader's position with the state update
        alone on this filter's own model -- the same arithmetic, so contracted and per-filter reads
        report the same background."""
        ref = getattr(self, "_contracted_reader", None)
        reader = ref() if ref is not None else None
        if reader is not None and getattr(reader, "_runner_obj", None) is not None and self.mode != "static":
            pos = getattr(reader, "_contracted_pos", 0)
            have = getattr(self, "_state_pos", None)
            if self._model is None or have is None or have > pos:
                self._seek_state(pos)
            elif have < pos:
                for a in range(have, pos, 32):
                    self._advance_state(np.stack([np.asarray(self._source.get_frame(k))
                                                  for k in range(a, min(pos, a + 32))]))
                self._state_pos = pos
            return self._model.state
        if self._model is None:
            self._reset_state()
            self._state_pos = 0
        return self._model.state


class FilterAnalysisChain(_SequentialStateFilter):
    """BUILD-DEFINED: the whole chain, fused and batched on the GPU.

        FilterBackground -> FilterBlur -> FilterThreshold -> FilterMorphology -> label

    Pulls `batch` frames from the source at a time, runs one `FrameEngine` pass over them and
    hands the results out frame by frame, so it is still a lazy `VideoFilterBase` for
    downstream code.  `output` selects what a frame of this video is: 'mask' (uint8),
    'labels' (int32) or 'filtered' (blurred, background-subtracted uint8).  After each frame
    `last_count` (number of objects) and, when max_labels > 0, `last_stats` are set.
    Results are identical to chaining the individual filters.
    """

    def __init__(self, source, background="mean", rate=0.02, sigma=5.0, threshold=20,
                 morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4,
                 output="mask", batch=32, max_labels=0, extra_outputs=(), static_background=None,
                 prepare=None, size=None, tap_rule="cv4"):
        if source.is_color and prepare is None:
            raise ValueError("FilterAnalysisChain expects a monochrome video")
        if output not in ("mask", "labels", "filtered"):
            raise ValueError("output must be 'mask', 'labels' or 'filtered'")
        if output == "labels" and not connectivity:
            raise ValueError("output='labels' needs connectivity 4 or 8")
        self.output = output
        self.extra_outputs = tuple(extra_outputs)        # further per-frame results kept in last_results
        self._static_background = static_background
        self.last_results = {}
        self.batch = int(batch)
        self._engine_args = dict(size=source.size if size is None else size, channels=1, dtype=np.uint8,
                                 max_batch=self.batch, background=background, bg_rate=rate, sigma=sigma,
                                 thresh=threshold, morphology=morphology, connectivity=connectivity,
                                 max_labels=max_labels, prepare=prepare, tap_rule=tap_rule)
        self._engine = None
        self._cache = {}            # frame index -> dict of per-frame results
        self.last_count = None
        self.last_stats = None
        super(FilterAnalysisChain, self).__init__(source, size=size, is_color=False)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = FrameEngine(**self._engine_args)
            if self._engine_args["background"] == "static":
                self._engine.set_background(self._static_background, 0)
        return self._engine

    @property
    def _needs_history(self):
        return self._engine_args["background"] in ("mean", "ema")

    def _wanted(self):
        want = {self.output} | set(self.extra_outputs)
        if self._engine_args["connectivity"]:
            want.add("counts")
        if self._engine_args["max_labels"] > 0:
            want.add("stats")
        return want

    def _reset_state(self):
        if self._engine_args["background"] in ("mean", "ema"):
            self.engine.set_background(None, 0)
        self._cache = {}

    def _advance_state(self, frames):
        if self._engine_args["background"] in ("mean", "ema"):
            for a in range(0, len(frames), self.batch):
                self.engine.run(frames[a:a + self.batch], want=())

    def _run_batch(self, first):
        last = min(self.frame_count, first + self.batch)
        frames = np.stack([np.asarray(self._source.get_frame(k)) for k in range(first, last)])
        if frames.dtype != np.uint8:
            raise TypeError("FilterAnalysisChain expects uint8 frames")
        res = self.engine.run(frames, want=self._wanted())
        self._cache = {first + i: {k: v[i] for k, v in res.items()} for i in range(last - first)}
        self._state_pos = last                      # the model has seen frames < last

    def _result(self, index):
        if index not in self._cache:
            if self._engine is None or index != getattr(self, "_state_pos", None):
                self._seek_state(index)
            self._run_batch(index)
        r = self._cache[index]
        self.last_count = int(r["counts"]) if "counts" in r else None
        self.last_stats = r.get("stats")
        self.last_results = r
        return r[self.output]

    def set_frame_pos(self, index):
        if index < 0:
            index += self.frame_count
        if not 0 <= index < self.frame_count:
            raise IndexError("Seeking to frame %d was not possible." % index)
        self._frame_pos = index

    def get_frame(self, index):
        if index < 0:
            index += self.frame_count
        if not 0 <= index < self.frame_count:
            raise IndexError("frame %d is out of range" % index)
        self._frame_pos = index
        return self._process_frame(self._result(index))

    def get_next_frame(self):
        if self._frame_pos >= self.frame_count:
            raise StopIteration
        out = self._result(self._frame_pos)
        self._frame_pos += 1
        return self._process_frame(out)

    def close(self, propagate=True):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        super(FilterAnalysisChain, self).close(propagate)
