"""Temporal statistics of a video (reference: video/analysis/video.py:14-55).

measure_mean / measure_mean_std keep their float64 state on the GPU and fold frames in with
exactly the reference's arithmetic (`mean*n/(n+1) + frame/(n+1)`; Welford for the variance)."""
import numpy as np


def reduce_video(video, function, initial_value=None):
    """folds `function(frame, result)` over the frames (host callable)"""
    result = initial_value
    for frame in video:
        result = frame if result is None else function(frame, result)
    return result


_FRAME_DTYPES = (np.uint8, np.int16, np.float32)


def _batches(video, batch):
    """stacks of `batch` frames; uint8, int16 (FilterTimeDifference) and float32 videos"""
    buf = []
    for frame in video:
        frame = np.asarray(frame)
        if frame.dtype not in _FRAME_DTYPES:
            raise TypeError("the GPU path takes uint8, int16 or float32 frames, got %s" % frame.dtype)
        buf.append(np.array(frame))
        if len(buf) == batch:
            yield np.stack(buf)
            buf = []
    if buf:
        yield np.stack(buf)


def measure_mean(video, batch=32):
    """mean of every pixel over time, float64 (reference :26-35)"""
    from .. import ops
    mean, n = None, 0
    model = None
    for frames in _batches(video, batch):
        if frames.dtype == np.uint8:            # device-resident state, division-free kernel
            if model is None:
                model = ops.BackgroundModel(video.shape[1:], "mean", dtype=np.uint8)
            model.process(frames, want_diff=False)
        else:                                   # int16 / float32 frames: NumPy's promotions restated
            mean = ops.running_mean(frames, mean, n)
        n += len(frames)
    if model is not None:
        return model.state
    return np.zeros(video.shape[1:]) if mean is None else mean


def measure_mean_std(video, batch=32):
    """mean and standard deviation of every pixel over time (reference :39-55)"""
    from .. import ops
    mean = m2 = None
    n = 0
    last = None
    for frames in _batches(video, batch):
        mean, m2 = ops.welford(frames, mean, m2, n)
        n += len(frames)
        last = frames[-1]
    if n == 0:
        raise ValueError("video is empty")
    if n - 1 < 2:                       # reference: `if n < 2` with n = last frame index
        return last, 0
    return mean, np.sqrt(m2 / (n - 1))
