"""Synthetic frame sources (role of the reference's `video/io/computed.py:15-41`).

`VideoGaussianNoise` produces frames of Gaussian noise.  Every frame has its own seed derived
from `(seed, index)`, so the video is seekable and `video[i]` is reproducible -- which the
reference's `np.random.randn` stream is not.
"""
import numpy as np

from .base import VideoBase


class VideoGaussianNoise(VideoBase):
    seekable = True

    def __init__(self, frame_count=100, size=(128, 128), mean=0, std=1, fps=None, is_color=False,
                 dtype=np.uint8, seed=0):
        super(VideoGaussianNoise, self).__init__(size=size, frame_count=frame_count, fps=fps,
                                                 is_color=is_color)
        self.mean = mean
        self.std = std
        self.seed = seed
        self.dtype = np.dtype(dtype)
        self._limits = None
        if np.issubdtype(self.dtype, np.integer):
            info = np.iinfo(self.dtype)
            self._limits = (info.min, info.max)

    def get_frame(self, index):
        index = index + self.frame_count if index < 0 else index
        if index < 0 or index >= self.frame_count:
            raise IndexError("frame %d is out of range" % index)
        noise = np.random.default_rng((self.seed, index)).standard_normal(self.shape[1:])
        frame = noise * self.std + self.mean
        if self._limits is not None:                 # integer videos saturate instead of wrapping
            frame = np.clip(frame, *self._limits)
        return frame.astype(self.dtype)
