"""synthetic sources (reference: video/io/computed.py:15-41)"""
import numpy as np

from .base import VideoBase


class VideoGaussianNoise(VideoBase):
    """frames of clipped Gaussian noise; seeded per frame so that get_frame(i) is repeatable"""
    seekable = True

    def __init__(self, frame_count=100, size=(128, 128), mean=0, std=1, fps=None,
                 is_color=False, dtype=np.uint8, seed=0):
        self.mean, self.std, self.dtype, self.seed = mean, std, np.dtype(dtype), seed
        super(VideoGaussianNoise, self).__init__(size=size, frame_count=frame_count, fps=fps,
                                                 is_color=is_color)

    def get_frame(self, index):
        if index < 0:
            index += self.frame_count
        if not 0 <= index < self.frame_count:
            raise IndexError("frame %d is out of range" % index)
        rng = np.random.default_rng((self.seed, index))
        frame = self.mean + self.std * rng.standard_normal(self.shape[1:])
        if np.issubdtype(self.dtype, np.integer):
            info = np.iinfo(self.dtype)
            frame = np.clip(frame, info.min, info.max)
        return frame.astype(self.dtype)
