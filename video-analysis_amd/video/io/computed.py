"""Synthetic frame sources (role of the reference's `video/io/computed.py:15-41`).

`VideoGaussianNoise` produces frames of Gaussian noise ON THE GPU: sample i of the video's noise
stream is a pure function of `(seed, i)` (Philox counter + Box-Muller, `va_gaussian_noise`), so
the video is seekable, `video[k]` is reproducible -- which the reference's unseeded
`np.random.randn` stream is not -- and a batch of frames can be generated straight into device
memory for the engine (`fill_device`), without ever crossing the host link.
"""
import numpy as np

from .base import VideoBase


class VideoGaussianNoise(VideoBase):
    seekable = True

    def __init__(self, frame_count=100, size=(128, 128), mean=0, std=1, fps=None, is_color=False,
                 dtype=None, seed=0):
        """dtype None: float64 frames like the reference (`mean + std*randn(...)`); integer videos
        (uint8) saturate instead of wrapping"""
        super(VideoGaussianNoise, self).__init__(size=size, frame_count=frame_count, fps=fps,
                                                 is_color=is_color)
        self.mean = mean
        self.std = std
        self.seed = seed
        self.dtype = np.dtype(np.float64 if dtype is None else dtype)
        if self.dtype not in (np.uint8, np.float32, np.float64):
            raise TypeError("VideoGaussianNoise: uint8, float32 and float64 frames on the GPU path")
        self._frame_shape = self.shape[1:]
        self._frame_samples = int(np.prod(self._frame_shape))

    def get_frame(self, index):
        from .. import ops
        index = index + self.frame_count if index < 0 else index
        if index < 0 or index >= self.frame_count:
            raise IndexError("frame %d is out of range" % index)
        return ops.gaussian_noise(self._frame_shape, self.dtype, self.mean, self.std, self.seed,
                                  first_index=index * self._frame_samples)

    def fill_device(self, dev_ptr, first_frame, count, stream=None):
        """generate frames [first_frame, first_frame + count) at device address `dev_ptr`
        (e.g. a torch tensor's data_ptr()): the input of `FrameEngine.run_device`, no host copy"""
        from .. import _hip
        if first_frame < 0 or first_frame + count > self.frame_count:
            raise IndexError("frames %d..%d are out of range" % (first_frame, first_frame + count))
        code = {np.dtype(np.uint8): _hip.VA_U8, np.dtype(np.float32): _hip.VA_F32,
                np.dtype(np.float64): _hip.VA_F64}[self.dtype]
        _hip.check(_hip.lib().va_gaussian_noise(dev_ptr, code, count * self._frame_samples, float(self.mean),
                                                float(self.std), int(self.seed) & (2 ** 64 - 1),
                                                first_frame * self._frame_samples, stream))
