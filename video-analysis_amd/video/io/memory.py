"""VideoMemory -- a video held in a NumPy array (reference: video/io/memory.py:17-64).

Also the natural host-side batch container for the GPU engine: `data[a:b]` is one contiguous
upload."""
import numpy as np

from .base import VideoBase


class VideoMemory(VideoBase):
    write_access = True
    seekable = True

    def __init__(self, data, fps=25, copy_data=True):
        data = np.array(data, copy=True) if copy_data else np.asarray(data)
        if data.ndim == 4 and data.shape[3] == 1:
            data = data[..., 0]                  # drop a singleton colour axis
        if data.ndim == 3:
            is_color = False
        elif data.ndim == 4 and data.shape[3] == 3:
            is_color = True
        else:
            raise ValueError("The last dimension of the data must be either 1 or 3.")
        self.data = data
        super(VideoMemory, self).__init__(size=(data.shape[2], data.shape[1]),
                                          frame_count=data.shape[0], fps=fps, is_color=is_color)

    def get_frame(self, index):
        if index < 0:
            index += self.frame_count
        if not 0 <= index < self.frame_count:
            raise IndexError("frame %d is out of range" % index)
        return self.data[index]                  # a view, like the reference

    def __getitem__(self, key):
        return self.data[key]

    def __setitem__(self, key, value):
        self.data[key] = value
