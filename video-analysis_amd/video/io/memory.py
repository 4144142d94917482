"""In-memory videos.

`VideoMemory` wraps a ``(frames, height, width[, 3])`` NumPy array as a seekable, writable video
(the reference's `video/io/memory.py:17-64`).  Because the array is contiguous in time it is also
the natural staging container for the GPU engine: `batches()` yields zero-copy slices that
`FrameEngine.run` / `StreamedEngine.submit` upload in one transfer.
"""
import numpy as np

from .base import VideoBase


def _format_of(array):
    """(frame_count, (width, height), is_color) of a video array, ValueError if it is not one"""
    if array.ndim == 3:
        return array.shape[0], (array.shape[2], array.shape[1]), False
    if array.ndim == 4 and array.shape[3] == 3:
        return array.shape[0], (array.shape[2], array.shape[1]), True
    raise ValueError("The last dimension of the data must be either 1 or 3.")


class VideoMemory(VideoBase):
    """all frames live in `self.data`; `get_frame` hands out views, writes go straight through"""

    write_access = True
    seekable = True

    def __init__(self, data, fps=25, copy_data=True):
        array = np.array(data, copy=True) if copy_data else np.asarray(data)
        if array.ndim == 4 and array.shape[3] == 1:      # (T, H, W, 1) is a monochrome video
            array = array.reshape(array.shape[:3])
        count, size, color = _format_of(array)
        self.data = array
        super(VideoMemory, self).__init__(size=size, frame_count=count, fps=fps, is_color=color)

    # frame access ---------------------------------------------------------------------------
    def get_frame(self, index):
        index = index + self.frame_count if index < 0 else index
        if index < 0 or index >= self.frame_count:
            raise IndexError("frame %d is out of range" % index)
        return self.data[index]

    def __getitem__(self, key):
        return self.data[key]

    def __setitem__(self, key, value):
        self.data[key] = value

    # batching for the GPU engine ------------------------------------------------------------
    def batches(self, batch_size, start=0, stop=None):
        """yields (first_frame_index, view of `batch_size` consecutive frames); the last batch
        may be shorter"""
        if batch_size < 1:
            raise ValueError("batch_size must be positive")
        stop = self.frame_count if stop is None else min(stop, self.frame_count)
        for first in range(start, stop, batch_size):
            yield first, self.data[first:min(stop, first + batch_size)]


class VideoMemoryBuffer(VideoBase):
    """placeholder for a video that buffers another one in memory; the reference declares it without a
    body as well (video/io/memory.py:68-71) -- the buffering role is `video.streaming.StreamedEngine` here"""
