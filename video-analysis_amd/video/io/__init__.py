"""video.io -- frame sources and the VideoBase / VideoFilterBase protocol (host plumbing).

Conventions (as in the reference, video/io/base.py:119-125): ``size = (width, height)``,
arrays are ``(frames, height, width[, 3])``; frames are NumPy arrays and sources may hand out
views of their storage, so consumers copy what they keep.
"""
from .base import (NotSeekableError, SynchronizationError, VideoBase, VideoFilterBase,  # noqa
                   VideoFork, VideoIterator, VideoSlice)
from .memory import VideoMemory  # noqa
from .computed import VideoGaussianNoise  # noqa
