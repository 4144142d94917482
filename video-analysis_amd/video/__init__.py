"""video -- MI355X-native drop-in for the per-frame hot path of david-zwicker/video-analysis.

Same package/module/class names as the reference (``video.io.base.VideoBase``,
``video.io.memory.VideoMemory``, ``video.filters.Filter*``, ``video.analysis.*``) so that
downstream tracking code keeps working, but every per-pixel operation runs in hand-written HIP
kernels for gfx950 through ``libvideoanalysis_hip.so`` (see ``video._hip``).  There is no
NumPy/OpenCV compute fallback: without the library and a GPU the filters raise
``video._hip.HipUnavailableError``.

Videos are iterated lazily, one frame at a time, exactly like the reference
(``video/__init__.py:4-10`` there); the batched fused path is ``video.engine.FrameEngine`` /
``video.filters.FilterAnalysisChain``.
"""
import logging

logging.getLogger("video").addHandler(logging.NullHandler())

__all__ = ["io", "filters", "analysis", "engine", "ops"]
