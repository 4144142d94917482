"""VideoBase / VideoFilterBase / VideoIterator / VideoSlice -- the drop-in boundary.

Python 3 restatement of the class protocol of the reference's video/io/base.py:26-474 (public
names, argument meaning and error behaviour kept; the Python-2 idioms and the latent bugs
listed in SURVEY.md F6 are not).  Pure host plumbing: no per-pixel work happens here.
"""
import logging

import numpy as np

logger = logging.getLogger("video.io")


class NotSeekableError(RuntimeError):
    """raised when a forward-only video is asked to seek backwards (base.py:21,172)"""


class SynchronizationError(RuntimeError):
    """raised when consumers of a shared video drift apart (base.py:22)"""


def _wrap_index(index, count):
    """Python's negative-index convention for frame numbers"""
    return index + count if index < 0 else index


class VideoBase(object):
    """A finite sequence of equally sized frames plus a read cursor.

    The protocol every consumer of the reference relies on (video/io/base.py:26-269): `size` is
    (width, height), `frame_count`, `fps`, `is_color`; frames come out of `get_frame(i)` (random
    access) or `get_next_frame()` (cursor), both routed through `_process_frame`, which is where
    listeners -- callables registered with `register_listener` -- see every frame handed out."""

    write_access = False
    seekable = False

    def __init__(self, size=(0, 0), frame_count=-1, fps=None, is_color=True):
        if len(size) != 2:
            raise ValueError("Videos must have two spatial dimensions.")
        self.size = tuple(size)
        self.frame_count = frame_count
        self.fps = fps if fps is not None else 25
        self.is_color = is_color
        self._listeners = []
        self._frame_pos = 0                      # the frame get_next_frame() returns next

    # ---- format ------------------------------------------------------------------------------
    width = property(lambda self: self.size[0])
    height = property(lambda self: self.size[1])
    bounds = property(lambda self: (0, 0) + tuple(self.size), doc="(left, top, width, height) of a frame")

    @property
    def shape(self):
        """(frames, height, width) -- plus a trailing 3 for colour videos"""
        w, h = self.size
        return (self.frame_count, h, w, 3) if self.is_color else (self.frame_count, h, w)

    @property
    def video_format(self):
        """the constructor arguments that describe this video's format"""
        return dict(size=self.size, frame_count=self.frame_count, fps=self.fps, is_color=self.is_color)

    def __len__(self):
        return self.frame_count

    def get_property_list(self):
        fmt = self.video_format
        return ("size=(%d, %d)" % tuple(fmt["size"]),) + tuple("%s=%s" % (k, fmt[k])
                                                                for k in ("frame_count", "fps", "is_color"))

    def _listener_suffix(self):
        n = len(self._listeners)
        return "[%d listener%s]" % (n, "" if n == 1 else "s") if n else ""

    def __str__(self):
        return "%s(%s)%s" % (type(self).__name__, ", ".join(self.get_property_list()), self._listener_suffix())

    def info(self):
        return "Video(%s)" % ", ".join(self.get_property_list())

    # ---- listeners ---------------------------------------------------------------------------
    def register_listener(self, listener_callback):
        self._listeners.append(listener_callback)

    def unregister_listener(self, listener_callback):
        self._listeners.remove(listener_callback)

    def _process_frame(self, frame):
        """every outgoing frame passes here; subclasses transform it and call up"""
        for notify in self._listeners:
            notify(frame)
        return frame

    # ---- cursor ------------------------------------------------------------------------------
    def get_frame_pos(self):
        return self._frame_pos

    def set_frame_pos(self, index):
        """random-access videos jump; forward-only ones read and drop frames up to `index` and
        refuse to go back (NotSeekableError)"""
        target = _wrap_index(index, self.frame_count)
        if not self.seekable:
            here = self.get_frame_pos()
            if target < here:
                raise NotSeekableError("Cannot seek to frame %d, because the video is already at "
                                       "frame %d" % (target, here))
            for _ in range(target - here):
                self.get_next_frame()
        elif 0 <= target < self.frame_count:
            self._frame_pos = target
        else:
            raise IndexError("Seeking to frame %d was not possible." % target)

    def rewind(self):
        self.set_frame_pos(0)

    # ---- frames ------------------------------------------------------------------------------
    def get_frame(self, index):
        raise NotImplementedError

    def get_next_frame(self):
        try:
            frame = self.get_frame(self._frame_pos)
        except IndexError:
            raise StopIteration
        self._frame_pos += 1
        return frame

    def abort_iteration(self):
        pass

    def close(self):
        pass

    def __iter__(self):
        return VideoIterator(self)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self.get_frame(int(key))
        if isinstance(key, slice):
            return VideoSlice(self, *key.indices(self.frame_count), _normalized=True)
        raise TypeError("Invalid key `%r` for indexing" % (key,))

    def __setitem__(self, key, value):
        raise ValueError("Writing to this video stream is prohibited.")

    def copy(self, dtype=np.uint8, disp=False):
        """reads the whole video into a VideoMemory (reference: base.py:248-269)"""
        from .memory import VideoMemory
        frames = np.empty(self.shape, dtype)
        for slot, frame in zip(frames, self):
            slot[...] = frame
        return VideoMemory(frames, fps=self.fps, copy_data=False)


class VideoIterator(object):
    """what `iter(video)` returns: starts at frame 0 and ends when the video does (base.py:273-283)"""

    def __init__(self, video):
        video.rewind()
        self._video = video

    def __iter__(self):
        return self

    def __next__(self):
        try:
            return self._video.get_next_frame()
        except IndexError:
            raise StopIteration

    next = __next__         # the spelling reference-era (Python 2) callers use


class VideoFilterBase(VideoBase):
    """A video computed from another one, frame by frame: subclasses override `_process_frame`
    (reference: video/io/base.py:313-388).  Whatever of the format is not given is the source's."""

    def __init__(self, source, size=None, frame_count=None, fps=None, is_color=None):
        self._source = source
        fmt = dict(source.video_format)
        given = dict(size=size, frame_count=frame_count, fps=fps, is_color=is_color)
        fmt.update({k: v for k, v in given.items() if v is not None})
        super(VideoFilterBase, self).__init__(**fmt)

    def __str__(self):
        return "%s +%s%s" % (self._source, type(self).__name__, self._listener_suffix())

    @property
    def seekable(self):
        return self._source.seekable

    def abort_iteration(self):
        self._source.abort_iteration()

    def get_frame_pos(self):
        return self._source.get_frame_pos()

    def set_frame_pos(self, index):
        self._frame_pos = _wrap_index(index, self.frame_count)
        self._source.set_frame_pos(self._frame_pos)

    def get_frame(self, index):
        index = _wrap_index(index, self.frame_count)
        raw = self._source.get_frame(index)
        self._frame_pos = index
        return self._process_frame(raw)

    def get_next_frame(self):
        raw = self._source.get_next_frame()
        self._frame_pos += 1
        return self._process_frame(raw)

    def close(self, propagate=True):
        """closes the source; with `propagate` the whole chain of filters below it"""
        if propagate and isinstance(self._source, VideoFilterBase):
            self._source.close(propagate=True)
        else:
            self._source.close()


class VideoSlice(VideoFilterBase):
    """`video[start:stop:step]` (reference: video/io/base.py:392-474).  The selection is a Python
    `range` over the source's frame numbers: frame k of the slice is frame `self._frames[k]` there."""

    def __init__(self, source, start=0, stop=None, step=1, _normalized=False):
        if step == 0:
            raise ValueError("step argument must not be zero.")
        if _normalized:             # start/stop/step come from slice.indices(): use them verbatim
            self._frames = range(start, stop, step)         # (stop == -1 with step < 0: down to frame 0)
        else:
            count = source.frame_count
            self._frames = range(_wrap_index(start, count), count if stop is None else _wrap_index(stop, count), step)
        if len(self._frames):
            source.set_frame_pos(self._frames[0])
        super(VideoSlice, self).__init__(source, frame_count=len(self._frames))
        if step < 0:
            logger.warning("Reversing a video can slow down the processing significantly.")

    def _source_index(self, index):
        index = _wrap_index(index, self.frame_count)
        if not 0 <= index < self.frame_count:
            raise IndexError("Cannot access frame %d in video of length %d" % (index, self.frame_count))
        return index, self._frames[index]

    def get_frame_pos(self):
        return self._frame_pos

    def set_frame_pos(self, index):
        if self.frame_count == 0 and index == 0:      # rewinding an empty slice is a no-op
            self._frame_pos = 0
            return
        self._frame_pos, there = self._source_index(index)
        self._source.set_frame_pos(there)

    def get_frame(self, index):
        _, there = self._source_index(index)
        return self._process_frame(self._source.get_frame(there))

    def get_next_frame(self):
        if self._frame_pos >= self.frame_count:
            self.abort_iteration()
            raise StopIteration
        if self._frames.step == 1:                    # consecutive frames: let the source stream
            raw = self._source.get_next_frame()
        else:
            raw = self._source.get_frame(self._frames[self._frame_pos])
        self._frame_pos += 1
        return self._process_frame(raw)


class _VideoForkClient(VideoBase):
    """one reader of a VideoFork (reference: video/io/base.py:478-513).  It holds no frames: every
    read goes to the fork, which knows which of its readers has taken the frame it holds."""

    def __init__(self, video_fork):
        self._parent = video_fork
        super(_VideoForkClient, self).__init__(**video_fork.video_format)

    def get_next_frame(self):
        frame = self._parent._serve(self, self._frame_pos)      # StopIteration past the last frame
        self._frame_pos += 1
        return frame

    def get_frame(self, index):
        raise NotSeekableError("clients of a VideoFork read forward only (the fork holds one frame)")

    def set_frame_pos(self, index):
        self._frame_pos = index + self.frame_count if index < 0 else index

    def abort_iteration(self):
        self._parent.abort_iteration()
        super(_VideoForkClient, self).abort_iteration()

    def close(self):
        """a reader that closes ends the iteration of every other reader (SystemExit there)"""
        self._parent.abort_iteration()


class VideoFork(VideoFilterBase):
    """One video, several readers in lock step (reference: video/io/base.py:516-662):

        fork = VideoFork(video)
        a, b = FilterBlur(fork.get_client(), 2), FilterCrop(fork.get_client(), region='left')
        for frame_a, frame_b in zip(a, b):
            ...

    The fork pulls every frame from its source once and HOLDS it until a reader asks for the next
    one.  Book-keeping is per reader -- the set of readers that have taken the held frame -- so
    `synchronized` means: the next frame is pulled only once every reader has had the held one
    (SynchronizationError for a reader that runs ahead), and a reader that asks twice for the same
    frame is still one reader.  Any request for a frame other than the held one or its successor is
    a SynchronizationError: the fork cannot go back and will not skip.  The GPU counterpart of this
    1 -> N fan-out -- one upload, several device-side consumers -- is `video.streaming`."""

    def __init__(self, source, synchronized=True, client_count=None):
        self.synchronized = bool(synchronized)
        self._fixed_clients = client_count      # None: as many readers as register
        self._clients = []
        self._held = None                       # frame most recently pulled from the source
        self._held_index = -1                   # its index; the successor is _held_index + 1
        self._taken_by = None                   # ids of the readers that have the held frame (None: nothing held)
        self._exhausted = False                 # the source ended: the successor does not exist
        self._aborted = False
        super(VideoFork, self).__init__(source)

    # -- what the reference exposes -------------------------------------------------------------
    @property
    def client_count(self):
        return len(self._clients) if self._fixed_clients is None else self._fixed_clients

    @property
    def state(self):
        return "aborting" if self._aborted else "normal"

    def get_client(self):
        """a new reader; iterate it (or a filter chain on top of it)"""
        if self._fixed_clients is not None and len(self._clients) >= self._fixed_clients:
            raise ValueError("this fork was created for %d readers and has them all" % self._fixed_clients)
        reader = _VideoForkClient(self)
        self._clients.append(reader)
        self._taken_by = None                   # a late joiner must not block (or be blocked by) the held frame
        return reader

    def __iter__(self):
        raise RuntimeError("a VideoFork is not iterated itself: iterate the readers from get_client()")

    # -- serving ---------------------------------------------------------------------------------
    def _pull(self):
        try:
            self._held = self.get_next_frame()          # (listeners of the fork see every frame once)
            self._exhausted = False
        except (StopIteration, IndexError):
            self._held, self._exhausted = None, True
        self._held_index += 1
        self._taken_by = set()

    def _serve(self, reader, index):
        if self._aborted:
            raise SystemExit("the iteration over this VideoFork was aborted by one of its readers")
        if index < 0:
            index += self.frame_count
        ahead = index - self._held_index
        if ahead == 1:
            waiting = self.synchronized and self._taken_by is not None and \
                len(self._taken_by) < self.client_count
            if waiting:
                raise SynchronizationError("frame %d has not been read by all %d readers yet; frame %d "
                                           "cannot be pulled" % (self._held_index, self.client_count, index))
            self._pull()
        elif ahead != 0:
            raise SynchronizationError("the fork holds frame %d and can only move on to frame %d, but frame "
                                       "%d was requested: its readers are out of step"
                                       % (self._held_index, self._held_index + 1, index))
        if self._taken_by is not None:
            self._taken_by.add(id(reader))
        if self._exhausted:
            raise StopIteration
        return self._held

    def get_frame(self, index):
        """frame `index` for an anonymous reader: the held frame or its successor (reference API)"""
        try:
            return self._serve(None, index)
        except StopIteration:
            return StopIteration

    def set_frame_pos(self, index):
        """moves the fork and all of its readers to frame `index`"""
        if index < 0:
            index += self.frame_count
        super(VideoFork, self).set_frame_pos(index)
        for reader in self._clients:
            reader.set_frame_pos(index)
        self._held, self._held_index, self._taken_by, self._exhausted = None, index - 1, None, False

    def clear(self):
        """forgets all readers (they may still read the held frame)"""
        self._clients = []

    def abort_iteration(self):
        """every reader's next read raises SystemExit"""
        self._aborted = True
        super(VideoFork, self).abort_iteration()
