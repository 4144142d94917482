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


class VideoBase(object):
    """A sequence of equally sized frames with a cursor (`frame_pos`) and listeners.

    Listeners are callables invoked with every frame the video hands out
    (reference: video/io/base.py:137-145, 182-189)."""

    write_access = False
    seekable = False

    def __init__(self, size=(0, 0), frame_count=-1, fps=None, is_color=True):
        if len(size) != 2:
            raise ValueError("Videos must have two spatial dimensions.")
        self.size = tuple(size)                  # (width, height)
        self.frame_count = frame_count
        self.fps = 25 if fps is None else fps
        self.is_color = is_color
        self._listeners = []
        self._frame_pos = 0                      # index of the frame handed out next

    # ------------------------------------------------------------------ description
    def get_property_list(self):
        return ("size=(%d, %d)" % tuple(self.size), "frame_count=%s" % self.frame_count,
                "fps=%s" % self.fps, "is_color=%s" % self.is_color)

    def _listener_suffix(self):
        k = len(self._listeners)
        return "" if k == 0 else ("[1 listener]" if k == 1 else "[%d listeners]" % k)

    def __str__(self):
        return "%s(%s)%s" % (type(self).__name__, ", ".join(self.get_property_list()),
                             self._listener_suffix())

    def info(self):
        return "Video(%s)" % ", ".join(self.get_property_list())

    # ------------------------------------------------------------------ geometry
    def __len__(self):
        return self.frame_count

    @property
    def width(self):
        return self.size[0]

    @property
    def height(self):
        return self.size[1]

    @property
    def bounds(self):
        return (0, 0, self.width, self.height)

    @property
    def shape(self):
        """shape of the whole video as an array: (frames, height, width[, 3])"""
        shape = (self.frame_count, self.size[1], self.size[0])
        return shape + (3,) if self.is_color else shape

    @property
    def video_format(self):
        return {"size": self.size, "frame_count": self.frame_count, "fps": self.fps,
                "is_color": self.is_color}

    # ------------------------------------------------------------------ listeners
    def register_listener(self, listener_callback):
        self._listeners.append(listener_callback)

    def unregister_listener(self, listener_callback):
        self._listeners.remove(listener_callback)

    def _process_frame(self, frame):
        """hook every frame passes through on its way out: notifies the listeners"""
        for listener in self._listeners:
            listener(frame)
        return frame

    # ------------------------------------------------------------------ cursor
    def get_frame_pos(self):
        return self._frame_pos

    def set_frame_pos(self, index):
        if index < 0:
            index += self.frame_count
        if self.seekable:
            if not 0 <= index < self.frame_count:
                raise IndexError("Seeking to frame %d was not possible." % index)
            self._frame_pos = index
        elif index >= self.get_frame_pos():
            for _ in range(self.get_frame_pos(), index):     # fast-forward
                self.get_next_frame()
        else:
            raise NotSeekableError("Cannot seek to frame %d, because the video is already at "
                                   "frame %d" % (index, self.get_frame_pos()))

    def rewind(self):
        self.set_frame_pos(0)

    # ------------------------------------------------------------------ frame access
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
        if isinstance(key, slice):
            return VideoSlice(self, *key.indices(self.frame_count), _normalized=True)
        if isinstance(key, (int, np.integer)):
            return self.get_frame(int(key))
        raise TypeError("Invalid key `%r` for indexing" % (key,))

    def __setitem__(self, key, value):
        raise ValueError("Writing to this video stream is prohibited.")

    def copy(self, dtype=np.uint8, disp=False):
        """materialise the video as a VideoMemory (base.py:248-269)"""
        from .memory import VideoMemory
        data = np.empty(self.shape, dtype)
        for k, frame in enumerate(self):
            data[k, ...] = frame
        return VideoMemory(data, fps=self.fps, copy_data=False)


class VideoIterator(object):
    """iterator protocol for videos; rewinds the video when created (base.py:273-283)"""

    def __init__(self, video):
        self._video = video
        self._video.rewind()

    def __iter__(self):
        return self

    def __next__(self):
        try:
            return self._video.get_next_frame()
        except IndexError:
            raise StopIteration

    next = __next__         # Python-2 spelling used by reference-era callers


class VideoFilterBase(VideoBase):
    """A view on another video that transforms each frame in `_process_frame`
    (reference: video/io/base.py:313-388).  Format properties default to the source's."""

    def __init__(self, source, size=None, frame_count=None, fps=None, is_color=None):
        self._source = source
        super(VideoFilterBase, self).__init__(
            size=source.size if size is None else size,
            frame_count=source.frame_count if frame_count is None else frame_count,
            fps=source.fps if fps is None else fps,
            is_color=source.is_color if is_color is None else is_color)

    def __str__(self):
        return "%s +%s%s" % (self._source, type(self).__name__, self._listener_suffix())

    @property
    def seekable(self):
        return self._source.seekable

    def abort_iteration(self):
        self._source.abort_iteration()

    def set_frame_pos(self, index):
        if index < 0:
            index += self.frame_count
        self._source.set_frame_pos(index)
        self._frame_pos = index

    def get_frame_pos(self):
        return self._source.get_frame_pos()

    def get_frame(self, index):
        if index < 0:
            index += self.frame_count
        frame = self._source.get_frame(index)
        self._frame_pos = index
        return self._process_frame(frame)

    def get_next_frame(self):
        frame = self._source.get_next_frame()
        self._frame_pos += 1
        return self._process_frame(frame)

    def close(self, propagate=True):
        if propagate and isinstance(self._source, VideoFilterBase):
            self._source.close(propagate=True)
        else:
            self._source.close()


class VideoSlice(VideoFilterBase):
    """video[start:stop:step] (reference: video/io/base.py:392-474)"""

    def __init__(self, source, start=0, stop=None, step=1, _normalized=False):
        count = source.frame_count
        if _normalized:             # start/stop come from slice.indices(): use them verbatim
            self._start, self._stop = start, stop   # (stop == -1 with step < 0: down to frame 0)
        else:
            self._start = start if start >= 0 else count + start
            if stop is None:
                self._stop = count
            else:
                self._stop = stop if stop >= 0 else count + stop
        if step == 0:
            raise ValueError("step argument must not be zero.")
        self._step = step
        frame_count = max(0, int(np.ceil((self._stop - self._start) / self._step)))
        if frame_count > 0:
            source.set_frame_pos(self._start)
        super(VideoSlice, self).__init__(source, frame_count=frame_count)
        if step < 0:
            logger.warning("Reversing a video can slow down the processing significantly.")

    def _check(self, index):
        if index < 0:
            index += self.frame_count
        if not 0 <= index < self.frame_count:
            raise IndexError("Cannot access frame %d in video of length %d"
                             % (index, self.frame_count))
        return index

    def set_frame_pos(self, index):
        if self.frame_count == 0 and index == 0:      # rewinding an empty slice is a no-op
            self._frame_pos = 0
            return
        index = self._check(index)
        self._source.set_frame_pos(self._start + index * self._step)
        self._frame_pos = index

    def get_frame_pos(self):
        return self._frame_pos

    def get_frame(self, index):
        index = self._check(index)
        return self._process_frame(self._source.get_frame(self._start + index * self._step))

    def get_next_frame(self):
        if self._frame_pos >= self.frame_count:
            self.abort_iteration()
            raise StopIteration
        if self._step == 1:
            frame = self._source.get_next_frame()
        else:
            frame = self._source.get_frame(self._start + self._frame_pos * self._step)
        self._frame_pos += 1
        return self._process_frame(frame)


class _VideoForkClient(VideoBase):
    """one consumer of a VideoFork: iterating it asks the fork for frame `_frame_pos`
    (reference: video/io/base.py:478-513)"""

    def __init__(self, video_fork):
        self._parent = video_fork
        super(_VideoForkClient, self).__init__(**video_fork.video_format)

    def get_next_frame(self):
        frame = self._parent.get_frame(self._frame_pos)
        self._frame_pos += 1
        if frame is StopIteration:
            raise StopIteration
        return frame

    def set_frame_pos(self, index):
        self._frame_pos = index + self.frame_count if index < 0 else index

    def abort_iteration(self):
        self._parent.abort_iteration()
        super(_VideoForkClient, self).abort_iteration()

    def close(self):
        """asks the fork to send SystemExit to all clients"""
        self._parent.abort_iteration()


class VideoFork(VideoFilterBase):
    """hands the frames of one video to several consumers that are iterated in lock step
    (reference: video/io/base.py:516-662):

        fork = VideoFork(video)
        a, b = FilterBlur(fork.get_client(), 2), FilterCrop(fork.get_client(), region='left')
        for frame_a, frame_b in zip(a, b):
            ...

    Every frame is pulled from the source once and cached until the next one is asked for.  With
    `synchronized` a client that runs ahead of the others raises SynchronizationError (the test
    counts how often the cached frame has been handed out); a client asking for any frame but
    the cached or the next one always does."""

    def __init__(self, source, synchronized=True, client_count=None):
        self.synchronized = synchronized
        self._client_count = client_count
        self._clients = []
        self._frame = None
        self._frame_index = -1
        self._retrieve_count = np.inf          # how often the cached frame has been handed out
        self.state = "normal"
        super(VideoFork, self).__init__(source)

    @property
    def client_count(self):
        return len(self._clients) if self._client_count is None else self._client_count

    def set_frame_pos(self, index):
        """positions the fork and all of its clients"""
        if index < 0:
            index += self.frame_count
        super(VideoFork, self).set_frame_pos(index)
        for client in self._clients:
            client.set_frame_pos(index)
        self._frame = None
        self._frame_index = index - 1

    def get_frame(self, index):
        """frame `index` for a client: the cached frame, or the next one of the source"""
        if self.state == "aborting":
            raise SystemExit("Another client of the VideoFork requested to abort the iteration.")
        if index < 0:
            index += self.frame_count
        if index == self._frame_index:
            self._retrieve_count += 1
        elif index == self._frame_index + 1:
            if self.synchronized and self._retrieve_count < self.client_count:
                raise SynchronizationError("The other clients have not yet read the previous frame.")
            self._frame_index = index
            try:
                self._frame = self.get_next_frame()
            except (StopIteration, IndexError):
                self._frame = StopIteration
            self._retrieve_count = 1
        else:
            raise SynchronizationError("The clients of the video fork ran out of sync. The parent "
                                       "process is at frame %d, while one client requested frame %d"
                                       % (self._frame_index, index))
        return self._frame

    def clear(self):
        """ends the iteration and forgets all clients (they may still read the cached frame)"""
        self._clients = []

    def abort_iteration(self):
        """sends SystemExit to all other clients"""
        self.state = "aborting"
        super(VideoFork, self).abort_iteration()

    def __iter__(self):
        raise RuntimeError("Cannot iterate over a VideoFork. Use the get_client() method to get an "
                           "iterable client.")

    def get_client(self):
        """a new client that can be iterated"""
        if self._client_count is not None and len(self._clients) >= self._client_count:
            raise ValueError("We already registered %d clients." % self._client_count)
        client = _VideoForkClient(self)
        self._clients.append(client)
        self._retrieve_count = np.inf
        return client
