"""Overlapped preprocessing: `VideoPreprocessor` (reference: video/io/parallel.py:386-488).

The reference reads the video in one thread and applies each preprocessing function in a thread
of its own, handing the consumer one dict per frame::

    for data in VideoPreprocessor(video, {'blur': blur_frame}):
        frame_raw, frame_blurred = data['raw'], data['blur']

Two forms exist here:

* `VideoPreprocessor(video, functions)` -- the same contract for arbitrary host callables, one
  worker thread per function plus one for the reader (ctypes releases the GIL during calls into
  libvideoanalysis_hip.so, which is what the reference asks of its callees, :398-400; the C ABI's
  stand-alone entry points keep no state between calls, so concurrent workers are safe).
* `VideoAnalysisPreprocessor(video, ...)` -- the dict-per-frame contract with the fused GPU chain
  underneath: frames are pulled in batches and run through one `FrameEngine` with the upload, the
  chain and the download of consecutive batches overlapped on three HIP streams
  (`video.streaming.StreamedEngine`), yielding ``{'raw', 'blur', 'mask', 'labels', 'count',
  'stats'}`` per frame.  The worker threads of the reference become streams.

The reader process / shared-memory pipe of the reference (:39-345) exists to keep a decoder off
the consumer's core; codec I/O is out of scope here.
"""
import queue
import threading

import numpy as np


class WorkerThread(object):
    """calls `function` in a background thread: `put(args)` starts a call, `get()` returns its
    result (re-raising its exception); with use_threads=False the call runs inside `get()`"""

    def __init__(self, function, use_threads=True):
        self.function = function
        self.use_threads = use_threads
        self._args = None
        if use_threads:
            self._in, self._out = queue.Queue(1), queue.Queue(1)
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    def _loop(self):
        while True:
            args = self._in.get()
            if args is None:
                return
            try:
                self._out.put((True, self.function(*args)))
            except BaseException as err:      # handed to the consumer in get()
                self._out.put((False, err))

    def put(self, *args):
        if self.use_threads:
            self._in.put(args)
        else:
            self._args = args

    def get(self):
        if not self.use_threads:
            return self.function(*self._args)
        ok, value = self._out.get()
        if not ok:
            raise value
        return value

    def close(self):
        if self.use_threads and self._thread.is_alive():
            self._in.put(None)


class VideoPreprocessor(object):
    """reads `video` in a background thread and applies every function of the dict `functions` to
    each frame in a thread of its own; iterating yields {'raw': frame, name: function(frame), ...}
    (reference: video/io/parallel.py:386-488)"""

    def __init__(self, video, functions, preprocess=None, use_threads=True):
        if "raw" in functions:
            raise KeyError("The key `raw` is reserved for the raw _frame and may not be used for functions.")
        self.length = len(video)
        self.video_iter = iter(video)
        self.functions = functions
        self.preprocess = preprocess
        self._frame = None
        self._worker_next_frame = WorkerThread(self._get_next_frame, use_threads=use_threads)
        self._workers = {name: WorkerThread(func, use_threads=use_threads)
                         for name, func in self.functions.items()}
        self._init_next_processing(self._get_next_frame())

    def __len__(self):
        return self.length

    def _get_next_frame(self):
        try:
            frame = next(self.video_iter)
        except StopIteration:
            return None
        frame = np.array(frame)               # sources may hand out views of a shared buffer
        return self.preprocess(frame) if self.preprocess else frame

    def _init_next_processing(self, frame_next):
        self._frame = frame_next
        if frame_next is None:
            return
        for worker in self._workers.values():
            worker.put(frame_next)
        self._worker_next_frame.put()

    def __iter__(self):
        return self

    def __next__(self):
        if self._frame is None:
            self.close()
            raise StopIteration
        result = {name: worker.get() for name, worker in self._workers.items()}
        result["raw"] = self._frame
        self._init_next_processing(self._worker_next_frame.get())
        return result

    next = __next__

    def close(self):
        for worker in list(self._workers.values()) + [self._worker_next_frame]:
            worker.close()


class VideoAnalysisPreprocessor(object):
    """dict-per-frame iteration with the fused GPU chain underneath.

    outputs: any of 'blur' (blurred, background-subtracted frame), 'mask', 'labels', 'count',
    'stats'; 'raw' is always there.  Engine parameters as for `FilterAnalysisChain`."""

    _ENGINE_NAME = {"blur": "filtered", "mask": "mask", "labels": "labels", "count": "counts", "stats": "stats"}

    def __init__(self, video, outputs=("blur", "mask", "labels", "count"), background="mean", rate=0.02,
                 sigma=5.0, threshold=20, morphology=(("dilate", "rect", 5), ("erode", "rect", 5)),
                 connectivity=4, max_labels=0, batch=32, slots=3):
        from ..engine import FrameEngine
        from ..streaming import StreamedEngine
        unknown = set(outputs) - set(self._ENGINE_NAME)
        if unknown:
            raise KeyError("unknown outputs %r (have %r)" % (sorted(unknown), sorted(self._ENGINE_NAME)))
        if video.is_color:
            raise ValueError("VideoAnalysisPreprocessor expects a monochrome video")
        if "stats" in outputs and max_labels <= 0:
            raise ValueError("'stats' needs max_labels > 0")
        self.length = len(video)
        self.outputs = tuple(outputs)
        self.batch = int(batch)
        self._video_iter = iter(video)
        self._engine = FrameEngine(size=video.size, max_batch=self.batch, background=background, bg_rate=rate,
                                   sigma=sigma, thresh=threshold, morphology=morphology,
                                   connectivity=connectivity, max_labels=max_labels)
        want = tuple(sorted({self._ENGINE_NAME[o] for o in self.outputs}))
        self._stream = StreamedEngine(self._engine, want=want, slots=slots)
        self._ready = []                      # per-frame dicts waiting to be handed out
        self._exhausted = False

    def __len__(self):
        return self.length

    def _split(self, results):
        for res in results:
            raw = res["tag"]
            for i in range(len(raw)):
                d = {"raw": raw[i]}
                for o in self.outputs:
                    v = res[self._ENGINE_NAME[o]][i]
                    d[o] = int(v) if o == "count" else v
                self._ready.append(d)

    def _fill(self):
        """keeps batches in flight until per-frame results are ready (or the video ends)"""
        while not self._ready and not self._exhausted:
            frames = []
            for frame in self._video_iter:
                frames.append(np.array(frame))
                if len(frames) == self.batch:
                    break
            if frames:
                block = np.stack(frames)
                if block.dtype != np.uint8:
                    raise TypeError("VideoAnalysisPreprocessor expects uint8 frames")
                self._split(self._stream.submit(block, tag=block))
            if len(frames) < self.batch:
                self._exhausted = True
                self._split(self._stream.drain())

    def __iter__(self):
        return self

    def __next__(self):
        self._fill()
        if not self._ready:
            self.close()
            raise StopIteration
        return self._ready.pop(0)

    next = __next__

    def close(self):
        if self._stream is not None:
            self._stream.close()
            self._engine.close()
            self._stream = None
