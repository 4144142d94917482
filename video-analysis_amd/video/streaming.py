"""StreamedEngine -- overlap of frame upload, the GPU chain and result download.

The reference overlaps frame production with processing through a reader process feeding a
shared-memory buffer and one worker thread per preprocessing function
(`video/io/parallel.py:345-488`, results as a dict per frame).  Here the same role is played by
three HIP streams and pinned host buffers: while batch k runs through the chain, batch k+1 is
uploaded and the results of batch k-1 are downloaded.  Results come back per batch as a dict
(`'counts'`, `'stats'`, optionally `'mask'` / `'labels'` / `'filtered'`), like the dict of
`VideoPreprocessor`.

    eng = FrameEngine(...)
    with StreamedEngine(eng, want=("counts", "stats"), slots=3) as s:
        for batch in batches:
            for result in s.submit(batch):       # yields results of EARLIER batches when ready
                ...
        for result in s.drain():
            ...
"""
import ctypes as C

import numpy as np

from . import _hip
from ._hip import check


class _Pinned(object):
    """page-locked host buffer exposed as a NumPy array"""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._ptr = C.c_void_p()
        check(_hip.lib().va_host_alloc(C.byref(self._ptr), max(self.nbytes, 1)))
        buf = (C.c_char * max(self.nbytes, 1)).from_address(self._ptr.value)
        self.array = np.frombuffer(buf, self.dtype, int(np.prod(self.shape))).reshape(self.shape)

    @property
    def ptr(self):
        return self._ptr.value

    def free(self):
        if self._ptr is not None and self._ptr.value:
            self.array = None
            _hip.load_library().va_host_free(self._ptr)
            self._ptr = C.c_void_p()


class StreamedEngine(object):

    OUTPUTS = ("filtered", "mask", "labels", "counts", "stats")

    def __init__(self, engine, want=("counts",), slots=3, copy_results=True):
        """copy_results=False hands out views of the pinned result buffers: zero-copy, valid
        until `slots` more batches have been submitted"""
        self.engine = engine
        self.copy_results = bool(copy_results)
        self.want = tuple(want)
        unknown = set(self.want) - set(self.OUTPUTS)
        if unknown:
            raise ValueError("unknown outputs %r" % sorted(unknown))
        if slots < 2:
            raise ValueError("need at least 2 slots to overlap anything")
        self._lib = _hip.lib()
        e = engine
        n, h, w = e.max_batch, e.height, e.width
        fshape = (n,) + e.frame_shape
        self._shapes = {"filtered": (fshape, e.dtype), "mask": ((n, h, w), np.uint8),
                        "labels": ((n, h, w), np.int32), "counts": ((n,), np.int32),
                        "stats": ((n, max(e.max_labels, 1), _hip.STATS_STRIDE), np.int64)}
        self._streams = {}
        for name in ("h2d", "compute", "d2h"):
            s = C.c_void_p()
            check(self._lib.va_stream_create(C.byref(s)))
            self._streams[name] = s
        self._slots = []
        for _ in range(slots):
            slot = {"in_host": _Pinned(fshape, e.dtype), "in_dev": _hip.DeviceBuffer(int(np.prod(fshape)) * e.dtype.itemsize),
                    "dev": {}, "host": {}, "events": {}, "n": 0, "busy": False, "tag": None}
            need = set(self.want)
            if "stats" in need:
                need.add("counts")
            for k in need:
                shape, dt = self._shapes[k]
                slot["dev"][k] = _hip.DeviceBuffer(int(np.prod(shape)) * np.dtype(dt).itemsize)
                slot["host"][k] = _Pinned(shape, dt)
            for k in ("uploaded", "computed", "downloaded"):
                ev = C.c_void_p()
                check(self._lib.va_event_create(C.byref(ev)))
                slot["events"][k] = ev
            self._slots.append(slot)
        self._next = 0
        self._pending = []            # slot indices in submission order

    # ------------------------------------------------------------------ pipeline
    def _collect(self, idx):
        slot = self._slots[idx]
        check(self._lib.va_event_sync(slot["events"]["downloaded"]))
        n = slot["n"]
        out = {k: (np.array(slot["host"][k].array[:n]) if self.copy_results
                   else slot["host"][k].array[:n]) for k in self.want}
        out["tag"] = slot["tag"]
        slot["busy"] = False
        return out

    def _retire_for(self, idx):
        done = []
        if self._slots[idx]["busy"]:
            while self._pending:                   # retire in order up to and including idx
                j = self._pending.pop(0)
                done.append(self._collect(j))
                if j == idx:
                    break
        return done

    def input_buffer(self):
        """(pinned (max_batch, H, W[, C]) array of the NEXT slot, results retired to free it).
        A producer (decoder, camera) can write frames straight into it and then call
        `submit(None, n=...)`: no host-side copy at all."""
        done = self._retire_for(self._next)
        return self._slots[self._next]["in_host"].array, done

    def submit(self, frames, tag=None, n=None):
        """enqueue one batch (n <= engine.max_batch frames); returns the list of results of
        earlier batches that had to be retired to make room (possibly empty).
        frames=None: the batch was written into `input_buffer()` already (give `n`)."""
        e = self.engine
        idx = self._next
        if frames is None:
            if n is None or not 0 < n <= e.max_batch:
                raise ValueError("submit(None) needs the number of frames n")
            done = self._retire_for(idx)
        else:
            frames = np.asarray(frames)
            n = frames.shape[0]
            if frames.shape[1:] != e.frame_shape or n > e.max_batch:
                raise ValueError("batch of shape %r does not fit the engine" % (frames.shape,))
            done = self._retire_for(idx)
            self._slots[idx]["in_host"].array[:n] = frames   # the one host copy (pageable -> pinned)
        self._next = (self._next + 1) % len(self._slots)
        slot = self._slots[idx]
        nbytes = n * int(np.prod(e.frame_shape)) * e.dtype.itemsize
        L, st, ev = self._lib, self._streams, slot["events"]
        check(L.va_memcpy_h2d(slot["in_dev"].ptr, slot["in_host"].ptr, nbytes, st["h2d"]))
        check(L.va_event_record(ev["uploaded"], st["h2d"]))
        check(L.va_stream_wait_event(st["compute"], ev["uploaded"]))
        g = lambda k: slot["dev"][k].ptr if k in slot["dev"] else None
        e.run_device(slot["in_dev"].ptr, n, g("filtered"), g("mask"), g("labels"), g("counts"),
                     g("stats"), st["compute"])
        check(L.va_event_record(ev["computed"], st["compute"]))
        check(L.va_stream_wait_event(st["d2h"], ev["computed"]))
        for k in self.want:
            shape, dt = self._shapes[k]
            per_frame = int(np.prod(shape[1:])) * np.dtype(dt).itemsize
            check(L.va_memcpy_d2h(slot["host"][k].ptr, slot["dev"][k].ptr, n * per_frame, st["d2h"]))
        check(L.va_event_record(ev["downloaded"], st["d2h"]))
        slot["n"], slot["busy"], slot["tag"] = n, True, tag
        self._pending.append(idx)
        return done

    def drain(self):
        """wait for and return the results of every batch still in flight, in order"""
        out = [self._collect(j) for j in self._pending]
        self._pending = []
        return out

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if not self._slots:
            return
        self.drain()
        for slot in self._slots:
            slot["in_host"].free()
            slot["in_dev"].free()
            for b in list(slot["dev"].values()) + list(slot["host"].values()):
                b.free()
            for ev in slot["events"].values():
                self._lib.va_event_destroy(ev)
        for s in self._streams.values():
            self._lib.va_stream_destroy(s)
        self._slots = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
