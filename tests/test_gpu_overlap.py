"""GPU: overlapped runs (va_pipeline_overlap) -- the label-image write of batch k on the pipeline's
own stream beside the stages of batch k + 1 -- against the oracle, batch by batch.  Consecutive
batches differ in content, so a paint pass that read the next batch's masks or run tables, or a
labelling stage that wrote into a label image still being painted, shows up as a wrong label map."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _clip(n, h, w, seed, nblobs=7, salt=0.002):
    rng = np.random.default_rng(seed)
    bg = np.clip(rng.normal(100, 10, (h, w)), 0, 255)
    yy, xx = np.mgrid[:h, :w]
    pos = rng.uniform(0, 1, (nblobs, 2)) * (w, h)
    vel = rng.uniform(-4, 4, (nblobs, 2))
    rad = rng.uniform(min(h, w) / 30 + 2, min(h, w) / 7 + 3, nblobs)
    out = np.empty((n, h, w), np.uint8)
    for t in range(n):
        f = bg + rng.normal(0, 4, (h, w))
        for (cx, cy), r in zip(pos + vel * t, rad):
            f[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] += 60
        f[rng.random((h, w)) < salt] = 255
        out[t] = np.clip(f, 0, 255).astype(np.uint8)
    return out


@pytest.fixture()
def hip():
    from video import _hip
    _hip.lib()
    yield _hip
    _hip.lib().va_test_hook_labelling(0, 0)


@pytest.mark.parametrize("path,lds_runs", [(0, 0), (1, 0), (2, 0), (2, 40), (3, 0)])
@pytest.mark.parametrize("label_buffers", [1, 2])
def test_overlapped_batches_match_the_oracle(hip, oracle, path, lds_runs, label_buffers):
    """5 batches x 6 frames of different content through one overlapped pipeline: every labelling
    path (library's choice, chip-wide passes, per-frame kernel with run tables, its large-frame mode
    with the forest in the label image, sparse hand-off), one label buffer (the labelling stage must
    wait for the previous paint) and two alternating ones (it must not need to)."""
    from video._hip import DeviceBuffer, check
    from video.engine import FrameEngine
    L = hip.lib()
    check(L.va_test_hook_labelling(path, lds_runs))
    nb, n, h, w, ml = 5, 6, 120, 208, 64
    clip = _clip(nb * n, h, w, seed=11 + path)
    eng = FrameEngine(size=(w, h), max_batch=n, background="mean", sigma=2.0, thresh=20,
                      morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4, max_labels=ml)
    eng.overlap(True)
    src = [DeviceBuffer.from_array(clip[b * n:(b + 1) * n]) for b in range(nb)]
    lab = [DeviceBuffer(n * h * w * 4) for _ in range(label_buffers)]
    stats = [DeviceBuffer(n * ml * 16 * 8) for _ in range(label_buffers)]
    cnt = [DeviceBuffer(n * 4) for _ in range(nb)]
    keep_l = [DeviceBuffer(n * h * w * 4) for _ in range(nb)]
    keep_s = [DeviceBuffer(n * ml * 16 * 8) for _ in range(nb)]
    stream = C.c_void_p()
    check(L.va_stream_create(C.byref(stream)))
    for b in range(nb):
        j = b % label_buffers
        eng.run_device(src[b].ptr, n, None, None, lab[j].ptr, cnt[b].ptr, stats[j].ptr, stream)
        # the consumer of batch b: after the fence its labels are complete on the stream
        eng.fence(stream)
        check(L.va_memcpy_d2d(keep_l[b].ptr, lab[j].ptr, n * h * w * 4, stream))
        check(L.va_memcpy_d2d(keep_s[b].ptr, stats[j].ptr, n * ml * 16 * 8, stream))
    check(L.va_stream_sync(stream))
    state, n_seen = eng.get_background()
    mean, seen = None, 0
    for b in range(nb):
        mask, labels, counts, mean = oracle.chain_u8(clip[b * n:(b + 1) * n], 2.0, 20, morph_ksize=5,
                                                     connectivity=4, mean=mean, n_seen=seen)
        seen += n
        got_l = keep_l[b].download((n, h, w), np.int32)
        got_c = cnt[b].download((n,), np.int32)
        got_s = keep_s[b].download((n, ml, 16), np.int64)
        assert np.array_equal(got_c, counts), (b, got_c, counts)
        assert np.array_equal(got_l, labels), "batch %d" % b
        for f in range(n):
            k = min(int(counts[f]), ml)
            ref = oracle.region_stats(labels[f], int(counts[f]))[:k]
            assert np.array_equal(got_s[f, :k, :14], ref[:, :14]), (b, f)
    assert n_seen == nb * n and np.array_equal(state, mean)
    check(L.va_stream_destroy(stream))
    eng.close()


def test_overlap_switch_and_numpy_api(hip, oracle):
    """overlap on/off on one engine; the NumPy API fences by itself; counts-only runs do not paint"""
    from video.engine import FrameEngine
    n, h, w = 4, 96, 160
    clip = _clip(3 * n, h, w, seed=5)
    eng = FrameEngine(size=(w, h), max_batch=n, background="mean", sigma=2.0, thresh=20,
                      morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=8)
    mean, seen = None, 0
    for b, ov in enumerate((True, False, True)):
        eng.overlap(ov)
        out = eng.run(clip[b * n:(b + 1) * n], want=("labels", "counts") if b != 1 else ("counts",))
        mask, labels, counts, mean = oracle.chain_u8(clip[b * n:(b + 1) * n], 2.0, 20, morph_ksize=5,
                                                     connectivity=8, mean=mean, n_seen=seen)
        seen += n
        assert np.array_equal(out["counts"], counts)
        if "labels" in out:
            assert np.array_equal(out["labels"], labels)
    eng.close()
    plain = FrameEngine(size=(w, h), max_batch=n, sigma=2.0, thresh=20)
    with pytest.raises(Exception):
        plain.overlap(True)                      # no labelling stage, nothing to overlap
    plain.close()
