"""GPU: every BASELINE.json configuration at its full frame size, through `va_pipeline_run`.

The oracle cannot run a whole batch at these sizes in seconds, so each configuration is checked by
  * exact oracle comparisons on crops (background model and blur are local up to the blur radius;
    crops that touch the image border check the BORDER_REFLECT_101 handling on that side),
  * exact oracle comparisons of whole frames for the non-local stages (threshold, morphology,
    labelling) fed with the GPU's own blurred frames,
  * size-independent properties over the full batch.
cfg#2: 1080p, bg-sub + sigma=5 + threshold (no morphology, no labelling: the mask-only path)
cfg#4: 3840x2160, full chain, one rank's share of the batch (128 frames), both labelling paths
cfg#5: 1920x1080x3 float32, EMA background + sigma=9 blur, batch 256
plus a bounded, seeded slice of tools/stress_parity.py / tools/stress_f32.py.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(**kw):
    from video import _hip
    from video.engine import FrameEngine
    _hip.lib()                      # loud failure without the extension / a GPU
    return FrameEngine(**kw)


def _moving_blob_clip(n, h, w, seed, nblobs, salt, noise_tiles=8):
    """SURVEY.md 8(d) generator, cheap enough for full-size batches: static N(100,10) background,
    N(0,4) noise (a few tiles cycled), moving filled discs (+60), optional salt noise"""
    rng = np.random.default_rng(seed)
    base = np.clip(rng.normal(100, 10, (h, w)), 0, 255).astype(np.float32)
    noise = rng.normal(0, 4, (noise_tiles, h, w)).astype(np.float32)
    yy, xx = np.mgrid[:h, :w]
    cx0, cy0 = rng.uniform(0, w, nblobs), rng.uniform(0, h, nblobs)
    vx, vy = rng.uniform(-3, 3, nblobs), rng.uniform(-3, 3, nblobs)
    rad = rng.uniform(8, 60, nblobs)
    clip = np.empty((n, h, w), np.uint8)
    for t in range(n):
        f = base + noise[t % noise_tiles]
        for k in range(nblobs):
            x0, y0, r = int(cx0[k] + vx[k] * t), int(cy0[k] + vy[k] * t), int(rad[k])
            ya, yb, xa, xb = max(0, y0 - r), min(h, y0 + r + 1), max(0, x0 - r), min(w, x0 + r + 1)
            if ya < yb and xa < xb:
                sub = f[ya:yb, xa:xb]
                sub[(xx[ya:yb, xa:xb] - x0) ** 2 + (yy[ya:yb, xa:xb] - y0) ** 2 <= r * r] += 60
        if salt:
            sy = rng.integers(0, h, int(salt * h * w))
            sx = rng.integers(0, w, int(salt * h * w))
            f[sy, sx] = 255
        clip[t] = np.clip(f, 0, 255).astype(np.uint8)
    return clip


def _check_bg_blur_crops(oracle, clip, filtered, state, sigma, radius, size=96):
    """running mean + emitted difference + blur on crops, exact.  A crop's blur is exact wherever
    the window does not cross one of the crop's INNER edges; edges that coincide with the image
    border are exact as well (the oracle reflects there exactly like the full image)."""
    n, h, w = clip.shape
    for (ya, xa) in ((0, 0), (0, w - size), (h - size, 0), (h - size, w - size),
                     (h // 2 - 7, w // 2 + 5), (0, w // 3), (h - size, w // 3), (h // 3, 0), (h // 3, w - size)):
        crop = np.ascontiguousarray(clip[:, ya:ya + size, xa:xa + size])
        rd, rm = oracle.bg_mean_u8(crop)
        assert np.array_equal(state[ya:ya + size, xa:xa + size], rm), (ya, xa)
        blur = oracle.gaussian_u8(rd, sigma)
        y0 = 0 if ya == 0 else radius
        y1 = size if ya + size == h else size - radius
        x0 = 0 if xa == 0 else radius
        x1 = size if xa + size == w else size - radius
        assert np.array_equal(filtered[:, ya + y0:ya + y1, xa + x0:xa + x1], blur[:, y0:y1, x0:x1]), (ya, xa)


def test_cfg1_480p_literal_clip_through_the_filter_classes(oracle):
    """BASELINE.json configs[0], literally: seed 1, 64 frames of 640x480 uint8 (static N(100,10)
    background + N(0,4) noise + 3 moving discs, r = 12..25, +60), VideoMemory -> FilterBackground ->
    FilterBlur(sigma=2) -> FilterThreshold(20), read through the iterator protocol -- contracted into
    one engine and filter by filter -- against the oracle, frame for frame."""
    from video import filters as F
    from video.io.memory import VideoMemory
    rng = np.random.default_rng(1)
    n, h, w = 64, 480, 640
    base = np.clip(rng.normal(100, 10, (h, w)), 0, 255)
    yy, xx = np.mgrid[:h, :w]
    pos, vel, rad = rng.uniform(0, 1, (3, 2)) * (w, h), rng.uniform(-4, 4, (3, 2)), rng.uniform(12, 25, 3)
    clip = np.empty((n, h, w), np.uint8)
    for t in range(n):
        f = base + rng.normal(0, 4, (h, w))
        for (cx, cy), r in zip(pos + vel * t, rad):
            f[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] += 60
        clip[t] = np.clip(f, 0, 255).astype(np.uint8)
    diff, mean = oracle.bg_mean_u8(clip)
    ref = oracle.threshold_u8(oracle.gaussian_u8(diff, 2.0), 20)
    assert 0 < (ref != 0).mean() < 0.2
    for contract in (True, False):
        F._GpuStage.contract = contract
        try:
            bg = F.FilterBackground(VideoMemory(clip))
            video = F.FilterThreshold(F.FilterBlur(bg, 2), 20)
            assert len(video) == n and video.size == (w, h) and not video.is_color
            assert (video._runner() is not None) == contract
            for k, frame in enumerate(video):
                assert frame.dtype == np.uint8 and np.array_equal(frame, ref[k]), (contract, k)
            assert k == n - 1
            assert np.array_equal(video[17], ref[17]) and np.array_equal(video[-1], ref[-1])
            video.close()
        finally:
            F._GpuStage.contract = True


def test_cfg2_1080p_blur_threshold_mask_only(oracle):
    """BASELINE.json configs[1]: 1920x1080 uint8, batch 256, bg-sub + sigma=5 + threshold."""
    n, h, w = 256, 1080, 1920
    clip = _moving_blob_clip(n, h, w, seed=2, nblobs=40, salt=0.0)
    eng = _engine(size=(w, h), max_batch=n, background="mean", sigma=5.0, thresh=20)
    assert "mfma" in eng.description
    mask_only = eng.run(clip, want=("mask",))["mask"]          # the benchmarked path: no u8 blur output
    eng.set_background(None, 0)
    out = eng.run(clip, want=("mask", "filtered"))
    state, n_seen = eng.get_background()
    eng.close()
    assert n_seen == n
    assert np.array_equal(mask_only, out["mask"])
    assert np.array_equal(out["mask"], np.where(out["filtered"] > 20, 255, 0).astype(np.uint8))
    assert 0.001 < (mask_only != 0).mean() < 0.5
    _check_bg_blur_crops(oracle, clip, out["filtered"], state, 5.0, 15)


@pytest.mark.parametrize("shape,sigma,thresh,maxval", [((3, 97, 208), 5.0, 20, 255), ((2, 64, 64), 2.0, 100, 1),
                                                       ((1, 200, 528), 3.0, 0, 7), ((2, 33, 1936), 5.0, 254, 255),
                                                       ((1, 1080, 80), 1.0, 128, 200)])
def test_mask_only_chain_writes_bytes_from_the_gaussian(oracle, shape, sigma, thresh, maxval):
    """the chain that ends at the uint8 mask (no morphology, no labelling, no u8 blur requested):
    the matrix-core Gaussian's epilogue writes the 0 / maxval bytes itself"""
    rng = np.random.default_rng(shape[2] + thresh)
    clip = rng.integers(0, 256, shape, dtype=np.uint8)
    clip[0, :9, :] = 255
    clip[-1, :, -20:] = 0
    eng = _engine(size=(shape[2], shape[1]), max_batch=shape[0], sigma=sigma, thresh=thresh, maxval=maxval)
    assert "mfma" in eng.description
    eng.profile(True)
    got = eng.run(clip, want=("mask",))["mask"]
    stages = eng.stage_times()
    eng.close()
    assert "gauss_mfma_mask8" in stages and "mask_unpack" not in stages
    ref = oracle.threshold_u8(oracle.gaussian_u8(clip, sigma), thresh, maxval)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("shape,sigma,thresh", [((3, 97, 208), 6.0, 20), ((2, 64, 64), 8.0, 100),
                                                ((1, 200, 528), 10.0, 60), ((2, 70, 1936), 9.0, 127)])
def test_wide_sigma_chain_stays_on_the_single_launch_gaussian(oracle, shape, sigma, thresh):
    """sigma beyond the matrix-core kernel's radius 16: the chain keeps a single-launch Gaussian (dot4/dot2 with a
    32-column halo) that thresholds in its epilogue -- mask, blur and labels against the oracle"""
    rng = np.random.default_rng(shape[2] + thresh)
    clip = rng.integers(0, 256, shape, dtype=np.uint8)
    clip[0, :9, :] = 255
    clip[-1, :, -40:] = 0
    eng = _engine(size=(shape[2], shape[1]), max_batch=shape[0], sigma=sigma, thresh=thresh, connectivity=4,
                  max_labels=4096)
    assert "fused" in eng.description, eng.description
    out = eng.run(clip, want=("mask", "filtered", "labels", "counts"))
    eng.close()
    blur = oracle.gaussian_u8(clip, sigma)
    ref = oracle.threshold_u8(blur, thresh, 255)
    assert np.array_equal(out["filtered"], blur)
    assert np.array_equal(out["mask"], ref)
    for f in range(shape[0]):
        lab, cnt = oracle.label(ref[f], 4)
        assert cnt == out["counts"][f]
        assert np.array_equal(lab, out["labels"][f])


def test_cfg4_4k_full_chain_both_labelling_paths(oracle):
    """BASELINE.json configs[3]: 3840x2160 uint8 full chain; one rank's shard of the 1024-frame
    batch (128 frames) through va_pipeline_run, with the library's labelling choice (per-frame
    kernel, 4 rows per wave at 4K) and with the chip-wide passes (test hook path 1)."""
    from video import _hip
    n, h, w = 128, 2160, 3840
    clip = _moving_blob_clip(n, h, w, seed=4, nblobs=160, salt=0.002, noise_tiles=4)
    kw = dict(size=(w, h), max_batch=n, background="mean", sigma=5.0, thresh=20,
              morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4)
    eng = _engine(**kw)
    assert "mfma" in eng.description
    out = eng.run(clip, want=("filtered", "mask", "labels", "counts"))
    state, n_seen = eng.get_background()
    labels, mask, counts, filtered = out["labels"], out["mask"], out["counts"], out["filtered"]
    # -- properties over the whole shard
    assert n_seen == n and counts.min() >= 1
    assert np.array_equal(labels != 0, mask != 0)
    assert np.array_equal(labels.reshape(n, -1).max(1), counts)
    for f in range(0, n, 9):
        assert labels[f].ravel()[np.flatnonzero(labels[f].ravel())[0]] == 1      # raster order
        assert len(np.unique(labels[f][labels[f] > 0])) == counts[f]             # 1..L all used
    # -- exact: background + blur on crops (incl. all four borders and corners)
    _check_bg_blur_crops(oracle, clip, filtered, state, 5.0, 15)
    # -- exact: whole frames through threshold, close, labelling
    for f in (1, 77, 127):
        m = oracle.threshold_u8(filtered[f], 20)
        m = oracle.morph_u8(oracle.morph_u8(m, oracle.DILATE, oracle.RECT, 5), oracle.ERODE, oracle.RECT, 5)
        rl, rc = oracle.label(m, 4)
        assert np.array_equal(mask[f], m) and rc == counts[f] and np.array_equal(labels[f], rl), f
    # -- the chip-wide labelling passes give the same label maps; so does a counts-only run
    eng.set_background(None, 0)
    _hip.check(_hip.lib().va_test_hook_labelling(1, 0))
    try:
        out1 = eng.run(clip, want=("labels", "counts"))
    finally:
        _hip.check(_hip.lib().va_test_hook_labelling(0, 0))
    assert np.array_equal(out1["counts"], counts)
    assert np.array_equal(out1["labels"], labels)
    eng.set_background(None, 0)
    assert np.array_equal(eng.run(clip, want=("counts",))["counts"], counts)
    eng.close()


def _f32_clip(n, h, w, c, seed):
    """SURVEY.md 8(d) cfg#5: U(0,1) noise + a low-frequency sinusoid drifting 0.1 %/frame"""
    rng = np.random.default_rng(seed)
    yy = np.arange(h, dtype=np.float32)[:, None, None]
    xx = np.arange(w, dtype=np.float32)[None, :, None]
    ph = np.arange(c, dtype=np.float32)[None, None, :]
    clip = np.empty((n, h, w, c), np.float32)
    for t in range(n):
        wave = 0.25 * np.sin(xx * (2 * np.pi / 480) + yy * (2 * np.pi / 270) + ph + 0.001 * t * 2 * np.pi)
        clip[t] = rng.random((h, w, c), dtype=np.float32) * 0.5 + 0.25 + wave.astype(np.float32)
    return clip


def _check_f32_crops(oracle, clip, out, rate, sigma, radius, frames, size):
    n, h, w, c = clip.shape
    for (ya, xa) in ((0, 0), (h - size, w - size), (0, w // 2), (h // 2 - 3, 0), (h // 3, w // 3), (h - size, w // 4)):
        crop = np.ascontiguousarray(clip[:, ya:ya + size, xa:xa + size])
        diff, _ = oracle.bg_ema_f32(crop, rate=rate)
        y0 = 0 if ya == 0 else radius
        y1 = size if ya + size == h else size - radius
        x0 = 0 if xa == 0 else radius
        x1 = size if xa + size == w else size - radius
        for f in frames:
            blur = oracle.gaussian_f32(diff[f], sigma, layout="hwc")
            got = out[f, ya + y0:ya + y1, xa + x0:xa + x1]
            assert np.array_equal(got.view(np.uint32), blur[y0:y1, x0:x1].view(np.uint32)), (ya, xa, f)


def test_cfg5_1080p_f32x3_sigma9_ema(oracle):
    """BASELINE.json configs[4]: 1920x1080 float32 x 3 channels, batch 256, adaptive (EMA)
    background + per-channel sigma=9 Gaussian (73 taps) through va_pipeline_run.  Bit patterns
    against the oracle on crops (interior and all border kinds), whole-batch properties."""
    n, h, w, c = 256, 1080, 1920, 3
    rate, sigma, radius = 0.02, 9.0, 36
    clip = _f32_clip(n, h, w, c, seed=5)
    eng = _engine(size=(w, h), channels=c, dtype=np.float32, max_batch=n, background="ema",
                  bg_rate=rate, sigma=sigma)
    out = eng.run(clip, want=("filtered",))["filtered"]
    state, n_seen = eng.get_background()
    assert n_seen == n and out.shape == clip.shape and out.dtype == np.float32
    # properties: the first frame initialises the model (difference 0 -> blur exactly 0);
    # everything finite and non-negative (|diff| >= 0, taps > 0); EMA state on crops
    assert not out[0].any()
    assert np.isfinite(out).all() and out.min() >= 0.0
    assert out[1:].max() > 0.01
    crop = np.ascontiguousarray(clip[:, 500:532, 900:932])
    _, bg = oracle.bg_ema_f32(crop, rate=rate)
    assert np.array_equal(state[500:532, 900:932].view(np.uint32), bg.view(np.uint32))
    _check_f32_crops(oracle, clip, out, rate, sigma, radius, frames=(1, 100, 255), size=112)
    # splitting the batch must not change a bit (state hand-over between runs)
    eng.set_background(None, 0)
    a = eng.run(clip[:100], want=("filtered",))["filtered"]
    assert np.array_equal(a.view(np.uint32), out[:100].view(np.uint32))
    b = eng.run(clip[100:130], want=("filtered",))["filtered"]
    assert np.array_equal(b.view(np.uint32), out[100:130].view(np.uint32))
    eng.close()


def test_stress_slice_fused_chain(oracle):
    """a bounded, seeded slice of tools/stress_parity.py: random sizes / sigmas / thresholds /
    morphology chains / connectivities / batch sizes through the fused chain against the oracle"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stress_parity
    rng = np.random.default_rng(20261004)
    for case in range(1, 41):
        ok, desc = stress_parity.run_case(rng, case)
        assert ok, desc


def test_stress_slice_gaussian_f32(oracle):
    """a bounded, seeded slice of tools/stress_f32.py: bit patterns of the float32 Gaussian"""
    from video import ops
    rng = np.random.default_rng(55)
    for case in range(60):
        c = int(rng.choice([1, 3]))
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 400))
        if rng.random() < 0.15:
            w = int(rng.choice([1920, 1280, 640, 2000]))
        n = int(rng.integers(1, 4))
        sigma = float(rng.choice([0.5, 1.0, 2.0, 3.3, 5.0, 9.0, 11.0, 12.5, 15.0]))
        shape = (n, h, w, c) if c == 3 else (n, h, w)
        f = (rng.random(shape, dtype=np.float32) * 3 - 1).astype(np.float32)
        f.flat[::53] = 0.0
        f.flat[7::131] = -0.0
        ref = oracle.gaussian_f32(f, sigma)
        got = ops.gaussian_blur(f, sigma, color=(c == 3))
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (shape, sigma)


def test_gaussian_f32_column_kernels(oracle):
    """the compile-time-radius kernels (input-stationary row pass for sigma 2/3/5/9, unrolled 15/16-rows-per-
    thread column pass for integer sigmas) against the oracle, and the runtime-radius kernels (test hook)
    against the same oracle, over heights around the column kernels' 120/128-row steps"""
    from video import ops, _hip
    rng = np.random.default_rng(77)
    shapes = [(120, 64), (121, 68), (128, 200), (129, 64), (240, 96), (241, 132), (257, 64), (384, 68),
              (1, 64), (2, 72), (37, 64), (500, 36)]
    for h, w in shapes:
        for sigma in (1.0, 2.0, 3.0, 5.0, 7.0, 9.0):
            c = 3 if (h + int(sigma)) % 3 == 0 else 1
            shape = (2, h, w, 3) if c == 3 else (2, h, w)
            f = (rng.random(shape, dtype=np.float32) * 3 - 1).astype(np.float32)
            f.flat[::41] = 0.0
            f.flat[5::97] = -0.0
            ref = oracle.gaussian_f32(f, sigma)
            got = ops.gaussian_blur(f, sigma, color=(c == 3))
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (shape, sigma)
            _hip.check(_hip.lib().va_test_hook_gaussian_f32(3))
            try:
                gen = ops.gaussian_blur(f, sigma, color=(c == 3))
            finally:
                _hip.check(_hip.lib().va_test_hook_gaussian_f32(0))
            assert np.array_equal(gen.view(np.uint32), ref.view(np.uint32)), (shape, sigma, "generic")


def test_running_mean_user_state_in_and_out_of_u8_range(oracle):
    """the pipeline's running mean drops the saturation of |frame - mean| while its state is known to lie in
    [0, 255]; a state set from outside that range (va_bg_set_state) must bring the saturating kernel back"""
    rng = np.random.default_rng(8)
    n, h, w = 7, 48, 160                                   # 16-byte rows: the wide kernel
    clip = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    for lo, hi in ((0.0, 255.0), (-50.0, 700.0)):
        state0 = rng.uniform(lo, hi, (h, w))
        eng = _engine(size=(w, h), max_batch=n, background="mean")
        eng.set_background(state0, 5)
        got = eng.run(clip, want=("filtered",))["filtered"]
        state, seen = eng.get_background()
        eng.close()
        ref, mean = oracle.bg_mean_u8(clip, state0, 5)
        assert seen == 5 + n
        assert np.array_equal(got, ref), (lo, hi)
        assert np.array_equal(state.view(np.uint64), mean.view(np.uint64)), (lo, hi)


def test_f32_ema_short_runs_and_restaging(oracle):
    """regression for two races of the fused float32 EMA path (found by tools/stress_f32_kernels.py): frames fed
    one per run (a late workgroup must not see its neighbour's already advanced halo state), and several frames
    per run on small frames (a loader wave must not restage an LDS buffer another one still copies out)"""
    rng = np.random.default_rng(99)
    for rep in range(4):
        for (n, h, w, c, sigma, rate, per_run) in ((6, 138, 1920, 3, 2.0, 1.0, 1), (3, 240, 1920, 3, 2.0, 0.3, 3),
                                                  (4, 379, 1280, 1, 1.0, 0.3, 1), (6, 125, 1280, 3, 4.0, 1.0, 2)):
            shape = (n, h, w, c) if c == 3 else (n, h, w)
            clip = (rng.random(shape, dtype=np.float32) * 2 - 0.5).astype(np.float32)
            eng = _engine(size=(w, h), channels=c, dtype=np.float32, max_batch=n, background="ema", bg_rate=rate,
                          sigma=sigma)
            got = np.concatenate([eng.run(clip[i:i + per_run], want=("filtered",))["filtered"]
                                  for i in range(0, n, per_run)])
            state, seen = eng.get_background()
            eng.close()
            diff, bg = oracle.bg_ema_f32(clip.reshape(n, -1), rate=np.float32(rate))
            ref = oracle.gaussian_f32(diff.reshape(shape), sigma)
            assert seen == n
            assert np.array_equal(state.ravel().view(np.uint32), bg.ravel().view(np.uint32)), (shape, "state")
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (shape, sigma, rate, per_run)


def test_f32_ema_4k_more_chunks_than_cus(oracle):
    """3840x2160x3 float32: the row kernel's chunks outnumber the CUs, so workgroups of a later round start
    after earlier ones have finished -- they must still read the state from BEFORE the batch for their halos"""
    n, h, w, c, sigma, rate = 3, 2160, 3840, 3, 2.0, 0.3
    rng = np.random.default_rng(4)
    clip = (rng.random((n, h, w, c), dtype=np.float32) * 2 - 0.5).astype(np.float32)
    eng = _engine(size=(w, h), channels=c, dtype=np.float32, max_batch=n, background="ema", bg_rate=rate,
                  sigma=sigma)
    got = np.concatenate([eng.run(clip[:2], want=("filtered",))["filtered"],
                          eng.run(clip[2:], want=("filtered",))["filtered"]])
    state, seen = eng.get_background()
    eng.close()
    diff, bg = oracle.bg_ema_f32(clip.reshape(n, -1), rate=np.float32(rate))
    ref = oracle.gaussian_f32(diff.reshape(clip.shape), sigma)
    assert seen == n
    assert np.array_equal(state.ravel().view(np.uint32), bg.ravel().view(np.uint32))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_stress_slice_round2_paths(oracle):
    """a bounded, seeded slice of tools/stress_misc.py: float32 EMA + blur through the pipeline at random
    shapes with split batches, cv2.resize in every mode, contour moments of traced contours"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stress_misc
    rng = np.random.default_rng(424242)
    for fn in (stress_misc.case_f32_pipeline, stress_misc.case_resize, stress_misc.case_contour_moments):
        for _ in range(40):
            ok, desc = fn(rng)
            assert ok, desc


def test_repeated_runs_are_bit_identical():
    """the same batch through the same chain, several times: every run gives the same bits (a store-data hazard of
    16-byte buffer stores showed up as run-to-run differences in an experimental kernel: DESIGN.md 13.10,
    tools/debug/determinism_f32.py / determinism_u8.py run this at full batch size)"""
    rng = np.random.default_rng(99)
    n, h, w = 24, 1080, 1920
    clip = _moving_blob_clip(n, h, w, seed=9, nblobs=30, salt=0.002, noise_tiles=4)
    eng = _engine(size=(w, h), max_batch=n, background="mean", sigma=5.0, thresh=20,
                  morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4)
    first = None
    for _ in range(4):
        eng.set_background(np.full((h, w), 100.0), 50)
        out = eng.run(clip, want=("filtered", "labels", "counts"))
        if first is None:
            first = out
        else:
            for k in first:
                assert np.array_equal(out[k], first[k]), k
    eng.close()
    f = (rng.random((12, h, w, 3), dtype=np.float32) * 2 - 0.5).astype(np.float32)
    eng = _engine(size=(w, h), channels=3, dtype=np.float32, max_batch=12, background="ema", bg_rate=0.02, sigma=9.0)
    first = None
    for _ in range(4):
        eng.set_background(np.zeros((h, w, 3), np.float32), 5)
        out = eng.run(f, want=("filtered",))["filtered"]
        state, _ = eng.get_background()
        if first is None:
            first = (out, state)
        else:
            assert np.array_equal(out.view(np.uint32), first[0].view(np.uint32))
            assert np.array_equal(state.view(np.uint32), first[1].view(np.uint32))
    eng.close()
    from video import ops
    g = (rng.random((8, h, w), dtype=np.float32) * 3 - 1).astype(np.float32)
    ref = ops.gaussian_blur(g, 5.0)
    for _ in range(3):
        assert np.array_equal(ops.gaussian_blur(g, 5.0).view(np.uint32), ref.view(np.uint32))
