"""CPU: the oracle (oracle/va_oracle.c) against the committed known-answer vectors.

The vectors come from scipy.ndimage / literal NumPy restatements of the reference formulas /
scikit-image (tests/golden/make_golden.py), i.e. from the third-party code the reference
itself calls -- the reference has no tests or fixtures of its own.
"""
import json

import numpy as np
import pytest


def _names(golden):
    return [str(n) for n in golden["mask_names"]]


def test_label_matches_scipy_vectors(golden, oracle):
    for name in _names(golden):
        m = golden["mask_" + name]
        for conn in (4, 8):
            lab, cnt = oracle.label(m, conn)
            assert cnt == int(golden["count%d_%s" % (conn, name)]), (name, conn)
            assert lab.dtype == np.int32
            assert np.array_equal(lab, golden["labels%d_%s" % (conn, name)]), (name, conn)


def test_label_nonzero_is_foreground(oracle):
    lab, cnt = oracle.label(np.array([[0, 5, 0, 255]], np.uint8))
    assert cnt == 2 and lab.tolist() == [[0, 1, 0, 2]]


def test_areas_largest_region_and_tie(golden, oracle):
    for name in _names(golden):
        m = golden["mask_" + name]
        lab, cnt = oracle.label(m, 4)
        st = oracle.region_stats(lab, cnt)
        assert np.array_equal(st[:, 0], golden["areas4_" + name]), name
        if cnt:
            big, area = oracle.get_largest_region(m, ret_area=True)
            assert np.array_equal(big.astype(np.uint8), golden["largest4_" + name]), name
            assert area == golden["areas4_" + name].max()
        else:
            with pytest.raises(ValueError):
                oracle.get_largest_region(m)
    # first maximum wins on ties (np.argmax rule, video/analysis/regions.py:169)
    big = oracle.get_largest_region(golden["mask_tie"])
    assert big[2, 2] and not big[10, 20]


def test_raw_moments_and_bbox(golden, oracle):
    for name in _names(golden):
        m = golden["mask_" + name]
        lab, cnt = oracle.label(m, 4)
        st = oracle.region_stats(lab, cnt)
        assert np.array_equal(st[:, :10], golden["lmoments4_" + name]), name
        for l in range(1, cnt + 1):
            ys, xs = np.nonzero(lab == l)
            assert st[l - 1, 10:14].tolist() == [xs.min(), ys.min(), xs.max(), ys.max()]
        # whole-mask moments == sum over labels
        if cnt:
            assert np.array_equal(st[:, :10].sum(0), golden["moments_" + name])


def test_central_moments_match_skimage(golden, oracle):
    sk = json.loads(str(golden["skimage_json"]))
    if not sk:
        pytest.skip("fixture was generated without scikit-image")
    for name, ref in sk.items():
        raw = golden["moments_" + name].astype(np.float64)
        out = oracle.complete_moments(raw)
        got = dict(zip(["mu20", "mu11", "mu02", "mu30", "mu21", "mu12", "mu03"], out[:7]))
        assert raw[0] == ref["m00"] and raw[1] == ref["m10"] and raw[2] == ref["m01"]
        scale = max(1.0, abs(raw[6]), abs(raw[9]))
        for k, v in got.items():
            assert abs(v - ref[k]) <= 1e-9 * scale, (name, k, v, ref[k])


def test_running_mean_is_bit_exact_numpy(golden, oracle):
    for n in (1, 2, 8, 64, 256):
        fr = golden["bgframes_%d" % n]
        diff, mean = oracle.bg_mean_u8(fr)
        assert np.array_equal(mean, golden["mean_%d" % n])          # bit-exact f64
        # split batches + carried state == one batch
        if n >= 8:
            d1, m1 = oracle.bg_mean_u8(fr[:3])
            d2, m2 = oracle.bg_mean_u8(fr[3:], mean=m1, n_seen=3)
            assert np.array_equal(m2, mean)
            assert np.array_equal(np.concatenate([d1, d2]), diff)
        # emitted difference: |frame - bg_prev| truncated, bg_prev = mean of earlier frames
        prev = np.zeros(fr.shape[1:])
        for k in range(n):
            exp = np.trunc(np.abs(fr[k] - prev)).astype(np.uint8)
            assert np.array_equal(diff[k], exp)
            prev = prev * k / (k + 1) + fr[k] / (k + 1)
    frc = golden["bgframes_color"]
    assert np.array_equal(oracle.bg_mean_u8(frc)[1], golden["mean_color"])


def test_welford_matches_numpy(golden, oracle):
    for n in (8, 64, 256):
        fr = golden["bgframes_%d" % n]
        mean, m2 = oracle.welford_u8(fr)
        assert np.array_equal(mean, golden["wmean_%d" % n])
        assert np.array_equal(np.sqrt(m2 / (n - 1)), golden["wstd_%d" % n])


def test_morphology_matches_scipy_vectors(golden, oracle):
    for nm in ("img", "bin"):
        im = golden["morph_" + nm]
        for k in (3, 5, 7):
            assert np.array_equal(oracle.morph_u8(im, oracle.ERODE, oracle.RECT, k),
                                  golden["erode_rect%d_%s" % (k, nm)])
            assert np.array_equal(oracle.morph_u8(im, oracle.DILATE, oracle.RECT, k),
                                  golden["dilate_rect%d_%s" % (k, nm)])
        assert np.array_equal(oracle.morph_u8(im, oracle.ERODE, oracle.CROSS, 3),
                              golden["erode_cross3_" + nm])
        assert np.array_equal(oracle.morph_u8(im, oracle.DILATE, oracle.CROSS, 3),
                              golden["dilate_cross3_" + nm])


def test_structuring_elements(oracle):
    assert oracle.structuring_element(oracle.CROSS, 3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert oracle.structuring_element(oracle.RECT, 5).all()
    e = oracle.structuring_element(oracle.ELLIPSE, 5)
    assert e.tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1],
                          [0, 0, 1, 0, 0]]      # cv2.getStructuringElement(MORPH_ELLIPSE,(5,5))


def test_gaussian_taps(golden, oracle):
    for s, ks in ((0.5, 5), (1.0, 7), (2.0, 13), (3.0, 19), (5.0, 31)):
        t = oracle.gauss_taps_q8(s)
        assert len(t) == ks == oracle.gauss_ksize(s)          # cvRound(6 sigma + 1) | 1
        assert int(t.sum()) == 256                            # unity gain in q8.8
        assert np.array_equal(t, t[::-1])
        assert np.array_equal(t, golden["taps_q8_%g" % s])
    t = oracle.gauss_taps_f32(9.0)
    assert len(t) == 73 == oracle.gauss_ksize(9.0, False)     # cvRound(8 sigma + 1) | 1
    assert abs(float(t.sum(dtype=np.float64)) - 1) < 1e-6
    assert np.array_equal(t, golden["taps_f32_9"])


def test_gaussian_vectors_and_properties(golden, oracle):
    from scipy import ndimage
    for nm in ("imp", "step", "ramp", "noise", "tiny"):
        im = golden["gin_" + nm]
        for s in (2.0, 5.0):
            out = oracle.gaussian_u8(im, s)
            assert np.array_equal(out, golden["gout_%s_%g" % (nm, s)])
            if min(im.shape) > 3 * s:      # loose physical sanity bound only (not parity)
                ref = ndimage.gaussian_filter(im.astype(float), s, mode="mirror", truncate=3.0)
                assert np.abs(out - ref).max() <= 2.0
    # constant image is a fixed point (taps sum to exactly 256, rounding is exact)
    c = np.full((20, 30), 77, np.uint8)
    assert np.array_equal(oracle.gaussian_u8(c, 5.0), c)
    # impulse response = outer product of the taps, rounded
    t = oracle.gauss_taps_q8(2.0).astype(np.int64)
    imp = np.zeros((41, 41), np.uint8)
    imp[20, 20] = 255
    exp = ((np.outer(t, t) * 255 + 32768) >> 16).astype(np.uint8)
    assert np.array_equal(oracle.gaussian_u8(imp, 2.0)[14:27, 14:27], exp)
    col = golden["gin_color"]
    outc = oracle.gaussian_u8(col, 2.0, layout="hwc")
    assert np.array_equal(outc, golden["gout_color_2"])
    for ch in range(3):         # channels are independent
        assert np.array_equal(outc[..., ch], oracle.gaussian_u8(np.ascontiguousarray(col[..., ch]), 2.0))
    f = golden["gin_f32"]
    assert np.array_equal(oracle.gaussian_f32(f, 2.0), golden["gout_f32_2"])
    assert np.array_equal(oracle.gaussian_f32(f, 9.0), golden["gout_f32_9"])
    ref = ndimage.gaussian_filter(f.astype(np.float64), 2.0, mode="mirror", truncate=4.0)
    assert np.abs(oracle.gaussian_f32(f, 2.0) - ref).max() < 2e-3


def test_threshold_mono_diff(oracle):
    a = np.arange(256, dtype=np.uint8)
    assert np.array_equal(oracle.threshold_u8(a, 20), np.where(a > 20, 255, 0))
    rng = np.random.default_rng(3)
    c = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    assert np.array_equal(oracle.mono_mean_u8(c), np.mean(c, axis=2).astype(np.uint8))
    p = rng.integers(0, 256, (7, 9), dtype=np.uint8)
    q = rng.integers(0, 256, (7, 9), dtype=np.uint8)
    assert np.array_equal(oracle.time_difference_u8(p, q), p.astype(np.int16) - q)


def test_chain_vectors(golden, oracle):
    clip = golden["chain_clip"]
    mask, labels, counts, mean = oracle.chain_u8(clip, 2.0, 20, morph_ksize=5, connectivity=4)
    assert np.array_equal(mask, golden["chain_mask"])
    assert np.array_equal(labels, golden["chain_labels"])
    assert np.array_equal(counts, golden["chain_counts"])
    assert np.array_equal(mean, golden["chain_mean"])
    assert counts.max() >= 1
    # the chain equals its stages
    diff, mean2 = oracle.bg_mean_u8(clip)
    m = oracle.threshold_u8(oracle.gaussian_u8(diff, 2.0), 20)
    m = oracle.morph_u8(oracle.morph_u8(m, oracle.DILATE, oracle.RECT, 5), oracle.ERODE,
                        oracle.RECT, 5)
    assert np.array_equal(m, mask) and np.array_equal(mean, mean2)


def test_contours_known_answers_and_properties(golden, oracle):
    """A8: parity with cv2 itself is unpinned (no OpenCV offline); these pin the restatement
    to known OpenCV outputs for simple shapes and to geometric invariants."""
    from scipy import ndimage
    m = np.zeros((8, 10), np.uint8)
    m[2:5, 3:7] = 1
    cs = oracle.find_contours_external_simple(m)
    # a filled rectangle: OpenCV returns the 4 corners, starting top-left, going down first
    assert len(cs) == 1 and cs[0].reshape(-1, 2).tolist() == [[3, 2], [3, 4], [6, 4], [6, 2]]
    assert oracle.contour_area(cs[0]) == 6.0                        # (w-1)*(h-1)
    m = np.zeros((8, 10), np.uint8)
    m[1, 1] = m[3, 3] = m[4, 4] = 1
    m[6, 1:9] = 1
    cs = [c.reshape(-1, 2).tolist() for c in oracle.find_contours_external_simple(m)]
    assert cs == [[[1, 6], [8, 6]], [[3, 3], [4, 4]], [[1, 1]]]     # most recently found first
    ring = np.zeros((12, 12), np.uint8)
    ring[1:11, 1:11] = 1
    ring[3:9, 3:9] = 0
    ring[5:7, 5:7] = 1                                              # blob inside the hole
    cs = oracle.find_contours_external_simple(ring)
    assert len(cs) == 1 and cs[0].reshape(-1, 2).tolist() == [[1, 1], [1, 10], [10, 10], [10, 1]]
    assert oracle.find_contours_external_simple(np.zeros((5, 5), np.uint8)) == []
    with pytest.raises(RuntimeError):
        oracle.get_contour_from_largest_region(np.zeros((5, 5), np.uint8))
    # invariants on the golden masks: one external contour per 8-component that is not nested,
    # every point is a foreground pixel touching the background (or the frame edge)
    for name in [str(n) for n in golden["mask_names"]]:
        mask = golden["mask_" + name]
        cs = oracle.find_contours_external_simple(mask)
        assert len(cs) <= int(golden["count8_" + name])
        pad = np.pad(mask, 1)
        for c in cs:
            for x, y in c.reshape(-1, 2):
                assert mask[y, x]
                assert pad[y:y + 3, x:x + 3].min() == 0
        # first points in OpenCV order are the components' first raster pixels, bottom-most first
        firsts = [(int(c[0, 0, 1]), int(c[0, 0, 0])) for c in cs]
        assert firsts == sorted(firsts, reverse=True)
    # filled polygon of the contour == the (hole-filled) component, for blob-like shapes
    from PIL import Image, ImageDraw
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[:90, :120]
    for _ in range(10):
        m = np.zeros((90, 120), bool)
        for _ in range(4):
            cx, cy, a, b = rng.uniform(20, 100), rng.uniform(20, 70), rng.uniform(4, 18), rng.uniform(4, 18)
            m |= ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1
        contour, area = oracle.get_contour_from_largest_region(m, ret_area=True)
        lab, cnt = ndimage.label(m, np.ones((3, 3)))
        img = Image.new("L", (120, 90), 0)
        ImageDraw.Draw(img).polygon([tuple(p) for p in contour.astype(int)], fill=1, outline=1)
        poly = np.array(img, bool)
        comp = lab[int(contour[0][1]), int(contour[0][0])]
        filled = ndimage.binary_fill_holes(lab == comp)
        assert np.array_equal(poly, filled)
        assert area <= filled.sum()


def test_n2_numpy_oracle_matches_literal_scipy(oracle):
    """the oracle's detect_peaks / image_statistics are plain NumPy (the GPU box needs no SciPy);
    here, where SciPy is installed, they are pinned to the literal scipy.ndimage restatements of
    video/analysis/image.py:131-201, 267-306"""
    pytest.importorskip("scipy")
    rng = np.random.default_rng(5)
    for shape in ((1, 1), (1, 9), (7, 1), (3, 3), (40, 53), (65, 31)):
        for im in (rng.integers(0, 6, shape, dtype=np.uint8) * rng.integers(0, 2, shape, dtype=np.uint8),
                   (rng.integers(0, 256, shape) // 32 * 32).astype(np.uint8),
                   np.zeros(shape, np.uint8), np.full(shape, 9, np.uint8)):
            for plateaus in (True, False):
                assert np.array_equal(oracle.detect_peaks(im, plateaus),
                                      oracle.detect_peaks_scipy(im, plateaus)), (shape, plateaus)
    img = rng.integers(0, 256, (37, 45), dtype=np.uint8)
    for kernel in ("box", "ellipse"):
        for ksize in (1, 2, 5):
            for excl in (False, True):
                for prior in (0, 128, None, 100.5):
                    m, v = oracle.image_statistics(img, kernel, ksize, prior, excl)
                    rm, rv = oracle.image_statistics_scipy(img, kernel, ksize, prior, excl)
                    if prior in (0, 128):
                        assert np.array_equal(m, rm) and np.array_equal(v, rv)
                    else:
                        assert np.allclose(m, rm, rtol=1e-13, atol=1e-10)
                        assert np.allclose(v, rv, rtol=1e-10, atol=1e-7)


def test_contour_moments_known_answers(oracle):
    """cv2.moments(contour) restatement (video/analysis/image.py:355, shapes.py:533): analytic
    polygon moments of rectangles / triangles, orientation independence, degenerate contours,
    and agreement with the raster moments' derived quantities in the limit of large shapes"""
    sq = np.array([[0, 0], [4, 0], [4, 4], [0, 4]], np.int32)
    m = oracle.contour_moments(sq)
    assert (m["m00"], m["m10"], m["m01"], m["m11"], m["m30"]) == (16.0, 32.0, 32.0, 64.0, 256.0)
    assert abs(m["m20"] - 256 / 3) < 1e-12 and abs(m["m21"] - 512 / 3) < 1e-12
    assert abs(m["mu20"] - 64 / 3) < 1e-12 and m["mu11"] == 0.0
    assert abs(m["nu20"] - (64 / 3) / 256) < 1e-15
    r = oracle.contour_moments(sq[::-1])                       # clockwise: same moments
    assert all(r[k] == m[k] for k in ("m00", "m10", "m01", "m20", "m11", "m02"))
    tri = np.array([[0, 0], [6, 0], [0, 3]], np.float32)       # area 9, centroid (2, 1)
    t = oracle.contour_moments(tri)
    assert t["m00"] == 9.0 and t["m10"] == 18.0 and t["m01"] == 9.0
    assert abs(t["m20"] - 54.0) < 1e-12 and abs(t["m02"] - 13.5) < 1e-12 and abs(t["m11"] - 13.5) < 1e-12
    for degenerate in (np.array([[3, 3]], np.int32), np.array([[1, 1], [5, 1]], np.int32),
                       np.array([[0, 0], [2, 2], [4, 4]], np.int32)):
        assert all(v == 0.0 for v in oracle.contour_moments(degenerate).values())
    # (N,1,2) layout of cv2.findContours, float dtype = float32 points, shifted rectangle
    rect = np.array([[[10, 20]], [[10, 29]], [[49, 29]], [[49, 20]]], np.float64)
    q = oracle.contour_moments(rect)
    assert q["m00"] == 39.0 * 9.0 and q["m10"] / q["m00"] == 29.5 and q["m01"] / q["m00"] == 24.5
    # contour of a filled raster region: polygon through the boundary pixel centres
    yy, xx = np.mgrid[:80, :90]
    disc = ((xx - 45) ** 2 + (yy - 38) ** 2 <= 30 ** 2).astype(np.uint8)
    c = oracle.get_contour_from_largest_region(disc)
    cm = oracle.contour_moments(np.asarray(c, np.int32))
    assert abs(cm["m00"] - oracle.contour_area(np.asarray(c, np.int32))) < 1e-9
    assert abs(cm["m10"] / cm["m00"] - 45) < 1e-9 and abs(cm["m01"] / cm["m00"] - 38) < 1e-9
    assert abs(cm["mu20"] - cm["mu02"]) / cm["mu20"] < 1e-9 and abs(cm["mu11"]) < 1e-6


# ---------------------------------------------------------------------------- reference-era Gaussian taps
def _cv3_taps_numpy(sigma):
    """OpenCV 2.4 / 3.x: float32 getGaussianKernel, every tap cvRound(k * 256) on its own -- restated
    here with NumPy float32 arithmetic, independently of the C oracle"""
    n = int(np.rint(sigma * 6 + 1)) | 1
    x = np.arange(n) - (n - 1) * 0.5
    cf = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)
    inv = 1.0 / cf.astype(np.float64).sum()
    k = (cf.astype(np.float64) * inv).astype(np.float32)
    return np.rint(k.astype(np.float64) * 256.0).astype(np.uint16)          # rint: ties to even, as cvRound


@pytest.mark.parametrize("sigma", [0.5, 1.0, 1.7, 2.0, 2.4, 3.0, 5.0, 7.0])
def test_cv3_taps_and_blur(oracle, sigma):
    taps = oracle.gauss_taps_q8(sigma, "cv3")
    assert np.array_equal(taps, _cv3_taps_numpy(sigma))
    assert np.array_equal(taps, taps[::-1])
    s = int(taps.sum())
    assert abs(s - 256) <= len(taps) // 2           # not forced to 256 ...
    if sigma in (1.7, 2.0, 2.4, 7.0):
        assert s != 256                             # ... and it is not (258, 257, 259, 251)
    # a constant image c comes out as sat((c * S^2 + 2^15) >> 16): gain != 1 is part of the definition
    for c in (0, 1, 100, 254, 255):
        out = oracle.gaussian_u8(np.full((1, 40, 48), c, np.uint8), sigma, tap_rule="cv3")
        assert np.all(out == min(255, (c * s * s + 32768) >> 16)), (sigma, c)
    # the default rule keeps unity gain
    assert int(oracle.gauss_taps_q8(sigma).sum()) == 256
    # literal two-pass integer convolution with the NumPy taps (reflect-101 borders)
    rng = np.random.default_rng(int(sigma * 10))
    im = rng.integers(0, 256, (23, 37), dtype=np.uint8)
    r = len(taps) // 2
    pad = np.pad(im.astype(np.int64), ((0, 0), (r, r)), mode="reflect")
    rows = sum(int(taps[i]) * pad[:, i:i + im.shape[1]] for i in range(len(taps)))
    pad = np.pad(rows, ((r, r), (0, 0)), mode="reflect")
    acc = sum(int(taps[j]) * pad[j:j + im.shape[0], :] for j in range(len(taps)))
    ref = np.minimum((acc + 32768) >> 16, 255).astype(np.uint8)
    assert np.array_equal(oracle.gaussian_u8(im, sigma, tap_rule="cv3"), ref)


# ---------------------------------------------------------------------------- Lanczos-4 / float32 resize
def test_lanczos4_and_float_resize_known_answers(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (31, 45), dtype=np.uint8)
    fimg = rng.normal(0, 1, (31, 45)).astype(np.float32)
    for mode in ("nearest", "linear", "cubic", "area", "lanczos"):
        # same size: sample positions fall on the pixels, every kernel degenerates to a delta
        assert np.array_equal(oracle.resize_u8(img, (45, 31), mode), img), mode
        assert np.array_equal(oracle.resize_f32(fimg, (45, 31), mode), fimg), mode
        # constant images stay constant (float: to rounding; Lanczos u8: its quantised taps, +-1)
        flat = oracle.resize_u8(np.full((20, 30), 77, np.uint8), (47, 33), mode)
        assert np.abs(flat.astype(int) - 77).max() <= (1 if mode == "lanczos" else 0), mode
        fflat = oracle.resize_f32(np.full((20, 30), 0.25, np.float32), (47, 33), mode)
        assert np.abs(fflat - 0.25).max() < 1e-6, mode
    # float32 linear, growing 2x along x only: literal two-tap formula with OpenCV's sample positions
    row = np.arange(8, dtype=np.float32)[None, :] ** 2
    out = oracle.resize_f32(np.repeat(row, 3, 0), (16, 3), "linear")
    fx = ((np.arange(16) + 0.5) * 0.5 - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(int)
    a1 = (fx - sx).astype(np.float32)
    left, right = np.clip(sx, 0, 7), np.clip(sx + 1, 0, 7)
    a1 = np.where(sx < 0, 0, np.where(sx >= 7, 0, a1)).astype(np.float32)
    expect = row[0, left] * (np.float32(1) - a1) + row[0, right] * a1
    assert np.allclose(out[1], expect, rtol=0, atol=1e-5)
    # Lanczos shrinks a smooth ramp to (nearly) a ramp and overshoots at a step (negative lobes)
    ramp = np.tile(np.linspace(20, 220, 64).astype(np.float32), (8, 1))
    small = oracle.resize_f32(ramp, (32, 8), "lanczos")
    assert np.abs(np.diff(small[4, 4:-4]) - (200 / 63 * 2)).max() < 0.05
    step = np.zeros((8, 64), np.float32)
    step[:, 32:] = 1
    big = oracle.resize_f32(step, (128, 8), "lanczos")
    assert big.max() > 1.02 and big.min() < -0.02


# ---------------------------------------------------------------------------- temporal statistics, any dtype
@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32])
def test_temporal_statistics_follow_numpy_promotions(oracle, dtype):
    """measure_mean / measure_mean_std (video/analysis/video.py:26-55) executed LITERALLY with NumPy on
    frames of the dtypes the reference meets; the oracle restates the promotions (float32 frames:
    `frame/(n + 1)` is a float32 quotient) and must reproduce every bit"""
    rng = np.random.default_rng(7)
    if dtype == np.float32:
        frames = rng.normal(0.4, 0.3, (11, 9, 13)).astype(np.float32)
    elif dtype == np.int16:
        frames = rng.integers(-255, 256, (11, 9, 13)).astype(np.int16)
    else:
        frames = rng.integers(0, 256, (11, 9, 13)).astype(np.uint8)
    mean = np.zeros(frames.shape[1:])
    for n, frame in enumerate(frames):                       # video/analysis/video.py:32-33
        mean = mean * n / (n + 1) + frame / (n + 1)
    assert np.array_equal(oracle.mean_any(frames), mean)
    m, M2 = np.zeros(frames.shape[1:]), np.zeros(frames.shape[1:])
    for n, frame in enumerate(frames):                       # :47-50
        delta = frame - m
        m = m + delta / (n + 1)
        M2 = M2 + delta * (frame - m)
    om, oq = oracle.welford_any(frames)
    assert np.array_equal(om, m) and np.array_equal(oq, M2)
    # split batches continue the recurrence
    a = oracle.mean_any(frames[:4])
    assert np.array_equal(oracle.mean_any(frames[4:], a, 4), mean)
