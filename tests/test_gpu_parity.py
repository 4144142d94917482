"""GPU: the HIP path (through the C ABI, via video.ops / video.engine) against the CPU oracle
and the committed golden vectors.  Bit-exact for every integer/byte/index result and for the
float64 running mean; the float32 Gaussian is bit-exact against the oracle's fmaf definition
(tolerance vs real OpenCV per the north star: 1 ULP float32, unverifiable offline).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from video import _hip, ops as _ops
    _hip.lib()          # raises HipUnavailableError (loudly) if the extension/GPU is missing
    return _ops


def _blob_clip(n, h, w, seed, nblobs=6, salt=0.0):
    rng = np.random.default_rng(seed)
    bg = np.clip(rng.normal(100, 10, (h, w)), 0, 255)
    yy, xx = np.mgrid[:h, :w]
    pos = rng.uniform(0, 1, (nblobs, 2)) * (w, h)
    vel = rng.uniform(-3, 3, (nblobs, 2))
    rad = rng.uniform(min(h, w) / 30 + 2, min(h, w) / 8 + 3, nblobs)
    out = np.empty((n, h, w), np.uint8)
    for t in range(n):
        f = bg + rng.normal(0, 4, (h, w))
        for (cx, cy), r in zip(pos + vel * t, rad):
            f[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] += 60
        if salt:
            f[rng.random((h, w)) < salt] = 255
        out[t] = np.clip(f, 0, 255).astype(np.uint8)
    return out


# ---------------------------------------------------------------------------- Gaussian
@pytest.mark.parametrize("sigma", [0.5, 1.0, 1.7, 2.0, 2.4, 3.0, 4.0, 5.0, 7.0])
def test_gaussian_u8_reference_era_taps(ops, oracle, sigma):
    """tap_rule='cv3' (OpenCV 2.4 / 3.x: every float32 tap rounded on its own, sum 251..259): the
    matrix-core kernel where the tap set allows it (sum <= 256, taps <= 127), the generic path with
    16- or 32-bit row sums otherwise, saturation for sums above 256 -- against the oracle"""
    from video import _hip
    assert np.array_equal(_hip.gauss_taps_q8(sigma, "cv3"), oracle.gauss_taps_q8(sigma, "cv3"))
    rng = np.random.default_rng(int(sigma * 13))
    for shape, color in (((3, 97, 208), False), ((2, 64, 75), False), ((1, 40, 64, 3), True)):
        im = rng.integers(0, 256, shape, dtype=np.uint8)
        im[0, :12] = 255                                   # saturating regions: sums above 256 overshoot 255
        ref = oracle.gaussian_u8(im, sigma, tap_rule="cv3")
        assert np.array_equal(ops.gaussian_blur(im, sigma, color=color, tap_rule="cv3"), ref), (sigma, shape)


def test_filter_blur_tap_rule_survives_contraction(oracle):
    """FilterBlur(tap_rule='cv3') read through a contracted chain and filter by filter: the rule travels
    into the fused engine (sigma = 2: tap sum 257, so the two rules differ visibly)"""
    from video import filters as F
    from video.io.memory import VideoMemory
    clip = _blob_clip(40, 64, 96, seed=14)
    diff, _ = oracle.bg_mean_u8(clip)
    ref = oracle.threshold_u8(oracle.gaussian_u8(diff, 2.0, tap_rule="cv3"), 20)
    assert not np.array_equal(ref, oracle.threshold_u8(oracle.gaussian_u8(diff, 2.0), 20))
    for contract in (True, False):
        F._GpuStage.contract = contract
        try:
            v = F.FilterThreshold(F.FilterBlur(F.FilterBackground(VideoMemory(clip)), 2, tap_rule="cv3"), 20)
            assert (v._runner() is not None) == contract
            assert np.array_equal(np.stack([np.array(f) for f in v]), ref), contract
            v.close()
        finally:
            F._GpuStage.contract = True
    with pytest.raises(ValueError):
        F.FilterBlur(VideoMemory(clip), 2, tap_rule="cv5")


def test_pipeline_with_reference_era_taps(oracle):
    from video.engine import FrameEngine
    clip = _blob_clip(6, 96, 160, seed=21, salt=0.002)
    for sigma in (2.0, 5.0):                               # 257 (generic path), 256 (matrix cores)
        eng = FrameEngine(size=(160, 96), max_batch=6, background="mean", sigma=sigma, thresh=20, tap_rule="cv3",
                          morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4)
        out = eng.run(clip, want=("filtered", "mask", "labels", "counts"))
        eng.close()
        diff, _ = oracle.bg_mean_u8(clip)
        blur = oracle.gaussian_u8(diff, sigma, tap_rule="cv3")
        m = oracle.threshold_u8(blur, 20)
        m = oracle.morph_u8(oracle.morph_u8(m, oracle.DILATE, oracle.RECT, 5), oracle.ERODE, oracle.RECT, 5)
        rl, rc = oracle.label_batch(m, 4)
        assert np.array_equal(out["filtered"], blur) and np.array_equal(out["mask"], m)
        assert np.array_equal(out["labels"], rl) and np.array_equal(out["counts"], rc)


@pytest.mark.parametrize("impl", [None, "generic"])
def test_gaussian_u8_golden(ops, golden, impl):
    for nm in ("imp", "step", "ramp", "noise", "tiny"):
        im = golden["gin_" + nm]
        for s in (2.0, 5.0):
            out = ops.gaussian_blur(im, s, implementation=impl)
            assert np.array_equal(out, golden["gout_%s_%g" % (nm, s)]), (nm, s, impl)
    col = golden["gin_color"]
    assert np.array_equal(ops.gaussian_blur(col, 2.0, color=True, implementation=impl),
                          golden["gout_color_2"])


@pytest.mark.parametrize("shape,sigma", [((3, 97, 133), 5.0), ((2, 64, 256), 2.0),
                                         ((1, 200, 517), 3.0), ((2, 31, 40), 1.0),
                                         ((1, 17, 300), 0.5), ((1, 480, 640), 2.0),
                                         ((2, 130, 70), 10.0), ((1, 9, 11), 5.0)])
def test_gaussian_u8_vs_oracle(ops, oracle, shape, sigma):
    rng = np.random.default_rng(hash(shape) % 1000)
    im = rng.integers(0, 256, shape, dtype=np.uint8)
    ref = oracle.gaussian_u8(im, sigma)
    assert np.array_equal(ops.gaussian_blur(im, sigma), ref)
    assert np.array_equal(ops.gaussian_blur(im, sigma, implementation="generic"), ref)


@pytest.mark.parametrize("shape,sigma", [((2, 64, 64), 5.0), ((1, 33, 80), 1.0), ((2, 100, 208), 5.3),
                                         ((1, 32, 1936), 2.5), ((3, 77, 96), 0.7), ((1, 250, 336), 4.0),
                                         ((1, 65, 112), 0.5), ((2, 129, 128), 3.3), ((1, 40, 64), 0.85)])
def test_gaussian_u8_single_launch_kernels(ops, oracle, shape, sigma):
    """shapes both single-launch kernels accept (w % 16 == 0, radius <= 16): the matrix-core
    kernel (library's choice when every tap <= 127) and the dot4/dot2 kernel against the oracle"""
    rng = np.random.default_rng(shape[2] + int(sigma * 10))
    im = rng.integers(0, 256, shape, dtype=np.uint8)
    im[0, :8, :] = 255                       # saturated borders: the largest row sums
    im[0, -8:, :] = 0
    im[-1, :, :24] = 255
    im[-1, :, -24:] = 255
    ref = oracle.gaussian_u8(im, sigma)
    assert np.array_equal(ops.gaussian_blur(im, sigma), ref)
    assert np.array_equal(ops.gaussian_blur(im, sigma, implementation="valu"), ref)
    for v in (0, 255):                       # extremes of the int16 row sums / i8 offsets
        c = np.full(shape, v, np.uint8)
        assert np.array_equal(ops.gaussian_blur(c, sigma), c)


@pytest.mark.parametrize("shape,sigma", [((2, 64, 64), 5.5), ((1, 100, 208), 6.0), ((2, 77, 128), 7.0),
                                         ((1, 65, 1936), 8.0), ((1, 130, 336), 8.3), ((2, 96, 192), 10.0),
                                         ((1, 64, 400), 10.7), ((1, 1080, 256), 9.0)])
def test_gaussian_u8_wide_radius_kernels(ops, oracle, shape, sigma):
    """radii 17 ... 32 (sigma up to 10.7): the dot4/dot2 kernel with a 32-column halo (two mirrored vectors per
    row end, up to 32 mirrored rows) -- the library's choice there, since the matrix-core kernel stops at
    radius 16 -- against the oracle and the generic two-pass kernels"""
    assert 16 < len(oracle.gauss_taps_q8(sigma)) // 2 <= 32
    rng = np.random.default_rng(shape[2] + int(sigma * 10))
    im = rng.integers(0, 256, shape, dtype=np.uint8)
    im[0, :8, :] = 255
    im[0, -8:, :] = 0
    im[-1, :, :40] = 255
    im[-1, :, -40:] = 255
    ref = oracle.gaussian_u8(im, sigma)
    assert np.array_equal(ops.gaussian_blur(im, sigma), ref)
    assert np.array_equal(ops.gaussian_blur(im, sigma, implementation="valu"), ref)
    assert np.array_equal(ops.gaussian_blur(im, sigma, implementation="generic"), ref)
    for v in (0, 255):
        c = np.full(shape, v, np.uint8)
        assert np.array_equal(ops.gaussian_blur(c, sigma), c)


@pytest.mark.parametrize("shape,sigma", [((2, 48, 64, 3), 2.0), ((1, 33, 112, 3), 5.0), ((3, 40, 80, 2), 1.0),
                                         ((1, 64, 96, 4), 3.3), ((1, 35, 1936, 3), 4.0),
                                         ((2, 40, 67), 5.3), ((1, 33, 1918), 5.0), ((2, 50, 101, 3), 2.0),
                                         ((1, 64, 49), 5.0), ((1, 32, 33, 4), 5.3)])
def test_gaussian_u8_planes_path(ops, oracle, shape, sigma):
    """frames the matrix-core kernel takes after a re-layout into single-channel planes: colour
    frames (every channel filtered on its own) and widths that are no multiple of 16 (planes padded
    with the reflected continuation of each row)"""
    color = len(shape) == 4
    rng = np.random.default_rng(sum(shape))
    im = rng.integers(0, 256, shape, dtype=np.uint8)
    im[0, :4] = 255
    im[..., -3:, :] = 255 if not color else im[..., -3:, :]
    ref = oracle.gaussian_u8(im, sigma)
    assert np.array_equal(ops.gaussian_blur(im, sigma, color=color), ref)
    assert np.array_equal(ops.gaussian_blur(im, sigma, color=color, implementation="generic"), ref)


def test_gaussian_u8_extremes_and_color(ops, oracle):
    for v in (0, 255, 77):
        c = np.full((2, 50, 70), v, np.uint8)
        assert np.array_equal(ops.gaussian_blur(c, 5.0), c)          # unity gain, exact
    yy, xx = np.mgrid[:64, :96]
    checker = (((yy + xx) % 2) * 255).astype(np.uint8)
    assert np.array_equal(ops.gaussian_blur(checker, 2.0), oracle.gaussian_u8(checker, 2.0))
    rng = np.random.default_rng(5)
    col = rng.integers(0, 256, (2, 40, 50, 3), dtype=np.uint8)
    assert np.array_equal(ops.gaussian_blur(col, 3.0, color=True), oracle.gaussian_u8(col, 3.0))


def test_gaussian_f32(ops, oracle, golden):
    f = golden["gin_f32"]
    assert np.array_equal(ops.gaussian_blur(f, 2.0), golden["gout_f32_2"])
    assert np.array_equal(ops.gaussian_blur(f, 9.0), golden["gout_f32_9"])
    rng = np.random.default_rng(8)
    c = rng.random((2, 33, 47, 3), dtype=np.float32)
    assert np.array_equal(ops.gaussian_blur(c, 4.0, color=True), oracle.gaussian_f32(c, 4.0))


@pytest.mark.parametrize("shape,color,sigma", [((2, 37, 53), False, 2.0), ((1, 64, 100, 3), True, 9.0),
                                                ((2, 20, 31, 3), True, 1.0), ((1, 5, 7), False, 9.0),
                                                ((1, 130, 257), False, 4.0), ((1, 90, 410, 3), True, 5.0),
                                                ((1, 33, 20, 3), True, 10.0), ((1, 40, 60), False, 12.0)])
def test_gaussian_f32_fast_and_generic_paths(ops, oracle, shape, color, sigma):
    rng = np.random.default_rng(int(sigma * 10) + shape[1])
    f = (rng.random(shape, dtype=np.float32) * 2 - 0.5).astype(np.float32)
    f.flat[::97] = 0.0
    f.flat[5::211] = -0.0
    ref = oracle.gaussian_f32(f, sigma) if not color else oracle.gaussian_f32(f, sigma)
    got = ops.gaussian_blur(f, sigma, color=color)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (shape, sigma)   # bit pattern


def test_gaussian_1080p_properties(ops, oracle):
    """full BASELINE size: parity on a crop-sized oracle sample + size-independent properties"""
    rng = np.random.default_rng(11)
    im = rng.integers(0, 256, (2, 1080, 1920), dtype=np.uint8)
    out = ops.gaussian_blur(im, 5.0)
    assert np.array_equal(out[:1], oracle.gaussian_u8(im[:1], 5.0))
    assert np.array_equal(out, ops.gaussian_blur(im, 5.0, implementation="generic"))
    # mean preservation (taps sum to 256; REFLECT_101 keeps mass up to border effects)
    assert abs(out.mean() - im.mean()) < 0.5
    # flipping commutes with the (symmetric) filter
    assert np.array_equal(ops.gaussian_blur(im[:, ::-1, ::-1].copy(), 5.0), out[:, ::-1, ::-1])


# ------------------------------------------------------------------------------ rotation
@pytest.mark.parametrize("shape,dtype,color", [((3, 37, 53), np.uint8, False), ((2, 64, 96), np.uint8, False),
                                               ((1, 33, 31, 3), np.uint8, True), ((2, 40, 70), np.int16, False),
                                               ((2, 31, 65), np.float32, False), ((1, 20, 45, 3), np.float32, True),
                                               ((45, 100), np.float64, False), ((1, 1, 7), np.uint8, False)])
def test_rot90_matches_numpy(ops, shape, dtype, color):
    """N4 FilterRotate (video/filters.py:339-344): np.rot90 in the image plane, every quarter turn"""
    rng = np.random.default_rng(sum(shape))
    a = (rng.random(shape) * 200).astype(dtype)
    axes = (0, 1) if a.ndim == (3 if color else 2) else (1, 2)
    for k in (0, 1, 2, 3, 5, -1):
        assert np.array_equal(ops.rot90(a, k, color=color), np.rot90(a, k, axes=axes)), (shape, k)


def test_filter_rotate_plumbing(ops):
    from video.io.memory import VideoMemory
    from video.filters import FilterRotate
    rng = np.random.default_rng(3)
    clip = rng.integers(0, 256, (5, 30, 44), dtype=np.uint8)
    for angle in (0, 90, 180, 270, 450):
        rot = FilterRotate(VideoMemory(clip.copy()), angle)
        k = (angle % 360) // 90
        assert rot.size == ((30, 44) if k & 1 else (44, 30))       # (width, height)
        assert len(rot) == 5
        for i, frame in enumerate(rot):
            assert np.array_equal(frame, np.rot90(clip[i], k))
    with pytest.raises(ValueError):
        FilterRotate(VideoMemory(clip), 45)


# -------------------------------------------------------------------------- background
def test_running_mean_bit_exact(ops, golden, oracle):
    for n in (1, 2, 8, 64, 256):
        fr = golden["bgframes_%d" % n]
        model = ops.BackgroundModel(fr.shape[1:], "mean")
        diff = model.process(fr)
        assert np.array_equal(model.state, golden["mean_%d" % n])       # float64, bit-exact
        assert np.array_equal(diff, oracle.bg_mean_u8(fr)[0])
        assert model.n_seen == n
    frc = golden["bgframes_color"]
    model = ops.BackgroundModel(frc.shape[1:], "mean")
    model.process(frc, want_diff=False)
    assert np.array_equal(model.state, golden["mean_color"])


def test_running_mean_split_batches_and_odd_sizes(ops, oracle):
    rng = np.random.default_rng(21)
    # px % 8 != 0 (scalar kernel), px % 16 == 0 (16 px per thread), px % 16 == 8 (8 px per thread)
    for shape in ((37, 5, 7), (20, 33, 41), (9, 16, 24), (11, 5, 8), (6, 3, 24)):
        fr = rng.integers(0, 256, shape, dtype=np.uint8)
        rd, rm = oracle.bg_mean_u8(fr)
        model = ops.BackgroundModel(shape[1:], "mean")
        d = np.concatenate([model.process(fr[:4]), model.process(fr[4:5]), model.process(fr[5:])])
        assert np.array_equal(d, rd) and np.array_equal(model.state, rm)
        # state export / import (checkpoint-resume of a shard)
        m2 = ops.BackgroundModel(shape[1:], "mean")
        m2.process(fr[:6], want_diff=False)
        m3 = ops.BackgroundModel(shape[1:], "mean")
        m3.set_state(m2.state, 6)
        assert np.array_equal(m3.process(fr[6:]), rd[6:])


def test_running_mean_division_free_path_is_bit_exact(ops, oracle):
    """the kernel replaces (mean*n)/(n+1) by a reciprocal + two FMA corrections and frame/(n+1) by
    a table: both must equal IEEE division bit for bit, for any state and frame count"""
    rng = np.random.default_rng(23)
    shape = (40, 16, 24)                      # px % 8 == 0 -> the division-free kernel
    fr = rng.integers(0, 256, shape, dtype=np.uint8)
    for n_seen in (0, 1, 2, 3, 7, 1000, 123457, 2 ** 31 - 100, 2 ** 40 + 12345):
        state = rng.random(shape[1:]) * 255   # arbitrary float64 significands
        state[0, :6] = [0.0, 255.0, 1e-300, 254.99999999999997, 1.0 / 3.0, 5e-324]
        rd, rm = oracle.bg_mean_u8(fr, mean=state, n_seen=n_seen)
        m = ops.BackgroundModel(shape[1:], "mean")
        m.set_state(state, n_seen)
        d = m.process(fr)
        assert np.array_equal(d, rd), n_seen
        assert np.array_equal(m.state, rm), n_seen


def test_ema_static_welford(ops, oracle, golden):
    rng = np.random.default_rng(22)
    fr = rng.integers(0, 256, (30, 12, 20), dtype=np.uint8)
    rd, rb = oracle.bg_ema_u8(fr, rate=0.05)
    m = ops.BackgroundModel(fr.shape[1:], "ema", rate=0.05)
    d = np.concatenate([m.process(fr[:7]), m.process(fr[7:])])
    assert np.array_equal(d, rd) and np.array_equal(m.state, rb)
    ff = rng.random((12, 9, 11, 3), dtype=np.float32)
    rd, rb = oracle.bg_ema_f32(ff, rate=0.02)
    m = ops.BackgroundModel(ff.shape[1:], "ema", rate=0.02, dtype=np.float32)
    d = m.process(ff)
    assert np.array_equal(d, rd) and np.array_equal(m.state, rb)
    bg = rng.random((12, 20)) * 255
    m = ops.BackgroundModel((12, 20), "static", background=bg)
    assert np.array_equal(m.process(fr), oracle.bg_static_u8(fr, bg))
    for n in (8, 64, 256):
        fr = golden["bgframes_%d" % n]
        mean, m2 = ops.welford(fr)
        assert np.array_equal(mean, golden["wmean_%d" % n])
        assert np.array_equal(np.sqrt(m2 / (n - 1)), golden["wstd_%d" % n])


# --------------------------------------------------------------------------- pointwise
def test_pointwise_ops(ops, oracle):
    rng = np.random.default_rng(31)
    for shape in ((1,), (255,), (7, 9), (3, 64, 100)):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        b = rng.integers(0, 256, shape, dtype=np.uint8)
        for t in (0, 20, 254, 255):
            assert np.array_equal(ops.threshold(a, t), oracle.threshold_u8(a, t))
        assert np.array_equal(ops.threshold(a, 100, 1), oracle.threshold_u8(a, 100, 1))
        assert np.array_equal(ops.time_difference(a, b), a.astype(np.int16) - b)
    c = rng.integers(0, 256, (5, 17, 23, 3), dtype=np.uint8)
    assert np.array_equal(ops.mono_mean(c), np.mean(c, axis=3).astype(np.uint8))
    a = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    fmin, fmax = 30, 200
    alpha = 255 / (fmax - fmin)
    exp = ((np.clip(a, fmin, fmax) - fmin) * alpha + 0).astype(np.uint8)
    assert np.array_equal(ops.normalize(a, fmin, fmax, alpha, 0), exp)


# -------------------------------------------------------------------------- morphology
def test_morphology_golden_and_oracle(ops, golden, oracle):
    for nm in ("img", "bin"):
        im = golden["morph_" + nm]
        for k in (3, 5, 7):
            assert np.array_equal(ops.morph(im, "erode", "rect", k), golden["erode_rect%d_%s" % (k, nm)])
            assert np.array_equal(ops.morph(im, "dilate", "rect", k), golden["dilate_rect%d_%s" % (k, nm)])
        assert np.array_equal(ops.morph(im, "erode", "cross", 3), golden["erode_cross3_" + nm])
        assert np.array_equal(ops.morph(im, "dilate", "cross", 3), golden["dilate_cross3_" + nm])
    rng = np.random.default_rng(41)
    for shape in ((2, 30, 45), (1, 64, 64), (3, 17, 100), (1, 5, 3), (2, 40, 97)):
        g = rng.integers(0, 256, shape, dtype=np.uint8)
        b = ((rng.random(shape) < 0.45) * 255).astype(np.uint8)
        for op in ("erode", "dilate"):
            o = oracle.ERODE if op == "erode" else oracle.DILATE
            for sh, so in (("rect", oracle.RECT), ("cross", oracle.CROSS), ("ellipse", oracle.ELLIPSE)):
                for k in (1, 2, 3, 4, 5, 9):
                    assert np.array_equal(ops.morph(g, op, sh, k), oracle.morph_u8(g, o, so, k)), (shape, op, sh, k)
                    # bit-packed kernel used inside the pipeline == u8 kernel on binary masks
                    assert np.array_equal(ops.morph(b, op, sh, k, implementation="bits"),
                                          oracle.morph_u8(b, o, so, k)), (shape, op, sh, k, "bits")
    # rows of whole dwords: the four-samples-per-thread kernels (packed 16-bit extremes, v_alignbyte windows),
    # large elements included; the other widths above ran the one-sample kernels
    for shape in ((1, 40, 4), (2, 33, 128), (1, 20, 1920), (1, 70, 8), (3, 64, 36)):
        g = rng.integers(0, 256, shape, dtype=np.uint8)
        g[0, :2] = 255
        g[-1, :, -5:] = 0
        for op, o in (("erode", oracle.ERODE), ("dilate", oracle.DILATE)):
            for sh, so in (("rect", oracle.RECT), ("cross", oracle.CROSS), ("ellipse", oracle.ELLIPSE)):
                for k in (3, 6, 15, 31):
                    assert np.array_equal(ops.morph(g, op, sh, k), oracle.morph_u8(g, o, so, k)), (shape, op, sh, k)


# --------------------------------------------------------------------------- labelling
@pytest.fixture(params=["frame-lds", "frame-staged", "frame-large", "chip-wide"])
def ccl_mode(request):
    """the labelling code paths of launch_ccl (va_ccl.hip), selected through the library's test hooks: one
    workgroup per frame with the forest in LDS (the default for large batches; spans read straight from
    global memory where rows are 16-byte aligned), the same with every row staged in LDS, the same kernel's
    large-frame mode (forest in the label image; forced by a tiny LDS budget), and the chip-wide
    multi-pass path that frames taller than the LDS row table take"""
    # (without a hook the library picks by a cost model of frame size and batch size)
    path, lds_runs = {"frame-lds": (2, 0), "frame-staged": (4, 0), "frame-large": (2, 7),
                      "chip-wide": (1, 0)}[request.param]
    from video import _hip
    _hip.check(_hip.lib().va_test_hook_labelling(path, lds_runs))
    yield request.param
    _hip.check(_hip.lib().va_test_hook_labelling(0, 0))


def test_label_golden_scipy_vectors(ops, golden, ccl_mode):
    for name in [str(n) for n in golden["mask_names"]]:
        m = golden["mask_" + name]
        for conn in (4, 8):
            lab, cnt = ops.label(m, conn)
            assert cnt == int(golden["count%d_%s" % (conn, name)]), (name, conn)
            assert lab.dtype == np.int32
            assert np.array_equal(lab, golden["labels%d_%s" % (conn, name)]), (name, conn)
    lab, cnt = ops.label(np.array([[0, 5, 0, 255]], np.uint8))
    assert cnt == 2 and lab.tolist() == [[0, 1, 0, 2]]


@pytest.mark.parametrize("h,w,density,seed", [(64, 64, 0.5, 1), (100, 333, 0.6, 2),
                                               (257, 129, 0.55, 3), (480, 640, 0.45, 4),
                                               (31, 2100, 0.5, 5), (300, 65, 0.7, 6),
                                               (128, 4160, 0.59, 7), (1, 500, 0.5, 8),
                                               (500, 1, 0.5, 9), (70, 3840, 0.5, 10),
                                               (45, 2500, 0.62, 11), (100, 1920, 0.55, 12),
                                               (64, 128, 0.5, 13), (50, 256, 0.6, 14),
                                               (1080, 1920, 0.02, 15)])
def test_label_random_vs_oracle(ops, oracle, h, w, density, seed, ccl_mode):
    rng = np.random.default_rng(seed)
    m = (rng.random((3, h, w)) < density).astype(np.uint8)
    for conn in (4, 8):
        lab, cnt = ops.label(m, conn)
        rl, rc = oracle.label_batch(m, conn)
        assert np.array_equal(cnt, rc), conn
        assert np.array_equal(lab, rl), conn


def test_label_adversarial(ops, oracle, ccl_mode):
    cases = {}
    cases["empty"] = np.zeros((40, 70), np.uint8)
    cases["full"] = np.ones((40, 70), np.uint8)
    cases["full_wide"] = np.ones((5, 4200), np.uint8)        # runs spanning > 64 words
    yy, xx = np.mgrid[:200, :300]
    cases["checker"] = ((yy + xx) % 2).astype(np.uint8)      # 30000 singletons (> 65535/2 ...)
    big = ((np.mgrid[:600, :512][0] + np.mgrid[:600, :512][1]) % 2).astype(np.uint8)
    cases["checker_big"] = big                               # 153600 components > 65535
    serp = np.zeros((201, 333), np.uint8)
    for y in range(0, 201, 2):
        serp[y, :] = 1
        if y + 1 < 201:
            serp[y + 1, 332 if (y // 2) % 2 == 0 else 0] = 1
    cases["serpentine"] = serp                               # one component, long dependency chain
    comb = np.zeros((300, 257), np.uint8)
    comb[:, ::2] = 1
    comb[299, :] = 1                                          # vertical teeth joined at the bottom
    cases["comb"] = comb
    ucomb = comb[::-1].copy()                                 # joined at the top
    cases["comb_top"] = ucomb
    frame = np.zeros((120, 200), np.uint8)
    frame[0, :] = frame[-1, :] = 1
    frame[:, 0] = frame[:, -1] = 1
    frame[40:80, 60:140] = 1
    cases["border_ring"] = frame
    tall = np.ones((3000, 33), np.uint8)                      # tall blob: deep union chains
    cases["tall"] = tall
    for name, m in cases.items():
        for conn in (4, 8):
            lab, cnt = ops.label(m, conn)
            rl, rc = oracle.label(m, conn)
            assert cnt == rc, (name, conn, cnt, rc)
            assert np.array_equal(lab, rl), (name, conn)


def test_label_1080p_vs_oracle_and_properties(ops, oracle):
    clip = _blob_clip(2, 1080, 1920, seed=3, nblobs=40, salt=0.002)
    m = (clip > 128).astype(np.uint8)
    lab, cnt = ops.label(m, 4)
    rl, rc = oracle.label_batch(m, 4)
    assert np.array_equal(cnt, rc) and np.array_equal(lab, rl)
    # size-independent properties: labels are 0 exactly on background, cover 1..L, idempotent
    assert np.array_equal(lab == 0, m == 0)
    for f in range(2):
        assert np.array_equal(np.unique(lab[f]), np.arange(cnt[f] + 1))
    lab2, cnt2 = ops.label(lab, 4)            # any non-zero is foreground
    assert np.array_equal(cnt2, cnt) and np.array_equal(lab2, lab)


def test_label_4k_frame(ops, oracle):
    rng = np.random.default_rng(77)
    m = (rng.random((1, 2160, 3840)) < 0.3).astype(np.uint8)
    m[0, 500:900, 1000:3000] = 1
    lab, cnt = ops.label(m, 4)
    rl, rc = oracle.label_batch(m, 4)
    assert np.array_equal(cnt, rc) and np.array_equal(lab, rl)
    # blobs instead of noise: few enough runs for the per-frame kernel (4 rows per wave at 4K)
    from video import _hip
    b = np.zeros((1, 2160, 3840), np.uint8)
    yy, xx = np.mgrid[:2160, :3840]
    for k in range(60):
        cx, cy, r = rng.uniform(0, 3840), rng.uniform(0, 2160), rng.uniform(10, 120)
        b[0][(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 1
    b[0][rng.random((2160, 3840)) < 0.0005] = 1
    rl, rc = oracle.label_batch(b, 8)
    _hip.check(_hip.lib().va_test_hook_labelling(2, 0))
    try:
        lab, cnt = ops.label(b, 8)
    finally:
        _hip.check(_hip.lib().va_test_hook_labelling(0, 0))
    assert np.array_equal(cnt, rc) and np.array_equal(lab, rl)


# ------------------------------------------------------------------ other dtypes (int16 / float32 inputs)
@pytest.mark.parametrize("dtype", [np.int16, np.float32, np.uint8])
def test_temporal_statistics_any_dtype(ops, oracle, dtype):
    """measure_mean / measure_mean_std on int16 (FilterTimeDifference's output) and float32 videos: the
    oracle (pinned to literal NumPy in the CPU suite) bit for bit, through ops and through the
    analysis.video functions on a VideoMemory"""
    from video.analysis.video import measure_mean, measure_mean_std
    from video.io.memory import VideoMemory
    rng = np.random.default_rng(17)
    if dtype == np.float32:
        frames = rng.normal(0.4, 0.3, (70, 33, 47)).astype(np.float32)
    elif dtype == np.int16:
        frames = rng.integers(-255, 256, (70, 33, 47)).astype(np.int16)
    else:
        frames = rng.integers(0, 256, (70, 33, 47)).astype(np.uint8)
    ref_mean = oracle.mean_any(frames)
    rm, rq = oracle.welford_any(frames)
    assert np.array_equal(ops.running_mean(frames), ref_mean)
    part = ops.running_mean(frames[:29])
    assert np.array_equal(ops.running_mean(frames[29:], part, 29), ref_mean)
    gm, gq = ops.welford(frames)
    assert np.array_equal(gm, rm) and np.array_equal(gq, rq)
    video = VideoMemory(frames)
    assert np.array_equal(measure_mean(video), ref_mean)
    mean, std = measure_mean_std(video)
    assert np.array_equal(mean, rm) and np.array_equal(std, np.sqrt(rq / (len(frames) - 1)))


def test_peaks_and_statistics_on_float_images(ops, oracle):
    """detect_peaks / get_image_statistics on float32 maps (the reference feeds them distance and
    correlation maps): comparisons in float, statistics on the truncated values"""
    from video.analysis.image import detect_peaks, get_image_statistics
    rng = np.random.default_rng(23)
    yy, xx = np.mgrid[:90, :130]
    img = np.zeros((90, 130), np.float32)
    for cx, cy, s, a in ((20, 30, 6, 3.5), (80, 40, 9, 7.25), (100, 70, 4, -2.0), (5, 85, 5, 4.0)):
        img += (a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2.0 * s * s))).astype(np.float32)
    img[img < 1e-3] = 0                                    # exact-zero background with plateaus
    img[40:44, 60:66] = 2.5                                # a plateau
    for plateaus in (True, False):
        assert np.array_equal(detect_peaks(img, plateaus), oracle.detect_peaks(img, plateaus)), plateaus
        noise = rng.normal(0, 1, (64, 77)).astype(np.float32)
        assert np.array_equal(detect_peaks(noise, plateaus), oracle.detect_peaks(noise, plateaus))
    big = (img * 40).astype(np.float32) + rng.uniform(-3, 3, img.shape).astype(np.float32)
    for kernel, ksize, prior, excl in (("box", 3, 10, False), ("ellipse", 4, 0, True), ("box", 1, 2.5, False)):
        gm, gv = get_image_statistics(big, kernel, ksize, prior=prior, exclude_center=excl)
        rm, rv = oracle.image_statistics(big, kernel, ksize, prior, excl)
        if float(prior).is_integer():
            assert np.array_equal(gm, rm) and np.array_equal(gv, rv), (kernel, ksize)
        else:
            assert np.allclose(gm, rm, rtol=1e-12, atol=1e-12) and np.allclose(gv, rv, rtol=1e-10, atol=1e-9)
    gm = get_image_statistics(big, "box", 2, ret_var=False)            # prior = image mean
    assert np.allclose(gm, oracle.image_statistics(big, "box", 2)[0], rtol=1e-12, atol=1e-10)


# ------------------------------------------------------------------ stats / largest region
def test_region_stats_and_moments(ops, oracle, golden):
    for name in [str(n) for n in golden["mask_names"]]:
        lab = golden["labels4_" + name]
        cnt = int(golden["count4_" + name])
        st = ops.region_stats(lab, max(cnt, 1))
        assert np.array_equal(st[:cnt, :14], oracle.region_stats(lab, cnt)[:, :14]), name
        assert np.array_equal(st[:cnt, 0], golden["areas4_" + name])
        assert np.array_equal(st[:cnt, :10], golden["lmoments4_" + name])
    rng = np.random.default_rng(51)
    m = (rng.random((2, 150, 517)) < 0.55).astype(np.uint8)
    lab, cnt = ops.label(m, 8)
    st = ops.region_stats(lab, int(cnt.max()))
    for f in range(2):
        assert np.array_equal(st[f, :cnt[f], :14], oracle.region_stats(lab[f], int(cnt[f]))[:, :14])


def test_largest_region_and_regionprops(ops, oracle, golden, ccl_mode):
    from video.analysis import image, regions
    for name in [str(n) for n in golden["mask_names"]]:
        m = golden["mask_" + name]
        if int(golden["count4_" + name]) == 0:
            with pytest.raises(ValueError):
                regions.get_largest_region(m)
            continue
        region, area = regions.get_largest_region(m, ret_area=True)
        assert region.dtype == bool
        assert np.array_equal(region.astype(np.uint8), golden["largest4_" + name]), name
        assert area == int(golden["areas4_" + name].max())
    assert regions.find_bounding_box(golden["mask_tie"][0:8]) == (2, 2, 6, 4)
    with pytest.raises(IndexError):
        regions.find_bounding_box(np.zeros((4, 4), np.uint8))
    m = golden["mask_border_blobs"]
    rp = image.regionprops(mask=m)
    raw = golden["moments_border_blobs"].astype(np.float64)
    cm = oracle.complete_moments(raw)
    assert rp.area == raw[0]
    assert rp.centroid == (raw[1] / raw[0], raw[2] / raw[0])
    for k, v in zip(("mu20", "mu11", "mu02", "mu30", "mu21", "mu12", "mu03", "nu20", "nu11", "nu02",
                     "nu30", "nu21", "nu12", "nu03"), cm):
        assert rp.moments[k] == v, k
    assert 0 <= rp.eccentricity <= 1 and rp.major_axis_length >= rp.minor_axis_length > 0


def test_contour_of_largest_region(ops, oracle, golden, ccl_mode):
    """A8 get_contour_from_largest_region: GPU (8-conn label roots + border following) vs the
    oracle's full Suzuki-Abe scanner, incl. nested components, ties and single pixels"""
    from video.analysis import regions
    masks = [golden["mask_" + str(n)] for n in golden["mask_names"]]
    rng = np.random.default_rng(61)
    for h, w, d in ((40, 50, 0.2), (64, 200, 0.45), (33, 70, 0.6), (200, 333, 0.52), (17, 4100, 0.4)):
        masks.append((rng.random((h, w)) < d).astype(np.uint8))
    yy, xx = np.mgrid[:300, :400]
    blobs = np.zeros((300, 400), np.uint8)
    for cx, cy, r in ((60, 70, 30), (200, 150, 55), (330, 230, 40), (200, 150, 20)):
        blobs |= ((xx - cx) ** 2 + (yy - cy) ** 2 <= r * r).astype(np.uint8)
    ring = blobs.copy()
    ring[(xx - 200) ** 2 + (yy - 150) ** 2 <= 35 ** 2] = 0            # hole ...
    ring[(xx - 200) ** 2 + (yy - 150) ** 2 <= 12 ** 2] = 1            # ... with a nested blob
    two = np.zeros((40, 60), np.uint8)
    two[5:15, 5:15] = 1
    two[25:35, 40:50] = 1                                             # equal areas: tie rule
    masks += [blobs, ring, two, np.ones((30, 40), np.uint8), np.eye(25, dtype=np.uint8)]
    for m in masks:
        if not m.any():
            with pytest.raises(RuntimeError):
                regions.get_contour_from_largest_region(m)
            continue
        ref, ref_area = oracle.get_contour_from_largest_region(m, ret_area=True)
        got, area = regions.get_contour_from_largest_region(m, ret_area=True)
        assert got.dtype == np.float64 and got.shape == ref.shape
        assert np.array_equal(got, ref) and area == ref_area
    pts, area, count = ops.largest_contour(two, max_points=2)        # truncated output
    assert len(pts) == 2 and count == 2


def test_small_stencils_peaks_thinning_statistics(ops, oracle):
    """N2: detect_peaks / mask_thinning (python method) / get_image_statistics"""
    from video.analysis import image
    rng = np.random.default_rng(71)
    for shape in ((1, 1), (1, 9), (7, 1), (40, 53), (128, 200), (5, 4), (1, 8), (33, 64), (2, 12)):   # (w % 4 == 0: four pixels per thread)
        img = rng.integers(0, 6, shape, dtype=np.uint8) * rng.integers(0, 2, shape, dtype=np.uint8)
        smooth = (rng.integers(0, 256, shape) // 32 * 32).astype(np.uint8)     # plateaus
        for im in (img, smooth, np.zeros(shape, np.uint8), np.full(shape, 9, np.uint8)):
            for plateaus in (True, False):
                got = image.detect_peaks(im, plateaus)
                assert got.dtype == bool
                assert np.array_equal(got, oracle.detect_peaks(im, plateaus)), (shape, plateaus)
    yy, xx = np.mgrid[:90, :130]
    blob = (((xx - 40) / 30.0) ** 2 + ((yy - 45) / 18.0) ** 2 <= 1) | (abs(xx - 95) + abs(yy - 40) < 25)
    for m in (blob.astype(np.uint8) * 255, blob.astype(np.uint8), np.zeros((10, 12), np.uint8),
              np.ones((9, 9), np.uint8), (rng.random((60, 70)) < 0.7).astype(np.uint8) * 255,
              # rows of whole dwords: erosion and the rest of a step as four-sample kernels
              blob[:88, :128].astype(np.uint8) * 255, (rng.random((64, 72)) < 0.8).astype(np.uint8) * 200,
              rng.integers(0, 256, (40, 64), dtype=np.uint8), np.full((12, 8), 255, np.uint8)):
        skel, it = ops.mask_thinning(m)
        rs, rit = oracle.mask_thinning(m)
        assert it == rit and np.array_equal(skel, rs)
        assert np.array_equal(image.mask_thinning(m), rs)
    img = rng.integers(0, 256, (50, 64), dtype=np.uint8)
    for kernel in ("box", "ellipse"):
        for ksize in (1, 2, 5):
            for excl in (False, True):
                for prior in (0, 128, None, 100.5):
                    mean, var = image.get_image_statistics(img, kernel, ksize, prior=prior,
                                                           exclude_center=excl)
                    rm, rv = oracle.image_statistics(img, kernel, ksize, prior, excl)
                    if prior in (0, 128):          # integer data: every window sum is exact
                        assert np.array_equal(mean, rm) and np.array_equal(var, rv)
                    else:                          # float prior: summation order may differ
                        assert np.allclose(mean, rm, rtol=1e-12, atol=1e-9)
                        assert np.allclose(var, rv, rtol=1e-9, atol=1e-6)
    assert image.get_image_statistics(img, "box", 3, ret_var=False).shape == img.shape


def test_image_statistics_box_tiles(ops, oracle):
    """box windows up to 31 x 31 run in one LDS-tiled kernel (64 x 64 tiles, sliding sums); larger ones and ellipses
    go through the row-prefix kernels: same integers, same float64 arithmetic -- against the oracle, and a batch of
    frames through the C ABI in one call"""
    from video import _hip
    from video.analysis import image
    rng = np.random.default_rng(77)
    for shape, ksize in (((64, 64), 3), ((65, 130), 7), ((200, 333), 15), ((130, 70), 16), ((31, 500), 10),
                         ((1, 1), 4), ((3, 200), 15), ((129, 64), 1)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        img[:2] = 255
        for excl in (False, True):
            mean, var = image.get_image_statistics(img, "box", ksize, prior=128, exclude_center=excl)
            rm, rv = oracle.image_statistics(img, "box", ksize, 128, excl)
            assert np.array_equal(mean, rm) and np.array_equal(var, rv), (shape, ksize, excl)
    img = rng.integers(0, 256, (90, 150), dtype=np.uint8)
    m1, v1 = image.get_image_statistics(img, "box", 6, prior=100.5)
    rm, rv = oracle.image_statistics(img, "box", 6, 100.5, False)
    assert np.allclose(m1, rm, rtol=1e-12, atol=1e-9) and np.allclose(v1, rv, rtol=1e-9, atol=1e-6)
    # three frames in one call
    clip = rng.integers(0, 256, (3, 70, 97), dtype=np.uint8)
    d, dm, dv = ops._upload(clip), ops._take(clip.size * 8), ops._take(clip.size * 8)
    try:
        _hip.check(_hip.lib().va_image_statistics_u8(d.ptr, dm.ptr, dv.ptr, 3, 70, 97, 0, 9, 0.0, 0, None))
        gm, gv = dm.download(clip.shape, np.float64), dv.download(clip.shape, np.float64)
    finally:
        for b in (d, dm, dv):
            ops._give(b)
    for f in range(3):
        rm, rv = oracle.image_statistics(clip[f], "box", 9, 0, False)
        assert np.array_equal(gm[f], rm) and np.array_equal(gv[f], rv)


# ------------------------------------------------------------------------ fused pipeline
def _engine(**kw):
    from video.engine import FrameEngine
    return FrameEngine(**kw)


def test_pipeline_golden_chain(golden):
    clip = golden["chain_clip"]
    eng = _engine(size=(64, 48), max_batch=16, background="mean", sigma=2.0, thresh=20,
                  morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4)
    out = eng.run(clip, want=("mask", "labels", "counts"))
    assert np.array_equal(out["mask"], golden["chain_mask"])
    assert np.array_equal(out["labels"], golden["chain_labels"])
    assert np.array_equal(out["counts"], golden["chain_counts"])
    state, n_seen = eng.get_background()
    assert n_seen == 16 and np.array_equal(state, golden["chain_mean"])
    eng.close()


@pytest.mark.parametrize("h,w,sigma,morph", [(480, 640, 2.0, 5), (270, 480, 5.0, 5),
                                              (101, 203, 3.0, 3), (64, 96, 5.0, 0)])
def test_pipeline_vs_oracle_batches(oracle, h, w, sigma, morph):
    clip = _blob_clip(12, h, w, seed=h + w, salt=0.002)
    steps = (("dilate", "rect", morph), ("erode", "rect", morph)) if morph else ()
    rm, rl, rc, rmean = oracle.chain_u8(clip, sigma, 20, morph_ksize=morph, connectivity=4)
    eng = _engine(size=(w, h), max_batch=8, background="mean", sigma=sigma, thresh=20,
                  morphology=steps, connectivity=4, max_labels=64)
    o1 = eng.run(clip[:8], want=("mask", "labels", "counts", "stats", "filtered"))
    o2 = eng.run(clip[8:], want=("mask", "labels", "counts"))           # state carries over
    assert np.array_equal(np.concatenate([o1["mask"], o2["mask"]]), rm)
    assert np.array_equal(np.concatenate([o1["labels"], o2["labels"]]), rl)
    assert np.array_equal(np.concatenate([o1["counts"], o2["counts"]]), rc)
    assert np.array_equal(eng.get_background()[0], rmean)
    diff, _ = oracle.bg_mean_u8(clip[:8])
    assert np.array_equal(o1["filtered"], oracle.gaussian_u8(diff, sigma))
    for f in range(8):
        c = min(int(rc[f]), 64)
        assert np.array_equal(o1["stats"][f, :c, :14], oracle.region_stats(rl[f], int(rc[f]))[:c, :14])
    eng.close()


def test_pipeline_1080p_full_chain(oracle):
    """BASELINE cfg#3 shape (2 frames checked against the oracle; properties on the rest)"""
    clip = _blob_clip(6, 1080, 1920, seed=3, nblobs=40, salt=0.002)
    eng = _engine(size=(1920, 1080), max_batch=6, background="mean", sigma=5.0, thresh=20,
                  morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4)
    out = eng.run(clip, want=("mask", "labels", "counts"))
    rm, rl, rc, _ = oracle.chain_u8(clip[:2], 5.0, 20, morph_ksize=5, connectivity=4)
    assert np.array_equal(out["mask"][:2], rm)
    assert np.array_equal(out["labels"][:2], rl)
    assert np.array_equal(out["counts"][:2], rc)
    assert np.array_equal(out["labels"] != 0, out["mask"] != 0)
    assert np.array_equal(out["labels"].reshape(6, -1).max(1), out["counts"])
    eng.close()


def test_pipeline_full_baseline_batch_properties(oracle):
    """BASELINE.json configs[2] at full size (256 x 1080p): the oracle cannot label 256 frames in
    seconds, so the full batch is checked through size-independent properties plus exact oracle
    comparisons on crops (background + blur are local up to the 15-pixel halo) and on two whole
    frames chosen from the middle and the end of the batch."""
    n, h, w = 256, 1080, 1920
    rng = np.random.default_rng(3)
    base = np.clip(rng.normal(100, 10, (h, w)), 0, 255)
    clip = np.empty((n, h, w), np.uint8)
    yy, xx = np.mgrid[:h, :w]
    cx0, cy0 = rng.uniform(0, w, 24), rng.uniform(0, h, 24)
    vx, vy = rng.uniform(-3, 3, 24), rng.uniform(-3, 3, 24)
    rad = rng.uniform(8, 60, 24)
    noise = rng.normal(0, 4, (8, h, w)).astype(np.float32)
    for t in range(n):
        f = base + noise[t % 8]
        for k in range(24):
            x0, y0, r = int(cx0[k] + vx[k] * t), int(cy0[k] + vy[k] * t), int(rad[k])
            ya, yb, xa, xb = max(0, y0 - r), min(h, y0 + r + 1), max(0, x0 - r), min(w, x0 + r + 1)
            if ya < yb and xa < xb:
                sub = f[ya:yb, xa:xb]
                sub[(xx[ya:yb, xa:xb] - x0) ** 2 + (yy[ya:yb, xa:xb] - y0) ** 2 <= r * r] += 60
        clip[t] = np.clip(f, 0, 255).astype(np.uint8)
    eng = _engine(size=(w, h), max_batch=n, background="mean", sigma=5.0, thresh=20,
                  morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4)
    out = eng.run(clip, want=("filtered", "mask", "labels", "counts"))
    state, n_seen = eng.get_background()
    eng.close()
    labels, mask, counts = out["labels"], out["mask"], out["counts"]
    # -- properties over the whole batch
    assert n_seen == n and counts.min() >= 1
    assert np.array_equal(labels != 0, mask != 0)                       # labels cover the mask
    assert np.array_equal(labels.reshape(n, -1).max(1), counts)         # 1..L, L = count
    first = np.array([np.flatnonzero(l.ravel())[0] for l in labels])    # SciPy order: label 1 first
    assert all(labels[f].ravel()[first[f]] == 1 for f in range(n))
    assert set(np.unique(mask)) <= {0, 255}
    # -- exact: running mean and emitted differences on crops (pixel-local)
    for (ya, xa) in ((0, 0), (500, 900), (1016, 1856)):
        crop = np.ascontiguousarray(clip[:, ya:ya + 64, xa:xa + 64])
        rd, rm = oracle.bg_mean_u8(crop)
        assert np.array_equal(state[ya:ya + 64, xa:xa + 64], rm)
        blur = oracle.gaussian_u8(rd, 5.0)          # exact away from the crop's own borders
        assert np.array_equal(out["filtered"][:, ya + 15:ya + 49, xa + 15:xa + 49],
                              blur[:, 15:49, 15:49])
    # -- exact: two whole frames through the rest of the chain (threshold, close, label)
    for f in (131, 255):
        m = oracle.threshold_u8(out["filtered"][f], 20)
        m = oracle.morph_u8(oracle.morph_u8(m, oracle.DILATE, oracle.RECT, 5), oracle.ERODE, oracle.RECT, 5)
        rl, rc = oracle.label(m, 4)
        assert np.array_equal(mask[f], m) and rc == counts[f] and np.array_equal(labels[f], rl)


def test_pipeline_variants(oracle):
    clip = _blob_clip(5, 72, 100, seed=9)
    # no background model, 8-connectivity, open instead of close, maxval 1
    eng = _engine(size=(100, 72), max_batch=5, sigma=2.0, thresh=110, maxval=1,
                  morphology=(("erode", "cross", 3), ("dilate", "cross", 3)), connectivity=8)
    out = eng.run(clip)
    m = oracle.threshold_u8(oracle.gaussian_u8(clip, 2.0), 110, 1)
    m = oracle.morph_u8(oracle.morph_u8(m, oracle.ERODE, oracle.CROSS, 3), oracle.DILATE, oracle.CROSS, 3)
    rl, rc = oracle.label_batch(m, 8)
    assert np.array_equal(out["mask"], m) and np.array_equal(out["labels"], rl)
    assert np.array_equal(out["counts"], rc)
    eng.close()
    # blur only, colour frames, EMA background on float32 (cfg#5 shape family)
    rng = np.random.default_rng(4)
    ff = rng.random((4, 30, 44, 3), dtype=np.float32)
    eng = _engine(size=(44, 30), channels=3, dtype=np.float32, max_batch=4, background="ema",
                  bg_rate=0.02, sigma=3.0)
    out = eng.run(ff, want=("filtered",))
    d, _ = oracle.bg_ema_f32(ff, rate=0.02)
    assert np.array_equal(out["filtered"], oracle.gaussian_f32(d, 3.0))
    eng.close()
    # static background
    bg = rng.random((72, 100)) * 200
    eng = _engine(size=(100, 72), max_batch=5, background="static", sigma=0, thresh=30)
    eng.set_background(bg)
    out = eng.run(clip, want=("mask",))
    assert np.array_equal(out["mask"], oracle.threshold_u8(oracle.bg_static_u8(clip, bg), 30))
    eng.close()


@pytest.mark.parametrize("h,w", [(70, 96), (45, 4160), (131, 2000)])
def test_pipeline_morphology_chains(oracle, h, w):
    """every kernel behind FilterMorphology inside the pipeline: register-streaming (rect, one or
    two ops, several word chunks per row), LDS-fused (other shapes / longer chains)"""
    rng = np.random.default_rng(h * w)
    img = (rng.random((3, h, w)) < 0.35) * np.uint8(200)
    img[:, :, -3:] = 200                       # foreground touching the right frame border
    img[:, 0, :] = 200
    chains = [[("dilate", "rect", 5)], [("erode", "rect", 3)], [("erode", "rect", 7)],
              [("erode", "rect", 3), ("dilate", "rect", 7)], [("dilate", "rect", 7), ("erode", "rect", 5)],
              [("erode", "rect", 5), ("erode", "rect", 3)], [("dilate", "rect", 3), ("dilate", "rect", 3)],
              [("dilate", "ellipse", 5), ("erode", "cross", 3)],
              [("dilate", "rect", 3), ("erode", "rect", 3), ("dilate", "rect", 5)],
              [("erode", "rect", 9)], [("dilate", "rect", 4), ("erode", "rect", 4)]]
    codes = {"erode": oracle.ERODE, "dilate": oracle.DILATE, "rect": oracle.RECT, "cross": oracle.CROSS,
             "ellipse": oracle.ELLIPSE}
    for chain in chains:
        eng = _engine(size=(w, h), max_batch=3, thresh=100, morphology=chain, connectivity=8)
        out = eng.run(img, want=("mask", "labels", "counts"))
        m = oracle.threshold_u8(img, 100)
        for op, shape, k in chain:
            m = oracle.morph_u8(m, codes[op], codes[shape], k)
        rl, rc = oracle.label_batch(m, 8)
        assert np.array_equal(out["mask"], m), chain
        assert np.array_equal(out["labels"], rl) and np.array_equal(out["counts"], rc), chain
        # the mask alone (no labelling stage behind the morphology kernels)
        assert np.array_equal(eng.run(img, want=("mask",))["mask"], m), chain
        eng.close()


def test_streamed_engine_overlap_matches_synchronous_runs(oracle):
    """N3: uploads / chain / downloads on three streams with pinned buffers give exactly the
    results of the synchronous path, in submission order, with the background state carried"""
    from video.streaming import StreamedEngine
    clip = _blob_clip(23, 96, 128, seed=12, salt=0.002)
    kw = dict(size=(128, 96), max_batch=4, background="mean", sigma=2.0, thresh=20,
              morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4, max_labels=32)
    rm, rl, rc, _ = oracle.chain_u8(clip, 2.0, 20, morph_ksize=5, connectivity=4)
    eng = _engine(**kw)
    results = []
    with StreamedEngine(eng, want=("mask", "labels", "counts", "stats"), slots=3) as s:
        for a in range(0, len(clip), 4):
            results += s.submit(clip[a:a + 4], tag=a)
        results += s.drain()
    assert [r["tag"] for r in results] == list(range(0, 23, 4))
    assert np.array_equal(np.concatenate([r["mask"] for r in results]), rm)
    assert np.array_equal(np.concatenate([r["labels"] for r in results]), rl)
    assert np.array_equal(np.concatenate([r["counts"] for r in results]), rc)
    st = np.concatenate([r["stats"] for r in results])
    for f in (0, 7, 22):
        c = min(int(rc[f]), 32)
        assert np.array_equal(st[f, :c, :14], oracle.region_stats(rl[f], int(rc[f]))[:c, :14])
    eng.close()


def test_pipeline_error_paths():
    from video import _hip
    with pytest.raises(_hip.HipError):
        _engine(size=(64, 48), sigma=100.0)                 # > 255 taps
    with pytest.raises(_hip.HipError):
        _engine(size=(64, 48), channels=3, thresh=10)       # masks need 1 channel
    with pytest.raises(_hip.HipError):
        _engine(size=(64, 48), connectivity=4)              # labelling needs a threshold
    eng = _engine(size=(64, 48), max_batch=2, thresh=10, connectivity=4)
    with pytest.raises(ValueError):
        eng.run(np.zeros((3, 48, 64), np.uint8))
    with pytest.raises(ValueError):
        eng.run(np.zeros((1, 48, 65), np.uint8))
    eng.close()


# --------------------------------------------------------------------- drop-in filter API
def test_filter_classes_match_fused_chain_and_oracle(oracle):
    from video.analysis.video import measure_mean, measure_mean_std
    from video.filters import (FilterAnalysisChain, FilterBackground, FilterBlur, FilterMorphology,
                               FilterThreshold, FilterTimeDifference)
    from video.io.memory import VideoMemory
    clip = _blob_clip(10, 60, 80, seed=2, salt=0.002)
    src = VideoMemory(clip)
    chain = FilterMorphology(FilterThreshold(FilterBlur(FilterBackground(src), 2), 20), "close", 5)
    frames = [np.array(f) for f in chain]
    rm, rl, rc, _ = oracle.chain_u8(clip, 2.0, 20, morph_ksize=5, connectivity=4)
    assert np.array_equal(np.stack(frames), rm)
    assert np.array_equal(chain[7], rm[7]) and np.array_equal(chain[3], rm[3])   # random access
    fused = FilterAnalysisChain(VideoMemory(clip), sigma=2.0, threshold=20, batch=4, output="labels")
    got = []
    counts = []
    for f in fused:
        got.append(np.array(f))
        counts.append(fused.last_count)
    assert np.array_equal(np.stack(got), rl) and counts == rc.tolist()
    assert np.array_equal(fused[6], rl[6]) and fused.last_count == rc[6]
    fused.close()
    assert np.array_equal(measure_mean(src), oracle.measure_mean_numpy(clip))
    mean, std = measure_mean_std(src)
    rmean, rstd = oracle.measure_mean_std_numpy(clip)
    assert np.array_equal(mean, rmean) and np.array_equal(std, rstd)
    td = list(FilterTimeDifference(VideoMemory(clip)))
    assert len(td) == 9
    assert all(np.array_equal(td[k], clip[k + 1].astype(np.int16) - clip[k]) for k in range(9))


def test_rccl_gather_counts_single_rank():
    """the C ABI's own RCCL path (va_comm_* / va_gather_counts), world size 1 on this box; the
    multi-rank layout is covered on CPU in tests/test_sharding.py"""
    import ctypes as C
    from video import _hip
    from video._hip import DeviceBuffer, check
    L = _hip.lib()
    uid = (C.c_uint8 * 128)()
    check(L.va_comm_unique_id(uid))
    comm = C.c_void_p()
    check(L.va_comm_init(C.byref(comm), 1, 0, uid))
    counts = np.arange(17, dtype=np.int32) * 3
    send, recv = DeviceBuffer.from_array(counts), DeviceBuffer(counts.nbytes)
    check(L.va_gather_counts(comm, send.ptr, recv.ptr, len(counts), None))
    check(L.va_stream_sync(None))
    assert np.array_equal(recv.download(counts.shape, np.int32), counts)
    check(L.va_comm_destroy(comm))
    send.free()
    recv.free()


def test_two_streams_do_not_share_scratch(oracle):
    """the stand-alone entry points lease their device scratch per call on the call's own stream
    (va_api.hip ScratchLease): interleaved calls of different sizes on two streams -- the
    reference's VideoPreprocessor worker threads, video/io/parallel.py:398-400 -- must not see
    each other's intermediates (the generic Gaussian keeps its u16 row pass there; the bit-packed
    morphology both of its masks)"""
    import ctypes as C
    import threading
    from video import _hip
    from video._hip import DeviceBuffer, check
    L = _hip.lib()
    rng = np.random.default_rng(123)
    a = rng.integers(0, 256, (6, 301, 517), dtype=np.uint8)          # generic path (odd width)
    b = rng.integers(0, 256, (2, 97, 133, 3), dtype=np.uint8)
    m = ((rng.random((4, 120, 200)) < 0.4) * 255).astype(np.uint8)
    ref_a, ref_b = oracle.gaussian_u8(a, 3.0), oracle.gaussian_u8(b, 2.0)
    ref_m = oracle.morph_u8(m, oracle.DILATE, oracle.RECT, 5)
    s1, s2 = C.c_void_p(), C.c_void_p()
    check(L.va_stream_create(C.byref(s1)))
    check(L.va_stream_create(C.byref(s2)))
    da, db, dm = DeviceBuffer.from_array(a), DeviceBuffer.from_array(b), DeviceBuffer.from_array(m)
    oa, ob, om = DeviceBuffer(a.nbytes), DeviceBuffer(b.nbytes), DeviceBuffer(m.nbytes)
    errors = []

    def worker(stream, jobs):
        try:
            for _ in range(20):
                for fn, args in jobs:
                    check(fn(*args, stream))
            check(L.va_stream_sync(stream))
        except Exception as e:          # pragma: no cover
            errors.append(e)

    t1 = threading.Thread(target=worker, args=(s1, [(L.va_gaussian_u8_generic, (da.ptr, oa.ptr, 6, 301, 517, 1, 3.0))]))
    t2 = threading.Thread(target=worker, args=(s2, [
        (L.va_gaussian_u8_generic, (db.ptr, ob.ptr, 2, 97, 133, 3, 2.0)),
        (L.va_morph_bits_u8, (dm.ptr, om.ptr, 4, 120, 200, _hip.MORPH_DILATE, _hip.SHAPE_RECT, 5))]))
    t1.start(); t2.start(); t1.join(); t2.join()
    assert not errors, errors
    assert np.array_equal(oa.download(a.shape, np.uint8), ref_a)
    assert np.array_equal(ob.download(b.shape, np.uint8), ref_b)
    assert np.array_equal(om.download(m.shape, np.uint8), ref_m)
    for buf in (da, db, dm, oa, ob, om):
        buf.free()
    check(L.va_stream_destroy(s1))
    check(L.va_stream_destroy(s2))


def test_pipeline_counts_only_skips_paint(oracle, ccl_mode):
    """counts without labels/stats: the label image is never painted (va_pipeline_run passes
    paint=false), on every labelling path, and the counts still match the oracle"""
    n = 20 if ccl_mode == "chip-wide" else 100
    clip = _blob_clip(n, 90, 160, seed=31, salt=0.003)
    _, _, rc, _ = oracle.chain_u8(clip, 2.0, 20, morph_ksize=3, connectivity=8, want_labels=True)
    eng = _engine(size=(160, 90), max_batch=n, background="mean", sigma=2.0, thresh=20,
                  morphology=(("dilate", "rect", 3), ("erode", "rect", 3)), connectivity=8, max_labels=8)
    eng.profile(True)
    out = eng.run(clip, want=("counts",))
    stages = eng.stage_times()
    assert np.array_equal(out["counts"], rc)
    assert "ccl_paint" not in stages and ("ccl_frame" in stages or "ccl_rank" in stages)
    eng.set_background(None, 0)
    out = eng.run(clip, want=("counts", "stats"))          # stats come from the paint pass
    assert np.array_equal(out["counts"], rc)
    assert "ccl_paint" in eng.stage_times()
    eng.close()


def test_contour_moments_match_oracle_bit_for_bit(ops, oracle):
    """A9: cv2.moments(contour) on the GPU (va_contour_moments) against the oracle's restatement
    of OpenCV's contourMoments -- same float64 bits for int32 and float32 points, either
    orientation, degenerate contours, long contours at 1080p/4K coordinates (terms above 2^53)"""
    from video.analysis import image, regions, shapes
    rng = np.random.default_rng(17)
    cases = [np.array([[0, 0], [4, 0], [4, 4], [0, 4]], np.int32),
             np.array([[0, 0], [4, 0], [4, 4], [0, 4]], np.int32)[::-1],
             np.array([[3, 3]], np.int32), np.array([[1, 1], [5, 1]], np.int32),
             np.array([[0, 0], [6, 0], [0, 3]], np.float32),
             (rng.random((500, 2)) * (3840, 2160)).astype(np.float32),
             rng.integers(0, 3840, (3000, 2)).astype(np.int32),
             np.array([[[10, 20]], [[10, 29]], [[49, 29]], [[49, 20]]], np.float64)]
    th = np.linspace(0, 2 * np.pi, 700, endpoint=False)
    cases.append(np.stack([1900 + 1800 * np.cos(th), 1080 + 1000 * np.sin(th)], 1).round().astype(np.int32))
    for c in cases:
        ref = oracle.contour_moments(c)
        got = image.contour_moments(c)
        assert set(got) == set(ref)
        for k, v in ref.items():
            assert np.float64(got[k]).tobytes() == np.float64(v).tobytes(), (k, got[k], v, c.shape)
    # through the analysis API: contour of the largest region -> regionprops / Polygon
    yy, xx = np.mgrid[:300, :400]
    m = (((xx - 150) / 90.0) ** 2 + ((yy - 140) / 60.0) ** 2 <= 1).astype(np.uint8)
    m[200:260, 300:380] = 1
    contour = regions.get_contour_from_largest_region(m)
    ref = oracle.contour_moments(np.asarray(contour, np.float32))
    props = image.regionprops(contour=contour)
    poly = shapes.Polygon(contour)
    for k, v in ref.items():
        assert props.moments[k] == v and poly.moments[k] == v, k
    assert props.area == ref["m00"] and 0 < poly.eccentricity < 1
    rp = image.regionprops(mask=regions.get_largest_region(m))         # raster moments of the same blob
    assert abs(props.centroid[0] - rp.centroid[0]) < 0.5 and abs(props.orientation - rp.orientation) < 0.05
    # the fused form: moments from the device-resident points of va_largest_contour
    pts, area, count, mom = ops.largest_contour(m, moments=True)
    assert count == 2 and area == ref["m00"]
    assert np.array_equal(mom, [ref[k] for k in ("m00", "m10", "m01", "m20", "m11", "m02", "m30", "m21", "m12", "m03")])


def test_background_property_is_the_same_contracted_or_not(oracle):
    """FilterBackground.background after k frames read through Threshold(Blur(Background(v))): the
    fused chain works a batch ahead of the reader, the property still reports the state after the
    frames READ so far -- equal to the per-filter path and to the oracle, mid-batch and after a seek"""
    from video import filters as F
    from video.io.memory import VideoMemory
    clip = _blob_clip(50, 48, 64, seed=31)
    states = {}
    for contract in (True, False):
        old = F._GpuStage.contract
        F._GpuStage.contract = contract
        try:
            bg = F.FilterBackground(VideoMemory(clip))
            thr = F.FilterThreshold(F.FilterBlur(bg, 2), 20)
            assert (thr._runner() is not None) == contract
            got = []
            for k, _f in enumerate(thr):
                if k in (0, 6, 31, 32, 40):
                    got.append(np.array(bg.background))
            thr[10]                                    # random access: the state follows the reader
            got.append(np.array(bg.background))
            states[contract] = got
            thr.close()
        finally:
            F._GpuStage.contract = old
    for k, a, b in zip((1, 7, 32, 33, 41, 11), states[True], states[False]):
        _, ref = oracle.bg_mean_u8(clip[:k])
        assert np.array_equal(a, ref), "contracted, after %d frames" % k
        assert np.array_equal(b, ref), "per filter, after %d frames" % k


def test_pooled_ops_from_several_threads(ops, oracle):
    """VideoPreprocessor runs GPU callables from one worker thread per function: the device-buffer pool
    of video.ops is shared between them (same frame size -> same size class)"""
    import threading
    rng = np.random.default_rng(8)
    frames = rng.integers(0, 256, (24, 96, 128), dtype=np.uint8)
    ref_blur = oracle.gaussian_u8(frames, 2.0)
    ref_thr = oracle.threshold_u8(frames, 128)
    ops.pool_clear()
    errors = []

    def work(fn, ref):
        try:
            for _ in range(3):
                for k in range(frames.shape[0]):
                    if not np.array_equal(fn(frames[k]), ref[k]):
                        errors.append("mismatch at frame %d" % k)
        except Exception as e:                        # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(lambda f: ops.gaussian_blur(f, 2.0), ref_blur)),
               threading.Thread(target=work, args=(lambda f: ops.threshold(f, 128), ref_thr)),
               threading.Thread(target=work, args=(lambda f: ops.gaussian_blur(f, 2.0), ref_blur))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    ops.pool_clear()


def test_gpu_filter_chains_contract_into_one_engine(oracle):
    """consecutive GPU filters collapse into one FrameEngine with batched prefetch (the reference's
    own idiom for consecutive crops, video/filters.py:209-215 there): identical frames with and
    without the contraction, listeners of every stage still notified, random access, partial
    chains, and chains that must NOT contract (a listener on a stage whose frames the fused
    chain never materialises)"""
    from video import filters as F
    from video.io.memory import VideoMemory
    clip = _blob_clip(70, 72, 96, seed=12, salt=0.003)
    rm, _, _, _ = oracle.chain_u8(clip, 2.0, 20, morph_ksize=5, connectivity=4)
    rdiff, _ = oracle.bg_mean_u8(clip)
    rblur = oracle.gaussian_u8(rdiff, 2.0)

    def build(listen=()):
        src = VideoMemory(clip)
        bg = F.FilterBackground(src)
        blur = F.FilterBlur(bg, 2)
        thr = F.FilterThreshold(blur, 20)
        mor = F.FilterMorphology(thr, "close", 5)
        seen = {}
        for name, node in (("bg", bg), ("blur", blur), ("thr", thr), ("mor", mor)):
            if name in listen:
                seen[name] = []
                node.register_listener(lambda f, lst=seen[name]: lst.append(np.array(f)))
        return mor, blur, thr, bg, seen

    # contracted: one engine, blur-stage and final listeners see their frames
    mor, blur, thr, bg, seen = build(listen=("blur", "mor"))
    assert mor._runner() is not None and mor._runner().engine.description.count("morph=2")
    frames = [np.array(f) for f in mor]
    assert np.array_equal(np.stack(frames), rm)
    assert np.array_equal(np.stack(seen["mor"]), rm) and np.array_equal(np.stack(seen["blur"]), rblur)
    assert np.array_equal(mor[41], rm[41]) and np.array_equal(mor[3], rm[3]) and np.array_equal(mor[-1], rm[-1])
    mor.close()
    # intermediate stages read on their own contract over what lies below them
    mor, blur, thr, bg, _ = build()
    assert np.array_equal(np.stack([np.array(f) for f in blur]), rblur)
    assert np.array_equal(np.stack([np.array(f) for f in thr]), oracle.threshold_u8(rblur, 20))
    assert bg._runner() is None and np.array_equal(np.stack([np.array(f) for f in bg]), rdiff)
    mor.close()
    # a listener on the threshold stage (mask before the morphology): no contraction, same frames
    mor, blur, thr, bg, seen = build(listen=("thr",))
    assert mor._runner() is None
    assert np.array_equal(np.stack([np.array(f) for f in mor]), rm)
    assert np.array_equal(np.stack(seen["thr"]), oracle.threshold_u8(rblur, 20))
    # the switch: per-filter path on request, identical frames
    F._GpuStage.contract = False
    try:
        mor2 = build()[0]
        assert mor2._runner() is None
        assert np.array_equal(np.stack([np.array(f) for f in mor2[::7]]), rm[::7])
    finally:
        F._GpuStage.contract = True
    # static background + threshold without a blur; open instead of close; maxval != 255 stays apart
    static = np.full(clip.shape[1:], 100.0)
    t = F.FilterThreshold(F.FilterBackground(VideoMemory(clip), mode="static", background=static), 25)
    assert t._runner() is not None
    assert np.array_equal(np.stack([np.array(f) for f in t]),
                          oracle.threshold_u8(oracle.bg_static_u8(clip, static), 25))
    t1 = F.FilterMorphology(F.FilterThreshold(F.FilterBlur(VideoMemory(clip), 2), 110, maxval=1), "open", 3)
    assert t1._runner() is None                      # maxval 1 under a morphology: per-filter path
    m1 = oracle.threshold_u8(oracle.gaussian_u8(clip, 2.0), 110, 1)
    m1 = oracle.morph_u8(oracle.morph_u8(m1, oracle.ERODE, oracle.RECT, 3), oracle.DILATE, oracle.RECT, 3)
    assert np.array_equal(np.stack([np.array(f) for f in t1]), m1)


def test_contracted_chain_runs_one_engine_pass_per_batch_1080p():
    """the four-filter chain over 256 x 1080p frames, read frame by frame, must do what
    FilterAnalysisChain does: ONE engine pass per 32 frames and no per-filter kernels -- checked by
    counting engine passes (a property of the code, not of the clock; the timings of the three ways
    to run the chain are printed for DESIGN.md and measured properly by tools/bench_next_tier.py)"""
    import time
    from video import filters as F
    from video.engine import FrameEngine
    from video.io.memory import VideoMemory
    clip = _blob_clip(32, 1080, 1920, seed=5, nblobs=20, salt=0.002)
    clip = np.concatenate([clip] * 8)                      # 256 frames
    passes = []
    real_run = FrameEngine.run

    def counting_run(self, frames, want=("mask", "labels", "counts")):
        passes.append(len(frames))
        return real_run(self, frames, want)

    def run(video):
        t0 = time.perf_counter()
        acc = 0
        for f in video:
            acc += int(f[540, 960])
        return time.perf_counter() - t0, acc

    FrameEngine.run = counting_run
    try:
        fused = F.FilterAnalysisChain(VideoMemory(clip), sigma=5.0, threshold=20, connectivity=0, batch=32)
        run(fused)                                             # warm-up (engine creation, first touch)
        del passes[:]
        t_fused, a = run(fused)
        passes_fused = list(passes)
        chain = F.FilterMorphology(F.FilterThreshold(F.FilterBlur(F.FilterBackground(VideoMemory(clip)), 5), 20), "close", 5)
        run(chain)
        del passes[:]
        t_chain, b = run(chain)
        passes_chain = list(passes)
    finally:
        FrameEngine.run = real_run
    F._GpuStage.contract = False
    try:
        plain = F.FilterMorphology(F.FilterThreshold(F.FilterBlur(F.FilterBackground(VideoMemory(clip[:32])), 5), 20), "close", 5)
        t_plain, _ = run(plain)
    finally:
        F._GpuStage.contract = True
    print("\n[chain contraction] 256 x 1080p: FilterAnalysisChain %.3f s, four contracted filters %.3f s, "
          "uncontracted %.3f s per 256 frames" % (t_fused, t_chain, t_plain * 8))
    assert a == b
    assert passes_fused == [32] * 8 and passes_chain == [32] * 8


@pytest.mark.parametrize("mode", ["nearest", "linear", "cubic", "area", "lanczos"])
def test_resize_matches_oracle(ops, oracle, mode):
    """N4 FilterResize / cv2.resize restatement: every mode, shrinking and growing, integer and
    fractional factors, 1 and 3 channels, batches -- bit for bit against the oracle"""
    rng = np.random.default_rng(len(mode))
    img = rng.integers(0, 256, (2, 90, 120), dtype=np.uint8)
    col = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    img[0, :10] = 255
    img[1, :, -7:] = 0
    for size in ((60, 45), (40, 30), (30, 18), (240, 180), (77, 51), (121, 91), (119, 90), (7, 5), (1, 1), (300, 31)):
        assert np.array_equal(ops.resize(img, size, mode), oracle.resize_u8(img, size, mode)), (mode, size)
        assert np.array_equal(ops.resize(col, size, mode, color=True),
                              oracle.resize_u8(col, size, mode, layout="hwc")), (mode, size, "color")
    big = rng.integers(0, 256, (1, 1080, 1920), dtype=np.uint8)
    for size in ((960, 540), (640, 360), (1280, 720), (2560, 1440)):
        assert np.array_equal(ops.resize(big, size, mode), oracle.resize_u8(big, size, mode)), (mode, size)
    flat = np.full((33, 47), 201, np.uint8)                          # unity gain (Lanczos: its 8 quantised
    got = ops.resize(flat, (80, 21), mode)                           # taps need not sum to 2048 exactly)
    assert np.abs(got.astype(int) - 201).max() <= (1 if mode == "lanczos" else 0)
    assert np.array_equal(ops.resize(img, (120, 90), mode), img)     # same size: every mode is the identity


@pytest.mark.parametrize("mode", ["nearest", "linear", "cubic", "area", "lanczos"])
def test_resize_f32_matches_oracle(ops, oracle, mode):
    """FilterResize on float32 frames (the reference resizes whatever dtype the video has,
    video/filters.py:310-314): the float instantiations of the same algorithms, bit patterns equal
    to the oracle's (same products, same order of additions, no contraction)"""
    rng = np.random.default_rng(40 + len(mode))
    img = rng.normal(0.5, 0.3, (2, 90, 120)).astype(np.float32)
    col = rng.uniform(-10, 10, (60, 80, 3)).astype(np.float32)
    img[0, :10] = 1e6
    for size in ((60, 45), (40, 30), (30, 18), (240, 180), (77, 51), (121, 91), (7, 5), (1, 1), (300, 31)):
        a, b = ops.resize(img, size, mode), oracle.resize_f32(img, size, mode)
        assert a.dtype == np.float32 and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (mode, size)
        a, b = ops.resize(col, size, mode, color=True), oracle.resize_f32(col, size, mode, layout="hwc")
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (mode, size, "color")
    big = rng.uniform(0, 1, (1, 540, 960)).astype(np.float32)
    for size in ((480, 270), (1280, 720)):
        assert np.array_equal(ops.resize(big, size, mode).view(np.uint32),
                              oracle.resize_f32(big, size, mode).view(np.uint32)), (mode, size)
    assert np.array_equal(ops.resize(img, (120, 90), mode), img)


def test_resize_more_frames_than_one_grid_dimension(ops, oracle):
    """more than 65535 frames in one call: the launch goes out in pieces of gridDim.z frames"""
    rng = np.random.default_rng(2)
    tiny = rng.integers(0, 256, (65540, 4, 6), dtype=np.uint8)
    got = ops.resize(tiny, (3, 2), "linear")
    ref = oracle.resize_u8(tiny, (3, 2), "linear")
    assert got.shape == (65540, 2, 3) and np.array_equal(got, ref)


def test_filter_resize_plumbing(ops, oracle):
    from video.filters import FilterResize
    from video.io.memory import VideoMemory
    clip = _blob_clip(5, 60, 90, seed=8)
    src = VideoMemory(clip)
    half = FilterResize(src, 0.5)                                    # 'auto' on a shrink: area
    assert half.size == (45, 30) and half.interpolation == "area"
    assert np.array_equal(np.stack(list(half)), oracle.resize_u8(clip, (45, 30), "area"))
    up = FilterResize(src, (135, 90))                                # 'auto' on a growth: cubic
    assert up.interpolation == "cubic" and np.array_equal(up[2], oracle.resize_u8(clip[2], (135, 90), "cubic"))
    same = FilterResize(src, 1)
    assert same.interpolation is None and np.array_equal(same[1], clip[1])
    twice = FilterResize(FilterResize(src, 0.5), (31, 21), "linear", even_dimensions=True)
    assert twice._source is src and twice.size == (32, 22)           # contracted; even dimensions
    assert np.array_equal(twice[0], oracle.resize_u8(clip[0], (32, 22), "linear"))
    with pytest.raises(ValueError):
        FilterResize(src, 0.5, "bogus")
    lz = FilterResize(src, (135, 90), "lanczos")                     # INTER_LANCZOS4 (reference :293-294)
    assert np.array_equal(lz[3], oracle.resize_u8(clip[3], (135, 90), "lanczos"))
    fclip = clip.astype(np.float32) / 255
    fl = FilterResize(VideoMemory(fclip), 0.5)                       # float32 videos resize as float32
    got = np.stack(list(fl))
    assert got.dtype == np.float32 and np.array_equal(got, oracle.resize_f32(fclip, (45, 30), "area"))


def _philox_normals(seed, first_index, count):
    """NumPy restatement of va_gaussian_noise's stream: Philox4x32-10 on the pair counter, two
    53-bit uniforms, Box-Muller in float64 (checker only)"""
    pair0 = first_index >> 1
    npairs = ((first_index + count + 1) >> 1) - pair0
    g = (np.arange(npairs, dtype=np.uint64) + np.uint64(pair0))
    c = [(g & np.uint64(0xFFFFFFFF)), (g >> np.uint64(32)), np.zeros(npairs, np.uint64), np.zeros(npairs, np.uint64)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & M, p0 & M]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    u1 = ((((c[0] << np.uint64(32)) | c[1]) >> np.uint64(11)).astype(np.float64) + 1.0) / 9007199254740992.0
    u2 = (((c[2] << np.uint64(32)) | c[3]) >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
    rad = np.sqrt(-2.0 * np.log(u1))
    z = np.stack([rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)], 1).ravel()
    off = first_index - 2 * pair0
    return z[off:off + count]


def test_gaussian_noise_video_on_device(ops):
    """N4: VideoGaussianNoise frames come from the GPU: reproducible and seekable (sample i is a
    function of (seed, i)), the documented Philox / Box-Muller stream, right statistics, and
    straight into device memory for the engine"""
    from video._hip import DeviceBuffer
    from video.io.computed import VideoGaussianNoise
    g = VideoGaussianNoise(6, (50, 31), mean=3.0, std=2.0, seed=11)
    f2 = g[2]
    assert f2.dtype == np.float64 and f2.shape == (31, 50)
    assert np.array_equal(f2, g[2]) and not np.array_equal(g[1], f2)
    frames = np.stack(list(g))
    ref = _philox_normals(11, 0, frames.size).reshape(frames.shape) * 2.0 + 3.0
    assert np.allclose(frames, ref, rtol=0, atol=1e-12)                  # device libm vs NumPy: last bits only
    assert np.array_equal(frames[2], f2)                                # a frame on its own == inside the sequence
    odd = ops.gaussian_noise((7,), np.float64, 0.0, 1.0, seed=11, first_index=5)     # unaligned windows of the stream
    assert np.allclose(odd, _philox_normals(11, 5, 7), atol=1e-12)
    big = VideoGaussianNoise(2, (640, 480), mean=100, std=10, seed=5, dtype=np.float32)[1]
    assert abs(big.mean() - 100) < 0.1 and abs(big.std() - 10) < 0.1
    assert abs(((big - 100) ** 3).mean()) < 30 and abs(((big - 100) / 10) ** 4).mean() - 3 < 0.1
    u8 = VideoGaussianNoise(3, (64, 48), mean=250, std=20, seed=1, dtype=np.uint8, is_color=True)
    fr = u8[0]
    assert fr.dtype == np.uint8 and fr.shape == (48, 64, 3) and fr.max() == 255 and (fr == 255).mean() > 0.3
    want = np.clip(_philox_normals(1, 0, fr.size) * 20 + 250, 0, 255)
    assert np.abs(fr.ravel().astype(np.float64) - np.trunc(want)).max() <= 1      # (truncation next to an integer)
    # generated in place for the engine: no host copy
    src = VideoGaussianNoise(8, (96, 64), mean=100, std=30, seed=9, dtype=np.uint8)
    dev = DeviceBuffer(8 * 96 * 64)
    src.fill_device(dev.ptr, 0, 8)
    assert np.array_equal(dev.download((8, 64, 96), np.uint8), np.stack(list(src)))
    dev.free()


def test_filter_normalize_other_dtypes(ops):
    """A5: FilterNormalize for float32 frames / float targets (video/filters.py:101-135)"""
    from video.filters import FilterNormalize
    from video.io.memory import VideoMemory
    rng = np.random.default_rng(4)
    f32 = (rng.random((4, 30, 40), dtype=np.float32) * 3 - 1).astype(np.float32)
    v = FilterNormalize(VideoMemory(f32), 0.0, 1.5, dtype=np.uint8)
    ref = ((np.clip(f32.astype(np.float64), 0.0, 1.5) - 0.0) * (255 / 1.5) + 0).astype(np.int64).astype(np.uint8)
    assert np.array_equal(np.stack(list(v)), ref)
    u8 = rng.integers(0, 256, (3, 20, 30), dtype=np.uint8)
    w = FilterNormalize(VideoMemory(u8), 50, 200, dtype=np.float32)
    refw = ((np.clip(u8.astype(np.float64), 50, 200) - 50) * (1.0 / 150.0) + 0).astype(np.float32)
    assert np.array_equal(np.stack(list(w)), refw)
    auto = FilterNormalize(VideoMemory(f32))                         # limits and dtype learnt from the first frame
    a0 = auto[0]
    assert a0.dtype == np.float32 and a0.min() == 0.0 and abs(a0.max() - 1.0) < 1e-6
    d64 = FilterNormalize(VideoMemory(u8), 0, 255, dtype=np.float64)[1]
    assert d64.dtype == np.float64 and np.array_equal(d64, u8[1].astype(np.float64) * (1.0 / 255.0))


def test_fork_and_analysis_preprocessor_on_gpu(oracle):
    """N3: VideoFork feeding two GPU filter chains in lock step, and the dict-per-frame
    VideoAnalysisPreprocessor ({'raw','blur','mask','labels','count','stats'}) on the streamed engine"""
    from video import filters as F
    from video.io.base import VideoFork
    from video.io.memory import VideoMemory
    from video.io.parallel import VideoAnalysisPreprocessor, VideoPreprocessor
    clip = _blob_clip(45, 72, 104, seed=21, salt=0.003)
    rm, rl, rc, _ = oracle.chain_u8(clip, 2.0, 20, morph_ksize=5, connectivity=4)
    rdiff, _ = oracle.bg_mean_u8(clip)
    rblur = oracle.gaussian_u8(rdiff, 2.0)
    fork = VideoFork(VideoMemory(clip))
    masks = F.FilterMorphology(F.FilterThreshold(F.FilterBlur(F.FilterBackground(fork.get_client()), 2), 20), "close", 5)
    small = F.FilterResize(fork.get_client(), 0.5)
    k = 0
    for m, s in zip(masks, small):                     # NB: the contracted chain prefetches a batch ...
        assert np.array_equal(m, rm[k]) and np.array_equal(s, oracle.resize_u8(clip[k], (52, 36), "area"))
        k += 1
    assert k == 45
    pre = VideoAnalysisPreprocessor(VideoMemory(clip), outputs=("blur", "mask", "labels", "count", "stats"),
                                    sigma=2.0, threshold=20, max_labels=32, batch=16)
    assert len(pre) == 45
    n = 0
    for d in pre:
        assert set(d) == {"raw", "blur", "mask", "labels", "count", "stats"}
        assert np.array_equal(d["raw"], clip[n]) and np.array_equal(d["blur"], rblur[n])
        assert np.array_equal(d["mask"], rm[n]) and np.array_equal(d["labels"], rl[n]) and d["count"] == rc[n]
        c = min(int(rc[n]), 32)
        assert np.array_equal(d["stats"][:c, :14], oracle.region_stats(rl[n], int(rc[n]))[:c, :14])
        n += 1
    assert n == 45
    # the reference's form with GPU callables in worker threads (ctypes releases the GIL)
    from video import ops
    pre2 = VideoPreprocessor(VideoMemory(clip[:12]), {"blur": lambda f: ops.gaussian_blur(f, 2.0),
                                                     "big": lambda f: ops.threshold(f, 128)})
    for i, d in enumerate(pre2):
        assert np.array_equal(d["blur"], oracle.gaussian_u8(clip[i], 2.0))
        assert np.array_equal(d["big"], oracle.threshold_u8(clip[i], 128))


def test_pointwise_pre_stages_fold_into_the_engine(oracle):
    """A5: FilterCrop -> FilterMonochrome -> FilterNormalize in front of GPU stages run as ONE device
    pass inside the engine (va_prepare_u8) -- same frames as the filters applied one by one"""
    from video import _hip, filters as F
    from video._hip import DeviceBuffer, check
    from video.io.memory import VideoMemory
    rng = np.random.default_rng(33)
    col = rng.integers(0, 256, (40, 90, 130, 3), dtype=np.uint8)
    yy, xx = np.mgrid[:90, :130]
    for t in range(40):
        col[t][(xx - 40 - t) ** 2 + (yy - 45) ** 2 <= 144] = (250, 240, 245)

    def chain(src):
        crop = F.FilterCrop(src, rect=(10, 8, 96, 72))
        mono = F.FilterMonochrome(crop, "mean")
        norm = F.FilterNormalize(mono, 20, 220, dtype=np.uint8)
        return F.FilterMorphology(F.FilterThreshold(F.FilterBlur(F.FilterBackground(norm), 2), 25), "close", 3)

    fused = chain(VideoMemory(col))
    runner = fused._runner()
    assert runner is not None and runner.engine.prepare is not None and runner._source.is_color
    got = np.stack([np.array(f) for f in fused])
    F._GpuStage.contract = False
    try:
        ref = np.stack([np.array(f) for f in chain(VideoMemory(col))])
    finally:
        F._GpuStage.contract = True
    assert np.array_equal(got, ref)
    # and against the oracle, stage by stage
    c = col[:, 8:80, 10:106].astype(np.float64).sum(-1)
    m = (c / 3.0).astype(np.uint8)
    nm = ((np.clip(m.astype(np.float64), 20, 220) - 20) * (255 / 200.0) + 0).astype(np.int64).astype(np.uint8)
    rmask, _, _, _ = oracle.chain_u8(nm, 2.0, 25, morph_ksize=3, connectivity=4)
    assert np.array_equal(got, rmask)
    assert np.array_equal(fused[17], rmask[17])
    # channel pick through the crop, no normalisation, a single GPU stage on top
    blue = F.FilterBlur(F.FilterCrop(VideoMemory(col), region="left", color_channel="b"), 2)
    assert blue._runner() is not None
    assert np.array_equal(np.stack([np.array(f) for f in blue]), oracle.gaussian_u8(col[:, :, :65, 0], 2.0))
    # the kernel on its own: every mode
    L = _hip.lib()
    src = DeviceBuffer.from_array(col[:3])
    dst = DeviceBuffer(3 * 72 * 96 * 3)
    check(L.va_prepare_u8(src.ptr, dst.ptr, 3, 90, 130, 3, 10, 8, 96, 72, -1, 1, 20.0, 220.0, 255 / 200.0, 0.0, None))
    keep = dst.download((3, 72, 96, 3), np.uint8)
    want = ((np.clip(col[:3, 8:80, 10:106].astype(np.float64), 20, 220) - 20) * (255 / 200.0)).astype(np.int64).astype(np.uint8)
    assert np.array_equal(keep, want)
    assert L.va_prepare_u8(src.ptr, dst.ptr, 3, 90, 130, 3, 100, 8, 96, 72, -1, 0, 0.0, 0.0, 0.0, 0.0, None) != 0
    # every mono mode, with and without normalisation, on the four-samples-per-thread path (rows of whole dwords)
    # and on the one-sample path (width 95)
    for width, left in ((96, 10), (95, 10), (96, 8), (64, 0), (92, 12)):      # (left % 4 == 0: aligned rows, dword loads)
        crop = col[:3, 8:80, left:left + width]
        for mono in (-1, 0, 1, 2, 3):
            for normalize in (0, 1):
                out_c = 3 if mono < 0 else 1
                check(L.va_prepare_u8(src.ptr, dst.ptr, 3, 90, 130, 3, left, 8, width, 72, mono, normalize, 20.0, 220.0,
                                      255 / 200.0, 0.0, None))
                got1 = dst.download((3, 72, width, out_c), np.uint8)
                w0 = crop if mono < 0 else (crop[..., mono:mono + 1] if mono < 3 else
                                            (crop.astype(np.float64).sum(-1, keepdims=True) / 3.0).astype(np.uint8))
                if normalize:
                    w0 = ((np.clip(w0.astype(np.float64), 20, 220) - 20) * (255 / 200.0)).astype(np.int64).astype(np.uint8)
                assert np.array_equal(got1, w0), (width, left, mono, normalize)
    src.free(); dst.free()
