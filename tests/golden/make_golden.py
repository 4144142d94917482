#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz -- the committed known-answer vectors.

Run in the BUILD container (SciPy present; optional /opt/conda python3.9 with scikit-image):
    python tests/golden/make_golden.py

Sources of truth (none of them is the build's own HIP code):
  * labels_*      : scipy.ndimage.label -- the exact call the reference makes at
                    video/analysis/regions.py:162 (default 4-connectivity) and with a full
                    3x3 structure (8-connectivity).
  * mean_*/std_*  : the literal NumPy formulas of video/analysis/video.py:33 and :48-55.
  * erode_/dilate_: scipy.ndimage.grey_erosion / grey_dilation with a constant neutral border
                    (== OpenCV's default morphology border, video/analysis/image.py:250-251).
  * moments_*     : direct NumPy sums of x^p y^q over the mask (cv2.moments definition,
                    video/analysis/image.py:353) and, when available, scikit-image 0.18
                    `measure.moments` / `regionprops` as an independent check.
  * largest_*     : the reference's own recipe get_largest_region
                    (video/analysis/regions.py:159-174) restated with scipy + numpy.
  * gauss_*       : the ORACLE's own output (OpenCV is not installed: parity with cv2 is
                    unpinned; these vectors pin the oracle/HIP code against regressions and
                    carry the analytic fixed-point taps).
The reference itself cannot be imported here (Python 2 + missing cv2/utils).
"""
import json
import os
import subprocess
import sys

import numpy as np
from scipy import ndimage

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def spiral(n):
    m = np.zeros((n, n), bool)
    x0, y0, x1, y1 = 0, 0, n - 1, n - 1
    while x0 <= x1 and y0 <= y1:
        m[y0, x0:x1 + 1] = True
        m[y0:y1 + 1, x1] = True
        if y1 > y0 + 1:
            m[y1, x0 + 2 if x0 + 2 <= x1 else x1:x1 + 1] = True
        if x1 > x0 + 2:
            m[y0 + 2:y1 + 1, x0 + 2] = True
        x0 += 2
        y0 += 2
        x1 -= 2
        y1 -= 2
        # open the ring so it stays one long snake
        if y0 < n and x0 < n:
            m[y0 - 1, x0 - 2 if x0 >= 2 else 0] = False
    return m


def serpentine(h, w):
    m = np.zeros((h, w), bool)
    for y in range(0, h, 2):
        m[y, :] = True
        if y + 1 < h:
            if (y // 2) % 2 == 0:
                m[y + 1, w - 1] = True
            else:
                m[y + 1, 0] = True
    return m


def masks():
    rng = np.random.default_rng(1234)
    out = {}
    for i, (h, w, d) in enumerate([(1, 1, 1.0), (1, 7, 0.5), (9, 1, 0.5), (5, 5, 0.1),
                                   (16, 16, 0.3), (31, 33, 0.5), (32, 32, 0.6), (40, 64, 0.45),
                                   (64, 64, 0.2), (64, 64, 0.55), (64, 64, 0.9), (17, 70, 0.5),
                                   (64, 33, 0.7), (48, 96, 0.35)]):
        out["rand%02d" % i] = rng.random((h, w)) < d
    out["empty"] = np.zeros((12, 20), bool)
    out["full"] = np.ones((12, 20), bool)
    yy, xx = np.mgrid[:24, :40]
    out["checker"] = ((yy + xx) % 2 == 0)
    out["hlines"] = (yy % 2 == 0)
    out["vlines"] = (xx % 2 == 0)
    out["spiral"] = spiral(33)
    out["serpentine"] = serpentine(21, 37)
    b = np.zeros((30, 50), bool)
    b[0:4, 0:6] = True
    b[26:30, 44:50] = True
    b[10:20, 20:35] = True
    b[0:3, 47:50] = True
    out["border_blobs"] = b
    t = np.zeros((20, 40), bool)
    t[2:6, 2:8] = True          # area 24
    t[10:14, 20:26] = True      # area 24 -> tie, first one must win
    t[16:18, 30:33] = True
    out["tie"] = t
    u = np.zeros((10, 100), bool)   # runs crossing 32-bit word boundaries
    u[1, 30:34] = True
    u[2, 31:65] = True
    u[4, 0:100] = True
    u[6, 63:65] = True
    u[7, 64:96] = True
    u[8, 95:97] = True
    out["wordcross"] = u
    d8 = np.zeros((12, 12), bool)   # diagonal chains: 1 component for 8-conn, many for 4
    for i in range(12):
        d8[i, i] = True
        d8[i, 11 - i] = True
    out["diag"] = d8
    return out


def raw_moments(mask):
    ys, xs = np.nonzero(mask)
    xs = xs.astype(object)
    ys = ys.astype(object)
    return np.array([int(np.sum(xs ** p * ys ** q)) if len(xs) else 0
                     for p, q in [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2), (3, 0), (2, 1),
                                  (1, 2), (0, 3)]], dtype=np.int64)


def skimage_moments(mask_dict):
    """central moments via scikit-image in the side interpreter; {} if unavailable"""
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        return {}
    tmp = os.path.join(HERE, "_tmp_masks.npz")
    np.savez(tmp, **{k: v for k, v in mask_dict.items() if v.any()})
    code = r"""
import sys, json, numpy as np
from skimage import measure
d = np.load(sys.argv[1]); out = {}
for k in d.files:
    m = d[k].astype(np.uint8)
    M = measure.moments(m, order=3)                      # M[q_row, p_col]
    cy, cx = M[1, 0] / M[0, 0], M[0, 1] / M[0, 0]
    mu = measure.moments_central(m, center=(cy, cx), order=3)
    out[k] = dict(m00=M[0, 0], m10=M[0, 1], m01=M[1, 0], mu20=mu[0, 2], mu11=mu[1, 1],
                  mu02=mu[2, 0], mu30=mu[0, 3], mu21=mu[1, 2], mu12=mu[2, 1], mu03=mu[3, 0])
print(json.dumps(out))
"""
    try:
        res = subprocess.run([py, "-c", code, tmp], capture_output=True, text=True, check=True)
        return json.loads(res.stdout)
    except Exception as e:  # pragma: no cover
        print("skimage unavailable:", e)
        return {}
    finally:
        os.remove(tmp)


def main():
    g = {}
    names = []
    ms = masks()
    for name, m in ms.items():
        names.append(name)
        g["mask_" + name] = m.astype(np.uint8)
        l4, c4 = ndimage.label(m)
        l8, c8 = ndimage.label(m, structure=np.ones((3, 3)))
        assert l4.dtype == np.int32
        g["labels4_" + name] = l4.astype(np.int32)
        g["labels8_" + name] = l8.astype(np.int32)
        g["count4_" + name] = np.int32(c4)
        g["count8_" + name] = np.int32(c8)
        # reference recipe: areas by np.sum(labels == l), first max wins
        areas = np.array([np.sum(l4 == l) for l in range(1, c4 + 1)], np.int64)
        g["areas4_" + name] = areas
        if c4:
            lm = int(np.argmax(areas)) + 1
            g["largest4_" + name] = (l4 == lm).astype(np.uint8)
        g["moments_" + name] = raw_moments(m)
        # per-label raw moments (4-conn)
        g["lmoments4_" + name] = np.array([raw_moments(l4 == l) for l in range(1, c4 + 1)],
                                          np.int64).reshape(c4, 10)
    g["mask_names"] = np.array(names)
    sk = skimage_moments(ms)
    g["skimage_json"] = np.array(json.dumps(sk))

    rng = np.random.default_rng(99)
    # running mean / std, literal NumPy formulas of the reference
    for n in (1, 2, 8, 64, 256):
        fr = rng.integers(0, 256, (n, 6, 10), dtype=np.uint8)
        g["bgframes_%d" % n] = fr
        g["mean_%d" % n] = O.measure_mean_numpy(fr)
        mean, std = O.measure_mean_std_numpy(fr)
        g["wmean_%d" % n] = np.asarray(mean, np.float64)
        g["wstd_%d" % n] = np.asarray(std, np.float64) * np.ones(fr.shape[1:])
    frc = rng.integers(0, 256, (5, 4, 6, 3), dtype=np.uint8)
    g["bgframes_color"] = frc
    g["mean_color"] = O.measure_mean_numpy(frc)

    # morphology via scipy (odd sizes; neutral constant border)
    img = rng.integers(0, 256, (24, 37), dtype=np.uint8)
    binimg = ((rng.random((24, 37)) < 0.4) * 255).astype(np.uint8)
    g["morph_img"] = img
    g["morph_bin"] = binimg
    cross = ndimage.generate_binary_structure(2, 1)
    for nm, im in (("img", img), ("bin", binimg)):
        for k in (3, 5, 7):
            g["erode_rect%d_%s" % (k, nm)] = ndimage.grey_erosion(im, size=(k, k), mode="constant",
                                                                 cval=255)
            g["dilate_rect%d_%s" % (k, nm)] = ndimage.grey_dilation(im, size=(k, k),
                                                                   mode="constant", cval=0)
        g["erode_cross3_" + nm] = ndimage.grey_erosion(im, footprint=cross, mode="constant",
                                                       cval=255)
        g["dilate_cross3_" + nm] = ndimage.grey_dilation(im, footprint=cross, mode="constant",
                                                         cval=0)

    # Gaussian: analytic taps + oracle outputs on impulse / step / ramp / noise
    for s in (0.5, 1.0, 2.0, 3.0, 5.0):
        g["taps_q8_%g" % s] = O.gauss_taps_q8(s)
    g["taps_f32_9"] = O.gauss_taps_f32(9.0)
    g["taps_f32_2"] = O.gauss_taps_f32(2.0)
    imp = np.zeros((41, 45), np.uint8)
    imp[20, 22] = 255
    imp[0, 0] = 200
    imp[40, 44] = 100
    step = np.zeros((32, 48), np.uint8)
    step[:, 24:] = 255
    ramp = (np.arange(48)[None, :] * 5 + np.arange(32)[:, None]).astype(np.uint8)
    noise = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    tiny = rng.integers(0, 256, (3, 5), dtype=np.uint8)      # kernel wider than the image
    for nm, im in (("imp", imp), ("step", step), ("ramp", ramp), ("noise", noise),
                   ("tiny", tiny)):
        g["gin_" + nm] = im
        for s in (2.0, 5.0):
            g["gout_%s_%g" % (nm, s)] = O.gaussian_u8(im, s)
    col = rng.integers(0, 256, (19, 23, 3), dtype=np.uint8)
    g["gin_color"] = col
    g["gout_color_2"] = O.gaussian_u8(col, 2.0, layout="hwc")
    f32 = rng.random((30, 40), dtype=np.float32)
    g["gin_f32"] = f32
    g["gout_f32_2"] = O.gaussian_f32(f32, 2.0)
    g["gout_f32_9"] = O.gaussian_f32(f32, 9.0)

    # whole chain on a small seeded clip (cfg#1-like, 16 x 48 x 64)
    bgimg = np.clip(rng.normal(100, 10, (48, 64)), 0, 255)
    clip = np.empty((16, 48, 64), np.uint8)
    yy, xx = np.mgrid[:48, :64]
    for t in range(16):
        f = bgimg + rng.normal(0, 4, (48, 64))
        for (cx, cy, r) in ((10 + 2 * t, 12, 5), (50 - t, 30 + t // 2, 7)):
            f[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] += 60
        clip[t] = np.clip(f, 0, 255).astype(np.uint8)
    g["chain_clip"] = clip
    mask, labels, counts, mean = O.chain_u8(clip, 2.0, 20, morph_ksize=5, connectivity=4)
    g["chain_mask"] = mask
    g["chain_labels"] = labels
    g["chain_counts"] = counts
    g["chain_mean"] = mean

    out = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes,", len(g), "arrays; skimage:",
          "yes" if sk else "no")


if __name__ == "__main__":
    main()
