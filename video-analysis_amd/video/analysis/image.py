"""Image operations: region properties from image moments, morphology, statistics.

Reference: video/analysis/image.py -- set_image_border :205-210, regionprops :310-405.
The ten spatial moments are accumulated on the GPU from run segments (exact integers); the
central / normalised moments and the derived scalars are the reference's formulas evaluated in
float64 on the host (a few dozen flops per region).
"""
import math

import numpy as np

_SPATIAL = ("m00", "m10", "m01", "m20", "m11", "m02", "m30", "m21", "m12", "m03")


def set_image_border(img, size=1, color=0):
    """sets the border of an image to `color` (in place)"""
    img[:size, :] = color
    img[-size:, :] = color
    img[:, :size] = color
    img[:, -size:] = color


def moments_from_spatial(spatial):
    """dict with the 24 entries of cv2.moments() from the ten spatial moments, evaluated in the
    operation order of OpenCV's completeMomentState"""
    m = {k: float(v) for k, v in zip(_SPATIAL, spatial)}
    cx = cy = inv_m00 = 0.0
    if abs(m["m00"]) > 2.220446049250313e-16:
        inv_m00 = 1.0 / m["m00"]
        cx = m["m10"] * inv_m00
        cy = m["m01"] * inv_m00
    mu20 = m["m20"] - m["m10"] * cx
    mu11 = m["m11"] - m["m10"] * cy
    mu02 = m["m02"] - m["m01"] * cy
    m["mu20"], m["mu11"], m["mu02"] = mu20, mu11, mu02
    m["mu30"] = m["m30"] - cx * (3 * mu20 + cx * m["m10"])
    mu11 += mu11
    m["mu21"] = m["m21"] - cx * (mu11 + cx * m["m01"]) - cy * mu20
    m["mu12"] = m["m12"] - cy * (mu11 + cy * m["m10"]) - cx * mu02
    m["mu03"] = m["m03"] - cy * (3 * mu02 + cy * m["m01"])
    inv_sqrt_m00 = math.sqrt(abs(inv_m00))
    s2 = inv_m00 * inv_m00
    s3 = s2 * inv_sqrt_m00
    for k in ("20", "11", "02"):
        m["nu" + k] = m["mu" + k] * s2
    for k in ("30", "21", "12", "03"):
        m["nu" + k] = m["mu" + k] * s3
    return m


def image_moments(mask):
    """cv2.moments(mask.astype(np.uint8)) for a 0/1 mask (image.py:353)"""
    from .. import ops
    m = (np.asarray(mask) != 0).astype(np.int32)
    if m.ndim != 2:
        raise ValueError("mask must be 2-d")
    return moments_from_spatial(ops.region_stats(m, 1)[0][:10])


def contour_moments(contour):
    """cv2.moments(contour) (image.py:355; shapes.py:533): Green's-theorem moments of a closed
    polygon given as (N, 2) / (N, 1, 2) points.  Integer arrays are read as int32 points, all
    other dtypes as float32 points, like Polygon.moments' explicit cast in the reference."""
    from .. import ops
    return moments_from_spatial(ops.contour_moments(contour))


class regionprops(object):
    """properties of a region given by a boolean mask, by its contour or by precomputed moments
    (reference :310-405; the formulas follow scikit-image, as the reference notes)"""

    def __init__(self, mask=None, contour=None, moments=None):
        if moments is not None:
            self.moments = moments
        elif mask is not None:
            self.moments = image_moments(mask)
        elif contour is not None:
            self.moments = contour_moments(contour)
        else:
            raise ValueError("Either the mask or the moments must be given")

    @property
    def area(self):
        return self.moments["m00"]

    @property
    def centroid(self):
        m = self.moments
        return (m["m10"] / m["m00"], m["m01"] / m["m00"])

    @property
    def orientation(self):
        m = self.moments
        a, b, c = m["mu20"], m["mu11"], m["mu02"]
        if a - c == 0:
            return -math.pi / 4 if b > 0 else math.pi / 4
        return -math.atan2(2 * b, (a - c)) / 2

    @property
    def inertia_tensor_eigvals(self):
        m = self.moments
        a, b, c = m["mu20"] / m["m00"], -m["mu11"] / m["m00"], m["mu02"] / m["m00"]
        root = math.sqrt(4 * b ** 2 + (a - c) ** 2)
        return (a + c) + root, (a + c) - root

    @property
    def eccentricity(self):
        e1, e2 = self.inertia_tensor_eigvals
        return 0 if e1 == 0 else math.sqrt(1 - e2 / e1)

    @property
    def major_axis_length(self):
        return 4 * math.sqrt(self.inertia_tensor_eigvals[0])

    @property
    def minor_axis_length(self):
        return 4 * math.sqrt(self.inertia_tensor_eigvals[1])


def detect_peaks(img, include_plateaus=True):
    """boolean mask of the pixels that are maximal in their 8-neighbourhood; with plateaus the
    eroded zero-background is removed (reference :267-306)"""
    from .. import ops
    return ops.detect_peaks(img, include_plateaus)


def mask_thinning(img, method="auto"):
    """skeleton of a mask.  Only the reference's `python` method (iterated 3x3-cross
    erosion/dilation, :243-258) exists on the GPU; 'guo-hall' needs the external `thinning`
    module in the reference as well and is not provided."""
    if method == "guo-hall":
        raise ImportError("Using the `guo-hall` method for thinning requires the `thinning` "
                          "module, which is not part of the GPU path.")
    if method not in ("auto", "python"):
        raise ValueError("Unknown thinning method `%s`" % method)
    from .. import ops
    return ops.mask_thinning(img)[0]


def get_image_statistics(img, kernel="box", ksize=5, ret_var=True, prior=None,
                         exclude_center=False):
    """mean and variance in a window around every point of an image (reference :131-201).
    `prior` is subtracted before summing (default: the image mean)."""
    from .. import ops
    img = np.asarray(img)
    if prior is None:
        prior = img.mean()
    if kernel not in ("box", "ellipse", "circle"):
        raise ValueError("Unknown filter kernel `%s`" % kernel)
    return ops.image_statistics(img, kernel, int(ksize), prior, exclude_center, ret_var)
