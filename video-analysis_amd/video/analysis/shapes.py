"""Shapes: the moment-related part of the reference's Polygon (video/analysis/shapes.py:527-549).

The reference's shapes module is geometry on top of shapely (out of the hot-path scope,
SURVEY.md section 2); what the per-frame analysis path needs from it is `Polygon.moments` /
`Polygon.eccentricity`, i.e. cv2.moments of a contour, which run on the GPU here.
"""
import math

import numpy as np

from .image import contour_moments


class Polygon(object):
    """a closed polygon given by its contour, an (N, 2) sequence of (x, y) points"""

    def __init__(self, contour):
        contour = np.asarray(contour, np.double)
        if contour.ndim != 2 or contour.shape[1] != 2 or len(contour) < 3:
            raise ValueError("a polygon needs an (N, 2) contour with at least three points")
        self.contour = contour
        self._moments = None

    @property
    def moments(self):
        """all moments up to third order: cv2.moments(np.asarray(contour, np.float32))
        (reference :527-533; the float32 cast is the reference's own)"""
        if self._moments is None:
            self._moments = contour_moments(np.asarray(self.contour, np.float32))
        return self._moments

    @property
    def area(self):
        return self.moments["m00"]

    @property
    def centroid(self):
        m = self.moments
        return (m["m10"] / m["m00"], m["m01"] / m["m00"])

    @property
    def eccentricity(self):
        """0 for a circle ... 1 for a line (reference :537-549)"""
        m = self.moments
        a, b, c = m["mu20"], -m["mu11"], m["mu02"]
        root = math.sqrt(4 * b ** 2 + (a - c) ** 2)
        e1, e2 = (a + c) + root, (a + c) - root
        if e1 == 0:
            return 0
        return math.sqrt(1 - e2 / e1)
