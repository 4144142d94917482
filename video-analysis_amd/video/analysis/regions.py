"""Region helpers: rectangles (host arithmetic) and blob detection (GPU).

Reference: video/analysis/regions.py -- corners_to_rect :23-29, rect_to_corners :33-45,
rect_to_slices :49-53, find_bounding_box :113-149, expand_rectangle :153-155,
get_largest_region :159-174.
"""
import numpy as np


def corners_to_rect(p1, p2):
    """rectangle (left, top, width, height) spanned by two corner points, both included"""
    xmin, xmax = min(p1[0], p2[0]), max(p1[0], p2[0])
    ymin, ymax = min(p1[1], p2[1]), max(p1[1], p2[1])
    return (xmin, ymin, xmax - xmin + 1, ymax - ymin + 1)


def rect_to_corners(rect, count=2):
    """`count` (2 or 4) corner points of a rectangle; the points lie inside it"""
    p1 = (rect[0], rect[1])
    p2 = (rect[0] + rect[2] - 1, rect[1] + rect[3] - 1)
    if count == 2:
        return p1, p2
    if count == 4:
        return p1, (p2[0], p1[1]), p2, (p1[0], p2[1])
    raise ValueError("count must be 2 or 4 (cannot be %d)" % count)


def rect_to_slices(rect):
    """(slice_y, slice_x) selecting the rectangle from an array"""
    return slice(rect[1], rect[1] + rect[3]), slice(rect[0], rect[0] + rect[2])


def expand_rectangle(rect, amount=1):
    return (rect[0] - amount, rect[1] - amount, rect[2] + 2 * amount, rect[3] + 2 * amount)


def label(mask, connectivity=4):
    """(labels, num_features) like scipy.ndimage.label -- the call get_largest_region makes
    at regions.py:162 -- computed by the run-based union-find kernels on the GPU"""
    from .. import ops
    return ops.label(mask, connectivity)


def find_bounding_box(mask):
    """[left, top, width, height] of the white region of a mask.  Like the reference it is
    meant for a single connected region; an empty mask raises IndexError."""
    from .. import ops
    m = (np.asarray(mask) != 0).astype(np.int32)
    st = ops.region_stats(m, 1)[0]
    if st[0] == 0:
        raise IndexError("mask is empty")
    xmin, ymin, xmax, ymax = (int(v) for v in st[10:14])
    return (xmin, ymin, xmax - xmin + 1, ymax - ymin + 1)


def get_largest_region(mask, ret_area=False, connectivity=4):
    """boolean mask that only contains the largest 4-connected region (first one on ties);
    ValueError on an empty mask, like np.argmax([]) in the reference"""
    from .. import ops
    region, area, count = ops.largest_region(mask, connectivity)
    if count == 0:
        raise ValueError("attempt to get argmax of an empty sequence")
    if ret_area:
        return region, area
    return region


def get_contour_from_largest_region(mask, ret_area=False):
    """contour of the region with the largest contour area as an (N, 2) float64 array of (x, y)
    points -- cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + cv2.contourArea + argmax,
    reference: video/analysis/regions.py:178-197.  RuntimeError when the mask is empty."""
    from .. import ops
    points, area, count = ops.largest_contour(mask)
    if count == 0:
        raise RuntimeError("Could not find any contour")
    contour = np.squeeze(np.asarray(points.reshape(-1, 1, 2), np.double))
    if ret_area:
        return contour, area
    return contour
