"""Region helpers: rectangles (host arithmetic) and blob detection (GPU).

Reference: video/analysis/regions.py -- corners_to_rect :23-29, rect_to_corners :33-45,
rect_to_slices :49-53, get_overlapping_slices :57-110, find_bounding_box :113-149,
expand_rectangle :153-155, get_largest_region :159-174, triangle_area :430-451.
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


def get_overlapping_slices(t_pos, t_shape, i_shape, anchor='center', ret_rect=False):
    """slices that cut the common part out of a template placed in a larger image
    (reference :57-110).  t_pos = (x, y) of the template's anchor in the image, t_shape =
    (height, width) of the template, i_shape = (height, width) of the image.  Returns
    ((template_rows, template_cols), (image_rows, image_cols)) and, with ret_rect, the common
    rectangle (left, top, width, height) in image coordinates; RuntimeError if nothing overlaps."""
    th, tw = t_shape[0], t_shape[1]
    if anchor == 'center':
        left, top = t_pos[0] - tw // 2, t_pos[1] - th // 2
    elif anchor == 'upper left':
        left, top = t_pos[0], t_pos[1]
    else:
        raise ValueError('Unknown anchor point: %s' % anchor)

    def overlap(start, t_len, i_len):
        """(image start, template start, length) along one axis"""
        length = min(t_len, i_len - start)
        if length <= 0 or start <= -t_len:
            raise RuntimeError('Template and image do not overlap')
        if start >= 0:
            return start, 0, length
        return 0, -start, length + start

    i_x, t_x, w = overlap(left, tw, i_shape[1])
    i_y, t_y, h = overlap(top, th, i_shape[0])
    slices = ((slice(t_y, t_y + h), slice(t_x, t_x + w)),
              (slice(i_y, i_y + h), slice(i_x, i_x + w)))
    if ret_rect:
        return slices, (i_x, i_y, w, h)
    return slices


def triangle_area(a, b, c):
    """area of a triangle with side lengths a, b, c (numbers or arrays) by Heron's formula;
    radicands that rounding made negative give 0 (reference :430-451)"""
    s = (a + b + c) / 2
    radicand = s * (s - a) * (s - b) * (s - c)
    if isinstance(radicand, np.ndarray):
        return np.sqrt(np.where(radicand > 0, radicand, 0))
    return np.sqrt(radicand) if radicand > 0 else 0


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
