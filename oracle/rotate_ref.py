"""TEST INFRASTRUCTURE (oracle) - numpy restatement of the two non-D4 augmentations the reference ships ENABLED
(code/settings/CVPPP/training_settings.py:40 ROTATION, :50 CENTER_CUT), as `AlignCollate.__preprocess` applies them
(code/lib/dataset.py:236-269):

  annotation rotation   dataset.py:142,243-249 -> preprocess.py:311-327 `rotate(img, angle, Image.NEAREST, expand=True)`
                        per instance plane and for the semantic map: PIL Image.rotate -> Image.transform(AFFINE) ->
                        Geometry.c affine_fixed (nearest neighbour, 16.16 fixed point, zero fill)
  image rotation        dataset.py:140,241 -> preprocess.py:330-365 `rotate_with_random_bg`: RGB -> RGBA, rotate BILINEAR
                        (Geometry.c ImagingGenericTransform + bilinear_filter32RGB, double arithmetic, truncation),
                        composite over one of four background colours (white, black, per-channel int(mean), int(median)
                        of the source), back to RGB.  Alpha is 0 or 255 only (a sample point outside the source gives a
                        zero pixel, one inside interpolates alpha 255 with clamped neighbours), so the composite selects.
  centre cut            dataset.py:252-269 -> preprocess.py:239-264 `CenterCut`: a window of 2*h x 2*w around a chosen
                        object pixel, clamped to the image; a plane survives when its window sums to more than 30.

Pillow is the reference's dependency (unversioned there); every function here is pinned against the installed Pillow
in tests/test_oracle_rotate.py over all 19 angles the reference can draw (int(rand * 10), random sign) and several
sizes.  Angles: `rot_angle = int(np.random.rand() * 10)`, negated with probability 1/2 (dataset.py:237-239).
"""
import math

import numpy as np


def rotate_matrix(w, h, angle):
    """PIL.Image.rotate(angle, expand=True) with default centre / translate: (matrix[6], (nw, nh)), or None for the
    fast paths (angle % 360 in {0, 90, 180, 270}: plain transposes)."""
    angle = angle % 360.0
    if angle in (0, 90, 180, 270):
        return None
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]

    def tf(x, y):
        return m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]

    m[2], m[5] = tf(-cx, -cy)
    m[2] += cx
    m[5] += cy
    xs, ys = zip(*(tf(x, y) for x, y in ((0, 0), (w, 0), (w, h), (0, h))))
    nw = math.ceil(max(xs)) - math.floor(min(xs))
    nh = math.ceil(max(ys)) - math.floor(min(ys))
    m[2], m[5] = tf(-(nw - w) / 2.0, -(nh - h) / 2.0)
    return m, (nw, nh)


def fixed_coeffs(m):
    """Geometry.c affine_fixed: 16.16 fixed point, FIX(v) = floor(v * 65536 + 0.5); the half-pixel centre is folded into
    the two offsets."""
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5),
            fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate_nearest(a, angle):
    """a: uint8 [h, w] or [h, w, c]; Image.rotate(angle, NEAREST, expand=True), zero fill."""
    h, w = a.shape[:2]
    r = rotate_matrix(w, h, angle)
    if r is None:
        return np.ascontiguousarray(np.rot90(a, int(angle % 360) // 90))
    m, (nw, nh) = r
    a0, a1, a2, a3, a4, a5 = fixed_coeffs(m)
    y, x = np.mgrid[0:nh, 0:nw].astype(np.int64)
    xin = (a2 + a1 * y + a0 * x) >> 16
    yin = (a5 + a4 * y + a3 * x) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros((nh, nw) + a.shape[2:], a.dtype)
    out[ok] = a[yin[ok], xin[ok]]
    return out


def background(rgb, key):
    """preprocess.py:351-362: key 0 white, 1 black, 2 int(mean) per channel, 3 int(median) per channel."""
    if key == 0:
        return (255, 255, 255)
    if key == 1:
        return (0, 0, 0)
    if key == 2:
        return tuple(int(v) for v in rgb.mean((0, 1)))
    return tuple(int(v) for v in np.median(rgb, (0, 1)))


def rotate_bilinear_bg(rgb, angle, bg):
    """rgb: uint8 [h, w, 3]; rotate_with_random_bg with the background colour `bg` (3 ints) already drawn."""
    h, w = rgb.shape[:2]
    r = rotate_matrix(w, h, angle)
    if r is None:
        return np.ascontiguousarray(np.rot90(rgb, int(angle % 360) // 90))
    m, (nw, nh) = r
    y, x = np.mgrid[0:nh, 0:nw].astype(np.float64)
    xin = m[0] * (x + 0.5) + m[1] * (y + 0.5) + m[2]          # Geometry.c affine_transform
    yin = m[3] * (x + 0.5) + m[4] * (y + 0.5) + m[5]
    inside = (xin >= 0.0) & (xin < w) & (yin >= 0.0) & (yin < h)
    xs, ys = xin - 0.5, yin - 0.5
    x0, y0 = np.floor(xs), np.floor(ys)
    dx, dy = xs - x0, ys - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
    xa, xb = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    ya = np.clip(y0, 0, h - 1)
    has2 = (y0 + 1 >= 0) & (y0 + 1 < h)
    yb = np.clip(y0 + 1, 0, h - 1)
    src = rgb.astype(np.float64)
    out = np.empty((nh, nw, 3), np.uint8)
    for c in range(3):
        v1 = src[ya, xa, c] + (src[ya, xb, c] - src[ya, xa, c]) * dx
        v2 = src[yb, xa, c] + (src[yb, xb, c] - src[yb, xa, c]) * dx
        v2 = np.where(has2, v2, v1)
        v = v1 + (v2 - v1) * dy
        out[..., c] = np.where(inside, v.astype(np.uint8), np.uint8(bg[c]))     # (UINT8) truncation; composite = select
    return out


def draw_rotation(np_random):
    """dataset.py:237-239 in the reference's call order: angle, sign; then (image only) the background key, drawn inside
    rotate_with_random_bg AFTER the rotation (preprocess.py:349).  Returns (angle, draw_key) where draw_key() consumes the
    key draw at the right moment."""
    angle = int(np_random.rand() * 10)
    if np_random.rand() >= 0.5:
        angle = -1 * angle
    return angle


def center_cut_window(H, W, center, h, w):
    """preprocess.py:239-264: the window CenterCut takes from an H x W array for an output of h x w (it doubles both).
    Returns (h_0, w_0, height, width)."""
    h, w = 2 * h, 2 * w
    if center[0] - h // 2 < 0:
        h_0 = 0
    elif center[0] + h // 2 > H:
        h_0 = max(0, H - h)
    else:
        h_0 = center[0] - h // 2
    if center[1] - w // 2 < 0:
        w_0 = 0
    elif center[1] + w // 2 > W:
        w_0 = max(0, W - w)
    else:
        w_0 = center[1] - w // 2
    return int(h_0), int(w_0), min(H, h) if h_0 + min(H, h) <= H else H - h_0, min(W, w) if w_0 + min(W, w) <= W else W - w_0


def center_cut(image, sem, planes, pick, h, w):
    """dataset.py:252-269.  planes: uint8 [H, W, n]; `pick` in [0, #candidates): the reference's
    np.random.choice(len(centers)) over the row-major list of pixels where the planes sum to exactly 1.
    Returns (image crop, sem crop, surviving planes [h', w', n'], kept plane indices)."""
    ins_all = planes.astype(np.float32).sum(2)
    ys, xs = np.where(ins_all == 1)
    center = (int(ys[pick]), int(xs[pick]))
    H, W = sem.shape
    h0, w0, hh, ww = center_cut_window(H, W, center, h, w)
    crop = lambda a: a[h0:h0 + hh, w0:w0 + ww]
    keep = [i for i in range(planes.shape[2]) if int(crop(planes[:, :, i]).sum()) > 30]
    return crop(image), crop(sem), np.ascontiguousarray(crop(planes)[:, :, keep]), keep
