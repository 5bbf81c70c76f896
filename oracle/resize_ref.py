"""TEST INFRASTRUCTURE (oracle) — nearest-neighbour resize of annotation planes as the reference's collate function
does it: `ann_resizer` = `IU.image_resizer(h, w, interpolation=Image.NEAREST)` (code/lib/dataset.py:162-164,168-170;
utils.py:26-27 -> torchvision `Resize` -> `PIL.Image.resize((w, h), NEAREST)`), called once per instance plane and for
the semantic map (dataset.py:293-320).

The arithmetic lives in Pillow (third-party dependency, absent from /root/reference, no version pinned there):
`ImagingScaleAffine` in libImaging/Geometry.c - output column x reads source column int(xo), xo = 0.5 * s for x = 0
and xo += s for each further column, s = in / out, accumulated in double; rows the same.  Restated below;
tests/test_oracle_resize.py pins it against the installed Pillow over several hundred size pairs.
"""
import numpy as np


def scale_table(n_in, n_out):
    a = float(n_in) / float(n_out)
    o = a * 0.5
    tab = np.empty(n_out, np.int64)
    for i in range(n_out):
        tab[i] = min(max(int(o), 0), n_in - 1)
        o += a
    return tab


def resize_nearest(x, h, w):
    """x: [..., h0, w0, c] or [h0, w0]; returns the array resized to (h, w) over the two spatial axes."""
    x = np.asarray(x)
    if x.ndim == 2:
        return x[scale_table(x.shape[0], h)][:, scale_table(x.shape[1], w)]
    ty, tx = scale_table(x.shape[-3], h), scale_table(x.shape[-2], w)
    return np.ascontiguousarray(np.take(np.take(x, ty, axis=-3), tx, axis=-2))


# ---------------------------------------------------------------------------------------------------------------
# Bilinear image resize: `img_resizer` = `IU.image_resizer(h, w)` (code/lib/dataset.py:160-161,166-167,
# prediction.py:37; utils.py:26-27 -> PIL.Image.resize((w, h), BILINEAR)), applied to every RGB image before ImageEx.
# Pillow's arithmetic (libImaging/Resample.c, 8 bits per channel): a separable triangle filter whose support grows
# with the down-scaling factor (anti-aliasing), coefficients computed in double and rounded to 22-bit fixed point,
# a horizontal pass into a uint8 intermediate, then a vertical pass; each pass starts from 1 << 21 (round half up)
# and clips to [0, 255].  Restated below; tests/test_oracle_resize.py pins it against the installed Pillow.
PRECISION_BITS = 32 - 8 - 2


def bilinear_coeffs(n_in, n_out):
    """(bounds[n_out, 2] = (first source index, count), coeffs[n_out, ksize] int64 fixed point) - precompute_coeffs +
    normalize_coeffs_8bpc of Resample.c for the triangle filter over the whole source extent."""
    scale = float(n_in) / float(n_out)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((n_out, 2), np.int64)
    kk = np.zeros((n_out, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(n_out):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), n_in) - xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] = w[:xmax] / ww
        for x in range(xmax):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(x, n_out, axis):
    """One resampling pass of a uint8 array along `axis`."""
    x = np.moveaxis(np.asarray(x, np.uint8), axis, -1)
    n_in = x.shape[-1]
    if n_in == n_out:
        return np.moveaxis(x, -1, axis)
    bounds, kk = bilinear_coeffs(n_in, n_out)
    out = np.empty(x.shape[:-1] + (n_out,), np.uint8)
    xi = x.astype(np.int64)
    for xx in range(n_out):
        lo, cnt = bounds[xx]
        acc = (xi[..., lo:lo + cnt] * kk[xx, :cnt]).sum(-1) + (1 << (PRECISION_BITS - 1))
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, -1, axis)


def resize_bilinear(img, h, w):
    """img uint8 [h0, w0, c] (or [n, h0, w0, c]) -> uint8 [..., h, w, c]: horizontal pass first, then vertical
    (ImagingResample)."""
    img = np.asarray(img, np.uint8)
    tmp = _pass(img, w, axis=-2)
    return np.ascontiguousarray(_pass(tmp, h, axis=-3))
