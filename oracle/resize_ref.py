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
