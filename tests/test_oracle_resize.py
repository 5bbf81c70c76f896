"""oracle/resize_ref.py against Pillow's Image.resize(NEAREST), the call behind the reference's `ann_resizer`
(dataset.py:162-170,293-320): the sizes of the CVPPP A1 images (530x500 -> 256x256) and a few hundred random pairs."""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import resize_ref as R  # noqa: E402


def test_index_tables_match_pillow():
    rng = np.random.default_rng(0)
    pairs = [(530, 256), (500, 256), (256, 256), (1, 5), (7, 3), (3, 7), (600, 256), (257, 256), (255, 256), (2048, 256),
             (441, 512), (256, 768)] + [(int(a), int(b)) for a, b in rng.integers(1, 700, (400, 2))]
    for n_in, n_out in pairs:
        src = np.zeros((2, n_in), np.int32)
        src[:] = np.arange(n_in)
        want = np.array(Image.fromarray(src, mode="I").resize((n_out, 2), Image.NEAREST))[0]
        np.testing.assert_array_equal(R.scale_table(n_in, n_out), want, err_msg="%d -> %d" % (n_in, n_out))


def test_planes_match_pillow():
    rng = np.random.default_rng(1)
    for (h0, w0, h, w) in [(530, 500, 256, 256), (37, 91, 64, 48), (16, 16, 33, 7)]:
        planes = (rng.random((h0, w0, 3)) < 0.4).astype(np.uint8) * 255
        got = R.resize_nearest(planes, h, w)
        for k in range(3):
            want = np.array(Image.fromarray(planes[:, :, k]).resize((w, h), Image.NEAREST))
            np.testing.assert_array_equal(got[:, :, k], want)
        np.testing.assert_array_equal(R.resize_nearest(planes[:, :, 0], h, w), got[:, :, 0])


def test_bilinear_matches_pillow():
    """oracle/resize_ref.resize_bilinear against PIL.Image.resize(BILINEAR) - the reference's `img_resizer` - bit for
    bit: CVPPP's 530x500 -> 256x256, up-scaling, identity axes, extreme aspect changes, random sizes."""
    rng = np.random.default_rng(2)
    cases = [(530, 500, 256, 256), (256, 256, 256, 256), (37, 91, 64, 48), (16, 16, 33, 7), (300, 400, 512, 512),
             (1, 9, 4, 3), (1024, 1024, 256, 256), (256, 300, 256, 256), (301, 417, 256, 256)]
    cases += [tuple(int(v) for v in rng.integers(1, 400, 4)) for _ in range(40)]
    for (h0, w0, h, w) in cases:
        img = rng.integers(0, 256, (h0, w0, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))
        got = R.resize_bilinear(img, h, w)
        np.testing.assert_array_equal(got, want, err_msg="%dx%d -> %dx%d" % (h0, w0, h, w))
    # flat and saturated images stay exact (the rounding constant and the clip)
    for v in (0, 255, 128):
        img = np.full((50, 70, 3), v, np.uint8)
        np.testing.assert_array_equal(R.resize_bilinear(img, 31, 17), np.asarray(Image.fromarray(img).resize((17, 31), Image.BILINEAR)))
