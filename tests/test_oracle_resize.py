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
