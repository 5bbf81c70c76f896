"""oracle/augment_ref.py against Pillow, the library the reference calls (preprocess.py:171,218,286,323): every one of
the 32 op codes, on a non-symmetric RGB image, a single-channel map, and with both resample filters the reference uses
for the 90x rotation (BILINEAR for images, NEAREST for annotations, dataset.py:145-146)."""
import os
import sys

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import augment_ref as R  # noqa: E402


def pil_chain(a, op, resample):
    im = Image.fromarray(a)
    if op & 1:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    if op & 2:
        im = im.transpose(Image.FLIP_TOP_BOTTOM)
    if op & 4:
        im = im.transpose(Image.TRANSPOSE)
    im = im.rotate(90 * ((op >> 3) & 3), resample=resample, expand=True)
    return np.array(im)


@pytest.mark.parametrize("op", range(32))
def test_matches_pillow(op):
    rng = np.random.default_rng(op)
    rgb = rng.integers(0, 256, (12, 12, 3), dtype=np.uint8)
    plane = rng.integers(0, 2, (12, 12), dtype=np.uint8)
    np.testing.assert_array_equal(R.d4(rgb, op), pil_chain(rgb, op, Image.BILINEAR))
    np.testing.assert_array_equal(R.d4(plane, op), pil_chain(plane, op, Image.NEAREST))


def test_draw_ops_follows_the_reference_call_order():
    import random
    r1, n1 = random.Random(5), np.random.RandomState(7)
    ops = R.draw_ops(3, r1, n1)
    r2, n2 = random.Random(5), np.random.RandomState(7)
    want = []
    for _ in range(3):
        h = r2.random() < 0.5; v = r2.random() < 0.5; t = r2.random() < 0.5
        a = n2.choice([0, 90, 180, 270])
        want.append(int(h) | int(v) << 1 | int(t) << 2 | (int(a) // 90) << 3)
    assert ops == want and all(0 <= o < 32 for o in ops)
