"""oracle/rotate_ref.py (rotation by the reference's +-9 degree angles, nearest for annotations / bilinear with a drawn
background for the image; centre cut) pinned against the installed Pillow and against the reference's own CenterCut
arithmetic restated with PIL/numpy calls - the library the reference calls (preprocess.py:311-365, 239-264)."""
import os
import sys

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle")]
import rotate_ref as RR  # noqa: E402

ANGLES = list(range(-9, 10))
SIZES = [(37, 53), (64, 64), (100, 75), (130, 96)]


def pil_rotate_with_bg(rgb, angle, bg):
    """preprocess.py:330-365 with the background colour fixed."""
    img = Image.fromarray(rgb).convert('RGBA')
    img = img.rotate(angle, resample=Image.BILINEAR, expand=True)
    back = Image.new('RGBA', img.size, (bg[0], bg[1], bg[2], 255))
    return np.array(Image.composite(img, back, img).convert('RGB'))


@pytest.mark.parametrize("h,w", SIZES)
def test_rotate_nearest_matches_pillow(h, w):
    rs = np.random.RandomState(h * 131 + w)
    a = (rs.rand(h, w) < 0.4).astype(np.uint8) * rs.randint(1, 255, (h, w)).astype(np.uint8)
    for angle in ANGLES:
        ref = np.array(Image.fromarray(a).rotate(angle, resample=Image.NEAREST, expand=True))
        got = RR.rotate_nearest(a, angle)
        assert got.shape == ref.shape, (angle, got.shape, ref.shape)
        assert np.array_equal(got, ref), (angle, int((got != ref).sum()))


@pytest.mark.parametrize("h,w", SIZES)
def test_rotate_bilinear_with_background_matches_pillow(h, w):
    rs = np.random.RandomState(h * 17 + w)
    rgb = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    for angle in ANGLES:
        for key in range(4):
            bg = RR.background(rgb, key)
            ref = pil_rotate_with_bg(rgb, angle, bg)
            got = RR.rotate_bilinear_bg(rgb, angle, bg)
            assert got.shape == ref.shape, (angle, got.shape, ref.shape)
            assert np.array_equal(got, ref), (angle, key, int((got != ref).sum()))


def test_background_colours_follow_the_reference():
    rs = np.random.RandomState(3)
    rgb = rs.randint(0, 256, (40, 50, 3)).astype(np.uint8)
    assert RR.background(rgb, 0) == (255, 255, 255) and RR.background(rgb, 1) == (0, 0, 0)
    assert RR.background(rgb, 2) == tuple(map(int, rgb.mean((0, 1))))                 # preprocess.py:358
    assert RR.background(rgb, 3) == tuple(map(int, np.median(rgb, (0, 1))))           # preprocess.py:361


def reference_center_cut(img, center, h, w):
    """preprocess.py:239-264, line by line."""
    h *= 2; w *= 2
    H, W = img.shape[0], img.shape[1]
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
    img = img[h_0:h_0 + min(H, h), w_0:w_0 + min(W, w)]
    return img, img.sum() > 30


@pytest.mark.parametrize("H,W,h,w", [(120, 160, 32, 32), (50, 60, 32, 32), (200, 90, 40, 24), (64, 64, 16, 48)])
def test_center_cut_matches_the_reference_arithmetic(H, W, h, w):
    rs = np.random.RandomState(H + W)
    planes = np.zeros((H, W, 6), np.uint8)
    for i in range(6):
        y0, x0 = rs.randint(0, H - 8), rs.randint(0, W - 8)
        planes[y0:y0 + rs.randint(3, 12), x0:x0 + rs.randint(3, 12), i] = 1
    sem = (planes.sum(2) > 0).astype(np.uint8)
    image = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    ys, xs = np.where(planes.astype(np.float32).sum(2) == 1)
    for pick in (0, len(ys) // 3, len(ys) - 1):
        center = (ys[pick], xs[pick])
        img_c, sem_c, planes_c, keep = RR.center_cut(image, sem, planes, pick, h, w)
        ref_img, _ = reference_center_cut(image, center, h, w)
        ref_sem, _ = reference_center_cut(sem, center, h, w)
        assert np.array_equal(img_c, ref_img) and np.array_equal(sem_c, ref_sem)
        ref_planes = []
        for i in range(6):
            p, has = reference_center_cut(planes[:, :, i], center, h, w)
            if has:
                ref_planes.append(p)
        assert planes_c.shape[2] == len(ref_planes)
        for j, p in enumerate(ref_planes):
            assert np.array_equal(planes_c[:, :, j], p)
