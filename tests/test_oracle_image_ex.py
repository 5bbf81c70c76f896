"""Known-answer tests for oracle/image_ex_ref.py (the 21-channel colour expansion ahead of the network,
reference code/lib/utils.py:90-113).  scikit-image is absent here and unpinned upstream, so these are published
colour-science values, not outputs of the reference's dependency: parity with scikit-image itself stays
"unpinned" (oracle header)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import image_ex_ref as R  # noqa: E402


def px(r, g, b):
    return np.array([[[r, g, b]]], dtype=np.uint8)


def test_layout_and_raw_rgb_channels():
    img = np.random.default_rng(0).integers(0, 256, (5, 7, 3), dtype=np.uint8)
    out = R.image_ex(img)
    assert out.shape == (5, 7, 21) and out.dtype == np.float32
    np.testing.assert_array_equal(out[..., :3], img.astype(np.float32))     # rgb stays 0..255
    std = R.image_ex_standardized(img)
    assert std.shape == (21, 5, 7)
    np.testing.assert_allclose(std, (np.moveaxis(out, -1, 0) - 0.5) * 2, rtol=0, atol=0)


def test_lab_known_values():
    lab = lambda r, g, b: R.image_ex(px(r, g, b))[0, 0, 3:6]
    np.testing.assert_allclose(lab(255, 255, 255), [100.0, 0.0, 0.0], atol=2e-2)
    np.testing.assert_allclose(lab(0, 0, 0), [0.0, 0.0, 0.0], atol=1e-4)
    # sRGB primaries in CIE L*a*b* (D65 / 2 deg), Lindbloom's tables
    np.testing.assert_allclose(lab(255, 0, 0), [53.24, 80.09, 67.20], atol=5e-2)
    np.testing.assert_allclose(lab(0, 255, 0), [87.73, -86.18, 83.18], atol=5e-2)
    np.testing.assert_allclose(lab(0, 0, 255), [32.30, 79.19, -107.86], atol=5e-2)


def test_hsv_known_values():
    hsv = lambda r, g, b: R.image_ex(px(r, g, b))[0, 0, 6:9]
    np.testing.assert_allclose(hsv(255, 0, 0), [0.0, 1.0, 1.0], atol=1e-6)
    np.testing.assert_allclose(hsv(0, 255, 0), [1 / 3, 1.0, 1.0], atol=1e-6)
    np.testing.assert_allclose(hsv(0, 0, 255), [2 / 3, 1.0, 1.0], atol=1e-6)
    np.testing.assert_allclose(hsv(128, 128, 128), [0.0, 0.0, 128 / 255], atol=1e-6)
    np.testing.assert_allclose(hsv(0, 0, 0), [0.0, 0.0, 0.0], atol=0)
    np.testing.assert_allclose(hsv(255, 0, 255), [5 / 6, 1.0, 1.0], atol=1e-6)       # magenta: blue branch wins the tie


def test_bt601_families():
    o = R.image_ex(px(255, 255, 255))[0, 0]
    np.testing.assert_allclose(o[9:12], [1.0, 0.0, 0.0], atol=1e-6)                  # YUV of white
    np.testing.assert_allclose(o[12:15], [235.0, 128.0, 128.0], atol=1e-4)           # studio-range YCbCr
    np.testing.assert_allclose(o[18:21], [1.0, 0.0, 0.0], atol=1e-6)                 # YIQ of white
    k = R.image_ex(px(0, 0, 0))[0, 0]
    np.testing.assert_allclose(k[12:15], [16.0, 128.0, 128.0], atol=1e-6)
    r = R.image_ex(px(255, 0, 0))[0, 0]
    np.testing.assert_allclose(r[9], 0.299, atol=1e-6)
    np.testing.assert_allclose(r[12:15], [81.481, 90.203, 240.0], atol=1e-3)


def test_hed_properties():
    w = R.image_ex(px(255, 255, 255))[0, 0, 15:18]
    np.testing.assert_allclose(w, [0.0, 0.0, 0.0], atol=1e-7)                        # no stain on white
    img = np.random.default_rng(1).integers(0, 256, (16, 16, 3), dtype=np.uint8)
    hed = R.image_ex(img)[..., 15:18]
    assert (hed >= 0).all() and np.isfinite(hed).all()
    # forward model: optical density = stains @ RGB_FROM_HED wherever no stain was clipped at zero
    od = np.log(np.maximum(img / 255.0, 1e-6)) / np.log(1e-6)
    raw = od @ R.HED_FROM_RGB
    keep = (raw >= 0).all(-1)
    np.testing.assert_allclose((hed[keep].astype(np.float64) @ R.RGB_FROM_HED), od[keep], atol=1e-5)
