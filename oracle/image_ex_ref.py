"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the reference's 21-channel colour expansion, the step
immediately before the hot path.

Reference: `ImageEx.__call__` (code/lib/utils.py:90-113) concatenates, per pixel,
    [rgb (uint8 values 0..255), rgb2lab, rgb2hsv, rgb2yuv, rgb2ycbcr, rgb2hed, rgb2yiq]  -> float32 [H,W,21]
then `transforms.ToTensor()` (float ndarray: transpose to [21,H,W], no rescale) and
`Standardization` (code/lib/preprocess.py:192-195): (x - 0.5) * 2.

PARITY UNPINNED: the six conversions are `skimage.color` functions; scikit-image is not importable in this
image, the reference pins no version and holds no test vectors for this step.  The formulas below restate the
published algorithms of scikit-image >= 0.17 (`skimage/color/colorconv.py`: sRGB D65 / 2-degree observer
matrices, CIE L*a*b*, ITU-R BT.601 YUV/YCbCr/YIQ matrices, Ruifrok & Johnston haematoxylin-eosin-DAB
deconvolution with the log(1e-6) normalisation).  Before 0.17 `rgb2hed` used `-log10(rgb + 2)` instead: a
different channel 15-17.  tests/test_oracle_image_ex.py pins this file against published colour-science known
answers (white/black/primaries), not against scikit-image itself.
"""
import numpy as np

XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423],
                         [0.212671, 0.715160, 0.072169],
                         [0.019334, 0.119193, 0.950227]])
D65_WHITE = np.array([0.95047, 1.0, 1.08883])
YUV_FROM_RGB = np.array([[0.299, 0.587, 0.114],
                         [-0.14714119, -0.28886916, 0.43601035],
                         [0.61497538, -0.51496512, -0.10001026]])
YIQ_FROM_RGB = np.array([[0.299, 0.587, 0.114],
                         [0.59590059, -0.27455667, -0.32134392],
                         [0.21153661, -0.52273617, 0.31119955]])
YCBCR_FROM_RGB = np.array([[65.481, 128.553, 24.966],
                           [-37.797, -74.203, 112.0],
                           [112.0, -93.786, -18.214]])
RGB_FROM_HED = np.array([[0.65, 0.70, 0.29],
                         [0.07, 0.99, 0.11],
                         [0.27, 0.57, 0.78]])
HED_FROM_RGB = np.linalg.inv(RGB_FROM_HED)


def rgb2lab(f):
    lin = np.where(f > 0.04045, ((f + 0.055) / 1.055) ** 2.4, f / 12.92)
    xyz = lin @ XYZ_FROM_RGB.T / D65_WHITE
    t = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    x, y, z = t[..., 0], t[..., 1], t[..., 2]
    return np.stack([116.0 * y - 16.0, 500.0 * (x - y), 200.0 * (y - z)], -1)


def rgb2hsv(f):
    v = f.max(-1)
    delta = v - f.min(-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(delta == 0, 0.0, delta / v)
        r, g, b = f[..., 0], f[..., 1], f[..., 2]
        h = np.zeros_like(v)
        # later assignments win on ties, as in the reference implementation (red, then green, then blue)
        h = np.where(r == v, (g - b) / delta, h)
        h = np.where(g == v, 2.0 + (b - r) / delta, h)
        h = np.where(b == v, 4.0 + (r - g) / delta, h)
        h = (h / 6.0) % 1.0
    h = np.where(delta == 0, 0.0, h)
    return np.stack([h, s, v], -1)


def rgb2hed(f):
    f = np.maximum(f, 1e-6)
    stains = (np.log(f) / np.log(1e-6)) @ HED_FROM_RGB
    return np.maximum(stains, 0.0)


def image_ex(rgb_u8):
    """uint8 [..., H, W, 3] -> float32 [..., H, W, 21] (ImageEx), before ToTensor/Standardization."""
    rgb_u8 = np.asarray(rgb_u8)
    assert rgb_u8.dtype == np.uint8 and rgb_u8.shape[-1] == 3
    f = rgb_u8.astype(np.float64) / 255.0
    parts = [rgb_u8.astype(np.float64), rgb2lab(f), rgb2hsv(f), f @ YUV_FROM_RGB.T,
             f @ YCBCR_FROM_RGB.T + np.array([16.0, 128.0, 128.0]), rgb2hed(f), f @ YIQ_FROM_RGB.T]
    return np.concatenate(parts, -1).astype(np.float32)


def image_ex_standardized(rgb_u8):
    """The network input of the reference: [..., 21, H, W] float32 = (ImageEx - 0.5) * 2."""
    x = image_ex(rgb_u8)
    x = np.moveaxis(x, -1, -3)
    return ((x - np.float32(0.5)) * np.float32(2.0)).astype(np.float32)
