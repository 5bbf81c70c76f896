"""isa_resize_nearest_u8 against oracle/resize_ref.py (pinned against Pillow): a byte gather, bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import resize_ref as R  # noqa: E402


def _lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd import lib as L
    return L


def run(L, x, h, w):
    n, h0, w0, c = x.shape
    d = torch.from_numpy(x).cuda()
    out = torch.full((n, h, w, c), 201, dtype=torch.uint8, device="cuda")
    assert L.lib().isa_resize_nearest_u8(L.ptr(d), n, h0, w0, c, L.ptr(out), h, w, L.stream_ptr()) == 0
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("shape,h,w", [((2, 530, 500, 32), 256, 256), ((3, 37, 91, 1), 64, 48), ((1, 16, 16, 3), 33, 7),
                                       ((2, 1, 1, 16), 5, 4), ((1, 300, 41, 5), 768, 2), ((2, 64, 64, 32), 64, 64)])
def test_matches_oracle_bit_for_bit(shape, h, w):
    L = _lib()
    rng = np.random.default_rng(sum(shape))
    x = rng.integers(0, 256, shape, dtype=np.uint8)
    np.testing.assert_array_equal(run(L, x, h, w), R.resize_nearest(x, h, w))


def test_invalid_arguments():
    L = _lib()
    t = torch.zeros(64, dtype=torch.uint8, device="cuda")
    u = torch.zeros(64, dtype=torch.uint8, device="cuda")
    f = L.lib().isa_resize_nearest_u8
    assert f(L.ptr(t), 1, 4, 4, 1, L.ptr(t), 4, 4, L.stream_ptr()) != 0           # in place
    assert f(L.ptr(t), 1, 4, 4, 1, L.ptr(u), 769, 4, L.stream_ptr()) != 0         # table capacity
    assert f(L.ptr(t), 1, 0, 4, 1, L.ptr(u), 4, 4, L.stream_ptr()) != 0
    assert f(None, 1, 4, 4, 1, L.ptr(u), 4, 4, L.stream_ptr()) != 0


def test_bilinear_resize_bit_exact_vs_oracle_and_pillow():
    """isa_resize_bilinear_u8 (the reference's `img_resizer`) against oracle/resize_ref.resize_bilinear (pinned to Pillow
    on the CPU) and, where Pillow is importable, against PIL.Image.resize(BILINEAR) itself - bit for bit, including
    CVPPP's 530x500 -> 256x256, up-scaling, unchanged axes and a batch."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.data import resize_bilinear
    rng = np.random.default_rng(5)
    cases = [(2, 530, 500, 256, 256), (1, 256, 256, 256, 256), (3, 37, 91, 64, 48), (1, 16, 16, 33, 7), (1, 300, 400, 512, 512),
             (1, 256, 300, 256, 256), (1, 301, 417, 256, 256), (1, 1024, 768, 256, 256)]
    cases += [(1,) + tuple(int(v) for v in rng.integers(1, 300, 4)) for _ in range(12)]
    for (n, h0, w0, h, w) in cases:
        img = rng.integers(0, 256, (n, h0, w0, 3), dtype=np.uint8)
        got = resize_bilinear(torch.from_numpy(img), (h, w)).cpu().numpy()
        np.testing.assert_array_equal(got, R.resize_bilinear(img, h, w), err_msg="%s" % ((n, h0, w0, h, w),))
        try:
            from PIL import Image
            np.testing.assert_array_equal(got[0], np.asarray(Image.fromarray(img[0]).resize((w, h), Image.BILINEAR)))
        except ImportError:
            pass
