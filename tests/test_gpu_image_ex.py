"""isa_image_ex (ImageEx + ToTensor + Standardization on device) against the numpy oracle
(oracle/image_ex_ref.py; pinned by published colour values only: parity with scikit-image itself is unpinned),
and the uint8-RGB entrance of the drop-in class against its float [B,21,H,W] entrance."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import image_ex_ref as IX   # noqa: E402
import reseg_ref as R       # noqa: E402


def _setup(dtype):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.reseg import ReSeg
    m = ReSeg(2, False, dtype=dtype)
    m.load_state_dict(R.synth_state_dict(23, False))
    m.eval()
    return m


def _images(n, h, w):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    special = [(0, 0, 0), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 0, 255), (128, 128, 128),
               (1, 1, 1), (10, 10, 11), (254, 255, 255), (0, 255, 255), (255, 255, 0), (3, 2, 1)]
    for i, c in enumerate(special):                     # saturated / grey / near-threshold pixels
        img[0, 0, i] = c
    return img


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_image_ex_kernel_matches_oracle(dtype):
    m = _setup(dtype)
    img = _images(2, 32, 48)
    m.engine.begin(bn_train=False, record=False)
    a = m.net.image_ex(torch.from_numpy(img))
    torch.cuda.synchronize()
    got = a.buf.float().cpu().numpy()                   # [n,h,w,24]
    ref = np.moveaxis(IX.image_ex_standardized(img), 1, -1)        # [n,h,w,21]
    assert got.shape[-1] == 24 and np.all(got[..., 21:] == 0), "pad channels must be zero"
    scale = np.abs(ref).reshape(-1, 21).max(0) + 1e-6
    err = (np.abs(got[..., :21] - ref).reshape(-1, 21) / scale).max(0)
    tol = 2e-5 if dtype == torch.float32 else 8e-3      # bf16: one storage rounding (2^-8 relative)
    assert (err < tol).all(), err


def test_uint8_entrance_equals_float_entrance():
    m = _setup(torch.float32)
    img = _images(2, 64, 64)
    x21 = torch.from_numpy(IX.image_ex_standardized(img))
    with torch.no_grad():
        sem_a, arg_a = m(False, torch.from_numpy(img))
        sem_b, arg_b = m(False, x21)
    torch.cuda.synchronize()
    d = float((sem_a - sem_b).abs().max() / sem_b.abs().max())
    assert d < 1e-3, d
    assert float((arg_a != arg_b).float().mean()) < 1e-3        # index map: identical up to argmax near-ties
