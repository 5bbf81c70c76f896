"""isa_d4_augment against oracle/augment_ref.py (itself pinned against Pillow): byte permutations, bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import augment_ref as R  # noqa: E402


def _lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd import lib as L
    return L


def run(L, x, ops):
    d = torch.from_numpy(x).cuda()
    out = torch.full_like(d, 201)
    o = torch.tensor(ops, dtype=torch.int32, device="cuda")
    n, s, _, c = x.shape
    assert L.lib().isa_d4_augment(L.ptr(d), L.ptr(out), n, s, s, c, 0, L.ptr(o), L.stream_ptr()) == 0
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("c", [1, 3, 32, 16, 5])
@pytest.mark.parametrize("s", [1, 7, 16, 33])
def test_all_op_codes(c, s):
    L = _lib()
    rng = np.random.default_rng(c * 100 + s)
    x = rng.integers(0, 256, (32, s, s, c), dtype=np.uint8)
    ops = list(range(32))
    np.testing.assert_array_equal(run(L, x, ops), R.d4_batch(x, ops))


def test_full_size_batch_and_round_trip():
    L = _lib()
    rng = np.random.default_rng(1)
    x = (rng.random((16, 256, 256, 32)) < 0.3).astype(np.uint8)
    ops = [int(v) for v in rng.integers(0, 32, 16)]
    y = run(L, x, ops)
    for b in (0, 7, 15):
        np.testing.assert_array_equal(y[b], R.d4(x[b], ops[b]))
    assert y.sum() == x.sum()                                   # a permutation
    # size-independent property: four quarter turns are the identity; a flip twice is the identity
    z = x
    for _ in range(4):
        z = run(L, z, [1 << 3] * 16)
    np.testing.assert_array_equal(z, x)
    np.testing.assert_array_equal(run(L, run(L, x, [1] * 16), [1] * 16), x)
    np.testing.assert_array_equal(run(L, run(L, x, [4] * 16), [4] * 16), x)


def test_invalid_arguments():
    L = _lib()
    t = torch.zeros(64, dtype=torch.uint8, device="cuda")
    u = torch.zeros(64, dtype=torch.uint8, device="cuda")
    o = torch.zeros(4, dtype=torch.int32, device="cuda")
    f = L.lib().isa_d4_augment
    assert f(L.ptr(t), L.ptr(t), 1, 4, 4, 4, 0, L.ptr(o), L.stream_ptr()) != 0          # in place
    assert f(L.ptr(t), L.ptr(u), 1, 4, 4, 4, 0, None, L.stream_ptr()) != 0
    assert f(L.ptr(t), L.ptr(u), 0, 4, 4, 4, 0, L.ptr(o), L.stream_ptr()) != 0
    assert f(None, L.ptr(u), 1, 4, 4, 4, 0, L.ptr(o), L.stream_ptr()) != 0


def test_host_helper_shares_one_op_per_image():
    L = _lib()
    from isa_amd.data import d4_augment
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (4, 32, 32, 3), dtype=np.uint8)
    sem = rng.integers(0, 2, (4, 32, 32), dtype=np.uint8)
    ins = rng.integers(0, 2, (4, 32, 32, 32), dtype=np.uint8)
    ops = [5, 0, 27, 14]
    a, b, c = d4_augment([torch.from_numpy(rgb), torch.from_numpy(sem), torch.from_numpy(ins)], ops)
    np.testing.assert_array_equal(a.cpu().numpy(), R.d4_batch(rgb, ops))
    np.testing.assert_array_equal(b.cpu().numpy(), R.d4_batch(sem[..., None], ops)[..., 0])
    np.testing.assert_array_equal(c.cpu().numpy(), R.d4_batch(ins, ops))
