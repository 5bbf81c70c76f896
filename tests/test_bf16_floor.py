"""tests/bf16_grad_floor.py (the measurement behind test_bf16_gradients_256_vs_reference_f64's bounds): with the rounding
switched off, the harness must reproduce the reference's float64 gradients of the fixture - i.e. the floor it reports is
the effect of the rounding alone."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import golden_io as G          # noqa: E402
import bf16_grad_floor as B    # noqa: E402


def test_harness_without_rounding_reproduces_the_reference_gradients():
    z = np.load(os.path.join(ROOT, "tests", "golden", "train_64_drop_f64.npz"))
    g = B.oracle_gradients(z, 64)
    names = sorted(set(k.split("/")[1] for k in z.files if k.startswith("grad/")))
    gmax = max(float(np.sqrt(z["grad/%s/sums" % k][2])) for k in names)
    checked = 0
    for k in names:
        if float(np.sqrt(z["grad/%s/sums" % k][2])) <= 1e-6 * gmax:
            continue
        shape = tuple(int(v) for v in z["grad/%s/shape" % k])
        err, _ = G.compare(z, "grad/" + k, g[k].numpy().reshape(shape), 512)
        assert err < 1e-6, (k, err)
        checked += 1
    assert checked > 400


def test_bf16_storage_floor_is_tens_of_per_cent_at_64():
    cos, stats = B.floor(64)
    # measured: cosine 0.871; medians 0.46 (backbone), 0.55 (decoder), 0.16 (stems+heads)
    assert 0.5 < cos < 0.999
    assert stats["backbone"][0] > 0.05 and stats["decoder"][0] > 0.05
