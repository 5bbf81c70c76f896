"""Entry points: Model.fit (2 tiny epochs on synthetic data, best checkpoint saved, reloadable) and
Model.predict argument checks like the reference (model.py:39-40,468)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def test_fit_and_predict(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.model import Model
    from isa_amd.data import SyntheticLoader, synth_batch
    with pytest.raises(AssertionError):
        Model('COCO', 'ReSeg', 2, 32)
    m = Model('CVPPP', 'ReSeg', 2, 32, use_instance_segmentation=True)
    m.model.head.drop_rate = 0.5
    tr, te = SyntheticLoader(2, 2, 64, 64, seed=1), SyntheticLoader(1, 2, 64, 64, seed=2)
    before = m.model.state_dict()["base.inc.conv.conv.down_conv_0.conv.3.weight"].clone()
    m.fit('Multi', 0.5, 1.5, 2, 1.0, 0.001, 10.0, 0.5, 25, False, 'Adadelta', True, 2, None, tr, te, str(tmp_path), False)
    after = m.model.state_dict()["base.inc.conv.conv.down_conv_0.conv.3.weight"]
    assert not torch.equal(before, after) and torch.isfinite(after).all()
    ckpts = [f for f in os.listdir(str(tmp_path)) if f.endswith(".pth")]
    assert ckpts and os.path.isfile(os.path.join(str(tmp_path), "training.log"))
    m2 = Model('CVPPP', 'ReSeg', 2, 32, use_instance_segmentation=False,
               load_model_path=os.path.join(str(tmp_path), sorted(ckpts)[-1]))
    x = synth_batch(2, 64, 64, seed=3)[0]
    prob = m2.predict(x)
    assert prob.shape == (2, 2, 64, 64) and torch.allclose(prob.sum(1), torch.ones(2, 64, 64), atol=1e-5)
    with pytest.raises(AssertionError):
        m2.predict(x[0])                      # 4-D input check (model.py:468)


def test_device_prefetcher_yields_the_same_batches_in_order():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.data import DevicePrefetcher, SyntheticLoader
    host = list(SyntheticLoader(3, 2, 32, 32, seed=4, compact=True))
    loader = SyntheticLoader(3, 2, 32, 32, seed=4, compact=True)
    got = list(DevicePrefetcher(loader))
    torch.cuda.synchronize()
    assert len(got) == 3 == len(DevicePrefetcher(loader))
    for h, d in zip(host, got):
        for a, b in zip(h[:3], d[:3]):
            assert b.is_cuda and torch.equal(a, b.cpu())
        assert torch.equal(h[3], d[3]) and not d[3].is_cuda
    assert list(DevicePrefetcher([])) == []


def test_fit_semantic_only_model(tmp_path):
    """the reference trains and validates models without the instance head too (its default,
    model.py:244-270, 426-435): CE / Dice are logged in both phases and drive the plateau scheduler."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.model import Model
    from isa_amd.data import SyntheticLoader
    m = Model('CVPPP', 'ReSeg', 2, 32, use_instance_segmentation=False)
    tr, te = SyntheticLoader(2, 2, 64, 64, seed=1), SyntheticLoader(1, 2, 64, 64, seed=2)
    m.fit('Multi', 0.5, 1.5, 2, 1.0, 0.001, 10.0, 0.5, 25, False, 'Adadelta', True, 1, None, tr, te, str(tmp_path), False)
    log = open(os.path.join(str(tmp_path), "validation.log")).read().strip().splitlines()
    assert len(log) == 2 and 0.0 < float(log[1].split(",")[1]) < 1.5        # Dice Cost of the validation batch
    assert [f for f in os.listdir(str(tmp_path)) if f.endswith(".pth")]
