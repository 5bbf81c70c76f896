"""isa_collate_targets (tail of AlignCollate.__call__, code/lib/dataset.py:349-379) against oracle/collate_ref.py:
integer index work, bit-exact; and the compact-target entrance of the trainer."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import collate_ref as R  # noqa: E402


def _lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd import lib as L
    return L


def run(L, ins, sem):
    n, h, w, k = ins.shape
    d_ins = torch.from_numpy(ins).cuda()
    d_sem = torch.from_numpy(sem).cuda() if sem is not None else None
    out = torch.full((n, k, h, w), -7, dtype=torch.int64, device="cuda")
    oh = torch.full((n, 2, h, w), -7, dtype=torch.int64, device="cuda")
    rc = L.lib().isa_collate_targets(L.ptr(d_ins), L.ptr(d_sem), n, h, w, k, L.ptr(out), L.ptr(oh) if sem is not None else None,
                                     L.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    return out.cpu().numpy(), oh.cpu().numpy()


@pytest.mark.parametrize("shape", [(2, 16, 16, 32), (1, 7, 9, 32), (3, 5, 3, 5), (2, 33, 31, 16), (1, 1, 1, 1), (2, 20, 13, 48)])
def test_matches_oracle_bit_for_bit(shape):
    L = _lib()
    rng = np.random.default_rng(sum(shape))
    ins = (rng.random(shape) < 0.3).astype(np.uint8) * rng.choice(np.array([1, 255], np.uint8), shape)
    sem = (rng.random(shape[:3]) < 0.5).astype(np.uint8)
    want_oh, want_ins = R.collate_targets(ins, sem)
    got_ins, got_oh = run(L, ins, sem)
    np.testing.assert_array_equal(got_ins, want_ins)
    np.testing.assert_array_equal(got_oh, want_oh)


def test_full_size_and_null_semantic_map():
    L = _lib()
    rng = np.random.default_rng(5)
    ins = (rng.random((4, 256, 256, 32)) < 0.2).astype(np.uint8)
    got_ins, untouched = run(L, ins, None)
    assert (untouched == -7).all()
    # size-independent property: a permutation - every plane equals its source slice, totals preserved
    assert got_ins.sum() == int(ins.sum())
    for b, k in [(0, 0), (3, 31), (2, 17)]:
        np.testing.assert_array_equal(got_ins[b, k], ins[b, :, :, k])


def test_invalid_arguments():
    L = _lib()
    t = torch.zeros(16, dtype=torch.uint8, device="cuda")
    o = torch.zeros(16, dtype=torch.int64, device="cuda")
    f = L.lib().isa_collate_targets
    assert f(None, None, 1, 2, 2, 4, L.ptr(o), None, L.stream_ptr()) != 0
    assert f(L.ptr(t), None, 1, 2, 2, 0, L.ptr(o), None, L.stream_ptr()) != 0
    assert f(L.ptr(t), L.ptr(t), 1, 2, 2, 4, L.ptr(o), None, L.stream_ptr()) != 0      # sem without sem_out
    assert f(L.ptr(t), None, 1, 2, 2, 253, L.ptr(o), None, L.stream_ptr()) != 0


def test_trainer_accepts_compact_targets():
    """Same batch as int64 tensors and as the uint8 arrays the collate function holds before its last lines: the
    expanded targets are identical, so the deterministic parts of the step (semantic losses) agree exactly in fp32."""
    L = _lib()
    from isa_amd.reseg import ReSeg
    from isa_amd.trainer import Trainer
    from isa_amd.data import synth_batch
    x, sem, ins, n = synth_batch(2, 64, 64, seed=11)
    ins_u8 = ins.permute(0, 2, 3, 1).contiguous().to(torch.uint8)           # [B,H,W,K]
    sem_u8 = sem[:, 1].contiguous().to(torch.uint8)                          # [B,H,W]
    sel = [list(range(int(k))) for k in n.view(-1)]
    m = ReSeg(2, True, dtype=torch.float32)
    m.reset_parameters(seed=3)
    m.train()
    m.head.drop_rate = 0.0
    tr = Trainer(m)
    sem_e, ins_e = None, None
    m.engine.begin(bn_train=True, record=False)
    sem_e, ins_e = m.net.collate_targets(sem_u8, ins_u8)
    torch.cuda.synchronize()
    assert torch.equal(ins_e.cpu(), ins) and torch.equal(sem_e.cpu(), sem)
    a = tr.forward_backward(x, sem, ins, n, selected_idx=sel)
    sa = a["sem"].clone()
    b = tr.forward_backward(x, sem_u8, ins_u8, n, selected_idx=sel)
    torch.cuda.synchronize()
    assert torch.isfinite(b["sem"]).all() and torch.isfinite(b["head"]).all()
    assert torch.allclose(sa, b["sem"], rtol=1e-4, atol=1e-5)
    out = tr.train_step_graphed(x, sem_u8, ins_u8, n, selected_idx=sel)     # warm (eager)
    out = tr.train_step_graphed(x, sem_u8, ins_u8, n, selected_idx=sel)     # capture + replay
    torch.cuda.synchronize()
    assert torch.isfinite(out["sem"]).all()


def test_device_annotation_pipeline_matches_the_oracle_chain():
    """augment (source resolution) -> nearest resize -> int64 planes / one-hot, the order of dataset.py:185-379."""
    L = _lib()
    import augment_ref as A
    import resize_ref as Z
    from isa_amd.data import device_collate_targets
    rng = np.random.default_rng(9)
    n, s0, k, size = 3, 90, 32, 64
    planes = (rng.random((n, s0, s0, k)) < 0.3).astype(np.uint8)
    sem = (planes.sum(-1) > 0).astype(np.uint8)
    ops = [13, 0, 30]
    sem_oh, ins = device_collate_targets(torch.from_numpy(planes), torch.from_numpy(sem), ops, size)
    p_ref = Z.resize_nearest(A.d4_batch(planes, ops), size, size)
    s_ref = Z.resize_nearest(A.d4_batch(sem[..., None], ops), size, size)[..., 0]
    want_oh, want_ins = R.collate_targets(p_ref, s_ref)
    np.testing.assert_array_equal(ins.cpu().numpy(), want_ins)
    np.testing.assert_array_equal(sem_oh.cpu().numpy(), want_oh)
    # non-square sources without ops (validation mode of the reference: no augmentation)
    planes2 = (rng.random((2, 53, 50, k)) < 0.3).astype(np.uint8)
    sem2 = (planes2.sum(-1) > 0).astype(np.uint8)
    oh2, ins2 = device_collate_targets(torch.from_numpy(planes2), torch.from_numpy(sem2), None, size)
    w_oh, w_ins = R.collate_targets(Z.resize_nearest(planes2, size, size), Z.resize_nearest(sem2[..., None], size, size)[..., 0])
    np.testing.assert_array_equal(ins2.cpu().numpy(), w_ins)
    np.testing.assert_array_equal(oh2.cpu().numpy(), w_oh)
