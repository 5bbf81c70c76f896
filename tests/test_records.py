"""The reference's record schema (dataset.py:17-71, data/scripts/CVPPP/utils.py:13-61) over a directory store: what
`create_dataset` writes is what `RecordDataset` reads back, key for key; on the GPU a RecordLoader batch equals the
collate function's resizes done with Pillow on the host."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def _samples(n=3, seed=0):
    rng = np.random.default_rng(seed)
    imgs, sems, inss = [], [], []
    for i in range(n):
        h, w = (53, 50) if i % 2 == 0 else (40, 64)
        k = 2 + i
        ins = np.zeros((h, w, k), np.uint8)
        for j in range(k):
            y0, x0 = rng.integers(0, h - 8), rng.integers(0, w - 8)
            ins[y0:y0 + 8, x0:x0 + 6, j] = 1
        imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        sems.append((ins.sum(2) > 0).astype(np.uint8))
        inss.append(ins)
    return imgs, sems, inss


def test_schema_roundtrip(tmp_path):
    import isa_amd  # noqa: F401
    from isa_amd.records import create_dataset, RecordDataset, DirStore
    imgs, sems, inss = _samples()
    root = str(tmp_path / "train-lmdb")
    create_dataset(root, imgs, sems, inss)
    keys = sorted(os.listdir(root))
    assert "num-samples" in keys and "image-1" in keys and "instance-annotation-3" in keys and "n_objects-2" in keys
    assert len(keys) == 1 + 6 * 3                                        # the reference's six keys per sample
    st = DirStore(root)
    assert st.get(b"num-samples") == b"3" and st.get("height-1") == b"53" and st.get("width-2") == b"64"
    ds = RecordDataset(root)
    assert len(ds) == 3
    for i in range(3):
        img, sem, ins, n = ds[i]
        assert np.array_equal(np.asarray(img), imgs[i]) and np.array_equal(sem, sems[i]) and np.array_equal(ins, inss[i])
        assert n == inss[i].shape[2] and ins.dtype == np.uint8
    assert st.get("image-9") is None


@pytest.mark.gpu
def test_record_loader_batch_equals_host_collate(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    import isa_amd  # noqa: F401
    from isa_amd.records import create_dataset, RecordDataset, RecordLoader
    imgs, sems, inss = _samples(4, seed=3)
    root = str(tmp_path / "val-lmdb")
    create_dataset(root, imgs, sems, inss)
    loader = RecordLoader(RecordDataset(root), 2, 32, 48, mode='test')
    batches = list(loader)
    assert len(batches) == 2 == len(loader)
    for b, (rgb, sem, ins, n) in enumerate(batches):
        assert rgb.shape == (2, 32, 48, 3) and sem.shape == (2, 32, 48) and ins.shape == (2, 32, 48, 32)
        for j in range(2):
            i = 2 * b + j
            want_rgb = np.asarray(Image.fromarray(imgs[i]).resize((48, 32), Image.BILINEAR))      # img_resizer
            assert np.array_equal(rgb[j].cpu().numpy(), want_rgb)
            want_sem = np.asarray(Image.fromarray(sems[i]).resize((48, 32), Image.NEAREST))       # ann_resizer
            assert np.array_equal(sem[j].cpu().numpy(), want_sem)
            k = inss[i].shape[2]
            assert int(n[j]) == k
            for p in range(32):
                want = np.asarray(Image.fromarray(inss[i][:, :, p]).resize((48, 32), Image.NEAREST)) if p < k else 0
                assert np.array_equal(ins[j, :, :, p].cpu().numpy(), want + np.zeros((32, 48), np.uint8))
    # training mode: shuffled, rank-sharded, augmentation keeps the label multiset
    tr = RecordLoader(RecordDataset(root), 1, 32, 32, mode='training', seed=5, rank=1, world=2)
    assert len(list(tr)) == 2


@pytest.mark.gpu
def test_fit_from_records(tmp_path):
    """train.py --data: Model.fit straight from record stores - uint8 RGB and uint8 targets, every resize / expansion /
    colour-space step on the device."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.model import Model
    from isa_amd.records import create_dataset, RecordDataset, RecordLoader
    rng = np.random.default_rng(1)
    imgs, sems, inss = [], [], []
    for i in range(4):
        ins = np.zeros((80, 80, 3), np.uint8)
        for j in range(3):
            ins[10 + 20 * j:25 + 20 * j, 8 + 12 * i:40 + 12 * i, j] = 1
        imgs.append(rng.integers(0, 256, (80, 80, 3), dtype=np.uint8)); sems.append((ins.sum(2) > 0).astype(np.uint8)); inss.append(ins)
    create_dataset(str(tmp_path / "training-lmdb"), imgs, sems, inss)
    create_dataset(str(tmp_path / "validation-lmdb"), imgs[:2], sems[:2], inss[:2])
    m = Model('CVPPP', 'ReSeg', 2, 32, use_instance_segmentation=True)
    tr = RecordLoader(RecordDataset(str(tmp_path / "training-lmdb")), 2, 64, 64, mode='training', seed=3)
    te = RecordLoader(RecordDataset(str(tmp_path / "validation-lmdb")), 2, 64, 64, mode='test')
    out = str(tmp_path / "run")
    m.fit('Multi', 0.5, 1.5, 2, 1.0, 0.001, 10.0, 0.5, 25, False, 'Adadelta', True, 2, None, tr, te, out, False)
    assert [f for f in os.listdir(out) if f.endswith(".pth")]
    assert torch.isfinite(m.model.store.flat).all()


def test_rank_shards_have_equal_step_counts_and_keep_short_batches():
    """len(ds) % world != 0 must not give one rank a step more than the others (it would wait in its gradient
    all-reduce forever), and a validation set smaller than world * batch must still yield a batch on every rank.  The
    reference drops nothing: a short last batch is filled by repeating its first image (dataset.py:330-333)."""
    import isa_amd  # noqa: F401
    from isa_amd.records import RecordLoader

    class DS(object):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    for n, world, bs in ((195, 8, 1), (49, 8, 8), (7, 2, 4), (16, 4, 2), (5, 1, 2)):
        shards = []
        for rank in range(world):
            for mode in ('training', 'test'):
                ld = RecordLoader(DS(n), bs, mode=mode, seed=3, rank=rank, world=world, device='cpu')
                ld.epoch = 1
                idx = ld.indices()
                shards.append((mode, len(idx), len(ld)))
                assert len(ld) >= 1
                if mode == 'test':
                    assert sorted(set(idx)) == sorted(set(range(n)[rank::world]) | set(idx))     # own slice is covered
        assert len(set(shards[0::2])) == 1 and len(set(shards[1::2])) == 1, (n, world, bs, shards)
        # every sample is seen by some rank
        seen = set()
        for rank in range(world):
            ld = RecordLoader(DS(n), bs, mode='test', rank=rank, world=world, device='cpu')
            seen |= set(ld.indices())
        assert seen == set(range(n))
