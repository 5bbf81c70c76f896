"""The two non-D4 augmentations the reference ships enabled (training_settings.py:40 ROTATION, :50 CENTER_CUT) and the
rectangular D4 ops on the device, bit-exact against their oracles (oracle/rotate_ref.py, oracle/augment_ref.py - both
pinned against the installed Pillow), and the RecordLoader's training-mode sample against the reference's own sequence
of PIL calls (dataset.py:185-330) replayed on the host with the loader's recorded draws."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import augment_ref as AR   # noqa: E402
import rotate_ref as RR    # noqa: E402


def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd import data as D
    return D


@pytest.mark.parametrize("h,w,c", [(37, 53, 1), (64, 48, 32), (100, 75, 5), (530, 500, 32)])
def test_rotate_nearest_bit_exact(h, w, c):
    D = need_gpu()
    rs = np.random.RandomState(h + 7 * c)
    a = ((rs.rand(2, h, w, c) < 0.3) * rs.randint(1, 255, (2, h, w, c))).astype(np.uint8)
    for angle in range(-9, 10):
        got = D.rotate_nearest([torch.from_numpy(a)], angle)[0].cpu().numpy()
        for b in range(2):
            ref = RR.rotate_nearest(a[b], angle)
            assert got[b].shape == ref.shape and np.array_equal(got[b], ref), (angle, b)


@pytest.mark.parametrize("h,w", [(37, 53), (100, 75), (530, 500)])
def test_rotate_image_bit_exact(h, w):
    D = need_gpu()
    rs = np.random.RandomState(h)
    rgb = rs.randint(0, 256, (1, h, w, 3)).astype(np.uint8)
    for angle in range(-9, 10):
        for key in range(4):
            bg = D.background_colour(rgb[0], key)
            assert bg == RR.background(rgb[0], key)
            got = D.rotate_image(torch.from_numpy(rgb), angle, bg)[0].cpu().numpy()
            ref = RR.rotate_bilinear_bg(rgb[0], angle, bg)
            assert got.shape == ref.shape and np.array_equal(got, ref), (angle, key, int((got != ref).sum()))


@pytest.mark.parametrize("h,w,c", [(40, 64, 3), (53, 50, 32), (21, 34, 1)])
def test_d4_on_rectangular_images_bit_exact(h, w, c):
    D = need_gpu()
    rs = np.random.RandomState(h * w)
    a = rs.randint(0, 256, (3, h, w, c)).astype(np.uint8)
    for op in range(32):
        got = D.d4_augment([torch.from_numpy(a)], [op] * 3)[0].cpu().numpy()
        ref = AR.d4_batch(a, [op] * 3)
        assert got.shape == ref.shape and np.array_equal(got, ref), op
    with pytest.raises(AssertionError):                   # ops that disagree on exchanging the axes cannot share a call
        D.d4_augment([torch.from_numpy(a)], [0, 4, 0])


@pytest.mark.parametrize("H,W,h,w", [(120, 160, 32, 32), (50, 60, 32, 32), (200, 90, 40, 24)])
def test_center_cut_bit_exact(H, W, h, w):
    D = need_gpu()
    rs = np.random.RandomState(H + W)
    planes = np.zeros((H, W, 6), np.uint8)
    for i in range(6):
        y0, x0 = rs.randint(0, H - 12), rs.randint(0, W - 12)
        planes[y0:y0 + rs.randint(3, 12), x0:x0 + rs.randint(3, 12), i] = 1
    sem = (planes.sum(2) > 0).astype(np.uint8)
    image = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    total = int((planes.astype(np.float32).sum(2) == 1).sum())
    for pick in (0, total // 3, total - 1):
        seen = []

        def draw(count):
            seen.append(count)
            return pick

        r2, s2, p2, n2 = D.center_cut(torch.from_numpy(image[None]).cuda(), torch.from_numpy(sem[None, :, :, None]).cuda(),
                                      torch.from_numpy(planes[None]).cuda(), 6, draw, h, w, 32)
        assert seen == [total]
        img_c, sem_c, planes_c, keep = RR.center_cut(image, sem, planes, pick, h, w)
        assert n2 == len(keep)
        assert np.array_equal(r2[0].cpu().numpy(), img_c) and np.array_equal(s2[0, :, :, 0].cpu().numpy(), sem_c)
        got = p2[0].cpu().numpy()
        assert got.shape[2] == 32 and np.array_equal(got[:, :, :n2], planes_c) and not got[:, :, n2:].any()


def _host_sample(img, sem, ins, draws, out_h, out_w, k):
    """AlignCollate.__preprocess (dataset.py:175-330) with PIL / numpy on the host for the draws the loader made."""
    image = Image.fromarray(img)
    planes = [ins[:, :, i] for i in range(ins.shape[2])]
    op = draws["op"]

    def pil_d4(a, resample):
        im = a if isinstance(a, Image.Image) else Image.fromarray(a)
        if op & 1:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        if op & 2:
            im = im.transpose(Image.FLIP_TOP_BOTTOM)
        if op & 4:
            im = im.transpose(Image.TRANSPOSE)
        im = im.rotate(90 * ((op >> 3) & 3), resample=resample, expand=True)
        return im

    image = pil_d4(image, Image.BILINEAR)
    planes = [np.array(pil_d4(p, Image.NEAREST)) for p in planes]
    sem = np.array(pil_d4(sem, Image.NEAREST))
    angle = draws["angle"]
    if draws["bg_key"] is not None:
        src = np.array(image)
        bg = RR.background(src, draws["bg_key"])
        rgba = image.convert('RGBA').rotate(angle, resample=Image.BILINEAR, expand=True)
        back = Image.new('RGBA', rgba.size, (bg[0], bg[1], bg[2], 255))
        image = Image.composite(rgba, back, rgba).convert('RGB')
        planes = [np.array(Image.fromarray(p).rotate(angle, resample=Image.NEAREST, expand=True)) for p in planes]
        sem = np.array(Image.fromarray(sem).rotate(angle, resample=Image.NEAREST, expand=True))
    if draws["pick"] is not None:
        stack = np.stack(planes, 2)
        img_c, sem, stack, keep = RR.center_cut(np.array(image), sem, stack, draws["pick"], out_h, out_w)
        image = Image.fromarray(img_c)
        planes = [stack[:, :, i] for i in range(stack.shape[2])]
    rgb = np.asarray(image.resize((out_w, out_h), Image.BILINEAR))
    semr = np.asarray(Image.fromarray(sem).resize((out_w, out_h), Image.NEAREST))
    out = np.zeros((out_h, out_w, k), np.uint8)
    for i, p in enumerate(planes):
        out[:, :, i] = np.asarray(Image.fromarray(np.ascontiguousarray(p)).resize((out_w, out_h), Image.NEAREST))
    return rgb, semr, out, len(planes)


def test_record_loader_training_sample_equals_the_reference_sequence(tmp_path):
    need_gpu()
    from isa_amd.records import create_dataset, RecordDataset, RecordLoader
    rng = np.random.default_rng(11)
    imgs, sems, inss = [], [], []
    for i in range(6):
        h, w = [(150, 110), (96, 140), (128, 128)][i % 3]          # non-square originals, as in the reference's data
        kk = 3 + i % 3
        ins = np.zeros((h, w, kk), np.uint8)
        for j in range(kk):
            y0, x0 = rng.integers(0, h - 30), rng.integers(0, w - 30)
            ins[y0:y0 + rng.integers(8, 30), x0:x0 + rng.integers(8, 30), j] = 1
        imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)); sems.append((ins.sum(2) > 0).astype(np.uint8)); inss.append(ins)
    root = str(tmp_path / "training-lmdb")
    create_dataset(root, imgs, sems, inss)
    loader = RecordLoader(RecordDataset(root), 3, 32, 32, mode='training', seed=9)
    seen_ops, seen_angles = set(), set()
    for epoch in range(3):
        order = None
        for bi, (rgb, sem, ins, n) in enumerate(loader):
            order = loader.indices() if order is None else order
            for j, draws in enumerate(loader.last_draws):
                i = order[3 * bi + j]
                want_rgb, want_sem, want_ins, want_n = _host_sample(imgs[i], sems[i], inss[i], draws, 32, 32, 32)
                assert np.array_equal(rgb[j].cpu().numpy(), want_rgb), (epoch, bi, j, draws)
                assert np.array_equal(sem[j].cpu().numpy(), want_sem), (epoch, bi, j, draws)
                assert np.array_equal(ins[j].cpu().numpy(), want_ins), (epoch, bi, j, draws)
                assert int(n[j]) == want_n
                seen_ops.add(draws["op"]); seen_angles.add(draws["angle"])
    assert len(seen_ops) > 5 and len(seen_angles) > 5
