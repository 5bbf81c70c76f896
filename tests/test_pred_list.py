"""pred_list.py (batched foreground prediction, reference code/pred_list.py + lib/prediction.py:47-50): the
nearest-neighbour up-sampling rule on the CPU, and an end-to-end synthetic run on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_nearest_upsample_matches_inter_nearest_rule():
    import pred_list
    m = np.arange(6 * 5).reshape(6, 5).astype(np.uint8)
    for oh, ow in ((6, 5), (12, 10), (13, 7), (4, 3), (531, 500)):
        u = pred_list.nearest_upsample(m, oh, ow)
        assert u.shape == (oh, ow)
        for y in (0, oh // 3, oh - 1):
            for x in (0, ow // 2, ow - 1):          # cv2.INTER_NEAREST: src = min(floor(dst * src_size / dst_size), src_size - 1)
                assert u[y, x] == m[min(int(np.floor(y * 6 / oh)), 5), min(int(np.floor(x * 5 / ow)), 4)]


@pytest.mark.gpu
def test_pred_list_synthetic_run(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    out = str(tmp_path / "out")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "pred_list.py"), "--synthetic", "6", "--batch", "4", "--output", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = sorted(os.listdir(out))
    assert len(names) == 6
    for n in names:
        img = np.asarray(Image.open(os.path.join(out, n, n + ".png")))
        mask = np.asarray(Image.open(os.path.join(out, n, n + "-fg_mask.png")))
        assert mask.shape == img.shape[:2]                      # up-sampled back to the original size
        assert set(np.unique(mask)) <= {0, 255}


@pytest.mark.gpu
def test_pred_image_in_palette_png_out(tmp_path):
    """pred.py contract (code/pred.py:114-123): an image file in, <name>-fg_mask.png out - a palette ('P') PNG at the
    ORIGINAL size holding {0, 255}; and the same mask pred_list.py writes for that image."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (301, 417, 3), dtype=np.uint8)
    src = str(tmp_path / "leaf_007.png")
    Image.fromarray(img).save(src)
    out = str(tmp_path / "single")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "pred.py"), "--image", src, "--output", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    m = Image.open(os.path.join(out, "leaf_007-fg_mask.png"))
    assert m.mode == "P" and m.size == (417, 301)
    # what the reference's own conversion chain yields for a {0, 255} float image
    want_values = set(np.unique(np.asarray(Image.fromarray(np.array([[0.0, 255.0]], dtype=np.float32)).convert("P"))))
    assert set(np.unique(np.asarray(m))) <= want_values
    lst = str(tmp_path / "val_list.txt")
    open(lst, "w").write(src + "\n")
    out2 = str(tmp_path / "listed")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "pred_list.py"), "--lst", lst, "--output", out2],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    m2 = np.asarray(Image.open(os.path.join(out2, "leaf_007", "leaf_007-fg_mask.png")))
    assert np.array_equal(np.asarray(m.convert("L")) > 127, m2 > 127)
