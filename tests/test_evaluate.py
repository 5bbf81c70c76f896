"""evaluate.py (reference code/evaluate.py:18-56 metrics + :59-111 directory walk): known answers, the joint-histogram
Best Dice against the pairwise definition it replaces, the reference's empty-input behaviour, and one end-to-end walk
over a directory in the reference's on-disk layout."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import evaluate as EV  # noqa: E402


def pairwise_bd(a, b):
    """The definition: for every object of `a`, the best Dice over the objects of `b`; mean over `a`."""
    best = []
    for i in sorted(set(np.unique(a)) - {0}):
        ma = a == i
        best.append(max(2.0 * np.sum(ma & (b == j)) / (np.sum(ma) + np.sum(b == j)) for j in sorted(set(np.unique(b)) - {0})))
    return float(np.mean(best))


def test_known_answers():
    gt = np.array([[1, 1, 0, 0], [1, 1, 0, 2], [0, 0, 2, 2]])
    assert EV.calc_sbd(gt, gt) == 1.0
    assert EV.calc_sbd(gt, np.where(gt == 1, 7, np.where(gt == 2, 3, 0))) == 1.0          # label names do not matter
    pred = np.array([[1, 1, 0, 0], [0, 0, 0, 2], [0, 0, 2, 2]])                            # object 1 half found
    # gt->pred: obj1 2*2/(4+2)=2/3, obj2 1  -> 5/6 ; pred->gt identical pairs -> 5/6
    assert abs(EV.calc_bd(gt, pred) - 5 / 6) < 1e-12 and abs(EV.calc_sbd(gt, pred) - 5 / 6) < 1e-12
    merged = np.where(gt > 0, 1, 0)                                                        # one blob over both objects
    # gt->merged: 2*4/(4+7)=8/11 and 2*3/(3+7)=6/10 -> mean; merged->gt: max(8/11, 6/10) = 8/11
    assert abs(EV.calc_bd(gt, merged) - (8 / 11 + 0.6) / 2) < 1e-12
    assert abs(EV.calc_bd(merged, gt) - 8 / 11) < 1e-12
    assert abs(EV.calc_sbd(gt, merged) - (8 / 11 + 0.6) / 2) < 1e-12
    assert EV.calc_dic(5, np.array(3)) == 2 and EV.calc_dic(3, 5) == 2
    assert EV.calc_dice(np.array([1, 1, 0, 0], bool), np.array([1, 0, 1, 0], bool)) == 0.5


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_histogram_best_dice_equals_pairwise_definition(seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 9, (37, 53)) * (rng.random((37, 53)) < 0.7)
    b = rng.integers(0, 6, (37, 53)) * 3                               # non-contiguous ids
    assert abs(EV.calc_bd(a, b) - pairwise_bd(a, b)) < 1e-12
    assert abs(EV.calc_bd(b, a) - pairwise_bd(b, a)) < 1e-12
    assert abs(EV.calc_sbd(a, b) - min(pairwise_bd(a, b), pairwise_bd(b, a))) < 1e-12


def test_empty_inputs_behave_like_the_reference():
    gt = np.array([[1, 1], [0, 2]])
    none = np.zeros_like(gt)
    with pytest.raises(ValueError):                                    # np.max([]) in evaluate.py:45
        EV.calc_bd(gt, none)
    assert np.isnan(EV.calc_bd(none, gt))                              # np.mean([]) in evaluate.py:48
    with pytest.raises(ZeroDivisionError):                             # float(0)/float(0) in evaluate.py:27
        EV.calc_dice(none.astype(bool), none.astype(bool))
    with pytest.raises(ValueError):
        EV.calc_bd(gt, np.zeros((3, 3), int))


def test_directory_walk(tmp_path):
    from PIL import Image
    data, pred = tmp_path / "data", tmp_path / "pred"
    a1 = data / "raw/CVPPP/CVPPP2017_LSC_training/training/A1"
    meta = data / "metadata/CVPPP"
    a1.mkdir(parents=True); meta.mkdir(parents=True)
    rng = np.random.default_rng(0)
    stems = ["plant001", "plant002", "plant003"]
    (meta / "validation_image_paths.txt").write_text("".join("x/%s_rgb.png\n" % s for s in stems))
    (meta / "number_of_instances.txt").write_text("".join("%s,%d\n" % (s, 3) for s in stems))
    want_sbd, want_fg = [], []
    for k, s in enumerate(stems):
        gt = np.zeros((20, 24), np.uint8)
        gt[2:8, 2:10], gt[10:18, 4:12], gt[4:12, 14:22] = 1, 2, 3
        Image.fromarray(gt).save(a1 / (s + "_label.png"))
        Image.fromarray((gt > 0).astype(np.uint8)).save(a1 / (s + "_fg.png"))
        if k == 2:
            continue                                                   # no prediction for the third image: skipped
        p = np.roll(gt, k + 1, axis=1)
        d = pred / (s + "_rgb")
        d.mkdir(parents=True)
        Image.fromarray(p).save(d / (s + "_rgb-ins_mask.png"))
        Image.fromarray(((p > 0) * 255).astype(np.uint8)).save(d / (s + "_rgb-fg_mask.png"))
        np.save(d / (s + "_rgb-n_objects.npy"), np.array(3 + k))
        want_sbd.append(min(pairwise_bd(gt, p), pairwise_bd(p, gt)))
        want_fg.append(2.0 * np.sum((gt > 0) & (p > 0)) / (np.sum(gt > 0) + np.sum(p > 0)))
    sbds, dics, fg, scored = EV.evaluate_cvppp(str(pred), str(data))
    assert scored == ["plant001_rgb", "plant002_rgb"]
    np.testing.assert_allclose(sbds, want_sbd, rtol=0, atol=1e-12)
    np.testing.assert_allclose(fg, want_fg, rtol=0, atol=1e-12)
    assert [int(d) for d in dics] == [0, 1]
    # foreground-only scoring of a directory that holds no instance outputs (what pred_list.py writes)
    os.remove(pred / "plant001_rgb" / "plant001_rgb-n_objects.npy")
    _, _, fg2, scored2 = EV.evaluate_cvppp(str(pred), str(data), fg_only=True)
    assert scored2 == ["plant001_rgb", "plant002_rgb"]
    np.testing.assert_allclose(fg2, want_fg, rtol=0, atol=1e-12)
