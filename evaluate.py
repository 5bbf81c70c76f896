#!/usr/bin/env python3
"""evaluate.py — offline CVPPP metrics over a prediction directory, the reference's code/evaluate.py:1-112
(flags --pred_dir / --dataset; prints MEAN SBD, MEAN |DIC|, MEAN FG DICE over the validation images that have
predictions).  Host-side numpy, outside the GPU path.

Same function names and results as the reference (`calc_dic` :18, `calc_dice` :22, `calc_bd` :31, `calc_sbd` :52);
`calc_bd` is computed from ONE joint histogram of (gt label, predicted label) pairs instead of a mask product per
pair (O(HW) instead of O(G*P*HW)); tests/test_evaluate.py checks it against the pairwise definition.

Differences, all additive: `--data_root` (the reference hard-codes ../data), and `--fg_only` to score the
foreground masks that pred_list.py writes when no instance outputs exist (the reference skips such images: its
instance clustering is dead at HEAD, see pred_list.py)."""
import argparse
import os

import numpy as np


def calc_dic(n_objects_gt, n_objects_pred):
    """|difference in count| (evaluate.py:18-19)."""
    return np.abs(n_objects_gt - n_objects_pred)


def calc_dice(gt_seg, pred_seg):
    """2|A.B| / (|A| + |B|) of two boolean masks (evaluate.py:22-28); empty-vs-empty divides by zero as there."""
    gt_seg, pred_seg = np.asarray(gt_seg), np.asarray(pred_seg)
    nom = 2 * np.sum(gt_seg * pred_seg)
    denom = np.sum(gt_seg) + np.sum(pred_seg)
    return float(nom) / float(denom)


def _pair_dice(ins_a, ins_b):
    """dice[i, j] between object i of ins_a and object j of ins_b (label 0 = background), from one joint histogram."""
    a, b = np.asarray(ins_a).reshape(-1), np.asarray(ins_b).reshape(-1)
    if a.shape != b.shape:
        raise ValueError("label maps differ in size: %s vs %s" % (np.shape(ins_a), np.shape(ins_b)))
    ids_a, inv_a = np.unique(a, return_inverse=True)
    ids_b, inv_b = np.unique(b, return_inverse=True)
    joint = np.bincount(inv_a * len(ids_b) + inv_b, minlength=len(ids_a) * len(ids_b)).reshape(len(ids_a), len(ids_b))
    size_a, size_b = joint.sum(1), joint.sum(0)
    keep_a, keep_b = ids_a != 0, ids_b != 0
    inter = joint[keep_a][:, keep_b].astype(np.float64)
    return 2.0 * inter / (size_a[keep_a][:, None] + size_b[keep_b][None, :]).astype(np.float64)


def calc_bd(ins_seg_gt, ins_seg_pred):
    """Best Dice (evaluate.py:31-49): mean over the objects of the first map of their best Dice with any object of
    the second.  No object in the second map: ValueError, as the reference's np.max of an empty list; no object in
    the first: nan, as its np.mean of an empty list."""
    dice = _pair_dice(ins_seg_gt, ins_seg_pred)
    if dice.shape[0] == 0:
        return float("nan")
    if dice.shape[1] == 0:
        raise ValueError("zero-size array to reduction operation maximum which has no identity "
                         "(the second label map holds no object)")
    return float(np.mean(dice.max(1)))


def calc_sbd(ins_seg_gt, ins_seg_pred):
    """Symmetric Best Dice (evaluate.py:52-56)."""
    return min(calc_bd(ins_seg_gt, ins_seg_pred), calc_bd(ins_seg_pred, ins_seg_gt))


def evaluate_cvppp(pred_dir, data_root, fg_only=False):
    """Walk the validation list like evaluate.py:59-111.  Returns (sbds, dics, fg_dices, names_scored)."""
    from PIL import Image
    names = np.atleast_1d(np.loadtxt(os.path.join(data_root, 'metadata/CVPPP/validation_image_paths.txt'),
                                     dtype='str', delimiter=','))
    names = [os.path.splitext(os.path.basename(n))[0] for n in names]
    img_dir = os.path.join(data_root, 'raw/CVPPP/CVPPP2017_LSC_training/training/A1')
    counts = None
    sbds, dics, fg_dices, scored = [], [], [], []
    for name in names:
        base = os.path.join(pred_dir, name, name)
        has_ins = os.path.isfile(base + '-n_objects.npy')
        if not has_ins and not (fg_only and os.path.isfile(base + '-fg_mask.png')):
            continue                                             # evaluate.py:72-74
        stem = name.replace('_rgb', '')
        if has_ins and not fg_only:
            if counts is None:
                counts = np.atleast_2d(np.loadtxt(os.path.join(data_root, 'metadata/CVPPP/number_of_instances.txt'),
                                                  dtype='str', delimiter=','))
            n_objects_gt = int(counts[counts[:, 0] == stem][0][1])
            n_objects_pred = np.load(base + '-n_objects.npy')    # plain array: allow_pickle stays False
            ins_seg_gt = np.array(Image.open(os.path.join(img_dir, stem + '_label.png')))
            ins_seg_pred = np.array(Image.open(base + '-ins_mask.png'))
            sbds.append(calc_sbd(ins_seg_gt, ins_seg_pred))
            dics.append(calc_dic(n_objects_gt, n_objects_pred))
        fg_seg_gt = np.array(Image.open(os.path.join(img_dir, stem + '_fg.png'))) == 1
        fg_seg_pred = np.array(Image.open(base + '-fg_mask.png')) == 255
        fg_dices.append(calc_dice(fg_seg_gt, fg_seg_pred))
        scored.append(name)
    return sbds, dics, fg_dices, scored


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--pred_dir', required=True, help='Prediction directory')
    parser.add_argument('--dataset', type=str, required=True, help='Name of the dataset which is "CVPPP"')
    parser.add_argument('--data_root', default='../data', help='where metadata/ and raw/ live (reference: ../data)')
    parser.add_argument('--fg_only', action='store_true', help='score foreground masks only (no instance outputs)')
    opt = parser.parse_args()
    assert opt.dataset in ['CVPPP', ]
    sbds, dics, fg_dices, scored = evaluate_cvppp(opt.pred_dir, opt.data_root, opt.fg_only)
    if not opt.fg_only:
        print('MEAN SBD     : ', np.mean(sbds))
        print('MEAN |DIC|   : ', np.mean(dics))
    print('MEAN FG DICE : ', np.mean(fg_dices))


if __name__ == '__main__':
    main()
