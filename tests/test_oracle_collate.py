"""oracle/collate_ref.py (tail of AlignCollate.__call__, code/lib/dataset.py:349-379) against hand-written cases."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import collate_ref as R  # noqa: E402


def test_hand_case():
    ins = np.zeros((1, 2, 3, 4), np.uint8)            # bs, h, w, K
    ins[0, 0, 1, 0] = 1
    ins[0, 1, 2, 3] = 255                             # values are widened, not binarised (dtype='int' cast)
    sem = np.array([[[0, 1, 0], [1, 1, 0]]], np.uint8)
    one_hot, out = R.collate_targets(ins, sem)
    assert out.shape == (1, 4, 2, 3) and out.dtype == np.int64
    assert out[0, 0, 0, 1] == 1 and out[0, 3, 1, 2] == 255 and out.sum() == 256
    assert one_hot.shape == (1, 2, 2, 3) and one_hot.dtype == np.int64
    np.testing.assert_array_equal(one_hot[0, 1], sem[0])
    np.testing.assert_array_equal(one_hot[0, 0], 1 - sem[0].astype(np.int64))
    assert (one_hot.sum(1) == 1).all()


def test_out_of_range_class_raises_like_np_eye_indexing():
    with pytest.raises(IndexError):
        R.collate_targets(np.zeros((1, 2, 2, 1), np.uint8), np.full((1, 2, 2), 2, np.uint8))
