"""TEST INFRASTRUCTURE (oracle) — numpy restatement of the exact (index-permuting) augmentations of the reference's
collate function, `AlignCollate.__preprocess` (code/lib/dataset.py:185-233), applied to image, semantic map and every
instance plane with one shared random decision each:

    horizontal flip   preprocess.py:171   img.transpose(Image.FLIP_LEFT_RIGHT)        a[:, ::-1]
    vertical flip     preprocess.py:218   img.transpose(Image.FLIP_TOP_BOTTOM)        a[::-1]
    transpose         preprocess.py:286   img.transpose(Image.TRANSPOSE)              a.swapaxes(0, 1)
    90x rotation      preprocess.py:323   img.rotate(angle, resample, expand=True)    np.rot90(a, angle // 90)
                      (angle in {0, 90, 180, 270}; counter-clockwise; the resample filter does not matter there)

in that order (dataset.py:185, 197, 209, 221).  Pinned against Pillow itself - the reference's own dependency - in
tests/test_oracle_augment.py.  Op code of the GPU path: bit0 hflip, bit1 vflip, bit2 transpose, bits 3-4 angle//90.
"""
import numpy as np


def d4(a, op):
    """a: [s, s] or [s, s, c] array; returns the augmented array (a view chain made contiguous)."""
    if op & 1:
        a = a[:, ::-1]
    if op & 2:
        a = a[::-1]
    if op & 4:
        a = a.swapaxes(0, 1)
    a = np.rot90(a, (op >> 3) & 3)
    return np.ascontiguousarray(a)


def d4_batch(x, ops):
    """x: [n, s, s, c]; ops: n op codes."""
    return np.stack([d4(x[b], int(ops[b])) for b in range(x.shape[0])])


def draw_ops(n, py_random, np_random):
    """The reference's host-side decisions in its call order (dataset.py:186,198,210,222): three random.random() < 0.5
    and one np.random.choice([0, 90, 180, 270]) per image."""
    ops = []
    for _ in range(n):
        op = int(py_random.random() < 0.5)
        op |= int(py_random.random() < 0.5) << 1
        op |= int(py_random.random() < 0.5) << 2
        op |= (int(np_random.choice([0, 90, 180, 270])) // 90) << 3
        ops.append(op)
    return ops
