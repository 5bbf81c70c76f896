"""TEST INFRASTRUCTURE (oracle) — numpy restatement of the tail of the reference's collate function,
`AlignCollate.__call__` (code/lib/dataset.py:349-379), the step that turns the per-image uint8 arrays into the
network's integer targets:

    instance_annotations = np.array(instance_annotations, dtype='int')          # bs, h, w, n_ins      (:349-351)
    semantic_annotations = np.array(semantic_annotations, dtype='int')          # bs, h, w             (:355-356)
    one_hot = np.eye(n_classes, dtype='int')[semantic_annotations.flatten()].reshape(bs, h, w, n_classes)   (:357-361)
    instance_annotations = torch.LongTensor(instance_annotations).permute(0, 3, 1, 2)                       (:363-364)
    one_hot = torch.LongTensor(one_hot).permute(0, 3, 1, 2)                                                 (:366-369)

Pure integer index work: the GPU path (isa_collate_targets) must match bit for bit.  The reference's dataset
module imports lmdb / skimage (absent here) at module level, so this file restates the five lines above instead of
importing them; tests/test_oracle_collate.py pins it against hand-written cases.
"""
import numpy as np


def collate_targets(instance_annotations_u8, semantic_annotations_u8, n_classes=2):
    """uint8 [bs,h,w,K], uint8 [bs,h,w] -> (sem one-hot int64 [bs,n_classes,h,w], ins int64 [bs,K,h,w])."""
    ins = np.asarray(instance_annotations_u8).astype(np.int64)
    sem = np.asarray(semantic_annotations_u8).astype(np.int64)
    eye = np.eye(n_classes, dtype=np.int64)
    one_hot = eye[sem.reshape(-1)].reshape(sem.shape[0], sem.shape[1], sem.shape[2], n_classes)   # IndexError if > 1
    return np.ascontiguousarray(one_hot.transpose(0, 3, 1, 2)), np.ascontiguousarray(ins.transpose(0, 3, 1, 2))
