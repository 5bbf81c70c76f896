"""The reference's dataset records (code/lib/dataset.py:17-71 `SegDataset`, written by data/scripts/CVPPP/utils.py:13-61
`create_dataset`) over a plain directory instead of LMDB (`lmdb` is not part of this image; the key schema is what
matters to a drop-in).  Keys, exactly as the reference writes them - all values are bytes:

    num-samples                      str(n)
    image-{i}                        the encoded image file (PNG/JPEG bytes), i = 1..n
    semantic-annotation-{i}          uint8 [height, width].tobytes()
    instance-annotation-{i}          uint8 [height, width, n_objects].tobytes()
    height-{i} / width-{i} / n_objects-{i}    str(int)

A `DirStore` keeps one file per key (`<root>/<key>`); `RecordDataset` reads a sample the way `SegDataset.__load_data`
does and returns `(PIL image, semantic uint8 [h,w], instance uint8 [h,w,n], n_objects)`.  `RecordLoader` batches
samples into the compact hand-over of this build: uint8 RGB resized on the device (isa_resize_bilinear_u8, Pillow-exact)
and uint8 targets resized / padded to 32 planes on the device (isa_resize_nearest_u8) - what `Trainer.train_step`
and `ReSeg.forward` accept directly (isa_image_ex + isa_collate_targets expand them).  The reference's random
augmentations that are exact index permutations are available through `data.d4_augment`; the remaining ones
(+-10 degree rotation, centre cut, colour jitter, gamma, channel swap, grayscale, resolution: dataset.py:236-285)
are not part of this build."""
import io
import os
import random

import numpy as np
import torch


class DirStore(object):
    """bytes key -> bytes value, one file per key."""

    def __init__(self, root, create=False):
        self.root = root
        if create:
            os.makedirs(root, exist_ok=True)
        assert os.path.isdir(root), 'Cannot read records from {}'.format(root)

    def get(self, key):
        key = key.decode() if isinstance(key, bytes) else key
        path = os.path.join(self.root, key)
        if not os.path.isfile(path):
            return None
        with open(path, 'rb') as f:
            return f.read()

    def put(self, key, value):
        key = key.decode() if isinstance(key, bytes) else key
        value = value if isinstance(value, bytes) else str(value).encode()
        with open(os.path.join(self.root, key), 'wb') as f:
            f.write(value)


def create_dataset(output_path, images, semantic_annotations, instance_annotations):
    """data/scripts/CVPPP/utils.py:13-61 with in-memory inputs: images = encoded image bytes (or uint8 RGB arrays, PNG-
    encoded here), semantic = uint8 [h,w], instance = uint8 [h,w,n_objects]."""
    from PIL import Image
    n_images = len(images)
    assert n_images == len(semantic_annotations) == len(instance_annotations)
    st = DirStore(output_path, create=True)
    for i in range(n_images):
        img = images[i]
        if not isinstance(img, (bytes, bytearray)):
            buf = io.BytesIO()
            Image.fromarray(np.asarray(img, np.uint8)).save(buf, format='PNG')
            img = buf.getvalue()
        sem = np.ascontiguousarray(semantic_annotations[i], np.uint8)
        ins = np.ascontiguousarray(instance_annotations[i], np.uint8)
        k = i + 1
        st.put('image-{}'.format(k), img)
        st.put('semantic-annotation-{}'.format(k), sem.tobytes())
        st.put('instance-annotation-{}'.format(k), ins.tobytes())
        st.put('height-{}'.format(k), str(sem.shape[0]))
        st.put('width-{}'.format(k), str(sem.shape[1]))
        st.put('n_objects-{}'.format(k), str(ins.shape[2]))
    st.put('num-samples', str(n_images))
    return st


class RecordDataset(object):
    """SegDataset (dataset.py:17-71)."""

    def __init__(self, path):
        self.store = DirStore(path)
        n = self.store.get('num-samples')
        assert n is not None, 'Cannot read records from {}'.format(path)
        self.n_samples = int(n)

    def __len__(self):
        return self.n_samples

    def __getitem__(self, index):
        assert index <= len(self), 'index range error'                # dataset.py:63 (sic: <=)
        from PIL import Image
        g, k = self.store.get, index + 1
        img = Image.open(io.BytesIO(g('image-{}'.format(k))))
        height, width = int(g('height-{}'.format(k))), int(g('width-{}'.format(k)))
        n_objects = int(g('n_objects-{}'.format(k)))
        sem = np.frombuffer(g('semantic-annotation-{}'.format(k)), dtype=np.uint8).reshape(height, width)
        ins = np.frombuffer(g('instance-annotation-{}'.format(k)), dtype=np.uint8).reshape(height, width, n_objects)
        return img, sem, ins, n_objects


class RecordLoader(object):
    """Batches of a RecordDataset in the compact hand-over (see module docstring): yields
    (rgb uint8 [B,H,W,3], sem uint8 [B,H,W], ins uint8 [B,H,W,32], n_objects int32 [B]) with the resizes done on the
    device.  mode 'training' shuffles (seeded) and draws the exact D4 augmentations of dataset.py:185-233 per image;
    'test' keeps order and applies none.  Images of one batch may differ in size (one resize launch per image)."""

    def __init__(self, dataset, batch_size, height=256, width=256, max_n_objects=32, mode='test', seed=0, device='cuda',
                 rank=0, world=1):
        assert mode in ('training', 'test')
        self.ds, self.bs, self.h, self.w, self.k, self.mode = dataset, batch_size, height, width, max_n_objects, mode
        self.seed, self.epoch, self.device, self.rank, self.world = seed, 0, device, rank, world

    def indices(self):
        idx = list(range(len(self.ds)))
        if self.mode == 'training':
            random.Random(self.seed + self.epoch).shuffle(idx)         # same permutation on every rank
        return idx[self.rank::self.world]                              # rank shard (parallel.py: data sharding)

    def __len__(self):
        return len(self.indices()) // self.bs

    def __iter__(self):
        from .data import d4_augment, resize_bilinear
        from . import lib as L
        self.epoch += 1
        idx = self.indices()
        rng = random.Random(1000003 * self.seed + self.epoch + 7919 * self.rank)
        for s in range(0, len(idx) - self.bs + 1, self.bs):
            rgbs, sems, inss, ns = [], [], [], []
            for i in idx[s:s + self.bs]:
                img, sem, ins, n_obj = self.ds[i]
                rgb = torch.from_numpy(np.array(img.convert('RGB'))[None]).to(self.device)
                h0, w0 = sem.shape
                planes = np.zeros((1, h0, w0, self.k), np.uint8)       # zero planes up to max_n_objects (dataset.py:305-311)
                planes[0, :, :, :n_obj] = ins[:, :, :self.k]
                planes = torch.from_numpy(planes).to(self.device)
                semt = torch.from_numpy(np.array(sem)[None, :, :, None]).to(self.device)
                if self.mode == 'training' and h0 == w0:
                    # the reference's call order: hflip, vflip, transpose (random.random() < 0.5 each), then 90x rotation
                    op = int(rng.random() < 0.5) | (int(rng.random() < 0.5) << 1) | (int(rng.random() < 0.5) << 2) | \
                        (rng.choice([0, 1, 2, 3]) << 3)
                    rgb, planes, semt = d4_augment([rgb, planes, semt], [op], self.device)
                rgbs.append(resize_bilinear(rgb, (self.h, self.w), self.device))
                p2 = torch.empty((1, self.h, self.w, self.k), dtype=torch.uint8, device=self.device)
                s2 = torch.empty((1, self.h, self.w, 1), dtype=torch.uint8, device=self.device)
                L.check(L.lib().isa_resize_nearest_u8(L.ptr(planes), 1, h0, w0, self.k, L.ptr(p2), self.h, self.w, L.stream_ptr()),
                        "isa_resize_nearest_u8")
                L.check(L.lib().isa_resize_nearest_u8(L.ptr(semt), 1, h0, w0, 1, L.ptr(s2), self.h, self.w, L.stream_ptr()),
                        "isa_resize_nearest_u8")
                inss.append(p2); sems.append(s2[..., 0]); ns.append(min(n_obj, self.k))
            yield torch.cat(rgbs), torch.cat(sems), torch.cat(inss), torch.tensor(ns, dtype=torch.int32)
