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
and `ReSeg.forward` accept directly (isa_image_ex + isa_collate_targets expand them).  In 'training' mode the loader
applies the augmentations the reference ships ENABLED (settings/CVPPP/training_settings.py:36-50), on the device and in
the reference's order (dataset.py:185-269): horizontal / vertical flip, transpose, 90x rotation (`data.d4_augment`, on
the original non-square image), rotation by an integer angle in [-9, 9] with expand (annotations nearest, image
bilinear over a drawn background: `data.rotate_nearest`, `data.rotate_image`) and the centre cut with its has-object
filter (`data.center_cut`).  The ones it ships disabled (colour jitter, gamma, channel swap, grayscale, resolution:
dataset.py:271-285) are not part of this build."""
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
    (rgb uint8 [B,H,W,3], sem uint8 [B,H,W], ins uint8 [B,H,W,32], n_objects int32 [B]) with every image operation on
    the device.  mode 'training' shuffles (seeded) and augments as the reference's shipped settings do, drawing from
    `random` / `numpy.random`-style generators in the reference's call order (dataset.py:186,198,210,222,237-239,
    preprocess.py:349, dataset.py:256); 'test' keeps order and applies none.  Images of one batch may differ in size.
    Like the reference's DataLoader + AlignCollate no sample is dropped: a short last batch is filled by repeating its
    first image (dataset.py:330-333).  With world > 1 the permutation is padded (wrapping around) to a multiple of
    `world` before it is dealt out, so every rank runs the same number of steps - a rank with a step more than the
    others would wait in its gradient all-reduce forever."""

    def __init__(self, dataset, batch_size, height=256, width=256, max_n_objects=32, mode='test', seed=0, device='cuda',
                 rank=0, world=1, d4=True, rotation=True, center_cut=True):
        assert mode in ('training', 'test')
        self.ds, self.bs, self.h, self.w, self.k, self.mode = dataset, batch_size, height, width, max_n_objects, mode
        self.seed, self.epoch, self.device, self.rank, self.world = seed, 0, device, rank, world
        self.d4, self.rotation, self.center_cut = d4, rotation, center_cut
        self.last_draws = []             # per image of the last batch: dict(op, angle, bg_key, pick) - tests / logging

    def indices(self):
        idx = list(range(len(self.ds)))
        if self.mode == 'training':
            random.Random(self.seed + self.epoch).shuffle(idx)         # same permutation on every rank
        if self.world > 1 and idx:
            per = -(-len(idx) // self.world)
            idx = (idx * (per * self.world // len(idx) + 1))[:per * self.world]   # wrap around: equal shards
        return idx[self.rank::self.world]                              # rank shard (parallel.py: data sharding)

    def __len__(self):
        return -(-len(self.indices()) // self.bs)

    def _one(self, i, py_rng, np_rng):
        """One sample through AlignCollate.__preprocess (dataset.py:175-330) on the device."""
        from . import data as D
        from . import lib as L
        dev = self.device
        img, sem, ins, n_obj = self.ds[i]
        rgb_host = np.array(img.convert('RGB'))
        rgb = torch.from_numpy(rgb_host[None]).to(dev)
        h0, w0 = sem.shape
        n_obj = min(int(n_obj), self.k)
        planes = torch.from_numpy(np.array(ins[None, :, :, :n_obj])).to(dev)          # a writable copy of the record's bytes
        semt = torch.from_numpy(np.array(sem)[None, :, :, None]).to(dev)
        draws = dict(op=0, angle=0, bg_key=None, pick=None)
        if self.mode == 'training':
            if self.d4:
                # the reference's call order: hflip, vflip, transpose (random.random() < 0.5 each), then 90x rotation
                op = int(py_rng.random() < 0.5) | (int(py_rng.random() < 0.5) << 1) | (int(py_rng.random() < 0.5) << 2) | \
                    ((int(np_rng.choice([0, 90, 180, 270])) // 90) << 3)
                draws["op"] = op
                if op:
                    rgb, planes, semt = D.d4_augment([rgb, planes, semt], [op], dev)
                    rgb_host = None      # the background mean is taken over the augmented array (numpy's summation order)
            if self.rotation:
                angle = int(np_rng.rand() * 10)                          # dataset.py:237-239
                if np_rng.rand() >= 0.5:
                    angle = -1 * angle
                key = int(np_rng.choice([0, 1, 2, 3]))                   # preprocess.py:349, drawn for every angle
                draws.update(angle=angle, bg_key=key)
                if angle % 360 != 0:
                    src = rgb_host if rgb_host is not None else rgb[0].cpu().numpy()
                    rgb = D.rotate_image(rgb, angle, D.background_colour(src, key), dev)
                    planes, semt = D.rotate_nearest([planes, semt], angle, dev)
            if self.center_cut:
                def draw(count):
                    draws["pick"] = int(np_rng.choice(count, 1)[0])
                    return draws["pick"]
                rgb, semt, planes, n_obj = D.center_cut(rgb, semt, planes, n_obj, draw, self.h, self.w, self.k, dev)
        self.last_draws.append(draws)
        h1, w1 = planes.shape[1:3]
        if planes.shape[3] != self.k:                                   # zero planes up to max_n_objects (dataset.py:305-311)
            full = torch.zeros((1, h1, w1, self.k), dtype=torch.uint8, device=dev)
            full[..., :planes.shape[3]] = planes
            planes = full
        rgb2 = D.resize_bilinear(rgb, (self.h, self.w), dev)
        p2 = torch.empty((1, self.h, self.w, self.k), dtype=torch.uint8, device=dev)
        s2 = torch.empty((1, self.h, self.w, 1), dtype=torch.uint8, device=dev)
        L.check(L.lib().isa_resize_nearest_u8(L.ptr(planes), 1, h1, w1, self.k, L.ptr(p2), self.h, self.w, L.stream_ptr()),
                "isa_resize_nearest_u8")
        L.check(L.lib().isa_resize_nearest_u8(L.ptr(semt.contiguous()), 1, h1, w1, 1, L.ptr(s2), self.h, self.w, L.stream_ptr()),
                "isa_resize_nearest_u8")
        return rgb2, s2[..., 0], p2, n_obj

    def __iter__(self):
        self.epoch += 1
        idx = self.indices()
        py_rng = random.Random(1000003 * self.seed + self.epoch + 7919 * self.rank)
        np_rng = np.random.RandomState((1000003 * self.seed + self.epoch + 7919 * self.rank) % (2 ** 32))
        for s in range(0, len(idx), self.bs):
            chunk = idx[s:s + self.bs]
            chunk = chunk + [chunk[0]] * (self.bs - len(chunk))         # dataset.py:330-333: repeat the batch's first image
            self.last_draws = []
            rgbs, sems, inss, ns = zip(*(self._one(i, py_rng, np_rng) for i in chunk))
            yield torch.cat(rgbs), torch.cat(sems), torch.cat(inss), torch.tensor(ns, dtype=torch.int32)
