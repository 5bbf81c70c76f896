"""Synthetic stand-in for the LMDB dataset + AlignCollate output (dataset.py:368-378): the reference
ships no data.  Same tensors and dtypes the collate function produces; rectangles/ellipses as instances
(mirrors the zero-padded 32-instance layout, dataset.py:304-310).  numpy RNG only."""
import numpy as np
import torch


def synth_batch(batch, height, width, seed=0, max_objects=32, kmin=3, kmax=8):
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((batch, 21, height, width)).astype(np.float32)
    ins = np.zeros((batch, max_objects, height, width), dtype=np.int64)
    n = np.zeros((batch,), dtype=np.int32)
    yy, xx = np.mgrid[0:height, 0:width]
    for b in range(batch):
        k = int(rs.randint(kmin, kmax + 1))
        occupied = np.zeros((height, width), dtype=bool)
        placed = tries = 0
        while placed < k and tries < 200:
            tries += 1
            hh = int(rs.randint(max(2, height // 10), max(3, height // 3)))
            ww = int(rs.randint(max(2, width // 10), max(3, width // 3)))
            y0, x0 = int(rs.randint(0, height - hh + 1)), int(rs.randint(0, width - ww + 1))
            if rs.rand() < 0.5:
                m = (yy >= y0) & (yy < y0 + hh) & (xx >= x0) & (xx < x0 + ww)
            else:
                cy, cx = y0 + hh / 2.0, x0 + ww / 2.0
                m = ((yy + 0.5 - cy) / (hh / 2.0)) ** 2 + ((xx + 0.5 - cx) / (ww / 2.0)) ** 2 <= 1.0
            if m.sum() < 4 or (m & occupied).any():
                continue
            ins[b, placed][m] = 1
            occupied |= m
            placed += 1
        n[b] = placed
        x[b, :3] += occupied[None].astype(np.float32)
    fg = ins.sum(1) > 0
    sem = np.stack([~fg, fg], 1).astype(np.int64)
    return torch.from_numpy(x), torch.from_numpy(sem), torch.from_numpy(ins), torch.from_numpy(n)


class SyntheticLoader(object):
    """Iterable of `n_batches` collated minibatches, re-seeded per epoch like a shuffling DataLoader."""

    def __init__(self, n_batches, batch_size, height=256, width=256, seed=0, compact=False):
        """compact=True yields the targets as the reference's collate function holds them before its last five lines
        (dataset.py:349-379): sem uint8 [B,H,W], ins uint8 [B,H,W,32]; the model expands them on the device
        (isa_collate_targets) - 8x less host-to-device traffic per step."""
        self.n, self.bs, self.h, self.w, self.seed, self.epoch = n_batches, batch_size, height, width, seed, 0
        self.compact = compact

    def __len__(self):
        return self.n

    def __iter__(self):
        self.epoch += 1
        for i in range(self.n):
            x, sem, ins, n = synth_batch(self.bs, self.h, self.w, seed=self.seed + 1000 * self.epoch + i)
            if self.compact:
                sem, ins = sem[:, 1].contiguous().to(torch.uint8), ins.permute(0, 2, 3, 1).contiguous().to(torch.uint8)
            yield x, sem, ins, n


def d4_augment(tensors, ops, device="cuda"):
    """The exact augmentations of the reference's collate function (dataset.py:185-233: horizontal / vertical flip,
    transpose, 90x rotation) on the device, one shared op code per image for all of `tensors` (uint8 [n,s,s,c] or
    [n,s,s]: RGB image, semantic map, instance planes).  ops: per-image codes, bit0 hflip, bit1 vflip, bit2 transpose,
    bits 3-4 = rot_angle // 90; the random draws stay on the host in the reference's call order (three
    random.random() < 0.5, one np.random.choice([0, 90, 180, 270]) per image).  Returns new device tensors."""
    from . import lib as L
    ops_dev = torch.as_tensor(list(ops), dtype=torch.int32).to(device)
    out = []
    for t in tensors:
        assert t.dtype == torch.uint8 and t.dim() in (3, 4) and t.shape[1] == t.shape[2], "uint8 [n,s,s(,c)]"
        src = t.to(device).contiguous()
        dst = torch.empty_like(src)
        c = 1 if t.dim() == 3 else t.shape[3]
        L.check(L.lib().isa_d4_augment(L.ptr(src), L.ptr(dst), src.shape[0], src.shape[1], c, L.ptr(ops_dev),
                                       L.stream_ptr()), "isa_d4_augment")
        out.append(dst)
    return out


_RESIZE_WS = {}


def resize_bilinear(images, size, device="cuda"):
    """The reference's `img_resizer` (dataset.py:160-161, prediction.py:37: PIL Image.resize(BILINEAR)) on the device:
    uint8 [n,h0,w0,c] (c <= 4) -> uint8 [n,size_h,size_w,c], bit-identical to Pillow (isa_resize_bilinear_u8)."""
    from . import lib as L
    h, w = (size, size) if isinstance(size, int) else size
    src = images.to(device).contiguous()
    assert src.dtype == torch.uint8 and src.dim() == 4
    n, h0, w0, c = src.shape
    dst = torch.empty((n, h, w, c), dtype=torch.uint8, device=src.device)
    need = int(L.lib().isa_resize_bilinear_ws_bytes(n, h0, w0, c, h, w))
    ws = _RESIZE_WS.get(src.device)
    if ws is None or ws.numel() < need:
        ws = _RESIZE_WS[src.device] = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=src.device)
    L.check(L.lib().isa_resize_bilinear_u8(L.ptr(src), n, h0, w0, c, L.ptr(dst), h, w, L.ptr(ws), ws.numel(), L.stream_ptr()),
            "isa_resize_bilinear_u8")
    return dst


class DevicePrefetcher(object):
    """Wraps an iterable of collated host batches (x, sem, ins, n): batch i+1 travels to the device on its own HIP
    stream while step i computes, so the host-to-device time of the reference's hand-over (373 MB per step at bs=16:
    fp32 input, int64 targets) leaves the critical path.  Host tensors are pinned once per batch; `n` stays on the
    host (the model reads it there).  Yields device tensors that the consumer's stream may use immediately."""

    def __init__(self, loader, device="cuda"):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch):
        x, sem, ins, n = batch
        with torch.cuda.stream(self.stream):
            dev = [t.pin_memory().to(self.device, non_blocking=True) if not t.is_cuda else t for t in (x, sem, ins)]
        return dev[0], dev[1], dev[2], n

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            torch.cuda.current_stream(self.device).wait_stream(self.stream)     # batch i has landed
            for t in cur[:3]:
                t.record_stream(torch.cuda.current_stream(self.device))
            try:
                nxt = self._upload(next(it))                                     # batch i+1 starts moving now
            except StopIteration:
                nxt = None
            yield cur


def device_collate_targets(planes, sem, ops=None, size=256, device="cuda"):
    """The annotation side of the reference's collate function on the device, in its order (dataset.py:185-233 exact
    augmentations at the source resolution, :293-320 nearest resize, :349-379 int64 planes + one-hot):
    planes uint8 [n,h0,w0,K] (zero planes already appended up to K = 32, :305-311), sem uint8 [n,h0,w0], ops = per-image
    D4 op codes or None.  Square sources are required when an op transposes.  Returns (sem_onehot int64 [n,2,size,size],
    ins int64 [n,K,size,size]) on the device; three launches instead of ~70 PIL calls per image."""
    from . import lib as L
    planes, sem = planes.to(device).contiguous(), sem.to(device).contiguous()
    n, h0, w0, k = planes.shape
    sem4 = sem.reshape(n, h0, w0, 1)
    if ops is not None:
        planes, sem4 = d4_augment([planes, sem4], ops, device)
    st = L.stream_ptr()
    p2 = torch.empty((n, size, size, k), dtype=torch.uint8, device=device)
    s2 = torch.empty((n, size, size, 1), dtype=torch.uint8, device=device)
    L.check(L.lib().isa_resize_nearest_u8(L.ptr(planes), n, h0, w0, k, L.ptr(p2), size, size, st), "isa_resize_nearest_u8")
    L.check(L.lib().isa_resize_nearest_u8(L.ptr(sem4), n, h0, w0, 1, L.ptr(s2), size, size, st), "isa_resize_nearest_u8")
    ins_out = torch.empty((n, k, size, size), dtype=torch.int64, device=device)
    sem_out = torch.empty((n, 2, size, size), dtype=torch.int64, device=device)
    L.check(L.lib().isa_collate_targets(L.ptr(p2), L.ptr(s2), n, size, size, k, L.ptr(ins_out), L.ptr(sem_out), st),
            "isa_collate_targets")
    return sem_out, ins_out
