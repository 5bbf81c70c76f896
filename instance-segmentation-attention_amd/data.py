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


def d4_swaps(op):
    """True when op code `op` exchanges the image axes (transpose XOR an odd number of quarter turns)."""
    return bool(((op >> 2) ^ (op >> 3)) & 1)


def d4_augment(tensors, ops, device="cuda"):
    """The exact augmentations of the reference's collate function (dataset.py:185-233: horizontal / vertical flip,
    transpose, 90x rotation) on the device, one shared op code per image for all of `tensors` (uint8 [n,h,w,c] or
    [n,h,w]: RGB image, semantic map, instance planes).  ops: per-image codes, bit0 hflip, bit1 vflip, bit2 transpose,
    bits 3-4 = rot_angle // 90; the random draws stay on the host in the reference's call order (three
    random.random() < 0.5, one np.random.choice([0, 90, 180, 270]) per image).  Non-square images (the reference
    augments the original image, before the resize): every op of the call must agree on whether it exchanges the axes
    (d4_swaps), the outputs are [n,w,h,...] then.  Returns new device tensors."""
    from . import lib as L
    ops = [int(o) for o in ops]
    ops_dev = torch.as_tensor(ops, dtype=torch.int32).to(device)
    out = []
    for t in tensors:
        assert t.dtype == torch.uint8 and t.dim() in (3, 4), "uint8 [n,h,w(,c)]"
        src = t.to(device).contiguous()
        n, h, w = src.shape[:3]
        swap = d4_swaps(ops[0])
        assert h == w or all(d4_swaps(o) == swap for o in ops), "ops of one call must agree on exchanging the axes of a non-square image"
        swap = swap and h != w
        shape = (n, w, h) + tuple(src.shape[3:]) if swap else tuple(src.shape)
        dst = torch.empty(shape, dtype=torch.uint8, device=src.device)
        c = 1 if t.dim() == 3 else t.shape[3]
        L.check(L.lib().isa_d4_augment(L.ptr(src), L.ptr(dst), n, h, w, c, 1 if swap else 0, L.ptr(ops_dev),
                                       L.stream_ptr()), "isa_d4_augment")
        out.append(dst)
    return out


# ---- rotation by a small angle and centre cut: the two non-D4 augmentations the reference ships enabled
# (settings/CVPPP/training_settings.py:40,50; AlignCollate.__preprocess, dataset.py:236-269) -----------------------------
def rotate_geometry(w, h, angle):
    """PIL Image.rotate(angle, expand=True)'s own arithmetic (Python floats, Image.py): the affine matrix that maps an
    output pixel to the source and the expanded output size.  Returns (matrix[6], (nw, nh)), or None when Pillow takes a
    transpose fast path (angle % 360 in {0, 90, 180, 270}: see d4_augment)."""
    import math
    angle = angle % 360.0
    if angle in (0, 90, 180, 270):
        return None
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]

    def tf(x, y):
        return m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]

    m[2], m[5] = tf(-cx, -cy)
    m[2] += cx
    m[5] += cy
    xs, ys = zip(*(tf(x, y) for x, y in ((0, 0), (w, 0), (w, h), (0, h))))
    nw = math.ceil(max(xs)) - math.floor(min(xs))
    nh = math.ceil(max(ys)) - math.floor(min(ys))
    m[2], m[5] = tf(-(nw - w) / 2.0, -(nh - h) / 2.0)
    return m, (nw, nh)


def _fixed_coeffs(m):
    """Geometry.c affine_fixed: FIX(v) = floor(v * 65536 + 0.5), the half-pixel centre folded into the offsets."""
    import math
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def rotate_nearest(tensors, angle, device="cuda"):
    """The reference's annotation rotator (dataset.py:142,243-249: Image.rotate(angle, NEAREST, expand=True) per instance
    plane and for the semantic map) on the device for uint8 [n,h,w,c] tensors sharing one angle; bit-identical to Pillow."""
    import ctypes as C
    from . import lib as L
    out = []
    for t in tensors:
        src = t.to(device).contiguous()
        assert src.dtype == torch.uint8 and src.dim() == 4
        n, h, w, c = src.shape
        g = rotate_geometry(w, h, angle)
        if g is None:                       # Pillow's transpose fast paths: a quarter-turn op of d4_augment (0 = copy)
            q = int(angle % 360) // 90
            out.append(src.clone() if q == 0 else d4_augment([src], [q << 3] * n, device)[0])
            continue
        m, (nw, nh) = g
        coef = (C.c_int32 * 6)(*_fixed_coeffs(m))
        dst = torch.empty((n, nh, nw, c), dtype=torch.uint8, device=src.device)
        L.check(L.lib().isa_rotate_nearest_u8(L.ptr(src), n, h, w, c, L.ptr(dst), nh, nw, coef, L.stream_ptr()),
                "isa_rotate_nearest_u8")
        out.append(dst)
    return out


def background_colour(rgb_host, key):
    """preprocess.py:349-362: the four backgrounds of rotate_with_random_bg - 0 white, 1 black, 2 int(mean), 3 int(median)
    per channel of the source image (host numpy array [h,w,3]; the reference computes them on the host as well)."""
    import numpy as np
    if key == 0:
        return (255, 255, 255)
    if key == 1:
        return (0, 0, 0)
    if key == 2:
        return tuple(int(v) for v in rgb_host.mean((0, 1)))
    return tuple(int(v) for v in np.median(rgb_host, (0, 1)))


def rotate_image(rgb, angle, bg, device="cuda"):
    """The reference's image rotator (dataset.py:140,241 -> preprocess.py:330-365 rotate_with_random_bg: RGBA, BILINEAR,
    expand, composite over the drawn background `bg`) on the device: uint8 [n,h,w,3] -> [n,nh,nw,3], bit-identical to
    Pillow."""
    import ctypes as C
    from . import lib as L
    src = rgb.to(device).contiguous()
    assert src.dtype == torch.uint8 and src.dim() == 4 and src.shape[3] <= 4
    n, h, w, c = src.shape
    g = rotate_geometry(w, h, angle)
    if g is None:
        q = int(angle % 360) // 90
        return src.clone() if q == 0 else d4_augment([src], [q << 3] * n, device)[0]
    m, (nw, nh) = g
    mat = (C.c_double * 6)(*m)
    bgc = (C.c_uint8 * 4)(*([int(v) for v in bg] + [255] * (4 - len(bg))))
    dst = torch.empty((n, nh, nw, c), dtype=torch.uint8, device=src.device)
    L.check(L.lib().isa_rotate_bilinear_u8(L.ptr(src), n, h, w, c, L.ptr(dst), nh, nw, mat, bgc, L.stream_ptr()),
            "isa_rotate_bilinear_u8")
    return dst


def center_cut_window(H, W, center, h, w):
    """preprocess.py:239-264 CenterCut: the window (y0, x0, height, width) it takes from an H x W array around `center`
    for an h x w output (it doubles both and clamps to the image)."""
    h, w = 2 * h, 2 * w
    if center[0] - h // 2 < 0:
        y0 = 0
    elif center[0] + h // 2 > H:
        y0 = max(0, H - h)
    else:
        y0 = center[0] - h // 2
    if center[1] - w // 2 < 0:
        x0 = 0
    elif center[1] + w // 2 > W:
        x0 = max(0, W - w)
    else:
        x0 = center[1] - w // 2
    return int(y0), int(x0), min(H, y0 + min(H, h)) - int(y0), min(W, x0 + min(W, w)) - int(x0)


def center_cut(rgb, sem, planes, n_objects, draw, out_h, out_w, max_planes=32, device="cuda"):
    """dataset.py:252-269 on the device for ONE image: rgb uint8 [1,H,W,3], sem uint8 [1,H,W,1], planes uint8 [1,H,W,K]
    (the first n_objects are real).  `draw(count)` returns the reference's np.random.choice(count) - the index of the
    chosen centre in the row-major list of pixels covered by exactly one plane.  Returns (rgb, sem, planes [1,h',w',
    max_planes] with the surviving planes first, n_objects') - the has-object filter (window sum > 30) drops planes."""
    from . import lib as L
    lib, st = L.lib(), L.stream_ptr()
    _, H, W, K = planes.shape
    single = torch.empty((1, H, W), dtype=torch.uint8, device=device)
    rows = torch.empty((1, H), dtype=torch.int32, device=device)
    real = planes if n_objects == K else planes[..., :n_objects].contiguous()
    L.check(lib.isa_cover_rows_u8(L.ptr(real), 1, H, W, n_objects, L.ptr(single), L.ptr(rows), st), "isa_cover_rows_u8")
    counts = rows[0].cpu().numpy().astype("int64")
    total = int(counts.sum())
    assert total > 0, "no pixel is covered by exactly one instance (the reference raises here too)"
    pick = int(draw(total))
    cum = counts.cumsum()
    r = int((cum > pick).argmax())
    j = pick - int(cum[r] - counts[r])
    cols = single[0, r].cpu().numpy().nonzero()[0]
    center = (r, int(cols[j]))
    y0, x0, hh, ww = center_cut_window(H, W, center, out_h, out_w)
    sums = torch.zeros((1, n_objects), dtype=torch.int64, device=device)
    L.check(lib.isa_plane_sums_u8(L.ptr(real), 1, H, W, n_objects, y0, x0, hh, ww, L.ptr(sums), st), "isa_plane_sums_u8")
    keep = [i for i, v in enumerate(sums[0].cpu().tolist()) if v > 30]
    chan = torch.tensor(keep + [-1] * (max_planes - len(keep)), dtype=torch.int32, device=device)
    p2 = torch.empty((1, hh, ww, max_planes), dtype=torch.uint8, device=device)
    L.check(lib.isa_crop_planes_u8(L.ptr(real), 1, H, W, n_objects, y0, x0, L.ptr(p2), hh, ww, max_planes, L.ptr(chan), st),
            "isa_crop_planes_u8")
    r2 = torch.empty((1, hh, ww, rgb.shape[3]), dtype=torch.uint8, device=device)
    L.check(lib.isa_crop_planes_u8(L.ptr(rgb), 1, H, W, rgb.shape[3], y0, x0, L.ptr(r2), hh, ww, rgb.shape[3], None, st),
            "isa_crop_planes_u8")
    s2 = torch.empty((1, hh, ww, 1), dtype=torch.uint8, device=device)
    L.check(lib.isa_crop_planes_u8(L.ptr(sem), 1, H, W, 1, y0, x0, L.ptr(s2), hh, ww, 1, None, st), "isa_crop_planes_u8")
    return r2, s2, p2, len(keep)


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
    D4 op codes or None (on non-square sources the ops of one call must agree on exchanging the axes).  Returns (sem_onehot int64 [n,2,size,size],
    ins int64 [n,K,size,size]) on the device; three launches instead of ~70 PIL calls per image."""
    from . import lib as L
    planes, sem = planes.to(device).contiguous(), sem.to(device).contiguous()
    n, h0, w0, k = planes.shape
    sem4 = sem.reshape(n, h0, w0, 1)
    if ops is not None:
        planes, sem4 = d4_augment([planes, sem4], ops, device)
        n, h0, w0, k = planes.shape
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
