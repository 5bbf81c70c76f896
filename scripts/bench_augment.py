#!/usr/bin/env python3
"""isa_d4_augment on the collated batch (16 x 256 x 256: RGB, semantic map, 32 instance planes) vs the same
permutations with numpy on the host (what the reference does plane by plane through PIL).  python scripts/bench_augment.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import isa_amd  # noqa: F401
from isa_amd import lib as L
import augment_ref as R

B, S = 16, 256
rng = np.random.default_rng(0)
ops = [int(v) for v in rng.integers(0, 32, B)]
o = torch.tensor(ops, dtype=torch.int32, device="cuda")
for name, c in (("instance planes", 32), ("RGB image", 3), ("semantic map", 1)):
    x = rng.integers(0, 2, (B, S, S, c), dtype=np.uint8)
    d = torch.from_numpy(x).cuda(); out = torch.empty_like(d)
    f = lambda: L.lib().isa_d4_augment(L.ptr(d), L.ptr(out), B, S, S, c, 0, L.ptr(o), L.stream_ptr())
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    t1 = time.perf_counter(); ref = R.d4_batch(x, ops); th = time.perf_counter() - t1
    assert np.array_equal(out.cpu().numpy(), ref)
    mb = 2 * x.size / 1e6
    print("%-16s [%d,%d,%d,%2d] uint8: device %6.1f us (%.2f TB/s, %5.1f MB moved)   numpy on one host core %6.2f ms" %
          (name, B, S, S, c, dt * 1e6, mb / 1e6 / dt, mb, th * 1e3))

# annotation resize (ann_resizer): CVPPP A1 planes 530x500 -> 256x256
import resize_ref as RR
x = (rng.random((B, 530, 500, 32)) < 0.3).astype(np.uint8)
d = torch.from_numpy(x).cuda(); out = torch.empty((B, S, S, 32), dtype=torch.uint8, device="cuda")
f = lambda: L.lib().isa_resize_nearest_u8(L.ptr(d), B, 530, 500, 32, L.ptr(out), S, S, L.stream_ptr())
for _ in range(3): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
t1 = time.perf_counter(); ref = RR.resize_nearest(x, S, S); th = time.perf_counter() - t1
assert np.array_equal(out.cpu().numpy(), ref)
print("nearest resize   [%d,530,500,32] -> [%d,%d,%d,32] uint8: device %6.1f us (%.1f MB written)   numpy on one host core %6.2f ms" %
      (B, B, S, S, dt * 1e6, out.numel() / 1e6, th * 1e3))
