#!/usr/bin/env python3
"""What the compact-target entrance saves per step: host -> device time of the reference's int64 targets
(sem one-hot [B,2,H,W] + ins [B,32,H,W], pinned) against the uint8 arrays + isa_collate_targets.
python scripts/bench_collate.py [batch] [size]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isa_amd  # noqa: F401
from isa_amd import lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = 32
ins_u8 = (torch.rand(B, S, S, K) < 0.2).to(torch.uint8).pin_memory()
sem_u8 = (torch.rand(B, S, S) < 0.5).to(torch.uint8).pin_memory()
ins64 = ins_u8.permute(0, 3, 1, 2).contiguous().long().pin_memory()
sem64 = torch.stack([1 - sem_u8.long(), sem_u8.long()], 1).contiguous().pin_memory()
d_ins64, d_sem64 = torch.empty_like(ins64, device="cuda"), torch.empty_like(sem64, device="cuda")
d_ins8, d_sem8 = torch.empty_like(ins_u8, device="cuda"), torch.empty_like(sem_u8, device="cuda")
out_i, out_s = torch.empty_like(d_ins64), torch.empty_like(d_sem64)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def ref_path():
    d_ins64.copy_(ins64, non_blocking=True); d_sem64.copy_(sem64, non_blocking=True)


def compact_path():
    d_ins8.copy_(ins_u8, non_blocking=True); d_sem8.copy_(sem_u8, non_blocking=True)
    rc = L.lib().isa_collate_targets(L.ptr(d_ins8), L.ptr(d_sem8), B, S, S, K, L.ptr(out_i), L.ptr(out_s), L.stream_ptr())
    assert rc == 0


def kernel_only():
    L.lib().isa_collate_targets(L.ptr(d_ins8), L.ptr(d_sem8), B, S, S, K, L.ptr(out_i), L.ptr(out_s), L.stream_ptr())


a, b, c = timed(ref_path), timed(compact_path), timed(kernel_only, 50)
assert torch.equal(out_i.cpu(), ins64) and torch.equal(out_s.cpu(), sem64)
mb64 = (ins64.numel() + sem64.numel()) * 8 / 1e6
mb8 = (ins_u8.numel() + sem_u8.numel()) / 1e6
print("targets bs=%d %dx%d K=%d" % (B, S, S, K))
print("  int64 tensors over PCIe (reference collate output): %7.1f MB  %6.2f ms/step" % (mb64, a))
print("  uint8 arrays over PCIe + isa_collate_targets:       %7.1f MB  %6.2f ms/step" % (mb8, b))
print("  isa_collate_targets alone: %.1f us = %.2f TB/s (reads %.0f MB, writes %.0f MB)" % (c * 1e3, (mb8 + mb64) / 1e6 / (c * 1e-3), mb8, mb64))
