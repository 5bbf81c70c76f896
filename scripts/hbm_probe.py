"""Practical HBM ceilings on this box for the access mixes the path uses (torch elementwise kernels as neutral probes):
copy (1 read + 1 write), read-only reduction, 2 reads + 1 write; tensors large enough to defeat the 256 MB Infinity Cache,
and small enough to live in it (134 MB = one 256x256x64 bf16 activation at bs=16)."""
import torch
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
for mb in (134, 1024):
    n = mb * (1 << 20) // 2
    a = torch.randn(n, device="cuda").to(torch.bfloat16); b = torch.empty_like(a); c = torch.randn(n, device="cuda").to(torch.bfloat16)
    dt = t(lambda: b.copy_(a));            print("%5d MB tensors  copy (1R+1W)      %7.1f us  %5.2f TB/s" % (mb, dt * 1e6, 2 * n * 2 / dt / 1e12))
    dt = t(lambda: torch.add(a, c, out=b)); print("%5d MB tensors  add  (2R+1W)      %7.1f us  %5.2f TB/s" % (mb, dt * 1e6, 3 * n * 2 / dt / 1e12))
    dt = t(lambda: a.float().sum()) if False else t(lambda: torch.sum(a, dtype=torch.float32)); print("%5d MB tensors  sum  (1R)         %7.1f us  %5.2f TB/s" % (mb, dt * 1e6, n * 2 / dt / 1e12))
