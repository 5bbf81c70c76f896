"""Tile-size sweep of the streaming attention kernel (isa_sdp_attention, d_k = d_v = 12, n_head = 2 interleaved):
one query per image against L = H*W keys at 256x256 (L = 65 536, batch 16) and 1024x1024 (L = 1 048 576, batch 1 and 4),
storage f32 / bf16 / f16, LDS tiles of 256 / 512 / 1024 keys.  GB/s = K and V read once / time, against 8 TB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import isa_amd  # noqa
from isa_amd import attention_ops as A


def timeit(fn, reps=20):
    """Device time per call: the launches of `reps` calls are captured once in a hipGraph and replayed (the Python
    wrapper costs more host time per call than the kernels run for at the small shapes)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


print("%-10s %8s %6s %5s %9s %9s %7s" % ("dtype", "L", "batch", "tile", "us", "GB/s", "of 8TB/s"))
for dtype in (torch.float32, torch.bfloat16, torch.float16):
    for (b, L) in ((16, 65536), (1, 1048576), (4, 1048576)):
        q = torch.randn(b, 1, 24, device="cuda").to(dtype)
        k = torch.randn(b, L, 24, device="cuda").to(dtype)
        v = torch.randn(b, L, 24, device="cuda").to(dtype)
        for tile in (256, 512, 1024):
            us = timeit(lambda: A.scaled_dot_product_attention(q, k, v, 12 ** 0.5, None, return_attn=False, heads=2, tile_keys=tile))
            gbs = 2 * b * L * 24 * k.element_size() / us / 1e3
            print("%-10s %8d %6d %5d %9.1f %9.0f %6.1f%%" % (str(dtype).split(".")[1], L, b, tile, us, gbs, gbs / 80.0))
