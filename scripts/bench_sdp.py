"""isa_sdp_attention (d_k = d_v = 12, n_head = 2 interleaved): one query per image against L = H*W keys at 256x256
(L = 65 536, batch 16) and 1024x1024 (L = 1 048 576, batch 1 and 4), storage f32 / bf16 / f16, LDS tiles of 256 / 512 /
1024 keys.  GB/s = K and V read once / time, against 8 TB/s.
Rows: `warm` = the same operands back to back (<= 100 MB: Infinity-Cache assisted, an upper bound); `rot` = SDP_ROTATE
operand sets cycled so that every launch finds K and V in HBM (> 256 MB between two uses of a set); `rot+mask` adds the
byte mask the reference passes (utils.py:323), `rot+mask+attn` also returns the attention map like the reference does
(one extra fp32 write + normalise pass per key)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import isa_amd  # noqa
from isa_amd import attention_ops as A


def timeit(fns, reps=24):
    """Device time per call: `reps` calls (cycling through fns) are captured once in a hipGraph and replayed."""
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fns[i % len(fns)]()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


tiles = [int(t) for t in os.environ.get("SDP_TILES", "256,512,1024").split(",")]
print("%-9s %8s %6s %5s %-14s %9s %9s %7s" % ("dtype", "L", "batch", "tile", "operands", "us", "GB/s", "of 8TB/s"))
for dtype in (torch.float32, torch.bfloat16, torch.float16):
    for (b, L) in ((16, 65536), (1, 1048576), (4, 1048576)):
        esz = 4 if dtype == torch.float32 else 2
        nbytes = 2 * b * L * 24 * esz
        nsets = max(2, -(-(300 << 20) // nbytes) + 1)          # > 256 MB between two uses of the same set
        sets = [(torch.randn(b, 1, 24, device="cuda").to(dtype), torch.randn(b, L, 24, device="cuda").to(dtype),
                 torch.randn(b, L, 24, device="cuda").to(dtype)) for _ in range(nsets)]
        mask = torch.rand(b, 1, L, device="cuda") < 0.25
        for tile in tiles:
            def mk(q, k, v, m=None, attn=False):
                return lambda: A.scaled_dot_product_attention(q, k, v, 12 ** 0.5, m, return_attn=attn, heads=2, tile_keys=tile)
            rows = [("warm", [mk(*sets[0])]), ("rot", [mk(*s_) for s_ in sets])]
            if tile == 512:
                rows += [("rot+mask", [mk(*s_, m=mask) for s_ in sets]), ("rot+mask+attn", [mk(*s_, m=mask, attn=True) for s_ in sets])]
            for name, fns in rows:
                us = timeit(fns)
                gbs = nbytes / us / 1e3
                print("%-9s %8d %6d %5d %-14s %9.1f %9.0f %6.1f%%" % (str(dtype).split(".")[1], L, b, tile, name, us, gbs, gbs / 80.0))
