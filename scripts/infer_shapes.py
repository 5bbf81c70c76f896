#!/usr/bin/env python3
"""Per-launch breakdown of one instrumented (eager) GT-free inference pass at bs=16 (or argv[1]), bf16: launches grouped by
entry point and algorithmic bytes, sorted by total time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import isa_amd  # noqa
from isa_amd.reseg import ReSeg
from isa_amd.data import synth_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = ReSeg(2, False, dtype=torch.bfloat16).cuda().eval()
x = synth_batch(B, S, S, seed=0)[0].cuda()
with torch.no_grad():
    for _ in range(2):
        m(False, x)
    torch.cuda.synchronize()
    E = m.engine
    E.profile = True
    m(False, x)
torch.cuda.synchronize()
groups = {}
for name, s, e, nbytes in E.prof_events:
    g = groups.setdefault((name, int(nbytes)), [0, 0.0])
    g[0] += 1; g[1] += s.elapsed_time(e)
tot = sum(v[1] for v in groups.values())
print("total instrumented kernel time %.3f ms, %d launches" % (tot, sum(v[0] for v in groups.values())))
print("%-28s %10s %6s %9s %9s %8s" % ("entry point", "MB/launch", "calls", "total ms", "avg us", "GB/s"))
for (name, nb), (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-28s %10.2f %6d %9.3f %9.1f %8.0f" % (name, nb / 1e6, c, t, t / c * 1e3, nb / 1e9 / (t / c * 1e-3) if nb else 0))
