#!/usr/bin/env python3
"""Per-launch breakdown of one instrumented (eager, single-stream) training step: launches grouped by entry point and
algorithmic bytes (= shape), sorted by total time.  Shows which shapes of a kernel family carry its time.
usage: python scripts/step_shapes.py [family-substring] [bf16|f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ISA_STREAMS", "1")
import torch
import isa_amd  # noqa
from isa_amd.reseg import ReSeg
from isa_amd.trainer import Trainer
from isa_amd.data import synth_batch

which = sys.argv[1] if len(sys.argv) > 1 else ""
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
torch.manual_seed(0)
m = ReSeg(2, True, dtype=dtype).cuda().train()
tr = Trainer(m)
x, sem, ins, n = synth_batch(16, 256, 256, seed=0)
x, sem, ins = x.cuda(), sem.cuda(), ins.cuda()
for _ in range(2):
    tr.forward_backward(x, sem, ins, n)
torch.cuda.synchronize()
E = m.engine
E.profile = True
tr.forward_backward(x, sem, ins, n)
torch.cuda.synchronize()
groups = {}
for name, s, e, nbytes in E.prof_events:
    g = groups.setdefault((name, int(nbytes)), [0, 0.0])
    g[0] += 1; g[1] += s.elapsed_time(e)
E.prof_events = []
tot = sum(v[1] for v in groups.values())
print("total instrumented kernel time %.2f ms" % tot)
fam = {}
for (name, nb), (c, t) in groups.items():
    f = fam.setdefault(name, [0, 0.0]); f[0] += c; f[1] += t
for name, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-32s %5d launches %8.3f ms" % (name, c, t))
print()
print("%-32s %10s %6s %9s %9s %8s" % ("entry point", "MB/launch", "calls", "total ms", "avg us", "GB/s"))
rows = sorted(((k, v) for k, v in groups.items() if which in k[0]), key=lambda kv: -kv[1][1])
for (name, nb), (c, t) in rows[:int(os.environ.get("SHAPES_ROWS", "40"))]:
    print("%-32s %10.2f %6d %9.3f %9.1f %8.0f" % (name, nb / 1e6, c, t, t / c * 1e3, nb / 1e9 / (t / c * 1e-3) if nb else 0))
