"""temporary: log the tensor descriptors of the depthwise forward launches of one training step"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ISA_STREAMS", "1")
import torch
import isa_amd  # noqa
from isa_amd import lib as L
from isa_amd.reseg import ReSeg
from isa_amd.trainer import Trainer
from isa_amd.data import synth_batch
m = ReSeg(2, True, dtype=torch.bfloat16).cuda().train()
tr = Trainer(m)
x, sem, ins, n = synth_batch(16, 256, 256, seed=0)
x, sem, ins = x.cuda(), sem.cuda(), ins.cuda()
tr.forward_backward(x, sem, ins, n)
torch.cuda.synchronize()
E = m.engine
real = E.lib.isa_dwconv3x3
log = []
def spy(xd, xp, w, bias, yd, stats, stream, *rest):
    xt = C.cast(xd, C.POINTER(L.IsaTensor)).contents
    yt = C.cast(yd, C.POINTER(L.IsaTensor)).contents
    pro = C.cast(xp, C.POINTER(L.IsaPro)).contents if xp else None
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = real(xd, xp, w, bias, yd, stats, stream, *rest)
    e.record()
    log.append(((xt.n, xt.h, xt.w, xt.c, xt.ld), (yt.h, yt.w, yt.c, yt.ld), bool(pro and pro.bscale), pro.act if pro else -1, bool(stats), s, e))
    return rc
class LibProxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, k): return spy if k == "isa_dwconv3x3" else getattr(self._lib, k)
E.lib = LibProxy(E.lib)
tr.forward_backward(x, sem, ins, n)
torch.cuda.synchronize()
for r in log:
    if r[0][1] >= 128: print(r[0], r[1], "bscale", r[2], "act", r[3], "stats", r[4], "%.1f us" % (r[5].elapsed_time(r[6]) * 1e3))
