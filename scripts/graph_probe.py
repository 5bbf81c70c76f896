"""Capture/replay probe: one graphed training step at 64x64 with a given stream count (ISA_STREAMS)."""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import isa_amd  # noqa
from isa_amd.reseg import ReSeg
from isa_amd.trainer import Trainer
from isa_amd.data import synth_batch
m = ReSeg(2, True, dtype=torch.bfloat16); m.reset_parameters(seed=23); m.train()
tr = Trainer(m)
x, sem, ins, n = synth_batch(2, 64, 64, seed=3)
x, sem, ins = x.cuda(), sem.cuda(), ins.cuda()
sel = [list(range(int(k))) for k in n.view(-1)]
for i in range(4):
    tr.train_step_graphed(x, sem, ins, n, selected_idx=sel)
    torch.cuda.synchronize()
    print("step", i, "ok", [s.get("state") for s in tr._graphs.values()], flush=True)
print("finite", bool(torch.isfinite(m.store.flat).all()))
