"""Per-step parameter divergence: eager vs eager (noise floor) and eager vs graph replay."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import isa_amd  # noqa
from isa_amd.reseg import ReSeg
from isa_amd.trainer import Trainer
import reseg_ref as R

dtype = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "f32") else torch.bfloat16
x, sem, ins, n = R.synth_batch(2, 64, 64, seed=1)
orders = [[[0, 1], [1, 0]], [[1, 0], [0, 1]], [[0, 1], [0, 1]], [[1, 0], [1, 0]], [[0, 1], [1, 0]]]

def run(graphed):
    m = ReSeg(2, True, dtype=dtype); m.load_state_dict(R.synth_state_dict(23, True)); m.train(); m.head.drop_rate = 0.0
    tr = Trainer(m); snaps = []
    for sel in orders:
        out = (tr.train_step_graphed if graphed else tr.train_step)(x, sem, ins, n, selected_idx=sel)
        torch.cuda.synchronize()
        snaps.append((m.store.flat.clone(), m.store.grad.clone(), [float(v) for v in out["head"]]))
    return snaps

a, b, g = run(False), run(False), run(True)
for i in range(len(orders)):
    for name, u, v in (("eager/eager", a, b), ("eager/graph", a, g)):
        dp = float((u[i][0] - v[i][0]).abs().max()); dg = float((u[i][1] - v[i][1]).abs().max() / (u[i][1].abs().max() + 1e-30))
        print("step %d %-12s dparam %.3e dgrad(rel) %.3e head %s | %s" % (i, name, dp, dg, ["%.5f" % t for t in u[i][2]], ["%.5f" % t for t in v[i][2]]))
