import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import isa_amd  # noqa
from isa_amd.reseg import ReSeg
from isa_amd.trainer import Trainer
import reseg_ref as R
z = np.load(os.path.join(ROOT, "tests", "golden", "train_256.npz"))
x, sem, ins, n = R.synth_batch(2, 256, 256, seed=1)
sel = [[int(v) for v in row if v >= 0] for row in z["inject/selected_idx"]]
for dtype in (torch.float32,):
    junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(12)]   # 12 GiB of NaN
    del junk                                                                        # returned to the caching allocator
    m = ReSeg(2, True, dtype=dtype); m.load_state_dict(R.synth_state_dict(23, True)); m.train(); m.head.drop_rate = 0.0
    tr = Trainer(m)
    inj = [torch.tensor(row, dtype=torch.int32, device="cuda") for row in z["inject/s_t"]]
    out = tr.forward_backward(x, sem, ins, n, selected_idx=sel, injected_s_t=inj)
    torch.cuda.synchronize()
    bad = [k for k in m.store.offsets if "running" not in k and not torch.isfinite(m.store.gview(k)).all()]
    print(dtype, "non-finite grads:", len(bad))
    good = [k for k in m.store.offsets if k.startswith("base.") and "running" not in k and k not in bad]
    print("   finite base grads:", good)
    print("   non-finite outside base:", [k for k in bad if not k.startswith("base.")])
    for it, r in enumerate(m.last_record["iters"]):
        a = r["alpha"].view(2, -1)
        print("   it", it, "alpha[s_t] =", [float(a[b, int(inj[it][b])]) for b in range(2)], "s_t", inj[it].tolist())
    tr.apply_update(); torch.cuda.synchronize()
    print("   sqnorm", tr.sqnorm.tolist(), "flat finite:", bool(torch.isfinite(m.store.flat).all()))
