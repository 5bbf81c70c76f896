import os, sys
import numpy as np, torch
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import golden_io as G, reseg_ref as R
import isa_amd
from isa_amd.reseg import ReSeg
from isa_amd.trainer import Trainer
from test_gpu_train import setup, _grad_samples
z = np.load(os.path.join(ROOT, "tests/golden/train_64_f64.npz")); z32 = np.load(os.path.join(ROOT, "tests/golden/train_64.npz"))
m, tr, batch, sel, inj = setup(ReSeg, Trainer, z, torch.float32)
m.head.streams = int(os.environ.get("NS", "1"))
tr.forward_backward(*batch, selected_idx=sel, injected_s_t=inj); torch.cuda.synchronize()
names = sorted(set(k.split("/")[1] for k in z.files if k.startswith("grad/")))
rows = []
for k in names:
    shape = tuple(int(v) for v in z["grad/%s/shape" % k])
    g = m.store.gview(k).cpu().numpy().reshape(-1).astype(np.float64)
    ref = _grad_samples(z, k); mine = g if ref.size == g.size else g[::G.subsample_stride(g.size, 512)]
    r32 = _grad_samples(z32, k); nrm = max(np.linalg.norm(ref), 1e-30)
    rows.append((k, np.linalg.norm(mine - ref) / nrm, np.linalg.norm(r32 - ref) / nrm, nrm))
for k, e, f, n in rows:
    if any(s in k for s in ("sem_seg_output", "channelAttend", "pred.last_fc", "pred.l_i", "upAtten4.UpAtten.dilation_part2.1", "base.up4.conv.conv.down_conv_1", "base.inc", "ins_seg_output_2.6", "decoder.s_sp", "decoder.attend")):
        print("%-75s err %.2e  ref32 %.2e  norm %.2e" % (k, e, f, n))
es = np.array([r[1] for r in rows]); fs = np.array([r[2] for r in rows])
print("median err %.2e, median floor %.2e, p90 err %.2e, p90 floor %.2e" % (np.median(es), np.median(fs), np.percentile(es, 90), np.percentile(fs, 90)))
