"""bf16-vs-fp32 storage error statistics per captured tensor at 256x256 (diagnostic for the parity bounds)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import golden_io as G, reseg_ref as R
import isa_amd
from isa_amd.reseg import ReSeg
from test_gpu_model import _run_gt, load
z = load("train_256")
m32, out32, cap32 = _run_gt(ReSeg, z, torch.float32, True)
a32 = {k: v.nchw().cpu() for k, v in cap32.items() if k.startswith(("it", "unet.")) or k == "x_enc"}
a32["sem_out"] = out32[0].cpu()
del m32, cap32
m, out, cap = _run_gt(ReSeg, z, torch.bfloat16, True)
a16 = {k: cap[k].nchw().cpu() for k in a32 if k != "sem_out"}; a16["sem_out"] = out[0].cpu()
for k in sorted(a32):
    d = (a16[k] - a32[k]).double(); r = a32[k].double()
    line = "%-14s max-abs/max %.3e  rel-L2 %.3e  p99.9|d|/max %.3e" % (k, d.abs().max() / r.abs().max(), d.norm() / r.norm(), torch.quantile(d.abs().flatten()[:: max(1, d.numel() // 1000000)], 0.999) / r.abs().max())
    if k.endswith("pred") or k == "sem_out":
        l32, l16 = a32[k], a16[k]
        mg = (l32[:, 1] - l32[:, 0]).abs(); sc = float(l32.abs().max())
        m16, m32_ = l16[:, 1] > l16[:, 0], l32[:, 1] > l32[:, 0]
        for t in (0.02, 0.05, 0.1):
            keep = mg > 2 * t * sc
            inter = (m16 & m32_ & keep).sum().item(); uni = ((m16 | m32_) & keep).sum().item()
            line += "  | t=%.2f keep %.2f iou %.5f" % (t, keep.float().mean(), inter / max(uni, 1))
        line += "  raw iou %.4f" % ((m16 & m32_).sum().item() / max((m16 | m32_).sum().item(), 1))
    print(line)
print("scalars bf16", [float(v) for v in out[3:]], "fp32", [float(v) for v in out32[3:]])
