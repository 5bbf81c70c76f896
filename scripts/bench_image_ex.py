"""isa_image_ex throughput next to the numpy oracle on the host cores:
python scripts/bench_image_ex.py [batch] [size]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
import isa_amd  # noqa
from isa_amd import lib as L
from isa_amd.engine import Act

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device="cuda")
for dt, name, esz in ((torch.bfloat16, "bf16", 2), (torch.float32, "f32", 4)):
    out = Act(torch.empty(B, S, S, 24, dtype=dt, device="cuda"), 0, 21)
    for _ in range(3):
        L.check(L.lib().isa_image_ex(L.ptr(img), out.d(), L.stream_ptr()), "isa_image_ex")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    reps = 20
    for _ in range(reps):
        L.check(L.lib().isa_image_ex(L.ptr(img), out.d(), L.stream_ptr()), "isa_image_ex")
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    alg = B * S * S * (3 + 21 * esz)
    print("isa_image_ex %s  %dx%dx%d  %.1f us  %.0f images/s  %.2f TB/s algorithmic (3 B in + 21 ch out per pixel)" %
          (name, B, S, S, us, B / us * 1e6, alg / us / 1e6))
# cpu_baseline leg (the only use of the oracle here): the numpy restatement on one host thread
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import image_ex_ref as IX
host = img[:4].cpu().numpy()
t0 = time.perf_counter(); IX.image_ex_standardized(host); dt_ = time.perf_counter() - t0
print("numpy oracle (1 host thread): %.1f images/s" % (4 / dt_))
