import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import isa_amd  # noqa
from isa_amd import lib as L
from isa_amd.engine import Act, Engine, ParamStore, Pro
c, hw, B = 64, 256, 16
ps = ParamStore([("w", (c, c, 1, 1))], "cuda"); eng = Engine(ps, torch.bfloat16); eng.begin(True, False)
x = Act(torch.randn(B, hw, hw, c, device="cuda").bfloat16(), 0, c); dy = Act(torch.randn(B, hw, hw, c, device="cuda").bfloat16(), 0, c)
for _ in range(3):
    L.check(eng.lib.isa_conv_wgrad(x.d(), None, dy.d(), ps.gptr("w"), None, 0, 0, None, c, L.ptr(eng.ws), eng.ws.numel(), None, L.stream_ptr()), "wg")
torch.cuda.synchronize()
