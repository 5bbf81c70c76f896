import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import isa_amd  # noqa
from isa_amd import lib as L
from isa_amd.engine import Act, Engine, ParamStore, Pro
c, hw, B = 64, 256, 16
ps = ParamStore([("wd", (c, 1, 3, 3))], "cuda"); ps.load_state_dict({"wd": torch.randn(c, 1, 3, 3) / 3})
eng = Engine(ps, torch.bfloat16); eng.begin(True, False)
x = Act(torch.randn(B, hw, hw, c, device="cuda").bfloat16(), 0, c); y = Act(torch.empty(B, hw, hw, c, device="cuda", dtype=torch.bfloat16), 0, c)
dy = Act(torch.randn(B, hw, hw, c, device="cuda").bfloat16(), 0, c)
sc, sh = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
xl = x.with_pro(Pro(sc, sh, L.ACT_RELU6)); st = torch.zeros(16 * c, device="cuda")
reg = eng.reg_dw("wd"); eng.packer.pack()
for _ in range(3):
    L.check(eng.lib.isa_dwconv3x3(xl.d(), xl.p(), eng.packer.ptr(reg["fwd"]), None, y.d(), L.ptr(st), L.stream_ptr()), "dw")
    L.check(eng.lib.isa_dwconv3x3_dgrad(dy.d(), eng.packer.ptr(reg["dgrad"]), y.d(), 0, L.stream_ptr()), "dwd")
    L.check(eng.lib.isa_dwconv3x3_wgrad(xl.d(), xl.p(), dy.d(), ps.gptr("wd"), None, c, L.ptr(eng.ws), eng.ws.numel(), None, L.stream_ptr()), "dww")
torch.cuda.synchronize()
