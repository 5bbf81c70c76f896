"""Launches of the MFMA kernels at the shapes that dominate a training step (for rocprofv3 --pmc passes: MFMA-busy /
VALU-busy evidence, profiles/r02_pmc_mfma_busy.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import isa_amd  # noqa
from isa_amd import lib as L
from isa_amd.engine import Act, Engine, ParamStore, Pro
B = 16
for hw, cin, cout in ((256, 64, 32), (256, 32, 64), (64, 256, 128), (16, 1024, 256), (16, 512, 1024)):
    ps = ParamStore([("w", (cout, cin, 1, 1))], "cuda"); ps.load_state_dict({"w": torch.randn(cout, cin, 1, 1) * cin ** -0.5})
    eng = Engine(ps, torch.bfloat16); eng.begin(True, False)
    x = Act(torch.randn(B, hw, hw, cin, device="cuda").bfloat16(), 0, cin)
    y = Act(torch.empty(B, hw, hw, cout, device="cuda", dtype=torch.bfloat16), 0, cout)
    dy = Act(torch.randn(B, hw, hw, cout, device="cuda").bfloat16(), 0, cout)
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    xl = x.with_pro(Pro(sc, sh, L.ACT_RELU6)); st = torch.zeros(16 * cout, device="cuda")
    reg = eng.reg_conv("w"); eng.packer.pack()
    for _ in range(3):
        L.check(eng.lib.isa_conv_gemm(xl.d(), xl.p(), eng.packer.ptr(reg["fwd"]), reg["kp"], None, y.d(), 0, 0, L.ptr(st), 0, L.stream_ptr()), "fwd")
        L.check(eng.lib.isa_conv_wgrad(xl.d(), xl.p(), dy.d(), ps.gptr("w"), None, 0, 0, None, cin, L.ptr(eng.ws), eng.ws.numel(), None, L.stream_ptr()), "wg")
    torch.cuda.synchronize()
