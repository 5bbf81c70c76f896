"""One InvertedResidual block (pw-BN-ReLU6-dw-BN-ReLU6-pw-BN +x) forward+backward at the 256x256 level, bf16,
for rocprofv3 --pmc / --kernel-trace runs of the fused backward kernels: python scripts/pmc_block.py [reps]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import isa_amd  # noqa
from isa_amd import lib as L
from isa_amd.engine import Act, Engine, ParamStore
from isa_amd.network import Network

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, hw, cin, chid = 16, 256, 32, 64
g = torch.Generator().manual_seed(0)
t = {"b.conv.0.weight": torch.randn(chid, cin, 1, 1, generator=g) * cin ** -0.5,
     "b.conv.3.weight": torch.randn(chid, 1, 3, 3, generator=g) / 3,
     "b.conv.6.weight": torch.randn(cin, chid, 1, 1, generator=g) * chid ** -0.5}
for i, c in ((1, chid), (4, chid), (7, cin)):
    t.update({"b.conv.%d.weight" % i: torch.rand(c, generator=g) + 0.5, "b.conv.%d.bias" % i: torch.randn(c, generator=g) * 0.3,
              "b.conv.%d.running_mean" % i: torch.zeros(c), "b.conv.%d.running_var" % i: torch.ones(c)})
schema = [(k, tuple(v.shape)) for k, v in t.items()] + [("b.conv.%d.num_batches_tracked" % i, ()) for i in (1, 4, 7)]
ps = ParamStore(schema, "cuda"); ps.load_state_dict(t)
eng = Engine(ps, torch.bfloat16)
net = Network.__new__(Network); net.E = eng
x = Act(torch.randn(n, hw, hw, cin, device="cuda").bfloat16(), 0, cin)
for _ in range(reps):
    eng.begin(bn_train=True, record=True)
    out = eng.new_act(n, hw, hw, cin)
    net.block_ir(x, "b", out)
    gout = eng.grads.grad_of(out)
    gout.buf.normal_()
    eng.grads.written[out.buf.data_ptr()].append((0, cin))
    eng.backward()
torch.cuda.synchronize()
print("ok")
