#!/usr/bin/env python3
"""PCIe-inclusive training rate: every step copies a pinned host batch into the graph's static input buffers, then
replays the step.  Two hand-overs: the reference's tensors (x fp32 [B,21,H,W], sem int64 one-hot, ins int64 planes)
and the compact one (uint8 RGB, uint8 semantic map, uint8 instance planes; ImageEx and the collate tail run on the
device).  bench.py's `value` keeps inputs resident in HBM; this is the number with the host in the loop.
python scripts/bench_pcie.py [batch] [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isa_amd  # noqa: F401
from isa_amd.reseg import ReSeg
from isa_amd.trainer import Trainer
from isa_amd.data import synth_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
S = 256
x, sem, ins, n = synth_batch(B, S, S, seed=100)
rgb = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8)
sel = [list(range(int(k))) for k in n.view(-1)]


def run(label, hx, hsem, hins):
    m = ReSeg(2, True, dtype=torch.bfloat16)
    m.reset_parameters(seed=23)
    m.train()
    tr = Trainer(m)
    hx, hsem, hins = hx.pin_memory(), hsem.pin_memory(), hins.pin_memory()
    dx, dsem, dins = hx.cuda(), hsem.cuda(), hins.cuda()
    tr.train_step_graphed(dx, dsem, dins, n, selected_idx=sel)
    tr.train_step_graphed(dx, dsem, dins, n, selected_idx=sel)
    gx, gsem, gins = tr.static_inputs()[0]

    def step():
        gx.copy_(hx, non_blocking=True); gsem.copy_(hsem, non_blocking=True); gins.copy_(hins, non_blocking=True)
        tr.train_step_graphed(gx, gsem, gins, n, selected_idx=sel)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    mb = (hx.numel() * hx.element_size() + hsem.numel() * hsem.element_size() + hins.numel() * hins.element_size()) / 1e6
    print("%-58s %7.1f MB/step over PCIe  %6.2f ms/step  %6.1f images/s" % (label, mb, dt * 1e3, B / dt))
    del m, tr
    torch.cuda.empty_cache()


def run_prefetched(label, hx, hsem, hins):
    from isa_amd.data import DevicePrefetcher
    m = ReSeg(2, True, dtype=torch.bfloat16)
    m.reset_parameters(seed=23)
    m.train()
    tr = Trainer(m)
    hx, hsem, hins = hx.pin_memory(), hsem.pin_memory(), hins.pin_memory()

    class Loader(object):
        def __init__(self, k): self.k = k
        def __len__(self): return self.k
        def __iter__(self):
            for _ in range(self.k):
                yield hx, hsem, hins, n
    for b in DevicePrefetcher(Loader(5)):
        tr.train_step_graphed(b[0], b[1], b[2], b[3], selected_idx=sel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in DevicePrefetcher(Loader(K)):
        tr.train_step_graphed(b[0], b[1], b[2], b[3], selected_idx=sel)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-58s %7s                     %6.2f ms/step  %6.1f images/s" % (label, "", dt * 1e3, B / dt))
    del m, tr
    torch.cuda.empty_cache()


print("train step 256x256 bs=%d bf16, hipGraph replay, host batch copied in every step (serial copy + step)" % B)
run("reference hand-over (x fp32, sem/ins int64)", x, sem, ins)
run("compact hand-over (rgb/sem/ins uint8, expansion on device)", rgb, sem[:, 1].contiguous().to(torch.uint8),
    ins.permute(0, 2, 3, 1).contiguous().to(torch.uint8))
print("the same with isa_amd.data.DevicePrefetcher (batch i+1 uploads on a side stream during step i)")
run_prefetched("reference hand-over, prefetched", x, sem, ins)
run_prefetched("compact hand-over, prefetched", rgb, sem[:, 1].contiguous().to(torch.uint8),
               ins.permute(0, 2, 3, 1).contiguous().to(torch.uint8))
