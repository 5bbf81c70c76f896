"""GPU bring-up: backbone + semantic head vs the CPU oracle (run through gpurun)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import isa_amd  # noqa
from isa_amd.engine import Engine, ParamStore
from isa_amd.network import Network
import reseg_ref as R

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def run(dtype, bn_train, size=64, batch=2):
    sd = R.synth_state_dict(23, True)
    ps = ParamStore(R.state_dict_schema(True), "cuda")
    ps.load_state_dict(sd)
    eng = Engine(ps, dtype)
    net = Network(eng)
    x, sem, ins, n = R.synth_batch(batch, size, size, seed=1)
    ctx = R.Ctx(bn_train=bn_train, capture=True)
    with torch.no_grad():
        ref = R.reseg_forward(sd, x, ctx=ctx)
    eng.begin(bn_train=bn_train, record=False)
    xin = net.to_nhwc(x.cuda())
    y, feats = net.unet(xin)
    sem_a = net.sem_head(y)
    am = net.argmax_map(sem_a)
    torch.cuda.synchronize()
    out = {}
    for nm, a in zip(("x1", "x2", "x3", "x4", "x5"), feats):
        out[nm] = rel(a.nchw(), ctx.taps["unet." + nm])
    out["x_dec"] = rel(y.nchw(), ctx.taps["unet.x_dec"])
    so = net.to_nchw(sem_a)
    out["sem_out"] = rel(so, ref["sem_out"])
    mism = int((am.nchw().cpu() != ref["sem_argmax"]).sum())
    print("dtype=%s bn_train=%s size=%d" % (dtype, bn_train, size), {k: "%.2e" % v for k, v in out.items()}, "argmax mismatches", mism, flush=True)
    return out

if __name__ == "__main__":
    for dtype in (torch.float32, torch.bfloat16):
        for bn_train in (False, True):
            run(dtype, bn_train)
    run(torch.float32, False, size=256, batch=2)
