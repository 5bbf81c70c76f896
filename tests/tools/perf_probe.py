"""Time the inference forward (backbone + SE + sem head) at the benchmark shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import isa_amd  # noqa
from isa_amd.engine import Engine, ParamStore
from isa_amd.network import Network
import reseg_ref as R

def main(dtype, batch=16, size=256, steps=10, bn_train=False):
    sd = R.synth_state_dict(23, True)
    ps = ParamStore(R.state_dict_schema(True), "cuda"); ps.load_state_dict(sd)
    eng = Engine(ps, dtype); net = Network(eng)
    x = torch.randn(batch, 21, size, size, device="cuda")
    def step():
        eng.begin(bn_train=bn_train, record=False)
        xin = net.to_nhwc(x)
        y, feats = net.unet(xin)
        sem = net.sem_head(y)
        return net.argmax_map(sem)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.time() - t) / steps
    # graph capture
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(steps): g.replay()
    torch.cuda.synchronize()
    dg = (time.time() - t) / steps
    print("dtype=%s bn_train=%s batch=%d: eager %.2f ms (%.0f img/s)  graph %.2f ms (%.0f img/s)  arena %.1f MB" % (
        dtype, bn_train, batch, dt * 1e3, batch / dt, dg * 1e3, batch / dg, eng.arena.bytes() / 1e6), flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "bf16"): main(torch.bfloat16)
    if which in ("all", "f32"): main(torch.float32)
    if which in ("all",): main(torch.bfloat16, bn_train=True)
