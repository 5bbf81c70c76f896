"""Kernel micro-benchmarks at the shapes that dominate the train step (event-timed, L2-cold-ish)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import isa_amd  # noqa
from isa_amd import lib as L
from isa_amd.engine import Act, Engine, ParamStore, Pro

_FLUSH = None

def timeit(fn, reps=20):
    """KBENCH_FLUSH=1: a 512 MB fill runs between repetitions and every repetition has its own event pair, so the
    operands come from HBM and not from the 256 MB Infinity Cache a back-to-back loop leaves them in."""
    global _FLUSH
    for _ in range(3): fn()
    torch.cuda.synchronize()
    if os.environ.get("KBENCH_FLUSH") == "1":
        if _FLUSH is None:
            _FLUSH = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
        tot = 0.0
        for i in range(reps):
            _FLUSH.fill_(i & 1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record()
            torch.cuda.synchronize()
            tot += s.elapsed_time(e)
        return tot / reps * 1e3
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us

def rnd(n, h, w, c, dtype):
    return Act(torch.randn(n, h, w, (c + 7) // 8 * 8, device="cuda").to(dtype), 0, c)

def main(dtype=torch.bfloat16, which=None):
    esz = 2 if dtype == torch.bfloat16 else 4
    B = 16
    rows = []
    shapes = [(256, 32), (256, 64), (128, 128), (64, 256), (16, 1024)]
    if os.environ.get("KBENCH_SHAPES"):            # e.g. KBENCH_SHAPES=256x64,128x128
        shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["KBENCH_SHAPES"].split(",")]
    for hw, c in shapes:
        schema = [("w", (c, c, 1, 1)), ("wd", (c, 1, 3, 3)), ("bn.weight", (c,)), ("bn.bias", (c,)),
                  ("bn.running_mean", (c,)), ("bn.running_var", (c,)), ("bn.num_batches_tracked", ())]
        ps = ParamStore(schema, "cuda")
        ps.load_state_dict({"w": torch.randn(c, c, 1, 1) * c ** -0.5, "wd": torch.randn(c, 1, 3, 3) / 3,
                            "bn.weight": torch.ones(c), "bn.bias": torch.zeros(c), "bn.running_mean": torch.zeros(c),
                            "bn.running_var": torch.ones(c)})
        eng = Engine(ps, dtype)
        eng.begin(True, False)
        sc, sh = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
        nbytes = B * hw * hw * c * esz
        st = torch.zeros(16 * c, device="cuda")
        mean, inv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        red = torch.zeros(16 * c, device="cuda")
        reg = eng.reg_conv("w"); regd = eng.reg_dw("wd"); eng.packer.pack()
        lib = eng.lib
        import ctypes as C
        red2 = torch.rand(16 * c, device="cuda")
        xred = torch.zeros(16 * c, device="cuda")
        dgam, dbet = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        ybn = L.IsaBnBwd(sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), red2.data_ptr(), None,
                         dgam.data_ptr(), dbet.data_ptr(), float(B * hw * hw), L.ACT_RELU6)
        xbn = L.IsaBnBwd(sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), None, xred.data_ptr(), None, None,
                         float(B * hw * hw), L.ACT_RELU6)
        wlin = torch.randn(c, c, device="cuda") * c ** -0.5
        dwl = torch.zeros(c, c, device="cuda")
        # KBENCH_ROTATE=n: n independent operand sets are cycled, so with n x (4 tensors) > 256 MB every repetition finds its
        # inputs in HBM like a launch inside a training step does (back-to-back runs on one set are Infinity-Cache warm)
        nsets = int(os.environ.get("KBENCH_ROTATE", "1"))
        def make_tests():
            x, y, dy, dx = rnd(B, hw, hw, c, dtype), rnd(B, hw, hw, c, dtype), rnd(B, hw, hw, c, dtype), rnd(B, hw, hw, c, dtype)
            xl = x.with_pro(Pro(sc, sh, L.ACT_RELU6))
            tests = {
                "conv1x1": (lambda: lib.isa_conv_gemm(x.d(), None, eng.packer.ptr(reg["fwd"]), reg["kp"], None, y.d(), 0, 0, L.ptr(st), 0, L.stream_ptr()), 2),
                "conv1x1+pro": (lambda: lib.isa_conv_gemm(xl.d(), xl.p(), eng.packer.ptr(reg["fwd"]), reg["kp"], None, y.d(), 0, 0, L.ptr(st), 0, L.stream_ptr()), 2),
                "wgrad1x1": (lambda: lib.isa_conv_wgrad(xl.d(), xl.p(), dy.d(), ps.gptr("w"), None, 0, 0, None, c, L.ptr(eng.ws), eng.ws.numel(), None, L.stream_ptr()), 2),
                "wgrad1x1_nopro": (lambda: lib.isa_conv_wgrad(x.d(), None, dy.d(), ps.gptr("w"), None, 0, 0, None, c, L.ptr(eng.ws), eng.ws.numel(), None, L.stream_ptr()), 2),
                "dw+pro": (lambda: lib.isa_dwconv3x3(xl.d(), xl.p(), eng.packer.ptr(regd["fwd"]), None, y.d(), L.ptr(st), L.stream_ptr()), 2),
                "dw_dgrad": (lambda: lib.isa_dwconv3x3_dgrad(dy.d(), eng.packer.ptr(regd["dgrad"]), y.d(), 0, L.stream_ptr()), 2),
                "dw_wgrad": (lambda: lib.isa_dwconv3x3_wgrad(xl.d(), xl.p(), dy.d(), ps.gptr("wd"), None, c, L.ptr(eng.ws), eng.ws.numel(), None, L.stream_ptr()), 2),
                "dw_bn_bwd": (lambda: lib.isa_dwconv3x3_bn_backward(dy.d(), y.d(), C.byref(ybn), xl.d(), xl.p(), C.byref(xbn), eng.packer.ptr(regd["dgrad"]), ps.gptr("wd"), c, dx.d(), 0, None, L.ptr(eng.ws), eng.ws.numel(), None, L.stream_ptr()), 4),
                "pw_bn_bwd": ((lambda: lib.isa_conv1x1_bn_backward(dy.d(), y.d(), C.byref(ybn), xl.d(), xl.p(), C.byref(xbn), L.ptr(wlin), L.ptr(dwl), dx.d(), 0, None, L.ptr(eng.ws), eng.ws.numel(), None, L.stream_ptr())) if c <= 64 else None, 4),
                "bn_bwd_reduce": (lambda: lib.isa_bn_bwd_reduce(dy.d(), x.d(), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(inv), L.ACT_RELU6, None, L.ptr(red), L.stream_ptr()), 2),
                "bn_bwd_apply": (lambda: lib.isa_bn_bwd_apply(dy.d(), x.d(), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(inv), L.ACT_RELU6, None, ps.ptr("bn.weight"), L.ptr(red), float(B * hw * hw), 1, y.d(), None, None, L.stream_ptr()), 3),
                "materialize+res": (lambda: lib.isa_affine_act_res(xl.d(), xl.p(), dy.d(), None, None, y.d(), L.stream_ptr()), 3),
                "materialize": (lambda: lib.isa_affine_act_res(xl.d(), xl.p(), None, None, None, y.d(), L.stream_ptr()), 2),
            }
            return tests
        sets = [make_tests() for _ in range(nsets)]
        tests = sets[0]
        def rotating(name):
            fns = [t[name][0] for t in sets]
            if fns[0] is None: return None
            state = [0]
            def call():
                state[0] = (state[0] + 1) % len(fns)
                return fns[state[0]]()
            return call
        for name, (fn, ntens) in tests.items():
            if fn is None or (which and which not in name): continue
            us = timeit(rotating(name))
            rows.append((name, hw, c, us, ntens * nbytes / us / 1e3))
    print("%-18s %5s %5s %10s %10s" % ("kernel", "hw", "c", "us", "GB/s(alg)"))
    for r in sorted(rows):
        print("%-18s %5d %5d %10.1f %10.0f" % r)

if __name__ == "__main__":
    main(torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] != "f32") else torch.float32, sys.argv[2] if len(sys.argv) > 2 else None)
