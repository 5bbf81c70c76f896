"""Train-mode BatchNorm finalize inside the consumer of the lazy tensor (isa_pro.fin, include/isa_kernels.h; engine
PendingFin) against the stand-alone isa_bn_finalize launch it replaces (nn.BatchNorm2d train forward as the reference
uses it: MobileNetDenseASPP.py:68-123):
  * the constants a consumer derives (scale / shift / mean / invstd, running statistics) are BIT-identical to the
    launch's - same device function on the same sums - for every in-kernel consumer (streaming and LDS-tiled GEMM,
    depthwise, materialising pass), with statistic groups and the repeat count;
  * whole InvertedResidual blocks, forward and backward, agree between ISA_INLINE_FIN=1 and 0 (the sums themselves come
    from float atomics, so two runs of either mode differ in the last bits: tolerance as in test_gpu_groups.py), the
    running statistics are updated exactly once per BatchNorm and no isa_bn_finalize launch is left in the block;
  * a lazy tensor with two consumers, on one stream and on two, is finalized correctly for both and updates the running
    statistics once."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from test_gpu_ops import _gpu, make_engine, rand, rel, to_act  # noqa: E402
from test_gpu_groups import SHAPES, _params, _run  # noqa: E402


def _bn_schema(c):
    return [("bn.weight", (c,)), ("bn.bias", (c,)), ("bn.running_mean", (c,)), ("bn.running_var", (c,)),
            ("bn.num_batches_tracked", ())]


def _bn_tensors(c):
    return {"bn.weight": rand(c, seed=5).abs() + 0.5, "bn.bias": rand(c, seed=6),
            "bn.running_mean": rand(c, seed=7) * 0.1, "bn.running_var": rand(c, seed=8).abs() + 0.5}


def _stats(x, G):
    """[G][8][2c] sums spread over the replicas (the consumer must add them in replica order, like the launch)."""
    n, c = x.shape[0] // G, x.shape[1]
    st = torch.zeros(G, 8, 2 * c)
    for g in range(G):
        xg = x[g * n:(g + 1) * n]
        s1, s2 = xg.sum((0, 2, 3)), (xg * xg).sum((0, 2, 3))
        w = torch.rand(8, 1, generator=torch.Generator().manual_seed(90 + g))
        w = w / w.sum()
        st[g] = torch.cat([w * s1[None], w * s2[None]], 1)
    return st.reshape(-1).cuda()


CONSUMERS = ["conv", "conv_tiled", "dw", "bn_out"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("consumer", CONSUMERS)
@pytest.mark.parametrize("G,rep", [(1, 1), (2, 1), (1, 2)])
def test_constants_are_bit_identical_to_the_launch(dtype, consumer, G, rep, monkeypatch):
    L, Act, Engine, ParamStore, Pro = _gpu()
    c = 128 if consumer == "conv_tiled" else 40
    cout, B, h, w = 48, 3, 6, 10
    x = rand(G * B, c, h, w, seed=3, scale=2.0) + 0.5
    wt = rand(cout, c, 1, 1, seed=11, scale=c ** -0.5) if consumer != "dw" else rand(c, 1, 3, 3, seed=11, scale=1 / 3.0)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ISA_INLINE_FIN", mode)
        tensors = dict(_bn_tensors(c)); tensors["w.weight"] = wt
        schema = _bn_schema(c) + [("w.weight", tuple(wt.shape))]
        eng = make_engine(Engine, ParamStore, schema, tensors, dtype)
        eng.profile = True
        xa = to_act(Act, x, dtype)
        xa = Act(xa.buf, 0, c, groups=G)
        st = _stats(x.to(dtype).float(), G)
        with eng.repeated(rep) if rep > 1 else _null():
            if consumer == "bn_out":
                out = eng.new_act(G * B, h, w, c, groups=G)
                eng.bn_out(xa, st, "bn", L.ACT_RELU6, out)
                lazy_bn = None
            else:
                lazy = eng.bn(xa, st, "bn", L.ACT_RELU6)
                lazy_bn = lazy.bn
                if consumer == "dw":
                    out = eng.new_act(G * B, h, w, c, groups=G)
                    eng.dwconv(lazy, "w.weight", out)
                else:
                    out = eng.new_act(G * B, h, w, cout, groups=G)
                    eng.conv(lazy, "w.weight", out)
        calls = eng.profile_summary()
        torch.cuda.synchronize()
        if mode == "1":
            assert "isa_bn_finalize" not in calls, calls
        else:
            assert "isa_bn_finalize" in calls
        consts = [lazy_bn[k].clone().cpu() for k in ("scale", "shift", "mean", "invstd")] if lazy_bn is not None else []
        res[mode] = (out.nchw().float().cpu(), consts, eng.params.view("bn.running_mean").clone().cpu(),
                     eng.params.view("bn.running_var").clone().cpu(), eng.params.int_buffers["bn.num_batches_tracked"])
    a, b = res["1"], res["0"]
    assert torch.equal(a[0], b[0])                       # same constants, same kernel arithmetic, no atomics in between
    for u, v in zip(a[1], b[1]):
        assert torch.equal(u, v)
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert not torch.equal(a[2], _bn_tensors(c)["bn.running_mean"])     # and they did move
    assert a[4] == b[4] == G * rep


class _null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("G", [1, 2])
@pytest.mark.parametrize("cin,chid,cout,h,w,B", SHAPES)
def test_blocks_agree_with_the_launch_per_batchnorm_path(dtype, G, cin, chid, cout, h, w, B, monkeypatch):
    t = _params(cin, chid, cout)
    x = torch.cat([rand(B, cin, h, w, seed=60 + g) * (1.0 + 0.5 * g) + 0.3 * g for g in range(G)])
    dy = rand(G * B, cout, h, w, seed=70)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ISA_INLINE_FIN", mode)
        out[mode] = _run(t, dtype, True, x, dy, G)
    (o1, dx1, g1, b1, n1, c1), (o0, dx0, g0, b0, n0, c0) = out["1"], out["0"]
    assert "isa_bn_finalize" not in c1 and c0["isa_bn_finalize"][0] == 3
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel(o1, o0) < tol and rel(dx1, dx0) < tol
    for k in g0:
        assert rel(g1[k].view(-1), g0[k].view(-1)) < (2e-5 if dtype == torch.float32 else 1.5e-2), k
    for k in b0:
        assert rel(b1[k], b0[k]) < 1e-6, k
        assert rel(b1[k], t[k]) > 1e-3, k
    assert n1 == n0 and all(v == G for v in n1.values())


@pytest.mark.parametrize("streams", [1, 2])
def test_two_consumers_share_one_running_update(streams, monkeypatch):
    """A lazy BatchNorm output read by a GEMM and by a depthwise conv (both can finalize in-kernel) and then by an entry
    point that cannot (isa_chan_mean): with one stream the first consumer finalizes and the others read its arrays; with
    the second consumer on a side stream both finalize (identical values), only the first carries the running update."""
    L, Act, Engine, ParamStore, Pro = _gpu()
    dtype = torch.bfloat16
    c, cout, B, h, w = 32, 16, 2, 8, 8
    x = rand(B, c, h, w, seed=3, scale=2.0) + 0.5
    w1, w2 = rand(cout, c, 1, 1, seed=11, scale=c ** -0.5), rand(c, 1, 3, 3, seed=12, scale=1 / 3.0)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ISA_INLINE_FIN", mode)
        tensors = dict(_bn_tensors(c)); tensors["w1.weight"] = w1; tensors["w2.weight"] = w2
        schema = _bn_schema(c) + [("w1.weight", tuple(w1.shape)), ("w2.weight", tuple(w2.shape))]
        eng = make_engine(Engine, ParamStore, schema, tensors, dtype)
        xa = to_act(Act, x, dtype)
        st = _stats(x.to(dtype).float(), 1)
        lazy = eng.bn(xa, st, "bn", L.ACT_RELU6)
        o1, o2 = eng.new_act(B, h, w, cout), eng.new_act(B, h, w, c)
        eng.conv(lazy, "w1.weight", o1)
        if streams == 2:
            eng.sync(0, 1)
            with eng.on(1):
                eng.dwconv(lazy, "w2.weight", o2)
            eng.sync(1, 0)
        else:
            eng.dwconv(lazy, "w2.weight", o2)
        mean = torch.zeros(B, c, dtype=torch.float32, device="cuda")     # isa_chan_mean accumulates
        L.check(eng.lib.isa_chan_mean(lazy.d(), lazy.p(), L.ptr(mean), eng.st()), "isa_chan_mean")
        torch.cuda.synchronize()
        res[mode] = (o1.nchw().float().cpu(), o2.nchw().float().cpu(), mean.cpu(),
                     eng.params.view("bn.running_mean").clone().cpu(), eng.params.view("bn.running_var").clone().cpu())
    for i, (u, v) in enumerate(zip(res["1"], res["0"])):
        if i == 2:
            assert rel(u, v) < 1e-5          # isa_chan_mean adds workgroup partial sums with float atomics: order varies
        else:
            assert torch.equal(u, v)


def test_entry_points_without_the_kernel_form_launch_the_finalize_themselves(monkeypatch):
    """isa_pro.fin handed to isa_chan_mean / isa_conv_wgrad (no in-kernel form): the entry point runs isa_bn_finalize on the
    stream first, so the result equals the launch-per-BatchNorm path and the arrays / running statistics are written."""
    L, Act, Engine, ParamStore, Pro = _gpu()
    dtype = torch.bfloat16
    c, cout, B, h, w = 32, 16, 2, 8, 8
    x = rand(B, c, h, w, seed=3, scale=2.0) + 0.5
    dy = rand(B, cout, h, w, seed=4)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ISA_INLINE_FIN", mode)
        outs = []
        for which in ("chan_mean", "wgrad"):
            eng = make_engine(Engine, ParamStore, _bn_schema(c), _bn_tensors(c), dtype)
            xa = to_act(Act, x, dtype)
            lazy = eng.bn(xa, _stats(x.to(dtype).float(), 1), "bn", L.ACT_RELU6)
            if which == "chan_mean":
                out = torch.zeros(B, c, dtype=torch.float32, device="cuda")
                L.check(eng.lib.isa_chan_mean(lazy.d(), lazy.p_fin(), L.ptr(out), eng.st()), "isa_chan_mean")
            else:
                out = torch.zeros(cout, c, dtype=torch.float32, device="cuda")
                ws = torch.empty(8 << 20, dtype=torch.float32, device="cuda")
                L.check(eng.lib.isa_conv_wgrad(lazy.d(), lazy.p_fin(), to_act(Act, dy, dtype).d(), L.ptr(out), None, L.IN_1X1,
                                               L.OUT_PLAIN, None, 0, L.ptr(ws), ws.numel(), None, eng.st()), "isa_conv_wgrad")
            torch.cuda.synchronize()
            outs += [out.cpu(), lazy.bn["scale"].clone().cpu(), lazy.bn["invstd"].clone().cpu(),
                     eng.params.view("bn.running_var").clone().cpu()]
        res[mode] = outs
    for i, (u, v) in enumerate(zip(res["1"], res["0"])):
        if i == 0:
            assert rel(u, v) < 1e-5              # isa_chan_mean: float atomics
        else:
            assert torch.equal(u, v), i
    assert not torch.equal(res["1"][3], _bn_tensors(c)["bn.running_var"])
