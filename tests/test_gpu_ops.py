"""Op-level parity: every HIP entry point against a torch fp32 reference of the same op.

Run on the GPU box: python -m pytest tests -m gpu.  All calls go through the C-ABI library via
the engine wrappers; the torch functions here are only the checkers.
Tolerances: fp32 storage 2e-5 relative to max|ref| (exact-f32 MFMA, different summation order);
bf16 storage 2e-2 (one bf16 rounding of inputs and of the stored result, fp32 accumulation).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd import lib as L
    from isa_amd.engine import Act, Engine, ParamStore, Pro
    return L, Act, Engine, ParamStore, Pro


DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def to_act(Act, t_nchw, dtype, ld=None, c0=0):
    """NCHW cpu tensor -> Act on GPU (optionally inside a wider buffer at channel offset c0)."""
    n, c, h, w = t_nchw.shape
    ld = ld or (c + 7) // 8 * 8
    buf = torch.full((n, h, w, ld), 7.0, dtype=dtype, device="cuda")
    buf[..., c0:c0 + c] = t_nchw.permute(0, 2, 3, 1).to(dtype).cuda()
    return Act(buf, c0, c)


def q(t, dtype):
    """Quantise a reference input the way storage does."""
    return t.to(dtype).float()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def make_engine(Engine, ParamStore, schema, tensors, dtype):
    ps = ParamStore(schema, "cuda")
    ps.load_state_dict(tensors)
    eng = Engine(ps, dtype)
    eng.begin(bn_train=True, record=False)
    return eng


def rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,hw,batch", [(32, 32, 16, 2), (21, 32, 8, 1), (64, 246, 8, 2),
                                               (512, 1024, 4, 2), (256, 120, 8, 1), (24, 12, 8, 2),
                                               (32, 2, 16, 2),
                                               # the LDS-tiled small-GEMM path (bf16, K >= 128, K % 64 == 0, N % 16 == 0):
                                               # ragged M, column tails of both tile widths, several K steps
                                               (128, 80, 12, 3), (192, 64, 9, 2), (1024, 512, 16, 1), (128, 144, 5, 1)])
def test_conv1x1(dtype, cin, cout, hw, batch):
    L, Act, Engine, ParamStore, Pro = _gpu()
    w = rand(cout, cin, 1, 1, seed=1, scale=cin ** -0.5)
    b = rand(cout, seed=2)
    x = rand(batch, cin, hw, hw + 3, seed=3)
    eng = make_engine(Engine, ParamStore, [("w", w.shape), ("b", b.shape)], dict(w=w, b=b), dtype)
    xa = to_act(Act, x, dtype)
    ya = eng.new_act(batch, hw, hw + 3, cout, ld=(cout + 15) // 8 * 8)
    ya.buf.fill_(5.0)
    _, st = eng.conv(xa, "w", ya, bias="b", stats=True)
    torch.cuda.synchronize()
    ref = F.conv2d(q(x, dtype), q(w, dtype), b)
    assert rel(ya.nchw(), ref) < TOL[dtype]
    # untouched padding channels
    assert float((ya.buf[..., cout:].float() - 5.0).abs().max()) == 0.0
    # fused BN statistics
    s = st.cpu().view(8, -1).sum(0)
    assert rel(s[:cout], ref.sum((0, 2, 3))) < 1e-3 + TOL[dtype]
    assert rel(s[cout:2 * cout], (ref * ref).sum((0, 2, 3))) < 1e-3 + TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,hw,batch", [(48, 24, 8, 2), (128, 96, 10, 3), (256, 64, 7, 2)])
def test_conv1x1_lazy_prologue(dtype, cin, cout, hw, batch):
    # the last two shapes take the LDS-tiled path in bf16; images of 100 / 49 pixels straddle its 128-row tiles, so the
    # per-image channel multiplier changes inside a tile
    L, Act, Engine, ParamStore, Pro = _gpu()
    w = rand(cout, cin, 1, 1, seed=1, scale=cin ** -0.5)
    x = rand(batch, cin, hw, hw, seed=3, scale=3.0)
    sc, sh = rand(cin, seed=4).abs() + 0.5, rand(cin, seed=5)
    bs = (rand(batch, cin, seed=6) > 0).float() * 2.0
    eng = make_engine(Engine, ParamStore, [("w", w.shape)], dict(w=w), dtype)
    xa = to_act(Act, x, dtype).with_pro(Pro(sc.cuda(), sh.cuda(), L.ACT_RELU6, bs.cuda().contiguous()))
    ya = eng.new_act(batch, hw, hw, cout)
    eng.conv(xa, "w", ya)
    torch.cuda.synchronize()
    xt = torch.clamp(q(x, dtype) * sc[None, :, None, None] + sh[None, :, None, None], 0, 6) * bs[:, :, None, None]
    ref = F.conv2d(q(xt, dtype) if dtype == torch.bfloat16 else xt, q(w, dtype))
    assert rel(ya.nchw(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,hw,batch,taps,with_res,with_pro", [
    (32, 64, 12, 2, 1, False, False),       # expand conv: activated store
    (64, 32, 12, 2, 1, True, True),         # project conv: lazy input, BN + residual in the epilogue
    (256, 128, 9, 3, 1, True, False),       # LDS-tiled path (bf16), ragged M
    (128, 80, 7, 2, 1, True, True),         # LDS-tiled path, column tail
    (24, 16, 8, 2, 9, False, False),        # dense 3x3
])
def test_conv_eval_epilogue(dtype, cin, cout, hw, batch, taps, with_res, with_pro):
    """isa_conv_gemm_ep: y = act(scale * conv(pro(x)) + shift) (+ res) - eval-mode BatchNorm, activation and the block's
    residual add in the conv's output epilogue (Engine.conv_bn_eval), against torch."""
    L, Act, Engine, ParamStore, Pro = _gpu()
    k = 3 if taps == 9 else 1
    w = rand(cout, cin, k, k, seed=1, scale=(taps * cin) ** -0.5)
    gamma, beta = rand(cout, seed=2).abs() + 0.5, rand(cout, seed=3)
    rmean, rvar = rand(cout, seed=4) * 0.1, rand(cout, seed=5).abs() + 0.5
    x = rand(batch, cin, hw, hw + 1, seed=6, scale=2.0)
    res = rand(batch, cout, hw, hw + 1, seed=7)
    schema = [("w", w.shape), ("bn.weight", (cout,)), ("bn.bias", (cout,)), ("bn.running_mean", (cout,)),
              ("bn.running_var", (cout,)), ("bn.num_batches_tracked", ())]
    ps = ParamStore(schema, "cuda")
    ps.load_state_dict({"w": w, "bn.weight": gamma, "bn.bias": beta, "bn.running_mean": rmean, "bn.running_var": rvar})
    eng = Engine(ps, dtype)
    eng.begin(bn_train=False, record=False)
    assert eng.eval_fusable()
    xa = to_act(Act, x, dtype)
    xt = q(x, dtype)
    if with_pro:
        sc, sh = rand(cin, seed=8).abs() + 0.5, rand(cin, seed=9)
        xa = xa.with_pro(Pro(sc.cuda(), sh.cuda(), L.ACT_RELU6))
        xt = torch.clamp(xt * sc[None, :, None, None] + sh[None, :, None, None], 0, 6)
        if dtype == torch.bfloat16:
            xt = q(xt, dtype)
    ra = to_act(Act, res, dtype) if with_res else None
    act = L.ACT_NONE if with_res else L.ACT_RELU6
    ya = eng.new_act(batch, hw, hw + 1, cout, ld=(cout + 15) // 8 * 8)
    ya.buf.fill_(5.0)
    eng.conv_bn_eval(xa, "w", ya, "bn", act, res=ra, taps=taps)
    torch.cuda.synchronize()
    ref = F.conv2d(xt, q(w, dtype), padding=k // 2)
    inv = 1.0 / torch.sqrt(rvar + 1e-5)
    ref = ref * (gamma * inv)[None, :, None, None] + (beta - rmean * gamma * inv)[None, :, None, None]
    ref = torch.clamp(ref, 0, 6) if act == L.ACT_RELU6 else ref
    if with_res:
        ref = ref + q(res, dtype)
    assert rel(ya.nchw(), ref) < TOL[dtype]
    assert float((ya.buf[..., cout:].float() - 5.0).abs().max()) == 0.0     # padding channels untouched


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,hw", [(32, 16, 12), (16, 2, 16), (12, 1, 8), (256, 128, 4)])
def test_conv3x3_dense(dtype, cin, cout, hw):
    L, Act, Engine, ParamStore, Pro = _gpu()
    batch = 2
    w = rand(cout, cin, 3, 3, seed=1, scale=(9 * cin) ** -0.5)
    b = rand(cout, seed=2)
    x = rand(batch, cin, hw, hw + 1, seed=3)
    eng = make_engine(Engine, ParamStore, [("w", w.shape), ("b", b.shape)], dict(w=w, b=b), dtype)
    xa = to_act(Act, x, dtype)
    ya = eng.new_act(batch, hw, hw + 1, cout)
    eng.conv(xa, "w", ya, taps=9, bias="b")
    torch.cuda.synchronize()
    ref = F.conv2d(q(x, dtype), q(w, dtype), b, padding=1)
    assert rel(ya.nchw(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,hw", [(64, 32, 8), (512, 256, 4)])
def test_conv_transpose(dtype, cin, cout, hw):
    L, Act, Engine, ParamStore, Pro = _gpu()
    batch = 2
    w = rand(cin, cout, 2, 2, seed=1, scale=cin ** -0.5)
    b = rand(cout, seed=2)
    x = rand(batch, cin, hw, hw, seed=3)
    eng = make_engine(Engine, ParamStore, [("w", w.shape), ("b", b.shape)], dict(w=w, b=b), dtype)
    xa = to_act(Act, x, dtype)
    big = eng.new_act(batch, 2 * hw, 2 * hw, 2 * cout, ld=2 * cout)
    big.buf.fill_(3.0)
    ya = big.slice(cout, cout)                      # write into the second half of a concat buffer
    eng.conv(xa, "w", ya, bias="b", transposed=True)
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(q(x, dtype), q(w, dtype), b, stride=2)
    assert rel(ya.nchw(), ref) < TOL[dtype]
    assert float((big.buf[..., :cout].float() - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,h,w", [(32, 16, 16), (21, 9, 20), (48, 8, 8), (1024, 4, 4)])
def test_dwconv(dtype, c, h, w):
    L, Act, Engine, ParamStore, Pro = _gpu()
    batch = 2
    wt = rand(c, 1, 3, 3, seed=1, scale=1 / 3.0)
    b = rand(c, seed=2)
    x = rand(batch, c, h, w, seed=3)
    sc, sh = rand(c, seed=4).abs() + 0.5, rand(c, seed=5)
    eng = make_engine(Engine, ParamStore, [("w", wt.shape), ("b", b.shape)], dict(w=wt, b=b), dtype)
    xa = to_act(Act, x, dtype).with_pro(Pro(sc.cuda(), sh.cuda(), L.ACT_RELU6))
    ya = eng.new_act(batch, h, w, c)
    _, st = eng.dwconv(xa, "w", ya, bias="b", stats=True)
    torch.cuda.synchronize()
    xt = torch.clamp(q(x, dtype) * sc[None, :, None, None] + sh[None, :, None, None], 0, 6)
    ref = F.conv2d(xt, q(wt, dtype), b, padding=1, groups=c)
    assert rel(ya.nchw(), ref) < TOL[dtype]
    s = st.cpu().view(8, -1).sum(0)
    assert rel(s[:c], ref.sum((0, 2, 3))) < 1e-3 + TOL[dtype]
    assert rel(s[c:2 * c], (ref * ref).sum((0, 2, 3))) < 1e-3 + TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_finalize_and_materialize(dtype):
    L, Act, Engine, ParamStore, Pro = _gpu()
    c, batch, hw = 24, 2, 8
    x = rand(batch, c, hw, hw, seed=3, scale=2.0) + 0.5
    res = rand(batch, c, hw, hw, seed=4)
    g, b = rand(c, seed=5).abs() + 0.5, rand(c, seed=6)
    rm, rv = rand(c, seed=7) * 0.1, rand(c, seed=8).abs() + 0.5
    schema = [("bn.weight", (c,)), ("bn.bias", (c,)), ("bn.running_mean", (c,)), ("bn.running_var", (c,)),
              ("bn.num_batches_tracked", ())]
    for train in (True, False):
        eng = make_engine(Engine, ParamStore, schema,
                          {"bn.weight": g, "bn.bias": b, "bn.running_mean": rm, "bn.running_var": rv}, dtype)
        eng.bn_train = train
        xa = to_act(Act, x, dtype)
        xq = q(x, dtype)
        st = torch.zeros(8, 2 * c)
        st[0] = torch.cat([xq.sum((0, 2, 3)), (xq * xq).sum((0, 2, 3))])
        st = st.reshape(-1).cuda()
        out = eng.new_act(batch, hw, hw, c)
        eng.bn_out(xa, st, "bn", L.ACT_RELU6, out, res=to_act(Act, res, dtype))
        torch.cuda.synchronize()
        ref = F.batch_norm(xq, rm.clone(), rv.clone(), g, b, training=train, momentum=0.1, eps=1e-5)
        ref = torch.clamp(ref, 0, 6) + q(res, dtype)
        assert rel(out.nchw(), ref) < max(TOL[dtype], 1e-4)
        if train:
            rm2, rv2 = rm.clone(), rv.clone()
            F.batch_norm(xq, rm2, rv2, g, b, training=True, momentum=0.1, eps=1e-5)
            assert rel(eng.params.view("bn.running_mean"), rm2) < 1e-4
            assert rel(eng.params.view("bn.running_var"), rv2) < 1e-4
            assert eng.params.int_buffers["bn.num_batches_tracked"] == 1


@pytest.mark.parametrize("dtype", DTYPES)
def test_avgpool2_and_layout(dtype):
    L, Act, Engine, ParamStore, Pro = _gpu()
    from isa_amd.network import Network
    x = rand(2, 21, 8, 12, seed=1)
    eng = make_engine(Engine, ParamStore, [("dummy", (4,))], {}, dtype)
    net = Network(eng)
    xa = net.to_nhwc(x.cuda())
    torch.cuda.synchronize()
    assert xa.ld == 24 and rel(xa.nchw(), q(x, dtype)) < 1e-7
    back = net.to_nchw(xa)
    assert rel(back, q(x, dtype)) < 1e-7
    x2 = rand(2, 32, 8, 12, seed=2)
    a = to_act(Act, x2, dtype)
    out = eng.new_act(2, 4, 6, 32)
    eng.avgpool2(a, out)
    torch.cuda.synchronize()
    assert rel(out.nchw(), F.avg_pool2d(q(x2, dtype), 2)) < TOL[dtype]


def test_se_gate_and_argmax():
    L, Act, Engine, ParamStore, Pro = _gpu()
    dtype = torch.float32
    n, c, hw = 2, 32, 8
    x = rand(n, c, hw, hw, seed=1)
    w1, b1, w2, b2 = rand(16, c, seed=2) * 0.3, rand(16, seed=3), rand(c, 16, seed=4) * 0.3, rand(c, seed=5)
    eng = make_engine(Engine, ParamStore, [("d", (4,))], {}, dtype)
    xa = to_act(Act, x, dtype)
    mean = torch.zeros(n * c, device="cuda")
    L.check(eng.lib.isa_chan_mean(xa.d(), None, L.ptr(mean), eng.st()), "mean")
    gate, hid = torch.empty(n * c, device="cuda"), torch.empty(n * 16, device="cuda")
    d = [t.cuda().contiguous() for t in (w1, b1, w2, b2)]
    L.check(eng.lib.isa_se_fc(L.ptr(mean), L.ptr(d[0]), L.ptr(d[1]), L.ptr(d[2]), L.ptr(d[3]), n, c, 16,
                              L.ptr(hid), L.ptr(gate), eng.st()), "se")
    torch.cuda.synchronize()
    m = x.mean((2, 3))
    ref = torch.sigmoid(F.linear(F.relu(F.linear(m, w1, b1)), w2, b2))
    assert rel(mean.view(n, c), m) < 1e-5
    assert rel(gate.view(n, c), ref) < 1e-5
    lg = rand(2, 2, 8, 8, seed=9)
    la = to_act(Act, lg, dtype)
    out = eng.new_act(2, 8, 8, 1)
    L.check(eng.lib.isa_chan_argmax(la.d(), out.d(), eng.st()), "argmax")
    torch.cuda.synchronize()
    assert torch.equal(out.nchw().cpu()[:, 0], lg.argmax(1).float())


# ------------------------------------------------------------------------------------------------
# backward of the convolution family vs torch autograd (weight, bias and data gradients)
# ------------------------------------------------------------------------------------------------
BWD_TOL = {torch.float32: 1e-4, torch.bfloat16: 3e-2}


def _run_backward(eng, out_act, dy, Act):
    g = eng.grads.grad_of(out_act)
    g.buf[..., g.c0:g.c0 + g.c] = dy.permute(0, 2, 3, 1).to(g.buf.dtype).cuda()
    eng.grads.written[out_act.buf.data_ptr()].append((out_act.c0, out_act.c0 + out_act.c))
    eng.backward()
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("bscale", [False, True])      # per-image channel multiplier in the prologue: a Dropout2d mask
@pytest.mark.parametrize("kind,cin,cout,hw", [("1x1", 32, 32, 16), ("1x1", 64, 246, 8), ("1x1", 48, 24, 12),
                                              ("1x1", 256, 128, 8), ("3x3", 32, 16, 12), ("3x3", 16, 2, 16),
                                              ("T", 64, 32, 8), ("dw", 32, 32, 16), ("dw", 21, 21, 10)])
def test_conv_backward(dtype, kind, cin, cout, hw, bscale):
    L, Act, Engine, ParamStore, Pro = _gpu()
    batch = 2
    taps = 9 if kind == "3x3" else 1
    if kind == "T":
        w = rand(cin, cout, 2, 2, seed=1, scale=cin ** -0.5)
    elif kind == "dw":
        w = rand(cin, 1, 3, 3, seed=1, scale=1 / 3.0)
    else:
        k = 3 if kind == "3x3" else 1
        w = rand(cout, cin, k, k, seed=1, scale=(cin * k * k) ** -0.5)
    b = rand(cout, seed=2)
    x = rand(batch, cin, hw, hw + 2, seed=3)
    sc, sh = rand(cin, seed=4).abs() + 0.5, rand(cin, seed=5) + 1.0
    eng = make_engine(Engine, ParamStore, [("w", w.shape), ("b", b.shape)], dict(w=w, b=b), dtype)
    eng.begin(bn_train=True, record=True)
    # lazy input: act(scale*x+shift) with ReLU6, as after a BN; data gradient is w.r.t. that value
    # bscale: keep / (1 - p) per (image, channel), as F.dropout2d(p=.5) scales (utils.py:1104-1110); Act.grad is the
    # gradient w.r.t. the tensor's true value, i.e. after the multiplier
    bs = ((torch.rand(batch, cin, generator=torch.Generator().manual_seed(11)) < 0.5).float() * 2.0) if bscale else None
    xa = to_act(Act, x, dtype).with_pro(Pro(sc.cuda(), sh.cuda(), L.ACT_RELU6, bs.cuda() if bscale else None))
    oh, ow = (2 * hw, 2 * (hw + 2)) if kind == "T" else (hw, hw + 2)
    ya = eng.new_act(batch, oh, ow, cout)
    if kind == "dw":
        eng.dwconv(xa, "w", ya, bias="b")
    else:
        eng.conv(xa, "w", ya, taps=taps, bias="b", transposed=(kind == "T"))
    dy = rand(batch, cout, oh, ow, seed=7)
    _run_backward(eng, ya, dy, Act)
    # reference
    xt = torch.clamp(q(x, dtype) * sc[None, :, None, None] + sh[None, :, None, None], 0, 6)
    if bscale:
        xt = xt * bs[:, :, None, None]
    if dtype == torch.bfloat16 and kind != "dw":
        xt = q(xt, dtype)                       # the MFMA operand is rounded once to bf16
    xt.requires_grad_(True)
    wq = q(w, dtype).requires_grad_(True)
    bq = b.clone().requires_grad_(True)
    if kind == "T":
        yr = F.conv_transpose2d(xt, wq, bq, stride=2)
    elif kind == "dw":
        yr = F.conv2d(xt, wq, bq, padding=1, groups=cin)
    else:
        yr = F.conv2d(xt, wq, bq, padding=taps // 9)
    yr.backward(q(dy, dtype))
    tol = BWD_TOL[dtype]
    assert rel(eng.params.gview("w"), wq.grad) < tol, "dW"
    assert rel(eng.params.gview("b"), bq.grad) < tol, "db"
    dx = eng.grads.grad_of(xa)
    assert rel(dx.nchw(), xt.grad) < tol, "dX"


def test_slab_arena_abi_contract():
    """isa_slab_arena_* argument checks, the empty pass, and exhaustion as an error code (never a silent fallback);
    the numerics of deferred folds are covered by every backward test above (Engine.backward defers by default) and
    by tests/test_gpu_train.py."""
    import ctypes as C
    L, Act, Engine, ParamStore, Pro = _gpu()
    lib = L.lib()
    st = L.stream_ptr()
    n, used = C.c_int32(-1), C.c_int64(-1)
    h = C.c_void_p()
    small = torch.empty(1 << 20, dtype=torch.float32, device="cuda")
    assert lib.isa_slab_arena_create(L.ptr(small), small.numel(), C.byref(h)) != 0       # region below 16M floats
    assert lib.isa_slab_arena_create(None, 1 << 26, C.byref(h)) != 0
    assert lib.isa_slab_arena_flush(None, st, C.byref(n), C.byref(used)) != 0            # no handle
    arena = torch.empty(17 << 20, dtype=torch.float32, device="cuda")
    assert lib.isa_slab_arena_create(L.ptr(arena), arena.numel(), C.byref(h)) == 0 and h.value
    assert lib.isa_slab_arena_begin(h) == 0
    assert lib.isa_slab_arena_flush(h, st, C.byref(n), C.byref(used)) == 0
    assert (n.value, used.value) == (0, 0)
    assert lib.isa_slab_arena_destroy(h) == 0
    # an arena with less than 64 MB left refuses the call with ISA_ENOMEM; with ISA_DEFER_FOLD off the entry points
    # fold immediately from the caller's workspace (same result)
    w = rand(32, 32, 1, 1, seed=1)
    eng = make_engine(Engine, ParamStore, [("w", w.shape)], dict(w=w), torch.bfloat16)
    eng.fold_arena = torch.empty(16 << 20, dtype=torch.float32, device="cuda")     # exactly the minimum: usable once
    ref = None
    for rep in range(3):
        eng.begin(bn_train=True, record=True)
        eng.params.grad.zero_()
        xa = to_act(Act, rand(2, 32, 16, 16, seed=3), torch.bfloat16)
        y1, y2 = eng.new_act(2, 16, 16, 32), eng.new_act(2, 16, 16, 32)
        eng.conv(xa, "w", y1)
        if rep != 1:
            eng.conv(xa, "w", y2)
        for ya in ((y1, y2) if rep != 1 else (y1,)):
            g = eng.grads.grad_of(ya)
            g.buf.copy_(rand(2, 32, 16, 16, seed=5).permute(0, 2, 3, 1).to(g.buf.dtype))
            eng.grads.written[ya.buf.data_ptr()].append((0, 32))
        eng.defer_fold = rep != 0
        if rep == 2:                                   # two weight-gradient calls, room for one
            with pytest.raises(L.IsaError, match="ISA_ENOMEM"):
                eng.backward()
            continue
        eng.backward()
        torch.cuda.synchronize()
        if rep == 0:
            ref = eng.params.gview("w").clone()        # immediate folds, two convs
        else:
            assert eng.fold_stats[0] == 1, eng.fold_stats
            assert rel(2 * eng.params.gview("w"), ref) < 1e-5


def test_side_to_side_stream_edges_are_refused():
    """hipStreamEndCapture segfaults on ROCm 7.2 when a captured graph has side -> side edges (scripts/graph_probe.py;
    it killed a test process in round 2): Engine.sync turns the topology into an exception before anything is issued."""
    L, Act, Engine, ParamStore, Pro = _gpu()
    eng = make_engine(Engine, ParamStore, [("w", (8, 8, 1, 1))], dict(w=rand(8, 8, 1, 1)), torch.float32)
    eng.sync(0, 1); eng.sync(1, 0); eng.sync(2, 2)          # origin at one end / no edge at all: fine
    with pytest.raises(RuntimeError, match="side-to-side"):
        eng.sync(1, 2)


def test_uncaptured_arenas_are_bounded():
    """Every configuration key owns an arena (a full activation set); the un-captured ones are an LRU of ISA_ARENA_CACHE
    entries, frozen ones (a captured hipGraph points into them) are never dropped."""
    L, Act, Engine, ParamStore, Pro = _gpu()
    eng = make_engine(Engine, ParamStore, [("w", (8, 8, 1, 1))], dict(w=rand(8, 8, 1, 1)), torch.float32)
    eng.begin(False, False, key="graph")
    eng.new_act(1, 4, 4, 8)
    eng.freeze_arena()
    for i in range(12):
        eng.begin(False, False, key=("shape", i))
        eng.new_act(1, 4, 4, 8)
    keys = [k for k in eng.arenas if k is not None]
    assert (False, False, "graph") in keys and (False, False, ("shape", 11)) in keys
    assert len([k for k in keys if not eng.arenas[k].frozen]) <= eng.arena_cache + 1
    assert (False, False, ("shape", 0)) not in keys
