"""isa_dwconv3x3_bn_backward / isa_conv1x1_bn_backward (fused BN-apply + conv dgrad/wgrad + next BN-reduce) against
(a) torch autograd in fp32 and (b) the separate kernels it replaces, on the two block shapes that use
it: InvertedResidual (pw-BN-ReLU6-dw-BN-ReLU6-pw, MobileNetDenseASPP.py:96-123) and
InvertedV1Residual (dw-BN-ReLU6-pw, MobileNetDenseASPP.py:68-93)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from test_gpu_ops import _gpu, _run_backward, q, rand, rel, to_act  # noqa: E402


def _build(kind, cin, chid, cout, dtype, x, dy, fuse, tensors):
    L, Act, Engine, ParamStore, Pro = _gpu()
    schema = [(k, tuple(v.shape)) for k, v in tensors.items()]
    schema += [(k.replace("running_mean", "num_batches_tracked"), ()) for k in tensors if k.endswith("running_mean")]
    ps = ParamStore(schema, "cuda")
    ps.load_state_dict(tensors)
    eng = Engine(ps, dtype)
    eng.fuse_dw_bn = fuse
    eng.fuse_pw_bn = fuse
    eng.profile = True
    eng.begin(bn_train=True, record=True)
    xa = to_act(Act, x, dtype)
    n, _, h, w = x.shape
    cur = xa
    if kind == "ir":
        e = eng.new_act(n, h, w, chid)
        _, s = eng.conv(cur, "pw1.weight", e, stats=True)
        cur = eng.bn(e, s, "bn1", L.ACT_RELU6)
    d = eng.new_act(n, h, w, chid)
    _, s = eng.dwconv(cur, "dw.weight", d, stats=True)
    cur = eng.bn(d, s, "bn2", L.ACT_RELU6)
    y3 = eng.new_act(n, h, w, cout)
    _, s = eng.conv(cur, "pw2.weight", y3, stats=True)
    out = eng.new_act(n, h, w, cout)
    eng.bn_out(y3, s, "bn3", L.ACT_NONE, out, res=xa if cin == cout else None)
    _run_backward(eng, out, dy, Act)
    calls = eng.profile_summary()
    grads = {k: eng.params.gview(k).clone() for k in tensors if "running" not in k}
    return grads, eng.grads.grad_of(xa).nchw().float().cpu(), calls


def _reference(kind, dtype, x, dy, t):
    xt = q(x, dtype).requires_grad_(True)
    P = {k: (q(v, dtype) if v.dim() == 4 else v.clone()).requires_grad_(True) for k, v in t.items() if "running" not in k}
    cur = xt
    if kind == "ir":
        cur = F.conv2d(cur, P["pw1.weight"])
        cur = F.relu6(F.batch_norm(cur, None, None, P["bn1.weight"], P["bn1.bias"], True, 0.1, 1e-5))
    cur = F.conv2d(cur, P["dw.weight"], padding=1, groups=cur.shape[1])
    cur = F.relu6(F.batch_norm(cur, None, None, P["bn2.weight"], P["bn2.bias"], True, 0.1, 1e-5))
    out = F.conv2d(cur, P["pw2.weight"])
    out = F.batch_norm(out, None, None, P["bn3.weight"], P["bn3.bias"], True, 0.1, 1e-5)
    if out.shape[1] == xt.shape[1]:
        out = out + xt
    out.backward(dy)
    return {k: v.grad for k, v in P.items()}, xt.grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,cin,chid,cout,h,w", [("ir", 16, 96, 16, 12, 20), ("ir", 32, 48, 24, 9, 37),
                                                     ("ir", 32, 64, 32, 16, 16), ("ir", 64, 64, 64, 8, 24),
                                                     ("v1", 64, 64, 32, 16, 16), ("v1", 40, 40, 16, 7, 33),
                                                     ("v1", 32, 32, 32, 12, 20)])      # residual gradient rides along
def test_fused_dw_bn_backward(dtype, kind, cin, chid, cout, h, w):
    n = 3
    t = {}
    if kind == "ir":
        t["pw1.weight"] = rand(chid, cin, 1, 1, seed=1, scale=cin ** -0.5)
        t.update({"bn1.weight": rand(chid, seed=2).abs() + 0.5, "bn1.bias": rand(chid, seed=3) * 0.5 + 1.0,
                  "bn1.running_mean": torch.zeros(chid), "bn1.running_var": torch.ones(chid)})
    t["dw.weight"] = rand(chid, 1, 3, 3, seed=4, scale=1 / 3.0)
    t.update({"bn2.weight": rand(chid, seed=5).abs() + 0.5, "bn2.bias": rand(chid, seed=6) * 0.5 + 1.0,
              "bn2.running_mean": torch.zeros(chid), "bn2.running_var": torch.ones(chid)})
    t["pw2.weight"] = rand(cout, chid, 1, 1, seed=7, scale=chid ** -0.5)
    t.update({"bn3.weight": rand(cout, seed=10).abs() + 0.5, "bn3.bias": rand(cout, seed=11) * 0.5,
              "bn3.running_mean": torch.zeros(cout), "bn3.running_var": torch.ones(cout)})
    x = rand(n, cin, h, w, seed=8) + (0.5 if kind == "v1" else 0.0)
    dy = rand(n, cout, h, w, seed=9)

    g_f, dx_f, calls_f = _build(kind, cin, chid, cout, dtype, x, dy, True, t)
    g_u, dx_u, calls_u = _build(kind, cin, chid, cout, dtype, x, dy, False, t)
    assert "isa_dwconv3x3_bn_backward" in calls_f and "isa_dwconv3x3_wgrad" not in calls_f
    assert "isa_dwconv3x3_bn_backward" not in calls_u and "isa_conv1x1_bn_backward" not in calls_u
    ncalls = lambda calls, name: calls.get(name, (0,))[0]
    if dtype == torch.bfloat16 and chid <= 64:
        # project conv (+ the expand conv of an InvertedResidual) take the fused 1x1 path: no apply pass is left,
        # and the only stand-alone reduce is the block's last BN
        assert ncalls(calls_f, "isa_conv1x1_bn_backward") == (2 if kind == "ir" else 1)
        assert ncalls(calls_f, "isa_bn_bwd_apply") == 0 and ncalls(calls_f, "isa_bn_bwd_reduce") == 1
        assert ncalls(calls_f, "isa_conv_wgrad") == 0
    else:
        assert ncalls(calls_f, "isa_conv1x1_bn_backward") == 0
        assert ncalls(calls_f, "isa_bn_bwd_apply") == ncalls(calls_u, "isa_bn_bwd_apply") - 1

    # (b) same arithmetic as the separate kernels: differences are summation order only
    tol_same = 2e-5 if dtype == torch.float32 else 1.5e-2
    for k in g_u:
        assert rel(g_f[k], g_u[k]) < tol_same, "fused vs separate: " + k
    assert rel(dx_f, dx_u) < tol_same, "fused vs separate: dx"

    # (a) torch autograd
    if dtype == torch.float32:
        ref, dx_ref = _reference(kind, dtype, x, dy, t)
        for k, v in ref.items():
            assert rel(g_f[k].view(-1), v.reshape(-1)) < 2e-4, "vs torch: " + k
        assert rel(dx_f, dx_ref) < 2e-4, "vs torch: dx"
