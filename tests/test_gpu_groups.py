"""Statistic groups (isa_tensor.groups): a batch of G*B images whose G consecutive groups keep separate train-mode
BatchNorm statistics - how the decoder iterations of attenet2.py:384-399 (same weights, same backbone features,
different glimpse / dropout) run as ONE pass - must give exactly what G separate passes over B images give:
  * forward values and data gradients of group g == those of the separate pass g,
  * parameter gradients == the sum over the passes,
  * BatchNorm running statistics and num_batches_tracked == the passes applied one after the other.
Covers every group-aware entry point through InvertedResidual blocks (MobileNetDenseASPP.py:96-123) in the shapes
that select the streaming / LDS-tiled GEMM, the fused and the separate backward kernels, fp32 and bf16 storage, plus
the broadcast materialising pass (one evaluation of the iteration-independent `cross` block, G Dropout2d masks,
utils.py:984) and its gradient fold."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from test_gpu_ops import _gpu, _run_backward, rand, rel, to_act  # noqa: E402


def _params(cin, chid, cout):
    t = {"b.conv.0.weight": rand(chid, cin, 1, 1, seed=1, scale=cin ** -0.5),
         "b.conv.3.weight": rand(chid, 1, 3, 3, seed=4, scale=1 / 3.0),
         "b.conv.6.weight": rand(cout, chid, 1, 1, seed=7, scale=chid ** -0.5)}
    for i, (k, c) in enumerate((("b.conv.1", chid), ("b.conv.4", chid), ("b.conv.7", cout))):
        t[k + ".weight"] = rand(c, seed=20 + i).abs() + 0.5
        t[k + ".bias"] = rand(c, seed=30 + i) * 0.5 + (1.0 if i < 2 else 0.0)
        t[k + ".running_mean"] = rand(c, seed=40 + i) * 0.1
        t[k + ".running_var"] = rand(c, seed=50 + i).abs() + 0.5
    return t


def _engine(tensors, dtype, fuse):
    L, Act, Engine, ParamStore, Pro = _gpu()
    from isa_amd.network import Network
    schema = [(k, tuple(v.shape)) for k, v in tensors.items()]
    schema += [(k.replace("running_mean", "num_batches_tracked"), ()) for k in tensors if k.endswith("running_mean")]
    ps = ParamStore(schema, "cuda")
    ps.load_state_dict(tensors)
    eng = Engine(ps, dtype)
    eng.fuse_dw_bn = fuse
    eng.fuse_pw_bn = fuse
    eng.profile = True
    return eng, Network(eng, use_instance_seg=False), Act


def _run(tensors, dtype, fuse, x, dy, groups, oscale=None, bcast_from=None):
    """One InvertedResidual over x (NCHW cpu) with `groups` statistic groups; returns out, dx, grads, buffers."""
    eng, net, Act = _engine(tensors, dtype, fuse)
    eng.begin(bn_train=True, record=True)
    n, cin, h, w = x.shape
    cout = tensors["b.conv.6.weight"].shape[0]
    xa = to_act(Act, x, dtype)
    if bcast_from is None:
        xa = Act(xa.buf, 0, cin, groups=groups)
        out = eng.new_act(n, h, w, cout, groups=groups)
        net.block_ir(xa, "b", out, oscale=None if oscale is None else oscale.cuda())
    else:                                                   # one evaluation, G masked copies
        G = bcast_from
        out = eng.new_act(G * n, h, w, cout, groups=G)
        with eng.repeated(G):
            net.block_ir(xa, "b", out, oscale=oscale.cuda())
    _run_backward(eng, out, dy, Act)
    calls = eng.profile_summary()
    torch.cuda.synchronize()
    grads = {k: eng.params.gview(k).clone().cpu() for k in tensors if "running" not in k}
    bufs = {k: eng.params.view(k).clone().cpu() for k in tensors if "running" in k}
    nbt = dict(eng.params.int_buffers)
    return out.nchw().cpu(), eng.grads.grad_of(xa).nchw().float().cpu(), grads, bufs, nbt, calls


SHAPES = [(32, 64, 32, 12, 20, 3),      # <= 64 channels: fused 1x1 backward (bf16), streaming GEMM, residual
          (64, 128, 32, 8, 12, 3),      # K = 128: LDS-tiled GEMM for the project conv, ragged M (288 px per group)
          (256, 512, 128, 4, 4, 2),     # low-resolution level shape: 32 px per group, wide BN backward (chunked walk)
          (24, 48, 40, 9, 37, 2)]       # channel tails


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("cin,chid,cout,h,w,B", SHAPES)
def test_grouped_block_equals_separate_passes(dtype, fuse, cin, chid, cout, h, w, B):
    G = 2
    t = _params(cin, chid, cout)
    xs = [rand(B, cin, h, w, seed=60 + g) * (1.0 + 0.5 * g) + 0.3 * g for g in range(G)]    # different statistics per group
    dys = [rand(B, cout, h, w, seed=70 + g) for g in range(G)]
    masks = [(torch.rand(B, cout, generator=torch.Generator().manual_seed(80 + g)) < 0.5).float() * 2.0 for g in range(G)]
    out_g, dx_g, grads_g, bufs_g, nbt_g, calls = _run(t, dtype, fuse, torch.cat(xs), torch.cat(dys), G, oscale=torch.cat(masks))
    # the separate passes, running statistics carried from one to the next
    tt = dict(t)
    outs, dxs, gsum = [], [], None
    for g in range(G):
        o, dx, gr, bufs, nbt, _ = _run(tt, dtype, fuse, xs[g], dys[g], 1, oscale=masks[g])
        outs.append(o); dxs.append(dx)
        gsum = gr if gsum is None else {k: gsum[k] + gr[k] for k in gr}
        tt = dict(tt); tt.update(bufs)
    tol = 1e-5 if dtype == torch.float32 else 1e-2      # same kernels, same arithmetic: only the summation order of atomics
    assert rel(out_g, torch.cat(outs)) < tol
    assert rel(dx_g, torch.cat(dxs)) < tol
    for k in gsum:
        assert rel(grads_g[k].view(-1), gsum[k].view(-1)) < (2e-5 if dtype == torch.float32 else 1.5e-2), k
    for k in bufs:
        assert rel(bufs_g[k], tt[k]) < 1e-5, k
    assert all(v == G for v in nbt_g.values())
    if dtype == torch.bfloat16 and fuse and chid <= 64:
        assert "isa_conv1x1_bn_backward" in calls and "isa_dwconv3x3_bn_backward" in calls


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,chid,cout,h,w,B", [(32, 64, 32, 12, 20, 3), (64, 128, 32, 8, 12, 2)])
def test_broadcast_block_equals_separate_passes(dtype, cin, chid, cout, h, w, B):
    """The `cross` block ahead of its Dropout2d: one evaluation feeds G masked copies; gradients fold back."""
    G = 2
    t = _params(cin, chid, cout)
    x = rand(B, cin, h, w, seed=61)
    dys = [rand(B, cout, h, w, seed=70 + g) for g in range(G)]
    masks = [(torch.rand(B, cout, generator=torch.Generator().manual_seed(80 + g)) < 0.5).float() * 2.0 for g in range(G)]
    out_b, dx_b, grads_b, bufs_b, nbt_b, _ = _run(t, dtype, True, x, torch.cat(dys), 1, oscale=torch.cat(masks), bcast_from=G)
    tt = dict(t)
    outs, dxsum, gsum = [], None, None
    for g in range(G):
        o, dx, gr, bufs, nbt, _ = _run(tt, dtype, True, x, dys[g], 1, oscale=masks[g])
        outs.append(o)
        dxsum = dx if dxsum is None else dxsum + dx
        gsum = gr if gsum is None else {k: gsum[k] + gr[k] for k in gr}
        tt = dict(tt); tt.update(bufs)
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2
    assert rel(out_b, torch.cat(outs)) < tol
    assert rel(dx_b, dxsum) < (2e-5 if dtype == torch.float32 else 2e-2)
    for k in gsum:
        assert rel(grads_b[k].view(-1), gsum[k].view(-1)) < (2e-5 if dtype == torch.float32 else 2e-2), k
    for k in bufs:
        assert rel(bufs_b[k], tt[k]) < 1e-5, k
    assert all(v == G for v in nbt_b.values())
