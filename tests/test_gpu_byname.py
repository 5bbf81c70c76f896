"""The reference's named attention operators (a19-a21) on the GPU vs vectors produced by the reference
classes themselves (tests/golden/byname_ops.npz, see oracle/gen_golden.py:byname_cases)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd import attention_ops as A
    return A


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
def test_scaled_dot_product_attention(dtype, tol):
    A = ops()
    z = np.load(os.path.join(ROOT, "tests", "golden", "byname_ops.npz"))
    t = lambda k: torch.from_numpy(z[k])
    q, k, v = (t("sdp/" + n).cuda().to(dtype) for n in ("q", "k", "v"))
    out, attn = A.scaled_dot_product_attention(q, k, v, float(np.sqrt(12.0)), t("sdp/mask").cuda())
    torch.cuda.synchronize()
    assert rel(out.float(), t("sdp/out")) < tol
    assert rel(attn, t("sdp/attn")) < tol
    assert abs(float(attn.sum(-1).mean()) - 1.0) < 1e-4
    # property at full size: one query against L = 65 536 keys; probabilities sum to 1, masked keys get 0
    L = 65536
    g = torch.Generator().manual_seed(1)
    q2, k2, v2 = torch.randn(2, 1, 12, generator=g), torch.randn(2, L, 12, generator=g), torch.randn(2, L, 12, generator=g)
    m2 = torch.rand(2, 1, L, generator=g) < 0.5
    o2, a2 = A.scaled_dot_product_attention(q2.cuda().to(dtype), k2.cuda().to(dtype), v2.cuda().to(dtype), 12 ** 0.5, m2.cuda())
    ref = torch.softmax((q2.to(dtype).float() @ k2.to(dtype).float().transpose(1, 2) / 12 ** 0.5).masked_fill(m2, float("-inf")), 2)
    assert rel(a2, ref) < max(tol, 1e-4)
    assert float(a2[m2.cuda()].abs().max()) == 0.0
    assert rel(o2.float(), ref @ v2.to(dtype).float()) < max(tol, 1e-4)


def test_local_dilated_attention_and_point_query():
    A = ops()
    z = np.load(os.path.join(ROOT, "tests", "golden", "byname_ops.npz"))
    t = lambda k: torch.from_numpy(z[k]).cuda()
    QK, V = t("local/QK"), t("local/V")
    nomask = t("local/nomask").repeat(2, 1, 1, 1)           # exactly what the reference does (utils.py:271)
    att = A.local_dilated_attention(QK[:, :12], QK[:, 12:], V, nomask, int(z["local/dilation"][0]))
    b2, dv, h, w = att.shape
    att = att.permute(0, 2, 3, 1).reshape(b2, h, w, dv).permute(0, 3, 1, 2).reshape(b2 // 2, 2 * dv, h, w)
    assert rel(att, t("local/att")) < 1e-5
    pq = A.point_query_mask(t("pq/q"), t("pq/enc"))
    assert rel(pq, t("pq/out")) < 1e-5


@pytest.mark.parametrize("tile", [256, 512, 1024])
def test_sdp_tile_sizes_and_ragged_lengths(tile):
    """Every LDS tile size of the sweep, lengths that are not multiples of the tile or of the split, several queries,
    fully masked rows (NaN like the reference's softmax of all -inf) - against torch on the same operands."""
    A = ops()
    g = torch.Generator().manual_seed(tile)
    for (bh, lq, L) in ((3, 1, 1), (2, 3, 1000), (5, 1, 4097), (1, 6, 70001)):
        q, k, v = torch.randn(bh, lq, 12, generator=g), torch.randn(bh, L, 12, generator=g), torch.randn(bh, L, 12, generator=g)
        mask = torch.rand(bh, lq, L, generator=g) < 0.3
        if L > 1:
            mask[0, 0] = True                                    # one fully masked row
        out, attn = A.scaled_dot_product_attention(q.cuda(), k.cuda(), v.cuda(), 12 ** 0.5, mask.cuda(), tile_keys=tile)
        ref_a = torch.softmax((q @ k.transpose(1, 2) / 12 ** 0.5).masked_fill(mask, float("-inf")), 2)
        ref_o = ref_a @ v
        ok = ~torch.isnan(ref_o)
        assert torch.equal(torch.isnan(out.cpu()), torch.isnan(ref_o)), (bh, lq, L)
        assert float((out.cpu()[ok] - ref_o[ok]).abs().max()) < 1e-5 * max(1.0, float(ref_o[ok].abs().max()))
        oka = ~torch.isnan(ref_a)
        assert float((attn.cpu()[oka] - ref_a[oka]).abs().max()) < 1e-6 + 1e-5 * float(ref_a[oka].max())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2), (torch.float16, 4e-3)])
def test_multi_head_attention_whole_operator(dtype, tol):
    """MultiHeadAttention (utils.py:167-225) with the reference's own parameters and vectors: projections, the
    streaming attention over the interleaved heads, fc + LayerNorm, and the last=True sigmoid branch."""
    A = ops()
    z = np.load(os.path.join(ROOT, "tests", "golden", "byname_ops.npz"))
    t = lambda k: torch.from_numpy(z[k])
    m = A.MultiHeadAttention(2, 24, 12, 12, dtype=dtype).cuda().eval()
    m.load_state_dict({k[len("mha/sd/"):]: t(k) for k in z.files if k.startswith("mha/sd/")})
    q, k, v, mask = t("mha/q").cuda(), t("mha/k").cuda(), t("mha/v").cuda(), t("mha/mask").cuda()
    out, attn = m(q, k, v, mask=mask)
    last, none = m(q, k, v, mask=mask, last=True)
    torch.cuda.synchronize()
    assert none is None and tuple(attn.shape) == tuple(z["mha/attn"].shape) and tuple(last.shape) == tuple(z["mha/last"].shape)
    assert rel(out, t("mha/out")) < tol
    assert rel(attn, t("mha/attn")) < tol
    assert rel(last, t("mha/last")) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4),
                                       # bf16 storage (local_attn_kernel<bf16_t>): inputs, the three projected tensors and the
                                       # attended map are each rounded once (2^-9 relative), the InstanceNorm output once more
                                       (torch.bfloat16, 3e-2)])
def test_scale_pd_attention_whole_operator(dtype, tol):
    """_ScalePDAttention (utils.py:248-303) with the reference's parameters: block-diagonal projections, the local
    3x3-dilated softmax per head (incl. the reference's nomask.repeat indexing), fc and InstanceNorm2d(out + qk)."""
    A = ops()
    z = np.load(os.path.join(ROOT, "tests", "golden", "byname_ops.npz"))
    t = lambda k: torch.from_numpy(z[k])
    m = A.ScalePDAttention(12, 12, 24, int(z["local/dilation"][0]), n_head=2, dtype=dtype).cuda().eval()
    m.load_state_dict({k[len("local/sd/"):]: t(k) for k in z.files if k.startswith("local/sd/")})
    out = m(t("local/qk_in").cuda(), t("local/v_in").cuda(), t("local/nomask").cuda())
    torch.cuda.synchronize()
    assert rel(out, t("local/out")) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize("b,heads,lq,L,d", [(3, 1, 1, 1000, 12), (2, 2, 3, 4097, 12), (1, 1, 6, 70001, 12), (2, 1, 2, 300, 32),
                                            (1, 1, 4, 513, 16)])
def test_scaled_dot_product_attention_backward(dtype, tol, b, heads, lq, L, d):
    """isa_sdp_attention_bwd against torch autograd (float64 on the CPU) of the reference's formula
    softmax(mask(q k^T / T)) v, from the forward pass's own outputs; masked keys, interleaved heads, ragged lengths."""
    A = ops()
    if d != 12 and dtype == torch.float16:
        pytest.skip("the forward pass streams f16 only at the reference's head width (12)")
    g = torch.Generator().manual_seed(L + lq)
    q = torch.randn(b, lq, heads * d, generator=g)
    k, v = torch.randn(b, L, heads * d, generator=g), torch.randn(b, L, heads * d, generator=g)
    # with 70 001 keys a probability is ~1e-5 and dK / dV land in f16's subnormal range: scale the incoming gradient the way
    # a loss scaler would (the backward pass is linear in it)
    d_out = torch.randn(b, lq, heads * d, generator=g) * (1024.0 if L > 10000 else 1.0)
    mask = torch.rand(b, lq, L, generator=g) < 0.3
    temp = d ** 0.5
    qd, kd, vd, dod = (t.cuda().to(dtype) for t in (q, k, v, d_out))
    out, attn = A.scaled_dot_product_attention(qd, kd, vd, temp, mask.cuda(), heads=heads)
    dq, dk, dv = A.scaled_dot_product_attention_backward(qd, kd, vd, temp, out, attn, dod, heads=heads)
    torch.cuda.synchronize()
    # reference: per head, float64 autograd on the storage-rounded operands
    q64, k64, v64 = (t.float().cpu().double().requires_grad_(True) for t in (qd, kd, vd))
    do64 = dod.float().cpu().double()
    outs = []
    for h in range(heads):
        sl = slice(h * d, (h + 1) * d)
        s = (q64[..., sl] @ k64[..., sl].transpose(1, 2) / temp).masked_fill(mask, float("-inf"))
        outs.append(torch.softmax(s, 2) @ v64[..., sl])
    (torch.cat(outs, 2) * do64).sum().backward()
    # the saved forward output enters through D = dO . out: over 70 001 keys `out` is a mean of ~49 000 values (|out| ~ 1e-2)
    # and its storage rounding (5e-3 relative to its largest element in f16 and bf16 alike, measured) bounds dK / dV there
    bound = max(tol, 1e-4) if (dtype == torch.float32 or L <= 10000) else max(tol, 1e-2)
    assert rel(dq, q64.grad) < bound
    assert rel(dk.float(), k64.grad) < bound
    assert rel(dv.float(), v64.grad) < bound
