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


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
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
