"""Host wrappers of the reference's named attention operators (modules/utils.py:59-69,248-329).
Same argument meaning as the reference classes; compute is the HIP library (a19-a21 in SURVEY.md §8)."""
import torch

from . import lib as L
from .engine import Act


def scaled_dot_product_attention(q, k, v, temperature, mask=None, return_attn=True):
    """ScaledDotProductAttention.forward (utils.py:316-327), dropout off (eval).
    q[Bh,Lq,dk], k[Bh,Lk,dk], v[Bh,Lk,dv] on the GPU (f32 or bf16); mask[Bh,Lq,Lk] bool, True = masked."""
    assert q.is_cuda and q.dtype == k.dtype == v.dtype and q.dtype in (torch.float32, torch.bfloat16)
    bh, lq, dk = q.shape
    Lk, dv = k.shape[1], v.shape[2]
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    out = torch.empty(bh, lq, dv, dtype=q.dtype, device=q.device)
    attn = torch.empty(bh, lq, Lk, dtype=torch.float32, device=q.device) if return_attn else None
    m = mask.to(torch.uint8).contiguous() if mask is not None else None
    L.check(L.lib().isa_sdp_attention(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(m), L.ptr(out), L.ptr(attn), bh, lq, Lk, dk, dv,
                                      float(temperature), L.dtype_code(q.dtype), L.stream_ptr()), "isa_sdp_attention")
    return out, attn


def _nhwc(t, dtype):
    return Act(t.permute(0, 2, 3, 1).contiguous().to(dtype), 0, t.shape[1]) if t.shape[1] % 8 == 0 else \
        _pad_nhwc(t, dtype)


def _pad_nhwc(t, dtype):
    n, c, h, w = t.shape
    ld = (c + 7) // 8 * 8
    buf = torch.zeros(n, h, w, ld, dtype=dtype, device=t.device)
    buf[..., :c] = t.permute(0, 2, 3, 1).to(dtype)
    return Act(buf, 0, c)


def local_dilated_attention(Q, K, V, nomask, dilation, dtype=torch.float32):
    """Core of _ScalePDAttention.forward (utils.py:276-299) after the 1x1 projections.  NCHW in/out."""
    qa, ka, va = _nhwc(Q, dtype), _nhwc(K, dtype), _nhwc(V, dtype)
    n, dv, h, w = V.shape
    out = Act(torch.empty(n, h, w, (dv + 7) // 8 * 8, dtype=dtype, device=V.device), 0, dv)
    nm = nomask.reshape(n, -1).float().contiguous()
    L.check(L.lib().isa_local_attention(qa.d(), ka.d(), va.d(), L.ptr(nm), out.d(), int(dilation), L.stream_ptr()),
            "isa_local_attention")
    return out.nchw()


def point_query_mask(q, enc, dtype=torch.float32):
    """Decoder.forward (utils.py:59-69): sigmoid(q[B,C] . enc[B,C,H*W]) -> [B, H*W]."""
    ea = _nhwc(enc, dtype)
    n, c, h, w = enc.shape
    out = torch.empty(n, h * w, dtype=torch.float32, device=enc.device)
    qq = q.float().contiguous()
    L.check(L.lib().isa_point_query(L.ptr(qq), ea.d(), L.ptr(out), L.stream_ptr()), "isa_point_query")
    return out
