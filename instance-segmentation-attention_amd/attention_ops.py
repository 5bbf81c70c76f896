"""Host side of the reference's named attention operators (modules/utils.py:49-69,167-329; dead at HEAD, SURVEY §8
a19-a21): same class names, constructor arguments, parameter names (state_dict-compatible) and forward argument
meaning as the reference; all compute is the HIP library.  Eval-mode semantics (dropout off).

  ScaledDotProductAttention / scaled_dot_product_attention   utils.py:305-329   isa_sdp_attention (split-L streaming)
  MultiHeadAttention                                          utils.py:167-225   isa_linear_ln + isa_conv_gemm + isa_sdp_*
  _ScalePDAttention / ScalePDAttention                        utils.py:248-303   isa_conv_gemm + isa_local_attention +
                                                                                 isa_instance_norm_res
  point_query_mask (Decoder.forward)                          utils.py:59-69     isa_point_query
"""
import math

import torch
import torch.nn as nn

from . import lib as L
from .engine import Act, Engine, ParamStore

_WS = {}


def _ws(device, floats=1 << 18):
    t = _WS.get(device)
    if t is None or t.numel() < floats:
        t = _WS[device] = torch.empty(floats, dtype=torch.float32, device=device)
    return t


def scaled_dot_product_attention(q, k, v, temperature, mask=None, return_attn=True, heads=1, mask_per_head=False,
                                 tile_keys=0):
    """ScaledDotProductAttention.forward (utils.py:316-327), dropout off (eval).
    heads == 1: q[Bh,Lq,dk], k[Bh,Lk,dk], v[Bh,Lk,dv] (f32, bf16 or f16); mask[Bh,Lq,Lk] bool, True = masked.
    heads == G: the projected layout of MultiHeadAttention, q[B,Lq,G*d], k,v[B,Lk,G*d]; out[B,Lq,G*d]; attn rows are
    head-major [(G*B),Lq,Lk] like the reference's.  tile_keys: keys per LDS tile (0 = default)."""
    assert q.is_cuda and q.dtype == k.dtype == v.dtype and q.dtype in (torch.float32, torch.bfloat16, torch.float16)
    b, lq, dq = q.shape
    Lk = k.shape[1]
    dk, dv = dq // heads, v.shape[2] // heads
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    out = torch.empty(b, lq, heads * dv, dtype=q.dtype, device=q.device)
    attn = torch.empty(heads * b, lq, Lk, dtype=torch.float32, device=q.device) if return_attn else None
    m = mask.to(torch.uint8).contiguous() if mask is not None else None
    ws = _ws(q.device)
    L.check(L.lib().isa_sdp_attention(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(m), L.ptr(out), L.ptr(attn), b, lq, Lk, dk, dv,
                                      float(temperature), L.dtype_code(q.dtype), heads, 1 if mask_per_head else 0,
                                      L.ptr(ws), ws.numel(), int(tile_keys), L.stream_ptr()), "isa_sdp_attention")
    return out, attn


def scaled_dot_product_attention_backward(q, k, v, temperature, out, attn, d_out, heads=1):
    """Gradients of scaled_dot_product_attention w.r.t. q, k, v (what autograd gives for utils.py:316-327, dropout off)
    from the forward's own outputs `out` and `attn` (masked keys carry zero probability, so the mask is not needed again).
    Returns (dq fp32, dk, dv)."""
    b, lq, dq_ = q.shape
    Lk = k.shape[1]
    dk_, dv_ = dq_ // heads, v.shape[2] // heads
    q, k, v, out, d_out = (t.contiguous() for t in (q, k, v, out, d_out))
    dq = torch.zeros(b, lq, dq_, dtype=torch.float32, device=q.device)
    dk, dv = torch.empty_like(k), torch.empty_like(v)
    L.check(L.lib().isa_sdp_attention_bwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(attn.contiguous()), L.ptr(out), L.ptr(d_out),
                                          L.ptr(dq), L.ptr(dk), L.ptr(dv), b, lq, Lk, dk_, dv_, float(temperature),
                                          L.dtype_code(q.dtype), heads, L.stream_ptr()), "isa_sdp_attention_bwd")
    return dq, dk, dv


class ScaledDotProductAttention(nn.Module):
    """utils.py:305-329 (constructor signature kept; dropout is an eval-mode no-op)."""

    def __init__(self, temperature, attn_dropout=0.1):
        super().__init__()
        self.temperature = temperature

    def forward(self, q, k, v, mask=None, last=False):
        if last:
            b, lq, d = q.shape
            out = torch.empty(b, lq, k.shape[1], dtype=torch.float32, device=q.device)
            L.check(L.lib().isa_sdp_scores(L.ptr(q.contiguous()), L.ptr(k.contiguous()), L.ptr(out), b, 1, lq, k.shape[1], d,
                                           L.dtype_code(q.dtype), 0, L.stream_ptr()), "isa_sdp_scores")
            return out                           # raw q k^T (utils.py:310-313); MultiHeadAttention applies the sigmoid
        return scaled_dot_product_attention(q, k, v, self.temperature, mask)


def _linear_ln(x, w, bias, residual=None, gamma=None, beta=None, eps=1e-5):
    rows, k = x.shape
    n = w.shape[0]
    y = torch.empty(rows, n, dtype=torch.float32, device=x.device)
    L.check(L.lib().isa_linear_ln(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(residual), L.ptr(gamma), L.ptr(beta), float(eps),
                                  rows, k, n, L.ptr(y), L.stream_ptr()), "isa_linear_ln")
    return y


class _ConvHelper:
    """A private Engine over a handful of 1x1 weights: the K/V projections stream L = H*W rows through isa_conv_gemm
    (MFMA, bias fused) exactly like the backbone's 1x1 convolutions."""

    def __init__(self, schema, device, dtype):
        self.store = ParamStore([(n, tuple(s)) for n, s in schema], device)
        self.engine = Engine(self.store, dtype, device)

    def load(self, tensors):
        self.store.load_state_dict({k: v.detach().reshape(self.store.shapes[k]) for k, v in tensors.items()})
        if self.engine.packer.entries:
            self.engine.packer.pack()

    def conv(self, x: Act, wname, bias, cout):
        E = self.engine
        out = E.new_act(x.n, x.h, x.w, cout)
        E.conv(x, wname, out, bias=bias)
        return out


class MultiHeadAttention(nn.Module):
    """utils.py:167-225.  forward(q[b,lq,d_model], k[b,L,d_model], v[b,L,d_model], mask[b,lq,L] bool, last) ->
    (LayerNorm(fc(attention) + q), attn[(n_head*b),lq,L])  or, last=True, (sigmoid(q k^T)[(n_head*b),L], None).
    `dtype`: storage of the projected K/V stream (float32, bfloat16 or float16 - the fp16 attention path)."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1, dtype=torch.float32):
        super().__init__()
        self.n_head, self.d_k, self.d_v, self.d_model = n_head, d_k, d_v, d_model
        self.w_qs = nn.Linear(d_model, n_head * d_k)
        self.w_ks = nn.Linear(d_model, n_head * d_k)
        self.w_vs = nn.Linear(d_model, n_head * d_v)
        nn.init.normal_(self.w_qs.weight, mean=0, std=math.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_ks.weight, mean=0, std=math.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_vs.weight, mean=0, std=math.sqrt(2.0 / (d_model + d_v)))
        self.layer_norm = nn.LayerNorm(d_model)
        self.fc = nn.Linear(n_head * d_v, d_model)
        nn.init.xavier_normal_(self.fc.weight)
        self.storage = dtype
        self._helper = None

    def _project_kv(self, k, v):
        """w_ks / w_vs over the L rows: isa_conv_gemm on [b, L, 1, d_model] views; output [b, L, n_head*d] rows with the
        heads interleaved - the layout the streaming kernel reads, no head-major permute (utils.py:200-202)."""
        dev = k.device
        conv_dt = torch.float32 if self.storage == torch.float16 else self.storage     # conv_gemm stores f32 / bf16
        if self._helper is None or self._helper.engine.dtype != conv_dt:
            self._helper = _ConvHelper([("w_ks.weight", (self.n_head * self.d_k, self.d_model)),
                                        ("w_ks.bias", (self.n_head * self.d_k,)),
                                        ("w_vs.weight", (self.n_head * self.d_v, self.d_model)),
                                        ("w_vs.bias", (self.n_head * self.d_v,))], dev, conv_dt)
        H = self._helper
        H.engine.begin(False, False, key=("mha", tuple(k.shape)))
        H.load({"w_ks.weight": self.w_ks.weight, "w_ks.bias": self.w_ks.bias, "w_vs.weight": self.w_vs.weight,
                "w_vs.bias": self.w_vs.bias})
        b, Lk, dm = k.shape
        ka = Act(k.to(conv_dt).reshape(b, Lk, 1, dm).contiguous(), 0, dm, needs_grad=False)
        va = Act(v.to(conv_dt).reshape(b, Lk, 1, dm).contiguous(), 0, dm, needs_grad=False)
        kp = H.conv(ka, "w_ks.weight", "w_ks.bias", self.n_head * self.d_k)      # (a newly registered weight is packed
        vp = H.conv(va, "w_vs.weight", "w_vs.bias", self.n_head * self.d_v)      #  by the engine before its first launch)
        kt = kp.buf.reshape(b, Lk, -1)[..., :self.n_head * self.d_k]
        vt = vp.buf.reshape(b, Lk, -1)[..., :self.n_head * self.d_v]
        return kt.to(self.storage).contiguous(), vt.to(self.storage).contiguous()

    def forward(self, q, k, v, mask=None, last=False):
        assert q.is_cuda and q.dim() == 3
        b, lq, dm = q.shape
        Lk = k.shape[1]
        residual = q.float().reshape(b * lq, dm).contiguous()
        qp = _linear_ln(residual, self.w_qs.weight.detach().float().contiguous(), self.w_qs.bias.detach().float())
        qp = qp.reshape(b, lq, self.n_head * self.d_k).to(self.storage).contiguous()
        kp, vp = self._project_kv(k, v)
        if last:
            out = torch.empty(self.n_head * b, lq, Lk, dtype=torch.float32, device=q.device)
            L.check(L.lib().isa_sdp_scores(L.ptr(qp), L.ptr(kp), L.ptr(out), b, self.n_head, lq, Lk, self.d_k,
                                           L.dtype_code(self.storage), 1, L.stream_ptr()), "isa_sdp_scores")
            return out.squeeze(1), None
        o, attn = scaled_dot_product_attention(qp, kp, vp, math.sqrt(self.d_k), mask, heads=self.n_head)
        y = _linear_ln(o.float().reshape(b * lq, -1).contiguous(), self.fc.weight.detach().float().contiguous(),
                       self.fc.bias.detach().float(), residual, self.layer_norm.weight.detach().float(),
                       self.layer_norm.bias.detach().float(), self.layer_norm.eps)
        return y.reshape(b, lq, dm), attn


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


class ScalePDAttention(nn.Module):
    """_ScalePDAttention (utils.py:248-303): local 3x3-dilated attention with its projections.  forward(qk[b,d_model,h,w],
    v[b,d_model,h,w], nomask[b,1,h,w]) -> InstanceNorm2d(fc(attended) + qk).  The per-head 1x1 convolutions (the
    reference views the batch as b*n_head images of d_model/n_head channels) run as ONE block-diagonal 1x1 convolution
    over the d_model channels, so every head reads an aligned NHWC tensor."""

    def __init__(self, d_k, d_v, d_model, dilation_rate, n_head=2, dtype=torch.float32):
        super().__init__()
        self.qk_w = nn.Conv2d(d_model // n_head, 2 * d_k, 1)
        self.v_w = nn.Conv2d(d_model // n_head, d_v, 1)
        self.fc = nn.Conv2d(n_head * d_v, d_model, 1)
        nn.init.normal_(self.qk_w.weight, mean=0, std=math.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.v_w.weight, mean=0, std=math.sqrt(2.0 / (d_model + d_v)))
        nn.init.xavier_normal_(self.fc.weight)
        self.d, self.d_k, self.d_v, self.n_head, self.d_model = dilation_rate, d_k, d_v, n_head, d_model
        self.storage = dtype
        self._helper = None

    def forward(self, qk, v, nomask=None):
        dev, G, dk, dv, dm = qk.device, self.n_head, self.d_k, self.d_v, self.d_model
        b, _, h, w = qk.shape
        cin = dm // G
        if self._helper is None:
            self._helper = _ConvHelper([("qk.weight", (G * 2 * dk, dm)), ("qk.bias", (G * 2 * dk,)),
                                        ("v.weight", (G * dv, dm)), ("v.bias", (G * dv,)),
                                        ("fc.weight", (dm, G * dv)), ("fc.bias", (dm,))], dev, self.storage)
        H = self._helper
        # block-diagonal packing of the per-head weights (parameter plumbing on a few hundred floats)
        wqk = torch.zeros(G * 2 * dk, dm, device=dev)
        wv = torch.zeros(G * dv, dm, device=dev)
        for g in range(G):
            wqk[g * 2 * dk:(g + 1) * 2 * dk, g * cin:(g + 1) * cin] = self.qk_w.weight.detach().reshape(2 * dk, cin)
            wv[g * dv:(g + 1) * dv, g * cin:(g + 1) * cin] = self.v_w.weight.detach().reshape(dv, cin)
        H.engine.begin(False, False, key=("spd", tuple(qk.shape)))
        H.load({"qk.weight": wqk, "qk.bias": self.qk_w.bias.detach().repeat(G), "v.weight": wv,
                "v.bias": self.v_w.bias.detach().repeat(G), "fc.weight": self.fc.weight.detach().reshape(dm, G * dv),
                "fc.bias": self.fc.bias.detach()})
        xa, va = _nhwc(qk, self.storage), _nhwc(v, self.storage)
        xa.needs_grad = va.needs_grad = False
        QK = H.conv(xa, "qk.weight", "qk.bias", G * 2 * dk)
        V = H.conv(va, "v.weight", "v.bias", G * dv)
        att = H.engine.new_act(b, h, w, G * dv)
        # the reference repeats nomask along the (b*n_head) batch: image-head i uses nomask[i % b] (utils.py:271)
        nm = nomask.reshape(b, -1).float()
        for g in range(G):
            idx = torch.tensor([(bi * G + g) % b for bi in range(b)], device=dev)
            nmg = nm[idx].contiguous()
            L.check(L.lib().isa_local_attention(QK.slice(g * 2 * dk, dk).d(), QK.slice(g * 2 * dk + dk, dk).d(),
                                                V.slice(g * dv, dv).d(), L.ptr(nmg), att.slice(g * dv, dv).d(), int(self.d),
                                                L.stream_ptr()), "isa_local_attention")
        out = H.conv(att, "fc.weight", "fc.bias", dm)
        y = H.engine.new_act(b, h, w, dm)
        sums = torch.zeros(b * 2 * dm, dtype=torch.float32, device=dev)
        L.check(L.lib().isa_instance_norm_res(out.d(), xa.d(), y.d(), 1e-5, L.ptr(sums), L.stream_ptr()),
                "isa_instance_norm_res")
        return y.nchw()


def point_query_mask(q, enc, dtype=torch.float32):
    """Decoder.forward (utils.py:59-69): sigmoid(q[B,C] . enc[B,C,H*W]) -> [B, H*W]."""
    ea = _nhwc(enc, dtype)
    n, c, h, w = enc.shape
    out = torch.empty(n, h * w, dtype=torch.float32, device=enc.device)
    qq = q.float().contiguous()
    L.check(L.lib().isa_point_query(L.ptr(qq), ea.d(), L.ptr(out), L.stream_ptr()), "isa_point_query")
    return out
