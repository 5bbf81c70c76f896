"""Row softmaxes of the attention head with many workgroups per row (isa_sp_softmax / isa_ins_softmax with a `part`
scratch: chunk scores + online-softmax partials, then the normalisation) against the one-workgroup-per-row kernels
(part = NULL) and against torch: SpatialAttentionLayer's masked softmax times the mask count (utils.py:505-512) and
HardAttentionLayer's softmax over the pixels of the selected instance (utils.py:648-655, attenet2.py:342-343).  Ragged
row lengths, an empty mask / an instance without pixels (all zeros out, as NaN -> 0 in the reference), several decoder
iterations over the same images (nsrc)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from test_gpu_ops import _gpu, rand  # noqa: E402


@pytest.mark.parametrize("n,c,L", [(3, 24, 65536), (2, 24, 72 * 72), (4, 16, 4096), (2, 24, 100)])
def test_sp_softmax_chunked(n, c, L):
    Lm, Act, Engine, ParamStore, Pro = _gpu()
    dev = "cuda"
    dot = rand(n, L, seed=1).to(dev)
    m = (torch.rand(n, L, generator=torch.Generator().manual_seed(2)) < 0.4).float().to(dev)
    m[-1] = 0.0                                                     # an image without foreground
    chansum, lh = rand(n, c, seed=3).to(dev) * 50.0, rand(c, seed=4).to(dev)
    fcw, fcb = torch.tensor([1.7], device=dev), torch.tensor([-0.3], device=dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for chunked in (True, False):
        beta = torch.full((n, L), 7.0, device=dev)
        rowstat = torch.zeros(n, 4, device=dev)
        part = torch.zeros(n * 64 * 4, device=dev) if chunked else None
        Lm.check(Lm.lib().isa_sp_softmax(Lm.ptr(dot), Lm.ptr(m), Lm.ptr(chansum), Lm.ptr(lh), Lm.ptr(fcw), Lm.ptr(fcb), n, c, L,
                                         Lm.ptr(beta), Lm.ptr(rowstat), Lm.ptr(part), st), "isa_sp_softmax")
        torch.cuda.synchronize()
        outs.append((beta.cpu(), rowstat.cpu()))
    (b1, r1), (b0, r0) = outs
    ht = (chansum * lh[None]).sum(1, keepdim=True) / L
    z = (fcw * torch.tanh(dot + ht) + fcb).masked_fill(m < 0.5, float("-inf"))
    ref = torch.nan_to_num(torch.softmax(z.double(), 1), nan=0.0) * m.sum(1, keepdim=True).double()
    for b in (b1, b0):
        assert float((b.double() - ref.cpu()).abs().max() / ref.abs().max()) < 2e-6
    assert float((b1 - b0).abs().max() / b0.abs().max()) < 1e-6
    assert torch.allclose(r1[:, (0, 2, 3)], r0[:, (0, 2, 3)], rtol=0, atol=0) or float((r1 - r0).abs().max()) < 1e-5
    assert float(((r1[:, 1] - r0[:, 1]).abs() / r0[:, 1].abs().clamp_min(1e-30)).max()) < 1e-5
    assert float(b1[-1].abs().max()) == 0.0


@pytest.mark.parametrize("nsrc,iters,nobj,L", [(3, 2, 5, 65536), (2, 1, 4, 72 * 72), (2, 3, 3, 1000)])
def test_ins_softmax_chunked(nsrc, iters, nobj, L):
    Lm, Act, Engine, ParamStore, Pro = _gpu()
    dev = "cuda"
    n = nsrc * iters
    merge = rand(nsrc, L, seed=1, scale=3.0).to(dev)
    owner = torch.randint(0, nobj, (nsrc, L), generator=torch.Generator().manual_seed(2))
    ins = torch.stack([(owner == k) for k in range(nobj - 1)] + [torch.zeros(nsrc, L, dtype=torch.bool)], 1).long().to(dev)
    idx = torch.randint(0, nobj, (n,), generator=torch.Generator().manual_seed(3)).int()
    idx[0] = nobj - 1                                               # an instance without pixels
    idx = idx.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for chunked in (True, False):
        alpha = torch.full((n, L), 7.0, device=dev)
        rowstat = torch.zeros(n, 2, device=dev)
        part = torch.zeros(n * 64 * 4, device=dev) if chunked else None
        Lm.check(Lm.lib().isa_ins_softmax(Lm.ptr(merge), Lm.ptr(ins), Lm.ptr(idx), n, nobj, L, Lm.ptr(alpha), Lm.ptr(rowstat),
                                          nsrc, Lm.ptr(part), st), "isa_ins_softmax")
        torch.cuda.synchronize()
        outs.append((alpha.cpu(), rowstat.cpu()))
    (a1, r1), (a0, r0) = outs
    rows = []
    for b in range(n):
        bi = b % nsrc
        z = merge[bi].double().masked_fill(ins[bi, idx[b]] == 0, float("-inf"))
        rows.append(torch.nan_to_num(torch.softmax(z, 0), nan=0.0))
    ref = torch.stack(rows).cpu()
    for a in (a1, a0):
        assert float((a.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert float((a1 - a0).abs().max() / a0.abs().max()) < 1e-6
    assert float(a1[0].abs().max()) == 0.0
    assert torch.equal(r1[:, 0], r0[:, 0])
