"""N>1 path on CPU: world_size-2 gloo processes drive the SAME exchange code the GPU trainer uses
(parallel.exchange_and_update: all-reduce -> norm of the averaged gradient -> clip + Adadelta with the averaging
scale folded in), with torch stand-ins for the two HIP kernels (isa_sqnorm / isa_adadelta formulas), plus the buffer
policy (sync_buffers), the rank-averaged validation cost and the per-rank data seeds.  After two steps both ranks
must hold identical parameters, equal to a single-process run on the averaged gradients through
torch.optim.Adadelta + clip_grad_norm_ (model.py:145-166,273-278)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]

N_TOTAL, N_TRAIN, N_BUF0 = 1000, 800, 900        # flat layout: [trainable | never-trained | float buffers]
LR, RHO, EPS, WD, CLIP = 1.0, 0.9, 1e-6, 1e-3, 10.0


def _grad(rank, step):
    g = torch.Generator().manual_seed(100 * step + rank)
    return torch.randn(N_TOTAL, generator=g) * (3.0 + rank)      # large enough for the clip to bite


class _Store:                                     # the three attributes of ParamStore that sync_buffers reads
    def __init__(self, flat):
        self.flat, self.buffer_start, self.total = flat, N_BUF0, N_TOTAL


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    import isa_amd  # noqa: F401
    from isa_amd import parallel as P
    assert P.init_from_env("gloo") == (world, rank, rank) and dist.is_initialized()
    flat = torch.linspace(-1, 1, N_TOTAL).clone()
    sq, acc = torch.zeros(N_TRAIN), torch.zeros(N_TRAIN)
    state = dict(sqnorm=torch.zeros(1))

    def sqnorm_fn(grad, n, gscale):               # isa_sqnorm: out += sum (g * scale)^2
        state["sqnorm"] = ((grad[:n] * gscale) ** 2).sum().reshape(1)

    def update_fn(grad, n, gscale):               # isa_adadelta: clip + weight decay + Adadelta on the trained slice
        clip = min(1.0, CLIP / (float(state["sqnorm"].sqrt()) + 1e-6))
        w = flat[:n]
        g = grad[:n] * gscale * clip + WD * w
        sq.mul_(RHO).addcmul_(g, g, value=1 - RHO)
        delta = (acc + EPS).sqrt() / (sq + EPS).sqrt() * g
        acc.mul_(RHO).addcmul_(delta, delta, value=1 - RHO)
        w.sub_(LR * delta)

    for step in range(2):
        grad = _grad(rank, step)
        grad[N_TRAIN:] = 123.0 + rank             # never-trained / buffer region: must not be reduced
        scale = P.exchange_and_update(grad, N_TRAIN, world, sqnorm_fn, update_fn, CLIP)
        assert scale == 0.5 and torch.all(grad[N_TRAIN:] == 123.0 + rank)
        flat[N_BUF0:] += rank + 1.0               # "running statistics" drift per rank
    baseline = torch.tensor([float(rank)])
    P.sync_buffers(_Store(flat), baseline, world)
    val = P.mean_over_ranks(10.0 + rank, world, device="cpu")
    torch.save(dict(flat=flat, baseline=baseline, val=val, shard=P.shard_batch(64, rank, world),
                    seed=P.rank_seed(23, rank), main=P.is_main()), os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_exchange_and_update_world2(tmp_path):
    world, port = 2, 29731
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(world)]
    # both ranks took the same steps
    assert torch.equal(r[0]["flat"], r[1]["flat"])
    # ... and they are the steps of one process on the averaged gradient (torch's own optimizer + clip)
    p = torch.nn.Parameter(torch.linspace(-1, 1, N_TOTAL)[:N_TRAIN].clone())
    opt = torch.optim.Adadelta([p], lr=LR, rho=RHO, eps=EPS, weight_decay=WD)
    for step in range(2):
        p.grad = (0.5 * (_grad(0, step) + _grad(1, step)))[:N_TRAIN].clone()
        torch.nn.utils.clip_grad_norm_([p], CLIP)
        opt.step()
    assert float((r[0]["flat"][:N_TRAIN] - p.detach()).abs().max()) < 1e-6
    # never-trained slice untouched; buffers and the baseline averaged over the ranks (2 steps x (1 + 2) / 2 = 3)
    base = torch.linspace(-1, 1, N_TOTAL)
    assert torch.equal(r[0]["flat"][N_TRAIN:N_BUF0], base[N_TRAIN:N_BUF0])
    assert torch.allclose(r[0]["flat"][N_BUF0:], base[N_BUF0:] + 3.0)
    assert float(r[0]["baseline"]) == float(r[1]["baseline"]) == 0.5
    assert r[0]["val"] == r[1]["val"] == 10.5
    assert [r[i]["shard"] for i in range(world)] == [(0, 32), (32, 64)]
    assert r[0]["seed"] == 23 and r[1]["seed"] != 23 and [r[i]["main"] for i in range(world)] == [True, False]
