"""N>1 exchange step on CPU: world_size-2 gloo processes run the same flat all-reduce + averaging +
clip logic the GPU trainer uses (parallel.py), and must end with identical, correctly averaged grads."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import isa_amd  # noqa: F401
    from isa_amd.parallel import allreduce_flat_, shard_batch, clip_coef
    n_total, n_train = 1000, 900
    g = torch.full((n_total,), float(rank + 1))
    g[n_train:] = 123.0 + rank                     # never-grad / buffer region: must not be reduced
    scale = allreduce_flat_(g, n_train, world)
    lo, hi = shard_batch(64, rank, world)
    coef = clip_coef(float(((g[:n_train] * scale) ** 2).sum()), 10.0)
    torch.save(dict(g=g, scale=scale, shard=(lo, hi), coef=coef), os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_flat_allreduce_world2(tmp_path):
    world, port = 2, 29731
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % i)) for i in range(world)]
    for i in range(world):
        assert r[i]["scale"] == 0.5
        assert torch.allclose(r[i]["g"][:900], torch.full((900,), 3.0))           # 1 + 2, summed
        assert torch.allclose(r[i]["g"][900:], torch.full((100,), 123.0 + i))     # untouched
        assert r[i]["shard"] == (32 * i, 32 * (i + 1))
    assert r[0]["coef"] == r[1]["coef"]
    expect = min(1.0, 10.0 / ((900 * 1.5 ** 2) ** 0.5 + 1e-6))
    assert abs(r[0]["coef"] - expect) < 1e-9
