"""Data-parallel exchange step: ONE sum-all-reduce of the flat gradient buffer, averaged
by 1/world inside the optimizer kernel.  The reference has no distributed code at all; this is the
only collective on the path (4.78 M fp32 elements = 19 MB, latency-bound over xGMI), issued once per
step after backward on the compute stream.  Ranks may run different numbers of decoder iterations
(data-dependent), which a post-backward flat reduce is insensitive to; the 9 never-trained tensors
live outside the reduced slice.  backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests."""
import torch
import torch.distributed as dist


def allreduce_flat_(flat_grad: torch.Tensor, n_train: int, world: int) -> float:
    """In-place sum over ranks of flat_grad[:n_train]; returns the scale (1/world) the optimizer
    applies, so clipping (model.py:275-277) sees the averaged gradient on every rank."""
    if world <= 1:
        return 1.0
    dist.all_reduce(flat_grad[:n_train], op=dist.ReduceOp.SUM)
    return 1.0 / world


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous image shard of a global batch (weak scaling keeps per-rank batch fixed)."""
    per = global_batch // world
    assert per * world == global_batch, "global batch must divide by world size"
    return rank * per, (rank + 1) * per


def clip_coef(sum_sq: float, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ coefficient (host-side mirror of the device formula)."""
    if max_norm <= 0:
        return 1.0
    return min(1.0, max_norm / (sum_sq ** 0.5 + 1e-6))
