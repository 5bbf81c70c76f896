"""Data-parallel training across the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference has no distributed code at all.  What is exchanged, and what is not:
  * ONE sum-all-reduce per step of the trainable slice of the flat gradient buffer (4 779 392 fp32 = 19 MB,
    latency-bound over xGMI), issued after backward on the compute stream; the average (1/world) is applied inside
    the optimizer kernel, so the global-norm clip (model.py:275-277) sees the averaged gradient on every rank and all
    ranks take the same step.  Ranks may run different numbers of decoder iterations (data dependent): a post-backward
    flat reduce is insensitive to that.  The 9 never-trained tensors live outside the reduced slice.
  * BatchNorm batch statistics stay LOCAL (the reference has no SyncBN; parity is per rank at the local batch).
  * Policy for the state that drifts per rank - BN running statistics, maskBN buffers, the REINFORCE baseline
    (attenet2.py:266): they are local running estimates; `sync_buffers` averages them over the ranks (one all-reduce of
    the float-buffer slice of the flat store) before every validation pass / checkpoint, so that every rank validates
    the same model, the plateau scheduler (fed with the rank-averaged validation cost) takes the same decision
    everywhere, and rank 0's checkpoint is the model every rank holds.  Only rank 0 writes files.
backend 'nccl' is RCCL on ROCm; 'gloo' runs the same code on CPU tensors (tests/test_parallel_gloo.py).
"""
import os

import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------ process group
def env_world():
    """(world, rank, local_rank) from the torch.distributed.run environment; (1, 0, 0) without a launcher."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_from_env(backend=None):
    """Join the job described by WORLD_SIZE / RANK / LOCAL_RANK / MASTER_ADDR / MASTER_PORT.  Must run before anything
    else touches the GPU: it binds this process to its own device first.  Returns (world, rank, local_rank)."""
    world, rank, local_rank = env_world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver stack
        kw = dict(device_id=torch.device("cuda", local_rank)) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return world, rank, local_rank


def is_main(rank=None):
    return (env_world()[1] if rank is None else rank) == 0


# ------------------------------------------------------------------------------------------ the exchange step
def allreduce_flat_(flat_grad: torch.Tensor, n_train: int, world: int) -> float:
    """In-place sum over ranks of flat_grad[:n_train]; returns the scale (1/world) the optimizer
    applies, so clipping (model.py:275-277) sees the averaged gradient on every rank."""
    if world <= 1:
        return 1.0
    dist.all_reduce(flat_grad[:n_train], op=dist.ReduceOp.SUM)
    return 1.0 / world


def exchange_and_update(flat_grad, n_train, world, sqnorm_fn, update_fn, clip):
    """The ordering every rank follows after backward (Trainer.apply_update and the CPU test drive this same
    function): all-reduce (sum) -> squared norm of the AVERAGED gradient -> clip coefficient + Adadelta with the
    averaging scale folded in.  sqnorm_fn(grad, n, gscale) and update_fn(grad, n, gscale) are the two kernels
    (isa_sqnorm / isa_adadelta on the GPU; torch stand-ins on the CPU).  Returns gscale."""
    gscale = allreduce_flat_(flat_grad, n_train, world)
    if clip > 0:
        sqnorm_fn(flat_grad, n_train, gscale)
    update_fn(flat_grad, n_train, gscale)
    return gscale


def sync_buffers(store, baseline=None, world=None):
    """Average the float buffers (BN running statistics; layout of ParamStore.flat: [trainable | never-trained |
    float buffers]) and the REINFORCE baseline over the ranks.  See the module docstring for the policy."""
    world = (dist.get_world_size() if dist.is_initialized() else 1) if world is None else world
    if world <= 1:
        return
    buf = store.flat[store.buffer_start:store.total]
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    buf.mul_(1.0 / world)
    if baseline is not None:
        dist.all_reduce(baseline, op=dist.ReduceOp.SUM)
        baseline.mul_(1.0 / world)


def mean_over_ranks(value: float, world=None, device=None) -> float:
    """Rank-averaged scalar (validation cost): every rank feeds the plateau scheduler the same number."""
    world = (dist.get_world_size() if dist.is_initialized() else 1) if world is None else world
    if world <= 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / world


# ------------------------------------------------------------------------------------------ data sharding
def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous image shard of a global batch (weak scaling keeps per-rank batch fixed)."""
    per = global_batch // world
    assert per * world == global_batch, "global batch must divide by world size"
    return rank * per, (rank + 1) * per


def rank_seed(seed: int, rank: int) -> int:
    """Distinct, reproducible data / shuffling seed per rank (rank 0 keeps the single-process seed)."""
    return int(seed) + 1000003 * int(rank)


def clip_coef(sum_sq: float, max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ coefficient (host-side mirror of the device formula)."""
    if max_norm <= 0:
        return 1.0
    return min(1.0, max_norm / (sum_sq ** 0.5 + 1e-6))
