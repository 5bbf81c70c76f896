"""BASELINE configs[3] on the one GPU a test box has: the 512x512 shapes of `train.py --size 512` (per-rank view),
checked against the CPU oracle run live, and the world > 1 branch of the graphed training step (forward + backward
replayed from a hipGraph, RCCL all-reduce + optimizer eager) driven through a real single-member RCCL group with
Trainer.world forced to 2 - the collective, the capture in thread_local mode and the 1/world scaling all execute;
only the second GPU is missing (the multi-rank arithmetic itself is covered on CPU: tests/test_parallel_gloo.py)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import reseg_ref as R       # noqa: E402
from test_oracle_golden import assert_index_map    # noqa: E402


def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.reseg import ReSeg
    from isa_amd.trainer import Trainer
    return ReSeg, Trainer


def test_512_inference_vs_oracle():
    ReSeg, _ = need_gpu()
    x, _, _, _ = R.synth_batch(2, 512, 512, seed=5)
    sd = R.synth_state_dict(23, False)
    m = ReSeg(2, False, dtype=torch.float32)
    m.load_state_dict(sd)
    m.eval()
    sem_out, sem_arg = m(False, x)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = R.reseg_forward(sd, x, use_instance_seg=False)
    got = sem_out.cpu()
    err = float((got - ref["sem_out"]).abs().max() / ref["sem_out"].abs().max())
    assert err < 1e-3, err
    margin = (ref["sem_out"][:, 1] - ref["sem_out"][:, 0]).abs().numpy()
    assert_index_map(ref["sem_argmax"][:, 0].numpy() != 0, sem_arg.cpu().numpy()[:, 0] != 0, margin,
                     float(ref["sem_out"].abs().max()), "sem_argmax 512", rel=1e-5)


def test_512_train_forward_scalars_vs_oracle_and_step_runs():
    """ReSeg.forward(True, ...) at 512x512 (the pyramid factors and position codes follow the input size, not the
    reference's hard-coded 256: config.py:1, utils.py:885) against the oracle's loss scalars; then one full training
    step (backward + clip + Adadelta) at that size: finite, parameters moved."""
    ReSeg, Trainer = need_gpu()
    x, sem, ins, n = R.synth_batch(2, 512, 512, seed=6)
    sd = R.synth_state_dict(23, True)
    sel = [list(reversed(range(int(k)))) for k in n.view(-1)]
    m = ReSeg(2, True, dtype=torch.float32)
    m.load_state_dict(sd)
    m.train()
    m.head.drop_rate = 0.0
    m.head.sample_in_training = False                     # greedy glimpse point: the oracle takes the argmax too
    out = m(True, x, sem, ins, n, selected_idx=sel)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = R.reseg_forward(sd, x, sem, ins, n, ctx=R.Ctx(bn_train=True, training=True, drop_rate=0.0),
                              state=R.HeadState(), selected_idx=sel, sample_fn=lambda a: a.argmax(1))
    for got, key in zip(out[3:], ("criterion", "ins_ce_loss", "ins_dice_loss")):
        r = float(ref[key])
        assert abs(float(got) - r) <= 1e-3 * max(1.0, abs(r)), (key, float(got), r)
    err = float((out[0].cpu() - ref["sem_out"]).abs().max() / ref["sem_out"].abs().max())
    assert err < 1e-3, err
    m.load_state_dict(sd)
    tr = Trainer(m)
    before = m.store.flat[:m.store.n_train].clone()
    tr.train_step(x, sem, ins, n, selected_idx=sel)
    torch.cuda.synchronize()
    assert torch.isfinite(m.store.grad).all() and torch.isfinite(m.store.flat).all()
    assert float((m.store.flat[:m.store.n_train] - before).abs().max()) > 0


def test_world2_branch_of_the_graphed_step():
    ReSeg, Trainer = need_gpu()
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x, sem, ins, n = R.synth_batch(2, 64, 64, seed=1)
        m = ReSeg(2, True, dtype=torch.float32)
        m.load_state_dict(R.synth_state_dict(23, True))
        m.train()
        m.head.drop_rate = 0.0
        m.head.sample_in_training = False
        tr = Trainer(m, world_size=2)                    # one member, averaging for two: gradients are halved
        order = [[0, 1], [1, 0]]
        inj = [torch.tensor([64 * 20 + 9, 64 * 41 + 30], dtype=torch.int32, device="cuda"),
               torch.tensor([64 * 12 + 50, 64 * 33 + 17], dtype=torch.int32, device="cuda")]
        snap = dict(flat=m.store.flat.clone(), ib=dict(m.store.int_buffers))

        def run(step):
            m.store.flat.copy_(snap["flat"]); tr.sq.zero_(); tr.acc.zero_(); m.store.int_buffers.update(snap["ib"])
            m.head.baseline = None
            m.mark_weights_dirty()
            step(x, sem, ins, n, selected_idx=order, injected_s_t=inj)
            torch.cuda.synchronize()
            return m.store.flat.clone(), m.store.grad.clone()

        p_e, g_e = run(tr.train_step)
        p_e2, g_e2 = run(tr.train_step)                  # the eager step against itself: float atomics reorder sums and a
        # ReLU6 / clamp threshold flip moves every upstream gradient (see test_gpu_train._check_gradients), so two runs
        # of the SAME path already differ; the graph path is held to a few times that, not to a fixed epsilon
        noise_cos = 1.0 - float(torch.nn.functional.cosine_similarity(g_e2.double(), g_e.double(), dim=0))
        noise_p = float((p_e2 - p_e).abs().max())
        run(tr.train_step_graphed)                       # first sight: eager
        p_c, g_c = run(tr.train_step_graphed)            # capture (forward + backward), update outside the graph
        p_r, g_r = run(tr.train_step_graphed)            # replay
        assert any(s.get("state") == "ready" for s in tr._graphs.values()), "graph was never captured"
        for name, p, g in (("capture", p_c, g_c), ("replay", p_r, g_r)):
            cos = float(torch.nn.functional.cosine_similarity(g.double(), g_e.double(), dim=0))
            assert 1.0 - cos <= max(5e-4, 5.0 * noise_cos), (name, cos, noise_cos)
            assert float((p - p_e).abs().max()) <= max(2e-2, 5.0 * noise_p), (name, noise_p)
        # the averaging scale reached the optimizer: a world-1 trainer from the same state takes a different step
        tr1 = Trainer(m, world_size=1)
        m.store.flat.copy_(snap["flat"]); m.store.int_buffers.update(snap["ib"]); m.head.baseline = None
        m.mark_weights_dirty()
        tr1.train_step(x, sem, ins, n, selected_idx=order, injected_s_t=inj)
        torch.cuda.synchronize()
        assert float((m.store.flat - p_e).abs().max()) > 2e-5      # Adadelta is nearly scale-invariant: eps and weight decay only
    finally:
        dist.destroy_process_group()
