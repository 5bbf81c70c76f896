"""GPU bring-up: full train-step gradients vs oracle autograd (fp64 oracle as truth)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import isa_amd  # noqa
from isa_amd.reseg import ReSeg
from isa_amd.trainer import Trainer
import reseg_ref as R

def main(dtype=torch.float32, size=64, batch=2, use_ins=True):
    sd = R.synth_state_dict(23, True)
    x, sem, ins, n = R.synth_batch(batch, size, size, seed=1)
    sel = [list(reversed(range(int(k)))) for k in n.view(-1)]
    pick = (lambda a: torch.topk(a, 3, dim=1).indices[:, 2])
    # oracle in float64
    P = {k: (v.double().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.double() if v.dtype.is_floating_point else v)) for k, v in sd.items()}
    ctx = R.Ctx(bn_train=True, training=True, drop_rate=0.0)
    o = R.reseg_forward(P, x.double(), sem, ins, n, ctx=ctx, state=R.HeadState(), selected_idx=sel, sample_fn=pick, use_instance_seg=use_ins)
    ce, dice = R.sem_losses(o["sem_out"], sem)
    loss = ce + dice + (o["ins_cost_finite"] if use_ins else 0)
    loss.backward()
    model = ReSeg(2, use_ins, dtype=dtype); model.load_state_dict({k: v for k, v in sd.items() if k in model.state_dict()}); model.train(); model.head.drop_rate = 0.0
    tr = Trainer(model)
    inj = [torch.tensor(t["s_t"], dtype=torch.int32, device="cuda") for t in o["trace"]] if use_ins else None
    out = tr.forward_backward(x, sem, ins, n, selected_idx=sel, injected_s_t=inj)
    torch.cuda.synchronize()
    print("sem ce/dice mine", out["sem"].tolist(), "ref", float(ce), float(dice))
    if use_ins:
        print("head scal mine", out["head"].tolist(), "ref", float(o["ins_cost_finite"]), float(o["criterion"]), float(o["ins_ce_loss"]), float(o["ins_dice_loss"]))
    rows = []
    gmax = max(float(P[k].grad.norm()) for k in P if getattr(P[k], "grad", None) is not None)
    for k in model.store.names:
        if k not in model.store.offsets or "running" in k: continue
        g = P[k].grad
        mine = model.store.gview(k).double().cpu()
        if g is None:
            rows.append((float(mine.abs().max()), 0.0, k, "none")); continue
        gn = float(g.norm())
        err = float((mine - g).norm()) / (gn + 1e-30)
        rows.append((err if gn > 1e-6 * gmax else 0.0, gn, k, ""))
    rows.sort(reverse=True)
    print("dtype", dtype, "worst relative L2 grad errors:")
    for r in rows[:25]:
        print("   %.3e  |g|=%.3e  %s %s" % r)
    print("attention front / stems / SE:")
    for r in sorted(rows, key=lambda r: r[2]):
        if r[2].startswith(("decoder.s_sp", "decoder.attend", "ins_seg_output", "channelAttend", "sem_seg")):
            print("   %.3e  |g|=%.3e  %s %s" % r)
    bad = [r for r in rows if r[0] > 1e-2]
    print("n tensors with rel err > 1e-2:", len(bad), "of", len(rows))

if __name__ == "__main__":
    main(use_ins=(len(sys.argv) < 2 or sys.argv[1] != "sem"))
