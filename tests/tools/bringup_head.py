"""GPU bring-up: full forward with GT (eval + train-BN) vs the CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import isa_amd  # noqa
from isa_amd.engine import Engine, ParamStore
from isa_amd.network import Network
from isa_amd.instance_head import InstanceHead
import reseg_ref as R

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def run(dtype, training, size=64, batch=2):
    sd = R.synth_state_dict(23, True)
    ps = ParamStore(R.state_dict_schema(True), "cuda"); ps.load_state_dict(sd)
    eng = Engine(ps, dtype); net = Network(eng); head = InstanceHead(net); head.drop_rate = 0.0
    x, sem, ins, n = R.synth_batch(batch, size, size, seed=1)
    sel = [list(reversed(range(int(k)))) for k in n.view(-1)]
    ctx = R.Ctx(bn_train=training, training=training, capture=True, drop_rate=0.0)
    pick = (lambda a: torch.topk(a, 3, dim=1).indices[:, 2])
    with torch.no_grad():
        ref = R.reseg_forward(sd, x, sem, ins, n, ctx=ctx, state=R.HeadState(), selected_idx=sel, sample_fn=pick)
    eng.begin(bn_train=training, record=False)
    xin = net.to_nhwc(x.cuda())
    y, feats = net.unet(xin)
    sem_a = net.sem_head(y)
    sem_map = sem.argmax(1).float().reshape(batch, -1).cuda().contiguous()
    inj = None
    if training:
        inj = [torch.tensor(t["s_t"], dtype=torch.int32, device="cuda") for t in ref["trace"]]
    cap = {}
    rec = head.forward(y, feats, sem_map, ins.cuda().contiguous(), [int(v) for v in n.view(-1)], training, sel, inj, cap)
    torch.cuda.synchronize()
    out = {}
    out["x_enc"] = rel(cap["x_enc"].nchw(), ctx.taps["x_enc"])
    out["s_sp"] = rel(cap["s_sp"].nchw(), ctx.taps["s_sp.out"])
    out["merge"] = rel(cap["merge"].view(batch, 1, size, size), ctx.taps["attend.pro_merge"])
    for it, tr in enumerate(ref["trace"]):
        a_ref = torch.stack([ctx.taps["attend.pro_split"][b, tr["idx"][b]] for b in range(batch)])
        out["it%d.alpha" % it] = rel(rec["iters"][it]["alpha"].view(batch, size, size), a_ref)
        st = rec["iters"][it]["s_t"].cpu().tolist()
        out["it%d.s_t_eq" % it] = float(st == tr["s_t"])
        for lvl in range(5):
            out["it%d.L%d.x" % (it, lvl)] = rel(cap["it%d.L%d.x" % (it, lvl)].nchw(), ctx.taps["it%d.L%d.x" % (it, lvl)])
            out["it%d.L%d.pred" % (it, lvl)] = rel(cap["it%d.L%d.pred" % (it, lvl)].nchw(), ctx.taps["it%d.L%d.pred" % (it, lvl)])
            tg = rec["iters"][it]["targets"][lvl].view(tr["targets"][lvl].shape)
            out["it%d.L%d.tgt_eq" % (it, lvl)] = float(torch.equal(tg.cpu(), tr["targets"][lvl]))
    sc = InstanceHead.scalars_from_sums(rec, training)
    print("dtype=%s training=%s" % (dtype, training))
    for k, v in out.items():
        print("   %-16s %.2e" % (k, v))
    for k in ("ins_cost", "criterion", "ins_ce_loss", "ins_dice_loss"):
        print("   %-14s mine %.6f ref %.6f" % (k, sc[k], float(ref[k])))
    sys.stdout.flush()

if __name__ == "__main__":
    run(torch.float32, False)
    run(torch.float32, True)
    run(torch.bfloat16, True)
