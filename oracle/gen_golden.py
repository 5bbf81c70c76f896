"""Generate tests/golden/*.npz by running the UPSTREAM reference network on seeded inputs.

Run in the build container only (needs /root/reference):  python oracle/gen_golden.py
Fixtures are data (inputs are re-synthesised from seeds; expected outputs are stored): small
tensors in full, large ones as a strided subsample plus float64 checksums, integer maps
bit-packed in full.  Weights come from `reseg_ref.synth_state_dict` (numpy RandomState keyed by
tensor name), loaded into the reference with load_state_dict(strict=True).

Reference-side knobs touched (interpreter state only, nothing in /root/reference is edited):
  config.H/W   -> fixture size (the reference hard-codes 256, config.py:1)
  config.drop_rate = 0 before construction for the plain cases; the `*_drop` cases keep the reference's .5
                 (config.py:64) and replace nn.Dropout2d.forward / F.dropout2d by seeded per-(image, channel) keep masks
                 that are recorded under inject/drop/ (dropout parity is by injection)
  decoder.getRandomIdx / torch.multinomial -> deterministic choices recorded in the fixture
  decoder.vis -> no-op (debug JPEG writer, needs cv2)
"""
import io
import os
import sys
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim          # noqa: E402
import reseg_ref as R    # noqa: E402

from golden_io import pack, pack_bits  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def topk3(p, nsamp=1, *a, **k):
    """Deterministic stand-in for torch.multinomial: the 3rd most probable position."""
    return torch.topk(p, 3, dim=1).indices[:, 2:3]


class DropRecorder:
    """Stands in for nn.Dropout2d.forward (utils.py:984, the `cross` branch) and F.dropout2d (utils.py:1104-1110): the
    same Bernoulli(keep) / keep channel mask torch draws, but from a seeded numpy stream and recorded in call order -
    per decoder iteration and level: cross (module), d1, d2 (functional)."""

    def __init__(self, p, seed=77):
        self.p, self.rs = p, np.random.RandomState(seed)
        self.masks, self.n_module, self.n_func = {}, 0, 0

    def _mask(self, x, name):
        keep = 1.0 - self.p
        m = (self.rs.rand(x.shape[0], x.shape[1]) < keep).astype(np.float64) / keep
        self.masks[name] = m.astype(np.float32)
        return x * torch.from_numpy(m).to(x.dtype)[:, :, None, None]

    def module_forward(self, mod, x):
        if not mod.training:
            return x
        it, lvl = divmod(self.n_module, 5)
        self.n_module += 1
        return self._mask(x, "it%d.L%d.cross" % (it, lvl))

    def functional(self, x, p=0.5, training=True, inplace=False):
        if not training or p <= 0:
            return x
        it, r = divmod(self.n_func, 10)
        self.n_func += 1
        return self._mask(x, "it%d.L%d.d%d" % (it, r // 2, 1 + r % 2))

    @contextlib.contextmanager
    def installed(self):
        F = torch.nn.functional
        real_f, real_m = F.dropout2d, torch.nn.Dropout2d.forward
        rec = self
        F.dropout2d = self.functional
        torch.nn.Dropout2d.forward = lambda mod, x: rec.module_forward(mod, x)
        try:
            yield self
        finally:
            F.dropout2d, torch.nn.Dropout2d.forward = real_f, real_m


def build(reseg, config, size, use_ins, dtype, drop=0.0):
    config.H = config.W = size
    config.drop_rate = drop
    m = reseg.ReSeg(2, use_ins)
    sd = R.synth_state_dict(23, True)
    own = m.state_dict()
    m.load_state_dict({k: sd[k] for k in own}, strict=True)
    m.decoder.vis = lambda *a, **k: None
    return m.to(dtype)


def hook_outputs(m, store_list):
    """Forward hooks recording the intermediates named in SURVEY §8(c)."""
    hs = []
    rec = store_list

    def add(mod, key):
        hs.append(mod.register_forward_hook(lambda _m, _i, o, key=key: rec.append((key, o))))

    add(m.base, "unet")
    add(m.decoder.s_sp, "s_sp")
    add(m.decoder.attend, "attend")
    for lvl in range(5):
        add(getattr(m.decoder.bone, "upAtten%d" % lvl), "L%d" % lvl)
    hs.append(m.decoder.register_forward_pre_hook(lambda _m, i: rec.append(("x_enc", i[0]))))
    return hs


def run_case(reseg, config, name, size, batch, mode, dtype=torch.float32, seed=1):
    drop = 0.5 if name.endswith("_drop") or "_drop_" in name else 0.0     # config.py:64
    store = {}
    x, sem, ins, n = R.synth_batch(batch, size, size, seed=seed)
    x = x.to(dtype)
    store["meta/size_batch_seed"] = np.array([size, batch, seed], dtype=np.int64)
    if mode == "infer":
        m = build(reseg, config, size, False, dtype).eval()
        rec = []
        hs = hook_outputs(m, rec)
        with torch.no_grad():
            sem_out, sem_arg = m(False, x)
        for key, o in rec:
            if key == "unet":
                for nm, t in zip(("x_dec", "x1", "x2", "x3", "x4", "x5"), o):
                    pack("unet." + nm, t, store)
        pack("sem_out", sem_out, store)
        pack_bits("sem_argmax", sem_arg, store)
        pack_bits("sem_prob_gt_half", torch.softmax(sem_out, 1)[:, 1] > 0.5, store)
        return store

    m = build(reseg, config, size, True, dtype, drop)
    training = mode == "train"
    dropper = DropRecorder(drop) if drop > 0 else None
    m.train(training)
    order = [list(reversed(range(int(k)))) for k in n.view(-1)]
    m.decoder.getRandomIdx = lambda n_ins: [list(s) for s in order]
    store["inject/selected_idx"] = np.array([o + [-1] * (32 - len(o)) for o in order], dtype=np.int64)
    rec = []
    hook_outputs(m, rec)
    s_ts = []
    orig_sample = m.decoder.sample

    def sample(*a, **k):
        out = orig_sample(*a, **k)
        s_ts.append(list(out[0]))
        return out

    m.decoder.sample = sample
    real_multinomial = torch.multinomial
    torch.multinomial = topk3
    try:
        with contextlib.redirect_stdout(io.StringIO()), (dropper.installed() if dropper else contextlib.nullcontext()):
            if training:
                out = m(True, x, sem, ins, n)
            else:
                with torch.no_grad():
                    out = m(False, x, sem, ins, n)
    finally:
        torch.multinomial = real_multinomial
    sem_out, sem_arg, ins_cost, criterion, ins_ce, ins_dice = out
    pack("sem_out", sem_out, store)
    pack_bits("sem_argmax", sem_arg, store)
    store["scalars/ins_cost_isnan"] = np.array([bool(torch.isnan(ins_cost).all())])
    if not training:
        store["scalars/ins_cost"] = np.array([float(ins_cost)])
    store["scalars/criterion"] = np.array([float(criterion)])
    store["scalars/ins_ce_loss"] = np.array([float(ins_ce)])
    store["scalars/ins_dice_loss"] = np.array([float(ins_dice)])
    store["inject/s_t"] = np.array(s_ts, dtype=np.int64)
    if dropper:
        assert dropper.n_module == 5 * len(s_ts) and dropper.n_func == 10 * len(s_ts), (dropper.n_module, dropper.n_func)
        for k_, v_ in dropper.masks.items():
            store["inject/drop/" + k_] = v_
    it = {}
    for key, o in rec:
        if key == "unet":
            for nm, t in zip(("x_dec", "x1", "x2", "x3", "x4", "x5"), o):
                pack("unet." + nm, t, store)
        elif key == "x_enc":
            pack("x_enc", o, store)
        elif key == "s_sp":
            pack("s_sp.out", o, store)
        elif key == "attend":
            pack("attend.pro_split", o[0], store)
            pack("attend.pro_merge", o[1], store)
        else:
            k = it.get(key, 0)
            it[key] = k + 1
            pack("it%d.%s.x" % (k, key), o[0], store)
            pack("it%d.%s.pred" % (k, key), o[1], store)
            pack_bits("it%d.%s.mask_pred" % (k, key), o[1][:, 1] > o[1][:, 0], store)
            pack_bits("it%d.%s.target" % (k, key), o[2], store)
    if training:
        ce, dice = R.sem_losses(sem_out, sem)
        store["scalars/sem_ce"] = np.array([float(ce)])
        store["scalars/sem_dice"] = np.array([float(dice)])
        m.zero_grad()
        (ins_cost + ce + dice).backward()      # NaN-valued cost, finite grads
        names, gsum = [], []
        for k, p in m.named_parameters():
            if p.grad is None:
                store["grad_none/" + k] = np.array([1])
                continue
            pack("grad/" + k, p.grad, store, 512)
        store["scalars/baseline"] = np.array([float(m.decoder.baseline)])
        sd_after = m.state_dict()
        for k in sd_after:
            if k.endswith("running_mean") or k.endswith("running_var"):
                pack("buf/" + k, sd_after[k], store, 512)
        store["scalars/nbt_first_last"] = np.array(
            [int(sd_after["base.inc.conv.conv.down_conv_0.conv.1.num_batches_tracked"]),
             int(sd_after["decoder.bone.upAtten4.UpAtten.conv1.1.num_batches_tracked"]),
             int(sd_after["decoder.attend.bn.num_batches_tracked"])])
    return store


def byname_cases(store):
    """a19-a21: the reference's own (dead-at-HEAD) attention operator classes."""
    from modules import utils as U   # /root/reference/code/lib/archs/modules/utils.py
    rs = np.random.RandomState(5)
    torch.manual_seed(5)             # the operator classes draw their initial weights from torch's generator
    # a19: ScaledDotProductAttention with one query point against L keys (utils.py:305-329)
    bh, L, d = 4, 1024, 12
    q = torch.from_numpy(rs.standard_normal((bh, 1, d)).astype(np.float32))
    k = torch.from_numpy(rs.standard_normal((bh, L, d)).astype(np.float32))
    v = torch.from_numpy(rs.standard_normal((bh, L, d)).astype(np.float32))
    mask = torch.from_numpy(rs.rand(bh, 1, L) < 0.3)
    att = U.ScaledDotProductAttention(temperature=float(np.power(d, 0.5))).eval()
    with torch.no_grad():
        o, a = att(q, k, v, mask=mask)
    for nm, t in (("q", q), ("k", k), ("v", v), ("out", o), ("attn", a)):
        store["sdp/" + nm] = t.numpy()
    store["sdp/mask"] = mask.numpy()
    # a20: _ScalePDAttention (utils.py:248-303): record projected Q,K,V and the attended map
    mod = U._ScalePDAttention(d_k=12, d_v=12, d_model=24, dilation_rate=2, n_head=2).eval()
    got = {}
    mod.qk_w.register_forward_hook(lambda _m, _i, o: got.__setitem__("QK", o))
    mod.v_w.register_forward_hook(lambda _m, _i, o: got.__setitem__("V", o))
    mod.fc.register_forward_pre_hook(lambda _m, i: got.__setitem__("att", i[0]))
    b, h, w = 2, 16, 16
    qk = torch.from_numpy(rs.standard_normal((b, 24, h, w)).astype(np.float32))
    vv = torch.from_numpy(rs.standard_normal((b, 24, h, w)).astype(np.float32))
    nomask = torch.from_numpy((rs.rand(b, 1, h, w) < 0.2).astype(np.float32))
    with torch.no_grad():
        mod(qk, vv, nomask)
    store["local/QK"] = got["QK"].numpy()       # [(b*2), 24, h, w] = Q(12) | K(12)
    store["local/V"] = got["V"].numpy()         # [(b*2), 12, h, w]
    store["local/nomask"] = nomask.numpy()
    store["local/att"] = got["att"].numpy()     # [b, 24, h, w]
    store["local/dilation"] = np.array([2])
    # a21: Decoder.forward (utils.py:59-69)
    dec = U.Decoder(1, 24, 40, 2, 12, 12).eval()
    qq = torch.from_numpy(rs.standard_normal((2, 24)).astype(np.float32))
    enc = torch.from_numpy(rs.standard_normal((2, 24, 16, 16)).astype(np.float32))
    with torch.no_grad():
        o = dec(qq, enc, None)
    store["pq/q"], store["pq/enc"], store["pq/out"] = qq.numpy(), enc.numpy(), o.numpy()
    # a19 whole operator: MultiHeadAttention (utils.py:167-225) as configured (config.py:22-25), eval mode, both branches
    mha = U.MultiHeadAttention(2, 24, 12, 12).eval()
    for k_, v_ in mha.state_dict().items():
        store["mha/sd/" + k_] = v_.numpy()
    b, L = 2, 2048
    mq = torch.from_numpy(rs.standard_normal((b, 1, 24)).astype(np.float32))
    mk = torch.from_numpy(rs.standard_normal((b, L, 24)).astype(np.float32))
    mv = torch.from_numpy(rs.standard_normal((b, L, 24)).astype(np.float32))
    mmask = torch.from_numpy(rs.rand(b, 1, L) < 0.25)
    with torch.no_grad():
        o1, a1 = mha(mq, mk, mv, mask=mmask)
        o2, _ = mha(mq, mk, mv, mask=mmask, last=True)
    for nm, t in (("q", mq), ("k", mk), ("v", mv), ("out", o1), ("attn", a1), ("last", o2)):
        store["mha/" + nm] = t.numpy()
    store["mha/mask"] = mmask.numpy()
    # a20 whole operator: _ScalePDAttention.forward incl. projections, fc and InstanceNorm (utils.py:248-303)
    for k_, v_ in mod.state_dict().items():
        store["local/sd/" + k_] = v_.numpy()
    with torch.no_grad():
        full = mod(qk, vv, nomask)
    store["local/qk_in"], store["local/v_in"], store["local/out"] = qk.numpy(), vv.numpy(), full.numpy()


def main():
    torch.set_num_threads(8)
    reseg, config = ref_shim.install()
    os.makedirs(OUT, exist_ok=True)
    cases = [
        ("infer_32", 32, 2, "infer", torch.float32),
        ("infer_256", 256, 2, "infer", torch.float32),
        ("evalgt_64", 64, 2, "evalgt", torch.float32),
        ("train_64", 64, 2, "train", torch.float32),
        ("train_64_f64", 64, 2, "train", torch.float64),
        ("train_256", 256, 2, "train", torch.float32),
        ("train_256_f64", 256, 2, "train", torch.float64),     # the reference's own fp64 run: gradient floor at 256^2
        # Dropout2d ACTIVE at the reference's rate (the configuration train.py and bench.py run): masks injected
        ("train_64_drop", 64, 2, "train", torch.float32),
        ("train_64_drop_f64", 64, 2, "train", torch.float64),
        ("train_256_drop", 256, 2, "train", torch.float32),
        ("train_256_drop_f64", 256, 2, "train", torch.float64),
    ]
    only = set(sys.argv[1:])              # python oracle/gen_golden.py [case ...]: regenerate a subset
    for name, size, batch, mode, dtype in cases:
        if only and name not in only:
            continue
        st = run_case(reseg, config, name, size, batch, mode, dtype)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **st)
        print(name, len(st), "arrays", os.path.getsize(path) // 1024, "KiB")
    if only and "byname_ops" not in only:
        return
    st = {}
    byname_cases(st)
    path = os.path.join(OUT, "byname_ops.npz")
    np.savez_compressed(path, **st)
    print("byname_ops", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
