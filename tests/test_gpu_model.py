"""Model-level parity through the drop-in ReSeg class (C-ABI path) on the GPU.

Checks against (a) tests/golden/*.npz = outputs of the upstream reference itself and (b) the CPU
oracle run live on the same seeded inputs.  Tolerances (north star): fp32 storage <= 1e-3 relative
on float activations, index maps bit-exact (except 1-ulp near-ties, see assert_index_map);
bf16 storage: mask IoU and fp32-accumulated scalars.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import golden_io as G       # noqa: E402
import reseg_ref as R       # noqa: E402
from test_oracle_golden import assert_index_map    # noqa: E402


def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.reseg import ReSeg
    return ReSeg


def load(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def drop_masks(z):
    """{(iteration, level, kind): [n, C] keep mask} recorded by the reference run of a *_drop fixture (gen_golden.py)."""
    out = {}
    for k in z.files:
        if k.startswith("inject/drop/"):
            it, lvl, kind = k[len("inject/drop/"):].split(".")
            out[int(it[2:]), int(lvl[1:]), kind] = torch.from_numpy(z[k])
    return out


def build(ReSeg, use_ins, dtype, train, z=None):
    m = ReSeg(2, use_ins, dtype=dtype)
    sd = R.synth_state_dict(23, use_ins)
    m.load_state_dict(sd)
    m.train(train)
    masks = drop_masks(z) if z is not None else {}
    if masks:                               # Dropout2d active at the reference's rate (config.py:64), masks injected
        assert m.head.drop_rate == 0.5
        m.head.injected_masks = masks
    else:
        m.head.drop_rate = 0.0
    return m, sd


def iou(a, b):
    a, b = np.asarray(a, bool), np.asarray(b, bool)
    u = (a | b).sum()
    return 1.0 if u == 0 else float((a & b).sum()) / float(u)


def test_state_dict_roundtrip_and_keys():
    ReSeg = need_gpu()
    m, sd = build(ReSeg, True, torch.float32, False)
    out = m.state_dict()
    assert list(out.keys()) == [n for n, _ in R.state_dict_schema(True)]
    for k, v in sd.items():
        assert torch.equal(out[k].cpu(), v), k
    assert hasattr(m, "base") and len(list(m.base.parameters())) > 0
    assert sum(p.numel() for p in m.parameters()) == 4779392
    assert m.cuda() is m


@pytest.mark.parametrize("case", ["infer_32", "infer_256"])
def test_inference_vs_reference_golden(case):
    ReSeg = need_gpu()
    z = load(case)
    size, batch, seed = (int(v) for v in z["meta/size_batch_seed"])
    x, _, _, _ = R.synth_batch(batch, size, size, seed=seed)
    m, sd = build(ReSeg, False, torch.float32, False)
    sem_out, sem_arg = m(False, x)
    torch.cuda.synchronize()
    err, cs = G.compare(z, "sem_out", sem_out.cpu().numpy())
    assert err < 1e-3 and cs < 1e-3, (err, cs)
    so = sem_out.cpu()
    margin = (so[:, 1] - so[:, 0]).abs().numpy()
    assert_index_map(G.unpack_bits(z, "sem_argmax")[:, 0], sem_arg.cpu().numpy()[:, 0] != 0, margin,
                     float(so.abs().max()), "sem_argmax", rel=1e-5)
    # reference raises for GT-free instance mode; so do we (different exception type, same contract)
    m2, _ = build(ReSeg, True, torch.float32, False)
    with pytest.raises(RuntimeError):
        m2(False, x)


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def margin_iou(l16, l32, t):
    """IoU of the thresholded masks over the pixels whose fp32 margin |l1 - l0| exceeds 2 * t * max|logit|: a logit
    error below t * max|logit| cannot flip them.  Returns (iou, fraction of pixels kept, raw iou over all pixels)."""
    keep = ((l32[:, 1] - l32[:, 0]).abs() > 2 * t * float(l32.abs().max())).numpy()
    m16, m32 = (l16[:, 1] > l16[:, 0]).numpy(), (l32[:, 1] > l32[:, 0]).numpy()
    return iou(m16[keep], m32[keep]), float(keep.mean()), iou(m16, m32)


def test_inference_bf16_mask_iou():
    """bf16 storage against the fp32 oracle (the reference's algorithm on the CPU) at 256x256.  Raw conv outputs are
    rounded to bf16 BEFORE BatchNorm (as under torch autocast), so a channel whose mean is large against its spread
    carries a relative error of 2^-9 * |mean| / std after normalisation; at random init that accumulates to a few per
    cent of the logit scale (measured: DESIGN.md 2).  Bounds: relative L2 of the logits, and the thresholded mask
    identical (IoU >= 0.999) on every pixel whose oracle margin exceeds 10 % of the logit scale."""
    ReSeg = need_gpu()
    x, _, _, _ = R.synth_batch(2, 256, 256, seed=1)
    m, sd = build(ReSeg, False, torch.bfloat16, False)
    sem_out, sem_arg = m(False, x)
    with torch.no_grad():
        ref = R.reseg_forward(sd, x, use_instance_seg=False)["sem_out"]
    got = sem_out.cpu()
    l2, mx = _rel_l2(got, ref), float((got - ref).abs().max() / ref.abs().max())
    v, kept, raw = margin_iou(got, ref, 0.05)
    print("bf16 sem logits vs fp32 oracle: rel-L2 %.3e, max-abs/max %.3e; IoU %.5f on the %.1f%% confident pixels; raw IoU %.4f"
          % (l2, mx, v, 100 * kept, raw))
    assert l2 <= 0.10 and mx <= 0.30, (l2, mx)
    assert v >= 0.999 and kept > 0.3, (v, kept)


def _run_gt(ReSeg, z, dtype, training):
    size, batch, seed = (int(v) for v in z["meta/size_batch_seed"])
    x, sem, ins, n = R.synth_batch(batch, size, size, seed=seed)
    sel = [[int(v) for v in row if v >= 0] for row in z["inject/selected_idx"]]
    m, sd = build(ReSeg, True, dtype, training, z)
    inj = None
    if training:
        inj = [torch.tensor(row, dtype=torch.int32, device="cuda") for row in z["inject/s_t"]]
    cap = {}
    out = m(training, x, sem, ins, n, selected_idx=sel, injected_s_t=inj, capture=cap)
    torch.cuda.synchronize()
    return m, out, cap


@pytest.mark.parametrize("case,training", [("evalgt_64", False), ("train_64", True),
                                           ("train_64_drop", True), ("train_256_drop", True)])    # Dropout2d p=.5 active
def test_forward_with_gt_vs_reference_golden(case, training):
    ReSeg = need_gpu()
    z = load(case)
    m, out, cap = _run_gt(ReSeg, z, torch.float32, training)
    tol = 1e-3
    for nm, key in (("x_enc", "x_enc"), ("s_sp.out", "s_sp")):
        err, _ = G.compare(z, nm, cap[key].nchw().cpu().numpy())
        assert err < tol, (nm, err)
    for nm in ("x_dec", "x1", "x2", "x3", "x4", "x5"):                 # UNet.forward's six maps (unet_model.py:36)
        err, cs = G.compare(z, "unet." + nm, cap["unet." + nm].nchw().cpu().numpy())
        assert err < tol and cs < tol, (nm, err, cs)
    size = int(z["meta/size_batch_seed"][0])
    b = int(z["meta/size_batch_seed"][1])
    err, _ = G.compare(z, "attend.pro_merge", cap["merge"].view(b, 1, size, size).cpu().numpy())
    assert err < tol, err
    rec = m.last_record
    assert len(rec["iters"]) == z["inject/s_t"].shape[0]
    # alpha (utils.py:648-655): the reference evaluates pro_split[B,32,H,W]; here only the row of the instance each
    # iteration selects exists.  Compare those rows, as floats, at the fixture's stored sample positions.
    shape = tuple(int(v) for v in z["attend.pro_split/shape"])
    total = int(np.prod(shape))
    pos = np.arange(0, total, G.subsample_stride(total))
    ref = z["attend.pro_split/sub"].astype(np.float64)
    pb, pi, ppix = pos // (shape[1] * size * size), (pos // (size * size)) % shape[1], pos % (size * size)
    checked = 0
    for r in rec["iters"]:
        idx = r["idx"].cpu().numpy()
        alpha = r["alpha"].view(b, -1).cpu().numpy().astype(np.float64)
        selm = pi == idx[pb]
        checked += int(selm.sum())
        got = alpha[pb[selm], ppix[selm]]
        assert np.abs(got - ref[selm]).max() <= tol * max(np.abs(ref[selm]).max(), 1e-30), "alpha rows"
        assert abs(alpha.sum() - b) < 1e-3                              # each selected row is a distribution
    assert checked > 50, checked
    for it, r in enumerate(rec["iters"]):
        assert r["s_t"].cpu().tolist() == [int(v) for v in z["inject/s_t"][it]]          # exact
        for lvl in range(5):
            pre = "it%d.L%d" % (it, lvl)
            err, _ = G.compare(z, pre + ".x", cap[pre + ".x"].nchw().cpu().numpy())
            assert err < tol, (pre, err)
            pred = cap[pre + ".pred"].nchw().cpu()
            err, _ = G.compare(z, pre + ".pred", pred.numpy())
            assert err < tol, (pre, err)
            f = 16 >> lvl
            tgt = r["targets"][lvl].view(b, 1, size // f, size // f).cpu().numpy() != 0
            assert np.array_equal(G.unpack_bits(z, pre + ".target"), tgt)                  # exact
            margin = (pred[:, 1] - pred[:, 0]).abs().numpy()
            assert_index_map(G.unpack_bits(z, pre + ".mask_pred"), (pred[:, 1] > pred[:, 0]).numpy(), margin,
                             max(1.0, float(pred.abs().max())), pre, rel=1e-4)
    keys = ("criterion", "ins_ce_loss", "ins_dice_loss") + (() if training else ("ins_cost",))
    vals = dict(zip(("ins_cost", "criterion", "ins_ce_loss", "ins_dice_loss"), out[2:]))
    for k in keys:
        ref = float(z["scalars/" + k][0])
        assert abs(float(vals[k]) - ref) <= 1e-3 * max(1.0, abs(ref)), (k, float(vals[k]), ref)
    if training:
        assert bool(torch.isnan(vals["ins_cost"]))                 # reference quirk, attenet2.py:77
        # running statistics updated like the reference (decoder BNs twice: two iterations)
        sd_after = m.state_dict()
        for k in ("base.inc.conv.conv.down_conv_0.conv.1.running_mean",
                  "decoder.bone.upAtten4.UpAtten.conv1.1.running_var", "decoder.attend.bn.running_mean",
                  "decoder.s_sp.bn.running_var"):
            err, _ = G.compare(z, "buf/" + k, sd_after[k].cpu().numpy(), 512)
            assert err < 1e-3, (k, err)
        nbt = z["scalars/nbt_first_last"]
        assert int(sd_after["base.inc.conv.conv.down_conv_0.conv.1.num_batches_tracked"]) == nbt[0]
        assert int(sd_after["decoder.bone.upAtten4.UpAtten.conv1.1.num_batches_tracked"]) == nbt[1]
        assert int(sd_after["decoder.attend.bn.num_batches_tracked"]) == nbt[2]


def test_train_forward_256_bf16_per_level_bounds_and_margin_aware_iou():
    """bf16 storage (the benchmarked precision) at 256x256 against (a) the reference's own outputs (train_256.npz:
    loss scalars, mask bits) and (b) the fp32-storage run of the same HIP path (itself within 1e-3 of the reference:
    the tests above), which supplies the full-resolution logits the fixture only samples.
      * loss scalars within 1e-2 of the reference's (measured 3e-3);
      * per tensor relative-L2 error of the six UNet maps, x_enc and every level's x / pred (2 iterations x 5 levels):
        bounds at ~1.5x the measured values (scripts/bf16diag.py: UNet 0.7 - 2.4 %, x_enc 4.7 %, decoder levels 8 - 13 %;
        bf16 rounds the raw conv outputs ahead of BatchNorm, see test_inference_bf16_mask_iou);
      * margin-aware IoU: on the pixels whose fp32 margin |l1 - l0| exceeds 10 % of max|logit| (4 % for the
        semantic head) the bf16 mask equals the fp32 mask AND the reference's stored bits (IoU >= 0.999), for the
        semantic mask and all ten instance masks; the fraction of pixels kept and the raw IoU are printed."""
    ReSeg = need_gpu()
    z = load("train_256")
    m32, out32, cap32 = _run_gt(ReSeg, z, torch.float32, True)
    acts32 = {k: v.nchw().cpu() for k, v in cap32.items() if k.startswith(("it", "unet.")) or k == "x_enc"}
    sem32 = out32[0].cpu()
    del m32, cap32
    m, out, cap = _run_gt(ReSeg, z, torch.bfloat16, True)
    vals = dict(zip(("ins_cost", "criterion", "ins_ce_loss", "ins_dice_loss"), out[2:]))
    for k in ("criterion", "ins_ce_loss", "ins_dice_loss"):
        ref = float(z["scalars/" + k][0])
        assert abs(float(vals[k]) - ref) <= 1e-2 * max(1.0, abs(ref)), (k, float(vals[k]), ref)
    worst = {}
    for k, a32 in sorted(acts32.items()):
        a16 = cap[k].nchw().cpu()
        grp = "unet" if k.startswith("unet.") else ("x_enc" if k == "x_enc" else "decoder")
        bound = dict(unet=4e-2, x_enc=8e-2, decoder=0.2)[grp]
        e = _rel_l2(a16, a32)
        worst[grp] = max(worst.get(grp, 0.0), e)
        assert e <= bound, (k, e, bound)
        assert float((a16 - a32).abs().max() / a32.abs().max()) <= 2 * bound, k
    print("bf16 vs fp32 storage at 256x256, worst relative-L2 per group:", {k: "%.3e" % v for k, v in worst.items()})

    def check_mask(l16, l32, bits_ref, name, t):
        v, kept, raw = margin_iou(l16, l32, t)
        keep = ((l32[:, 1] - l32[:, 0]).abs() > 2 * t * float(l32.abs().max())).numpy()
        vr = iou((l16[:, 1] > l16[:, 0]).numpy()[keep], np.asarray(bits_ref, bool)[keep])
        print("%-8s kept %5.1f%% of the pixels; IoU on them %.5f (vs the reference's bits %.5f); raw IoU %.4f"
              % (name, 100.0 * kept, v, vr, raw))
        assert v >= 0.999 and vr >= 0.999, (name, v, vr)
        assert kept > 0.3, (name, kept)

    sem16 = out[0].cpu()
    assert _rel_l2(sem16, sem32) <= 5e-2
    check_mask(sem16, sem32, (sem32[:, 1] > sem32[:, 0]).numpy(), "sem", 0.02)
    for it in range(2):
        for lvl in range(5):
            pre = "it%d.L%d" % (it, lvl)
            check_mask(cap[pre + ".pred"].nchw().cpu(), acts32[pre + ".pred"], G.unpack_bits(z, pre + ".mask_pred"), pre, 0.05)


def test_graph_replayed_inference_matches_eager():
    """ReSeg.infer_graphed (hipGraph replay of the GT-free forward) against the eager launch loop, for two
    different inputs through the same captured graph (the SE block's channel means are float atomics: equal to
    summation order)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.reseg import ReSeg
    import reseg_ref as R
    m = ReSeg(2, False, dtype=torch.float32)
    m.load_state_dict(R.synth_state_dict(23, False))
    m.eval()
    xs = [R.synth_batch(2, 64, 64, seed=s)[0] for s in (1, 2, 3)]
    with torch.no_grad():
        eager = [tuple(t.clone() for t in m(False, x)) for x in xs]
    got = []
    for x in xs:                                   # call 1 eager (first sight), call 2 captures, call 3 replays
        out = m.infer_graphed(x)
        torch.cuda.synchronize()
        got.append(tuple(t.clone() for t in out))
    assert any(s.get("state") == "ready" for s in m._infer_graphs.values())
    for (se, ae), (sg, ag) in zip(eager, got):
        assert float((se - sg).abs().max() / se.abs().max()) < 1e-5
        assert float((ae != ag).float().mean()) < 1e-3


def test_whole_block_eval_fusion_matches_the_two_launch_path():
    """isa_dwpw_eval (InvertedV1Residual in one launch, the depthwise output resident in LDS) against the op-granular
    eval path (isa_dwconv3x3 + isa_conv_gemm_ep) it replaces on the 256x256 ... 32x32 backbone levels: same roundings
    by construction (the depthwise result is rounded to bf16 where the two-launch path stores it), so the two differ by
    the summation order of the 1x1 contraction only; and against the fp32 oracle like every bf16 run."""
    ReSeg = need_gpu()
    x, _, _, _ = R.synth_batch(2, 256, 256, seed=1)
    m, sd = build(ReSeg, False, torch.bfloat16, False)
    E = m.engine
    E.profile = True
    outs = {}
    for fuse in (False, True):
        E.fuse_block = fuse
        E.prof_events = []
        sem_out, _ = m(False, x)
        torch.cuda.synchronize()
        calls = E.profile_summary()
        outs[fuse] = sem_out.float().cpu().clone()
        if fuse:
            assert calls["isa_dwpw_eval"][0] >= 8 and calls.get("isa_dwconv3x3", (0,))[0] <= 10, calls
        else:
            assert "isa_dwpw_eval" not in calls
    E.profile = False
    with torch.no_grad():
        ref = R.reseg_forward(sd, x, use_instance_seg=False)["sem_out"]
    d = float((outs[True] - outs[False]).abs().max() / outs[False].abs().max())
    e_f, e_u = _rel_l2(outs[True], ref), _rel_l2(outs[False], ref)
    print("fused vs two-launch eval blocks: max-abs/max %.3e, rel-L2 %.3e; vs the fp32 oracle: fused %.3e, two-launch %.3e"
          % (d, _rel_l2(outs[True], outs[False]), e_f, e_u))
    # a different summation order flips last bits of bf16 roundings, which 40 layers amplify like any bf16 noise: the two
    # paths must be equally far from the fp32 oracle and no farther from each other than either is from it
    assert _rel_l2(outs[True], outs[False]) <= max(e_f, e_u) and d <= 0.10, d
    assert e_f <= 1.25 * e_u + 1e-3
    assert e_f <= 0.10
    v, kept, _ = margin_iou(outs[True], ref, 0.05)
    assert v >= 0.999 and kept > 0.3
