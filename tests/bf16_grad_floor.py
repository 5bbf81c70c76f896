#!/usr/bin/env python3
"""What bf16 STORAGE alone does to this network's gradients, measured on the reference's arithmetic.

The float64 CPU oracle (pinned to the reference: tests/test_oracle_golden.py) is run twice on a *_drop_f64 fixture's
inputs, weights, glimpse points and Dropout2d masks: once as it is, once with every raw convolution output and every
block output rounded to bf16 (Ctx.storage_round, straight-through backward) - nothing else changes, all arithmetic stays
float64.  The relative L2 distance between the two gradients, per parameter tensor and family, is the floor any bf16-
storage implementation of the reference sits on: tiny-batch BatchNorm, ReLU6 thresholds and ten decoder levels in series
amplify a 2^-9 rounding of the activations to tens of per cent on many tensors.
tests/test_gpu_train.py::test_bf16_gradients_256_vs_reference_f64 quotes the 256x256 output of this script.

usage: python tests/bf16_grad_floor.py [64|256]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle")]
import reseg_ref as R  # noqa: E402


def family(name):
    if name.startswith("base."):
        return "backbone"
    if name.startswith("decoder.bone."):
        return "decoder"
    return "stems+heads"


def oracle_gradients(z, size, storage_round=None, round_input=False):
    x, sem, ins, n = R.synth_batch(2, size, size, seed=int(z["meta/size_batch_seed"][2]))
    sel = [[int(v) for v in row if v >= 0] for row in z["inject/selected_idx"]]
    masks = {k[len("inject/drop/"):]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith("inject/drop/")}
    s_t = z["inject/s_t"]
    sd = R.synth_state_dict(23, True)
    P = {k: (v.double().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
         for k, v in sd.items()}
    ctx = R.Ctx(bn_train=True, training=True, drop_rate=0.5 if masks else 0.0, drop_masks=masks or None,
                storage_round=storage_round)
    it = [0]

    def pick(_alpha):
        r = torch.tensor(s_t[it[0]])
        it[0] += 1
        return r

    xin = x.bfloat16().double() if round_input else x.double()
    out = R.reseg_forward(P, xin, sem, ins, n, ctx=ctx, state=R.HeadState(), selected_idx=sel, sample_fn=pick)
    ce, dice = R.sem_losses(out["sem_out"], sem)
    (out["ins_cost_finite"] + ce + dice).backward()
    return {k: v.grad.detach() for k, v in P.items() if getattr(v, "grad", None) is not None}


def floor(size):
    z = np.load(os.path.join(ROOT, "tests", "golden", "train_%d_drop_f64.npz" % size))
    g0 = oracle_gradients(z, size)
    g1 = oracle_gradients(z, size, storage_round=lambda t: t.bfloat16().double(), round_input=True)
    gmax = max(float(v.norm()) for v in g0.values())
    fam, d = {}, [0.0, 0.0, 0.0]
    for k in g0:
        if float(g0[k].norm()) <= 1e-6 * gmax:
            continue
        fam.setdefault(family(k), []).append(float((g1[k] - g0[k]).norm() / g0[k].norm()))
        d[0] += float((g1[k] * g0[k]).sum()); d[1] += float((g1[k] ** 2).sum()); d[2] += float((g0[k] ** 2).sum())
    cos = d[0] / np.sqrt(d[1] * d[2])
    return cos, {f: (float(np.median(v)), float(np.percentile(v, 90)), float(np.max(v))) for f, v in fam.items()}


if __name__ == "__main__":
    torch.set_num_threads(8)
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cos, stats = floor(size)
    print("bf16 storage emulated in the float64 oracle at %dx%d: cosine to the unrounded gradient %.5f" % (size, size, cos))
    for f, (med, p90, mx) in sorted(stats.items()):
        print("  %-12s relative L2: median %.3f  p90 %.3f  max %.3f" % (f, med, p90, mx))
