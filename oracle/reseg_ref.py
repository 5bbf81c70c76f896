"""CPU oracle: functional fp32 restatement of the reference hot path (TEST INFRASTRUCTURE).

This file is the *checker*, never the product: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The product path (the HIP library behind
`include/isa_kernels.h`) never calls into it and fails loudly when the library is missing.

Parity status: PINNED.  `oracle/gen_golden.py` runs the upstream network itself
(`/root/reference/code/lib/archs/reseg.py`, imported through `oracle/ref_shim.py`) on seeded
inputs and commits outputs under `tests/golden/`; `tests/test_oracle_golden.py` replays the
same inputs through this file and compares.

It is a restatement, not a copy: the reference is an `nn.Module` class tree with module-global
configuration; this is a set of pure functions over a flat `state_dict` (same keys as the
reference) with every size derived from the tensors, so 64², 256², 512² and
1024² inputs all work.  Each function cites the reference lines it follows.

Conventions: NCHW float32 torch tensors on CPU.  `P` is the state_dict (name -> tensor).
`bn_train` selects batch statistics (and returns running-stat updates in `ctx.new_buffers`);
`training` is the flag the reference threads through `forward(training, ...)`.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

# ---- hyper-parameters the reference keeps in a module-global (modules/config.py) -------------
D_MODEL = 24          # config.py:23
D_K = 12              # config.py:24
MAX_ITER = 2          # config.py:56 (overrides :8)
LAMBDA_L = 0.5        # config.py:45
LAMBDA_R = 2.0        # config.py:46
LAMBDA_E = 5.0        # config.py:47
LAMBDA_INS = 1.0      # config.py:49
PYRAMID_W = (16.0, 8.0, 4.0, 2.0, 1.0)   # config.py:51
CE_WEIGHT = 10.0      # config.py:17
FOCAL_GAMMA = 2       # config.py:14
DROP_RATE = 0.5       # config.py:64
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


@dataclass
class Ctx:
    """Per-call state: mode flags, injected random choices, captured intermediates."""
    bn_train: bool = False
    training: bool = False
    drop_rate: float = DROP_RATE
    # name -> [B,C] multiplicative channel masks (already scaled by 1/(1-p)); None => identity
    drop_masks: Optional[Dict[str, torch.Tensor]] = None
    capture: bool = False
    # optional storage-rounding emulation (tests/bf16_grad_floor.py): applied to every raw convolution output ahead of
    # its BatchNorm and to every block output, straight-through in the backward pass - the reference's arithmetic with
    # activations STORED in a narrower type.  None (default) = the reference's own arithmetic.
    storage_round: Optional[object] = None
    taps: Dict[str, torch.Tensor] = field(default_factory=dict)
    new_buffers: Dict[str, torch.Tensor] = field(default_factory=dict)

    def tap(self, name, t):
        if self.capture:
            self.taps[name] = t

    def q(self, x):
        if self.storage_round is None:
            return x
        return x + (self.storage_round(x.detach()) - x.detach())


# ---------------------------------------------------------------------------------------------
# primitive layers
# ---------------------------------------------------------------------------------------------
def batchnorm(P, pre, x, ctx: Ctx):
    """nn.BatchNorm2d semantics (train: biased batch var, running update with unbiased var)."""
    w, b = P[pre + ".weight"], P[pre + ".bias"]
    x = ctx.q(x)
    if ctx.bn_train:
        dims = (0, 2, 3)
        n = x.numel() // x.shape[1]
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        with torch.no_grad():
            # a layer applied twice in one forward (decoder iterations) updates twice
            rm, rv, nb = (ctx.new_buffers.get(pre + s, P[pre + s]) for s in
                          (".running_mean", ".running_var", ".num_batches_tracked"))
            ctx.new_buffers[pre + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
            ctx.new_buffers[pre + ".running_var"] = \
                (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * (n / max(n - 1, 1))
            ctx.new_buffers[pre + ".num_batches_tracked"] = nb + 1
    else:
        mean, var = P[pre + ".running_mean"], P[pre + ".running_var"]
    scale = w / torch.sqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * scale[None, :, None, None] + b[None, :, None, None]


def relu6(x):
    return torch.clamp(x, 0.0, 6.0)


def block_v1(P, pre, x, ctx):
    """dw3x3-BN-ReLU6-pw-BN (+x when Cin==Cout).  MobileNetDenseASPP.py:68-93."""
    cin = x.shape[1]
    y = F.conv2d(x, P[pre + ".conv.0.weight"], None, 1, 1, 1, groups=cin)
    y = relu6(batchnorm(P, pre + ".conv.1", y, ctx))
    y = F.conv2d(y, P[pre + ".conv.3.weight"])
    y = batchnorm(P, pre + ".conv.4", y, ctx)
    return ctx.q(x + y if y.shape[1] == cin else y)


def block_ir(P, pre, x, ctx):
    """pw(x2)-BN-ReLU6-dw3x3-BN-ReLU6-pw-BN (+x when Cin==Cout).  MobileNetDenseASPP.py:96-123."""
    cin = x.shape[1]
    y = F.conv2d(x, P[pre + ".conv.0.weight"])
    y = relu6(batchnorm(P, pre + ".conv.1", y, ctx))
    y = F.conv2d(y, P[pre + ".conv.3.weight"], None, 1, 1, 1, groups=y.shape[1])
    y = relu6(batchnorm(P, pre + ".conv.4", y, ctx))
    y = F.conv2d(y, P[pre + ".conv.6.weight"])
    y = batchnorm(P, pre + ".conv.7", y, ctx)
    return ctx.q(x + y if y.shape[1] == cin else y)


def dropout2d(x, name, ctx: Ctx, active: bool):
    """Channel dropout.  Randomness is injected (SURVEY.md §7 'RNG parity')."""
    if not active or ctx.drop_rate <= 0:
        return x
    if ctx.drop_masks is None or name not in ctx.drop_masks:
        raise RuntimeError("dropout active at %s but no mask injected" % name)
    return x * ctx.drop_masks[name][:, :, None, None]


# ---------------------------------------------------------------------------------------------
# backbone  (unet_model.py:23-36, unet_parts.py:7-93)
# ---------------------------------------------------------------------------------------------
def double_v1(P, pre, x, ctx):
    x = block_v1(P, pre + ".conv.down_conv_0", x, ctx)
    return block_v1(P, pre + ".conv.down_conv_1", x, ctx)


def unet_down(P, pre, x, ctx):
    # interpolate(0.5, bilinear, align_corners=False) on even sizes == 2x2 mean (unet_parts.py:58)
    xd = F.avg_pool2d(x, 2)
    y = double_v1(P, pre + ".mpconv", xd, ctx)
    return torch.cat([y, xd], 1)


def unet_up(P, pre, xlow, xskip, ctx):
    u = F.conv_transpose2d(xlow, P[pre + ".up.weight"], P[pre + ".up.bias"], stride=2)
    dy, dx = xskip.shape[2] - u.shape[2], xskip.shape[3] - u.shape[3]
    if dy or dx:
        u = F.pad(u, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    return double_v1(P, pre + ".conv", torch.cat([xskip, u], 1), ctx)


def unet(P, x, ctx):
    x1 = double_v1(P, "base.inc.conv", x, ctx)
    x2 = unet_down(P, "base.down1", x1, ctx)
    x3 = unet_down(P, "base.down2", x2, ctx)
    x4 = unet_down(P, "base.down3", x3, ctx)
    x5 = unet_down(P, "base.down4", x4, ctx)
    y = unet_up(P, "base.up1", x5, x4, ctx)
    y = unet_up(P, "base.up2", y, x3, ctx)
    y = unet_up(P, "base.up3", y, x2, ctx)
    y = unet_up(P, "base.up4", y, x1, ctx)
    for n, t in zip(("x_dec", "x1", "x2", "x3", "x4", "x5"), (y, x1, x2, x3, x4, x5)):
        ctx.tap("unet." + n, t)
    return y, [x1, x2, x3, x4, x5]


# ---------------------------------------------------------------------------------------------
# heads on the decoder output  (reseg.py:72-102,112-123; utils.py:402-420)
# ---------------------------------------------------------------------------------------------
def se_gate(P, x):
    y = x.mean((2, 3))
    y = F.relu(F.linear(y, P["channelAttend.fc.0.weight"], P["channelAttend.fc.0.bias"]))
    y = torch.sigmoid(F.linear(y, P["channelAttend.fc.2.weight"], P["channelAttend.fc.2.bias"]))
    return x * y[:, :, None, None]


def sem_head(P, x_dec, ctx):
    out = F.conv2d(se_gate(P, x_dec), P["sem_seg_output.weight"], P["sem_seg_output.bias"])
    ctx.tap("sem_out", out)
    return out


def ins_stems(P, x_dec, ctx):
    p1, p2 = "ins_seg_output_1", "ins_seg_output_2"
    y = F.conv2d(x_dec, P[p1 + ".0.weight"], P[p1 + ".0.bias"], 1, 1, 1, groups=x_dec.shape[1])
    y = relu6(batchnorm(P, p1 + ".1", y, ctx))
    y = F.conv2d(y, P[p1 + ".3.weight"], P[p1 + ".3.bias"])
    e1 = relu6(batchnorm(P, p1 + ".4", y, ctx))
    y = F.conv2d(e1, P[p2 + ".0.weight"], P[p2 + ".0.bias"])
    y = relu6(batchnorm(P, p2 + ".1", y, ctx))
    y = F.conv2d(y, P[p2 + ".3.weight"], P[p2 + ".3.bias"], 1, 1, 1, groups=y.shape[1])
    y = relu6(batchnorm(P, p2 + ".4", y, ctx))
    y = F.conv2d(y, P[p2 + ".6.weight"], P[p2 + ".6.bias"])
    y = batchnorm(P, p2 + ".7", y, ctx)
    out = y + e1
    ctx.tap("x_enc", out)
    return out


# ---------------------------------------------------------------------------------------------
# attention front: spatial additive attention + hard attention  (utils.py:457-523,529-591,613-663)
# ---------------------------------------------------------------------------------------------
def spatial_attention(P, x, m, ctx):
    """x[B,24,H,W], m[B,1,H,W] in {0,1}.  utils.py:484-523 with h_t=None, multiply=True."""
    pre = "decoder.s_sp"
    b, c, h, w = x.shape
    xm = x * m
    base = F.conv2d(xm, P[pre + ".l_v.weight"], P[pre + ".l_v.bias"])
    ht = F.linear(xm.reshape(b, c, -1).mean(2), P[pre + ".l_h.weight"])
    base = base + ht[:, :, None, None]
    beta = F.conv2d(torch.tanh(base), P[pre + ".spatial_fc.1.weight"], P[pre + ".spatial_fc.1.bias"])
    beta = beta.masked_fill(m < 0.5, float("-inf")).reshape(b, 1, -1)
    msum = m.sum((1, 2, 3), keepdim=True)
    beta = torch.softmax(beta, 2).reshape(b, 1, h, w) * msum
    ctx.tap("s_sp.beta", beta)
    out = x + batchnorm(P, pre + ".bn", x * beta, ctx) * m
    ctx.tap("s_sp.out", out)
    return out


def mask_bn(P, pre, x, m, ctx):
    """maskBN (utils.py:568-591).  Denominator is sum(mask)+1; the running update uses the
    reference's inverted convention run*f + (1-f)*new with f = momentum."""
    b, c, h, w = x.shape
    w_, b_ = P[pre + ".weight"], P[pre + ".bias"]
    if ctx.bn_train:
        den = m.reshape(b, -1).sum(1) + 1
        xf, mf = x.reshape(b, c, -1), m.reshape(b, 1, -1)
        mean = ((xf * mf).sum(2) / den[:, None]).mean(0)
        var = ((((xf - mean[None, :, None]) ** 2) * mf).sum(2) / den[:, None]).mean(0)
        with torch.no_grad():
            f = BN_MOMENTUM
            rm, rv, nb = (ctx.new_buffers.get(pre + s, P[pre + s]) for s in
                          (".running_mean", ".running_var", ".num_batches_tracked"))
            ctx.new_buffers[pre + ".running_mean"] = rm * f + (1 - f) * mean
            ctx.new_buffers[pre + ".running_var"] = rv * f + (1 - f) * var
            ctx.new_buffers[pre + ".num_batches_tracked"] = nb + 1
    else:
        mean, var = P[pre + ".running_mean"], P[pre + ".running_var"]
    return (x - mean[None, :, None, None]) / torch.pow(var[None, :, None, None] + BN_EPS, 0.5) \
        * w_[None, :, None, None] + b_[None, :, None, None]


def hard_attention(P, s, sem, ins, ctx):
    """s[B,24,H,W], sem[B,1,H,W] float {0,1}, ins[B,n,H,W] {0,1}.  utils.py:631-663.
    Returns pro_split[B,n,H,W] (per-instance softmax over H*W, fully masked rows -> 0) and
    pro_merge[B,1,H,W]."""
    pre = "decoder.attend"
    b, n, h, w = ins.shape
    s = F.avg_pool2d(s, 3, 1, 1)
    e = F.conv2d(s, P[pre + ".l1.weight"], P[pre + ".l1.bias"])
    e = F.conv2d(torch.tanh(e), P[pre + ".attend_fc.1.weight"], P[pre + ".attend_fc.1.bias"], 1, 1)
    e = mask_bn(P, pre + ".bn", e, sem, ctx)
    merge = F.avg_pool2d(e, 3, 1, 1) * sem
    logits = merge.expand(-1, n, -1, -1).masked_fill(ins < 0.5, float("-inf")).reshape(b, n, -1)
    sm = torch.softmax(logits, 2)
    split = torch.where(torch.isnan(sm), torch.zeros_like(sm), sm).reshape(b, n, h, w)
    ctx.tap("attend.pro_merge", merge)
    ctx.tap("attend.pro_split", split)
    return split, merge


# ---------------------------------------------------------------------------------------------
# pyramid mask decoder  (attenet2.py:410-473, utils.py:816-892,946-1112,696-774)
# ---------------------------------------------------------------------------------------------
def position_code(rows: List[int], cols: List[int], factor: int):
    """utils.py:823-835: coarse cell + MSB-first binary code of (row%f, col%f)."""
    nb = int(math.log(factor, 2))
    pr = [r // factor for r in rows]
    pc = [c // factor for c in cols]
    codes = []
    for r, c in zip(rows, cols):
        rr, cc = r % factor, c % factor
        bits = [(rr >> (nb - 1 - k)) & 1 for k in range(nb)] + \
               [(cc >> (nb - 1 - k)) & 1 for k in range(nb)]
        codes.append(bits)
    return pr, pc, codes


def l0_pred(P, pre, x):
    y = F.conv2d(x, P[pre + ".l_i.weight"], P[pre + ".l_i.bias"], 1, 1)
    return F.conv2d(F.leaky_relu(y, 0.01), P[pre + ".last_fc.1.weight"], P[pre + ".last_fc.1.bias"], 1, 1)


def up_atten_level(P, lvl, xprev, xskip, rows, cols, sem_mask, gold, pred_prev, ctx, tag):
    """One UpDecoderLayer (utils.py:869-892) = resize + UpAttenLayer.forward (utils.py:1058-1112)
    + L0Layer.  Returns (x, pred[B,2,h,w], target[B,1,h,w])."""
    pre = "decoder.bone.upAtten%d" % lvl
    ua = pre + ".UpAtten"
    b, _, h, w = xskip.shape
    factor = sem_mask.shape[2] // h
    pr, pc, codes = position_code(rows, cols, factor)
    mask_all = F.max_pool2d(sem_mask, factor) if factor > 1 else sem_mask
    target = F.max_pool2d(gold, factor) if factor > 1 else gold

    def cross(x):
        x = block_ir(P, ua + ".cross.up_feature.0", x, ctx)
        # nn.Dropout2d *module*: follows the module's train/eval mode, not the `training` arg
        x = dropout2d(x, "%s.L%d.cross" % (tag, lvl), ctx, ctx.bn_train)
        return block_ir(P, ua + ".cross.up_feature.2", x, ctx)

    if lvl == 0:
        x = cross(xskip)
        up = None
    else:
        up = F.conv_transpose2d(xprev, P[ua + ".up.weight"], P[ua + ".up.bias"], stride=2)
        g = F.interpolate(pred_prev, (h, w), mode="bilinear", align_corners=False)
        g = torch.softmax(g, 1)[:, 1:2]
        x = torch.cat([cross(xskip), up * g], 1)
    nb = int(math.log(factor, 2))
    pos = torch.zeros(b, 2 * nb + 1, h, w, dtype=x.dtype)
    for i in range(b):
        pos[i, 2 * nb, pr[i], pc[i]] = 1.0
        for t in range(2 * nb):
            pos[i, t, pr[i], pc[i]] = float(codes[i][t])
    x = torch.cat([x, mask_all, pos], 1)
    ctx.tap("%s.L%d.concat" % (tag, lvl), x)
    x = F.conv2d(x, P[ua + ".conv1.0.weight"])
    x = F.relu(batchnorm(P, ua + ".conv1.1", x, ctx))
    x = dropout2d(x, "%s.L%d.d1" % (tag, lvl), ctx, ctx.training)
    x = block_ir(P, ua + ".dilation_part1.0", x, ctx)
    x = block_ir(P, ua + ".dilation_part1.1", x, ctx)
    if up is not None:
        x = x + up
    x = dropout2d(x, "%s.L%d.d2" % (tag, lvl), ctx, ctx.training)
    x = block_ir(P, ua + ".dilation_part2.0", x, ctx)
    x = block_ir(P, ua + ".dilation_part2.1", x, ctx)
    pred = l0_pred(P, pre + ".pred", x)
    ctx.tap("%s.L%d.x" % (tag, lvl), x)
    ctx.tap("%s.L%d.pred" % (tag, lvl), pred)
    return x, pred, target


def pyramid_decoder(P, feats, rows, cols, sem_mask, gold, ctx, tag):
    x1, x2, x3, x4, x5 = feats
    skips = [x5, x4, x3, x2, x1]
    x, pred = None, None
    preds, targets = [], []
    for lvl in range(5):
        x, pred, tgt = up_atten_level(P, lvl, x, skips[lvl], rows, cols, sem_mask, gold, pred, ctx, tag)
        preds.append(pred)
        targets.append(tgt)
    return targets, preds


# ---------------------------------------------------------------------------------------------
# losses  (dice.py:10-85, multi_loss.py:27-42, attenet2.py:86-141,204-290)
# ---------------------------------------------------------------------------------------------
def dice_fg_loss(logits, target, time=1, smooth=1.0):
    """1 - dice of the foreground channel, per image.  logits[B,2,h,w], target[B,1,h,w]."""
    p = torch.softmax(logits, 1)[:, 1]
    t = target[:, 0]
    num = (p * t).sum((1, 2))
    if time == 1:
        den = p.sum((1, 2)) + t.sum((1, 2))
    else:
        den = (p * p).sum((1, 2)) + (t * t).sum((1, 2))
    return 1.0 - (2 * num + smooth) / (den + smooth)


def focal_map(logits, target):
    """Per-pixel focal loss, gamma=2, alpha=0, detached modulating factor.  [B,h,w]."""
    p = torch.softmax(logits, 1)
    pt = p.detach()
    pc = p.clamp(1e-7, 1.0 - 1e-7)
    t = target[:, 0]
    f1 = -((1 - pt[:, 1]) ** FOCAL_GAMMA) * torch.log(pc[:, 1]) * t
    f0 = -((1 - pt[:, 0]) ** FOCAL_GAMMA) * torch.log(pc[:, 0]) * (1 - t)
    return f1 + f0


def ce_mean(logits, target):
    b = logits.shape[0]
    return F.cross_entropy(logits.permute(0, 2, 3, 1).reshape(-1, 2), target.reshape(-1).long())


def sem_losses(sem_out, sem_onehot):
    """Trainer-side semantic losses (model.py:255-269): CE + Dice(time=1, fg only, mean)."""
    ce = F.cross_entropy(sem_out.permute(0, 2, 3, 1).reshape(-1, sem_out.shape[1]),
                         sem_onehot.argmax(1).reshape(-1))
    dice = dice_fg_loss(sem_out, sem_onehot[:, 1:2].float(), time=1).mean()
    return ce, dice


@dataclass
class HeadState:
    baseline: float = 0.0     # REINFORCE EMA baseline (attenet2.py:47,266)


def atten_loss(preds, targets, alpha, s_t, training, state: HeadState):
    """attenet2.py:239-290.  Returns dict with the reference's four outputs plus `loss_finite`
    (the same loss without the -lambda_e*H term, whose value is NaN and whose gradient is
    exactly zero in the reference because of the min>max clamp at attenet2.py:77)."""
    b = alpha.shape[0]
    with torch.no_grad():
        eval_ce = ce_mean(preds[-1], targets[-1])
        eval_dice = dice_fg_loss(preds[-1], targets[-1], time=1)
    if not training:
        with torch.no_grad():
            loss = dice_fg_loss(preds[-1], targets[-1], time=2)
        return dict(loss=loss, loss_finite=loss, criterion=eval_ce + eval_dice,
                    ce=eval_ce, dice=eval_dice)
    loss_pred = 0
    dice_last = None
    for p, t, wl in zip(preds, targets, PYRAMID_W):
        dice_last = dice_fg_loss(p, t, time=1)
        focal = focal_map(p, t).reshape(b, -1).mean(1)
        loss_pred = loss_pred + (CE_WEIGHT * focal + dice_last) * wl
    with torch.no_grad():
        ce = ce_mean(preds[-1], targets[-1])
        log_p_y = -eval_dice
        state.baseline = 0.9 * state.baseline + 0.1 * float(log_p_y.mean())
    a = alpha.reshape(b, -1)
    picked = torch.stack([a[i, s_t[i]] for i in range(b)])
    loss_r = -(log_p_y - state.baseline) * torch.log(picked)
    criterion = ce + dice_last.sum().detach()
    per_img = LAMBDA_L * loss_pred + LAMBDA_R * loss_r
    loss_finite = LAMBDA_INS * per_img.sum() / b
    loss = loss_finite + float("nan")        # -lambda_e*H/b with H = NaN (attenet2.py:77)
    return dict(loss=loss, loss_finite=loss_finite, criterion=criterion, ce=eval_ce,
                dice=eval_dice, loss_pred=loss_pred, loss_r=loss_r)


# ---------------------------------------------------------------------------------------------
# instance head driver  (attenet2.py:357-407) and whole network (reseg.py:106-130)
# ---------------------------------------------------------------------------------------------
def instance_head(P, x_enc, sem_mask, ins, n_ins, feats, ctx: Ctx, state: HeadState,
                  selected_idx: List[List[int]], sample_fn=None):
    """selected_idx[b] = the per-image shuffled instance order (attenet2.py:349-355), injected.
    sample_fn(alpha[B,L]) -> list of B flat indices replaces multinomial in training mode; eval
    mode uses argmax (attenet2.py:321-326)."""
    b, _, h, w = x_enc.shape
    s = spatial_attention(P, x_enc, sem_mask, ctx)
    split, merge = hard_attention(P, s, sem_mask, ins.float(), ctx)
    nmin = int(min(int(v) for v in n_ins))
    max_iter = min(MAX_ITER, nmin) if ctx.training else nmin
    tot = dict(loss=0.0, loss_finite=0.0, criterion=0.0, ce=0.0, dice=0.0)
    trace = []
    for it in range(max_iter):
        idx = [selected_idx[i][it] for i in range(b)]
        gold = torch.stack([ins[i, idx[i]] for i in range(b)]).unsqueeze(1).float()
        alpha = torch.stack([split[i, idx[i]] for i in range(b)]).unsqueeze(1)
        with torch.no_grad():
            flat = alpha.reshape(b, -1)
            if ctx.training:
                s_t = [int(v) for v in sample_fn(flat)]
            else:
                s_t = [int(v) for v in flat.argmax(1)]
        rows, cols = [v // w for v in s_t], [v % w for v in s_t]
        tag = "it%d" % it
        targets, preds = pyramid_decoder(P, feats, rows, cols, sem_mask, gold, ctx, tag)
        out = atten_loss(preds, targets, alpha, s_t, ctx.training, state)
        tot["loss"] = tot["loss"] + out["loss"]
        tot["loss_finite"] = tot["loss_finite"] + out["loss_finite"]
        tot["criterion"] = tot["criterion"] + out["criterion"]
        tot["ce"] = tot["ce"] + out["ce"]
        tot["dice"] = tot["dice"] + out["dice"].mean()
        trace.append(dict(idx=idx, s_t=s_t, preds=preds, targets=targets))
    res = dict(
        ins_cost=(tot["loss"] / max_iter).mean(),
        ins_cost_finite=(tot["loss_finite"] / max_iter).mean(),
        criterion=(tot["criterion"] / max_iter).mean(),
        ins_ce_loss=tot["ce"] / max_iter,
        ins_dice_loss=tot["dice"] / max_iter,
        trace=trace,
    )
    return res


def reseg_forward(P, x, sem_onehot=None, ins=None, n_ins=None, *, use_instance_seg=True,
                  ctx: Optional[Ctx] = None, state: Optional[HeadState] = None,
                  selected_idx=None, sample_fn=None):
    """reseg.py:106-130.  Without GT returns dict(sem_out, sem_argmax); with GT adds the head's
    four scalars (and `ins_cost_finite`, see atten_loss)."""
    ctx = ctx or Ctx()
    x_dec, feats = unet(P, x, ctx)
    sem_out = sem_head(P, x_dec, ctx)
    if sem_onehot is not None:
        sem_argmax = sem_onehot.argmax(1).unsqueeze(1).float()
    else:
        sem_argmax = sem_out.argmax(1).unsqueeze(1).float()
    out = dict(sem_out=sem_out, sem_argmax=sem_argmax)
    if use_instance_seg and sem_onehot is not None:
        x_enc = ins_stems(P, x_dec, ctx)
        out.update(instance_head(P, x_enc, sem_argmax, ins, n_ins, feats, ctx,
                                 state or HeadState(), selected_idx, sample_fn))
    return out


# ---------------------------------------------------------------------------------------------
# by-name attention operators (dead at HEAD, covered with their own vectors)
# ---------------------------------------------------------------------------------------------
def sdp_attention(q, k, v, temperature, mask=None):
    """ScaledDotProductAttention.forward (utils.py:316-327), dropout off.
    q[Bh,Lq,d], k[Bh,Lk,d], v[Bh,Lk,dv]; mask[Bh,Lq,Lk] True = masked."""
    attn = torch.bmm(q, k.transpose(1, 2)) / temperature
    if mask is not None:
        attn = attn.masked_fill(mask, float("-inf"))
    attn = torch.softmax(attn, 2)
    return torch.bmm(attn, v), attn


def local_dilated_attention(Q, K, V, nomask, d):
    """Core of _ScalePDAttention.forward (utils.py:276-299) after the 1x1 projections.
    Q,K[B,dk,H,W], V[B,dv,H,W], nomask[B,1,H,W] (non-zero = masked key), dilation d.
    Returns [B,dv,H,W]: per pixel softmax over its 3x3 dilated neighbourhood."""
    b, dk, h, w = K.shape
    Kp, Vp, Mp = (F.pad(t, (d, d, d, d)) for t in (K, V, nomask))
    logits, vals = [], []
    for i in range(9):
        oy, ox = (i // 3) * d, (i % 3) * d
        kk = Kp[:, :, oy:oy + h, ox:ox + w]
        logits.append(((kk * Q).sum(1) * dk ** -0.5).masked_fill(Mp[:, 0, oy:oy + h, ox:ox + w] != 0,
                                                                  float("-inf")))
        vals.append(Vp[:, :, oy:oy + h, ox:ox + w])
    p = torch.softmax(torch.stack(logits, 1), 1)
    p = torch.where(torch.isnan(p), torch.zeros_like(p), p)
    return (torch.stack(vals, 1) * p[:, :, None]).sum(1)


def point_query_mask(q, enc):
    """Decoder.forward (utils.py:59-69): sigmoid(q[B,C] . enc[B,C,H*W]) -> [B,H*W]."""
    b, c = enc.shape[:2]
    return torch.sigmoid(torch.bmm(q.unsqueeze(1), enc.reshape(b, c, -1))).squeeze(1)


# ---------------------------------------------------------------------------------------------
# deterministic parameter / input synthesis shared by fixtures, tests and bench
# ---------------------------------------------------------------------------------------------
def state_dict_schema(use_instance_seg=True):
    """(name, shape) list in the reference's registration order (probe of reseg.ReSeg(2,...))."""
    S = []

    def bn(pre, c):
        S.extend([(pre + ".weight", (c,)), (pre + ".bias", (c,)), (pre + ".running_mean", (c,)),
                  (pre + ".running_var", (c,)), (pre + ".num_batches_tracked", ())])

    def v1(pre, ci, co):
        S.append((pre + ".conv.0.weight", (ci, 1, 3, 3))); bn(pre + ".conv.1", ci)
        S.append((pre + ".conv.3.weight", (co, ci, 1, 1))); bn(pre + ".conv.4", co)

    def ir(pre, ci, co):
        S.append((pre + ".conv.0.weight", (2 * ci, ci, 1, 1))); bn(pre + ".conv.1", 2 * ci)
        S.append((pre + ".conv.3.weight", (2 * ci, 1, 3, 3))); bn(pre + ".conv.4", 2 * ci)
        S.append((pre + ".conv.6.weight", (co, 2 * ci, 1, 1))); bn(pre + ".conv.7", co)

    def dbl(pre, ci, co):
        v1(pre + ".conv.down_conv_0", ci, co); v1(pre + ".conv.down_conv_1", co, co)

    def l0(pre, c):
        S.extend([(pre + ".l_i.weight", (c // 2, c, 3, 3)), (pre + ".l_i.bias", (c // 2,)),
                  (pre + ".last_fc.1.weight", (2, c // 2, 3, 3)), (pre + ".last_fc.1.bias", (2,))])

    dbl("base.inc.conv", 21, 32)
    for i, c in enumerate((32, 64, 128, 256)):
        dbl("base.down%d.mpconv" % (i + 1), c, c)
    for i, c in enumerate((512, 256, 128, 64)):
        S.extend([("base.up%d.up.weight" % (i + 1), (c, c // 2, 2, 2)), ("base.up%d.up.bias" % (i + 1), (c // 2,))])
        dbl("base.up%d.conv" % (i + 1), c, c // 2)
    l0("decoder.pred", 64)
    skip = (512, 256, 128, 64, 32)
    outc = (256, 128, 64, 32, 32)
    for lvl in range(5):
        pre = "decoder.bone.upAtten%d" % lvl
        ua = pre + ".UpAtten"
        nn_ = 2 * (4 - lvl) + 2
        if lvl > 0:
            cin_up = outc[lvl - 1]
            S.extend([(ua + ".up.weight", (cin_up, outc[lvl], 2, 2)), (ua + ".up.bias", (outc[lvl],))])
        ir(ua + ".cross.up_feature.0", skip[lvl], outc[lvl])
        ir(ua + ".cross.up_feature.2", outc[lvl], outc[lvl] - nn_)
        cin1 = outc[lvl] if lvl == 0 else 2 * outc[lvl]
        S.append((ua + ".conv1.0.weight", (outc[lvl], cin1, 1, 1))); bn(ua + ".conv1.1", outc[lvl])
        for part in ("dilation_part1", "dilation_part2"):
            for j in (0, 1):
                ir("%s.%s.%d" % (ua, part, j), outc[lvl], outc[lvl])
        l0(pre + ".pred", outc[lvl])
    S.extend([("decoder.s_sp.l_v.weight", (1, 24, 1, 1)), ("decoder.s_sp.l_v.bias", (1,)),
              ("decoder.s_sp.l_h.weight", (1, 24)),
              ("decoder.s_sp.spatial_fc.1.weight", (1, 1, 1, 1)), ("decoder.s_sp.spatial_fc.1.bias", (1,))])
    bn("decoder.s_sp.bn", 24)
    S.extend([("decoder.attend.l1.weight", (12, 24, 1, 1)), ("decoder.attend.l1.bias", (12,)),
              ("decoder.attend.l2.weight", (12, 24)),
              ("decoder.attend.attend_fc.1.weight", (1, 12, 3, 3)), ("decoder.attend.attend_fc.1.bias", (1,))])
    bn("decoder.attend.bn", 1)
    S.extend([("decoder.embedding.sigma.0.weight", (12, 24)), ("decoder.embedding.sigma.0.bias", (12,)),
              ("decoder.embedding.sigma.2.weight", (1, 12)), ("decoder.embedding.sigma.2.bias", (1,))])
    S.extend([("channelAttend.fc.0.weight", (16, 32)), ("channelAttend.fc.0.bias", (16,)),
              ("channelAttend.fc.2.weight", (32, 16)), ("channelAttend.fc.2.bias", (32,)),
              ("sem_seg_output.weight", (2, 32, 1, 1)), ("sem_seg_output.bias", (2,))])
    if use_instance_seg:
        p1, p2 = "ins_seg_output_1", "ins_seg_output_2"
        S.extend([(p1 + ".0.weight", (32, 1, 3, 3)), (p1 + ".0.bias", (32,))]); bn(p1 + ".1", 32)
        S.extend([(p1 + ".3.weight", (24, 32, 1, 1)), (p1 + ".3.bias", (24,))]); bn(p1 + ".4", 24)
        S.extend([(p2 + ".0.weight", (48, 24, 1, 1)), (p2 + ".0.bias", (48,))]); bn(p2 + ".1", 48)
        S.extend([(p2 + ".3.weight", (48, 1, 3, 3)), (p2 + ".3.bias", (48,))]); bn(p2 + ".4", 48)
        S.extend([(p2 + ".6.weight", (24, 48, 1, 1)), (p2 + ".6.bias", (24,))]); bn(p2 + ".7", 24)
    return S


def synth_state_dict(seed=23, use_instance_seg=True):
    """Deterministic, torch-RNG-independent weights: every tensor from its own
    numpy RandomState(crc32(name) ^ seed) stream, scaled like a He/fan-in init so activations stay
    O(1) through ~50 layers.  BN gamma in [0.5,1.5], beta small, running stats non-trivial."""
    import zlib
    import numpy as np
    sd = {}
    for name, shape in state_dict_schema(use_instance_seg):
        rs = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
            continue
        if name.endswith("running_mean"):
            a = rs.uniform(-0.1, 0.1, shape)
        elif name.endswith("running_var"):
            a = rs.uniform(0.5, 1.5, shape)
        elif len(shape) == 1 and name.endswith(".weight"):       # BN gamma
            a = rs.uniform(0.5, 1.5, shape)
        elif len(shape) == 1:                                      # biases / BN beta
            a = rs.uniform(-0.1, 0.1, shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            if len(shape) == 4 and name.endswith("up.weight"):     # ConvTranspose2d [Cin,Cout,2,2]
                fan_in = shape[0]
            a = rs.standard_normal(shape) * math.sqrt(2.0 / max(fan_in, 1))
        sd[name] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def synth_batch(batch, height, width, seed=0, max_objects=32, kmin=3, kmax=8):
    """Synthetic inputs shaped like the reference's collate output (dataset.py:368-378):
    x[B,21,H,W] f32, sem one-hot [B,2,H,W] i64, ins [B,32,H,W] i64 (k non-overlapping
    rectangles/ellipses in channels 0..k-1), N[B,1] i32.  numpy RandomState only."""
    import numpy as np
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((batch, 21, height, width)).astype(np.float32)
    ins = np.zeros((batch, max_objects, height, width), dtype=np.int64)
    n = np.zeros((batch, 1), dtype=np.int32)
    yy, xx = np.mgrid[0:height, 0:width]
    for b in range(batch):
        k = int(rs.randint(kmin, kmax + 1))
        occupied = np.zeros((height, width), dtype=bool)
        placed = 0
        tries = 0
        while placed < k and tries < 200:
            tries += 1
            hh = int(rs.randint(max(2, height // 10), max(3, height // 3)))
            ww = int(rs.randint(max(2, width // 10), max(3, width // 3)))
            y0 = int(rs.randint(0, height - hh + 1))
            x0 = int(rs.randint(0, width - ww + 1))
            if rs.rand() < 0.5:
                m = (yy >= y0) & (yy < y0 + hh) & (xx >= x0) & (xx < x0 + ww)
            else:
                cy, cx = y0 + hh / 2.0, x0 + ww / 2.0
                m = ((yy + 0.5 - cy) / (hh / 2.0)) ** 2 + ((xx + 0.5 - cx) / (ww / 2.0)) ** 2 <= 1.0
            if m.sum() < 4 or (m & occupied).any():
                continue
            ins[b, placed][m] = 1
            occupied |= m
            placed += 1
        n[b, 0] = placed
        # correlate the image with the masks a little so heads see structure
        x[b, :3] += occupied[None].astype(np.float32)
    fg = ins.sum(1) > 0
    sem = np.stack([~fg, fg], 1).astype(np.int64)
    return (torch.from_numpy(x), torch.from_numpy(sem), torch.from_numpy(ins), torch.from_numpy(n))
