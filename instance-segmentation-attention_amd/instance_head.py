"""Attention mask-prediction head composed from engine ops.

Reference: reseg.py:78-102,122-126 (stems), attenet2.py:357-407 (DecoderLayer.forward driver),
attenet2.py:443-473 (AttenDecoder), utils.py:457-663 (attention layers), utils.py:816-1112
(UpDecoderLayer / UpAttenLayer / L0Layer), attenet2.py:86-141,204-290 (losses).

What changes vs. the reference's control flow (same results, MI355X-first):
  * nothing leaves the device inside the iteration loop: the reference copies alpha to the CPU,
    samples there and rebuilds Python int lists (attenet2.py:306-332); here argmax / injected samples
    stay in a device int32 vector that the position-code kernel reads;
  * HardAttentionLayer's [B,32,H,W] softmax (utils.py:648-655) is evaluated only for the instance
    each iteration actually selects (B rows, not 32*B);
  * concat order inside the decoder is [gated_up | cross | mask_all | position] so every slice
    starts 16-byte aligned; conv1's weight columns are permuted to match when packed.
"""
import math
import os

import torch

from . import lib as L
from .engine import Act, Engine, Pro

SKIP_CH = (512, 256, 128, 64, 32)
OUT_CH = (256, 128, 64, 32, 32)
FACTORS = (16, 8, 4, 2, 1)
PYRAMID_W = (16.0, 8.0, 4.0, 2.0, 1.0)     # config.py:51
CE_WEIGHT = 10.0                            # config.py:17
LAMBDA_L, LAMBDA_R = 0.5, 2.0               # config.py:45-46
MAX_ITER = 2                                # config.py:56
DROP_RATE = 0.5                             # config.py:64
ROW_PART = 64 * 4                           # floats of chunk partials per softmax row (ISA_ROW_CHUNKS * 4, isa_kernels.h)


class InstanceHead:
    def __init__(self, net):
        self.net = net
        self.E: Engine = net.E
        self.drop_rate = DROP_RATE
        # concurrent HIP streams for the pyramid passes: 1 = none, 2 = odd iterations on a side stream, 3 = cross
        # chains on the step's stream + one side stream per iteration's level chain (see forward)
        self.streams = int(os.environ.get("ISA_STREAMS", "2"))
        # all decoder iterations of a train-mode pass as ONE batch of max_iter * B images with a BatchNorm statistic group
        # per iteration (attenet2.py:384-399 runs them one after the other through the same weights): half the launches,
        # twice the work per launch on the low-resolution levels.  0 = one pass per iteration (A/B, bisecting).
        self.batch_iters = os.environ.get("ISA_BATCH_ITERS", "1") != "0"
        self.injected_masks = None       # {(iteration, level, "cross"|"d1"|"d2"): [n, C] keep/(1-p) mask}: parity by injection
        self.sample_in_training = True   # False: greedy point (argmax) also in training; parity tests inject s_t instead
        import ctypes as _C
        self._level_w = (_C.c_float * 5)(*PYRAMID_W)
        self.baseline = None            # device scalar (REINFORCE EMA baseline, attenet2.py:47,266)

    # ------------------------------------------------------------------ stems (reseg.py:78-102)
    def stems(self, x_dec: Act) -> Act:
        E, net = self.E, self.net
        n, h, w = x_dec.n, x_dec.h, x_dec.w
        p1, p2 = "ins_seg_output_1", "ins_seg_output_2"
        y = E.new_act(n, h, w, 32)
        _, s = E.dwconv(x_dec, p1 + ".0.weight", y, bias=p1 + ".0.bias", stats=True)
        y = E.bn(y, s, p1 + ".1", L.ACT_RELU6)
        z = E.new_act(n, h, w, 24)
        _, s = E.conv(y, p1 + ".3.weight", z, bias=p1 + ".3.bias", stats=True)
        e1 = E.new_act(n, h, w, 24)
        E.bn_out(z, s, p1 + ".4", L.ACT_RELU6, e1)
        y = E.new_act(n, h, w, 48)
        _, s = E.conv(e1, p2 + ".0.weight", y, bias=p2 + ".0.bias", stats=True)
        y = E.bn(y, s, p2 + ".1", L.ACT_RELU6)
        z = E.new_act(n, h, w, 48)
        _, s = E.dwconv(y, p2 + ".3.weight", z, bias=p2 + ".3.bias", stats=True)
        z = E.bn(z, s, p2 + ".4", L.ACT_RELU6)
        y = E.new_act(n, h, w, 24)
        _, s = E.conv(z, p2 + ".6.weight", y, bias=p2 + ".6.bias", stats=True)
        x_enc = E.new_act(n, h, w, 24)
        return E.bn_out(y, s, p2 + ".7", L.ACT_NONE, x_enc, res=e1)

    # ------------------------------------------------------------------ a7 SpatialAttentionLayer
    def spatial_attention(self, x: Act, sem: torch.Tensor) -> Act:
        E, P = self.E, self.E.params
        n, c, Lp = x.n, x.c, x.h * x.w
        pre = "decoder.s_sp"
        dot, chansum = E.scratch(n * Lp), E.scratch(n * c)
        L.check(E.lib.isa_mask_dot(x.d(), L.ptr(sem), P.ptr(pre + ".l_v.weight"), P.ptr(pre + ".l_v.bias"),
                                   L.ptr(dot), L.ptr(chansum), E.st()), "isa_mask_dot")
        beta, rowstat = E.f32(n * Lp), E.f32(n * 4)
        L.check(E.lib.isa_sp_softmax(L.ptr(dot), L.ptr(sem), L.ptr(chansum), P.ptr(pre + ".l_h.weight"),
                                     P.ptr(pre + ".spatial_fc.1.weight"), P.ptr(pre + ".spatial_fc.1.bias"),
                                     n, c, Lp, L.ptr(beta), L.ptr(rowstat), L.ptr(E.f32(n * ROW_PART)), E.st()),
                "isa_sp_softmax")
        scale, shift, mean, invstd = (E.f32(c) for _ in range(4))
        stats = E.scratch(2 * c * 8)
        if E.bn_train:
            L.check(E.lib.isa_scaled_stats(x.d(), L.ptr(beta), L.ptr(stats), E.st()), "isa_scaled_stats")
            P.int_buffers[pre + ".bn.num_batches_tracked"] += 1
        L.check(E.lib.isa_bn_finalize(L.ptr(stats) if E.bn_train else None, float(n * Lp), P.ptr(pre + ".bn.weight"),
                                      P.ptr(pre + ".bn.bias"), P.ptr(pre + ".bn.running_mean"),
                                      P.ptr(pre + ".bn.running_var"), E.BN_MOMENTUM, E.BN_EPS, L.ptr(scale),
                                      L.ptr(shift), L.ptr(mean), L.ptr(invstd), c, 1, 1, E.st()), "isa_bn_finalize")
        out = E.new_act(n, x.h, x.w, c)
        L.check(E.lib.isa_sp_apply(x.d(), L.ptr(beta), L.ptr(sem), L.ptr(scale), L.ptr(shift), out.d(), E.st()),
                "isa_sp_apply")
        if E.record:
            train = 1 if E.bn_train else 0

            def bwd():
                scr = E.scratch(3 * c + 2 * ((n + 3) // 4 * 4) + 2 * n * Lp)
                acc = E.grads.claim(x, E)
                L.check(E.lib.isa_sp_bwd(E.grads.grad_of(out).d(), x.d(), L.ptr(beta), L.ptr(sem), L.ptr(dot),
                                         L.ptr(rowstat), L.ptr(chansum), L.ptr(scale), L.ptr(mean), L.ptr(invstd),
                                         P.ptr(pre + ".l_v.weight"), P.ptr(pre + ".l_h.weight"),
                                         P.ptr(pre + ".spatial_fc.1.weight"), float(n * Lp), train, L.ptr(scr),
                                         E.grads.grad_of(x).d(), acc, P.gptr(pre + ".bn.weight"), P.gptr(pre + ".bn.bias"),
                                         P.gptr(pre + ".l_v.weight"), P.gptr(pre + ".l_v.bias"), P.gptr(pre + ".l_h.weight"),
                                         P.gptr(pre + ".spatial_fc.1.weight"), P.gptr(pre + ".spatial_fc.1.bias"),
                                         E.st()), "isa_sp_bwd")
            E.tape.append(bwd)
        return out

    # ------------------------------------------------------------------ a8/a9 HardAttentionLayer front
    def hard_attention_scores(self, s: Act, sem: torch.Tensor) -> torch.Tensor:
        """Returns pro_merge as an fp32 [n, h*w] map."""
        E, P = self.E, self.E.params
        n, h, w = s.n, s.h, s.w
        pre = "decoder.attend"
        sp = E.new_act(n, h, w, s.c)
        L.check(E.lib.isa_avgpool3(s.d(), None, sp.d(), 0, E.st()), "isa_avgpool3")
        if E.record:
            def bwd_pool():      # 3x3 mean with zero padding is self-adjoint
                acc = E.grads.claim(s, E)
                L.check(E.lib.isa_avgpool3(E.grads.grad_of(sp).d(), None, E.grads.grad_of(s).d(), acc, E.st()),
                        "isa_avgpool3(bwd)")
            E.tape.append(bwd_pool)
        e1 = E.new_act(n, h, w, 12)
        E.conv(sp, pre + ".l1.weight", e1, bias=pre + ".l1.bias")
        e2 = E.new_act(n, h, w, 1)
        e1t = E.act_out(e1, L.ACT_TANH, E.new_act(n, h, w, 12))        # tanh once, not once per tap
        E.conv(e1t, pre + ".attend_fc.1.weight", e2, taps=9, bias=pre + ".attend_fc.1.bias")
        am, v, mean_var = E.f32(2 * n), E.f32(n), E.f32(2)
        train = 1 if E.bn_train else 0
        rm, rv = P.ptr(pre + ".bn.running_mean"), P.ptr(pre + ".bn.running_var")
        if train:
            L.check(E.lib.isa_maskbn_stats(e2.d(), L.ptr(sem), None, L.ptr(am), E.st()), "isa_maskbn_stats")
        L.check(E.lib.isa_maskbn_finalize(L.ptr(am), L.ptr(v), n, 0, L.ptr(mean_var), rm, rv, E.BN_MOMENTUM, train,
                                          E.st()), "isa_maskbn_finalize")
        if train:
            L.check(E.lib.isa_maskbn_stats(e2.d(), L.ptr(sem), L.ptr(mean_var), L.ptr(v), E.st()), "isa_maskbn_stats")
            L.check(E.lib.isa_maskbn_finalize(L.ptr(am), L.ptr(v), n, 1, L.ptr(mean_var), rm, rv, E.BN_MOMENTUM, 1,
                                              E.st()), "isa_maskbn_finalize")
            P.int_buffers[pre + ".bn.num_batches_tracked"] += 1
        merge = E.f32(n * h * w)
        L.check(E.lib.isa_maskbn_apply_pool(e2.d(), L.ptr(sem), L.ptr(mean_var), P.ptr(pre + ".bn.weight"),
                                            P.ptr(pre + ".bn.bias"), E.BN_EPS, L.ptr(merge), E.st()),
                "isa_maskbn_apply_pool")
        self.dmerge = None
        if E.record:
            dmerge = E.scratch(n * h * w)       # zero at step start; filled by the per-iteration closures
            self.dmerge = dmerge

            def bwd_bn():
                red4, k2 = E.scratch(4), E.f32(2)
                acc = E.grads.claim(e2, E)
                L.check(E.lib.isa_maskbn_bwd(e2.d(), L.ptr(sem), L.ptr(dmerge), L.ptr(mean_var), P.ptr(pre + ".bn.weight"),
                                             E.BN_EPS, L.ptr(am), train, L.ptr(red4), L.ptr(k2), P.gptr(pre + ".bn.weight"),
                                             P.gptr(pre + ".bn.bias"), E.grads.grad_of(e2).d(), acc, E.st()),
                        "isa_maskbn_bwd")
            E.tape.append(bwd_bn)
        return merge

    # ------------------------------------------------------------------ pyramid decoder
    def _draw_masks(self, n, max_iter, training):
        """All Dropout2d masks of a step from ONE torch.rand draw (3 per level and iteration: utils.py:869-892).
        Per-mask draws were ~120 five-microsecond launches per step."""
        self._mask_pool, self._mask_cursor = None, 0
        if self.drop_rate <= 0 or not (training or self.E.bn_train) or self.injected_masks is not None:
            return
        total = max_iter * sum(3 * n * oc for oc in OUT_CH)
        keep = 1.0 - self.drop_rate
        u = torch.rand(total, device=self.E.device)          # torch supplies the random numbers (graph-safe generator)
        self._mask_pool = torch.empty_like(u)
        L.check(self.E.lib.isa_dropout_mask(L.ptr(u), total, keep, L.ptr(self._mask_pool), self.E.st()), "isa_dropout_mask")

    def _masks_batched(self, n, G, training):
        """Dropout2d masks of a pass that batches the G iterations: per level {cross, d1, d2} as [G*n, C] arrays
        ([iteration][image] rows, the order the batched tensors use).  Inactive d1 / d2 masks are absent; an inactive
        cross mask is all ones (the broadcast materialising pass always takes a per-image scale)."""
        E = self.E
        keep = 1.0 - self.drop_rate
        act_cross = self.drop_rate > 0 and E.bn_train
        act_d = self.drop_rate > 0 and training
        inj = self.injected_masks
        out = []
        pool, cur = None, 0
        if inj is None and (act_cross or act_d):
            total = sum((int(act_cross) + 2 * int(act_d)) * G * n * oc for oc in OUT_CH)
            u = torch.rand(total, device=E.device)
            pool = torch.empty_like(u)
            L.check(E.lib.isa_dropout_mask(L.ptr(u), total, keep, L.ptr(pool), E.st()), "isa_dropout_mask")

        def take(lvl, kind, oc):
            nonlocal cur
            if inj is not None:
                return torch.cat([inj[it, lvl, kind].to(E.device, torch.float32).reshape(n, oc) for it in range(G)]).contiguous()
            m = pool[cur:cur + G * n * oc].view(G * n, oc)
            cur += G * n * oc
            return m

        for lvl, oc in enumerate(OUT_CH):
            masks = {}
            if act_cross:
                masks["cross"] = take(lvl, "cross", oc)
            else:
                ones = getattr(self, "_ones", None)
                if ones is None or ones.numel() < G * n * oc:
                    ones = self._ones = torch.ones(G * n * max(OUT_CH), device=E.device)
                masks["cross"] = ones[:G * n * oc].view(G * n, oc)
            if act_d:
                masks["d1"] = take(lvl, "d1", oc)
                masks["d2"] = take(lvl, "d2", oc)
            out.append(masks)
        return out

    def _drop_mask(self, n, c, active, key=None):
        if not active or self.drop_rate <= 0:
            return None
        if self.injected_masks is not None:
            return self.injected_masks[key].to(self.E.device, torch.float32).reshape(n, c).contiguous()
        m = self._mask_pool[self._mask_cursor:self._mask_cursor + n * c].view(n, c)
        self._mask_cursor += n * c
        return m

    def level_cross(self, lvl, skip: Act, masks, G=1):
        """The `cross` branch of one UpAttenLayer (utils.py:1058-1075): IR -> Dropout2d(module) -> IR on the backbone
        feature.  It does not depend on the previous level, so it runs on its own stream next to the level chain.
        Returns the concat buffer (its cross slice written).
        G > 1 (all decoder iterations in one pass): the first block sees the same input and weights in every
        iteration, so it is evaluated ONCE (E.repeated(G): running statistics advance G times) and only its Dropout2d
        differs - the materialising pass writes G masked copies (masks["cross"] is [G*n, C])."""
        E, net = self.E, self.net
        ua = "decoder.bone.upAtten%d.UpAtten" % lvl
        n, h, w = skip.n, skip.h, skip.w
        out_ch, f = OUT_CH[lvl], FACTORS[lvl]
        nb = int(math.log2(f))
        naux = 2 * nb + 2
        ccross = out_ch - naux
        width = out_ch if lvl == 0 else 2 * out_ch
        cat = E.new_act(G * n, h, w, width, ld=width, groups=G)
        cross_off = 0 if lvl == 0 else out_ch
        mid = E.new_act(G * n, h, w, out_ch, groups=G)
        if G > 1:
            with E.repeated(G):
                net.block_ir(skip, ua + ".cross.up_feature.0", mid, oscale=masks["cross"])
        else:
            net.block_ir(skip, ua + ".cross.up_feature.0", mid, oscale=masks.get("cross"))
        net.block_ir(mid, ua + ".cross.up_feature.2", cat.slice(cross_off, ccross))
        return cat

    def level_main(self, lvl, cat: Act, x_prev: Act, pred_prev: Act, mask_all, s_t, W_full, training, masks):
        """The rest of one UpDecoderLayer (utils.py:869-892) once the cross slice of `cat` is there.  Returns (x, pred).
        With cat.groups > 1 the batch holds all decoder iterations ([iteration][image]); mask_all is the one map per
        image they share, s_t and the masks are per (iteration, image)."""
        E, net = self.E, self.net
        pre = "decoder.bone.upAtten%d" % lvl
        ua = pre + ".UpAtten"
        n, h, w = cat.n, cat.h, cat.w
        out_ch, f = OUT_CH[lvl], FACTORS[lvl]
        nb = int(math.log2(f))
        naux = 2 * nb + 2
        ccross = out_ch - naux
        cross_off = 0 if lvl == 0 else out_ch
        up = None
        if lvl > 0:
            up = E.like(cat, out_ch)
            E.conv(x_prev, ua + ".up.weight", up, bias=ua + ".up.bias", transposed=True)
            gated = cat.slice(0, out_ch)
            gmap = E.f32(n * h * w) if E.record else None
            L.check(E.lib.isa_gate(up.d(), pred_prev.d(), gated.d(), L.ptr(gmap), E.st()), "isa_gate")
            if E.record:
                def bwd_gate(up=up, pred_prev=pred_prev, gated=gated, gmap=gmap):
                    du = E.scratch(n * h * w)
                    acc_up = E.grads.claim(up, E)
                    acc_pred = E.grads.claim(pred_prev, E)
                    L.check(E.lib.isa_gate_bwd(E.grads.grad_of(gated).d(), up.d(), L.ptr(gmap), E.grads.grad_of(up).d(),
                                               acc_up, L.ptr(du), E.grads.grad_of(pred_prev).d(), acc_pred, E.st()),
                            "isa_gate_bwd")
                E.tape.append(bwd_gate)
        aux = cat.slice(cross_off + ccross, naux)
        L.check(E.lib.isa_concat_aux(aux.d(), L.ptr(mask_all), L.ptr(s_t), W_full, f, nb, n // cat.groups, E.st()),
                "isa_concat_aux")
        if lvl == 0:
            kmap = None
        else:   # physical [gated(out) | cross(ccross) | aux] -> source [cross | gated | aux]
            kmap = [ccross + k for k in range(out_ch)] + list(range(ccross)) + \
                   [ccross + out_ch + k for k in range(naux)]
        y = E.like(cat, out_ch)
        _, s = E.conv(cat, ua + ".conv1.0.weight", y, stats=True, kmap=kmap)
        x = E.like(cat, out_ch)
        E.bn_out(y, s, ua + ".conv1.1", L.ACT_RELU, x, bscale=masks.get("d1"))
        x2 = E.like(cat, out_ch)
        net.block_ir(x, ua + ".dilation_part1.0", x2)
        x3 = E.like(cat, out_ch)
        self._block_ir_ext(x2, ua + ".dilation_part1.1", x3, res2=up, oscale=masks.get("d2"))
        x4 = E.like(cat, out_ch)
        net.block_ir(x3, ua + ".dilation_part2.0", x4)
        x5 = E.like(cat, out_ch)
        net.block_ir(x4, ua + ".dilation_part2.1", x5)
        # L0Layer (utils.py:696-774): conv3x3 -> LeakyReLU -> conv3x3
        hmid = E.like(cat, out_ch // 2)
        E.conv(x5, pre + ".pred.l_i.weight", hmid, taps=9, bias=pre + ".pred.l_i.bias")
        pred = E.like(cat, 2)
        hact = E.act_out(hmid, L.ACT_LEAKY, E.like(cat, out_ch // 2))     # once, not once per tap
        E.conv(hact, pre + ".pred.last_fc.1.weight", pred, taps=9, bias=pre + ".pred.last_fc.1.bias")
        return x5, pred

    def _block_ir_ext(self, x, pre, out, res2=None, oscale=None):
        """InvertedResidual whose materialising pass also adds `res2` and applies a Dropout2d mask
        to the sum (utils.py:1104-1110: x = dilation_part1(x); x = x + x1; x = dropout2d(x))."""
        E = self.E
        chid = E.params.shapes[pre + ".conv.0.weight"][0]
        cout = E.params.shapes[pre + ".conv.6.weight"][0]
        y1 = E.like(x, chid)
        _, s1 = E.conv(x, pre + ".conv.0.weight", y1, stats=True)
        y1 = E.bn(y1, s1, pre + ".conv.1", L.ACT_RELU6)
        y2 = E.like(x, chid)
        _, s2 = E.dwconv(y1, pre + ".conv.3.weight", y2, stats=True)
        y2 = E.bn(y2, s2, pre + ".conv.4", L.ACT_RELU6)
        y3 = E.like(x, cout)
        _, s3 = E.conv(y2, pre + ".conv.6.weight", y3, stats=True)
        return E.bn_out(y3, s3, pre + ".conv.7", L.ACT_NONE, out, res=x if x.c == cout else None,
                        res2=res2, oscale=oscale)

    # ------------------------------------------------------------------ driver (attenet2.py:357-407)
    def forward(self, x_dec: Act, feats, sem_map: torch.Tensor, ins: torch.Tensor, n_ins, training: bool,
                selected_idx, injected_s_t=None, capture=None, idx_dev=None):
        """sem_map: fp32 [n, h*w] {0,1}; ins: int64 [n,32,h,w] on device; n_ins: host ints;
        selected_idx[b]: instance order; injected_s_t: optional list of int32 device vectors (train
        sampling is injected, SURVEY §7 'RNG parity').  Returns dict of per-iteration records."""
        E = self.E
        n, H, W = x_dec.n, x_dec.h, x_dec.w
        Lp = H * W
        nobj = ins.shape[1]
        x_enc = self.stems(x_dec)
        s = self.spatial_attention(x_enc, sem_map)
        merge = self.hard_attention_scores(s, sem_map)
        nmin = int(min(int(v) for v in n_ins))
        max_iter = min(MAX_ITER, nmin) if training else nmin
        mask_all = []
        for f in FACTORS:
            if f == 1:
                mask_all.append(sem_map)
            else:
                m = E.f32(n * (H // f) * (W // f))
                L.check(E.lib.isa_pool_target(None, None, L.ptr(sem_map), nobj, n, H, W, f, L.ptr(m), n, E.st()),
                        "isa_pool_target(sem)")
                mask_all.append(m)
        if idx_dev is None:          # [max_iter, n] int32 instance order; pre-staged by the graph-captured step
            idx_dev = self.order_tensor(selected_idx, max_iter, n).to(E.device)
        skips = [feats[4], feats[3], feats[2], feats[1], feats[0]]
        iters = []
        scal = E.scratch(4)          # {ins_cost (finite part), criterion, ins_ce_loss, ins_dice_loss}
        if self.baseline is None:
            self.baseline = torch.zeros(1, dtype=torch.float32, device=E.device)
        if capture is not None:
            capture.update(x_enc=x_enc, s_sp=s, merge=merge)
        if self.batch_iters and E.bn_train and 2 <= max_iter <= 4:
            return self._forward_batched(x_enc, merge, skips, sem_map, ins, n, H, W, nobj, max_iter, training, idx_dev,
                                         injected_s_t, capture, mask_all, scal)
        self._draw_masks(n, max_iter, training)
        # ---- per-iteration front: instance softmax, glimpse point, pyramid targets (main stream) ----------------
        pre = []
        for it in range(max_iter):
            idx = idx_dev[it]
            alpha, rowstat = E.f32(n * Lp), E.f32(2 * n)
            L.check(E.lib.isa_ins_softmax(L.ptr(merge), L.ptr(ins), L.ptr(idx), n, nobj, Lp, L.ptr(alpha),
                                          L.ptr(rowstat), n, L.ptr(E.f32(n * ROW_PART)), E.st()), "isa_ins_softmax")
            if injected_s_t is not None:
                s_t = injected_s_t[it]
            else:
                # attenet2.py:304-321 `sample`: training draws s_t ~ Multinomial(alpha), eval takes the argmax.
                # The draw is argmax(alpha / Exp(1)) (the exponential-race form torch.multinomial itself uses
                # for one sample); torch supplies the random numbers (graph-safe generator), the row argmax is ours.
                race = None
                if training and self.sample_in_training:
                    race = torch.empty(n, Lp, dtype=torch.float32, device=alpha.device).exponential_(1.0)
                s_t = E.arena.alloc((n,), torch.int32)
                L.check(E.lib.isa_row_argmax(L.ptr(alpha), L.ptr(race), n, Lp, L.ptr(s_t), E.st()), "isa_row_argmax")
            targets = []
            for f in FACTORS:
                t = E.f32(n * (H // f) * (W // f))
                L.check(E.lib.isa_pool_target(L.ptr(ins), L.ptr(idx), None, nobj, n, H, W, f, L.ptr(t), n, E.st()),
                        "isa_pool_target")
                targets.append(t)
            sums_all = E.scratch(5 * 8 * n)                 # [level][image][8], contiguous for isa_head_loss
            pre.append((idx, alpha, s_t, targets, sums_all))
        # ---- the pyramid passes.  The iterations share weights and backbone features but not data (attenet2.py:384-
        # 399 runs them one after the other), and inside an iteration the cross branch of every level only reads the
        # backbone.  Most of these launches are too small to fill 256 CUs (16x16 ... 64x64 maps), so independent
        # chains run on separate HIP streams.  streams == 2 (default): odd iterations run whole on side stream 1, one
        # fork and one join per pass (34.3 -> 30.0 ms/step at the BASELINE shape under hipGraph replay).  streams == 3:
        # the cross chains of all iterations stay on the step's own stream, the level chain of iteration i runs on
        # side stream 1 + i % 2, one edge per level; measured slower (31.9 ms: every extra cross-stream edge costs
        # more than the finer overlap returns; cross chains on their own side streams 32.4 - 33.9 ms).  Every edge has
        # the origin stream at one end: hipGraph capture of side -> side edges crashes inside ROCm 7.2's
        # hipStreamEndCapture (scripts/graph_probe.py).
        # What the sequential order guaranteed is restored explicitly: BatchNorm running statistics of work that
        # runs beside iteration 0 are applied after the join in iteration order (isa_bn_running_update), gradients of
        # backbone features read from a side stream are accumulated per stream and merged (GradBook.merge_shared),
        # loss assembly stays on the origin stream.
        nstreams = self.streams if max_iter >= 2 else 1
        assert nstreams in (1, 2, 3), "ISA_STREAMS must be 1, 2 or 3"
        for sk in skips:
            E.grads.share(sk)
        if E.record and nstreams > 1:
            E.tape.append(lambda: E.grads.merge_shared(E))      # runs after the reversed forks have joined stream 0
        forked = [] if nstreams == 1 else ([1] if nstreams == 2 else [1, 2][:max_iter])
        for sid in forked:                  # all forks up front: a fork issued later would wait for iteration 0's chain
            E.sync(0, sid)
        masks_of = {}
        for it in range(max_iter):          # Dropout2d masks in the reference's call order (iteration-major)
            for lvl in range(5):
                masks = {}
                if self.drop_rate > 0:
                    oc = OUT_CH[lvl]
                    masks = dict(cross=self._drop_mask(n, oc, E.bn_train, (it, lvl, "cross")),
                                 d1=self._drop_mask(n, oc, training, (it, lvl, "d1")),
                                 d2=self._drop_mask(n, oc, training, (it, lvl, "d2")))
                    masks = {k: v for k, v in masks.items() if v is not None}
                masks_of[it, lvl] = masks
        state = [dict(x=None, pred=None, preds=[]) for _ in range(max_iter)]
        for lvl in range(5):                # level-major issue order: the cross chains of all iterations interleave
            for it in range(max_iter):
                idx, alpha, s_t, targets, sums_all = pre[it]
                if nstreams == 3:
                    sc, sm = 0, 1 + it % 2
                elif nstreams == 2:
                    sc = sm = it % 2
                else:
                    sc = sm = 0
                st_ = state[it]
                masks = masks_of[it, lvl]
                # a BatchNorm layer must see iteration 0's update first: only iteration 0's in-launch update is kept
                # when iterations can overlap, later ones are queued for flush_bn_running()
                E.defer_bn_running = E.bn_train and it >= 1 and nstreams > 1
                with E.on(sc):
                    cat = self.level_cross(lvl, skips[lvl], masks)
                E.sync(sc, sm)
                with E.on(sm):
                    x, pred = self.level_main(lvl, cat, st_["x"], st_["pred"], mask_all[lvl], s_t, W, training, masks)
                    L.check(E.lib.isa_mask_loss_sums(pred.d(), L.ptr(targets[lvl]), None,
                                                     L.ptr(sums_all[lvl * 8 * n:(lvl + 1) * 8 * n]), E.st()),
                            "isa_mask_loss_sums")
                st_["x"], st_["pred"] = x, pred
                st_["preds"].append(pred)
                if capture is not None:
                    capture["it%d.L%d.x" % (it, lvl)] = x
                    capture["it%d.L%d.pred" % (it, lvl)] = pred
        E.defer_bn_running = False
        for sid in forked:
            E.sync(sid, 0)
        E.flush_bn_running()
        recs = [(state[it]["preds"], [pre[it][4][l * 8 * n:(l + 1) * 8 * n] for l in range(5)]) for it in range(max_iter)]
        for it in range(max_iter):
            idx, alpha, s_t, targets, sums_all = pre[it]
            preds, sums = recs[it]
            coef, adv = E.f32(5 * 4 * n), E.f32(n)
            L.check(E.lib.isa_head_loss(L.ptr(sums_all), L.ptr(alpha), L.ptr(s_t), Lp, n, self._level_w, CE_WEIGHT,
                                        LAMBDA_L, LAMBDA_R, 1.0 / max_iter, L.ptr(self.baseline), 1 if training else 0,
                                        L.ptr(coef), L.ptr(adv), L.ptr(scal), 1, E.st()), "isa_head_loss")
            if E.record:
                def bwd_losses(preds=preds, targets=targets, coef=coef, alpha=alpha, idx=idx, s_t=s_t, adv=adv):
                    for lvl in range(5):
                        acc = E.grads.claim(preds[lvl], E)
                        L.check(E.lib.isa_mask_loss_grad(preds[lvl].d(), L.ptr(targets[lvl]), None,
                                                         L.ptr(coef[lvl * 4 * n:(lvl + 1) * 4 * n]),
                                                         E.grads.grad_of(preds[lvl]).d(), acc, E.st()), "isa_mask_loss_grad")
                    L.check(E.lib.isa_ins_softmax_bwd(L.ptr(alpha), L.ptr(ins), L.ptr(idx), L.ptr(s_t), L.ptr(adv), n, nobj,
                                                      Lp, L.ptr(self.dmerge), n, E.st()), "isa_ins_softmax_bwd")
                E.tape.append(bwd_losses)
            iters.append(dict(idx=idx, alpha=alpha, s_t=s_t, targets=targets, preds=preds, sums=sums))
        return dict(iters=iters, max_iter=max_iter, merge=merge, x_enc=x_enc, scal=scal)

    def _forward_batched(self, x_enc, merge, skips, sem_map, ins, n, H, W, nobj, G, training, idx_dev, injected_s_t,
                         capture, mask_all, scal):
        """The G decoder iterations as one pass over G*n images, rows ordered [iteration][image] (attenet2.py:384-399
        runs them one after the other; they share weights and backbone features and do not read each other's outputs).
        BatchNorm statistics stay per iteration (isa_tensor.groups = G), running statistics take the G updates in
        iteration order inside isa_bn_finalize, the REINFORCE baseline EMA runs in iteration order inside isa_head_loss.
        The cross branches (backbone features only) run beside the level chain on side stream 1 (ISA_STREAMS >= 2)."""
        E = self.E
        Lp, R = H * W, G * n
        idx_flat = idx_dev.reshape(-1).contiguous()          # [G*n]
        alpha, rowstat = E.f32(R * Lp), E.f32(2 * R)
        L.check(E.lib.isa_ins_softmax(L.ptr(merge), L.ptr(ins), L.ptr(idx_flat), R, nobj, Lp, L.ptr(alpha), L.ptr(rowstat), n,
                                      L.ptr(E.f32(R * ROW_PART)), E.st()), "isa_ins_softmax")
        s_t = E.arena.alloc((R,), torch.int32)
        if injected_s_t is not None:
            for it in range(G):
                s_t[it * n:(it + 1) * n].copy_(injected_s_t[it])
        else:
            race = None                                      # attenet2.py:304-321 `sample`, see the per-iteration path
            if training and self.sample_in_training:
                race = torch.empty(R, Lp, dtype=torch.float32, device=alpha.device).exponential_(1.0)
            L.check(E.lib.isa_row_argmax(L.ptr(alpha), L.ptr(race), R, Lp, L.ptr(s_t), E.st()), "isa_row_argmax")
        targets = []
        for f in FACTORS:
            t = E.f32(R * (H // f) * (W // f))
            L.check(E.lib.isa_pool_target(L.ptr(ins), L.ptr(idx_flat), None, nobj, R, H, W, f, L.ptr(t), n, E.st()),
                    "isa_pool_target")
            targets.append(t)
        sums_all = E.scratch(5 * 8 * R)                      # [level][iteration][image][8]
        masks = self._masks_batched(n, G, training)
        sc = 1 if self.streams >= 2 else 0                   # the cross chain's stream
        if sc:
            for sk in skips:
                E.grads.share(sk)
            if E.record:
                E.tape.append(lambda: E.grads.merge_shared(E))
            E.sync(0, sc)
        x = pred = None
        preds = []
        for lvl in range(5):
            with E.on(sc):
                cat = self.level_cross(lvl, skips[lvl], masks[lvl], G)
            E.sync(sc, 0)
            x, pred = self.level_main(lvl, cat, x, pred, mask_all[lvl], s_t, W, training, masks[lvl])
            L.check(E.lib.isa_mask_loss_sums(pred.d(), L.ptr(targets[lvl]), None, L.ptr(sums_all[lvl * 8 * R:(lvl + 1) * 8 * R]),
                                             E.st()), "isa_mask_loss_sums")
            preds.append(pred)
            if capture is not None:
                for it in range(G):
                    capture["it%d.L%d.x" % (it, lvl)] = x.images(it * n, n)
                    capture["it%d.L%d.pred" % (it, lvl)] = pred.images(it * n, n)
        coef, adv = E.f32(5 * 4 * R), E.f32(R)
        L.check(E.lib.isa_head_loss(L.ptr(sums_all), L.ptr(alpha), L.ptr(s_t), Lp, n, self._level_w, CE_WEIGHT, LAMBDA_L,
                                    LAMBDA_R, 1.0 / G, L.ptr(self.baseline), 1 if training else 0, L.ptr(coef), L.ptr(adv),
                                    L.ptr(scal), G, E.st()), "isa_head_loss")
        if E.record:
            def bwd_losses():
                for lvl in range(5):
                    acc = E.grads.claim(preds[lvl], E)
                    L.check(E.lib.isa_mask_loss_grad(preds[lvl].d(), L.ptr(targets[lvl]), None,
                                                     L.ptr(coef[lvl * 4 * R:(lvl + 1) * 4 * R]), E.grads.grad_of(preds[lvl]).d(),
                                                     acc, E.st()), "isa_mask_loss_grad")
                L.check(E.lib.isa_ins_softmax_bwd(L.ptr(alpha), L.ptr(ins), L.ptr(idx_flat), L.ptr(s_t), L.ptr(adv), R, nobj, Lp,
                                                  L.ptr(self.dmerge), n, E.st()), "isa_ins_softmax_bwd")
            E.tape.append(bwd_losses)
        iters = []
        for it in range(G):                                  # per-iteration views, as the one-pass-per-iteration path returns them
            hw = [(H // f) * (W // f) for f in FACTORS]
            iters.append(dict(idx=idx_flat[it * n:(it + 1) * n], alpha=alpha[it * n * Lp:(it + 1) * n * Lp],
                              s_t=s_t[it * n:(it + 1) * n],
                              targets=[targets[l][it * n * hw[l]:(it + 1) * n * hw[l]] for l in range(5)],
                              preds=[p.images(it * n, n) for p in preds],
                              sums=[sums_all[l * 8 * R + it * 8 * n:l * 8 * R + (it + 1) * 8 * n] for l in range(5)]))
        return dict(iters=iters, max_iter=G, merge=merge, x_enc=x_enc, scal=scal)

    @staticmethod
    def order_tensor(selected_idx, max_iter, n):
        return torch.tensor([[selected_idx[b][it] for b in range(n)] for it in range(max_iter)],
                            dtype=torch.int32).reshape(max_iter, n)

    # ------------------------------------------------------------------ host-side scalar assembly
    @staticmethod
    def scalars_from_sums(rec, training, baseline=0.0):
        """Reference outputs (ins_cost, criterion, ins_ce_loss, ins_dice_loss) from the per-level
        per-image sums.  Host arithmetic on 5*B*7 floats (validation / logging path; the training
        path assembles the same quantities on device in isa_head_loss)."""
        tot = dict(loss=0.0, criterion=0.0, ce=0.0, dice=0.0)
        nan = False
        max_iter = rec["max_iter"]
        for itrec in rec["iters"]:
            S = [s.view(-1, 8).double().cpu() for s in itrec["sums"]]
            last = S[-1]
            nimg = last.shape[0]
            cnt = last[:, 6]
            eval_ce = float(last[:, 4].sum() / cnt.sum())
            dice1 = 1.0 - (2 * last[:, 0] + 1.0) / (last[:, 1] + last[:, 2] + 1.0)
            if not training:
                dice2 = 1.0 - (2 * last[:, 0] + 1.0) / (last[:, 5] + last[:, 2] + 1.0)
                tot["loss"] = tot["loss"] + dice2
                tot["criterion"] = tot["criterion"] + (eval_ce + dice1)
            else:
                nan = True
                tot["criterion"] = tot["criterion"] + (eval_ce + float(dice1.sum()))
            tot["ce"] += eval_ce
            tot["dice"] += float(dice1.mean())
        out = dict(ins_ce_loss=tot["ce"] / max_iter, ins_dice_loss=tot["dice"] / max_iter)
        if training:
            out["ins_cost"] = float("nan")
            out["criterion"] = float(tot["criterion"]) / max_iter
        else:
            out["ins_cost"] = float((tot["loss"] / max_iter).mean())
            out["criterion"] = float((tot["criterion"] / max_iter).mean())
        return out
