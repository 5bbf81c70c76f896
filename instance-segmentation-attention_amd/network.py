"""The ReSeg hot path composed from engine ops (reference: code/lib/archs/reseg.py:106-130).

Buffer plan (NHWC, one buffer per pyramid level, zero-copy concats — see DESIGN.md §layout):
  L256 buf[.,64]  = [ x1 (inc out, 32) | up4 convT out (32) ]      -> up4 reads the whole row
  L128 buf[.,128] = [ conv(32) | mean2x2(32) | up3 convT (64) ]    x2 = first 64
  L64  buf[.,256] = [ conv(64) | mean(64)   | up2 convT (128) ]    x3 = first 128
  L32  buf[.,512] = [ conv(128)| mean(128)  | up1 convT (256) ]    x4 = first 256
  L16  buf[.,512] = [ conv(256)| mean(256) ]                       x5
so torch.cat in unet_parts.py:60,91 costs nothing: producers write into channel slices.
"""
import torch

from . import lib as L
from .engine import Act, Engine, Pro, rup


class Network:
    def __init__(self, eng: Engine, use_instance_seg=True):
        self.E = eng
        self.use_instance_seg = use_instance_seg

    # ------------------------------------------------------------------ blocks
    def block_v1(self, x: Act, pre: str, out: Act):
        """InvertedV1Residual (MobileNetDenseASPP.py:68-93): dw3x3-BN-ReLU6-pw-BN (+x)."""
        E = self.E
        cin = x.c
        cout = E.params.shapes[pre + ".conv.3.weight"][0]
        if E.block_v1_eval_fusable(x, cout):               # eval: the whole block in one launch, the dw output stays in LDS
            return E.block_v1_eval(x, pre, out, res=x if cin == cout else None)
        y1 = E.like(x, cin)
        if E.eval_fusable():                               # eval: BN.4 and the residual ride in the 1x1 conv's epilogue
            E.dwconv(x, pre + ".conv.0.weight", y1, stats=False)
            y1 = E.bn(y1, None, pre + ".conv.1", L.ACT_RELU6)
            return E.conv_bn_eval(y1, pre + ".conv.3.weight", out, pre + ".conv.4", L.ACT_NONE, res=x if cin == cout else None)
        _, s1 = E.dwconv(x, pre + ".conv.0.weight", y1, stats=True)
        y1 = E.bn(y1, s1, pre + ".conv.1", L.ACT_RELU6)
        y2 = E.like(x, cout)
        _, s2 = E.conv(y1, pre + ".conv.3.weight", y2, stats=True)
        return E.bn_out(y2, s2, pre + ".conv.4", L.ACT_NONE, out, res=x if cin == cout else None)

    def block_ir(self, x: Act, pre: str, out: Act, oscale=None):
        """InvertedResidual (MobileNetDenseASPP.py:96-123): pw-BN-ReLU6-dw-BN-ReLU6-pw-BN (+x).
        `oscale` folds a following Dropout2d (which scales the block output, residual included)
        into the materialising pass."""
        E = self.E
        cin = x.c
        chid = E.params.shapes[pre + ".conv.0.weight"][0]
        cout = E.params.shapes[pre + ".conv.6.weight"][0]
        y1 = E.like(x, chid)
        if E.eval_fusable() and oscale is None:
            # eval: expand conv stores ReLU6(BN(.)) (no prologue in the depthwise conv), the project conv applies BN.7 and
            # the residual in its epilogue (no materialising pass)
            E.conv_bn_eval(x, pre + ".conv.0.weight", y1, pre + ".conv.1", L.ACT_RELU6)
            y2 = E.like(x, chid)
            E.dwconv(y1, pre + ".conv.3.weight", y2, stats=False)
            y2 = E.bn(y2, None, pre + ".conv.4", L.ACT_RELU6)
            return E.conv_bn_eval(y2, pre + ".conv.6.weight", out, pre + ".conv.7", L.ACT_NONE, res=x if cin == cout else None)
        _, s1 = E.conv(x, pre + ".conv.0.weight", y1, stats=True)
        y1 = E.bn(y1, s1, pre + ".conv.1", L.ACT_RELU6)
        y2 = E.like(x, chid)
        _, s2 = E.dwconv(y1, pre + ".conv.3.weight", y2, stats=True)
        y2 = E.bn(y2, s2, pre + ".conv.4", L.ACT_RELU6)
        y3 = E.like(x, cout)
        _, s3 = E.conv(y2, pre + ".conv.6.weight", y3, stats=True)
        return E.bn_out(y3, s3, pre + ".conv.7", L.ACT_NONE, out, res=x if cin == cout else None,
                        oscale=oscale)

    def double_v1(self, x: Act, pre: str, out: Act):
        E = self.E
        cmid = E.params.shapes[pre + ".conv.down_conv_0.conv.3.weight"][0]
        mid = E.like(x, cmid)
        self.block_v1(x, pre + ".conv.down_conv_0", mid)
        return self.block_v1(mid, pre + ".conv.down_conv_1", out)

    # ------------------------------------------------------------------ backbone
    def unet(self, x_in: Act):
        """UNet.forward (unet_model.py:23-36).  x_in: [n,h,w,21] view (ld 24)."""
        E = self.E
        n, h, w = x_in.n, x_in.h, x_in.w
        widths = [(64, 32), (128, 64), (256, 128), (512, 256), (512, 512)]   # (buffer ld, x_k width)
        bufs = []
        for lvl, (ld, _) in enumerate(widths):
            bufs.append(E.new_act(n, h >> lvl, w >> lvl, ld, ld=ld))
        x1 = bufs[0].slice(0, 32)
        self.double_v1(x_in, "base.inc.conv", x1)
        feats = [x1]
        cur = x1
        for lvl in range(1, 5):
            cw = widths[lvl][1] // 2                       # conv half / pooled half
            pooled = bufs[lvl].slice(cw, cw)
            E.avgpool2(cur, pooled)
            self.double_v1(pooled, "base.down%d.mpconv" % lvl, bufs[lvl].slice(0, cw))
            cur = bufs[lvl].slice(0, 2 * cw)
            feats.append(cur)
        y = feats[4]
        for i, lvl in enumerate((3, 2, 1, 0)):
            pre = "base.up%d" % (i + 1)
            skip_w = widths[lvl][1]
            co = E.params.shapes[pre + ".up.weight"][1]
            up_dst = bufs[lvl].slice(skip_w, co)
            E.conv(y, pre + ".up.weight", up_dst, bias=pre + ".up.bias", transposed=True)
            cat = bufs[lvl].slice(0, skip_w + co)
            out = E.new_act(n, h >> lvl, w >> lvl, co)
            y = self.double_v1(cat, pre + ".conv", out)
        return y, feats

    # ------------------------------------------------------------------ heads
    def sem_head(self, x_dec: Act):
        """channelAttend (utils.py:402-420) + sem_seg_output (reseg.py:73-75,115-116)."""
        E = self.E
        P = E.params
        n, c = x_dec.n, x_dec.c
        mean = E.scratch(n * c)
        L.check(E.lib.isa_chan_mean(x_dec.d(), None, L.ptr(mean), E.st()), "isa_chan_mean")
        hid, gate = E.f32(n * 16), E.f32(n * c)
        L.check(E.lib.isa_se_fc(L.ptr(mean), P.ptr("channelAttend.fc.0.weight"), P.ptr("channelAttend.fc.0.bias"),
                                P.ptr("channelAttend.fc.2.weight"), P.ptr("channelAttend.fc.2.bias"), n, c, 16,
                                L.ptr(hid), L.ptr(gate), E.st()), "isa_se_fc")
        gated = x_dec.with_pro(Pro(bscale=gate))
        sem = E.new_act(n, x_dec.h, x_dec.w, 2)
        E.conv(gated, "sem_seg_output.weight", sem, bias="sem_seg_output.bias", record_bwd=False)
        reg = E._last_conv["reg"]
        if E.record:
            def bwd():
                # sem conv: weight/bias grads see x*gate; its data gradient is w.r.t. x*gate (dxa)
                dy = E.grads.grad_of(sem)
                L.check(E.lib.isa_conv_wgrad(gated.d(), gated.p(), dy.d(), P.gptr("sem_seg_output.weight"),
                                             P.gptr("sem_seg_output.bias"), L.IN_1X1, L.OUT_PLAIN, None, c,
                                             L.ptr(E.ws), E.ws.numel(), E.defer_handle(), E.st()),
                        "isa_conv_wgrad(sem)")
                dxa = E.new_act(n, x_dec.h, x_dec.w, c)
                L.check(E.lib.isa_conv_gemm(dy.d(), None, E.packer.ptr(reg["dgrad"]), reg["kp_d"], None, dxa.d(),
                                            L.IN_1X1, L.OUT_PLAIN, None, 0, E.st()), "isa_conv_gemm(dgrad sem)")
                dg, dmean = E.scratch(n * c), E.f32(n * c)
                acc = E.grads.claim(x_dec, E)
                L.check(E.lib.isa_se_bwd(dxa.d(), x_dec.d(), L.ptr(gate), L.ptr(hid), L.ptr(mean),
                                         P.ptr("channelAttend.fc.0.weight"), P.ptr("channelAttend.fc.2.weight"), 16,
                                         L.ptr(dg), L.ptr(dmean), P.gptr("channelAttend.fc.0.weight"),
                                         P.gptr("channelAttend.fc.0.bias"), P.gptr("channelAttend.fc.2.weight"),
                                         P.gptr("channelAttend.fc.2.bias"), E.grads.grad_of(x_dec).d(), acc, E.st()),
                        "isa_se_bwd")
            E.tape.append(bwd)
        return sem

    def sem_loss(self, sem: Act, sem_onehot: torch.Tensor):
        """Trainer-side CE + Dice(time=1) on the semantic logits (model.py:255-269).
        Returns device tensor [ce, dice]; records d(sem)."""
        E = self.E
        n = sem.n
        sums = E.scratch(8 * n)
        L.check(E.lib.isa_mask_loss_sums(sem.d(), None, L.ptr(sem_onehot), L.ptr(sums), E.st()), "isa_mask_loss_sums")
        coef, scal = E.f32(4 * n), E.f32(2)
        L.check(E.lib.isa_sem_loss(L.ptr(sums), n, L.ptr(coef), L.ptr(scal), E.st()), "isa_sem_loss")
        if E.record:
            def bwd():
                acc = E.grads.claim(sem, E)
                L.check(E.lib.isa_mask_loss_grad(sem.d(), None, L.ptr(sem_onehot), L.ptr(coef),
                                                 E.grads.grad_of(sem).d(), acc, E.st()), "isa_mask_loss_grad(sem)")
            E.tape.append(bwd)
        return scal

    def onehot_map(self, sem_onehot: torch.Tensor) -> torch.Tensor:
        """int64 one-hot [n,2,h,w] -> fp32 {0,1} map [n, h*w] (sem_seg_argmax of reseg.py:118, on the device)."""
        E = self.E
        n, c, h, w = sem_onehot.shape
        assert c == 2 and sem_onehot.dtype == torch.int64
        out = E.f32(n, h * w)
        L.check(E.lib.isa_onehot_map(L.ptr(sem_onehot), n, h * w, L.ptr(out), E.st()), "isa_onehot_map")
        return out

    def softmax_nchw(self, logits: Act) -> torch.Tensor:
        out = torch.empty((logits.n, logits.c, logits.h, logits.w), dtype=torch.float32, device=logits.buf.device)
        L.check(self.E.lib.isa_softmax_nchw(logits.d(), L.ptr(out), self.E.st()), "isa_softmax_nchw")
        return out

    def argmax_map(self, logits: Act):
        E = self.E
        out = E.new_act(logits.n, logits.h, logits.w, 1)
        L.check(E.lib.isa_chan_argmax(logits.d(), out.d(), E.st()), "isa_chan_argmax")
        return out

    # ------------------------------------------------------------------ boundary
    def to_nhwc(self, x: torch.Tensor, c_pad=None) -> Act:
        """NCHW fp32 (reference layout) -> NHWC activation view."""
        E = self.E
        n, c, h, w = x.shape
        x = x.contiguous()
        dst = E.new_act(n, h, w, c, ld=rup(c, 8))
        dst.needs_grad = False
        self._keep = x
        L.check(E.lib.isa_nchw_to_nhwc(L.ptr(x), c, dst.d(), E.st()), "isa_nchw_to_nhwc")
        return dst

    def image_ex(self, rgb: torch.Tensor) -> Act:
        """uint8 RGB [n,h,w,3] -> the 21-channel standardized NHWC input (ImageEx + ToTensor + Standardization,
        lib/utils.py:90-113, preprocess.py:192-195) in one kernel; replaces the six skimage conversions per image
        of the reference's data pipeline (parity unpinned, see oracle/image_ex_ref.py)."""
        E = self.E
        assert rgb.dtype == torch.uint8 and rgb.dim() == 4 and rgb.shape[-1] == 3, "expects uint8 [n,h,w,3]"
        n, h, w, _ = rgb.shape
        rgb = rgb.to(E.device).contiguous()
        dst = E.new_act(n, h, w, 21, ld=24)
        dst.needs_grad = False
        self._keep = rgb
        L.check(E.lib.isa_image_ex(L.ptr(rgb), dst.d(), E.st()), "isa_image_ex")
        return dst

    def input_view(self, x: torch.Tensor) -> Act:
        """Network input: the reference's [n,21,h,w] float tensor, or raw uint8 RGB [n,h,w,3] expanded on device."""
        if x.dtype == torch.uint8:
            return self.image_ex(x)
        return self.to_nhwc(x.to(device=self.E.device, dtype=torch.float32))

    def collate_targets(self, sem: torch.Tensor, ins: torch.Tensor):
        """Targets as the reference's collate function leaves them before its last five lines (dataset.py:349-379):
        ins uint8 [n,h,w,K] instance planes, sem uint8 [n,h,w] -> (sem one-hot int64 [n,2,h,w], ins int64 [n,K,h,w])
        on the device (isa_collate_targets).  8.6x less PCIe traffic than shipping the int64 tensors."""
        E = self.E
        assert ins.dtype == torch.uint8 and ins.dim() == 4 and sem.dtype == torch.uint8 and sem.dim() == 3
        n, h, w, k = ins.shape
        assert tuple(sem.shape) == (n, h, w)
        ins, sem = ins.to(E.device).contiguous(), sem.to(E.device).contiguous()
        ins_out = E.arena.alloc((n, k, h, w), torch.int64)
        sem_out = E.arena.alloc((n, 2, h, w), torch.int64)
        L.check(E.lib.isa_collate_targets(L.ptr(ins), L.ptr(sem), n, h, w, k, L.ptr(ins_out), L.ptr(sem_out), E.st()),
                "isa_collate_targets")
        return sem_out, ins_out

    def to_nchw(self, a: Act) -> torch.Tensor:
        E = self.E
        out = torch.empty((a.n, a.c, a.h, a.w), dtype=torch.float32, device=a.buf.device)
        L.check(E.lib.isa_nhwc_to_nchw(a.d(), L.ptr(out), E.st()), "isa_nhwc_to_nchw")
        return out
