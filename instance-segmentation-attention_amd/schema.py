"""state_dict schema of the reference network (names, shapes, registration order).

Mirrors `reseg.ReSeg(2, use_instance_seg).state_dict()` of the reference (891
tensors / 4 824 330 elements with the instance stems).  Derived from the constructors:
reseg.py:52-102, unet_model.py:8-21, unet_parts.py:7-93, MobileNetDenseASPP.py:68-123,
attenet2.py:17-50,410-430, utils.py:402-420,457-483,613-630,696-710,777-786,816-822,946-1025.
"""


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


def is_bn_prefix(schema_names, prefix):
    return (prefix + ".running_mean") in schema_names
