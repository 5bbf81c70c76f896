"""ctypes binding of libisa_kernels.so (include/isa_kernels.h).

There is deliberately NO fallback: if the HIP library is missing or a symbol is absent, importing
the product path raises.  The CPU oracle under oracle/ is never consulted from here.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ISA_KERNELS_LIB: another build of the same library (A/B kernel timings inside one GPU session; scripts/kbench.py)
LIB_PATH = os.environ.get("ISA_KERNELS_LIB") or os.path.join(_HERE, "libisa_kernels.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_LEAKY, ACT_TANH = 0, 1, 2, 3, 4
IN_1X1, IN_3X3, IN_GATHER2 = 0, 1, 2
OUT_PLAIN, OUT_SHUFFLE2 = 0, 1

_ERR = {-1: "ISA_EINVAL", -2: "ISA_EALIGN", -3: "ISA_EDTYPE", -4: "ISA_ELAUNCH", -5: "ISA_ENOMEM"}


class IsaTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("c", C.c_int32), ("ld", C.c_int32), ("dtype", C.c_int32), ("groups", C.c_int32)]


class IsaBnFin(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p),
                ("invstd", C.c_void_p), ("count", C.c_float), ("momentum", C.c_float), ("eps", C.c_float),
                ("repeat", C.c_int32)]


class IsaPro(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("bscale", C.c_void_p),
                ("act", C.c_int32), ("fin", C.POINTER(IsaBnFin))]


class IsaConvEp(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("act", C.c_int32), ("res", C.c_void_p)]


class IsaBnBwd(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("red", C.c_void_p), ("out_red", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("count", C.c_float), ("act", C.c_int32)]


class IsaBnUpd(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("count", C.c_float), ("c", C.c_int32)]


class IsaPackEntry(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("dst_off", C.c_int64), ("kind", C.c_int32),
                ("n", C.c_int32), ("k", C.c_int32), ("taps", C.c_int32), ("kp", C.c_int32),
                ("kmap_off", C.c_int32), ("rows", C.c_int32)]


P_T, P_PRO, VP, I32, F = C.POINTER(IsaTensor), C.POINTER(IsaPro), C.c_void_p, C.c_int32, C.c_float
I64 = C.c_int64
P_BN = C.POINTER(IsaBnBwd)

# name -> argtypes, exactly mirroring include/isa_kernels.h
SIGNATURES = {
    "isa_pack_weights": [VP, I32, VP, VP, VP, I32, VP],
    "isa_conv_gemm": [P_T, P_PRO, VP, I32, VP, P_T, I32, I32, VP, I32, VP],
    "isa_conv_gemm_ep": [P_T, P_PRO, VP, I32, VP, P_T, I32, VP, VP],
    "isa_dwpw_eval": [P_T, VP, VP, VP, VP, I32, VP, P_T, VP],
    "isa_conv_wgrad": [P_T, P_PRO, P_T, VP, VP, I32, I32, VP, I32, VP, I64, VP, VP],
    "isa_colsum": [P_T, VP, VP],
    "isa_dwconv3x3": [P_T, P_PRO, VP, VP, P_T, VP, VP],
    "isa_dwconv3x3_dgrad": [P_T, VP, P_T, I32, VP],
    "isa_dwconv3x3_wgrad": [P_T, P_PRO, P_T, VP, VP, I32, VP, I64, VP, VP],
    "isa_dwconv3x3_bn_backward": [P_T, P_T, P_BN, P_T, P_PRO, P_BN, VP, VP, I32, P_T, I32, P_T, VP, I64, VP, VP],
    "isa_conv1x1_bn_backward": [P_T, P_T, P_BN, P_T, P_PRO, P_BN, VP, VP, P_T, I32, P_T, VP, I64, VP, VP],
    "isa_slab_arena_create": [VP, I64, C.POINTER(C.c_void_p)],
    "isa_slab_arena_destroy": [VP],
    "isa_slab_arena_begin": [VP],
    "isa_slab_arena_flush": [VP, VP, VP, VP],
    "isa_bn_running_update": [C.POINTER(IsaBnUpd), I32, F, VP],
    "isa_d4_augment": [VP, VP, I32, I32, I32, I32, I32, VP, VP],
    "isa_rotate_nearest_u8": [VP, I32, I32, I32, I32, VP, I32, I32, VP, VP],
    "isa_rotate_bilinear_u8": [VP, I32, I32, I32, I32, VP, I32, I32, VP, VP, VP],
    "isa_cover_rows_u8": [VP, I32, I32, I32, I32, VP, VP, VP],
    "isa_plane_sums_u8": [VP, I32, I32, I32, I32, I32, I32, I32, I32, VP, VP],
    "isa_crop_planes_u8": [VP, I32, I32, I32, I32, I32, I32, VP, I32, I32, I32, VP, VP],
    "isa_resize_nearest_u8": [VP, I32, I32, I32, I32, VP, I32, I32, VP],
    "isa_resize_bilinear_u8": [VP, I32, I32, I32, I32, VP, I32, I32, VP, I64, VP],
    "isa_collate_targets": [VP, VP, I32, I32, I32, I32, VP, VP, VP],
    "isa_bn_finalize": [VP, F, VP, VP, VP, VP, F, F, VP, VP, VP, VP, I32, I32, I32, VP],
    "isa_bn_bwd_reduce": [P_T, P_T, VP, VP, VP, VP, I32, VP, VP, VP],
    "isa_bn_bwd_apply": [P_T, P_T, VP, VP, VP, VP, I32, VP, VP, VP, F, I32, P_T, VP, VP, VP],
    "isa_affine_act_res": [P_T, P_PRO, P_T, P_T, VP, P_T, VP],
    "isa_axpy": [P_T, P_T, F, I32, VP],
    "isa_avgpool2": [P_T, P_T, VP],
    "isa_avgpool2_bwd": [P_T, P_T, I32, VP],
    "isa_pool_f": [P_T, P_T, I32, I32, VP],
    "isa_avgpool3": [P_T, P_T, P_T, I32, VP],
    "isa_chan_mean": [P_T, P_PRO, VP, VP],
    "isa_se_fc": [VP, VP, VP, VP, VP, I32, I32, I32, VP, VP, VP],
    "isa_chan_argmax": [P_T, P_T, VP],
    "isa_mask_dot": [P_T, VP, VP, VP, VP, VP, VP],
    "isa_sp_softmax": [VP, VP, VP, VP, VP, VP, I32, I32, I64, VP, VP, VP, VP],
    "isa_scaled_stats": [P_T, VP, VP, VP],
    "isa_sp_apply": [P_T, VP, VP, VP, VP, P_T, VP],
    "isa_maskbn_stats": [P_T, VP, VP, VP, VP],
    "isa_maskbn_finalize": [VP, VP, I32, I32, VP, VP, VP, F, I32, VP],
    "isa_maskbn_apply_pool": [P_T, VP, VP, VP, VP, F, VP, VP],
    "isa_ins_softmax": [VP, VP, VP, I32, I32, I64, VP, VP, I32, VP, VP],
    "isa_row_argmax": [VP, VP, I32, I64, VP, VP],
    "isa_onehot_map": [VP, I32, I64, VP, VP],
    "isa_dropout_mask": [VP, I64, F, VP, VP],
    "isa_softmax_nchw": [P_T, VP, VP],
    "isa_pool_target": [VP, VP, VP, I32, I32, I32, I32, I32, VP, I32, VP],
    "isa_concat_aux": [P_T, VP, VP, I32, I32, I32, I32, VP],
    "isa_gate": [P_T, P_T, P_T, VP, VP],
    "isa_mask_loss_sums": [P_T, VP, VP, VP, VP],
    "isa_head_loss": [VP, VP, VP, I64, I32, VP, F, F, F, F, VP, I32, VP, VP, VP, I32, VP],
    "isa_sem_loss": [VP, I32, VP, VP, VP],
    "isa_mask_loss_grad": [P_T, VP, VP, VP, P_T, I32, VP],
    "isa_ins_softmax_bwd": [VP, VP, VP, VP, VP, I32, I32, I64, VP, I32, VP],
    "isa_maskbn_bwd": [P_T, VP, VP, VP, VP, F, VP, I32, VP, VP, VP, VP, P_T, I32, VP],
    "isa_sp_bwd": [P_T, P_T, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, F, I32, VP, P_T, I32,
                   VP, VP, VP, VP, VP, VP, VP, VP],
    "isa_gate_bwd": [P_T, P_T, VP, P_T, I32, VP, P_T, I32, VP],
    "isa_se_bwd": [P_T, P_T, VP, VP, VP, VP, VP, I32, VP, VP, VP, VP, VP, VP, P_T, I32, VP],
    "isa_scale_bc": [P_T, VP, P_T, I32, VP],
    "isa_sqnorm": [VP, I64, F, VP, VP],
    "isa_adadelta": [VP, VP, VP, VP, I64, F, F, F, F, VP, F, F, VP, VP],
    "isa_sdp_attention": [VP, VP, VP, VP, VP, VP, I32, I32, I64, I32, I32, F, I32, I32, I32, VP, I64, I32, VP],
    "isa_sdp_attention_bwd": [VP, VP, VP, VP, VP, VP, VP, VP, VP, I32, I32, I64, I32, I32, F, I32, I32, VP],
    "isa_sdp_scores": [VP, VP, VP, I32, I32, I32, I64, I32, I32, I32, VP],
    "isa_linear_ln": [VP, VP, VP, VP, VP, VP, F, I32, I32, I32, VP, VP],
    "isa_instance_norm_res": [P_T, P_T, P_T, F, VP, VP],
    "isa_local_attention": [P_T, P_T, P_T, VP, P_T, I32, VP],
    "isa_point_query": [VP, P_T, VP, VP],
    "isa_image_ex": [VP, P_T, VP],
    "isa_nchw_to_nhwc": [VP, I32, P_T, VP],
    "isa_nhwc_to_nchw": [P_T, VP, VP],
}


class IsaError(RuntimeError):
    pass


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "HIP kernel library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C instance-segmentation-attention_amd/csrc`. There is no CPU fallback."
            % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: intended
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.isa_resize_bilinear_ws_bytes.argtypes = [I32] * 6       # the one entry point that returns a size, not a status
    lib.isa_resize_bilinear_ws_bytes.restype = C.c_int64
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def check(rc, what):
    if rc != 0:
        raise IsaError("%s failed: %s" % (what, _ERR.get(rc, rc)))


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16            # attention operators only
    raise IsaError("unsupported activation dtype %s" % dt)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def addr(t):
    """Device address of a torch tensor as a plain int (None -> NULL) for struct fields."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
