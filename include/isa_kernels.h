/*
 * isa_kernels.h — C-ABI of the MI355X (gfx950) kernel library for the ReSeg hot path.
 *
 * The reference (Snoworday/instance-segmentation-attention) has no FFI: its boundary is the Python
 * class `ReSeg` (code/lib/archs/reseg.py:52-130) whose layers dispatch to torch/cuDNN.  This header
 * is the layer the drop-in `ReSeg` class calls instead; each entry point names
 * the reference operator(s) it replaces.  Conventions:
 *   - plain C, no torch types; the caller owns every buffer (activations, parameters, scratch);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates,
 *     never synchronises, so a whole step can be captured into a hipGraph;
 *   - return 0 on success, a negative ISA_E* code on argument errors (nothing is launched then);
 *   - activations are NHWC "views": element (b,y,x,c) lives at data[((b*h+y)*w+x)*ld + c], so
 *     channel concatenation is done by writing into a slice of a wider buffer (reseg/unet `cat`);
 *   - an input may be LAZY: its true value is act(scale[c]*raw+shift[c])*bscale[b][c] (isa_pro).
 *     That is how train-mode BatchNorm+ReLU6 (137 BN modules) costs no extra pass over HBM: the
 *     producer writes raw conv output and per-channel sum/sumsq, isa_bn_finalize turns those into
 *     scale/shift, the consumer applies them while loading.
 */
#ifndef ISA_KERNELS_H
#define ISA_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ISA_F32 = 0, ISA_BF16 = 1,
       ISA_F16 = 2 };  /* fp16 storage: accepted by the attention operators (isa_sdp_attention, isa_sdp_scores) only */
enum { ISA_ACT_NONE = 0, ISA_ACT_RELU = 1, ISA_ACT_RELU6 = 2, ISA_ACT_LEAKY = 3, ISA_ACT_TANH = 4 };
enum { ISA_STAT_REPLICAS = 8 };   /* layout of every per-channel statistics buffer: [8][2*C] */
enum { ISA_OK = 0, ISA_EINVAL = -1, ISA_EALIGN = -2, ISA_EDTYPE = -3, ISA_ELAUNCH = -4, ISA_ENOMEM = -5 };

/* conv_gemm input addressing */
enum { ISA_IN_1X1 = 0,       /* K = Cin                       (nn.Conv2d k=1)                    */
       ISA_IN_3X3 = 1,       /* K = 9*Cin, zero pad 1         (nn.Conv2d k=3 p=1, dense)         */
       ISA_IN_GATHER2 = 2 }; /* K = 4*Cin, tap (dy,dx) reads pixel (2y+dy,2x+dx) of a 2x image
                                (data-gradient of ConvTranspose2d k=2 s=2)                       */
/* conv_gemm output addressing */
enum { ISA_OUT_PLAIN = 0,
       ISA_OUT_SHUFFLE2 = 1 }; /* N = 4*Cout, column (dy,dx,co) goes to pixel (2y+dy,2x+dx)
                                  (forward of ConvTranspose2d k=2 s=2)                          */

typedef struct isa_tensor {
    void*   data;   /* device pointer to element (0,0,0,0) of the view */
    int32_t n, h, w;
    int32_t c;      /* channels in the view */
    int32_t ld;     /* pixel stride in elements (>= c) */
    int32_t dtype;  /* ISA_F32 | ISA_BF16 */
    int32_t groups; /* statistic groups G (0 or 1: one group).  G > 1: the n images are G consecutive groups of n/G
                       images with separate train-mode BatchNorm statistics (the decoder iterations of attenet2.py:384-399
                       batched into one pass share weights but not batch statistics).  Every per-channel array that
                       accompanies a grouped tensor then carries a leading [G] dimension: isa_pro.scale/shift [G][c],
                       stats [G][ISA_STAT_REPLICAS][2N], isa_bn_bwd scale/shift/mean/invstd [G][c] and red/out_red
                       [G][ISA_STAT_REPLICAS][2c]; per-image arrays (bscale, oscale) are [n][c] as before and parameter
                       gradients (dw, dbias, dgamma, dbeta) are sums over all groups.  Entry points without per-channel
                       statistics or constants ignore the field. */
} isa_tensor;

/* A train-mode BatchNorm finalize that has NOT run yet, handed to the consumer of the lazy tensor (isa_pro.fin): the
 * consuming kernel derives scale/shift from the producer's statistics itself - every workgroup for its own statistic
 * group, into LDS - and ONE workgroup of the launch does what isa_bn_finalize does to memory: scale/shift/mean/invstd
 * [G][c] (read later by the backward entry points) and, when running_mean/var are given, the G ordered running-statistic
 * updates.  The arithmetic is isa_bn_finalize's (the same device function).  It removes a ~5 us launch from the dependency
 * chain of every BatchNorm (nn.BatchNorm2d forward in train mode, torch/nn/modules/batchnorm.py; the reference's
 * MobileNetDenseASPP.py:68-123 blocks have three each).  Entry points that take an isa_pro but have no in-kernel form
 * (isa_chan_mean, the non-tiled depthwise fallback, the weight-gradient and fused backward entry points, the eval-epilogue
 * GEMM) run isa_bn_finalize on the same stream first, so
 * a non-NULL fin is always honoured.  c <= ISA_FIN_MAX_C in-kernel, wider layers take the launch. */
#define ISA_FIN_MAX_C 1024
typedef struct isa_bn_fin {
    const float* stats;        /* [G][ISA_STAT_REPLICAS][2c] sums of the batch, complete on this stream */
    const float* gamma;        /* [c] or NULL (1) */
    const float* beta;         /* [c] or NULL (0) */
    float* running_mean;       /* [c] or NULL: no running-statistics update by this consumer */
    float* running_var;
    float* scale;              /* [G][c] outputs, as isa_bn_finalize; scale/shift must equal isa_pro.scale/shift */
    float* shift;
    float* mean;               /* may be NULL */
    float* invstd;             /* may be NULL */
    float count, momentum, eps;
    int32_t repeat;            /* as isa_bn_finalize */
} isa_bn_fin;

typedef struct isa_pro {       /* lazy-input prologue; all pointers may be NULL */
    const float* scale;        /* [c]   ([G][c] for a grouped tensor, see isa_tensor.groups) */
    const float* shift;        /* [c]   */
    const float* bscale;       /* [n,c] per-image channel multiplier (Dropout2d mask, SE gate) */
    int32_t      act;          /* ISA_ACT_* applied after the affine, before bscale */
    const isa_bn_fin* fin;     /* NULL, or the pending finalize that produces scale/shift (HOST pointer, read during the call) */
} isa_pro;

typedef struct isa_conv_ep {   /* output epilogue of isa_conv_gemm_ep: y = act(scale[n]*(conv+bias) + shift[n]) + res */
    const float* scale;        /* [N] eval-mode BatchNorm scale  gamma / sqrt(running_var + eps)          */
    const float* shift;        /* [N]                     shift  beta - running_mean * scale               */
    int32_t      act;          /* ISA_ACT_* applied after the affine                                        */
    const isa_tensor* res;     /* optional residual branch, same shape and dtype as y (may be NULL)         */
} isa_conv_ep;

typedef struct isa_slab_arena isa_slab_arena;   /* opaque: deferred weight-gradient folds, see below */

typedef struct isa_pack_entry { /* one parameter tensor to repack (see isa_pack_weights) */
    int64_t src_off;            /* element offset into the fp32 master buffer */
    int64_t dst_off;            /* element offset into the packed buffer */
    int32_t kind;               /* 0 conv fwd   src[N][K][taps]  -> dst[rows=N][taps][kp]
                                   1 conv dgrad src[N][K][taps]  -> dst[rows=Kphys][taps flipped][kp>=N]
                                   2 convT fwd  src[K][Co][2][2] -> dst[rows=4*Co][kp]      (n = Co)
                                   3 convT dgrad                 -> dst[rows=Kphys][4][kp>=Co] (n = Co)
                                   4 depthwise  src[C][1][3][3]  -> dst[9][rows]            (n = C)
                                   5 depthwise, taps flipped (data gradient)                        */
    int32_t n, k, taps;         /* logical dims of the source tensor */
    int32_t kp;                 /* padded contraction length per tap in the destination (%32==0) */
    int32_t kmap_off;           /* >=0: offset into kmap[] giving, per destination channel of the
                                   K axis (kinds 0,2: contraction index; kinds 1,3: row; kinds 4,5:
                                   column), the source channel or -1 for zero: channel padding and
                                   concat reordering.  <0: identity */
    int32_t rows;               /* destination rows (see kind) */
} isa_pack_entry;

/* ---- parameter repacking -------------------------------------------------------------------
 * One launch repacks every conv weight from the reference's NCHW fp32 `state_dict` layout into
 * the MFMA-operand layouts (row = output channel, contiguous contraction, zero padded to 32). */
int isa_pack_weights(const isa_pack_entry* table_dev, int32_t n_entries, const int32_t* kmap_dev,
                     const float* src_base, void* dst_base, int32_t dst_dtype, void* stream);

/* ---- convolution as MFMA GEMM ----------------------------------------------------------------
 * y[M=n*h*w, N] (+)= pro(x)[M, K] * W^T (+ bias).  Replaces nn.Conv2d k=1 / k=3 dense and
 * nn.ConvTranspose2d(k=2,s=2) forward and their data-gradients (MobileNetDenseASPP.py:68-123 1x1
 * convs, utils.py:703-707 L0Layer, unet_parts.py:73 / utils.py:975 up-convs, reseg.py:73 head).
 * w: packed weights from isa_pack_weights ([Nrows][taps][kp], dtype = x.dtype).
 * stats: optional float[ISA_STAT_R][2*N] (zeroed by the caller); per-channel sum and sum of squares of
 *        the fp32 result are atomically added into replica (workgroup & 7): train-mode BatchNorm
 *        statistics fused into the producer without serialising on a handful of L2 atomics.
 * accumulate != 0: y += result (gradient accumulation for multi-consumer tensors).            */
int isa_conv_gemm(const isa_tensor* x, const isa_pro* pro, const void* w, int32_t kp,
                  const float* bias, const isa_tensor* y, int32_t in_mode, int32_t out_mode,
                  float* stats, int32_t accumulate, void* stream);

/* The same convolution with an output epilogue: eval-mode BatchNorm2d (a constant per-channel affine), the activation and
 * the block's residual add are applied before the store, so `model.eval()` needs neither a lazy prologue in the consumer
 * nor the materialising pass of a block output (reference: nn.Sequential(conv, BatchNorm2d, ReLU6) in eval mode and
 * `x + self.conv(x)`, MobileNetDenseASPP.py:68-123).  No statistics, no accumulate, plain output layout.           */
int isa_conv_gemm_ep(const isa_tensor* x, const isa_pro* pro, const void* w, int32_t kp,
                     const float* bias, const isa_tensor* y, int32_t in_mode, const isa_conv_ep* ep, void* stream);

/* Whole-block forward of InvertedV1Residual in eval mode (MobileNetDenseASPP.py:68-93 under model.eval()) in ONE pass:
 *   y = ep( pw1x1( ReLU6( bn1_scale * dw3x3(x) + bn1_shift ) ) )        ep as in isa_conv_gemm_ep (BN2 affine, residual)
 * The depthwise output - the block's widest tensor - stays in LDS (read C + write C' instead of 3C + C' per pixel).
 * bf16 storage; x plain (no prologue), x->c % 32 == 0 and <= 128, y->c % 16 == 0 and <= 64, h % 8 == 0, w % 32 == 0
 * (the 256x256 ... 32x32 levels of the backbone; other shapes: ISA_EINVAL, callers use isa_dwconv3x3 +
 * isa_conv_gemm_ep).  w_dw: packed [9][C] (isa_pack_weights kind 4), w_pw: packed [N][kp] (kind 0). */
int isa_dwpw_eval(const isa_tensor* x, const void* w_dw, const float* bn1_scale, const float* bn1_shift,
                  const void* w_pw, int32_t kp, const isa_conv_ep* ep, const isa_tensor* y, void* stream);

/* Weight gradient of the same family, accumulated straight into the reference's state_dict layout:
 * dw[N][Ksrc][kh][kw] (or [K][Co][2][2] for ISA_OUT_SHUFFLE2)
 *   += sum_m dy[m,n] * pro(x)[m@tap, kd],  kd -> k through kmap (NULL = identity, -1 = padding).
 * dbias[N] += sum_m dy (optional; for SHUFFLE2 the sum runs over all four output quadrants).
 * Two launches, no atomics (deterministic): split-M workgroups write partial slabs into `ws`
 * (caller scratch, >= a few MB; the split factor adapts to ws_floats), a reduce kernel folds them.
 * `defer` (optional, all weight-gradient entry points): see isa_slab_arena below. */
int isa_conv_wgrad(const isa_tensor* x, const isa_pro* pro, const isa_tensor* dy,
                   float* dw, float* dbias, int32_t in_mode, int32_t out_mode,
                   const int32_t* kmap, int32_t ksrc, float* ws, int64_t ws_floats, isa_slab_arena* defer,
                   void* stream);
/* out[c] += sum over all pixels of x[.,c]  (bias gradients) */
int isa_colsum(const isa_tensor* x, float* out, void* stream);

/* ---- depthwise 3x3, pad 1 (MobileNetDenseASPP.py:77,109; reseg.py:79,93) ---------------------
 * y = dw3x3(pro(x)) (+bias); w packed [9][C] (kind 4); stats as above.                         */
int isa_dwconv3x3(const isa_tensor* x, const isa_pro* pro, const void* w, const float* bias,
                  const isa_tensor* y, float* stats, void* stream);
/* dx (+)= dw3x3^T(dy): w = tap-flipped packing (kind 5).
 * wgrad: dw[C][1][3][3] (reference layout) += sum dy * pro(x) shifted; dbias += sum dy; only the
 * first csrc channels are real parameters (the 21-channel input is padded to 24).              */
int isa_dwconv3x3_dgrad(const isa_tensor* dy, const void* w, const isa_tensor* dx,
                        int32_t accumulate, void* stream);
int isa_dwconv3x3_wgrad(const isa_tensor* x, const isa_pro* pro, const isa_tensor* dy,
                        float* dw, float* dbias, int32_t csrc, float* ws, int64_t ws_floats, isa_slab_arena* defer,
                        void* stream);

/* Backward constants of one train-mode BatchNorm2d layer (what torch autograd keeps for
 * native_batch_norm_backward): scale/shift/mean/invstd from isa_bn_finalize, `red` = the
 * [ISA_STAT_REPLICAS][2C] sums written by isa_bn_bwd_reduce (the consumer adds the replicas),
 * `out_red` = where a fused producer writes those sums, dgamma/dbeta accumulate (may be NULL).     */
typedef struct isa_bn_bwd {
    const float* scale; const float* shift; const float* mean; const float* invstd;
    float* red; float* out_red;
    float* dgamma; float* dbeta;
    float count; int32_t act;
} isa_bn_bwd;

/* Fused backward of the  x -> dw3x3 -> y -> BN(train)+act  segment of InvertedResidual /
 * InvertedV1Residual (MobileNetDenseASPP.py:77-80,109-111), one pass over HBM:
 *   dy   = BN-backward(g, y; ybn)                 (what isa_bn_bwd_apply would store)
 *   dx  (+)= dw3x3^T(dy),  dw += pro(x) (*) dy    (isa_dwconv3x3_dgrad + isa_dwconv3x3_wgrad)
 *   xbn != NULL: xbn->out_red += sums of BN(x)'s backward over the finished dx
 *                (isa_bn_bwd_reduce of the layer that produced x; requires dx to be complete, i.e.
 *                every other consumer of x has already accumulated into it).
 * addend (optional, shape of x): dx += addend - the gradient arriving through the block's residual branch.
 * g, y, x, dx: same shape and dtype, C % 8 == 0; xpro without bscale; ybn->red already reduced.     */
int isa_dwconv3x3_bn_backward(const isa_tensor* g, const isa_tensor* y, const isa_bn_bwd* ybn,
                              const isa_tensor* x, const isa_pro* xpro, const isa_bn_bwd* xbn,
                              const void* w_flipped, float* dw, int32_t csrc,
                              const isa_tensor* dx, int32_t accumulate, const isa_tensor* addend,
                              float* ws, int64_t ws_floats, isa_slab_arena* defer, void* stream);

/* The same fusion for a bias-free 1x1 convolution y = W x (W fp32 [N][K], state_dict layout) followed by a
 * train-mode BatchNorm2d (MobileNetDenseASPP.py:105-107,113-114 expand / project; :81-82 InvertedV1):
 *   dy = BN-backward(g, y; ybn);  dw += dy^T . pro(x);  dx (+)= dy . W;  xbn as above.
 * `addend` (optional, shape of x): dx += addend, e.g. the gradient arriving through the block's residual branch
 * (saves the separate accumulate pass).
 * bf16 storage only, N <= 64, K <= 64, both multiples of 8; otherwise ISA_EINVAL (callers then use
 * isa_bn_bwd_apply + isa_conv_wgrad + isa_conv_gemm).                                               */
int isa_conv1x1_bn_backward(const isa_tensor* g, const isa_tensor* y, const isa_bn_bwd* ybn,
                            const isa_tensor* x, const isa_pro* xpro, const isa_bn_bwd* xbn,
                            const float* w, float* dw, const isa_tensor* dx, int32_t accumulate,
                            const isa_tensor* addend, float* ws, int64_t ws_floats, isa_slab_arena* defer,
                            void* stream);

/* ---- deferred weight-gradient folds ------------------------------------------------------------
 * Every weight-gradient entry point is two-stage: partial slabs, then a small fold kernel that adds them into
 * dw/dbias.  Nothing reads a weight gradient before the optimizer (torch autograd gives the same freedom to the
 * reference's loss.backward(), train.py:312-322 -> model.py:204-216), so a backward pass may postpone the folds.
 * The state is an explicit handle (no global or thread-local state): an isa_slab_arena wraps a caller-owned fp32
 * region (256-byte aligned, >= 16M floats).  Weight-gradient entry points that receive the handle (`defer` != NULL)
 * ignore their `ws`, place their slabs in the arena one after the other and record the fold; isa_slab_arena_flush
 * adds every recorded slab set into its dw/dbias with one launch per 32 folds (~190 small launches fewer per
 * training step at the BASELINE configuration).  When the arena cannot hold a call's slabs the call returns
 * ISA_ENOMEM and launches nothing.  A handle is used by one host thread at a time; distinct handles are
 * independent.  The region must not be reused before the flush's kernels have run.                        */
int isa_slab_arena_create(float* region, int64_t region_floats, isa_slab_arena** out);
int isa_slab_arena_destroy(isa_slab_arena* a);
int isa_slab_arena_begin(isa_slab_arena* a);                 /* rewind; forget folds recorded but never flushed */
int isa_slab_arena_flush(isa_slab_arena* a, void* stream, int32_t* n_folds, int64_t* floats_used);

/* ---- BatchNorm2d pieces (torch.nn.BatchNorm2d train/eval semantics) --------------------------
 * finalize: stats[2C] (sum, sumsq over `count` values) -> scale/shift (and mean/invstd for the
 * backward); updates running_mean/var (momentum, unbiased var) when running_* != NULL.
 * eval mode: pass stats == NULL and scale/shift are derived from running_*.
 * groups G > 1 (isa_tensor.groups): stats [G][ISA_STAT_REPLICAS][2c] and `count` values per group ->
 * scale/shift/mean/invstd [G][c]; the running statistics take the G updates in group order, as the reference's
 * decoder iterations pass through the same module one after the other (attenet2.py:384-399).
 * repeat R > 1: every update is applied R times (a sub-network whose input does not depend on the iteration is
 * evaluated once here where the reference evaluates it R times with identical batch statistics).           */
int isa_bn_finalize(const float* stats, float count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps,
                    float* scale, float* shift, float* mean, float* invstd, int32_t c,
                    int32_t groups, int32_t repeat, void* stream);
/* Running-statistics updates of `n` train-mode layers, applied in array order by ONE launch per 64 layers: what
 * isa_bn_finalize does to running_mean/var when it is given them.  For statistics produced on another HIP stream
 * than the one that must own the update order (the decoder iterations run concurrently, attenet2.py:384-399 runs
 * them one after the other: iteration 0 updates inside its finalize, iteration 1's updates are applied here after
 * the join).  upd: HOST array, copied into the kernel arguments (a captured hipGraph keeps it by value). */
typedef struct isa_bn_upd {
    const float* stats;           /* [ISA_STAT_REPLICAS][2C] sums of the batch (device) */
    float* running_mean; float* running_var;
    float count; int32_t c;
} isa_bn_upd;
int isa_bn_running_update(const isa_bn_upd* upd, int32_t n, float momentum, void* stream);
/* backward of t = act(scale*y+shift): reduce pass  red[2C] += (sum dz, sum dz*yhat),
 * then apply pass  dy = gamma*invstd*(dz - red0/count - yhat*red1/count)  (in place over dt ok) */
int isa_bn_bwd_reduce(const isa_tensor* dt, const isa_tensor* y, const float* scale,
                      const float* shift, const float* mean, const float* invstd, int32_t act,
                      const float* bscale, float* red, void* stream);
int isa_bn_bwd_apply(const isa_tensor* dt, const isa_tensor* y, const float* scale,
                     const float* shift, const float* mean, const float* invstd, int32_t act,
                     const float* bscale, const float* gamma, const float* red, float count,
                     int32_t train, const isa_tensor* dy, float* dgamma, float* dbeta,
                     void* stream);

/* ---- materialise a lazy tensor: out = pro(x) (+ res)  (residual adds of Inverted*Residual,
 * F.dropout2d, torch.cat placement) and its backward pieces ----------------------------------- */
/* out = (pro(x) (+res) (+res2)) * oscale[n,c]   (oscale: F.dropout2d after a residual sum).
 * Broadcast form: out has G groups (out->groups = G, out->n = G * x->n) while x, res hold ONE group: every output
 * group g is (pro(x) + res (+ res2[g])) * oscale[g] - the Dropout2d that follows an iteration-independent block. */
int isa_affine_act_res(const isa_tensor* x, const isa_pro* pro, const isa_tensor* res,
                       const isa_tensor* res2, const float* oscale, const isa_tensor* out,
                       void* stream);
/* dst (+)= src  (NHWC views; gradient fan-in of residual/skip connections) */
int isa_axpy(const isa_tensor* src, const isa_tensor* dst, float alpha, int32_t accumulate,
             void* stream);

/* ---- pooling / resampling ------------------------------------------------------------------ */
/* 2x2 mean, stride 2 == F.interpolate(scale=0.5,bilinear) on even sizes (unet_parts.py:58) */
int isa_avgpool2(const isa_tensor* x, const isa_tensor* y, void* stream);
int isa_avgpool2_bwd(const isa_tensor* dy, const isa_tensor* dx, int32_t accumulate, void* stream);
/* f x f max / mean pooling with stride f of small-channel maps (utils.py:841-846) */
int isa_pool_f(const isa_tensor* x, const isa_tensor* y, int32_t f, int32_t is_max, void* stream);
/* 3x3 mean, stride 1, pad 1, count_include_pad (utils.py:634,645); optional per-pixel mask mul;
 * y (+)= ...: the operator is self-adjoint, so the same call with accumulate is its backward */
int isa_avgpool3(const isa_tensor* x, const isa_tensor* mask, const isa_tensor* y, int32_t accumulate,
                 void* stream);

/* ---- squeeze-excite gate + heads (utils.py:402-420, reseg.py:72-75,116-120) ------------------ */
/* mean over h*w of pro(x), ADDED to out[n,c] (float atomics from several workgroups per image): zero it first */
int isa_chan_mean(const isa_tensor* x, const isa_pro* pro, float* out, void* stream);
/* gate[n,c] = sigmoid(W2 relu(W1 m + b1) + b2); hidden/intermediates kept for backward */
int isa_se_fc(const float* mean, const float* w1, const float* b1, const float* w2,
              const float* b2, int32_t n, int32_t c, int32_t hidden, float* hid, float* gate,
              void* stream);
/* argmax over channels -> float map in {0..c-1} (first max wins, torch.argmax) */
int isa_chan_argmax(const isa_tensor* x, const isa_tensor* y, void* stream);

/* ---- attention mask head (utils.py:457-663, attenet2.py:304-347) ---------------------------------
 * Single-channel maps (masks, scores, probabilities, targets) are fp32 [n, h*w] arrays.
 * SpatialAttentionLayer (utils.py:484-523), four steps: */
int isa_mask_dot(const isa_tensor* x, const float* m, const float* w, const float* bias,
                 float* dot /*zeroed*/, float* chansum /*[n,c] zeroed*/, void* stream);
/* `part` (isa_sp_softmax, isa_ins_softmax): caller scratch of n * ISA_ROW_CHUNKS * 4 floats, or NULL.  With it (and
 * L <= 4096 * ISA_ROW_CHUNKS) a row is worked on by L / 4096 workgroups in two launches - chunk scores and online-softmax
 * partials, then the normalisation - instead of by one workgroup per row (n = 16 ... 32 rows of 65 536 pixels left most of the
 * chip idle for 86 us); same values up to the order of the fp32 sums. */
#define ISA_ROW_CHUNKS 64
int isa_sp_softmax(const float* dot, const float* m, const float* chansum, const float* lh,
                   const float* fcw, const float* fcb, int32_t n, int32_t c, int64_t L,
                   float* beta, float* rowstat /*[n,4]: max,sumexp,count,h_t*/, float* part, void* stream);
int isa_scaled_stats(const isa_tensor* x, const float* beta, float* stats /*[2c] zeroed*/, void* stream);
int isa_sp_apply(const isa_tensor* x, const float* beta, const float* m, const float* scale,
                 const float* shift, const isa_tensor* out, void* stream);
/* maskBN (utils.py:568-591) on the 1-channel score map, fused with AvgPool3x3 * sem (utils.py:645) */
int isa_maskbn_stats(const isa_tensor* e, const float* m, const float* mean_in, float* out, void* stream);
int isa_maskbn_finalize(const float* am, const float* v, int32_t n, int32_t stage, float* mean_var,
                        float* running_mean, float* running_var, float f, int32_t train, void* stream);
int isa_maskbn_apply_pool(const isa_tensor* e, const float* sem, const float* mean_var, const float* w,
                          const float* b, float eps, float* merge, void* stream);
/* HardAttentionLayer softmax for the selected instance of each image (utils.py:648-655 + the
 * gather of attenet2.py:342-343): alpha[b,:] = softmax over pixels of ins[b, idx[b]] */
int isa_ins_softmax(const float* merge, const int64_t* ins, const int32_t* idx, int32_t n, int32_t nobj,
                    int64_t L, float* alpha, float* rowstat /*[n,2]*/, int32_t nsrc, float* part, void* stream);
/* `nsrc` (here and in isa_pool_target / isa_ins_softmax_bwd; 0 = n): the n rows are n/nsrc decoder iterations over the same
 * nsrc images - row b reads merge / ins of image b % nsrc with its own idx[b], so the fronts of all iterations
 * (attenet2.py:384-399) are one launch.  isa_concat_aux's `mask_n` is the same for its shared mask_all map. */
/* DecoderLayer.sample (attenet2.py:304-324): first argmax per row, on device.  race == NULL: the eval branch
 * (argmax of alpha).  race != NULL ([n, L] draws from Exp(1)): argmax of alpha / race = one draw from
 * Multinomial(alpha), the training branch (torch.multinomial's own single-sample form). */
int isa_row_argmax(const float* a, const float* race, int32_t n, int64_t L, int32_t* out, void* stream);
/* sem_seg_argmax = GT.argmax(1) (reseg.py:118) of the int64 one-hot target [n,2,h,w] as an fp32 {0,1} map [n, h*w] */
int isa_onehot_map(const int64_t* onehot, int32_t n, int64_t hw, float* out, void* stream);
/* F.dropout2d masks (utils.py:984,1104-1110) from uniform draws: out = (u < keep) / keep */
int isa_dropout_mask(const float* u, int64_t n, float keep, float* out, void* stream);
/* softmax over the channels of an NHWC logit map, written NCHW fp32 (Model.predict, model.py:486) */
int isa_softmax_nchw(const isa_tensor* x, float* out, void* stream);
/* UpDecoderLayer.resize (utils.py:841-846): f x f max-pool of the instance plane / of an fp32 map */
int isa_pool_target(const int64_t* ins, const int32_t* idx, const float* src, int32_t nobj, int32_t n,
                    int32_t H, int32_t W, int32_t f, float* out, int32_t nsrc, void* stream);
/* mask_all + conPosition channels (utils.py:1027-1045,1085) written into a concat slice */
int isa_concat_aux(const isa_tensor* dst, const float* mask_all, const int32_t* s_t, int32_t W_full,
                   int32_t f, int32_t nb, int32_t mask_n, void* stream);
/* UpAttenLayer.Mask (utils.py:1047-1056): out = up * softmax2(bilinear_x2(pred))[1] */
int isa_gate(const isa_tensor* up, const isa_tensor* pred, const isa_tensor* out, float* gmap, void* stream);
/* per-image sums for dice / focal / CE of a 2-class prediction (dice.py:10-51, multi_loss.py:27-42):
 * sums[b][0..6] = {sum p1*t, sum p1, sum t, sum focal, sum ce, sum p1^2, count}; stride 8 floats */
int isa_mask_loss_sums(const isa_tensor* pred, const float* target, const int64_t* onehot, float* sums,
                       void* stream);

/* ---- losses and hand-derived backward of the head (the reference relies on autograd) ------------
 * isa_head_loss: attenet2.py:239-290 on device (no host sync): per-level gradient coefficients,
 * REINFORCE advantage with the EMA baseline (device scalar), and scal[0..3] += {ins_cost without the
 * NaN entropy term (attenet2.py:77, zero gradient), criterion, ins_ce_loss, ins_dice_loss}/max_iter.
 * iters > 1: the rows of `iters` decoder iterations at once - sums [5][iters*B][8], alpha / s_t / adv [iters*B], coef
 * [5][iters*B][4] - processed in iteration order, the EMA baseline carried from one to the next (attenet2.py:266). */
int isa_head_loss(const float* sums /*[5][B][8]*/, const float* alpha, const int32_t* s_t, int64_t L, int32_t B,
                  const float* level_w /*host[5]*/, float ce_weight, float lambda_l, float lambda_r, float inv_iter,
                  float* baseline, int32_t training, float* coef /*[5][B][4]*/, float* adv /*[B]*/,
                  float* scal /*[4]*/, int32_t iters, void* stream);
/* trainer-side CE + Dice on the semantic logits (model.py:255-269): scal = {ce, dice}, coef[B][4] */
int isa_sem_loss(const float* sums /*[B][8]*/, int32_t B, float* coef, float* scal, void* stream);
int isa_mask_loss_grad(const isa_tensor* pred, const float* target, const int64_t* onehot, const float* coef /*[B][4]*/,
                       const isa_tensor* dpred, int32_t accumulate, void* stream);
int isa_ins_softmax_bwd(const float* alpha, const int64_t* ins, const int32_t* idx, const int32_t* s_t,
                        const float* adv, int32_t n, int32_t nobj, int64_t L, float* dmerge /*[nsrc, L]*/, int32_t nsrc,
                        void* stream);
int isa_maskbn_bwd(const isa_tensor* e, const float* sem, const float* dmerge, const float* mean_var,
                   const float* w, float eps, const float* am, int32_t train, float* red4 /*zeroed*/, float* k2,
                   float* dw, float* db, const isa_tensor* de, int32_t accumulate, void* stream);
int isa_sp_bwd(const isa_tensor* dout, const isa_tensor* x, const float* beta, const float* m, const float* dot,
               const float* rowstat, const float* chansum, const float* scale, const float* mean,
               const float* invstd, const float* wv, const float* lh, const float* fcw, float count, int32_t train,
               float* scratch /*zeroed: 3C + 2*rup(n,4) + 2*n*L floats*/, const isa_tensor* dx, int32_t accumulate,
               float* d_gamma, float* d_beta, float* d_wv, float* d_bv, float* d_lh, float* d_fcw, float* d_fcb,
               void* stream);
int isa_gate_bwd(const isa_tensor* dout, const isa_tensor* up, const float* gmap, const isa_tensor* dup,
                 int32_t acc_up, float* du /*zeroed [n*H*W]*/, const isa_tensor* dpred, int32_t acc_pred, void* stream);
int isa_se_bwd(const isa_tensor* dxa, const isa_tensor* x, const float* gate, const float* hid, const float* mean,
               const float* w1, const float* w2, int32_t hidden, float* dg /*zeroed [n*c]*/, float* dmean /*[n*c]*/,
               float* dw1, float* db1, float* dw2, float* db2, const isa_tensor* dx, int32_t accumulate, void* stream);
/* dst (+)= src * s[n,c]  (Dropout2d backward on a residual sum).  src may hold G * dst.n images (a tensor that was
 * broadcast to G statistic groups, each with its own per-image scale): dst[b] (+)= sum_g src[g*n+b] * s[g*n+b]. */
int isa_scale_bc(const isa_tensor* src, const float* s_bc, const isa_tensor* dst, int32_t accumulate, void* stream);
/* optimizer on the flat parameter buffer (model.py:145-166,273-278): out += sum (g*scale)^2, then
 * clip_grad_norm_(max_norm) + Adadelta(lr, rho, eps, weight_decay) in one pass.  lr_dev (optional, device float[1])
 * overrides `lr` at run time, so a launch recorded in a hipGraph follows the scheduler (model.py:164,437). */
int isa_sqnorm(const float* g, int64_t n, float scale, float* out /*zeroed*/, void* stream);
int isa_adadelta(float* p, const float* g, float* sq, float* acc, int64_t n, float lr, float rho, float eps, float wd,
                 const float* sqnorm, float max_norm, float gscale, const float* lr_dev, void* stream);

/* ---- the reference's named attention operators (modules/utils.py; dead at HEAD) --
 * a19 ScaledDotProductAttention.forward (utils.py:316-327) as MultiHeadAttention calls it (utils.py:167-225):
 *     out = softmax(mask(q k^T / T)) v, attn[(head*b), lq, L] optional (fp32).  Few queries against L = H*W keys.
 *     Operands: k, v [bh, L, heads*d], q [bh, lq, heads*d], out [bh, lq, heads*d] with `heads` heads interleaved in a
 *     row - the layout the Linear projections produce, so the reference's head-major permute (utils.py:200-202) is
 *     never materialised; heads = 1 means plain contiguous [bh, L, d] operands (then bh = batch*heads).
 *     dtype ISA_F32 | ISA_BF16 | ISA_F16.  mask: bytes, non-zero = masked, [bh, lq, L] or, with mask_per_head,
 *     [heads*bh, lq, L] (the reference repeats the mask per head).  attn rows are ordered head-major ((h*bh + b)*lq + q)
 *     like the reference's (n*b) batch.
 *     d_k = d_v = 12 (config.py:22-25): split-L streaming kernel - K/V read once in 16-byte coalesced loads, LDS-staged
 *     tiles of `tile_keys` keys (0 = 512; 256/512/1024), lane = key, online softmax, shuffle merges, one partial per
 *     workgroup in `ws` and a flash-decoding merge.  Other head dims <= 32: a one-workgroup-per-query fallback
 *     (heads = 1, f32/bf16).  ws: fp32 scratch, bh*heads*lq*(nsplit*(2+d)+2) floats (64 K floats always suffice). */
int isa_sdp_attention(const void* q, const void* k, const void* v, const uint8_t* mask, void* out,
                      float* attn, int32_t bh, int32_t lq, int64_t L, int32_t dk, int32_t dv,
                      float temperature, int32_t dtype, int32_t heads, int32_t mask_per_head,
                      float* ws, int64_t ws_floats, int32_t tile_keys, void* stream);
/* Backward of isa_sdp_attention (what autograd computes for utils.py:316-327 with dropout off), from the tensors the
 * forward pass returns: attn (normalised probabilities, head-major rows, zero on masked keys) and out.  dq: fp32
 * [bh, lq, heads*d], ZERO-INITIALISED by the caller (partials are added atomically); dk, dv: storage dtype, same layout
 * as k, v, written once.  One streaming pass over K and V.  lq <= 8, d <= 32, lq*d <= 96.  `bh` is the batch (rows of
 * q), `heads` the interleaved heads per row, exactly as in the forward call. */
int isa_sdp_attention_bwd(const void* q, const void* k, const void* v, const float* attn, const void* out,
                          const void* dout, float* dq, void* dk, void* dv, int32_t bh, int32_t lq, int64_t L,
                          int32_t dk_dim, int32_t dv_dim, float temperature, int32_t dtype, int32_t heads, void* stream);
/* The `last=True` branch (utils.py:203-209, 310-313): out[(h*b), lq, L] = q k^T (ScaledDotProductAttention) or
 * sigmoid(q k^T) (MultiHeadAttention: sigmoid != 0), no temperature, no softmax; same operand layout. */
int isa_sdp_scores(const void* q, const void* k, float* out, int32_t b, int32_t heads, int32_t lq, int64_t L,
                   int32_t d, int32_t dtype, int32_t sigmoid, void* stream);
/* y[r,:] = LayerNorm(W x[r,:] + bias + residual[r,:]): the small dense layers around the attention core (w_qs on the few
 * query rows, fc + layer_norm on the attended rows, utils.py:189-201).  fp32, n <= 64; gamma == NULL: no LayerNorm. */
int isa_linear_ln(const float* x, const float* w, const float* bias, const float* residual, const float* gamma,
                  const float* beta, float eps, int32_t rows, int32_t k, int32_t n, float* y, void* stream);
/* out = InstanceNorm2d(x + res) (utils.py:301-302: affine=False, biased variance); sums: zeroed float[n][2c]. */
int isa_instance_norm_res(const isa_tensor* x, const isa_tensor* res, const isa_tensor* out, float eps, float* sums,
                          void* stream);
/* a20 _ScalePDAttention core (utils.py:276-299): per-pixel softmax over the 3x3 neighbourhood at `dilation` */
int isa_local_attention(const isa_tensor* q, const isa_tensor* k, const isa_tensor* v, const float* nomask,
                        const isa_tensor* out, int32_t dilation, void* stream);
/* a21 Decoder.forward (utils.py:59-69): out[b,p] = sigmoid(<q[b,:], enc[b,p,:]>) */
int isa_point_query(const float* q, const isa_tensor* enc, float* out, void* stream);

/* ---- input expansion: ImageEx + ToTensor + Standardization, code/lib/utils.py:90-113 and
 * code/lib/preprocess.py:192-195.  rgb: uint8 [n,h,w,3] (device); out: NHWC view with c == 21 (ld 24: the three pad
 * channels are written as zero).  out = ([rgb, lab, hsv, yuv, ycbcr, hed, yiq] - 0.5) * 2.                     */
int isa_image_ex(const uint8_t* rgb, const isa_tensor* out, void* stream);

/* ---- target expansion (the tail of the collate function): AlignCollate.__call__,
 * code/lib/dataset.py:349-379, widens the uint8 instance planes to int64 [bs,K,h,w] and builds the int64 semantic
 * one-hot [bs,2,h,w] on the host.  Here: ins uint8 [n,h,w,k] (the array `instance_annotations` of :349-351 before
 * the cast) -> ins_out int64 [n,k,h,w], values preserved; sem uint8 [n,h,w] (may be NULL) -> sem_out int64 [n,2,h,w],
 * channel c = (sem == c) as np.eye(2)[sem] (:356-361; a value > 1 is an IndexError there and all-zero here).  k <= 252. */
int isa_collate_targets(const uint8_t* ins, const uint8_t* sem, int32_t n, int32_t h, int32_t w, int32_t k,
                        int64_t* ins_out, int64_t* sem_out, void* stream);

/* ---- exact augmentations: the index-permuting part of AlignCollate.__preprocess,
 * code/lib/dataset.py:185-233 - horizontal flip, vertical flip, transpose, rotation by k*90 degrees (preprocess.py:171,
 * 218,286,323; PIL FLIP_LEFT_RIGHT / FLIP_TOP_BOTTOM / TRANSPOSE / rotate(expand=True)), in the reference's order.
 * src: uint8 [n,h,w,c] (c = 3 image, 1 semantic map, 32 instance planes), out of place; dst: [n,h,w,c], or [n,w,h,c] when
 * swap_axes != 0 - an op that transposes XOR turns an odd number of quarters exchanges the axes, so one call covers the
 * images whose ops agree in that parity (any mix on square images).  The reference augments the original, non-square
 * image (e.g. 2362 x 672) before the resize.
 * ops_dev: int32 [n] on the device, per image: bit0 hflip, bit1 vflip, bit2 transpose, bits 3-4 = quarter turns
 * counter-clockwise (rot_angle / 90).  The random draws stay on the host as in the reference (:186,198,210,222). */
int isa_d4_augment(const uint8_t* src, uint8_t* dst, int32_t n, int32_t h, int32_t w, int32_t c, int32_t swap_axes,
                   const int32_t* ops_dev, void* stream);

/* ---- the non-D4 augmentations the reference ships enabled (settings/CVPPP/training_settings.py:40 ROTATION, :50
 * CENTER_CUT), AlignCollate.__preprocess code/lib/dataset.py:236-269 ------------------------------------------------
 * Rotation by `angle` degrees (an integer in [-9, 9], dataset.py:237-239) with expand=True.  The geometry - output size
 * and the six affine coefficients - is Image.rotate's own Python float arithmetic and stays on the host
 * (isa_amd.data.rotate_geometry); h, w are the expanded output dims.
 * nearest (annotation planes and semantic map, preprocess.py:311-327 Image.NEAREST): Pillow's Geometry.c affine_fixed,
 *   coef = the six 16.16 fixed-point values {a0,a1,a2,a3,a4,a5} (HOST array), zero fill; any channel count.
 * bilinear (the RGB image, preprocess.py:330-365 rotate_with_random_bg): ImagingGenericTransform + bilinear_filter32RGB
 *   in IEEE double, (UINT8) truncation; pixels whose sample point lies outside the source take bg (HOST array, c bytes:
 *   the drawn background colour - white, black, int(mean) or int(median) of the source); c <= 4.
 * Both are bit-identical to the installed Pillow through oracle/rotate_ref.py (tests/test_oracle_rotate.py). */
int isa_rotate_nearest_u8(const uint8_t* src, int32_t n, int32_t h0, int32_t w0, int32_t c, uint8_t* dst, int32_t h,
                          int32_t w, const int32_t* coef_host, void* stream);
int isa_rotate_bilinear_u8(const uint8_t* src, int32_t n, int32_t h0, int32_t w0, int32_t c, uint8_t* dst, int32_t h,
                           int32_t w, const double* matrix_host, const uint8_t* bg_host, void* stream);
/* CenterCut (preprocess.py:239-264, dataset.py:252-269).
 * cover_rows: single[n,h,w] = 1 where exactly one instance plane covers the pixel (`ins_all == 1`, the candidate
 *   centres in row-major order), row_counts[n,h] = their number per row - the host picks the drawn candidate from h ints
 *   and one row of `single` instead of downloading the planes.
 * plane_sums: sums[n,c] (zeroed by the caller) += the sum of every channel over the window [y0,y0+h) x [x0,x0+w): the
 *   reference keeps a plane when its window sums to more than 30.
 * crop_planes: dst[n,h,w,cd] = the window of the channels chan_dev[j] (device int32 [cd]; -1 or out of range = a zero
 *   plane, NULL = identity): crop, compaction of the surviving planes and the zero planes up to 32 (dataset.py:304-311)
 *   in one pass; with c = cd = 3 / 1 the image and semantic-map crops. */
int isa_cover_rows_u8(const uint8_t* planes, int32_t n, int32_t h, int32_t w, int32_t k, uint8_t* single,
                      int32_t* row_counts, void* stream);
int isa_plane_sums_u8(const uint8_t* src, int32_t n, int32_t H, int32_t W, int32_t c, int32_t y0, int32_t x0, int32_t h,
                      int32_t w, int64_t* sums, void* stream);
int isa_crop_planes_u8(const uint8_t* src, int32_t n, int32_t H, int32_t W, int32_t c, int32_t y0, int32_t x0, uint8_t* dst,
                       int32_t h, int32_t w, int32_t cd, const int32_t* chan_dev, void* stream);

/* Nearest-neighbour resize of annotation planes: the `ann_resizer` of AlignCollate
 * (code/lib/dataset.py:162,168,293-320 -> utils.py:26-27 -> PIL Image.resize(NEAREST), once per instance plane and
 * semantic map on the host).  src uint8 [n,h0,w0,c] -> dst uint8 [n,h,w,c], out of place; source row/column tables
 * follow Pillow's ImagingScaleAffine exactly (half-step start, step accumulated in double), h, w <= 768.          */
int isa_resize_nearest_u8(const uint8_t* src, int32_t n, int32_t h0, int32_t w0, int32_t c, uint8_t* dst,
                          int32_t h, int32_t w, void* stream);

/* Bilinear image resize: the `img_resizer` of AlignCollate and Prediction (code/lib/dataset.py:160-161,
 * 166-167, prediction.py:37 -> utils.py:26-27 -> PIL Image.resize(BILINEAR)) for uint8 images [n,h0,w0,c] (c <= 4) ->
 * [n,h,w,c], bit-identical to Pillow's Resample.c (anti-aliased triangle filter, 22-bit fixed-point coefficients,
 * horizontal pass into a uint8 intermediate, then vertical).  ws: device scratch of isa_resize_bilinear_ws_bytes(...)
 * bytes (coefficient tables + the intermediate image); ISA_ENOMEM when smaller. */
int64_t isa_resize_bilinear_ws_bytes(int32_t n, int32_t h0, int32_t w0, int32_t c, int32_t h, int32_t w);
int isa_resize_bilinear_u8(const uint8_t* src, int32_t n, int32_t h0, int32_t w0, int32_t c, uint8_t* dst,
                           int32_t h, int32_t w, void* ws, int64_t ws_bytes, void* stream);

/* ---- boundary layout converters (the reference passes NCHW fp32: reseg.py:106-110) ----------- */
int isa_nchw_to_nhwc(const float* src, int32_t csrc, const isa_tensor* dst, void* stream);
int isa_nhwc_to_nchw(const isa_tensor* src, float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISA_KERNELS_H */
