// Shared device/host helpers for the gfx950 kernel library.  CDNA4 only: 64-wide wavefronts,
// MFMA 32x32 tiles, 160 KB LDS/CU.  No portability macros on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/isa_kernels.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ISA_WAVE 64
// per-channel statistics are accumulated into ISA_STAT_R replicas ([R][2C], replica = workgroup & 7)
// so that thousands of workgroups do not serialise on the same few L2 atomic addresses; consumers
// (isa_bn_finalize, isa_bn_bwd_apply) sum the replicas.
#define ISA_STAT_R 8

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int launch_status() {
    return hipGetLastError() == hipSuccess ? ISA_OK : ISA_ELAUNCH;
}

static inline bool tensor_ok(const isa_tensor* t, int align_elems) {
    if (!t || !t->data || t->n <= 0 || t->h <= 0 || t->w <= 0 || t->c <= 0 || t->ld < t->c) return false;
    if (t->dtype != ISA_F32 && t->dtype != ISA_BF16) return false;
    if (align_elems > 1) {
        size_t esz = t->dtype == ISA_F32 ? 4 : 2;
        if ((reinterpret_cast<uintptr_t>(t->data) % (align_elems * esz)) != 0) return false;
        if (t->ld % align_elems) return false;
    }
    return true;
}

// ---- storage-type traits --------------------------------------------------------------------
template <typename T> struct st;
template <> struct st<float> {
    static constexpr int dtype = ISA_F32;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void stv(float* p, float v) { *p = v; }
};
template <> struct st<bf16_t> {
    static constexpr int dtype = ISA_BF16;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void stv(bf16_t* p, float v) { *p = (bf16_t)v; }
};

// 8 consecutive elements <-> 8 floats (16 B for bf16, 2 x 16 B for f32)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}

// guarded forms: only the first `nv` (1..8) elements exist (channel tail of a view); the full-vector
// fast path is taken when nv == 8
template <typename T> __device__ __forceinline__ void load8g(const T* p, float (&v)[8], int nv) {
    if (nv >= 8) { load8<T>(p, v); return; }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = i < nv ? st<T>::ld(p + i) : 0.f;
}
template <typename T> __device__ __forceinline__ void store8g(T* p, const float (&v)[8], int nv) {
    if (nv >= 8) { store8<T>(p, v); return; }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < nv) st<T>::stv(p + i, v[i]);
}

// Sum over the lanes of a 16-lane row that share (lane % CLS), CLS = 4 or 8: DPP row rotations, no LDS traffic.  The
// end-of-kernel folds used __shfl_xor trees (ds_bpermute: a dependent LDS round trip per step) per value: a phase
// trace of dw_bn_bwd on a one-tile problem showed 26 k of the launch's 47 k cycles in its fold.  Lanes 0..CLS-1 of
// every ROW then hold a partial (16 / CLS lanes folded); the caller adds the four rows of the wave.
template <int CLS> __device__ __forceinline__ float row_fold(float v) {
    static_assert(CLS == 4 || CLS == 8, "lane classes per row");
    if constexpr (CLS == 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));                           // row_ror:8
    return v;
}

// ---- activations ----------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float z, int act) {
    switch (act) {
        case ISA_ACT_RELU:  return fmaxf(z, 0.f);
        case ISA_ACT_RELU6: return fminf(fmaxf(z, 0.f), 6.f);
        case ISA_ACT_LEAKY: return z > 0.f ? z : 0.01f * z;
        case ISA_ACT_TANH:  return tanhf(z);
        default:            return z;
    }
}
// derivative w.r.t. the pre-activation z (torch semantics: relu6/hardtanh grad is 1 on 0<z<6;
// threshold_backward for relu: z>0)
__device__ __forceinline__ float act_grad(float z, int act) {
    switch (act) {
        case ISA_ACT_RELU:  return z > 0.f ? 1.f : 0.f;
        case ISA_ACT_RELU6: return (z > 0.f && z < 6.f) ? 1.f : 0.f;
        case ISA_ACT_LEAKY: return z > 0.f ? 1.f : 0.01f;
        case ISA_ACT_TANH:  { float t = tanhf(z); return 1.f - t * t; }
        default:            return 1.f;
    }
}

// compile-time activation: ACT = ISA_ACT_RELU6 / ISA_ACT_NONE are specialised (branch-free inner loops),
// ACT = -1 takes the runtime switch (its tanhf branch alone triples the prologue code).  Hot kernels are
// instantiated for {RELU6, LEAKY where the prediction heads use it, runtime}.
constexpr int ACT_RT = -1;
template <int ACT> __device__ __forceinline__ float act_t(float z, int rt) {
    // v_med3_f32: one instruction for the clamp (fminf(fmaxf()) compiles to v_max + v_min); same result for every
    // input including NaN (-> 0, as fmaxf(NaN, 0) = 0)
    if constexpr (ACT == ISA_ACT_RELU6) return __builtin_amdgcn_fmed3f(z, 0.f, 6.f);
    else if constexpr (ACT == ISA_ACT_NONE) return z;
    else if constexpr (ACT == ISA_ACT_LEAKY) return z > 0.f ? z : 0.01f * z;
    else return act_apply(z, rt);
}
template <int ACT> __device__ __forceinline__ float act_grad_t(float z, int rt) {
    if constexpr (ACT == ISA_ACT_RELU6) return (z > 0.f && z < 6.f) ? 1.f : 0.f;
    else if constexpr (ACT == ISA_ACT_NONE) return 1.f;
    else if constexpr (ACT == ISA_ACT_LEAKY) return z > 0.f ? 1.f : 0.01f;
    else return act_grad(z, rt);
}

// device-side copy of isa_pro with nulls normalised
struct ProDev {
    const float* scale; const float* shift; const float* bscale; int act;
};
static inline ProDev make_pro(const isa_pro* p) {
    ProDev d{nullptr, nullptr, nullptr, ISA_ACT_NONE};
    if (p) { d.scale = p->scale; d.shift = p->shift; d.bscale = p->bscale; d.act = p->act; }
    return d;
}
static inline bool pro_trivial(const ProDev& p) {
    return !p.scale && !p.shift && !p.bscale && p.act == ISA_ACT_NONE;
}

// ---- BatchNorm finalize, stand-alone (isa_bn_finalize) and inside the consumer of the lazy tensor (isa_pro.fin) ------
struct FinDev {
    const float *stats, *gamma, *beta; float *rm, *rv, *scale, *shift, *mean, *invstd;
    float count, momentum, eps; int repeat;
};
static inline FinDev make_fin(const isa_pro* p) {
    FinDev f{}; f.repeat = 1;
    if (p && p->fin) {
        const isa_bn_fin* q = p->fin;
        f = FinDev{q->stats, q->gamma, q->beta, q->running_mean, q->running_var, q->scale, q->shift, q->mean, q->invstd,
                   q->count, q->momentum, q->eps, q->repeat < 1 ? 1 : q->repeat};
    }
    return f;
}
static inline bool fin_valid(const isa_pro* p) {         // scale/shift of the prologue must be the finalize's outputs
    return !p || !p->fin || (p->fin->stats && p->fin->scale && p->fin->shift && p->fin->scale == p->scale && p->fin->shift == p->shift);
}
// stand-alone launch for entry points without the in-kernel form (elementwise.hip)
int fin_standalone(const isa_pro* p, int c, int groups, hipStream_t s);

// Channel i of a BatchNorm finalize.  all_groups: the memory-writing form (isa_bn_finalize; the writer workgroup of a
// consumer): groups 0..G-1 in order, scale/shift/mean/invstd [G][c] and the running statistics; else group gsel only,
// nothing written to memory.  tab (LDS, or NULL): scale -> tab[i], shift -> tab[tab_ld + i] of group gsel.
// stats == NULL: eval mode, constants from the running statistics (stand-alone only).
__device__ __forceinline__ void bn_fin_channel(const FinDev& f, int i, int c, int groups, bool all_groups, int gsel,
                                               float* tab, int tab_ld) {
    const float g = f.gamma ? f.gamma[i] : 1.f, b = f.beta ? f.beta[i] : 0.f;
    const bool run = all_groups || !f.stats;
    float rmi = (run && f.rm) ? f.rm[i] : 0.f, rvi = (run && f.rv) ? f.rv[i] : 0.f;
    const int g0 = all_groups ? 0 : gsel, g1 = all_groups ? groups : gsel + 1;
    for (int gi = g0; gi < g1; ++gi) {
        float mean, var;
        if (f.stats) {
            const float* st = f.stats + (long)gi * ISA_STAT_R * 2 * c;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < ISA_STAT_R; ++r) { s1 += st[r * 2 * c + i]; s2 += st[r * 2 * c + c + i]; }
            mean = s1 / f.count;
            var = fmaxf(s2 / f.count - mean * mean, 0.f);
            if (all_groups)
                for (int k = 0; k < f.repeat; ++k) {
                    rmi = (1.f - f.momentum) * rmi + f.momentum * mean;
                    rvi = (1.f - f.momentum) * rvi + f.momentum * var * (f.count / fmaxf(f.count - 1.f, 1.f));
                }
        } else {
            mean = rmi; var = rvi;
        }
        const float inv = 1.0f / sqrtf(var + f.eps);
        const float sc = g * inv, sh = b - mean * g * inv;
        if (all_groups) {
            f.scale[gi * c + i] = sc;
            f.shift[gi * c + i] = sh;
            if (f.mean) f.mean[gi * c + i] = mean;
            if (f.invstd) f.invstd[gi * c + i] = inv;
        }
        if (tab && gi == gsel) { tab[i] = sc; tab[tab_ld + i] = sh; }
    }
    if (all_groups && f.stats) {
        if (f.rm) f.rm[i] = rmi;
        if (f.rv) f.rv[i] = rvi;
    }
}
// In a consumer kernel, called by all NTHREADS threads of every workgroup before the prologue constants are read:
// fills tab[ch - cbeg] / tab[tab_ld + ch - cbeg] (LDS) with scale / shift of this workgroup's statistic group for the
// channels [cbeg, cbeg + ncb) the workgroup works on; `writer` workgroups (exactly one per channel range of the launch)
// also write what isa_bn_finalize writes.  Ends with a __syncthreads().  No-op (false) without a pending finalize.
template <int NTHREADS>
__device__ __forceinline__ bool bn_fin_inline(const FinDev& f, int c, int groups, int gsel, float* tab, int tab_ld, int tid,
                                              int cbeg, int ncb, bool writer) {
    if (!f.stats) return false;
    for (int i = tid; i < ncb; i += NTHREADS)
        if (cbeg + i < c) bn_fin_channel(f, cbeg + i, c, groups, writer, gsel, tab - cbeg, tab_ld);
    __syncthreads();
    return true;
}
template <int NTHREADS>
__device__ __forceinline__ bool bn_fin_inline(const FinDev& f, int c, int groups, int gsel, float* tab, int tab_ld, int tid) {
    // the writer is the LAST workgroup in x: in the persistent tile loops it has the fewest tiles, its extra work hides there
    return bn_fin_inline<NTHREADS>(f, c, groups, gsel, tab, tab_ld, tid, 0, c,
                                   blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && blockIdx.z == 0);
}

// ---- wave reductions ------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// For `item = pixel*cg + channel_group` grid-stride loops: round the grid so that the stride (grid*256) is a multiple
// of cg.  A lane then keeps ONE channel group for its whole life and its per-channel partial sums are flushed
// once, not on every trip (cg = 3 for the 24-channel attention tensors: 2048*256 % 3 != 0 flushed every trip).
static inline int grid_keep_cg(int grid, int cg) {
    int m = cg;
    while (m % 2 == 0) m /= 2;                       // 256 already supplies the powers of two
    if (m <= 1 || grid < m) return grid;
    return grid - grid % m;
}

// In `item = pixel*cg + group` walks whose stride keeps a lane's group fixed (grid_keep_cg), lanes l, l+cg, l+2cg, ...
// of a wave hold partial sums of the SAME channel group.  Fold them (strided doubling tree, any cg < 64); lanes < cg
// end up with the wave totals and are the only ones that touch the LDS accumulator: same-address LDS atomics from
// every lane serialised and dominated these reductions.
__device__ __forceinline__ float fold_stride(float v, int cg, int lane) {
    for (int off = cg; off < 64; off <<= 1) {
        const float t = __shfl_down(v, off, 64);
        if (lane + off < 64) v += t;
    }
    return v;
}

// ---- statistic groups (isa_tensor.groups) -----------------------------------------------------
// A grouped launch is G independent sub-problems of n/G images each that differ only in their BatchNorm statistics
// and constants.  A workgroup works for ONE group: workgroups [g*per, (g+1)*per) of gridDim.x belong to group g
// (the host rounds gridDim.x to a multiple of G).  The kernel offsets its tensor pointers by g sub-problems, its
// per-channel pointers by g arrays, and then runs its usual loops with (bx, nbx) in place of (blockIdx.x, gridDim.x);
// slab and replica indices keep the absolute blockIdx.x.
struct GroupSel { int g, bx, nbx; };
__device__ __forceinline__ GroupSel group_sel(int G) {
    if (G <= 1) return GroupSel{0, (int)blockIdx.x, (int)gridDim.x};
    const int per = (int)gridDim.x / G;
    const int g = (int)blockIdx.x / per;
    return GroupSel{g, (int)blockIdx.x - g * per, per};
}
template <typename P> __device__ __forceinline__ P* goff(P* p, long off) { return p ? p + off : p; }
static inline int tensor_groups(const isa_tensor* t) { return (t && t->groups > 1) ? t->groups : 1; }
// grid of a grouped launch: a multiple of G, at least G
static inline long group_grid(long gx, int G) { if (G <= 1) return gx < 1 ? 1 : gx; gx = gx / G * G; return gx < G ? G : gx; }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int grid_cap(long blocks, int cap = 256 * 8) { return (int)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap); }

// K workgroup-wide sums added to dst[0..K) with ONE wave-wide atomic request per workgroup (sh: >= (blockDim/64)*K
// floats).  K single-lane atomicAdds per workgroup are K memory-side requests on the same line: 2048 workgroups x 4
// of them made a 15 us reduction take 116 us.
template <int K>
__device__ __forceinline__ void block_sums_atomic(float (&a)[K], float* sh, float* dst) {
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = wave_sum(a[k]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh[wave * K + k] = a[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < K) {
        float r = 0.f;
        for (int w = 0; w < nw; ++w) r += sh[w * K + threadIdx.x];
        if (r != 0.f) atomicAdd(dst + threadIdx.x, r);
    }
}

// ---- deferred slab folds (isa_slab_arena_*; defined in conv_wgrad.hip) -------------------------------
// Weight gradients are leaves of the backward graph: nothing reads dW before the optimizer.  A weight-gradient
// entry point that is handed an isa_slab_arena takes its partial-slab region from the arena (defer_ws) and records
// the second-stage fold in it (defer_push) instead of launching it; isa_slab_arena_flush folds every recorded slab
// set in a handful of launches (descriptors travel as kernel arguments, so a captured hipGraph keeps them by
// value).  All state lives in the caller's handle: no globals, no thread-locals.
#include <vector>
struct FoldDesc {
    const float* ws; float* dw; float* dbias; const int32_t* kmap;
    int kind;                    // 0: MFMA-fragment slabs (conv_wgrad family), 1: depthwise tile slabs [gx][gy][10*tk]
    int gx, gy, taps, groups_k, tn, tk, N, cin, ksrc, out_mode, rsplit;
    int first_block, blocks;
};
struct isa_slab_arena {
    float* base; long floats, used, peak;
    bool from_arena;             // between a defer_ws that handed out arena memory and the defer_push that records it
    std::vector<FoldDesc> tab;
};
// a == NULL: *ws / *ws_floats unchanged (immediate fold from the caller's workspace).  Otherwise the arena cursor and
// the floats left; ISA_ENOMEM when less than the minimum slab budget is left (never a silent fallback).
int defer_ws(isa_slab_arena* a, float** ws, long* ws_floats);
bool defer_push(isa_slab_arena* a, FoldDesc d, long used_floats);   // true: recorded, the caller skips its own reduce launch
