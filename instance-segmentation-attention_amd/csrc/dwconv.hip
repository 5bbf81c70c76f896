// Depthwise 3x3 (pad 1, stride 1) on NHWC: forward, data gradient, weight gradient.
// Replaces nn.Conv2d(groups=C) in InvertedV1Residual / InvertedResidual / the instance stems
// (MobileNetDenseASPP.py:77,109; reseg.py:79,93).  No MFMA here: it is a 9-tap per-channel
// stencil, pure HBM streaming.  Layout choices for gfx950:
//   * a lane owns 8 (bf16) / 4 (f32) consecutive channels = one 16-byte vector, and walks a
//     horizontal strip of outputs with a register sliding window, so each input vector is loaded
//     once per row it participates in (3 loads per output, 2 of them L1/L2 hits);
//   * consecutive lanes take consecutive channel groups => every load/store instruction covers
//     whole contiguous NHWC pixel rows;
//   * the lazy BN+ReLU6 prologue is applied to each loaded vector in registers; zero padding is
//     applied AFTER it (the pad value of the true tensor is 0, not act(shift));
//   * per-channel sum/sumsq of the output (next BN's batch statistics) accumulate in registers,
//     are combined per workgroup with LDS float atomics, then one global atomic per channel.
#include "common.hpp"

// LDS-tiled v2 (dwconv_tiled.hip): taken whenever the channel count is a multiple of 8
int dw2_forward(const isa_tensor* x, const isa_pro* pro, const void* w, const float* bias, const isa_tensor* y,
                float* stats, int accumulate, void* stream);
int dw2_wgrad(const isa_tensor* x, const isa_pro* pro, const isa_tensor* dy, float* dw, float* dbias, int csrc,
              float* ws, long ws_floats, isa_slab_arena* defer, void* stream);

namespace {

constexpr int STRIP = 16;
constexpr int WSTRIP = 64;       // weight-gradient strips are longer: the per-strip reduction tail is amortised 4x

struct DwParams {
    const void* x; const void* w; const float* bias; void* y;
    int n, h, w_, c, ldx, ldy, wld;
    ProDev pro;
    float* stats; int accumulate;
    long items; int nstrips, cg;
};

// nv = number of real channels in this lane's group (channel tail of views like the 21-ch input)
template <typename T, int CH>
__device__ __forceinline__ void loadv(const T* p, float (&v)[CH], int nv) {
    if (nv >= CH) {
        if constexpr (CH == 8) { load8<T>(p, v); }
        else {
            f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = a[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = i < nv ? st<T>::ld(p + i) : 0.f;
    }
}
template <typename T, int CH>
__device__ __forceinline__ void storev(T* p, const float (&v)[CH], int nv) {
    if (nv >= CH) {
        if constexpr (CH == 8) { store8<T>(p, v); }
        else {
            f32x4 a;
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = v[i];
            *reinterpret_cast<f32x4*>(p) = a;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CH; ++i) if (i < nv) st<T>::stv(p + i, v[i]);
    }
}

template <typename T, int CH, bool HAS_PRO, int ACT>
__global__ __launch_bounds__(256) void dw_fwd_kernel(DwParams p) {
    extern __shared__ float red[];        // [2*C] when stats
    if (p.stats) {
        for (int i = threadIdx.x; i < 2 * p.c; i += 256) red[i] = 0.f;
        __syncthreads();
    }
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const bool active = item < p.items;
    float ssum[CH], ssq[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
    int c0 = 0, nv = CH;
    if (active) {
        const int cgi = (int)(item % p.cg); long q = item / p.cg;
        const int s = (int)(q % p.nstrips); q /= p.nstrips;
        const int y = (int)(q % p.h); const int b = (int)(q / p.h);
        c0 = cgi * CH;
        nv = min(CH, p.c - c0);
        const int x0 = s * STRIP, x1 = min(p.w_, x0 + STRIP);
        const T* xin = reinterpret_cast<const T*>(p.x);
        const T* wp = reinterpret_cast<const T*>(p.w);
        float wt[9][CH];
#pragma unroll
        for (int t = 0; t < 9; ++t) loadv<T, CH>(wp + (long)t * p.wld + c0, wt[t], CH);
        float sc[CH], sh[CH], bs[CH], bv[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = min(c0 + j, p.c - 1);
            sc[j] = (HAS_PRO && p.pro.scale) ? p.pro.scale[c] : 1.f;
            sh[j] = (HAS_PRO && p.pro.shift) ? p.pro.shift[c] : 0.f;
            bs[j] = (HAS_PRO && p.pro.bscale) ? p.pro.bscale[(long)b * p.c + c] : 1.f;
            bv[j] = p.bias ? p.bias[c] : 0.f;
        }
        // three running accumulators: outputs at columns xc-1, xc, xc+1
        float a0[CH], a1[CH], a2[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) { a0[j] = 0.f; a1[j] = 0.f; a2[j] = 0.f; }
        T* yout = reinterpret_cast<T*>(p.y);
        for (int xc = x0 - 1; xc <= x1; ++xc) {        // input column
            if (xc >= 0 && xc < p.w_) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1;
                    if (yy < 0 || yy >= p.h) continue;
                    float v[CH];
                    loadv<T, CH>(xin + (((long)b * p.h + yy) * p.w_ + xc) * p.ldx + c0, v, nv);
                    if constexpr (HAS_PRO) {
#pragma unroll
                        for (int j = 0; j < CH; ++j) v[j] = act_t<ACT>(fmaf(v[j], sc[j], sh[j]), p.pro.act);
                        if (p.pro.bscale) {
#pragma unroll
                            for (int j = 0; j < CH; ++j) v[j] *= bs[j];
                        }
                    }
                    // input (yy,xc) feeds output (y, xc+1-tx) through tap (dy, tx)
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        a0[j] = fmaf(v[j], wt[dy * 3 + 2][j], a0[j]);   // output xc-1 uses tx=2
                        a1[j] = fmaf(v[j], wt[dy * 3 + 1][j], a1[j]);   // output xc   uses tx=1
                        a2[j] = fmaf(v[j], wt[dy * 3 + 0][j], a2[j]);   // output xc+1 uses tx=0
                    }
                }
            }
            const int xo = xc - 1;                       // column whose accumulator is complete
            if (xo >= x0 && xo < x1) {
                float o[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    o[j] = a0[j] + bv[j];
                    ssum[j] += o[j]; ssq[j] += o[j] * o[j];
                }
                T* dst = yout + (((long)b * p.h + y) * p.w_ + xo) * p.ldy + c0;
                if (p.accumulate) {
                    float old[CH];
                    loadv<T, CH>(dst, old, nv);
#pragma unroll
                    for (int j = 0; j < CH; ++j) o[j] += old[j];
                }
                storev<T, CH>(dst, o, nv);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; a2[j] = 0.f; }
        }
    }
    if (p.stats) {
        if (active) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (j < nv) {
                    atomicAdd(&red[c0 + j], ssum[j]);
                    atomicAdd(&red[p.c + c0 + j], ssq[j]);
                }
            }
        }
        __syncthreads();
        float* rep = p.stats + (blockIdx.x & (ISA_STAT_R - 1)) * 2 * p.c;
        for (int i = threadIdx.x; i < 2 * p.c; i += 256)
            if (red[i] != 0.f) atomicAdd(rep + i, red[i]);
    }
}

struct DwWgParams {
    const void* x; const void* dy; float* dw; float* dbias;
    int n, h, w_, c, ldx, ldd;
    ProDev pro;
    long items; int nstrips, cg; int csrc; int cg_pad;
    float* ws;                 // [gridDim.x][10*C] per-workgroup partial sums
};

// dw[c][t] (reference [C,1,3,3] layout) += sum_p dy[p,c] * xt[p + off(t), c];  dbias[c] += sum_p dy[p,c]
// Persistent: at most 512 workgroups; a lane keeps ONE channel group for its whole life (items are
// laid out strip-major, channel-group-minor and the grid stride is a multiple of the group count), walks
// many row strips accumulating its 9x8 products in registers and flushes once (LDS, then one global
// atomic per (tap, channel) per workgroup).
template <typename T, int CH, bool HAS_PRO, int ACT>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(DwWgParams p) {
    extern __shared__ float red[];        // [10*C]
    for (int i = threadIdx.x; i < 10 * p.c; i += 256) red[i] = 0.f;
    __syncthreads();
    const int cg_pad = p.cg_pad;                          // power of two >= cg, <= 256
    const int cgi = threadIdx.x & (cg_pad - 1);
    const int ssub = threadIdx.x / cg_pad, spb = 256 / cg_pad;      // strips per block pass
    if (cgi < p.cg) {
        const int c0 = cgi * CH;
        const int nv = min(CH, p.c - c0);
        const T* xin = reinterpret_cast<const T*>(p.x);
        const T* din = reinterpret_cast<const T*>(p.dy);
        float sc[CH], sh[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = min(c0 + j, p.c - 1);
            sc[j] = (HAS_PRO && p.pro.scale) ? p.pro.scale[c] : 1.f;
            sh[j] = (HAS_PRO && p.pro.shift) ? p.pro.shift[c] : 0.f;
        }
        float acc[9][CH], db[CH];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[t][j] = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j) db[j] = 0.f;
        const long nstrip_total = (long)p.n * p.h * p.nstrips;
        for (long sidx = (long)blockIdx.x * spb + ssub; sidx < nstrip_total; sidx += (long)gridDim.x * spb) {
            const unsigned su = (unsigned)sidx;
            const unsigned q = su / (unsigned)p.nstrips;
            const int s = (int)(su - q * (unsigned)p.nstrips);
            const int b = (int)(q / (unsigned)p.h), y = (int)(q - (unsigned)b * (unsigned)p.h);
            const int x0 = s * WSTRIP, x1 = min(p.w_, x0 + WSTRIP);
            float bs[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j)
                bs[j] = (HAS_PRO && p.pro.bscale) ? p.pro.bscale[(long)b * p.c + min(c0 + j, p.c - 1)] : 1.f;
            float d0[CH], d1[CH], d2[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) { d0[j] = 0.f; d1[j] = 0.f; d2[j] = 0.f; }
            const T* drow = din + (((long)b * p.h + y) * p.w_) * p.ldd + c0;
            for (int xc = x0 - 1; xc <= x1; ++xc) {
#pragma unroll
                for (int j = 0; j < CH; ++j) { d0[j] = d1[j]; d1[j] = d2[j]; }
                const int xn = xc + 1;
                if (xn >= x0 && xn < x1) loadv<T, CH>(drow + (long)xn * p.ldd, d2, nv);
                else {
#pragma unroll
                    for (int j = 0; j < CH; ++j) d2[j] = 0.f;
                }
                if (xc >= x0 && xc < x1) {
#pragma unroll
                    for (int j = 0; j < CH; ++j) db[j] += d1[j];
                }
                if (xc < 0 || xc >= p.w_) continue;
#pragma unroll
                for (int dyy = 0; dyy < 3; ++dyy) {
                    const int yy = y + dyy - 1;
                    if (yy < 0 || yy >= p.h) continue;
                    float v[CH];
                    loadv<T, CH>(xin + (((long)b * p.h + yy) * p.w_ + xc) * p.ldx + c0, v, nv);
                    if constexpr (HAS_PRO) {
#pragma unroll
                        for (int j = 0; j < CH; ++j) v[j] = act_t<ACT>(fmaf(v[j], sc[j], sh[j]), p.pro.act);
                        if (p.pro.bscale) {
#pragma unroll
                            for (int j = 0; j < CH; ++j) v[j] *= bs[j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        acc[dyy * 3 + 2][j] = fmaf(v[j], d0[j], acc[dyy * 3 + 2][j]);
                        acc[dyy * 3 + 1][j] = fmaf(v[j], d1[j], acc[dyy * 3 + 1][j]);
                        acc[dyy * 3 + 0][j] = fmaf(v[j], d2[j], acc[dyy * 3 + 0][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < CH; ++j) if (j < nv) atomicAdd(&red[t * p.c + c0 + j], acc[t][j]);
#pragma unroll
        for (int j = 0; j < CH; ++j) if (j < nv) atomicAdd(&red[9 * p.c + c0 + j], db[j]);
    }
    __syncthreads();
    float* slab = p.ws + (long)blockIdx.x * 10 * p.c;       // plain coalesced stores; dw_wgrad_reduce_kernel folds
    for (int i = threadIdx.x; i < 10 * p.c; i += 256) slab[i] = red[i];
}

// dw[c][t] += sum over workgroups of slab[t*C + c]; dbias[c] += slab[9*C + c]
constexpr int DW_RSPLIT = 16;
__global__ __launch_bounds__(256) void dw_wgrad_reduce_kernel(const float* ws, int nblk, int C, int csrc, float* dw, float* dbias) {
    const int split = blockIdx.y;
    const int per = (nblk + DW_RSPLIT - 1) / DW_RSPLIT;
    const int b0 = split * per, b1 = min(nblk, b0 + per);
    if (b0 >= b1) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 10 * C; i += gridDim.x * 256) {
        float s = 0.f;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += ws[(long)b * 10 * C + i];
        const int t = i / C, c = i - t * C;
        if (c >= csrc) continue;
        if (t < 9) atomicAdd(dw + c * 9 + t, s);
        else if (dbias) atomicAdd(dbias + c, s);
    }
}

template <typename T, int CH>
int launch_fwd(DwParams& p, bool has_pro, hipStream_t s) {
    p.cg = (p.c + CH - 1) / CH;
    p.wld = ((p.c + 7) / 8) * 8;            // packed [9][rup(C,8)]
    p.nstrips = (p.w_ + STRIP - 1) / STRIP;
    p.items = (long)p.n * p.h * p.nstrips * p.cg;
    const int grid = cdiv(p.items, 256);
    const size_t lds = p.stats ? 2 * (size_t)p.c * 4 : 0;
    if (has_pro && p.pro.act == ISA_ACT_RELU6) hipLaunchKernelGGL((dw_fwd_kernel<T, CH, true, ISA_ACT_RELU6>), dim3(grid), dim3(256), lds, s, p);
    else if (has_pro) hipLaunchKernelGGL((dw_fwd_kernel<T, CH, true, ACT_RT>), dim3(grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((dw_fwd_kernel<T, CH, false, ISA_ACT_NONE>), dim3(grid), dim3(256), lds, s, p);
    return launch_status();
}

int dw_forward(const isa_tensor* x, const isa_pro* pro, const void* w, const float* bias,
               const isa_tensor* y, float* stats, int accumulate, void* stream) {
    if (!tensor_ok(x, 8) || !tensor_ok(y, 8) || !w || x->dtype != y->dtype) return ISA_EINVAL;
    if (x->n != y->n || x->h != y->h || x->w != y->w || x->c != y->c) return ISA_EINVAL;
    // the tiled kernels work on whole 8-channel vectors: views whose row pitch covers rup(C,8) qualify (the 21-channel
    // input lives in 24-channel rows; pad lanes compute garbage that no consumer reads and no statistic sees)
    const int c8 = (x->c + 7) / 8 * 8;
    if (x->ld >= c8 && y->ld >= c8) return dw2_forward(x, pro, w, bias, y, stats, accumulate, stream);
    if (pro && pro->fin) { if (int rc = fin_standalone(pro, x->c, tensor_groups(x), as_stream(stream))) return rc; }   // no in-kernel form here
    DwParams p{};
    p.x = x->data; p.w = w; p.bias = bias; p.y = y->data;
    p.n = x->n; p.h = x->h; p.w_ = x->w; p.c = x->c; p.ldx = x->ld; p.ldy = y->ld;
    p.pro = make_pro(pro); p.stats = stats; p.accumulate = accumulate;
    const bool has_pro = !pro_trivial(p.pro);
    if (x->dtype == ISA_BF16) return launch_fwd<bf16_t, 8>(p, has_pro, as_stream(stream));
    return launch_fwd<float, 4>(p, has_pro, as_stream(stream));
}

template <typename T, int CH>
int launch_wg(DwWgParams& p, bool has_pro, long ws_floats, hipStream_t s) {
    p.cg = (p.c + CH - 1) / CH;
    p.cg_pad = 1;
    while (p.cg_pad < p.cg) p.cg_pad <<= 1;
    if (p.cg_pad > 256) return ISA_EINVAL;
    p.nstrips = (p.w_ + WSTRIP - 1) / WSTRIP;
    p.items = (long)p.n * p.h * p.nstrips * p.cg;
    const long strips = (long)p.n * p.h * p.nstrips;
    if (strips >= (1L << 32)) return ISA_EINVAL;
    int grid = grid_cap(cdiv(strips, 256 / p.cg_pad), 1024);
    const long ws_cap = ws_floats / (10L * p.c);
    if (ws_cap < 1) return ISA_EINVAL;
    if (grid > ws_cap) grid = (int)ws_cap;
    const size_t lds = 10 * (size_t)p.c * 4;
    if (has_pro && p.pro.act == ISA_ACT_RELU6) hipLaunchKernelGGL((dw_wgrad_kernel<T, CH, true, ISA_ACT_RELU6>), dim3(grid), dim3(256), lds, s, p);
    else if (has_pro) hipLaunchKernelGGL((dw_wgrad_kernel<T, CH, true, ACT_RT>), dim3(grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((dw_wgrad_kernel<T, CH, false, ISA_ACT_NONE>), dim3(grid), dim3(256), lds, s, p);
    hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(cdiv(10 * p.c, 256), DW_RSPLIT), dim3(256), 0, s, p.ws, grid, p.c, p.csrc, p.dw, p.dbias);
    return launch_status();
}

}  // namespace

extern "C" int isa_dwconv3x3(const isa_tensor* x, const isa_pro* pro, const void* w,
                             const float* bias, const isa_tensor* y, float* stats, void* stream) {
    return dw_forward(x, pro, w, bias, y, stats, 0, stream);
}

// w must be the tap-flipped packing of the forward weights (isa_pack_weights kind 5)
extern "C" int isa_dwconv3x3_dgrad(const isa_tensor* dy, const void* w, const isa_tensor* dx,
                                   int32_t accumulate, void* stream) {
    return dw_forward(dy, nullptr, w, nullptr, dx, nullptr, accumulate, stream);
}

extern "C" int isa_dwconv3x3_wgrad(const isa_tensor* x, const isa_pro* pro, const isa_tensor* dy,
                                   float* dw, float* dbias, int32_t csrc, float* ws, int64_t ws_floats, isa_slab_arena* defer,
                                   void* stream) {
    if (pro && pro->fin) { if (int rc = fin_standalone(pro, x->c, tensor_groups(x), as_stream(stream))) return rc; }   // no in-kernel form here
    if (!tensor_ok(x, 8) || !tensor_ok(dy, 8) || !dw || x->dtype != dy->dtype) return ISA_EINVAL;
    if (x->n != dy->n || x->h != dy->h || x->w != dy->w || x->c != dy->c) return ISA_EINVAL;
    if (!ws && !defer) return ISA_EINVAL;
    const int c8 = (x->c + 7) / 8 * 8;
    if (x->ld >= c8 && dy->ld >= c8) return dw2_wgrad(x, pro, dy, dw, dbias, csrc, ws, ws_floats, defer, stream);
    if (10 * (size_t)x->c * 4 > 60 * 1024) return ISA_EINVAL;
    DwWgParams p{};
    p.x = x->data; p.dy = dy->data; p.dw = dw; p.dbias = dbias;
    p.n = x->n; p.h = x->h; p.w_ = x->w; p.c = x->c; p.ldx = x->ld; p.ldd = dy->ld;
    p.pro = make_pro(pro);
    p.csrc = (csrc > 0 && csrc < x->c) ? csrc : x->c;
    if (!ws) return ISA_EINVAL;
    p.ws = ws;
    const bool has_pro = !pro_trivial(p.pro);
    if (x->dtype == ISA_BF16) return launch_wg<bf16_t, 8>(p, has_pro, ws_floats, as_stream(stream));
    return launch_wg<float, 4>(p, has_pro, ws_floats, as_stream(stream));
}
