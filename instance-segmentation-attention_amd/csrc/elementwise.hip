// Streaming / reduction kernels around the convolutions: parameter repacking, BatchNorm finalize and
// backward, lazy-tensor materialisation (+residual), pooling, squeeze-excite, channel argmax.
// All are HBM-bound: 16-byte vectors per lane over the NHWC channel axis, consecutive lanes on
// consecutive channel groups, grid-strided with ~8 workgroups per CU; per-channel reductions go
// registers -> LDS float atomics -> one global atomic per channel per workgroup.
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// weight repacking (one launch for the whole parameter set)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_kernel(const isa_pack_entry* tab, const int32_t* kmap, const float* src, T* dst) {
    const isa_pack_entry e = tab[blockIdx.y];
    const float* s = src + e.src_off;
    T* d = dst + e.dst_off;
    const int32_t* km = e.kmap_off >= 0 ? kmap + e.kmap_off : nullptr;
    long total;
    const int rows = e.rows;
    switch (e.kind) {
        case 0: case 1: total = (long)rows * e.taps * e.kp; break;
        case 2: total = (long)rows * e.kp; break;
        case 3: total = (long)rows * 4 * e.kp; break;
        default: total = (long)9 * rows; break;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (e.kind == 0) {               // [N][taps][kp] <- src[N][K][taps]
            const int kd = (int)(i % e.kp); long q = i / e.kp;
            const int t = (int)(q % e.taps); const int n = (int)(q / e.taps);
            const int k = km ? km[kd] : (kd < e.k ? kd : -1);
            if (k >= 0) v = s[((long)n * e.k + k) * e.taps + t];
        } else if (e.kind == 1) {        // [rows=Kphys][taps][np] <- src[N][K][taps], taps flipped
            const int nd = (int)(i % e.kp); long q = i / e.kp;
            const int t = (int)(q % e.taps); const int kd = (int)(q / e.taps);
            const int k = km ? km[kd] : kd;
            if (k >= 0 && nd < e.n) v = s[((long)nd * e.k + k) * e.taps + (e.taps - 1 - t)];
        } else if (e.kind == 2) {        // convT fwd: [4*Co][kp] <- src[K][Co][2][2]
            const int kd = (int)(i % e.kp); const int row = (int)(i / e.kp);
            const int co_n = e.n;        // Co
            const int qd = row / co_n, co = row - qd * co_n;
            const int k = km ? km[kd] : (kd < e.k ? kd : -1);
            if (k >= 0) v = s[((long)k * co_n + co) * 4 + qd];
        } else if (e.kind == 3) {        // convT dgrad: [K][4][cop] <- src[K][Co][2][2]
            const int cod = (int)(i % e.kp); long q = i / e.kp;
            const int qd = (int)(q % 4); const int kd = (int)(q / 4);
            const int k = km ? km[kd] : kd;
            if (k >= 0 && cod < e.n) v = s[((long)k * e.n + cod) * 4 + qd];
        } else {                          // depthwise: [9][rows] <- src[C][9]; kind 5 = flipped
            const int cd = (int)(i % rows); const int t = (int)(i / rows);
            const int c = km ? km[cd] : (cd < e.n ? cd : -1);
            if (c >= 0) v = s[(long)c * 9 + (e.kind == 5 ? 8 - t : t)];
        }
        d[i] = (T)v;
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm finalize
// ------------------------------------------------------------------------------------------
// groups > 1: stats [G][R][2c] -> scale/shift/mean/invstd [G][c]; the running statistics take the G updates in group
// order (the reference runs the groups - its decoder iterations - one after the other through the same module);
// repeat > 1: the same update applied `repeat` times (a layer whose identical forward the reference runs `repeat` times).
__global__ void bn_finalize_kernel(FinDev f, int c, int groups) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c; i += gridDim.x * blockDim.x)
        bn_fin_channel(f, i, c, groups, true, 0, nullptr, 0);
}

// Ordered running-statistics updates of many train-mode BatchNorm layers in one launch (isa_bn_running_update):
// one workgroup per layer; the descriptors travel by value so a captured hipGraph keeps them.
constexpr int BN_UPD_CHUNK = 64;
struct BnUpdChunk { isa_bn_upd d[BN_UPD_CHUNK]; };
__global__ __launch_bounds__(256) void bn_running_update_kernel(BnUpdChunk ch, float momentum) {
    const isa_bn_upd u = ch.d[blockIdx.x];
    const int c = u.c;
    for (int i = threadIdx.x; i < c; i += 256) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < ISA_STAT_R; ++r) { s1 += u.stats[r * 2 * c + i]; s2 += u.stats[r * 2 * c + c + i]; }
        const float mean = s1 / u.count;
        const float var = fmaxf(s2 / u.count - mean * mean, 0.f);
        u.running_mean[i] = (1.f - momentum) * u.running_mean[i] + momentum * mean;
        u.running_var[i] = (1.f - momentum) * u.running_var[i] + momentum * var * (u.count / fmaxf(u.count - 1.f, 1.f));
    }
}

// ------------------------------------------------------------------------------------------
// generic vectorised NHWC walker
// ------------------------------------------------------------------------------------------
struct View { void* data; int n, h, w, c, ld; };
static inline View mkview(const isa_tensor* t) { return View{t->data, t->n, t->h, t->w, t->c, t->ld}; }

struct BnBwdParams {
    View dt, y, dy;
    const float *scale, *shift, *mean, *invstd, *bscale, *gamma, *red;
    float* out_red; float* dgamma; float* dbeta;
    float inv_count; int act, train;
    long pixels; int cg;
    int groups;                  // statistic groups (blockIdx.z): pixels / wk.pixels are per group
};

// Division-free walk over (pixel, 8-channel group): a workgroup is laid out as
// [256 >> sh pixels] x [1 << sh channel groups] (sh = ceil log2 of the group count), so a lane keeps
// ONE channel group for its whole life: per-channel constants and partial sums live in registers,
// addresses advance by adds only (64-bit integer division costs ~100 instructions on CDNA).
struct Walk { int cg, sh; long pixels; };
static inline Walk mkwalk(int c, long pixels) {
    Walk w; w.cg = (c + 7) / 8; w.sh = 0;
    while ((1 << w.sh) < w.cg) ++w.sh;
    if (w.sh > 8) w.sh = 8;
    w.pixels = pixels; return w;
}
// `iters` = pixel iterations a workgroup should own at least: kernels with a per-workgroup prologue/epilogue of
// O(C) global accesses (the BN-backward passes) must not be launched as thousands of one-iteration workgroups on
// the wide, low-resolution levels (4096 px x 512 ch: the epilogue atomics outnumbered the data 16:1)
static inline int walk_grid(const Walk& w, int iters = 1) { return grid_cap(cdiv(w.pixels, (long)(256 >> w.sh) * iters)); }

// z = scale*y+shift ; dz = dt * bscale * act'(z)
template <typename T, bool APPLY, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdParams p, Walk wk) {
    extern __shared__ float red[];              // [2*C]: reduce pass partial sums / apply pass folded sums
    const int C = p.y.c;
    if (p.groups > 1 && blockIdx.z) {           // this workgroup's statistic group: its images and its constants
        const long g = blockIdx.z, gp = g * wk.pixels;
        p.dt.data = reinterpret_cast<T*>(p.dt.data) + gp * p.dt.ld;
        p.y.data = reinterpret_cast<T*>(p.y.data) + gp * p.y.ld;
        if (APPLY) p.dy.data = reinterpret_cast<T*>(p.dy.data) + gp * p.dy.ld;
        p.scale = goff(p.scale, g * C); p.shift = goff(p.shift, g * C); p.mean = goff(p.mean, g * C);
        p.invstd = goff(p.invstd, g * C);
        p.bscale = goff(p.bscale, g * (wk.pixels / ((long)p.y.h * p.y.w)) * C);
        p.red = goff(p.red, g * ISA_STAT_R * 2 * C); p.out_red = goff(p.out_red, g * ISA_STAT_R * 2 * C);
    }
    if (!APPLY) {
        for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
        __syncthreads();
    } else if (p.train) {
        // fold the ISA_STAT_R replicas of the reduce pass once per workgroup, for the channels this workgroup
        // owns (blockIdx.y = channel chunk on wide tensors)
        const int cw = 8 << wk.sh;                                // channels per chunk
        for (int cb = blockIdx.y * cw; cb < C; cb += gridDim.y * cw) {
            for (int i = threadIdx.x; i < 2 * cw; i += 256) {
                const int which = i / cw, c = cb + (i - which * cw);
                if (c >= C) continue;
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < ISA_STAT_R; ++r) s += p.red[r * 2 * C + which * C + c];
                red[which * C + c] = s;
                if (blockIdx.x == 0) {
                    if (which == 1 && p.dgamma) atomicAdd(p.dgamma + c, s);
                    if (which == 0 && p.dbeta) atomicAdd(p.dbeta + c, s);
                }
            }
        }
        __syncthreads();
    }
    const int ppb = 256 >> wk.sh;
    const int psub = threadIdx.x >> wk.sh;
    const unsigned hw = (unsigned)p.y.h * (unsigned)p.y.w;
    for (int cgi = (blockIdx.y << wk.sh) + (threadIdx.x & ((1 << wk.sh) - 1)); cgi < wk.cg; cgi += (gridDim.y << wk.sh)) {
        const int c0 = cgi * 8;
        const int nv = min(8, C - c0);
        float sc[8], sh[8], mu[8], is[8], k0[8], k1[8], s0[8], s1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = min(c0 + j, C - 1);
            sc[j] = p.scale ? p.scale[c] : 1.f;
            sh[j] = p.shift ? p.shift[c] : 0.f;
            mu[j] = p.mean ? p.mean[c] : 0.f;
            is[j] = p.invstd ? p.invstd[c] : 1.f;
            k0[j] = (APPLY && p.train) ? red[c] * p.inv_count : 0.f;
            k1[j] = (APPLY && p.train) ? red[C + c] * p.inv_count : 0.f;
            s0[j] = 0.f; s1[j] = 0.f;
        }
        const long step = (long)gridDim.x * ppb;
        for (long pix = (long)blockIdx.x * ppb + psub; pix < wk.pixels; pix += step) {
            float dt[8], yv[8], out[8];
            load8g<T>(reinterpret_cast<const T*>(p.dt.data) + pix * p.dt.ld + c0, dt, nv);
            load8g<T>(reinterpret_cast<const T*>(p.y.data) + pix * p.y.ld + c0, yv, nv);
            const float* bs = p.bscale ? p.bscale + (long)((unsigned)pix / hw) * C : nullptr;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float z = fmaf(yv[j], sc[j], sh[j]);
                float dz = dt[j] * act_grad_t<ACT>(z, p.act);
                if (bs) dz *= bs[min(c0 + j, C - 1)];
                const float yh = (yv[j] - mu[j]) * is[j];
                if (!APPLY) { s0[j] += dz; s1[j] += dz * yh; }
                else out[j] = sc[j] * (dz - k0[j] - yh * k1[j]);
            }
            if (APPLY) store8g<T>(reinterpret_cast<T*>(p.dy.data) + pix * p.dy.ld + c0, out, nv);
        }
        if (!APPLY) {
            // lanes that differ only above bit `sh` hold the same channel group: fold them inside the wave first, so a
            // wave issues one LDS atomic per channel instead of 64 >> sh (they all hit the same address and serialise)
            const int lane = threadIdx.x & 63;
            if (wk.sh < 6) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    for (int off = 1 << wk.sh; off < 64; off <<= 1) {
                        s0[j] += __shfl_xor(s0[j], off, 64); s1[j] += __shfl_xor(s1[j], off, 64);
                    }
                }
            }
            if (wk.sh >= 6 || (lane >> wk.sh) == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < nv) { atomicAdd(&red[c0 + j], s0[j]); atomicAdd(&red[C + c0 + j], s1[j]); }
            }
        }
    }
    if (!APPLY) {
        __syncthreads();
        float* rep = p.out_red + (blockIdx.x & (ISA_STAT_R - 1)) * 2 * C;
        for (int i = threadIdx.x; i < 2 * C; i += 256)
            if (red[i] != 0.f) atomicAdd(rep + i, red[i]);
    }
}

// out = (pro(x) (+ res) (+ res2)) * oscale[b,c]
// groups (blockIdx.z): pixels / wk.pixels are per group.  bcast: x (and res) hold ONE group of images that every output
// group reads (an identical sub-network evaluated once for all decoder iterations; only the per-image oscale differs).
struct MatParams { View x, res, res2, out; ProDev pro; const float* oscale; long pixels; int cg; int has_res, has_res2; int groups, bcast; FinDev fin; };
template <typename T, int ACT>
__global__ __launch_bounds__(256) void materialize_kernel(MatParams p, Walk wk) {
    const int C = p.x.c;
    // a pending finalize of x's BatchNorm runs here (isa_pro.fin): x's statistic group of this workgroup into LDS
    __shared__ float fin_tab[2 * ISA_FIN_MAX_C];
    const bool fin = bn_fin_inline<256>(p.fin, C, p.bcast ? 1 : p.groups, p.bcast ? 0 : (int)blockIdx.z, fin_tab, ISA_FIN_MAX_C,
                                        threadIdx.x);
    if (p.groups > 1 && blockIdx.z) {
        const long g = blockIdx.z, gp = g * wk.pixels;
        const long gimg = g * (wk.pixels / ((long)p.x.h * p.x.w)) * C;
        p.out.data = reinterpret_cast<T*>(p.out.data) + gp * p.out.ld;
        if (p.has_res2) p.res2.data = reinterpret_cast<T*>(p.res2.data) + gp * p.res2.ld;
        p.oscale = goff(p.oscale, gimg);
        if (!p.bcast) {
            p.x.data = reinterpret_cast<T*>(p.x.data) + gp * p.x.ld;
            if (p.has_res) p.res.data = reinterpret_cast<T*>(p.res.data) + gp * p.res.ld;
            p.pro.scale = goff(p.pro.scale, g * C); p.pro.shift = goff(p.pro.shift, g * C);
            p.pro.bscale = goff(p.pro.bscale, gimg);
        }
    }
    const int ppb = 256 >> wk.sh;
    const int psub = threadIdx.x >> wk.sh;
    const unsigned hw = (unsigned)p.x.h * (unsigned)p.x.w;
    for (int cgi = threadIdx.x & ((1 << wk.sh) - 1); cgi < wk.cg; cgi += (1 << wk.sh)) {
        const int c0 = cgi * 8;
        const int nv = min(8, C - c0);
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = min(c0 + j, C - 1);
            sc[j] = fin ? fin_tab[c] : (p.pro.scale ? p.pro.scale[c] : 1.f);
            sh[j] = fin ? fin_tab[ISA_FIN_MAX_C + c] : (p.pro.shift ? p.pro.shift[c] : 0.f);
        }
        const bool per_image = p.pro.bscale || p.oscale;
        const long step = (long)gridDim.x * ppb;
        for (long pix = (long)blockIdx.x * ppb + psub; pix < wk.pixels; pix += step) {
            float v[8];
            load8g<T>(reinterpret_cast<const T*>(p.x.data) + pix * p.x.ld + c0, v, nv);
            const long bofs = per_image ? (long)((unsigned)pix / hw) * C : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = act_t<ACT>(fmaf(v[j], sc[j], sh[j]), p.pro.act);
            if (p.pro.bscale) {                              // wave-uniform: not a test per element
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= p.pro.bscale[bofs + min(c0 + j, C - 1)];
            }
            if (p.has_res) {
                float rr[8];
                load8g<T>(reinterpret_cast<const T*>(p.res.data) + pix * p.res.ld + c0, rr, nv);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rr[j];
            }
            if (p.has_res2) {
                float rr[8];
                load8g<T>(reinterpret_cast<const T*>(p.res2.data) + pix * p.res2.ld + c0, rr, nv);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rr[j];
            }
            if (p.oscale) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= p.oscale[bofs + min(c0 + j, C - 1)];
            }
            store8g<T>(reinterpret_cast<T*>(p.out.data) + pix * p.out.ld + c0, v, nv);
        }
    }
}

struct AxpyParams { View src, dst; float alpha; int accumulate; long pixels; };
template <typename T>
__global__ __launch_bounds__(256) void axpy_kernel(AxpyParams p) {
    const int C = p.src.c;
    const long items = p.pixels * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C); const long pix = i / C;
        // alpha == 0 is a FILL: never touch src (0 * uninitialised NaN would poison the destination)
        float v = p.alpha == 0.f ? 0.f : p.alpha * st<T>::ld(reinterpret_cast<const T*>(p.src.data) + pix * p.src.ld + c);
        T* d = reinterpret_cast<T*>(p.dst.data) + pix * p.dst.ld + c;
        if (p.accumulate) v += st<T>::ld(d);
        st<T>::stv(d, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void axpy8_kernel(AxpyParams p) {
    const int cg = p.src.c / 8;
    const long items = p.pixels * cg;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cg) * 8; const long pix = i / cg;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p.alpha != 0.f) load8<T>(reinterpret_cast<const T*>(p.src.data) + pix * p.src.ld + c0, v);
        T* d = reinterpret_cast<T*>(p.dst.data) + pix * p.dst.ld + c0;
        if (p.accumulate) {
            float o[8]; load8<T>(d, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(p.alpha, v[j], o[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
        }
        store8<T>(d, v);
    }
}

// 2x2 mean pooling and its backward
struct PoolParams { View x, y; int accumulate; int f; int is_max; };
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void avgpool2_kernel(PoolParams p) {
    // fwd: x big -> y small.  bwd: (x = dy small) -> (y = dx big), dx (+)= dy/4
    const View big = BWD ? p.y : p.x, small = BWD ? p.x : p.y;
    const int cg = small.c / 8;
    const long items = (long)small.n * small.h * small.w * cg;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cg) * 8; long q = i / cg;
        const int xs = (int)(q % small.w); q /= small.w;
        const int ys = (int)(q % small.h); const int b = (int)(q / small.h);
        T* sp = reinterpret_cast<T*>(small.data) + (((long)b * small.h + ys) * small.w + xs) * small.ld + c0;
        T* bp = reinterpret_cast<T*>(big.data) + (((long)b * big.h + 2 * ys) * big.w + 2 * xs) * big.ld + c0;
        if (!BWD) {
            float a[8], acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                load8<T>(bp + ((long)(d >> 1) * big.w + (d & 1)) * big.ld, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += a[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] *= 0.25f;
            store8<T>(sp, acc);
        } else {
            float g[8];
            load8<T>(sp, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] *= 0.25f;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                T* dp = bp + ((long)(d >> 1) * big.w + (d & 1)) * big.ld;
                float o[8];
                if (p.accumulate) {
                    load8<T>(dp, o);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += g[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = g[j];
                }
                store8<T>(dp, o);
            }
        }
    }
}

// f x f pooling (stride f) of narrow maps; scalar per element
template <typename T>
__global__ __launch_bounds__(256) void pool_f_kernel(PoolParams p) {
    const long items = (long)p.y.n * p.y.h * p.y.w * p.y.c;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long)gridDim.x * 256) {
        const int c = (int)(i % p.y.c); long q = i / p.y.c;
        const int xo = (int)(q % p.y.w); q /= p.y.w;
        const int yo = (int)(q % p.y.h); const int b = (int)(q / p.y.h);
        float acc = p.is_max ? -INFINITY : 0.f;
        for (int dy = 0; dy < p.f; ++dy)
            for (int dx = 0; dx < p.f; ++dx) {
                const float v = st<T>::ld(reinterpret_cast<const T*>(p.x.data) +
                    (((long)b * p.x.h + yo * p.f + dy) * p.x.w + xo * p.f + dx) * p.x.ld + c);
                acc = p.is_max ? fmaxf(acc, v) : acc + v;
            }
        if (!p.is_max) acc /= (float)(p.f * p.f);
        st<T>::stv(reinterpret_cast<T*>(p.y.data) + (((long)b * p.y.h + yo) * p.y.w + xo) * p.y.ld + c, acc);
    }
}

// 3x3 mean, stride 1, pad 1, divisor always 9; optional single-channel mask multiply
struct Pool3Params { View x, mask, y; int has_mask; int accumulate; };
template <typename T>
__global__ __launch_bounds__(256) void avgpool3_kernel(Pool3Params p) {
    const long items = (long)p.y.n * p.y.h * p.y.w * p.y.c;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long)gridDim.x * 256) {
        const int c = (int)(i % p.y.c); long q = i / p.y.c;
        const int xo = (int)(q % p.y.w); q /= p.y.w;
        const int yo = (int)(q % p.y.h); const int b = (int)(q / p.y.h);
        float acc = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = yo + dy, xx = xo + dx;
                if (yy >= 0 && yy < p.x.h && xx >= 0 && xx < p.x.w)
                    acc += st<T>::ld(reinterpret_cast<const T*>(p.x.data) + (((long)b * p.x.h + yy) * p.x.w + xx) * p.x.ld + c);
            }
        acc *= (1.f / 9.f);
        const long pix = ((long)b * p.y.h + yo) * p.y.w + xo;
        if (p.has_mask) acc *= st<T>::ld(reinterpret_cast<const T*>(p.mask.data) + pix * p.mask.ld);
        T* o = reinterpret_cast<T*>(p.y.data) + pix * p.y.ld + c;
        if (p.accumulate) acc += st<T>::ld(o);
        st<T>::stv(o, acc);
    }
}

// vector form for C % 8 == 0: an item is (pixel, 8-channel group), nine 16-byte loads, 32-bit index math
template <typename T>
__global__ __launch_bounds__(256) void avgpool3_vec_kernel(Pool3Params p) {
    const unsigned cg = (unsigned)p.y.c / 8, W = (unsigned)p.y.w, H = (unsigned)p.y.h;
    const unsigned items = (unsigned)p.y.n * H * W * cg;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < items; i += gridDim.x * 256u) {
        const unsigned c0 = (i % cg) * 8; unsigned q = i / cg;
        const int xo = (int)(q % W); q /= W;
        const int yo = (int)(q % H); const int b = (int)(q / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = yo + dy, xx = xo + dx;
                if (yy >= 0 && yy < p.x.h && xx >= 0 && xx < p.x.w) {
                    float v[8];
                    load8<T>(reinterpret_cast<const T*>(p.x.data) + (((long)b * p.x.h + yy) * p.x.w + xx) * p.x.ld + c0, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[j];
                }
            }
        const long pix = ((long)b * p.y.h + yo) * p.y.w + xo;
        float m = 1.f / 9.f;
        if (p.has_mask) m *= st<T>::ld(reinterpret_cast<const T*>(p.mask.data) + pix * p.mask.ld);
        T* o = reinterpret_cast<T*>(p.y.data) + pix * p.y.ld + c0;
        float old[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p.accumulate) load8<T>(o, old);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc[j] * m + old[j];
        store8<T>(o, acc);
    }
}

// per-(image,channel) mean of pro(x): grid = (blocks, n)
struct MeanParams { View x; ProDev pro; float* out; float inv_hw; };
template <typename T>
__global__ __launch_bounds__(256) void chan_mean_kernel(MeanParams p) {
    extern __shared__ float red[];            // [C]
    const int C = p.x.c, cg = C / 8, b = blockIdx.y;
    for (int i = threadIdx.x; i < C; i += 256) red[i] = 0.f;
    __syncthreads();
    const long hw = (long)p.x.h * p.x.w, items = hw * cg;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int last_c0 = -1;
    for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < items; item += (long)gridDim.x * 256) {
        const int c0 = (int)(item % cg) * 8; const long pix = item / cg;
        if (c0 != last_c0 && last_c0 >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { atomicAdd(&red[last_c0 + j], s[j]); s[j] = 0.f; }
        }
        last_c0 = c0;
        float v[8];
        load8<T>(reinterpret_cast<const T*>(p.x.data) + ((long)b * hw + pix) * p.x.ld + c0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = v[j];
            if (p.pro.scale) z *= p.pro.scale[c0 + j];
            if (p.pro.shift) z += p.pro.shift[c0 + j];
            s[j] += act_apply(z, p.pro.act);
        }
    }
    if (last_c0 >= 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&red[last_c0 + j], s[j]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256)
        if (red[i] != 0.f) atomicAdd(p.out + (long)b * C + i, red[i] * p.inv_hw);
}

// SE gate: one workgroup per image
__global__ void se_fc_kernel(const float* mean, const float* w1, const float* b1, const float* w2,
                             const float* b2, int c, int hidden, float* hid, float* gate) {
    extern __shared__ float sm[];             // [c + hidden]
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < c; i += blockDim.x) sm[i] = mean[(long)b * c + i];
    __syncthreads();
    for (int j = threadIdx.x; j < hidden; j += blockDim.x) {
        float a = b1[j];
        for (int i = 0; i < c; ++i) a = fmaf(w1[j * c + i], sm[i], a);
        a = fmaxf(a, 0.f);
        sm[c + j] = a;
        if (hid) hid[(long)b * hidden + j] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
        float a = b2[i];
        for (int j = 0; j < hidden; ++j) a = fmaf(w2[i * hidden + j], sm[c + j], a);
        gate[(long)b * c + i] = 1.f / (1.f + expf(-a));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void chan_argmax_kernel(View x, View y) {
    const long pixels = (long)x.n * x.h * x.w;
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < pixels; pix += (long)gridDim.x * 256) {
        const T* px = reinterpret_cast<const T*>(x.data) + pix * x.ld;
        float best = st<T>::ld(px); int bi = 0;
        for (int c = 1; c < x.c; ++c) {
            const float v = st<T>::ld(px + c);
            if (v > best) { best = v; bi = c; }
        }
        st<T>::stv(reinterpret_cast<T*>(y.data) + pix * y.ld, (float)bi);
    }
}

// boundary layout converters: the reference hands over NCHW fp32 tensors (reseg.py:106-110)
template <typename T, bool TO_NHWC>
__global__ __launch_bounds__(256) void layout_kernel(float* nchw, View v, int csrc) {
    // NHWC-major indexing so the strided side is the fp32 NCHW one (read once / written once)
    const long total = (long)v.n * v.h * v.w * v.c;
    const long hw = (long)v.h * v.w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % v.c); const long pix = i / v.c;
        const long b = pix / hw, sp = pix - b * hw;
        T* q = reinterpret_cast<T*>(v.data) + pix * v.ld + c;
        if (TO_NHWC) st<T>::stv(q, c < csrc ? nchw[(b * csrc + c) * hw + sp] : 0.f);
        else if (c < csrc) nchw[(b * csrc + c) * hw + sp] = st<T>::ld(q);
    }
}

// NCHW fp32 -> NHWC rows, a thread per pixel: for a fixed channel consecutive lanes read consecutive fp32 pixels of the
// plane (coalesced), then every lane writes its own pixel row - rows of consecutive lanes are adjacent in memory, so the
// stores of a wave cover one contiguous piece.  The element-wise walk above reads with a stride of h*w floats between
// lanes (every lane its own cache line): 194 us for the 16 x 21 x 256 x 256 network input.  CP = padded channels (<= 32).
template <typename T, int CP>
__global__ __launch_bounds__(256) void nchw_to_nhwc_rows_kernel(const float* nchw, View v, int csrc) {
    // A workgroup converts 1024 consecutive pixels: every thread takes four of them - one 16-byte load per channel plane -
    // and parks its four rows in LDS; the 1024 rows are one contiguous piece of the NHWC tensor (ld == CP), which the
    // workgroup then writes out linearly, 16 bytes per lane.  (Rows written straight from the registers were 16-byte
    // pieces 4 * ld elements apart per store instruction: 60 us for the 16 x 21 x 256 x 256 input, as slow as the scalar walk.)
    extern __shared__ __attribute__((aligned(16))) char lrows[];          // [1024][CP] of T
    T* rows = reinterpret_cast<T*>(lrows);
    const long hw = (long)v.h * v.w, total = (long)v.n * hw;
    for (long base = (long)blockIdx.x * 1024; base < total; base += (long)gridDim.x * 1024) {
        const long pix = base + threadIdx.x * 4;
        if (pix < total) {
            const long b = pix / hw, sp = pix - b * hw;                     // hw % 4 == 0: the four pixels share the image
            const float* s = nchw + b * csrc * hw + sp;
            f32x4 val[CP];
#pragma unroll
            for (int c = 0; c < CP; ++c) val[c] = c < csrc ? *reinterpret_cast<const f32x4*>(s + (long)c * hw) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int c0 = 0; c0 < CP; c0 += 8) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = val[c0 + j][k];
                    store8<T>(rows + (threadIdx.x * 4 + k) * CP + c0, o);
                }
            }
        }
        __syncthreads();
        const long npix = min((long)1024, total - base);
        constexpr int VPR = CP * (int)sizeof(T) / 16;                       // 16-byte vectors per row
        const long nvec = npix * VPR;
        char* dst = reinterpret_cast<char*>(reinterpret_cast<T*>(v.data) + base * v.ld);
        for (long i = threadIdx.x; i < nvec; i += 256)
            *reinterpret_cast<f32x4*>(dst + i * 16) = *reinterpret_cast<const f32x4*>(lrows + i * 16);
        __syncthreads();
    }
}

static inline bool same_shape(const isa_tensor* a, const isa_tensor* b) {
    return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c && a->dtype == b->dtype;
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
    do { if ((dtype) == ISA_BF16) { CALL_BF16; } else { CALL_F32; } } while (0)

extern "C" int isa_pack_weights(const isa_pack_entry* table_dev, int32_t n_entries,
                                const int32_t* kmap_dev, const float* src_base, void* dst_base,
                                int32_t dst_dtype, void* stream) {
    if (!table_dev || n_entries <= 0 || !src_base || !dst_base) return ISA_EINVAL;
    dim3 grid(32, n_entries);
    DISPATCH_T(dst_dtype,
        hipLaunchKernelGGL(pack_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), table_dev, kmap_dev, src_base, (bf16_t*)dst_base),
        hipLaunchKernelGGL(pack_kernel<float>, grid, dim3(256), 0, as_stream(stream), table_dev, kmap_dev, src_base, (float*)dst_base));
    return launch_status();
}

extern "C" int isa_bn_finalize(const float* stats, float count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, float* scale, float* shift,
                               float* mean, float* invstd, int32_t c, int32_t groups, int32_t repeat, void* stream) {
    if (c <= 0 || !scale || !shift || (!stats && (!running_mean || !running_var))) return ISA_EINVAL;
    if (groups < 1) groups = 1;
    if (repeat < 1) repeat = 1;
    if (!stats && groups != 1) return ISA_EINVAL;
    const FinDev f{stats, gamma, beta, running_mean, running_var, scale, shift, mean, invstd, count, momentum, eps, repeat};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(c, 256)), dim3(256), 0, as_stream(stream), f, c, groups);
    return launch_status();
}

int fin_standalone(const isa_pro* p, int c, int groups, hipStream_t s) {
    if (!p || !p->fin) return ISA_OK;
    if (!fin_valid(p) || c <= 0) return ISA_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(c, 256)), dim3(256), 0, s, make_fin(p), c, groups < 1 ? 1 : groups);
    return launch_status();
}

extern "C" int isa_bn_running_update(const isa_bn_upd* upd, int32_t n, float momentum, void* stream) {
    if (n < 0 || (n > 0 && !upd)) return ISA_EINVAL;
    for (int i = 0; i < n; ++i)
        if (!upd[i].stats || !upd[i].running_mean || !upd[i].running_var || upd[i].c <= 0 || !(upd[i].count > 0)) return ISA_EINVAL;
    for (int i0 = 0; i0 < n; i0 += BN_UPD_CHUNK) {
        BnUpdChunk ch{};
        const int m = n - i0 < BN_UPD_CHUNK ? n - i0 : BN_UPD_CHUNK;
        for (int i = 0; i < m; ++i) ch.d[i] = upd[i0 + i];
        hipLaunchKernelGGL(bn_running_update_kernel, dim3(m), dim3(256), 0, as_stream(stream), ch, momentum);
        if (launch_status() != ISA_OK) return ISA_ELAUNCH;
    }
    return ISA_OK;
}

static int bn_bwd_common(const isa_tensor* dt, const isa_tensor* y, const isa_tensor* dy,
                         BnBwdParams& p, bool apply, void* stream) {
    if (!tensor_ok(dt, 8) || !tensor_ok(y, 8) || !same_shape(dt, y)) return ISA_EINVAL;
    p.dt = mkview(dt); p.y = mkview(y);
    if (apply) { if (!tensor_ok(dy, 8) || !same_shape(dy, y)) return ISA_EINVAL; p.dy = mkview(dy); }
    const int G = tensor_groups(y);
    if (y->n % G) return ISA_EINVAL;
    p.groups = G;
    p.pixels = (long)(y->n / G) * y->h * y->w; p.cg = (y->c + 7) / 8;     // per group
    if (p.pixels * G >= (1L << 32)) return ISA_EINVAL;
    // wide tensors (> 128 channels): blockIdx.y owns a 32-channel chunk, so a workgroup's O(C) prologue /
    // epilogue (replica fold, per-channel atomics) covers 32 channels instead of C, and the pixel dimension can
    // be split across enough workgroups to fill the chip (16x16x16 px x 1024 ch used to run as 128 workgroups
    // of 16 serial iterations each)
    Walk wk = mkwalk(y->c, p.pixels);
    int gy = 1, grid;
    if (wk.cg > 16) {
        // 32-channel chunks x 64 pixel rows per trip; the reduce pass keeps <= 128 pixel splits so that its
        // splits x 2C epilogue atomics stay small next to the data, the apply pass splits further
        wk.sh = 2; gy = (wk.cg + 3) / 4;
        const long trips = cdiv(p.pixels, 64);
        grid = (int)(apply ? (trips + 1) / 2 : (trips + 3) / 4);
        const int cap = apply ? 512 : 128;
        if (grid > cap) grid = cap;
        if (grid < 1) grid = 1;
    } else {
        // 16 trips per workgroup on the big tensors (amortises the O(C) epilogue); on the small ones fewer trips
        // so that at least ~512 workgroups exist (64x64x16 px x 64 ch ran as 128 workgroups of 16 serial trips)
        const long per = 256 >> wk.sh;
        int iters = 16;
        if (cdiv(p.pixels, per * 16) < 512) { iters = (int)(p.pixels / (per * 512)); if (iters < 2) iters = 2; }
        grid = walk_grid(wk, iters);
        if (!apply && grid > 1024) grid = 1024;     // every block ends with 2C global atomics (8 replicas)
    }
    const size_t lds = 2 * (size_t)y->c * 4;
#define BN_BWD_LAUNCH(AP, ACTV) \
    DISPATCH_T(y->dtype, \
        hipLaunchKernelGGL((bn_bwd_kernel<bf16_t, AP, ACTV>), dim3(grid, gy, G), dim3(256), lds, as_stream(stream), p, wk), \
        hipLaunchKernelGGL((bn_bwd_kernel<float, AP, ACTV>), dim3(grid, gy, G), dim3(256), lds, as_stream(stream), p, wk))
    if (apply) {
        if (p.act == ISA_ACT_RELU6) BN_BWD_LAUNCH(true, ISA_ACT_RELU6);
        else if (p.act == ISA_ACT_NONE) BN_BWD_LAUNCH(true, ISA_ACT_NONE);
        else BN_BWD_LAUNCH(true, ACT_RT);
    } else {
        if (p.act == ISA_ACT_RELU6) BN_BWD_LAUNCH(false, ISA_ACT_RELU6);
        else if (p.act == ISA_ACT_NONE) BN_BWD_LAUNCH(false, ISA_ACT_NONE);
        else BN_BWD_LAUNCH(false, ACT_RT);
    }
#undef BN_BWD_LAUNCH
    return launch_status();
}

extern "C" int isa_bn_bwd_reduce(const isa_tensor* dt, const isa_tensor* y, const float* scale,
                                 const float* shift, const float* mean, const float* invstd,
                                 int32_t act, const float* bscale, float* red, void* stream) {
    if (!red) return ISA_EINVAL;
    BnBwdParams p{};
    p.scale = scale; p.shift = shift; p.mean = mean; p.invstd = invstd; p.bscale = bscale;
    p.act = act; p.out_red = red; p.train = 1;
    return bn_bwd_common(dt, y, nullptr, p, false, stream);
}

extern "C" int isa_bn_bwd_apply(const isa_tensor* dt, const isa_tensor* y, const float* scale,
                                const float* shift, const float* mean, const float* invstd,
                                int32_t act, const float* bscale, const float* gamma,
                                const float* red, float count, int32_t train,
                                const isa_tensor* dy, float* dgamma, float* dbeta, void* stream) {
    if (train && !red) return ISA_EINVAL;
    BnBwdParams p{};
    p.scale = scale; p.shift = shift; p.mean = mean; p.invstd = invstd; p.bscale = bscale;
    p.gamma = gamma; p.red = red; p.inv_count = count > 0 ? 1.f / count : 0.f;
    p.act = act; p.train = train; p.dgamma = dgamma; p.dbeta = dbeta;
    return bn_bwd_common(dt, y, dy, p, true, stream);
}

extern "C" int isa_affine_act_res(const isa_tensor* x, const isa_pro* pro, const isa_tensor* res,
                                  const isa_tensor* res2, const float* oscale,
                                  const isa_tensor* out, void* stream) {
    if (!tensor_ok(x, 8) || !tensor_ok(out, 8)) return ISA_EINVAL;
    // broadcast form: x (and res) hold n images, out G*n - every output group is (pro(x) + res (+ res2[g])) * oscale[g]
    const int G = tensor_groups(out);
    const bool bcast = G > 1 && out->n == G * x->n && tensor_groups(x) == 1;
    if (bcast) {
        if (x->h != out->h || x->w != out->w || x->c != out->c || x->dtype != out->dtype) return ISA_EINVAL;
    } else if (!same_shape(x, out)) return ISA_EINVAL;
    if (out->n % G) return ISA_EINVAL;
    if (res && (!tensor_ok(res, 8) || !same_shape(res, x))) return ISA_EINVAL;
    if (res2 && (!tensor_ok(res2, 8) || !same_shape(res2, out))) return ISA_EINVAL;
    MatParams p{};
    p.x = mkview(x); p.out = mkview(out); p.pro = make_pro(pro); p.has_res = res != nullptr;
    p.has_res2 = res2 != nullptr; p.oscale = oscale;
    if (res) p.res = mkview(res);
    if (res2) p.res2 = mkview(res2);
    p.groups = G; p.bcast = bcast;
    if (pro && pro->fin) {
        if (!fin_valid(pro) || (!bcast && tensor_groups(x) != G)) return ISA_EINVAL;
        if (x->c <= ISA_FIN_MAX_C) p.fin = make_fin(pro);
        else if (int rc = fin_standalone(pro, x->c, tensor_groups(x), as_stream(stream))) return rc;
    }
    p.pixels = (long)(out->n / G) * x->h * x->w; p.cg = (x->c + 7) / 8;       // per group
    if (p.pixels * G >= (1L << 32)) return ISA_EINVAL;
    const Walk wk = mkwalk(x->c, p.pixels);
    const dim3 grid(walk_grid(wk), 1, G);
    if (p.pro.act == ISA_ACT_NONE)
        DISPATCH_T(x->dtype,
            hipLaunchKernelGGL((materialize_kernel<bf16_t, ISA_ACT_NONE>), grid, dim3(256), 0, as_stream(stream), p, wk),
            hipLaunchKernelGGL((materialize_kernel<float, ISA_ACT_NONE>), grid, dim3(256), 0, as_stream(stream), p, wk));
    else
        DISPATCH_T(x->dtype,
            hipLaunchKernelGGL((materialize_kernel<bf16_t, ACT_RT>), grid, dim3(256), 0, as_stream(stream), p, wk),
            hipLaunchKernelGGL((materialize_kernel<float, ACT_RT>), grid, dim3(256), 0, as_stream(stream), p, wk));
    return launch_status();
}

extern "C" int isa_axpy(const isa_tensor* src, const isa_tensor* dst, float alpha,
                        int32_t accumulate, void* stream) {
    if (!tensor_ok(src, 1) || !tensor_ok(dst, 1) || !same_shape(src, dst)) return ISA_EINVAL;
    AxpyParams p{mkview(src), mkview(dst), alpha, accumulate, (long)src->n * src->h * src->w};
    const bool vec = tensor_ok(src, 8) && tensor_ok(dst, 8) && src->c % 8 == 0;
    const int grid = grid_cap(cdiv(p.pixels * (vec ? src->c / 8 : src->c), 256));
    if (vec)
        DISPATCH_T(src->dtype,
            hipLaunchKernelGGL(axpy8_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), p),
            hipLaunchKernelGGL(axpy8_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), p));
    else
        DISPATCH_T(src->dtype,
            hipLaunchKernelGGL(axpy_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), p),
            hipLaunchKernelGGL(axpy_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), p));
    return launch_status();
}

extern "C" int isa_avgpool2(const isa_tensor* x, const isa_tensor* y, void* stream) {
    if (!tensor_ok(x, 8) || !tensor_ok(y, 8) || x->dtype != y->dtype || x->c != y->c || x->c % 8 ||
        x->n != y->n || x->h != 2 * y->h || x->w != 2 * y->w) return ISA_EINVAL;
    PoolParams p{mkview(x), mkview(y), 0, 2, 0};
    const int grid = grid_cap(cdiv((long)y->n * y->h * y->w * (y->c / 8), 256));
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL((avgpool2_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, as_stream(stream), p),
        hipLaunchKernelGGL((avgpool2_kernel<float, false>), dim3(grid), dim3(256), 0, as_stream(stream), p));
    return launch_status();
}

extern "C" int isa_avgpool2_bwd(const isa_tensor* dy, const isa_tensor* dx, int32_t accumulate,
                                void* stream) {
    if (!tensor_ok(dy, 8) || !tensor_ok(dx, 8) || dy->dtype != dx->dtype || dy->c != dx->c ||
        dy->c % 8 || dy->n != dx->n || dx->h != 2 * dy->h || dx->w != 2 * dy->w) return ISA_EINVAL;
    PoolParams p{mkview(dy), mkview(dx), accumulate, 2, 0};
    const int grid = grid_cap(cdiv((long)dy->n * dy->h * dy->w * (dy->c / 8), 256));
    DISPATCH_T(dy->dtype,
        hipLaunchKernelGGL((avgpool2_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, as_stream(stream), p),
        hipLaunchKernelGGL((avgpool2_kernel<float, true>), dim3(grid), dim3(256), 0, as_stream(stream), p));
    return launch_status();
}

extern "C" int isa_pool_f(const isa_tensor* x, const isa_tensor* y, int32_t f, int32_t is_max,
                          void* stream) {
    if (!tensor_ok(x, 1) || !tensor_ok(y, 1) || x->dtype != y->dtype || x->c != y->c || f < 1 ||
        x->n != y->n || x->h != f * y->h || x->w != f * y->w) return ISA_EINVAL;
    PoolParams p{mkview(x), mkview(y), 0, f, is_max};
    const int grid = grid_cap(cdiv((long)y->n * y->h * y->w * y->c, 256));
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(pool_f_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), p),
        hipLaunchKernelGGL(pool_f_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), p));
    return launch_status();
}

extern "C" int isa_avgpool3(const isa_tensor* x, const isa_tensor* mask, const isa_tensor* y,
                            int32_t accumulate, void* stream) {
    if (!tensor_ok(x, 1) || !tensor_ok(y, 1) || !same_shape(x, y)) return ISA_EINVAL;
    if (mask && (!tensor_ok(mask, 1) || mask->n != x->n || mask->h != x->h || mask->w != x->w ||
                 mask->dtype != x->dtype)) return ISA_EINVAL;
    Pool3Params p{mkview(x), mask ? mkview(mask) : View{}, mkview(y), mask != nullptr, accumulate};
    const long vec_items = (long)y->n * y->h * y->w * (y->c / 8);
    const bool aligned = (reinterpret_cast<uintptr_t>(x->data) % 16 == 0) && (reinterpret_cast<uintptr_t>(y->data) % 16 == 0);
    if (y->c % 8 == 0 && x->ld % 8 == 0 && y->ld % 8 == 0 && aligned && vec_items < (1L << 31)) {
        const int gridv = grid_cap(cdiv(vec_items, 256), 256 * 16);
        DISPATCH_T(x->dtype,
            hipLaunchKernelGGL(avgpool3_vec_kernel<bf16_t>, dim3(gridv), dim3(256), 0, as_stream(stream), p),
            hipLaunchKernelGGL(avgpool3_vec_kernel<float>, dim3(gridv), dim3(256), 0, as_stream(stream), p));
        return launch_status();
    }
    const int grid = grid_cap(cdiv((long)y->n * y->h * y->w * y->c, 256));
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(avgpool3_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), p),
        hipLaunchKernelGGL(avgpool3_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), p));
    return launch_status();
}

extern "C" int isa_chan_mean(const isa_tensor* x, const isa_pro* pro, float* out, void* stream) {
    if (pro && pro->fin) { if (int rc = fin_standalone(pro, x->c, tensor_groups(x), as_stream(stream))) return rc; }   // no in-kernel form here
    if (!tensor_ok(x, 8) || x->c % 8 || !out) return ISA_EINVAL;
    MeanParams p{mkview(x), make_pro(pro), out, 1.f / ((float)x->h * x->w)};
    const long items = (long)x->h * x->w * (x->c / 8);
    dim3 grid(grid_cap(cdiv(items, 256), 128), x->n);
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(chan_mean_kernel<bf16_t>, grid, dim3(256), x->c * 4, as_stream(stream), p),
        hipLaunchKernelGGL(chan_mean_kernel<float>, grid, dim3(256), x->c * 4, as_stream(stream), p));
    return launch_status();
}

extern "C" int isa_se_fc(const float* mean, const float* w1, const float* b1, const float* w2,
                         const float* b2, int32_t n, int32_t c, int32_t hidden, float* hid,
                         float* gate, void* stream) {
    if (!mean || !w1 || !b1 || !w2 || !b2 || !gate || n <= 0 || c <= 0 || hidden <= 0) return ISA_EINVAL;
    hipLaunchKernelGGL(se_fc_kernel, dim3(n), dim3(64), (c + hidden) * 4, as_stream(stream), mean, w1, b1,
                       w2, b2, c, hidden, hid, gate);
    return launch_status();
}

extern "C" int isa_chan_argmax(const isa_tensor* x, const isa_tensor* y, void* stream) {
    if (!tensor_ok(x, 1) || !tensor_ok(y, 1) || x->dtype != y->dtype || x->n != y->n ||
        x->h != y->h || x->w != y->w) return ISA_EINVAL;
    const int grid = grid_cap(cdiv((long)x->n * x->h * x->w, 256));
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(chan_argmax_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(x), mkview(y)),
        hipLaunchKernelGGL(chan_argmax_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(x), mkview(y)));
    return launch_status();
}

extern "C" int isa_nchw_to_nhwc(const float* src, int32_t csrc, const isa_tensor* dst, void* stream) {
    if (!src || !tensor_ok(dst, 1) || csrc <= 0 || csrc > dst->c) return ISA_EINVAL;
    const int cp = (dst->c + 7) / 8 * 8;
    if (tensor_ok(dst, 8) && dst->c <= 32 && dst->ld == cp && ((long)dst->h * dst->w) % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(src) % 16) == 0) {
        // the network input (21 -> 24) and other narrow maps whose padded rows are contiguous: row form (pad channels = 0)
        const int gridp = grid_cap(cdiv((long)dst->n * dst->h * dst->w, 1024), 256 * 8);
        const size_t esz = dst->dtype == ISA_BF16 ? 2 : 4;
#define ROWS(CPV) DISPATCH_T(dst->dtype, \
            hipLaunchKernelGGL((nchw_to_nhwc_rows_kernel<bf16_t, CPV>), dim3(gridp), dim3(256), 1024 * CPV * esz, as_stream(stream), src, mkview(dst), csrc), \
            hipLaunchKernelGGL((nchw_to_nhwc_rows_kernel<float, CPV>), dim3(gridp), dim3(256), 1024 * CPV * esz, as_stream(stream), src, mkview(dst), csrc))
        if (cp <= 8) ROWS(8); else if (cp <= 16) ROWS(16); else if (cp <= 24) ROWS(24); else ROWS(32);
#undef ROWS
        return launch_status();
    }
    const int grid = grid_cap(cdiv((long)dst->n * dst->h * dst->w * dst->c, 256));
    DISPATCH_T(dst->dtype,
        hipLaunchKernelGGL((layout_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, as_stream(stream), (float*)src, mkview(dst), csrc),
        hipLaunchKernelGGL((layout_kernel<float, true>), dim3(grid), dim3(256), 0, as_stream(stream), (float*)src, mkview(dst), csrc));
    return launch_status();
}

extern "C" int isa_nhwc_to_nchw(const isa_tensor* src, float* dst, void* stream) {
    if (!dst || !tensor_ok(src, 1)) return ISA_EINVAL;
    const int grid = grid_cap(cdiv((long)src->n * src->h * src->w * src->c, 256));
    DISPATCH_T(src->dtype,
        hipLaunchKernelGGL((layout_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, as_stream(stream), dst, mkview(src), src->c),
        hipLaunchKernelGGL((layout_kernel<float, false>), dim3(grid), dim3(256), 0, as_stream(stream), dst, mkview(src), src->c));
    return launch_status();
}
