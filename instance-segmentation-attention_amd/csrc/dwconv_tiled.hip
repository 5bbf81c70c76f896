// Depthwise 3x3 v2: LDS-staged halo tiles.  (v1 in dwconv.hip walks row strips with a register window; PMC
// showed it latency-bound: 62 % SQ_WAIT_ANY at 2 waves/SIMD with 3 dependent loads per column.)
//   * a workgroup owns an 8 x 32 output tile x 32 channels.  The (8+2) x (32+2) input halo is staged ONCE
//     in LDS as fp32 after the lazy BN/ReLU6 prologue: every lane issues its 5-6 independent 16-byte
//     loads back to back, so ~24 KB per workgroup is in flight instead of 3 loads per wave.
//   * compute: lane = (8-channel group, 4 consecutive x): 18 row vectors from LDS feed 36 FMA-vectors for
//     4 outputs (horizontal register reuse), pixel stride padded to 36 floats so the four x-groups of a
//     16-lane ds_read_b128 group land on disjoint banks.
//   * forward/dgrad: bias, next-BN statistics (16-lane shuffle tree, then 4 LDS atomics per wave, then one
//     global atomic per channel into the replicated buffer), optional read-modify-write accumulate.
//   * wgrad: persistent over tiles of one channel block; 9x8 products per lane accumulated in registers
//     across tiles, reduced once (shuffle tree + LDS) into a per-workgroup slab (no global atomics).
#include "common.hpp"
#include <type_traits>

namespace {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's GLOBAL stores (vmcnt(0)
// ahead of s_barrier): in the persistent tile loops below that exposed the latency of the output stores once per tile.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int TH = 8, TW = 32, CB = 32, PS = 36;         // tile rows/cols, channel block, pixel stride (floats)
constexpr int HALO = (TH + 2) * (TW + 2);

struct Dw2Params {
    const void* x; const void* w; const float* bias; void* y; const void* dy;
    int n, h, w_, c, ldx, ldy, ldd, wld;
    ProDev pro;
    float* stats; int accumulate;
    int tiles_x, tiles_y; long ntiles;
    float* ws; int csrc;
    int G;                       // statistic groups: n and ntiles are per group (common.hpp)
    FinDev fin;                  // pending BatchNorm finalize of the lazy input (forward only), or stats == NULL
};

// 8 storage elements kept packed in registers (4 VGPRs for bf16) until they are consumed
template <typename T> struct raw8;
template <> struct raw8<bf16_t> {
    bf16x8 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ float get(int j) const { return (float)v[j]; }
    __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
    __device__ __forceinline__ void zero() { v = bf16x8{}; }
};
template <> struct raw8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
    __device__ __forceinline__ float get(int j) const { return j < 4 ? a[j] : b[j - 4]; }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b; }
    __device__ __forceinline__ void zero() { a = f32x4{}; b = f32x4{}; }
};

// per-lane prologue constants: a lane stages the same 8-channel group on every iteration (i += 256 keeps i & 3)
struct ProRegs { float sc[8], sh[8], bs[8]; };

template <bool HAS_PRO>
__device__ __forceinline__ void load_pro(const Dw2Params& p, ProRegs& r, int c0, int b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = min(c0 + j, p.c - 1);
        r.sc[j] = (HAS_PRO && p.pro.scale) ? p.pro.scale[c] : 1.f;
        r.sh[j] = (HAS_PRO && p.pro.shift) ? p.pro.shift[c] : 0.f;
        r.bs[j] = (HAS_PRO && p.pro.bscale) ? p.pro.bscale[(long)b * p.c + c] : 1.f;
    }
}

template <typename T, bool HAS_PRO, int ACT>
__device__ __forceinline__ void stage_tile(const Dw2Params& p, const ProRegs& r, float* tile, int b, int ty, int tx, int c_base) {
    const T* xin = reinterpret_cast<const T*>(p.x);
    const int cg = threadIdx.x & 3;
    const int c0 = c_base + cg * 8;
    const bool cok = c0 < p.c;
    constexpr int NIT = (HALO * 4 + 255) / 256;
    float v[NIT][8];
    bool ok[NIT];
    // all global loads first (independent, in flight together), then prologue + LDS stores
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pix = (threadIdx.x + it * 256) >> 2;
        const int rr = pix / (TW + 2), cc = pix - rr * (TW + 2);
        const int gy = ty * TH + rr - 1, gx = tx * TW + cc - 1;
        ok[it] = pix < HALO && cok && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_;
        if (ok[it]) load8<T>(xin + (((long)b * p.h + gy) * p.w_ + gx) * p.ldx + c0, v[it]);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pix = (threadIdx.x + it * 256) >> 2;
        if (pix >= HALO) continue;
        f32x4 a, bb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = 0.f;
            if (ok[it]) {
                z = v[it][j];
                if constexpr (HAS_PRO) {
                    z = act_t<ACT>(fmaf(z, r.sc[j], r.sh[j]), p.pro.act);
                    if (p.pro.bscale) z *= r.bs[j];
                }
            }
            if (j < 4) a[j] = z; else bb[j - 4] = z;
        }
        *reinterpret_cast<f32x4*>(tile + pix * PS + cg * 8) = a;
        *reinterpret_cast<f32x4*>(tile + pix * PS + cg * 8 + 4) = bb;
    }
}

// acc[j] += x[j] * w[j] for 8 channels as four v_pk_fma_f32 (packed fp32: two FMAs per lane per issue)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fma8(const float (&x)[8], const float (&w)[8], float (&acc)[8]) {
#ifdef ISA_DW_SCALAR_FMA
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(x[j], w[j], acc[j]);
#else
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 r = __builtin_elementwise_fma(f32x2{x[j], x[j + 1]}, f32x2{w[j], w[j + 1]}, f32x2{acc[j], acc[j + 1]});
        acc[j] = r[0]; acc[j + 1] = r[1];
    }
#endif
}

__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}

// XCD-aware tile walk for the persistent kernels: workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD has its
// own L2.  Adjacent tiles share halo rows / columns, so each XCD gets ONE contiguous eighth of the tile sequence and
// its workgroups stride inside it: the halo overlap is re-read from that XCD's L2 instead of from seven other ones.
struct TileRange { long t0, end, step; };
// (bx, nbx): the workgroup's index and the workgroup count inside its statistic group (GroupSel); with nbx % 8 == 0 a
// group starts on XCD 0, so bx & 7 is still the XCD
__device__ __forceinline__ TileRange tile_range(long ntiles, int bx, int nbx) {
    if ((nbx & 7) == 0 && ntiles >= 64) {
        const int xcd = bx & 7, j = bx >> 3;
        const long chunk = (ntiles + 7) / 8;
        const long end = min(ntiles, (xcd + 1) * chunk);
        return TileRange{xcd * chunk + j, end, (long)(nbx >> 3)};
    }
    return TileRange{(long)bx, ntiles, (long)nbx};
}

// Tile coordinates kept incrementally: the tile sequence of a workgroup is t0, t0 + step, ... and a 64-bit
// `t % tiles_x`, `t / tiles_x % tiles_y` pair per tile per role cost more scalar instructions than the tile's loads
// (ablation: with every load, LDS access, FMA and store removed the forward kernel still took 39 of its 93 us - the
// per-tile bookkeeping; one wave per role and SIMD issues an instruction every >= 4 cycles).  All fields are
// wave-uniform (SGPRs).
struct TileIter {
    long t, end, step;
    int tx, ty, b, sx, sy, sb, ntx, nty;
    __device__ __forceinline__ void init(const TileRange& r, int tiles_x, int tiles_y) {
        t = r.t0; end = r.end; step = r.step; ntx = tiles_x; nty = tiles_y;
        long q = t / tiles_x; tx = (int)(t - q * tiles_x); b = (int)(q / tiles_y); ty = (int)(q - (long)b * tiles_y);
        long qs = step / tiles_x; sx = (int)(step - qs * tiles_x); sb = (int)(qs / tiles_y); sy = (int)(qs - (long)sb * tiles_y);
    }
    __device__ __forceinline__ bool valid() const { return t < end; }
    __device__ __forceinline__ void next() {
        t += step;
        tx += sx; const int c = tx >= ntx ? 1 : 0; tx -= c * ntx;
        ty += sy + c; const int c2 = ty >= nty ? 1 : 0; ty -= c2 * nty;
        b += sb + c2;
    }
};

// Persistent over tiles of one channel block.  DB (bf16): 512 threads, waves 4-7 stage the next tile into the other
// LDS buffer while waves 0-3 run the stencil on the current one (the same role split as dw_bn_bwd_kernel below);
// weights, prologue constants and the statistic partial sums live across tiles and are flushed once.
template <typename T, bool HAS_PRO, int ACT, bool DB>
__global__ __launch_bounds__(DB ? 768 : 256) __attribute__((amdgpu_waves_per_eu(2, 3))) void dw2_fwd_kernel(Dw2Params p) {
    // DB: waves 0-3 compute, waves 4-11 stage (three waves per SIMD: the staging arithmetic of a tile is spread over
    // twice the lanes and the SIMD has one more wave to issue from while the others wait)
    constexpr int NTHR = DB ? 768 : 256;
    constexpr int LTHR = DB ? 512 : 256;              // threads that stage a tile
    constexpr int NBUF = DB ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* tile_base = sm;                          // [NBUF][HALO][PS]
    float* wts = sm + NBUF * HALO * PS;             // [9][CB]
    float* red = wts + 9 * CB;                      // [2*CB]
    float* fin_tab = red + 2 * CB;                  // [2][CB] scale / shift of this channel block when the finalize runs here
    const int tid = threadIdx.x;
    const bool loader = DB && tid >= 256;
    const int ltid = loader ? tid - 256 : tid;        // index inside the role group
    const int c_base = blockIdx.y * CB;
    const GroupSel gs = group_sel(p.G);
    if (gs.g) {                                       // this workgroup's statistic group: its n images, its constants
        const long img = (long)gs.g * p.n, pix = img * p.h * p.w_;
        p.x = reinterpret_cast<const T*>(p.x) + pix * p.ldx;
        p.y = reinterpret_cast<T*>(p.y) + pix * p.ldy;
        p.pro.scale = goff(p.pro.scale, (long)gs.g * p.c); p.pro.shift = goff(p.pro.shift, (long)gs.g * p.c);
        p.pro.bscale = goff(p.pro.bscale, img * p.c);
        p.stats = goff(p.stats, (long)gs.g * ISA_STAT_R * 2 * p.c);
    }
    const T* wp = reinterpret_cast<const T*>(p.w);
    for (int i = tid; i < 9 * CB; i += NTHR) {
        const int tp = i / CB, cc = i - tp * CB;
        wts[i] = (c_base + cc < p.c) ? st<T>::ld(wp + (long)tp * p.wld + c_base + cc) : 0.f;
    }
    if (tid < 2 * CB) red[tid] = 0.f;
    const int cg = ltid & 3, g = ltid >> 2, row = g >> 3, x0 = (g & 7) * 4;
    const int c0 = c_base + cg * 8;
    const bool cok = c0 < p.c;
    const T* xin = reinterpret_cast<const T*>(p.x);
    constexpr int NIT = (HALO * 4 + LTHR - 1) / LTHR;

    // per-channel prologue constants: loaded ONCE per lane (they were re-read from global memory at the top of every
    // tile: a dependent round trip ahead of the tile's own loads); only the per-image scale changes with the tile
    // a pending finalize of the input's BatchNorm runs here, for this workgroup's 32 channels; the last workgroup of
    // each channel block writes the arrays the backward pass reads
    const bool fin = HAS_PRO && bn_fin_inline<NTHR>(p.fin, p.c, p.G, gs.g, fin_tab, CB, tid, c_base, CB,
                                                    blockIdx.x == gridDim.x - 1 && blockIdx.z == 0);
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = min(c0 + j, p.c - 1);
        sc[j] = fin ? fin_tab[c - c_base] : ((HAS_PRO && p.pro.scale) ? p.pro.scale[c] : 1.f);
        sh[j] = fin ? fin_tab[CB + c - c_base] : ((HAS_PRO && p.pro.shift) ? p.pro.shift[c] : 0.f);
    }
    // tile-invariant part of the staging addresses: halo pixel (rr, cc) of slot `it` and its element offset from the
    // tile's halo origin; per tile only a scalar base pointer and (on border tiles) four scalar bounds remain
    int hrc[NIT], hoff[NIT];
    bool hok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pix = (ltid + it * LTHR) >> 2;
        const int rr = pix / (TW + 2), cc = pix - rr * (TW + 2);
        hrc[it] = (rr << 16) | cc;
        hoff[it] = (rr * p.w_ + cc) * p.ldx + (cok ? c0 : c_base);
        hok[it] = pix < HALO && cok;
    }
    const int ooff = (row * p.w_ + x0) * p.ldy + c0;           // output element offset from the tile's origin

    auto stage = [&](int b, int ty, int tx, float* tile) {
        float bs[8];
        if (HAS_PRO && p.pro.bscale) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bs[j] = p.pro.bscale[(long)b * p.c + min(c0 + j, p.c - 1)];
        }
        const T* base = xin + (((long)b * p.h + ty * TH - 1) * p.w_ + tx * TW - 1) * p.ldx;   // halo origin (may lie outside)
        // scalar bounds of the in-image part of the halo, in halo coordinates
        const int rlo = 1 - ty * TH, rhi = p.h + 1 - ty * TH, clo = 1 - tx * TW, chi = p.w_ + 1 - tx * TW;
        const bool interior = rlo <= 0 && rhi >= TH + 2 && clo <= 0 && chi >= TW + 2;
        raw8<T> v[NIT]; bool ok[NIT];
        if (interior) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) { ok[it] = hok[it]; if (ok[it]) v[it].load(base + hoff[it]); }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int rr = hrc[it] >> 16, cc = hrc[it] & 0xffff;
                ok[it] = hok[it] && rr >= rlo && rr < rhi && cc >= clo && cc < chi;
                if (ok[it]) v[it].load(base + hoff[it]);
            }
        }
        const bool has_bs = HAS_PRO && p.pro.bscale;             // wave-uniform: hoisted out of the element loops
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (ltid + it * LTHR) >> 2;
            if (pix >= HALO) continue;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
            if (ok[it]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float z = v[it].get(j);
                    if constexpr (HAS_PRO) z = act_t<ACT>(fmaf(z, sc[j], sh[j]), p.pro.act);
                    o[j] = z;
                }
                if (has_bs) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] *= bs[j];
                }
            }
            store8<float>(tile + pix * PS + cg * 8, o);
        }
    };

    auto compute = [&](int b, int ty, int tx, const float* tile, float (&s1)[8], float (&s2)[8]) {
        T* ybase = reinterpret_cast<T*>(p.y) + (((long)b * p.h + ty * TH) * p.w_ + tx * TW) * p.ldy;   // tile origin
        const bool rowok = ty * TH + row < p.h && cok;
        const int xlim = p.w_ - tx * TW;                          // x0 + o < xlim
        raw8<T> oc[4];
        if (p.accumulate && rowok) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (x0 + o < xlim) oc[o].load(ybase + ooff + o * p.ldy);
        }
        float acc[4][8];
        if (p.bias) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][j] = (c0 + j < p.c) ? p.bias[c0 + j] : 0.f;
        } else {
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
        }
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
            float wr[3][8], in[6][8];
#pragma unroll
            for (int k = 0; k < 3; ++k) ld8(wts + (dy * 3 + k) * CB + cg * 8, wr[k]);
#pragma unroll
            for (int k = 0; k < 6; ++k) ld8(tile + ((row + dy) * (TW + 2) + x0 + k) * PS + cg * 8, in[k]);
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    fma8(in[o + k], wr[k], acc[o]);
        }
        if (rowok) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (x0 + o >= xlim) continue;
                T* dst = ybase + ooff + o * p.ldy;
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1[j] += acc[o][j]; s2[j] += acc[o][j] * acc[o][j]; }
                if (p.accumulate) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[o][j] += oc[o].get(j);
                }
                store8<T>(dst, acc[o]);
            }
        }
    };

    __syncthreads();                                             // weights + zeroed `red` visible
    if (loader) {
        TileIter ti; ti.init(tile_range(p.ntiles, gs.bx, gs.nbx), p.tiles_x, p.tiles_y);
        int buf = 0;
        if (ti.valid()) stage(ti.b, ti.ty, ti.tx, tile_base);
        __syncthreads();
        while (ti.valid()) {
            ti.next();                                           // the tile the compute waves will consume next
            if (ti.valid()) stage(ti.b, ti.ty, ti.tx, tile_base + (buf ^ 1) * HALO * PS);
            lds_barrier();
            buf ^= 1;
        }
    } else {
        float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (DB) {
            TileIter ti; ti.init(tile_range(p.ntiles, gs.bx, gs.nbx), p.tiles_x, p.tiles_y);
            int buf = 0;
            __syncthreads();
            for (; ti.valid(); ti.next()) {
                compute(ti.b, ti.ty, ti.tx, tile_base + buf * HALO * PS, s1, s2);
                lds_barrier();
                buf ^= 1;
            }
        } else {
            TileIter ti; ti.init(TileRange{(long)gs.bx, p.ntiles, (long)gs.nbx}, p.tiles_x, p.tiles_y);
            for (; ti.valid(); ti.next()) {
                stage(ti.b, ti.ty, ti.tx, tile_base);
                __syncthreads();
                compute(ti.b, ti.ty, ti.tx, tile_base, s1, s2);
                __syncthreads();
            }
        }
        if (p.stats) {
            // lanes with equal (lane & 3) share a channel group: fold the 16 of them, then 4 LDS atomics per wave
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] = row_fold<4>(s1[j]); s2[j] = row_fold<4>(s2[j]);
            }
            if ((tid & 15) < 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { atomicAdd(&red[cg * 8 + j], s1[j]); atomicAdd(&red[CB + cg * 8 + j], s2[j]); }
            }
        }
    }
    __syncthreads();
    if (p.stats && tid < 2 * CB) {
        const int cc = tid & (CB - 1), which = tid / CB;
        if (c_base + cc < p.c && red[tid] != 0.f) {
            float* rep = p.stats + ((blockIdx.x + blockIdx.y) & (ISA_STAT_R - 1)) * 2 * p.c;
            atomicAdd(rep + which * p.c + c_base + cc, red[tid]);
        }
    }
}

// wgrad: slab[blockIdx.x][t*CB + cc] (t < 9) and [9*CB + cc] (bias) for channel block blockIdx.y
template <typename T, bool HAS_PRO, int ACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dw2_wgrad_kernel(Dw2Params p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* tile = sm;                       // [HALO][PS]
    float* red = sm + HALO * PS;            // [10*CB]
    const int tid = threadIdx.x;
    const int c_base = blockIdx.y * CB;
    const int cg = tid & 3, g = tid >> 2, row = g >> 3, x0 = (g & 7) * 4;
    const int c0 = c_base + cg * 8;
    const GroupSel gs = group_sel(p.G);
    if (gs.g) {
        const long img = (long)gs.g * p.n, pix = img * p.h * p.w_;
        p.x = reinterpret_cast<const T*>(p.x) + pix * p.ldx;
        p.dy = reinterpret_cast<const T*>(p.dy) + pix * p.ldd;
        p.pro.scale = goff(p.pro.scale, (long)gs.g * p.c); p.pro.shift = goff(p.pro.shift, (long)gs.g * p.c);
        p.pro.bscale = goff(p.pro.bscale, img * p.c);
    }
    const T* din = reinterpret_cast<const T*>(p.dy);
    float acc[9][8], db[8];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[tp][j] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) db[j] = 0.f;
    for (long t = gs.bx; t < p.ntiles; t += gs.nbx) {
        const int tx = (int)(t % p.tiles_x); const long q = t / p.tiles_x;
        const int ty = (int)(q % p.tiles_y); const int b = (int)(q / p.tiles_y);
        ProRegs pr;
        load_pro<HAS_PRO>(p, pr, c0, b);
        __syncthreads();
        stage_tile<T, HAS_PRO, ACT>(p, pr, tile, b, ty, tx, c_base);
        // this lane's four output gradients straight from global, issued once the staging registers are free;
        // they stay in flight across the barrier
        float d[4][8];
        const int oy = ty * TH + row;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int ox = tx * TW + x0 + o;
#pragma unroll
            for (int j = 0; j < 8; ++j) d[o][j] = 0.f;
            if (oy < p.h && ox < p.w_ && c0 < p.c) load8<T>(din + (((long)b * p.h + oy) * p.w_ + ox) * p.ldd + c0, d[o]);
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) db[j] += d[o][j];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            float in[6][8];
#pragma unroll
            for (int k = 0; k < 6; ++k) ld8(tile + ((row + dy) * (TW + 2) + x0 + k) * PS + cg * 8, in[k]);
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    fma8(in[o + k], d[o], acc[dy * 3 + k]);
            __builtin_amdgcn_sched_barrier(0);       // keep one row of LDS reads live at a time (register budget)
        }
    }
    __syncthreads();
    for (int i = tid; i < 10 * CB; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int tp = 0; tp < 10; ++tp) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = tp < 9 ? acc[tp < 9 ? tp : 0][j] : db[j];
            v = row_fold<4>(v);
            if ((tid & 15) < 4) atomicAdd(&red[tp * CB + cg * 8 + j], v);
        }
    }
    __syncthreads();
    float* slab = p.ws + ((long)blockIdx.x * gridDim.y + blockIdx.y) * 10 * CB;
    for (int i = tid; i < 10 * CB; i += 256) slab[i] = red[i];
}

constexpr int DW2_RSPLIT = 16;
// dw[c][t] += sum_b slab[b][cb][t*CB+cc]; dbias[c] += slab[...][9*CB+cc]
__global__ __launch_bounds__(256) void dw2_wgrad_reduce_kernel(const float* ws, int nblk, int ncb, int C, int csrc, float* dw, float* dbias) {
    const int cb = blockIdx.x, split = blockIdx.y;
    const int per = (nblk + DW2_RSPLIT - 1) / DW2_RSPLIT;
    const int b0 = split * per, b1 = min(nblk, b0 + per);
    if (b0 >= b1) return;
    for (int i = threadIdx.x; i < 10 * CB; i += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += ws[((long)b * ncb + cb) * 10 * CB + i];
        const int tp = i / CB, c = cb * CB + (i - tp * CB);
        if (c >= csrc || c >= C) continue;
        if (tp < 9) atomicAdd(dw + c * 9 + tp, s);
        else if (dbias) atomicAdd(dbias + c, s);
    }
}

template <typename T, bool HAS_PRO, int ACT>
int launch_fwd2_inst(Dw2Params& p, dim3 grid, hipStream_t s) {
    constexpr bool DB = sizeof(T) == 2;
    constexpr size_t lds = ((size_t)(DB ? 2 : 1) * HALO * PS + 9 * CB + 2 * CB + 2 * CB) * 4;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dw2_fwd_kernel<T, HAS_PRO, ACT, DB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISA_ELAUNCH;
        configured = true;
    }
    hipLaunchKernelGGL((dw2_fwd_kernel<T, HAS_PRO, ACT, DB>), grid, dim3(DB ? 768 : 256), lds, s, p);
    return launch_status();
}

template <typename T>
int launch_fwd2(Dw2Params& p, bool has_pro, hipStream_t s) {
    p.tiles_x = (p.w_ + TW - 1) / TW; p.tiles_y = (p.h + TH - 1) / TH;
    p.ntiles = (long)p.n * p.tiles_x * p.tiles_y;            // p.n: images per statistic group
    if (p.ntiles * p.G >= (1L << 31)) return ISA_EINVAL;
    const int ncb = (p.c + CB - 1) / CB;
    // persistent: bf16 = one 512-thread double-buffered workgroup per CU, f32 = three 256-thread ones
    long gx = (256L * (sizeof(T) == 2 ? 1 : 3)) / ncb;
    if (gx < 1) gx = 1;
    if (gx > p.ntiles * p.G) gx = p.ntiles * p.G;
    gx = group_grid(gx, p.G);
    dim3 grid((unsigned)gx, ncb);
    if (has_pro && p.pro.act == ISA_ACT_RELU6) return launch_fwd2_inst<T, true, ISA_ACT_RELU6>(p, grid, s);
    if (has_pro) return launch_fwd2_inst<T, true, ACT_RT>(p, grid, s);
    return launch_fwd2_inst<T, false, ISA_ACT_NONE>(p, grid, s);
}

template <typename T>
int launch_wg2(Dw2Params& p, bool has_pro, long ws_floats, isa_slab_arena* sa, hipStream_t s) {
    p.tiles_x = (p.w_ + TW - 1) / TW; p.tiles_y = (p.h + TH - 1) / TH;
    p.ntiles = (long)p.n * p.tiles_x * p.tiles_y;
    const int ncb = (p.c + CB - 1) / CB;
    long gx = (256L * 2) / ncb;          // 2 resident workgroups per CU
    if (gx < 1) gx = 1;
    if (gx > p.ntiles * p.G) gx = p.ntiles * p.G;
    if (int rc = defer_ws(sa, &p.ws, &ws_floats)) return rc;
    const long ws_cap = ws_floats / (10L * CB * ncb);
    if (ws_cap < p.G) return sa ? ISA_ENOMEM : ISA_EINVAL;
    if (gx > ws_cap) gx = ws_cap;
    gx = group_grid(gx, p.G);
    dim3 grid((unsigned)gx, ncb);
    const size_t lds = ((size_t)HALO * PS + 10 * CB) * 4;
    if (has_pro && p.pro.act == ISA_ACT_RELU6) hipLaunchKernelGGL((dw2_wgrad_kernel<T, true, ISA_ACT_RELU6>), grid, dim3(256), lds, s, p);
    else if (has_pro) hipLaunchKernelGGL((dw2_wgrad_kernel<T, true, ACT_RT>), grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL((dw2_wgrad_kernel<T, false, ISA_ACT_NONE>), grid, dim3(256), lds, s, p);
    if (launch_status() != ISA_OK) return ISA_ELAUNCH;
    if (defer_push(sa, FoldDesc{p.ws, (float*)p.y, (float*)p.bias, nullptr, 1, (int)gx, ncb, 9, 1, 0, CB, p.c, 0, p.csrc, 0, 0, 0, 0},
                   gx * ncb * 10L * CB)) return ISA_OK;
    hipLaunchKernelGGL(dw2_wgrad_reduce_kernel, dim3(ncb, DW2_RSPLIT), dim3(256), 0, s, p.ws, (int)gx, ncb, p.c, p.csrc, (float*)p.y, (float*)p.bias);
    return launch_status();
}


// ------------------------------------------------------------------------------------------------
// Fused backward of  x --dw3x3--> y --BN(train)+act--> ...   (InvertedResidual / InvertedV1Residual middle):
//   given g = dL/d act(BN(y)) and the already reduced sums of BN(y)'s backward, one pass over the tiles does
//     1. dy = gamma*invstd*(g*act'(z) - mean(g') - yhat*mean(g'*yhat))   (the BN-backward "apply", never stored)
//     2. dx (+)= dw3x3_flipped(dy)                                          (depthwise data gradient)
//     3. dW += sum  pro(x)[window] * dy                                     (depthwise weight gradient)
//     4. if x is itself a lazy BN output: the sums (sum g_x, sum g_x*xhat) its BN backward needs, from the
//        dx tile still in registers                                         (the next BN-backward "reduce")
//   replacing bn_bwd_apply (3 tensor passes) + dw_wgrad (2) + dw_dgrad (2) + bn_bwd_reduce (2) by
//   3 reads + 1 write.  dy is rounded to the storage type in LDS exactly as the unfused path rounds it in HBM.
struct FusedParams {
    const void *g, *y, *x, *w; void* dx;
    int n, h, w_, c, ldg, ldy, ldx, lddx, wld;
    const float *ysc, *ysh, *ymu, *yis, *yred; float ycnt_inv; int yact; float *ydgamma, *ydbeta;
    const float *xsc, *xsh, *xmu, *xis; int xact; float* xred;
    int accumulate, tiles_x, tiles_y; long ntiles; float* ws; int csrc; float* dw;
    const void* addend; int lda;      // optional: dx += addend (gradient of the block's residual branch)
    int G;                            // statistic groups: n and ntiles are per group (common.hpp)
};


template <typename T> struct dd_stride { static constexpr int v = 36; };       // floats: 144 B / pixel
template <> struct dd_stride<bf16_t> { static constexpr int v = 40; };         // 80 B / pixel: 4 x-groups tile 256 B

// DB (bf16): 512 threads, two LDS tile buffers.  Waves 4-7 (loader role) only stage tiles - 16-byte loads, the
// BN-backward arithmetic of dy, LDS stores - while waves 0-3 (compute role) run the two stencils on the current tile;
// one LDS-only barrier per tile swaps the buffers.  The single-buffer form (f32 storage: the tiles do not fit twice)
// runs the same phases back to back in 256 threads.
// What the measurements said (256x256x64, bf16, 223 us at the start):
//  * PMC: the single-buffer kernel waited on memory for most of each tile -> the role split.
//  * Ablation: no loads 120 us, no loader arithmetic 184 us, no stencils 201 us - every part additive.
//  * A cycle trace of one workgroup: the COMPUTE wave was the critical path.  Its epilogue loaded raw x (for the
//    BN(x) sums) from global memory, and those few loads queued in the CU's in-order vector-memory pipeline behind the
//    loader's bulk stream: every tile waited most of a tile's HBM time for them, the loader idled at the barrier.
//  -> the LDS tile now holds RAW x (storage type).  The compute role applies the lazy prologue itself (the unfused
//     arithmetic, bit for bit), takes the centre pixels for the BN(x) sums from the same tile and, in the common case,
//     reads no global memory at all.  LDS drops from 152 to 109 KB and the weight-gradient window reads halve.
//  -> the loader is software-pipelined across tiles (a tile's worth of loads always in flight) and keeps its
//     per-channel constants in registers: it is VALU-issue bound next to the compute wave of the same SIMD.
// EPI: the tile epilogue has extra operands (accumulate into old dx and / or a residual addend): rare, so the common
// variant compiles their loads and registers out.
template <typename T, int YACT, int XMODE, bool DB, bool EPI>
__global__ __launch_bounds__(DB ? 512 : 256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dw_bn_bwd_kernel(FusedParams p) {
    constexpr int PSD = dd_stride<T>::v;
    constexpr int XACT = XMODE == 1 ? ISA_ACT_RELU6 : (XMODE == 0 ? ISA_ACT_NONE : ACT_RT);
    constexpr int NBUF = DB ? 2 : 1;
    constexpr int NTHR = DB ? 512 : 256;
    constexpr int TILE_ELEMS = HALO * PSD;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    T* xt_base = reinterpret_cast<T*>(sm);                                   // [NBUF][HALO][PSD] RAW x (0 outside the image)
    T* dt_base = xt_base + NBUF * TILE_ELEMS;                                // [NBUF][HALO][PSD] dy
    float* wts = reinterpret_cast<float*>(dt_base + NBUF * TILE_ELEMS);      // [9][CB] flipped taps
    float* cst = wts + 9 * CB;                                               // [10][CB] per-channel constants
    float* red = cst + 10 * CB;                                              // [11*CB]
    const int tid = threadIdx.x;
    const int ltid = tid & 255;                                              // index inside the role group
    const bool loader = DB && tid >= 256;
    const int c_base = blockIdx.y * CB;
    const GroupSel gs = group_sel(p.G);
    if (gs.g) {                                                              // this workgroup's statistic group
        const long pix = (long)gs.g * p.n * p.h * p.w_, gc = (long)gs.g * p.c;
        p.g = reinterpret_cast<const T*>(p.g) + pix * p.ldg; p.y = reinterpret_cast<const T*>(p.y) + pix * p.ldy;
        p.x = reinterpret_cast<const T*>(p.x) + pix * p.ldx; p.dx = reinterpret_cast<T*>(p.dx) + pix * p.lddx;
        if (p.addend) p.addend = reinterpret_cast<const T*>(p.addend) + pix * p.lda;
        p.ysc += gc; p.ysh += gc; p.ymu += gc; p.yis += gc; p.yred += gc * ISA_STAT_R * 2;
        p.xsc = goff(p.xsc, gc); p.xsh = goff(p.xsh, gc); p.xmu = goff(p.xmu, gc); p.xis = goff(p.xis, gc);
        p.xred = goff(p.xred, gc * ISA_STAT_R * 2);
    }
    const int cg = ltid & 3, g4 = ltid >> 2, row = g4 >> 3, x0 = (g4 & 7) * 4;
    const int c0 = c_base + cg * 8;
    const bool cok = c0 < p.c;
    const T* wp = reinterpret_cast<const T*>(p.w);
    for (int i = tid; i < 9 * CB; i += NTHR) {
        const int tp = i / CB, cc = i - tp * CB;
        wts[i] = (c_base + cc < p.c) ? st<T>::ld(wp + (long)tp * p.wld + c_base + cc) : 0.f;
    }
    if (tid < CB) {
        const int c = min(c_base + tid, p.c - 1);
        cst[0 * CB + tid] = p.ysc[c]; cst[1 * CB + tid] = p.ysh[c];
        cst[2 * CB + tid] = p.ymu[c]; cst[3 * CB + tid] = p.yis[c];
        float r0 = 0.f, r1 = 0.f;                                // fold the ISA_STAT_R replicas of BN(y)'s backward sums
#pragma unroll
        for (int r = 0; r < ISA_STAT_R; ++r) { r0 += p.yred[r * 2 * p.c + c]; r1 += p.yred[r * 2 * p.c + p.c + c]; }
        cst[4 * CB + tid] = r0 * p.ycnt_inv; cst[5 * CB + tid] = r1 * p.ycnt_inv;
        cst[6 * CB + tid] = (XMODE && p.xsc) ? p.xsc[c] : 1.f; cst[7 * CB + tid] = (XMODE && p.xsh) ? p.xsh[c] : 0.f;
        cst[8 * CB + tid] = (XMODE && p.xmu) ? p.xmu[c] : 0.f; cst[9 * CB + tid] = (XMODE && p.xis) ? p.xis[c] : 1.f;
        if (gs.bx == 0 && c_base + tid < p.c) {                  // BN(y) parameter gradients: dbeta = sum g', dgamma = sum g'*yhat (per group)
            if (p.ydgamma) atomicAdd(p.ydgamma + c, r1);
            if (p.ydbeta) atomicAdd(p.ydbeta + c, r0);
        }
    }
    for (int i = tid; i < 11 * CB; i += NTHR) red[i] = 0.f;     // rows 0-8: dW taps, 9-10: BN(x) sums
    const T* gin = reinterpret_cast<const T*>(p.g);
    const T* yin = reinterpret_cast<const T*>(p.y);
    const T* xin = reinterpret_cast<const T*>(p.x);
    T* dxo = reinterpret_cast<T*>(p.dx);
    constexpr int NIT = (HALO * 4 + 255) / 256;
    constexpr int NB = 3;                            // single-buffer form: loads in chunks of 3 slots (registers)
    static_assert(NIT % NB == 0, "staging chunks");
    const bool want_xred = XMODE == 1 || (XMODE == 2 && p.xred != nullptr);

    // tile-invariant halo slot geometry (see dw2_fwd_kernel): per tile a scalar base per tensor and, on border tiles,
    // four scalar bounds
    int hrc[NIT], hog[NIT], hoy[NIT], hox[NIT];
    bool hok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pix = (ltid + it * 256) >> 2;
        const int rr = pix / (TW + 2), cc = pix - rr * (TW + 2);
        const int cc0 = cok ? c0 : c_base;
        hrc[it] = (rr << 16) | cc;
        hog[it] = (rr * p.w_ + cc) * p.ldg + cc0;
        hoy[it] = (rr * p.w_ + cc) * p.ldy + cc0;
        hox[it] = (rr * p.w_ + cc) * p.ldx + cc0;
        hok[it] = pix < HALO && cok;
    }
    const int oodx = (row * p.w_ + x0) * p.lddx + c0, ooa = (row * p.w_ + x0) * p.lda + c0;

    // ---- per-tile geometry shared by both staging forms
    struct Geo { const T *gb, *yb, *xb; int rlo, rhi, clo, chi, sg, sy, sx; bool interior; };
    auto geo = [&](int b, int ty, int tx) {
        Geo g;
        const long horg = ((long)b * p.h + ty * TH - 1) * p.w_ + tx * TW - 1;           // halo origin pixel (may lie outside)
        g.gb = gin + horg * p.ldg; g.yb = yin + horg * p.ldy; g.xb = xin + horg * p.ldx;
        g.rlo = 1 - ty * TH; g.rhi = p.h + 1 - ty * TH; g.clo = 1 - tx * TW; g.chi = p.w_ + 1 - tx * TW;
        g.interior = g.rlo <= 0 && g.rhi >= TH + 2 && g.clo <= 0 && g.chi >= TW + 2;
        const int cc0 = cok ? c0 : c_base;
        g.sg = (p.w_ + 1) * p.ldg + cc0; g.sy = (p.w_ + 1) * p.ldy + cc0; g.sx = (p.w_ + 1) * p.ldx + cc0;   // halo (1,1) = tile origin
        return g;
    };
    auto slot_ok = [&](const Geo& g, int it) {
        const int rr = hrc[it] >> 16, cc = hrc[it] & 0xffff;
        return hok[it] && (g.interior || (rr >= g.rlo && rr < g.rhi && cc >= g.clo && cc < g.chi));
    };
    // per-channel constants of the staging arithmetic: tile-invariant, so they live in registers for the whole kernel
    struct StageK { float sc[8], sh[8], mu[8], is[8], k0[8], k1[8]; };
    // one halo slot: dy = BN-backward(g, y) and raw x, both as storage type into the LDS tiles
    auto convert = [&](bool ok, int it, const StageK& K, const raw8<T>& gv, const raw8<T>& yv, const raw8<T>& xv, T* xt, T* dt) {
        const int pix = (ltid + it * 256) >> 2;
        if (pix >= HALO) return;
        float o[8];
        raw8<T> q = xv;                                           // raw x travels as it is: 16 bytes, no conversion
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float yy = yv.get(j);
            const float z = fmaf(yy, K.sc[j], K.sh[j]);
            const float dz = gv.get(j) * act_grad_t<YACT>(z, p.yact);
            const float yh = (yy - K.mu[j]) * K.is[j];
            o[j] = K.sc[j] * (dz - K.k0[j] - yh * K.k1[j]);
        }
        // out-of-image halo slots were fetched from the tile's origin pixel: zero them.  Interior tiles have none and the
        // test is uniform over the wave there, so the 16 selects are skipped
        if (__builtin_amdgcn_ballot_w64(!ok) != 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = ok ? o[j] : 0.f;
            if (!ok) q.zero();
        }
        store8<T>(dt + pix * PSD + cg * 8, o);
        q.store(xt + pix * PSD + cg * 8);
    };
    // Loads are unconditional (clamped address) so a chunk loop is straight-line code.  The empty asm keeps them where
    // they are written: LLVM otherwise sinks a prefetch down to its first use - across the LDS-only barrier and the loop
    // back-edge, into the `ok` branch - which undoes it and turns every partial vmcnt wait into a full one.
    auto issue = [&](const Geo& g, int it, raw8<T>& gv, raw8<T>& yv, raw8<T>& xv) {
        const bool ok = slot_ok(g, it);
        gv.load(g.gb + (ok ? hog[it] : g.sg));
        yv.load(g.yb + (ok ? hoy[it] : g.sy));
        xv.load(g.xb + (ok ? hox[it] : g.sx));
        asm volatile("" ::: "memory");
    };

    // single-buffer form: one tile, chunks of NB slots
    auto stage = [&](int b, int ty, int tx, T* xt, T* dt) {
        const Geo g = geo(b, ty, tx);
        StageK K;                                                 // per tile here: these threads also hold the accumulators
        ld8(cst + 0 * CB + cg * 8, K.sc); ld8(cst + 1 * CB + cg * 8, K.sh); ld8(cst + 2 * CB + cg * 8, K.mu);
        ld8(cst + 3 * CB + cg * 8, K.is); ld8(cst + 4 * CB + cg * 8, K.k0); ld8(cst + 5 * CB + cg * 8, K.k1);
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += NB) {
            raw8<T> gv[NB], yv[NB], xv[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) issue(g, it0 + u, gv[u], yv[u], xv[u]);
#pragma unroll
            for (int u = 0; u < NB; ++u) convert(slot_ok(g, it0 + u), it0 + u, K, gv[u], yv[u], xv[u], xt, dt);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto compute = [&](int b, int ty, int tx, const T* xt, const T* dt, float (&acc)[9][8], float (&s0)[8], float (&s1)[8]) {
        const long torg = ((long)b * p.h + ty * TH) * p.w_ + tx * TW;                      // tile origin pixel
        T* dxb = dxo + torg * p.lddx + oodx;
        const T* adb = reinterpret_cast<const T*>(p.addend) + torg * p.lda + ooa;
        const bool rowok = ty * TH + row < p.h && cok;
        const int xlim = p.w_ - tx * TW;                                                   // x0 + o < xlim
        // rare epilogue operands (old dx for accumulate, the residual addend): requested before the stencil
        raw8<T> oc[4], ad[4];
        if constexpr (EPI) {
            if (rowok) {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (x0 + o >= xlim) continue;
                    if (p.accumulate) oc[o].load(dxb + o * p.lddx);
                    if constexpr (XMODE == 0) { if (p.addend) ad[o].load(adb + o * p.lda); }
                }
            }
        }
        {   // ---- data gradient: dx tile = flipped taps over dy (halo), then BN(x)-backward sums
            float a[4][8];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[o][j] = 0.f;
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy) {
                float wr[3][8], in[6][8];
#pragma unroll
                for (int k = 0; k < 3; ++k) ld8(wts + (dy * 3 + k) * CB + cg * 8, wr[k]);
#pragma unroll
                for (int k = 0; k < 6; ++k) load8<T>(dt + ((row + dy) * (TW + 2) + x0 + k) * PSD + cg * 8, in[k]);
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        fma8(in[o + k], wr[k], a[o]);
            }
            if (rowok) {
                float xs[8], xh[8], xm[8], xi[8];
                if (want_xred) {
                    ld8(cst + 6 * CB + cg * 8, xs); ld8(cst + 7 * CB + cg * 8, xh);
                    ld8(cst + 8 * CB + cg * 8, xm); ld8(cst + 9 * CB + cg * 8, xi);
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (x0 + o >= xlim) continue;
                    T* dst = dxb + o * p.lddx;
                    if constexpr (EPI) {
                        if (p.accumulate) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) a[o][j] += oc[o].get(j);
                        }
                        if constexpr (XMODE == 0) {          // plain-tensor input: the only case with a residual branch
                            if (p.addend) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) a[o][j] += ad[o].get(j);
                            }
                        }
                    }
                    store8<T>(dst, a[o]);
                    if (want_xred) {
                        // the unfused reduce reads the stored (rounded) gradient: round the same way
                        float xr[8];
                        load8<T>(xt + ((row + 1) * (TW + 2) + x0 + 1 + o) * PSD + cg * 8, xr);      // raw x, centre pixel
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float gq = (float)(T)a[o][j];
                            const float z = fmaf(xr[j], xs[j], xh[j]);
                            const float dz = gq * act_grad_t<XACT>(z, p.xact);
                            s0[j] += dz; s1[j] += dz * ((xr[j] - xm[j]) * xi[j]);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // ---- weight gradient: 9 x 8 products per lane, dy centre x pro(x) window
            float d[4][8];
#pragma unroll
            for (int o = 0; o < 4; ++o) load8<T>(dt + ((row + 1) * (TW + 2) + x0 + 1 + o) * PSD + cg * 8, d[o]);
            float ps[8], ph[8];
            if constexpr (XMODE != 0) { ld8(cst + 6 * CB + cg * 8, ps); ld8(cst + 7 * CB + cg * 8, ph); }
            // Zero padding applies to pro(x), not to x: on border tiles the out-of-image window slots are masked after
            // the prologue.  Two copies of the stencil behind a uniform branch - as one body with per-slot predicates
            // the 18 lane masks stayed live across the loop and the accumulators spilled.
            const int rlo = 1 - ty * TH, rhi = p.h + 1 - ty * TH, clo = 1 - tx * TW, chi = p.w_ + 1 - tx * TW;
            const bool interior = XMODE == 0 || (rlo <= 0 && rhi >= TH + 2 && clo <= 0 && chi >= TW + 2);
            auto window = [&](auto border) {
                constexpr bool BORDER = decltype(border)::value;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const bool row_in = row + dy >= rlo && row + dy < rhi;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {                // one window pixel at a time: 8 live registers, not 48
                        float in[8];
                        load8<T>(xt + ((row + dy) * (TW + 2) + x0 + k) * PSD + cg * 8, in);
                        if constexpr (XMODE != 0) {
                            float z[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) z[j] = ph[j];
                            fma8(in, ps, z);                     // packed fma; the clamp has no packed form
#pragma unroll
                            for (int j = 0; j < 8; ++j) in[j] = act_t<XACT>(z[j], p.xact);
                            if constexpr (BORDER) {
                                const bool in_img = row_in && x0 + k >= clo && x0 + k < chi;
#pragma unroll
                                for (int j = 0; j < 8; ++j) in[j] = in_img ? in[j] : 0.f;
                            }
                        }
#pragma unroll
                        for (int o = 0; o < 4; ++o)
                            if (k - o >= 0 && k - o < 3) fma8(in, d[o], acc[dy * 3 + (k - o)]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (interior) window(std::false_type{});
            else window(std::true_type{});
        }
    };

    // The accumulators exist only on the compute side of the role split, so the register allocator sees
    // max(loader set, compute set) instead of their sum.  Both sides execute the same number of barriers.
    __syncthreads();                                             // constants + zeroed `red` visible
    if (loader) {
        // Software-pipelined across tiles: the loads of the tile after next are issued chunk by chunk WHILE the next tile
        // is converted - a chunk's registers are refilled right after its arithmetic consumed them, so a tile's worth of
        // 16-byte loads (3 x NIT per lane) is always in flight and a chunk only waits for loads issued a tile earlier.
        // Barriers: one per staged tile (the first is the compute side's "first tile staged", every later one closes the
        // compute side's previous tile) plus one closing its last tile: n + 1 on both sides.  The steady branch is
        // straight-line code with the same loads in flight at its top and bottom, so its vmcnt waits stay partial.
        StageK K;
        ld8(cst + 0 * CB + cg * 8, K.sc); ld8(cst + 1 * CB + cg * 8, K.sh); ld8(cst + 2 * CB + cg * 8, K.mu);
        ld8(cst + 3 * CB + cg * 8, K.is); ld8(cst + 4 * CB + cg * 8, K.k0); ld8(cst + 5 * CB + cg * 8, K.k1);
        TileIter tn; tn.init(tile_range(p.ntiles, gs.bx, gs.nbx), p.tiles_x, p.tiles_y);     // the tile whose loads are in flight
        raw8<T> gv[NIT], yv[NIT], xv[NIT];
        Geo gs, gn;
        bool have = tn.valid();
        if (have) {
            gn = geo(tn.b, tn.ty, tn.tx);
#pragma unroll
            for (int it = 0; it < NIT; ++it) issue(gn, it, gv[it], yv[it], xv[it]);
        }
        int buf = 0;
        while (have) {
            gs = gn;
            tn.next();
            T* xt = xt_base + buf * TILE_ELEMS; T* dt = dt_base + buf * TILE_ELEMS;
            if (tn.valid()) {
                gn = geo(tn.b, tn.ty, tn.tx);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    convert(slot_ok(gs, it), it, K, gv[it], yv[it], xv[it], xt, dt);
                    issue(gn, it, gv[it], yv[it], xv[it]);         // refill the chunk's registers: the tile after this one
                }
            } else {
#pragma unroll
                for (int it = 0; it < NIT; ++it) convert(slot_ok(gs, it), it, K, gv[it], yv[it], xv[it], xt, dt);
                have = false;
            }
            lds_barrier();
            buf ^= 1;
        }
        lds_barrier();
    } else {
        float acc[9][8], s0[8], s1[8];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[tp][j] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
        if constexpr (DB) {
            TileIter ti; ti.init(tile_range(p.ntiles, gs.bx, gs.nbx), p.tiles_x, p.tiles_y);
            int buf = 0;
            lds_barrier();                                       // first tile staged
            for (; ti.valid(); ti.next()) {
                compute(ti.b, ti.ty, ti.tx, xt_base + buf * TILE_ELEMS, dt_base + buf * TILE_ELEMS, acc, s0, s1);
                lds_barrier();
                buf ^= 1;
            }
        } else {
            TileIter ti; ti.init(TileRange{(long)gs.bx, p.ntiles, (long)gs.nbx}, p.tiles_x, p.tiles_y);
            for (; ti.valid(); ti.next()) {
                stage(ti.b, ti.ty, ti.tx, xt_base, dt_base);
                __syncthreads();
                compute(ti.b, ti.ty, ti.tx, xt_base, dt_base, acc, s0, s1);
                __syncthreads();                                 // tile fully consumed before it is restaged
            }
        }
        // Fold this lane's 88 sums WITHOUT atomics (a phase trace of a one-tile launch: 26 k of its 47 k cycles sat in the
        // old fold - a ds_bpermute tree per value plus same-address LDS float atomics from four waves).  Row sums by
        // DPP, then the 16 rows of the 4 compute waves park their partials in the (now idle) tile buffers:
        // part[(v * 4 + cg) * 16 + wave * 4 + row]; 352 threads add 16 partials each after the barrier below.
        float* part = reinterpret_cast<float*>(sm);
        const int slot = cg * 16 + (tid >> 6) * 4 + ((tid >> 4) & 3);
        const bool writer = (tid & 15) < 4;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = row_fold<4>(acc[tp][j]);
                if (writer) part[(tp * 8 + j) * 64 + slot] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v0 = row_fold<4>(s0[j]), v1 = row_fold<4>(s1[j]);
            if (writer) { part[(72 + j) * 64 + slot] = v0; part[(80 + j) * 64 + slot] = v1; }
        }
    }
    __syncthreads();
    {
        const float* part = reinterpret_cast<const float*>(sm);
        for (int o = tid; o < 88 * 4; o += NTHR) {               // o = v * 4 + cg
            const f32x4 a = *reinterpret_cast<const f32x4*>(part + o * 16), b = *reinterpret_cast<const f32x4*>(part + o * 16 + 4),
                        c = *reinterpret_cast<const f32x4*>(part + o * 16 + 8), d = *reinterpret_cast<const f32x4*>(part + o * 16 + 12);
            const f32x4 t = (a + b) + (c + d);
            const int v = o >> 2, g = o & 3;
            red[(v >> 3) * CB + g * 8 + (v & 7)] = (t[0] + t[1]) + (t[2] + t[3]);
        }
    }
    __syncthreads();
    float* slab = p.ws + ((long)blockIdx.x * gridDim.y + blockIdx.y) * 10 * CB;
    for (int i = tid; i < 10 * CB; i += NTHR) slab[i] = i < 9 * CB ? red[i] : 0.f;       // row 9 = conv bias slot (none here)
    if (want_xred && tid < 2 * CB) {
        const int cc = tid & (CB - 1), which = tid / CB;
        const float v = red[9 * CB + tid];
        if (c_base + cc < p.c && v != 0.f) {
            float* rep = p.xred + ((blockIdx.x + blockIdx.y) & (ISA_STAT_R - 1)) * 2 * p.c;
            atomicAdd(rep + which * p.c + c_base + cc, v);
        }
    }
}

template <typename T, int YACT, int XMODE, bool EPI = true>
int launch_fused_inst(FusedParams& p, dim3 grid, hipStream_t s) {
    constexpr bool DB = sizeof(T) == 2;                          // two pairs of tile buffers: 109 KB for bf16, 196 KB for f32
    if constexpr (EPI) {
        if (!p.accumulate && !p.addend) return launch_fused_inst<T, YACT, XMODE, false>(p, grid, s);
    }
    constexpr int NBUF = DB ? 2 : 1;
    constexpr size_t lds = NBUF * 2 * (size_t)HALO * dd_stride<T>::v * sizeof(T) + (9 + 10 + 11) * CB * 4;   // raw x + dy tiles
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bn_bwd_kernel<T, YACT, XMODE, DB, EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISA_ELAUNCH;
        configured = true;
    }
    hipLaunchKernelGGL((dw_bn_bwd_kernel<T, YACT, XMODE, DB, EPI>), grid, dim3(DB ? 512 : 256), lds, s, p);
    return ISA_OK;
}

template <typename T>
int launch_fused(FusedParams& p, int xmode, long ws_floats, isa_slab_arena* sa, hipStream_t s) {
    p.tiles_x = (p.w_ + TW - 1) / TW; p.tiles_y = (p.h + TH - 1) / TH;
    p.ntiles = (long)p.n * p.tiles_x * p.tiles_y;              // p.n: images per statistic group
    const int ncb = (p.c + CB - 1) / CB;
    const int per_cu = 1;                                        // LDS: 156 KB (bf16, double-buffered) / 100 KB (f32)
    long gx = (256L * per_cu) / ncb;
    if (gx < 1) gx = 1;
    if (gx > p.ntiles * p.G) gx = p.ntiles * p.G;
    if (int rc = defer_ws(sa, &p.ws, &ws_floats)) return rc;
    const long ws_cap = ws_floats / (10L * CB * ncb);
    if (ws_cap < p.G) return sa ? ISA_ENOMEM : ISA_EINVAL;
    if (gx > ws_cap) gx = ws_cap;
    gx = group_grid(gx, p.G);
    dim3 grid((unsigned)gx, ncb);
    int rc;
    const bool y6 = p.yact == ISA_ACT_RELU6;
    if (xmode == 0) rc = y6 ? launch_fused_inst<T, ISA_ACT_RELU6, 0>(p, grid, s) : launch_fused_inst<T, ACT_RT, 0>(p, grid, s);
    else if (xmode == 1) rc = y6 ? launch_fused_inst<T, ISA_ACT_RELU6, 1>(p, grid, s) : launch_fused_inst<T, ACT_RT, 1>(p, grid, s);
    else rc = y6 ? launch_fused_inst<T, ISA_ACT_RELU6, 2>(p, grid, s) : launch_fused_inst<T, ACT_RT, 2>(p, grid, s);
    if (rc != ISA_OK) return rc;
    if (defer_push(sa, FoldDesc{p.ws, p.dw, nullptr, nullptr, 1, (int)gx, ncb, 9, 1, 0, CB, p.c, 0, p.csrc, 0, 0, 0, 0},
                   gx * ncb * 10L * CB)) return ISA_OK;
    hipLaunchKernelGGL(dw2_wgrad_reduce_kernel, dim3(ncb, DW2_RSPLIT), dim3(256), 0, s, p.ws, (int)gx, ncb, p.c, p.csrc, p.dw, (float*)nullptr);
    return launch_status();
}

}  // namespace

// v2 entry points used by dwconv.hip's C-ABI functions when the view allows (C % 8 == 0)
int dw2_forward(const isa_tensor* x, const isa_pro* pro, const void* w, const float* bias, const isa_tensor* y,
                float* stats, int accumulate, void* stream) {
    Dw2Params p{};
    p.x = x->data; p.w = w; p.bias = bias; p.y = y->data;
    p.n = x->n; p.h = x->h; p.w_ = x->w; p.c = x->c; p.ldx = x->ld; p.ldy = y->ld; p.wld = ((x->c + 7) / 8) * 8;
    p.pro = make_pro(pro); p.stats = stats; p.accumulate = accumulate;
    const bool has_pro = !pro_trivial(p.pro);
    p.G = 1;
    const int G = tensor_groups(x);
    if (G > 1 && (stats || p.pro.scale || p.pro.shift)) {
        if (x->n % G || tensor_groups(y) != G) return ISA_EINVAL;
        p.G = G; p.n = x->n / G;
    }
    if (pro && pro->fin) {
        if (!fin_valid(pro)) return ISA_EINVAL;
        p.fin = make_fin(pro);
    }
    if (x->dtype == ISA_BF16) return launch_fwd2<bf16_t>(p, has_pro, as_stream(stream));
    return launch_fwd2<float>(p, has_pro, as_stream(stream));
}

int dw2_wgrad(const isa_tensor* x, const isa_pro* pro, const isa_tensor* dy, float* dw, float* dbias, int csrc,
              float* ws, long ws_floats, isa_slab_arena* defer, void* stream) {
    Dw2Params p{};
    p.x = x->data; p.dy = dy->data; p.y = dw; p.bias = dbias;        // y/bias slots carry the output pointers
    p.n = x->n; p.h = x->h; p.w_ = x->w; p.c = x->c; p.ldx = x->ld; p.ldd = dy->ld;
    p.pro = make_pro(pro); p.ws = ws; p.csrc = (csrc > 0 && csrc < x->c) ? csrc : x->c;
    const bool has_pro = !pro_trivial(p.pro);
    p.G = 1;
    const int G = tensor_groups(x);
    if (G > 1 && (p.pro.scale || p.pro.shift)) {
        if (x->n % G) return ISA_EINVAL;
        p.G = G; p.n = x->n / G;
    }
    if (x->dtype == ISA_BF16) return launch_wg2<bf16_t>(p, has_pro, ws_floats, defer, as_stream(stream));
    return launch_wg2<float>(p, has_pro, ws_floats, defer, as_stream(stream));
}

// Fused BN-apply + depthwise dgrad/wgrad + next BN-reduce; see dw_bn_bwd_kernel.
extern "C" int isa_dwconv3x3_bn_backward(const isa_tensor* g, const isa_tensor* y, const isa_bn_bwd* ybn,
                                         const isa_tensor* x, const isa_pro* xpro, const isa_bn_bwd* xbn,
                                         const void* w_flipped, float* dw, int32_t csrc,
                                         const isa_tensor* dx, int32_t accumulate, const isa_tensor* addend,
                                         float* ws, int64_t ws_floats, isa_slab_arena* defer, void* stream) {
    if (xpro && xpro->fin) { if (int rc = fin_standalone(xpro, x->c, tensor_groups(x), as_stream(stream))) return rc; }   // no in-kernel form here
    if (!tensor_ok(g, 8) || !tensor_ok(y, 8) || !tensor_ok(x, 8) || !tensor_ok(dx, 8)) return ISA_EINVAL;
    if (!ybn || !ybn->scale || !ybn->shift || !ybn->mean || !ybn->invstd || !ybn->red || !(ybn->count > 0)) return ISA_EINVAL;
    if (!w_flipped || !dw || (!ws && !defer)) return ISA_EINVAL;
    if (x->c % 8 != 0) return ISA_EINVAL;
    const isa_tensor* ts[3] = {y, x, dx};
    for (const isa_tensor* t : ts)
        if (t->n != g->n || t->h != g->h || t->w != g->w || t->c != g->c || t->dtype != g->dtype) return ISA_EINVAL;
    const ProDev xp = make_pro(xpro);
    if (xp.bscale) return ISA_EINVAL;                             // per-image scales are not folded here
    if (xbn && (!xbn->mean || !xbn->invstd || !xbn->out_red)) return ISA_EINVAL;
    FusedParams p{};
    p.g = g->data; p.y = y->data; p.x = x->data; p.w = w_flipped; p.dx = dx->data;
    p.n = g->n; p.h = g->h; p.w_ = g->w; p.c = g->c; p.ldg = g->ld; p.ldy = y->ld; p.ldx = x->ld; p.lddx = dx->ld;
    p.wld = ((g->c + 7) / 8) * 8;
    p.ysc = ybn->scale; p.ysh = ybn->shift; p.ymu = ybn->mean; p.yis = ybn->invstd; p.yred = ybn->red;
    p.ycnt_inv = 1.f / ybn->count; p.yact = ybn->act; p.ydgamma = ybn->dgamma; p.ydbeta = ybn->dbeta;
    p.xsc = xp.scale; p.xsh = xp.shift; p.xact = xp.act;
    p.xmu = xbn ? xbn->mean : nullptr; p.xis = xbn ? xbn->invstd : nullptr; p.xred = xbn ? xbn->out_red : nullptr;
    p.accumulate = accumulate; p.ws = ws; p.dw = dw;
    if (addend) {
        if (!tensor_ok(addend, 8) || addend->dtype != g->dtype || addend->c != g->c || addend->n != g->n || addend->h != g->h ||
            addend->w != g->w) return ISA_EINVAL;
        p.addend = addend->data; p.lda = addend->ld;
        if (!pro_trivial(make_pro(xpro)) || xbn) return ISA_EINVAL;      // only for a plain-tensor x
    }
    p.csrc = (csrc > 0 && csrc < g->c) ? csrc : g->c;
    p.G = tensor_groups(g);                                          // BN(y) constants and sums are per statistic group
    if (g->n % p.G) return ISA_EINVAL;
    p.n = g->n / p.G;
    int xmode = 0;
    if (!pro_trivial(xp) || xbn) xmode = (xbn && xp.act == ISA_ACT_RELU6) ? 1 : 2;
    if (g->dtype == ISA_BF16) return launch_fused<bf16_t>(p, xmode, ws_floats, defer, as_stream(stream));
    return launch_fused<float>(p, xmode, ws_floats, defer, as_stream(stream));
}
