// Dense 3x3 convolution (pad 1) for the narrow prediction heads: <= 32 input channels, <= 32 output channels, bf16,
// no lazy prologue.  L0Layer convs (utils.py:696-774: C -> C/2 -> 2 at the 256x256 / 128x128 levels), attend_fc
// (utils.py:631-663: 12 -> 1) and their data gradients.  isa_conv_gemm dispatches here; the generic kernel streams
// the A operand from global memory once per tap (nine dependent loads per pixel tile: 57-100 us at 256x256x32->16,
// 10x off the HBM time).  Here the (8+2) x (32+2) input halo tile is staged ONCE in LDS and the nine taps are nine
// shifted `ds_read_b128` row reads feeding v_mfma_f32_32x32x16_bf16; the 9 x kp x N weights live in registers.
#include "common.hpp"
#include "tr_lds.hpp"

int wgrad_slab_reduce_launch(float* ws, float* dw, float* dbias, int gx, int tn, int tk, int N, int cin, int taps,
                             isa_slab_arena* sa, hipStream_t s);

namespace {

constexpr int TH = 8, TW = 32, HALO = (TH + 2) * (TW + 2);
constexpr int XS = 80;                         // bytes per staged pixel: 32 bf16 + 16 pad (32 rows x 16 B hit 64 banks once)

struct C3Params {
    const bf16_t* x; int n, h, w, cin, ldx;
    const bf16_t* wt; const float* bias;
    bf16_t* y; int N, ldy, accumulate;
    int tiles_x, tiles_y; long ntiles;
};

__global__ __launch_bounds__(256) void conv3x3_tiled_kernel(C3Params p) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* xt = sm;                                             // [HALO][XS]
    float* stage = reinterpret_cast<float*>(sm + HALO * XS) + (threadIdx.x >> 6) * (32 * 33);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int nks = p.cin > 16 ? 2 : 1;                        // 16-channel contraction steps per tap
    // B fragments: W[n = r][tap][16 ks + 8 hh .. +8], packed rows of 9 x 32 (isa_pack_weights kind 0 / 1)
    bf16x8 wb[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wb[t][ks] = bf16x8{0};
            if (r < p.N && ks < nks) wb[t][ks] = *reinterpret_cast<const bf16x8*>(p.wt + ((long)r * 9 + t) * 32 + ks * 16 + 8 * hh);
        }
    const float bv = (p.bias && r < p.N) ? p.bias[r] : 0.f;
    const int cg = tid & 3;                                    // staging: lane keeps one 8-channel group
    const int c0 = cg * 8;
    const int nvalid = max(0, min(8, p.cin - c0));

    for (long t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const int tx = (int)(t % p.tiles_x); const long q = t / p.tiles_x;
        const int ty = (int)(q % p.tiles_y); const int b = (int)(q / p.tiles_y);
        __syncthreads();                                       // previous tile fully consumed
        constexpr int NIT = (HALO * 4 + 255) / 256;
        bf16x8 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + it * 256) >> 2;
            const int rr = pix / (TW + 2), cc = pix - rr * (TW + 2);
            const int gy = ty * TH + rr - 1, gx = tx * TW + cc - 1;
            v[it] = bf16x8{0};
            if (pix < HALO && nvalid > 0 && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w)
                v[it] = *reinterpret_cast<const bf16x8*>(p.x + (((long)b * p.h + gy) * p.w + gx) * p.ldx + c0);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + it * 256) >> 2;
            if (pix >= HALO) continue;
            bf16x8 o = v[it];
            if (nvalid < 8) {                                  // channel tail of the view: lanes beyond cin are not data
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j >= nvalid) o[j] = (bf16_t)0.f;
            }
            *reinterpret_cast<bf16x8*>(xt + pix * XS + c0 * 2) = o;
        }
        __syncthreads();

#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = wave * 2 + half;                   // tile row owned by this wave: 32 pixels = one MFMA M tile
            f32x16 acc = f32x16{0};
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int dy = tp / 3, dx = tp % 3;
                const char* base = xt + ((row + dy) * (TW + 2) + r + dx) * XS + 16 * hh;
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(base);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wb[tp][0], acc, 0, 0, 0);
                if (nks == 2) {
                    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wb[tp][1], acc, 0, 0, 0);
                }
            }
            // C/D layout: lane r = output channel, 16 pixel rows; transpose through the wave's LDS stage
#pragma unroll
            for (int i = 0; i < 16; ++i) stage[((i & 3) + 8 * (i >> 2) + 4 * hh) * 33 + r] = acc[i] + bv;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int px = lane >> 1, cseg = (lane & 1) * 16;
            const int oy = ty * TH + row, ox = tx * TW + px;
            if (oy < p.h && ox < p.w && cseg < p.N) {
                bf16_t* dst = p.y + (((long)b * p.h + oy) * p.w + ox) * p.ldy + cseg;
                float o16[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) o16[j] = stage[px * 33 + cseg + j];
                const bool full = (cseg + 16 <= p.N) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
                if (full) {
                    float lo[8], hi[8];
                    if (p.accumulate) {
                        load8<bf16_t>(dst, lo); load8<bf16_t>(dst + 8, hi);
#pragma unroll
                        for (int j = 0; j < 8; ++j) { lo[j] += o16[j]; hi[j] += o16[8 + j]; }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { lo[j] = o16[j]; hi[j] = o16[8 + j]; }
                    }
                    store8<bf16_t>(dst, lo); store8<bf16_t>(dst + 8, hi);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (cseg + j < p.N) {
                            float o = o16[j];
                            if (p.accumulate) o += (float)dst[j];
                            dst[j] = (bf16_t)o;
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                   // the stage is reused by the next half
        }
    }
}

}  // namespace

// x: [n,h,w,cin] bf16 view (cin <= 32), w: packed rows of [9][32] bf16, y: [n,h,w,N] bf16 view (N <= 32)
int conv3x3_tiled_launch(const isa_tensor* x, const void* w, const float* bias, const isa_tensor* y, int accumulate,
                         hipStream_t s) {
    C3Params p{};
    p.x = (const bf16_t*)x->data; p.n = x->n; p.h = x->h; p.w = x->w; p.cin = x->c; p.ldx = x->ld;
    p.wt = (const bf16_t*)w; p.bias = bias; p.y = (bf16_t*)y->data; p.N = y->c; p.ldy = y->ld; p.accumulate = accumulate;
    p.tiles_x = (p.w + TW - 1) / TW; p.tiles_y = (p.h + TH - 1) / TH;
    p.ntiles = (long)p.n * p.tiles_x * p.tiles_y;
    long gx = p.ntiles < 256 * 2 ? p.ntiles : 256 * 2;         // register-limited: two resident workgroups per CU
    const size_t lds = (size_t)HALO * XS + 4 * 32 * 33 * 4;
    hipLaunchKernelGGL(conv3x3_tiled_kernel, dim3((unsigned)gx), dim3(256), lds, s, p);
    return launch_status();
}

// ---- weight gradient of the same convs: dW[n][k][tap] += sum_px dy[px][n] * x[px + tap][k] -----------------------
// One pass: the x halo tile and the dy tile are staged once; every 16-pixel step fetches ONE dy fragment and nine
// shifted x fragments (ds_read_b64_tr_b16: the contraction index is the pixel) for nine MFMAs.  The generic kernel
// runs nine tap slices that each re-read x and dy.
namespace {

constexpr int WS_ = 64;                        // bytes per staged pixel (32 bf16), both tiles

struct C3WParams {
    const bf16_t* x; const bf16_t* dy; int n, h, w, cin, ldx, N, ldd;
    float* ws; int tiles_x, tiles_y; long ntiles;
};

__global__ __launch_bounds__(256) void conv3x3_wgrad_tiled_kernel(C3WParams p) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* xt = sm;                              // [HALO][64 B]
    char* dt = sm + HALO * WS_;                 // [TH*TW][64 B]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int cg = tid & 3, c0 = cg * 8;
    const int xvalid = max(0, min(8, p.cin - c0)), dvalid = max(0, min(8, p.N - c0));
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x16{0};
    float dbs = 0.f;

    for (long t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const int tx = (int)(t % p.tiles_x); const long q = t / p.tiles_x;
        const int ty = (int)(q % p.tiles_y); const int b = (int)(q / p.tiles_y);
        __syncthreads();
        constexpr int NIT = (HALO * 4 + 255) / 256;
        bf16x8 v[NIT], d[4];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + it * 256) >> 2;
            const int rr = pix / (TW + 2), cc = pix - rr * (TW + 2);
            const int gy = ty * TH + rr - 1, gx = tx * TW + cc - 1;
            v[it] = bf16x8{0};
            if (pix < HALO && xvalid > 0 && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w)
                v[it] = *reinterpret_cast<const bf16x8*>(p.x + (((long)b * p.h + gy) * p.w + gx) * p.ldx + c0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pix = (tid + it * 256) >> 2;                    // 0..255: (row, col) of the output tile
            const int gy = ty * TH + (pix >> 5), gx = tx * TW + (pix & 31);
            d[it] = bf16x8{0};
            if (dvalid > 0 && gy < p.h && gx < p.w)
                d[it] = *reinterpret_cast<const bf16x8*>(p.dy + (((long)b * p.h + gy) * p.w + gx) * p.ldd + c0);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = (tid + it * 256) >> 2;
            if (pix >= HALO) continue;
            bf16x8 o = v[it];
            if (xvalid < 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j >= xvalid) o[j] = (bf16_t)0.f;
            }
            *reinterpret_cast<bf16x8*>(xt + pix * WS_ + c0 * 2) = o;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pix = (tid + it * 256) >> 2;
            bf16x8 o = d[it];
            if (dvalid < 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j >= dvalid) o[j] = (bf16_t)0.f;
            }
            *reinterpret_cast<bf16x8*>(dt + pix * WS_ + c0 * 2) = o;
        }
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = wave * 2 + half;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 a = tr_frag(dt, WS_, row * TW + 16 * s, 0, lane);            // A[n][pixel]
#pragma unroll
                for (int j = 0; j < 8; ++j) dbs += (float)a[j];
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int dy = tp / 3, dx = tp % 3;
                    const bf16x8 bfr = tr_frag(xt, WS_, (row + dy) * (TW + 2) + 16 * s + dx, 0, lane);   // B[pixel][k]
                    acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr, acc[tp], 0, 0, 0);
                }
            }
        }
    }
    // ---- fold the four waves, one slab set per workgroup: [tap][1024 fragment floats + 32 bias sums]
    __syncthreads();
    float* red = reinterpret_cast<float*>(sm);
    constexpr int ACC_FLOATS = 9 * 1024;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float* qd = red + (tp * 16 + e) * 64 + lane;
                    *qd = (w == 0 ? 0.f : *qd) + acc[tp][e];
                }
            const float sv = dbs + __shfl_xor(dbs, 32, 64);
            if (hh == 0) { float* qd = red + ACC_FLOATS + r; *qd = (w == 0 ? 0.f : *qd) + sv; }
        }
        __syncthreads();
    }
    constexpr int SLABF = 1024 + 32;
    for (int i = tid; i < 9 * SLABF; i += 256) {
        const int tp = i / SLABF, idx = i - tp * SLABF;
        float val = idx < 1024 ? red[tp * 1024 + idx] : (tp == 0 ? red[ACC_FLOATS + idx - 1024] : 0.f);
        p.ws[((long)blockIdx.x * 9 + tp) * SLABF + idx] = val;
    }
}

}  // namespace

int conv3x3_wgrad_tiled_launch(const isa_tensor* x, const isa_tensor* dy, float* dw, float* dbias, float* ws, long ws_floats,
                               isa_slab_arena* sa, hipStream_t s) {
    C3WParams p{};
    p.x = (const bf16_t*)x->data; p.dy = (const bf16_t*)dy->data; p.n = x->n; p.h = x->h; p.w = x->w;
    if (int rc = defer_ws(sa, &ws, &ws_floats)) return rc;
    p.cin = x->c; p.ldx = x->ld; p.N = dy->c; p.ldd = dy->ld; p.ws = ws;
    p.tiles_x = (p.w + TW - 1) / TW; p.tiles_y = (p.h + TH - 1) / TH;
    p.ntiles = (long)p.n * p.tiles_x * p.tiles_y;
    long gx = p.ntiles < 256 ? p.ntiles : 256;
    const long cap = ws_floats / (9L * (1024 + 32));
    if (cap < 1) return sa ? ISA_ENOMEM : ISA_EINVAL;
    if (gx > cap) gx = cap;
    const size_t tiles = (size_t)HALO * WS_ + (size_t)TH * TW * WS_, redb = (9 * 1024 + 32) * 4;
    const size_t lds = tiles > redb ? tiles : redb;
    hipLaunchKernelGGL(conv3x3_wgrad_tiled_kernel, dim3((unsigned)gx), dim3(256), lds, s, p);
    if (launch_status() != ISA_OK) return ISA_ELAUNCH;
    return wgrad_slab_reduce_launch(ws, dw, dbias, (int)gx, 1, 1, p.N, p.cin, 9, sa, s);
}
