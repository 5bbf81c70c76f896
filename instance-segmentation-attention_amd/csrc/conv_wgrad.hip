// Weight gradient of the conv_gemm family: dW[n][k][tap] += sum_m dy[m,n] * pro(x)[m@tap, k],
// written straight into the reference's state_dict layout (fp32) with float atomics.
//
// The contraction runs over pixels, so for v_mfma_f32_32x32x2_f32 both operands are ROW reads of
// the NHWC tiles: A[i=n][k'=pixel] -> lanes 0..31 read 32 consecutive channels of pixel 2s,
// lanes 32..63 of pixel 2s+1 (and the same for B with the input channels).  No transpose, no
// strided gathers.  Tiles are staged once per workgroup in LDS as fp32 (the lazy BN/ReLU6
// prologue is applied there, once per element), then every wave feeds its 32x32 output tiles from
// conflict-free ds_read_b32 rows.  bf16 storage converts to fp32 while staging: wgrad is <10 % of
// step FLOPs at 1/16 MFMA rate vs. HBM time of the same tensors, so this stays memory-bound; the
// bf16-MFMA + ds_read_b64_tr_b16 variant is listed in DESIGN.md as follow-up.
//
// Work split: blockIdx.x strides over 32-pixel chunks (split-M, combined by atomics),
// blockIdx.y = (n-block, k-block) of <=128x128 outputs, blockIdx.z = tap.
#include "common.hpp"

namespace {

constexpr int PM = 32;          // pixels per staged chunk
constexpr int TB = 128;         // max output rows/cols per block

struct WgParams {
    const void* x; int xh, xw, cin, ldx;          // input image
    const void* dy; int dh, dw_, cdy, ldd;        // output-gradient image
    int mh, mw; long M;                           // M-grid (see conv_gemm)
    ProDev pro;
    float* dw; float* dbias;
    const int32_t* kmap; int ksrc;                // physical->source channel map, source K
    int N, taps, in_mode, out_mode, cout;
    int nb_n, nb_k;                               // number of 128-blocks along N and K
    int nchunks;
};

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int bn = blockIdx.y / p.nb_k, bk = blockIdx.y % p.nb_k;
    const int tap = blockIdx.z;
    const int n0 = bn * TB, k0 = bk * TB;
    const int nb = min(TB, p.N - n0), kb = min(TB, p.cin - k0);     // valid extents
    const int nt = (nb + 31) / 32, kt = (kb + 31) / 32;             // 32-tiles
    const int ldn = nt * 32 + 1, ldk = kt * 32 + 1;                 // +1: rows land on distinct banks
    float* sD = lds;                       // [PM][ldn]
    float* sX = lds + PM * ldn;            // [PM][ldk]
    const int ntiles = nt * kt;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x16{0};
    float dbsum = 0.f;                     // thread tid<nb accumulates column tid of dY
    const T* xin = reinterpret_cast<const T*>(p.x);
    const T* din = reinterpret_cast<const T*>(p.dy);

    for (int chunk = blockIdx.x; chunk < p.nchunks; chunk += gridDim.x) {
        const long mbase = (long)chunk * PM;
        __syncthreads();
        // ---- stage dY rows: element (pix, n) for n in [n0, n0+nt*32) ------------------------
        for (int i = tid; i < PM * nt * 32; i += 256) {
            const int col = i % (nt * 32), pix = i / (nt * 32);
            const long m = mbase + pix;
            float v = 0.f;
            if (m < p.M && col < nb) {
                const int px = (int)(m % p.mw); const long q = m / p.mw;
                const int py = (int)(q % p.mh); const int pb = (int)(q / p.mh);
                int n = n0 + col; long off;
                if (p.out_mode == ISA_OUT_SHUFFLE2) {
                    // forward wrote column (tap q, co) to pixel (2y+dy,2x+dx); here tap == q
                    off = (((long)pb * p.dh + 2 * py + (tap >> 1)) * p.dw_ + 2 * px + (tap & 1)) * p.ldd + n;
                } else {
                    off = m * p.ldd + n;
                }
                v = st<T>::ld(din + off);
            }
            sD[pix * ldn + col] = v;
        }
        // ---- stage X rows with the lazy prologue ---------------------------------------------
        for (int i = tid; i < PM * kt * 32; i += 256) {
            const int col = i % (kt * 32), pix = i / (kt * 32);
            const long m = mbase + pix;
            float v = 0.f;
            if (m < p.M && col < kb) {
                const int px = (int)(m % p.mw); const long q = m / p.mw;
                const int py = (int)(q % p.mh); const int pb = (int)(q / p.mh);
                int sy = py, sx = px; bool ok = true;
                if (p.in_mode == ISA_IN_3X3) {
                    sy = py + tap / 3 - 1; sx = px + tap % 3 - 1;
                    ok = sy >= 0 && sy < p.xh && sx >= 0 && sx < p.xw;
                }
                if (ok) {
                    const int k = k0 + col;
                    v = st<T>::ld(xin + (((long)pb * p.xh + sy) * p.xw + sx) * p.ldx + k);
                    if (p.pro.scale) v *= p.pro.scale[k];
                    if (p.pro.shift) v += p.pro.shift[k];
                    v = act_apply(v, p.pro.act);
                    if (p.pro.bscale) v *= p.pro.bscale[(long)pb * p.cin + k];
                }
            }
            sX[pix * ldk + col] = v;
        }
        __syncthreads();
        if (p.dbias && bk == 0 && tap == 0 && tid < nb) {
#pragma unroll 8
            for (int pix = 0; pix < PM; ++pix) dbsum += sD[pix * ldn + tid];
        }
#pragma unroll
        for (int s = 0; s < PM / 2; ++s) {
            const int row = 2 * s + hh;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = wave + 4 * i;
                if (t < ntiles) {
                    const int ni = t / kt, ki = t - ni * kt;
                    const float a = sD[row * ldn + ni * 32 + r];
                    const float b = sX[row * ldk + ki * 32 + r];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: D[row = n][col = k] -> atomics into the reference layout -----------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = wave + 4 * i;
        if (t >= ntiles) continue;
        const int ni = t / kt, ki = t - ni * kt;
        const int kd = k0 + ki * 32 + r;
        if (kd >= p.cin) continue;
        const int k = p.kmap ? p.kmap[kd] : kd;
        if (k < 0) continue;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = n0 + ni * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh;
            if (n >= p.N) continue;
            long off;
            if (p.out_mode == ISA_OUT_SHUFFLE2) off = ((long)k * p.N + n) * 4 + tap;   // [K][Co][2][2]
            else off = ((long)n * p.ksrc + k) * p.taps + tap;                          // [N][K][kh][kw]
            atomicAdd(p.dw + off, acc[i][j]);
        }
    }
    if (p.dbias && bk == 0 && tap == 0 && tid < nb) atomicAdd(p.dbias + n0 + tid, dbsum);
}

}  // namespace

extern "C" int isa_conv_wgrad(const isa_tensor* x, const isa_pro* pro, const isa_tensor* dy,
                              float* dw, float* dbias, int32_t in_mode, int32_t out_mode,
                              const int32_t* kmap, int32_t ksrc, void* stream) {
    if (!tensor_ok(x, 1) || !tensor_ok(dy, 1) || !dw || x->dtype != dy->dtype) return ISA_EINVAL;
    if (in_mode == ISA_IN_GATHER2) return ISA_EINVAL;
    WgParams p{};
    p.x = x->data; p.xh = x->h; p.xw = x->w; p.cin = x->c; p.ldx = x->ld;
    p.dy = dy->data; p.dh = dy->h; p.dw_ = dy->w; p.cdy = dy->c; p.ldd = dy->ld;
    p.mh = x->h; p.mw = x->w; p.M = (long)x->n * x->h * x->w;
    p.pro = make_pro(pro); p.dw = dw; p.dbias = dbias; p.kmap = kmap; p.ksrc = ksrc > 0 ? ksrc : x->c;
    p.in_mode = in_mode; p.out_mode = out_mode;
    if (out_mode == ISA_OUT_SHUFFLE2) {
        if (in_mode != ISA_IN_1X1 || dy->h != 2 * x->h || dy->w != 2 * x->w || dy->n != x->n) return ISA_EINVAL;
        p.taps = 4; p.N = dy->c;
        // dbias of a transposed conv sums over all four quadrants: handled by the caller via
        // isa_colsum on dy (kept out of this kernel so tap blocks stay independent)
        if (dbias) return ISA_EINVAL;
    } else {
        if (dy->h != x->h || dy->w != x->w || dy->n != x->n) return ISA_EINVAL;
        p.taps = in_mode == ISA_IN_3X3 ? 9 : 1; p.N = dy->c;
    }
    p.cout = dy->c;
    p.nb_n = (p.N + TB - 1) / TB; p.nb_k = (p.cin + TB - 1) / TB;
    p.nchunks = (int)((p.M + PM - 1) / PM);
    const int tiles_yz = p.nb_n * p.nb_k * p.taps;
    int gx = p.nchunks;
    const int cap = max(1, 1024 / tiles_yz);
    if (gx > cap) gx = cap;
    dim3 grid(gx, p.nb_n * p.nb_k, p.taps);
    const int mx = min(TB, ((p.N + 31) / 32) * 32), kx = min(TB, ((p.cin + 31) / 32) * 32);
    const size_t lds = (size_t)PM * ((mx + 1) + (kx + 1)) * 4;
    if (x->dtype == ISA_BF16)
        hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), lds, as_stream(stream), p);
    else
        hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), lds, as_stream(stream), p);
    return launch_status();
}

// column sums of an NHWC view: out[c] += sum over pixels (bias gradients of transposed convs, heads)
namespace {
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, long pixels, int c, int ld, float* out) {
    extern __shared__ float red[];
    for (int i = threadIdx.x; i < c; i += 256) red[i] = 0.f;
    __syncthreads();
    // thread -> channel (tid % c) when c <= 256, rows strided
    const int lanes_c = c < 256 ? c : 256;
    const int rows_per_pass = 256 / lanes_c;
    const int ch = threadIdx.x % lanes_c, rsub = threadIdx.x / lanes_c;
    if (rsub < rows_per_pass) {
        for (int cc = ch; cc < c; cc += lanes_c) {
            float s = 0.f;
            for (long pix = (long)blockIdx.x * rows_per_pass + rsub; pix < pixels; pix += (long)gridDim.x * rows_per_pass)
                s += st<T>::ld(x + pix * ld + cc);
            atomicAdd(&red[cc], s);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256)
        if (red[i] != 0.f) atomicAdd(out + i, red[i]);
}
}  // namespace

extern "C" int isa_colsum(const isa_tensor* x, float* out, void* stream) {
    if (!tensor_ok(x, 1) || !out) return ISA_EINVAL;
    const long pixels = (long)x->n * x->h * x->w;
    const int grid = grid_cap(cdiv(pixels, 64), 512);
    if (x->dtype == ISA_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(grid), dim3(256), x->c * 4, as_stream(stream),
                           (const bf16_t*)x->data, pixels, x->c, x->ld, out);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, dim3(grid), dim3(256), x->c * 4, as_stream(stream),
                           (const float*)x->data, pixels, x->c, x->ld, out);
    return launch_status();
}
