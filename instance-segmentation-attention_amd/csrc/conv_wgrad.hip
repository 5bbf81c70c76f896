// Weight gradient of the conv_gemm family: dW[n][k][tap] += sum_m dy[m,n] * pro(x)[m@tap, k],
// written straight into the reference's state_dict layout (fp32) with float atomics.
//
// The contraction runs over pixels, so for v_mfma_f32_32x32x2_f32 both operands are ROW reads of
// NHWC tiles: A[i=n][k'=pixel] -> lanes 0..31 read 32 consecutive channels of pixel 2s, lanes 32..63
// of pixel 2s+1 (same for B with the input channels): no transpose, no strided gathers.
// v2 structure (gfx950):
//   * every WAVE owns its own 16-pixel chunks: it stages dY[16][TN*32] and pro(X)[16][TK*32] as fp32
//     in a wave-private LDS slab with 16-byte global loads (a lane = 8 channels of one pixel; the
//     lazy BN/ReLU6 prologue is applied in registers, once per element), then feeds TN x TK output
//     tiles from conflict-free ds_read_b32 rows.  No __syncthreads in the main loop, the next
//     chunk's global loads are issued before the current chunk's MFMAs.
//   * at the end the four waves of a workgroup add their accumulators in LDS and ONE wave issues the
//     float atomics: 4x fewer atomics on the (small, hot) dW addresses.
//   * blockIdx.x strides over chunks (split-M), blockIdx.y = (n-group, k-group) of <= 2x2 tiles,
//     blockIdx.z = tap (3x3 dense / transposed-conv quadrant).
// bf16 storage converts to fp32 while staging; exact-fp32 MFMA keeps weight gradients at fp32 accuracy
// in both storage modes.  (bf16-MFMA + ds_read_b64_tr_b16 variant: DESIGN.md follow-ups.)
#include "common.hpp"

namespace {

constexpr int PM = 16;          // pixels per wave-chunk (8 MFMA k-steps)

struct WgParams {
    const void* x; int xh, xw, cin, ldx;
    const void* dy; int dh, dw_, ldd;
    int mh, mw; long M;
    ProDev pro;
    float* dw; float* dbias;
    const int32_t* kmap; int ksrc;
    int N, taps, in_mode, out_mode;
    int groups_k;
    long nchunks;
};

template <typename T>
__device__ __forceinline__ void ldvec(const T* p, float (&v)[8], int nvalid) {
    if (nvalid >= 8) load8<T>(p, v);
    else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = i < nvalid ? st<T>::ld(p + i) : 0.f;
    }
}

template <typename T, int TN, int TK, int ACT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LDN = TN * 32 + 4, LDK = TK * 32 + 4;      // +4 floats: rows stay 16-byte aligned
    constexpr int SLAB = PM * (LDN + LDK);
    constexpr int VN = TN * 4, VK = TK * 4;                  // 8-channel vectors per pixel row
    constexpr int LOADS_N = PM * VN / 64, LOADS_K = PM * VK / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int gn = blockIdx.y / p.groups_k, gk = blockIdx.y % p.groups_k;
    const int tap = blockIdx.z;
    const int n0 = gn * TN * 32, k0 = gk * TK * 32;
    float* sD = lds + wave * SLAB;
    float* sX = sD + PM * LDN;
    const T* xin = reinterpret_cast<const T*>(p.x);
    const T* din = reinterpret_cast<const T*>(p.dy);
    const bool plain = p.in_mode == ISA_IN_1X1 && p.out_mode == ISA_OUT_PLAIN;

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = f32x16{0};
    float dbs[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) dbs[i] = 0.f;

    float rd[LOADS_N][8], rx[LOADS_K][8];

    // global -> registers for one chunk (prologue applied to x here)
    auto fetch = [&](long chunk) {
        const long mbase = chunk * PM;
#pragma unroll
        for (int v = 0; v < LOADS_N; ++v) {
            const int idx = v * 64 + lane, pix = idx / VN, c0 = n0 + (idx % VN) * 8;
            const long m = mbase + pix;
#pragma unroll
            for (int j = 0; j < 8; ++j) rd[v][j] = 0.f;
            if (m < p.M && c0 < p.N) {
                long row = m;
                if (p.out_mode == ISA_OUT_SHUFFLE2) {
                    const int px = (int)(m % p.mw); const long q = m / p.mw;
                    const int py = (int)(q % p.mh); const long pb = q / p.mh;
                    row = (pb * p.dh + 2 * py + (tap >> 1)) * p.dw_ + 2 * px + (tap & 1);
                }
                ldvec<T>(din + row * p.ldd + c0, rd[v], p.N - c0);
            }
        }
#pragma unroll
        for (int v = 0; v < LOADS_K; ++v) {
            const int idx = v * 64 + lane, pix = idx / VK, c0 = k0 + (idx % VK) * 8;
            const long m = mbase + pix;
#pragma unroll
            for (int j = 0; j < 8; ++j) rx[v][j] = 0.f;
            if (m < p.M && c0 < p.cin) {
                long row = m; bool ok = true; long pb = 0;
                if (!plain || p.pro.bscale) {
                    const int px = (int)(m % p.mw); const long q = m / p.mw;
                    const int py = (int)(q % p.mh); pb = q / p.mh;
                    if (p.in_mode == ISA_IN_3X3) {
                        const int sy = py + tap / 3 - 1, sx = px + tap % 3 - 1;
                        ok = sy >= 0 && sy < p.xh && sx >= 0 && sx < p.xw;
                        row = (pb * p.xh + sy) * p.xw + sx;
                    }
                }
                if (ok) {
                    ldvec<T>(xin + row * p.ldx + c0, rx[v], p.cin - c0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int c = min(c0 + j, p.cin - 1);
                        float z = rx[v][j];
                        if (p.pro.scale) z *= p.pro.scale[c];
                        if (p.pro.shift) z += p.pro.shift[c];
                        z = act_t<ACT>(z, p.pro.act);
                        if (p.pro.bscale) z *= p.pro.bscale[pb * p.cin + c];
                        rx[v][j] = (c0 + j < p.cin) ? z : 0.f;
                    }
                }
            }
        }
    };
    auto stash = [&]() {          // registers -> wave-private LDS slab
#pragma unroll
        for (int v = 0; v < LOADS_N; ++v) {
            const int idx = v * 64 + lane, pix = idx / VN, col = (idx % VN) * 8;
            f32x4 a, b;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = rd[v][j]; b[j] = rd[v][4 + j]; }
            *reinterpret_cast<f32x4*>(sD + pix * LDN + col) = a;
            *reinterpret_cast<f32x4*>(sD + pix * LDN + col + 4) = b;
        }
#pragma unroll
        for (int v = 0; v < LOADS_K; ++v) {
            const int idx = v * 64 + lane, pix = idx / VK, col = (idx % VK) * 8;
            f32x4 a, b;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = rx[v][j]; b[j] = rx[v][4 + j]; }
            *reinterpret_cast<f32x4*>(sX + pix * LDK + col) = a;
            *reinterpret_cast<f32x4*>(sX + pix * LDK + col + 4) = b;
        }
    };

    const long stride = (long)gridDim.x * 4;
    long chunk = (long)blockIdx.x * 4 + wave;
    if (chunk < p.nchunks) fetch(chunk);
    for (; chunk < p.nchunks; chunk += stride) {
        stash();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (chunk + stride < p.nchunks) fetch(chunk + stride);          // overlap with the MFMAs below
#pragma unroll
        for (int s = 0; s < PM / 2; ++s) {
            const int row = 2 * s + hh;
            float a[TN], b[TK];
#pragma unroll
            for (int i = 0; i < TN; ++i) { a[i] = sD[row * LDN + i * 32 + r]; dbs[i] += a[i]; }
#pragma unroll
            for (int j = 0; j < TK; ++j) b[j] = sX[row * LDK + j * 32 + r];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- cross-wave reduction in LDS, then one atomic set per workgroup -----------------------
    __syncthreads();
    float* red = lds;                                   // [TN*TK*16*64] + [TN*32] (fits the 4 slabs)
    constexpr int ACC_FLOATS = TN * TK * 16 * 64;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float* q = red + ((i * TK + j) * 16 + e) * 64 + lane;
                        *q = (w == 0 ? 0.f : *q) + acc[i][j][e];
                    }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const float s = dbs[i] + __shfl_xor(dbs[i], 32, 64);
                if (hh == 0) { float* q = red + ACC_FLOATS + i * 32 + r; *q = (w == 0 ? 0.f : *q) + s; }
            }
        }
        __syncthreads();
    }
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int kd = k0 + j * 32 + r;
            if (kd >= p.cin) continue;
            const int k = p.kmap ? p.kmap[kd] : kd;
            if (k < 0) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (n >= p.N) continue;
                const float v = red[((i * TK + j) * 16 + e) * 64 + lane];
                long off;
                if (p.out_mode == ISA_OUT_SHUFFLE2) off = ((long)k * p.N + n) * 4 + tap;   // [K][Co][2][2]
                else off = ((long)n * p.ksrc + k) * p.taps + tap;                          // [N][K][kh][kw]
                atomicAdd(p.dw + off, v);
            }
        }
    if (p.dbias && gk == 0 && tap == 0 && hh == 0) {
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + i * 32 + r;
            if (n < p.N) atomicAdd(p.dbias + n, red[ACC_FLOATS + i * 32 + r]);
        }
    }
}

template <typename T, int TN, int TK>
int launch_wg(WgParams& p, int groups_n, hipStream_t s) {
    constexpr int LDN = TN * 32 + 4, LDK = TK * 32 + 4;
    const size_t slab = (size_t)PM * (LDN + LDK) * 4 * 4;
    const size_t redb = ((size_t)TN * TK * 16 * 64 + TN * 32) * 4;
    const size_t lds = slab > redb ? slab : redb;
    const int gy = groups_n * p.groups_k;
    long want = (p.nchunks + 3) / 4;
    long cap = (256L * 2) / ((long)gy * p.taps);
    if (cap < 1) cap = 1;
    const int gx = (int)(want < cap ? want : cap);
    dim3 grid(gx, gy, p.taps);
    if (p.pro.act == ISA_ACT_RELU6) hipLaunchKernelGGL((conv_wgrad_kernel<T, TN, TK, ISA_ACT_RELU6>), grid, dim3(256), lds, s, p);
    else if (p.pro.act == ISA_ACT_NONE) hipLaunchKernelGGL((conv_wgrad_kernel<T, TN, TK, ISA_ACT_NONE>), grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<T, TN, TK, ACT_RT>), grid, dim3(256), lds, s, p);
    return launch_status();
}

template <typename T>
int dispatch_wg(WgParams& p, hipStream_t s) {
    const int nt = (p.N + 31) / 32, kt = (p.cin + 31) / 32;
    int tn, tk;                         // tiles per wave: up to 4 accumulators (64 VGPRs)
    if (nt == 1) { tn = 1; tk = kt >= 4 ? 4 : (kt >= 2 ? 2 : 1); }
    else if (kt == 1) { tk = 1; tn = nt >= 4 ? 4 : (nt >= 2 ? 2 : 1); }
    else { tn = 2; tk = 2; }
    const int groups_n = (nt + tn - 1) / tn;
    p.groups_k = (kt + tk - 1) / tk;
    if (tn == 1 && tk == 1) return launch_wg<T, 1, 1>(p, groups_n, s);
    if (tn == 1 && tk == 2) return launch_wg<T, 1, 2>(p, groups_n, s);
    if (tn == 2 && tk == 1) return launch_wg<T, 2, 1>(p, groups_n, s);
    if (tn == 2 && tk == 2) return launch_wg<T, 2, 2>(p, groups_n, s);
    if (tn == 1 && tk == 4) return launch_wg<T, 1, 4>(p, groups_n, s);
    if (tn == 4 && tk == 1) return launch_wg<T, 4, 1>(p, groups_n, s);
    return ISA_EINVAL;
}

}  // namespace

extern "C" int isa_conv_wgrad(const isa_tensor* x, const isa_pro* pro, const isa_tensor* dy,
                              float* dw, float* dbias, int32_t in_mode, int32_t out_mode,
                              const int32_t* kmap, int32_t ksrc, void* stream) {
    if (!tensor_ok(x, 8) || !tensor_ok(dy, 8) || !dw || x->dtype != dy->dtype) return ISA_EINVAL;
    if (in_mode == ISA_IN_GATHER2) return ISA_EINVAL;
    WgParams p{};
    p.x = x->data; p.xh = x->h; p.xw = x->w; p.cin = x->c; p.ldx = x->ld;
    p.dy = dy->data; p.dh = dy->h; p.dw_ = dy->w; p.ldd = dy->ld;
    p.mh = x->h; p.mw = x->w; p.M = (long)x->n * x->h * x->w;
    p.pro = make_pro(pro); p.dw = dw; p.dbias = dbias; p.kmap = kmap; p.ksrc = ksrc > 0 ? ksrc : x->c;
    p.in_mode = in_mode; p.out_mode = out_mode;
    if (out_mode == ISA_OUT_SHUFFLE2) {
        if (in_mode != ISA_IN_1X1 || dy->h != 2 * x->h || dy->w != 2 * x->w || dy->n != x->n) return ISA_EINVAL;
        p.taps = 4; p.N = dy->c;
        if (dbias) return ISA_EINVAL;       // bias of a transposed conv sums all quadrants: isa_colsum
    } else {
        if (dy->h != x->h || dy->w != x->w || dy->n != x->n) return ISA_EINVAL;
        p.taps = in_mode == ISA_IN_3X3 ? 9 : 1; p.N = dy->c;
    }
    p.nchunks = (p.M + PM - 1) / PM;
    if (x->dtype == ISA_BF16) return dispatch_wg<bf16_t>(p, as_stream(stream));
    return dispatch_wg<float>(p, as_stream(stream));
}

// column sums of an NHWC view: out[c] += sum over pixels (bias gradients of transposed convs, heads)
namespace {
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, long pixels, int c, int ld, float* out) {
    extern __shared__ float red[];
    for (int i = threadIdx.x; i < c; i += 256) red[i] = 0.f;
    __syncthreads();
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
