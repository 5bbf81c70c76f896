// conv_gemm: y[M,N] (+)= pro(x)[M,K] * W^T (+bias), M = pixels, on MFMA 32x32 tiles.
//
// Replaces the reference's nn.Conv2d(k=1), nn.Conv2d(k=3,p=1) and nn.ConvTranspose2d(k=2,s=2)
// (MobileNetDenseASPP.py:68-123, utils.py:703-707, unet_parts.py:73, utils.py:975) and their
// data gradients.  Design for gfx950:
//   * HBM-bound: K and N are tiny (21..1024) while M is 4K..1M pixels, so the A operand (pixels)
//     streams from HBM exactly once, straight into MFMA operand registers: lane (r,h) of a wave
//     owns pixel r of the wave's 32-pixel tile and reads 16 *contiguous* channels per K32 group
//     (the contraction index is permuted identically in A and B, so fragments are contiguous 32 B
//     (bf16) / 64 B (f32) NHWC row pieces and need no LDS transpose).
//   * weights (B operand) are staged once per workgroup in LDS and reused by every pixel tile of
//     a persistent, grid-strided loop; rows padded by 16 B => conflict-free ds_read_b128.
//   * the lazy-input prologue (BN scale/shift + ReLU6/... + per-image channel gate) is applied in
//     registers between the global load and the MFMA: train-mode BN costs no extra HBM pass.
//   * epilogue: bias, per-channel sum/sumsq for the *next* BN (lane == channel in the 32x32 C/D
//     layout, so this is 16 adds per lane + one cross-half shuffle), transpose through a 4 KB
//     per-wave LDS tile, 16 B coalesced row stores (optionally read-modify-write accumulate,
//     optionally pixel-shuffled for ConvTranspose2d).
//   * bf16 storage: v_mfma_f32_32x32x16_bf16, fp32 accumulate.  f32 storage: v_mfma_f32_32x32x2_f32
//     (exact fp32 FMA chain) — used for the 1e-3 parity mode.
#include "common.hpp"

int conv3x3_tiled_launch(const isa_tensor* x, const void* w, const float* bias, const isa_tensor* y, int accumulate,
                         hipStream_t s);          // conv3x3_tiled.hip: narrow dense 3x3 convs from an LDS halo tile

namespace {

struct GemmParams {
    const void* x; int xh, xw, cin, ldx;       // input image dims and channels per tap
    int mh, mw;                                 // M-grid dims (rows of the GEMM are (b,y,x) here)
    long M;
    ProDev pro;
    const void* w; int kp, taps, total_groups, groups_per_chunk, ldb;
    const float* bias;
    void* y; int oh, ow, N, ldy, cout;          // cout: channels per shuffle quadrant
    float* stats; int accumulate;
    int ntiles;
};

template <typename T> struct Frag;      // 16 contiguous channels of one pixel
template <> struct Frag<bf16_t> { bf16x8 v[2]; };
template <> struct Frag<float> { f32x4 v[4]; };

template <typename T>
__device__ __forceinline__ void frag_zero(Frag<T>& f) {
    if constexpr (sizeof(T) == 2) { f.v[0] = bf16x8{0}; f.v[1] = bf16x8{0}; }
    else { for (int i = 0; i < 4; ++i) f.v[i] = f32x4{0}; }
}

// load 16 channels starting at p (k .. k+15), guarding k+j < cin
template <typename T>
__device__ __forceinline__ void frag_load(Frag<T>& f, const T* p, int k, int cin) {
    constexpr int V = sizeof(T) == 2 ? 8 : 4;       // elements per 16-byte vector
    constexpr int NV = 16 / V;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int kk = k + i * V;
        if (kk + V <= cin) {
            if constexpr (sizeof(T) == 2) f.v[i] = *reinterpret_cast<const bf16x8*>(p + i * V);
            else f.v[i] = *reinterpret_cast<const f32x4*>(p + i * V);
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                T val = (kk + j < cin) ? p[i * V + j] : (T)0.f;
                f.v[i][j] = val;
            }
        }
    }
}

template <typename T>
__device__ __forceinline__ float frag_get(const Frag<T>& f, int j) {
    constexpr int V = sizeof(T) == 2 ? 8 : 4;
    return (float)f.v[j / V][j % V];
}
template <typename T>
__device__ __forceinline__ void frag_set(Frag<T>& f, int j, float x) {
    constexpr int V = sizeof(T) == 2 ? 8 : 4;
    f.v[j / V][j % V] = (T)x;
}

template <typename T, int NT, int IN_MODE, int OUT_MODE, int PRO>
__global__ __launch_bounds__(256) void conv_gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int N_BLK = 32 * NT;
    constexpr bool HAS_PRO = PRO != 0;
    constexpr int ACT = PRO == 1 ? ISA_ACT_RELU6 : (PRO == 3 ? ISA_ACT_LEAKY : ACT_RT);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.y * N_BLK;
    // LDS carve: [B tile][pro table][4 x stage]
    T* ldsB = reinterpret_cast<T*>(smem);
    const int b_bytes = (N_BLK * p.ldb * (int)sizeof(T) + 15) & ~15;
    float* pro_tab = reinterpret_cast<float*>(smem + b_bytes);          // [2][kp]
    const int pro_bytes = HAS_PRO ? 2 * p.kp * 4 : 0;
    float* stage = reinterpret_cast<float*>(smem + b_bytes + pro_bytes) + wave * (32 * 33);

    if constexpr (HAS_PRO) {
        for (int k = tid; k < p.kp; k += 256) {
            float sc = 1.f, sh = 0.f;
            if (k < p.cin) {
                if (p.pro.scale) sc = p.pro.scale[k];
                if (p.pro.shift) sh = p.pro.shift[k];
            }
            pro_tab[k] = sc; pro_tab[p.kp + k] = sh;
        }
    }
    const int gpt = p.kp / 32;                         // K32 groups per tap
    const int nchunks = (p.total_groups + p.groups_per_chunk - 1) / p.groups_per_chunk;
    const bool resident = nchunks == 1;
    bool loaded = false;
    float st_sum[NT], st_sq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { st_sum[t] = 0.f; st_sq[t] = 0.f; }
    const T* xin = reinterpret_cast<const T*>(p.x);
    const T* wg = reinterpret_cast<const T*>(p.w);
    const long wrow = (long)p.taps * p.kp;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const long m = (long)tile * 128 + wave * 32 + r;
        const bool mvalid = m < p.M;
        int pb = 0, py = 0, px = 0;
        constexpr bool NEED_YX = IN_MODE != ISA_IN_1X1;
        if (mvalid && (NEED_YX || (HAS_PRO && p.pro.bscale))) {     // 32-bit: M < 2^31 checked on the host
            const unsigned mu = (unsigned)m, q = mu / (unsigned)p.mw;
            px = (int)(mu - q * (unsigned)p.mw); pb = (int)(q / (unsigned)p.mh); py = (int)(q - (unsigned)pb * (unsigned)p.mh);
        }
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};

        // address of this lane's A fragment for K32 group g (nullptr => zero contribution)
        auto a_ptr = [&](int g, int& kout) -> const T* {
            const int tap = g / gpt;
            const int k = (g - tap * gpt) * 32 + 16 * hh;
            kout = k;
            if (!mvalid || k >= p.cin) return nullptr;
            int sy = py, sx = px;
            if constexpr (IN_MODE == ISA_IN_3X3) {
                sy = py + tap / 3 - 1; sx = px + tap % 3 - 1;
                if (sy < 0 || sy >= p.xh || sx < 0 || sx >= p.xw) return nullptr;
            } else if constexpr (IN_MODE == ISA_IN_GATHER2) {
                sy = 2 * py + (tap >> 1); sx = 2 * px + (tap & 1);
            }
            if constexpr (IN_MODE == ISA_IN_1X1) return xin + m * p.ldx + k;
            return xin + (((long)pb * p.xh + sy) * p.xw + sx) * p.ldx + k;
        };

        Frag<T> cur, nxt;
        int kcur = 0, knxt = 0;
        bool vcur, vnxt = false;
        {
            const T* ap = a_ptr(0, kcur);
            vcur = ap != nullptr;
            if (vcur) frag_load<T>(cur, ap, kcur, p.cin); else frag_zero<T>(cur);
        }
        for (int c = 0; c < nchunks; ++c) {
            const int g_begin = c * p.groups_per_chunk;
            const int g_end = min(p.total_groups, g_begin + p.groups_per_chunk);
            if (!(resident && loaded)) {
                __syncthreads();
                // stage W[n0 .. n0+N_BLK) x groups [g_begin, g_end) into LDS, zero rows >= N
                constexpr int V = 16 / (int)sizeof(T);
                const int len = (g_end - g_begin) * 32;
                const int vec_per_row = len / V;
                for (int i = tid; i < N_BLK * vec_per_row; i += 256) {
                    const int row = i / vec_per_row, col = (i - row * vec_per_row) * V;
                    f32x4 val = f32x4{0};
                    if (n0 + row < p.N)
                        val = *reinterpret_cast<const f32x4*>(wg + (long)(n0 + row) * wrow + g_begin * 32 + col);
                    *reinterpret_cast<f32x4*>(ldsB + row * p.ldb + col) = val;
                }
                __syncthreads();
                loaded = true;
            }
            for (int g = g_begin; g < g_end; ++g) {
                if (g + 1 < p.total_groups) {
                    const T* ap = a_ptr(g + 1, knxt);
                    vnxt = ap != nullptr;
                    if (vnxt) frag_load<T>(nxt, ap, knxt, p.cin); else frag_zero<T>(nxt);
                }
                if constexpr (HAS_PRO) {
                    if (vcur) {
                        // scale / shift of the fragment's 16 channels as eight ds_read_b128.  Channels >= cin arrive as 0
                        // from frag_load and are multiplied by zero weight columns, so only finiteness matters there and
                        // there is no per-element bounds test; the per-image multiplier is a wave-uniform branch around
                        // its own loop.  The element-wise form compiled to 32 ds_read_b32, 16 waits and 16 branches per
                        // fragment.
                        float sc[16], sh[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 a = *reinterpret_cast<const f32x4*>(pro_tab + kcur + 4 * q);
                            const f32x4 b = *reinterpret_cast<const f32x4*>(pro_tab + p.kp + kcur + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { sc[4 * q + e] = a[e]; sh[4 * q + e] = b[e]; }
                        }
                        float v[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) v[j] = act_t<ACT>(fmaf(frag_get<T>(cur, j), sc[j], sh[j]), p.pro.act);
                        if (p.pro.bscale) {
#pragma unroll
                            for (int j = 0; j < 16; ++j) v[j] *= (kcur + j < p.cin) ? p.pro.bscale[(long)pb * p.cin + kcur + j] : 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < 16; ++j) frag_set<T>(cur, j, v[j]);
                    }
                }
                const int gl = g - g_begin;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const T* brow = ldsB + (t * 32 + r) * p.ldb + gl * 32 + 16 * hh;
                    if constexpr (sizeof(T) == 2) {
                        bf16x8 b0 = *reinterpret_cast<const bf16x8*>(brow);
                        bf16x8 b1 = *reinterpret_cast<const bf16x8*>(brow + 8);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.v[0], b0, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.v[1], b1, acc[t], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 bq = *reinterpret_cast<const f32x4*>(brow + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.v[q][e], bq[e], acc[t], 0, 0, 0);
                        }
                    }
                }
                cur = nxt; kcur = knxt; vcur = vnxt;
            }
        }

        // ---------------- epilogue: lane r <-> output channel n0+t*32+r -------------------------
        const long mbase = (long)tile * 128 + wave * 32;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + t * 32 + r;
            // bias is per output channel; in pixel-shuffle mode GEMM column n = quadrant*cout + co
            const float bv = (p.bias && n < p.N) ? p.bias[OUT_MODE == ISA_OUT_SHUFFLE2 ? n % p.cout : n] : 0.f;
            float s = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                float v = acc[t][i] + bv;
                if (mbase + row < p.M) { s += v; s2 += v * v; }
                stage[row * 33 + r] = v;
            }
            if (p.stats) {
                s += __shfl_xor(s, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                st_sum[t] += s; st_sq[t] += s2;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // row-wise stores: lane -> (row = lane>>1, 16 channels at (lane&1)*16)
            const int row = lane >> 1, cseg = (lane & 1) * 16;
            const long mr = mbase + row;
            const int nseg = n0 + t * 32 + cseg;
            if (mr < p.M && nseg < p.N) {
                T* dst;
                if constexpr (OUT_MODE == ISA_OUT_SHUFFLE2) {
                    const unsigned mu = (unsigned)mr, q = mu / (unsigned)p.mw;
                    const int ox = (int)(mu - q * (unsigned)p.mw);
                    const int ob = (int)(q / (unsigned)p.mh); const int oy = (int)(q - (unsigned)ob * (unsigned)p.mh);
                    const int quad = nseg / p.cout, co = nseg - quad * p.cout;
                    dst = reinterpret_cast<T*>(p.y) +
                          (((long)ob * p.oh + 2 * oy + (quad >> 1)) * p.ow + 2 * ox + (quad & 1)) * p.ldy + co;
                } else {
                    dst = reinterpret_cast<T*>(p.y) + mr * p.ldy + nseg;
                }
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = stage[row * 33 + cseg + j];
                const bool full = (nseg + 16 <= p.N) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
                if (full) {
                    float lo[8], hi[8];
                    if (p.accumulate) {
                        load8<T>(dst, lo); load8<T>(dst + 8, hi);
#pragma unroll
                        for (int j = 0; j < 8; ++j) { lo[j] += v[j]; hi[j] += v[8 + j]; }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { lo[j] = v[j]; hi[j] = v[8 + j]; }
                    }
                    store8<T>(dst, lo); store8<T>(dst + 8, hi);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (nseg + j < p.N) {
                            float o = v[j];
                            if (p.accumulate) o += st<T>::ld(dst + j);
                            st<T>::stv(dst + j, o);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (p.stats) {
        // the four waves of the workgroup are folded in LDS first: one global atomic per channel and workgroup instead
        // of four (the replicated counters still see gridDim.x / 8 adders each, serialised at the L2)
        __shared__ float sred[2 * 128];
        if (tid < 2 * N_BLK) sred[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (hh == 0) { atomicAdd(&sred[t * 32 + r], st_sum[t]); atomicAdd(&sred[N_BLK + t * 32 + r], st_sq[t]); }
        }
        __syncthreads();
        if (tid < 2 * N_BLK) {
            const int which = tid / N_BLK, n = n0 + (tid - which * N_BLK);
            if (n < p.N) {
                float* rep = p.stats + ((blockIdx.x + blockIdx.y) & (ISA_STAT_R - 1)) * 2 * p.N;
                atomicAdd(rep + which * p.N + n, sred[tid]);
            }
        }
    }
}

template <typename T, int NT, int IN_MODE, int OUT_MODE>
int launch2(const GemmParams& p, bool has_pro, dim3 grid, size_t lds, hipStream_t s) {
    if (has_pro && p.pro.act == ISA_ACT_RELU6) hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 1>), grid, dim3(256), lds, s, p);
    else if (has_pro && p.pro.act == ISA_ACT_LEAKY && IN_MODE == ISA_IN_3X3) hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 3>), grid, dim3(256), lds, s, p);
    else if (has_pro) hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 2>), grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 0>), grid, dim3(256), lds, s, p);
    return launch_status();
}

template <typename T, int NT>
int launch1(const GemmParams& p, bool has_pro, int in_mode, int out_mode, dim3 grid, size_t lds, hipStream_t s) {
    if (out_mode == ISA_OUT_SHUFFLE2) {
        if (in_mode != ISA_IN_1X1) return ISA_EINVAL;
        return launch2<T, NT, ISA_IN_1X1, ISA_OUT_SHUFFLE2>(p, has_pro, grid, lds, s);
    }
    switch (in_mode) {
        case ISA_IN_1X1: return launch2<T, NT, ISA_IN_1X1, ISA_OUT_PLAIN>(p, has_pro, grid, lds, s);
        case ISA_IN_3X3: return launch2<T, NT, ISA_IN_3X3, ISA_OUT_PLAIN>(p, has_pro, grid, lds, s);
        case ISA_IN_GATHER2: return launch2<T, NT, ISA_IN_GATHER2, ISA_OUT_PLAIN>(p, has_pro, grid, lds, s);
    }
    return ISA_EINVAL;
}

template <typename T>
int launch0(GemmParams& p, bool has_pro, int in_mode, int out_mode, hipStream_t s) {
    int nt = p.N <= 32 ? 1 : (p.N <= 64 ? 2 : 4);
    // low-resolution levels: few 128-pixel tiles but hundreds of output channels.  Narrower column tiles put more
    // workgroups on the chip (the re-read A operand is L2-resident at these sizes): 4096 px x 512 ch ran as 128
    // workgroups on 256 CUs.
    const long tiles_m = (p.M + 127) / 128;
    while (nt > 1 && tiles_m * ((p.N + 32 * nt - 1) / (32 * nt)) < 512) nt /= 2;
    const int n_blk = 32 * nt;
    // weight chunk sized to <= 48 KB of LDS
    int gpc = (48 * 1024) / (n_blk * 32 * (int)sizeof(T));
    if (gpc < 1) gpc = 1;
    if (gpc > p.total_groups) gpc = p.total_groups;
    p.groups_per_chunk = gpc;
    p.ldb = gpc * 32 + 16 / (int)sizeof(T);
    const size_t b_bytes = ((size_t)n_blk * p.ldb * sizeof(T) + 15) & ~(size_t)15;
    const size_t lds = b_bytes + (has_pro ? 2 * (size_t)p.kp * 4 : 0) + 4 * 32 * 33 * 4;
    p.ntiles = (int)((p.M + 127) / 128);
    const int gy = (p.N + n_blk - 1) / n_blk;
    int gx = p.ntiles;
    const int cap = max(1, (256 * 3) / gy);          // three resident workgroups per CU (measured: 2 -> 36.5 ms, 3 -> 36.3, 4 -> 36.9)
    if (gx > cap) gx = cap;
    dim3 grid(gx, gy);
    switch (nt) {
        case 1: return launch1<T, 1>(p, has_pro, in_mode, out_mode, grid, lds, s);
        case 2: return launch1<T, 2>(p, has_pro, in_mode, out_mode, grid, lds, s);
        default: return launch1<T, 4>(p, has_pro, in_mode, out_mode, grid, lds, s);
    }
}

}  // namespace

extern "C" int isa_conv_gemm(const isa_tensor* x, const isa_pro* pro, const void* w, int32_t kp,
                             const float* bias, const isa_tensor* y, int32_t in_mode,
                             int32_t out_mode, float* stats, int32_t accumulate, void* stream) {
    if (!tensor_ok(x, 8) || !tensor_ok(y, 1) || !w || x->dtype != y->dtype) return ISA_EINVAL;
    if (kp <= 0 || kp % 32 || kp < x->c) return ISA_EINVAL;
    if (y->ld % 8) return ISA_EALIGN;
    GemmParams p{};
    p.x = x->data; p.xh = x->h; p.xw = x->w; p.cin = x->c; p.ldx = x->ld;
    p.pro = make_pro(pro);
    p.w = w; p.kp = kp;
    p.bias = bias;
    p.y = y->data; p.oh = y->h; p.ow = y->w; p.ldy = y->ld;
    p.stats = stats; p.accumulate = accumulate;
    p.taps = in_mode == ISA_IN_3X3 ? 9 : (in_mode == ISA_IN_GATHER2 ? 4 : 1);
    p.total_groups = p.taps * (kp / 32);
    if (in_mode == ISA_IN_GATHER2) {
        if (x->h != 2 * y->h || x->w != 2 * y->w || x->n != y->n) return ISA_EINVAL;
        p.mh = y->h; p.mw = y->w;
    } else {
        p.mh = x->h; p.mw = x->w;
    }
    if (out_mode == ISA_OUT_SHUFFLE2) {
        if (y->h != 2 * x->h || y->w != 2 * x->w || y->n != x->n || y->c % 16 || stats) return ISA_EINVAL;
        p.N = 4 * y->c; p.cout = y->c;
    } else {
        if (y->h != p.mh || y->w != p.mw || y->n != x->n) return ISA_EINVAL;
        p.N = y->c; p.cout = y->c;
    }
    p.M = (long)x->n * p.mh * p.mw;
    if (p.M >= (1L << 31)) return ISA_EINVAL;
    const bool has_pro = !pro_trivial(p.pro);
    if (in_mode == ISA_IN_3X3 && out_mode == ISA_OUT_PLAIN && x->dtype == ISA_BF16 && !has_pro && !stats && kp == 32 &&
        y->c <= 32)
        return conv3x3_tiled_launch(x, w, bias, y, accumulate, as_stream(stream));
    if (x->dtype == ISA_BF16) return launch0<bf16_t>(p, has_pro, in_mode, out_mode, as_stream(stream));
    return launch0<float>(p, has_pro, in_mode, out_mode, as_stream(stream));
}
