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
#include <cstdlib>

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
    // optional output epilogue (isa_conv_gemm_ep): y = act(ep_scale[n] * (conv + bias) + ep_shift[n]) + res
    const float *ep_scale, *ep_shift; int ep_act; const void* res; int ldres;
    int G;                                      // statistic groups (1x1 plain only): M, ntiles are per group (common.hpp)
    FinDev fin;                                 // pending BatchNorm finalize of the lazy input (isa_pro.fin), or stats == NULL
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

// EP: the output epilogue of isa_conv_gemm_ep is its own instantiation - folded into the common one it cost the training
// launches 4 % (registers and epilogue code on every launch).
// FAST (1x1, cin == kp): every A-fragment load is unconditional - rows >= M read the last valid row (the epilogue masks
// them), no channel guards - so the K loop is straight-line code and the compiler's s_waitcnt pass can count: with the
// guarded, lane-divergent loads of the general path it falls back to vmcnt(0) ahead of every MFMA and any further
// lookahead (including the next tile's first fragment, requested here before the epilogue) is drained at once.
template <typename T, int NT, int IN_MODE, int OUT_MODE, int PRO, bool EP = false, bool FAST = false>
__global__ __launch_bounds__(256) void conv_gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int N_BLK = 32 * NT;
    constexpr bool HAS_PRO = PRO != 0;
    constexpr int ACT = PRO == 1 ? ISA_ACT_RELU6 : (PRO == 3 ? ISA_ACT_LEAKY : ACT_RT);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.y * N_BLK;
    const GroupSel gs = group_sel(p.G);
    if (gs.g) {                                 // this workgroup's statistic group: rows [g*M, (g+1)*M) and its constants
        p.x = reinterpret_cast<const T*>(p.x) + (long)gs.g * p.M * p.ldx;
        p.y = reinterpret_cast<T*>(p.y) + (long)gs.g * p.M * p.ldy;
        p.pro.scale = goff(p.pro.scale, (long)gs.g * p.cin); p.pro.shift = goff(p.pro.shift, (long)gs.g * p.cin);
        p.pro.bscale = goff(p.pro.bscale, (long)gs.g * (p.M / ((long)p.mh * p.mw)) * p.cin);
        p.stats = goff(p.stats, (long)gs.g * ISA_STAT_R * 2 * p.N);
    }
    // LDS carve: [B tile][pro table][4 x stage]
    T* ldsB = reinterpret_cast<T*>(smem);
    const int b_bytes = (N_BLK * p.ldb * (int)sizeof(T) + 15) & ~15;
    float* pro_tab = reinterpret_cast<float*>(smem + b_bytes);          // [2][kp]
    const int pro_bytes = HAS_PRO ? 2 * p.kp * 4 : 0;
    float* stage = reinterpret_cast<float*>(smem + b_bytes + pro_bytes) + wave * (32 * (NT == 2 ? 65 : 33));

    // A pending finalize of the input's BatchNorm (isa_pro.fin, train mode) runs in this kernel: this group's scale / shift
    // go straight into the table.  One dword of the cache line of this lane's first fragment is requested ahead of it, so
    // the round trip to the input overlaps the one to the statistics instead of following it (running the finalize inside
    // the tile loop, behind the real fragment load, costs these kernels an occupancy step).
    bool fin_live = false;
    if constexpr (HAS_PRO && !EP) {
        if (p.fin.stats != nullptr) {
            long m = (long)gs.bx * 128 + wave * 32 + r;
            if (m > p.M - 1) m = p.M - 1;
            const T* touch = reinterpret_cast<const T*>(p.x) + (IN_MODE == ISA_IN_GATHER2 ? 0 : m * p.ldx + 16 * hh);
            const unsigned probe = *reinterpret_cast<const unsigned*>(touch);
            asm volatile("" ::: "memory");                        // keep the request ahead of the finalize
            bn_fin_inline<256>(p.fin, p.cin, p.G, gs.g, pro_tab, p.kp, tid);
            asm volatile("" :: "v"(probe));
            fin_live = true;
        }
    }
    if constexpr (HAS_PRO) {
        const bool fin = fin_live;
        for (int k = tid; k < p.kp; k += 256) {
            if (fin && k < p.cin) continue;
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

    Frag<T> pre;                                                 // FAST: first fragment of the next tile, requested before the epilogue
    bool have_pre = false;
    for (int tile = gs.bx; tile < p.ntiles; tile += gs.nbx) {
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

        if constexpr (FAST && IN_MODE == ISA_IN_1X1) {
            const long mrow = mvalid ? m : p.M - 1;
            const T* ab = xin + mrow * p.ldx + 16 * hh;
            auto fload = [&](Frag<T>& f, const T* base, int g) {
                if constexpr (sizeof(T) == 2) {
                    f.v[0] = *reinterpret_cast<const bf16x8*>(base + g * 32);
                    f.v[1] = *reinterpret_cast<const bf16x8*>(base + g * 32 + 8);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) f.v[q] = *reinterpret_cast<const f32x4*>(base + g * 32 + 4 * q);
                }
                asm volatile("" ::: "memory");                     // keep the request here (LLVM sinks loads to their use)
            };
            auto consume = [&](Frag<T>& f, int g, int gl) {
                const int kq = g * 32 + 16 * hh;
                if constexpr (HAS_PRO) {
                    float sc[16], sh[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(pro_tab + kq + 4 * q);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(pro_tab + p.kp + kq + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { sc[4 * q + e] = a[e]; sh[4 * q + e] = b[e]; }
                    }
                    float v[16];
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) v[jj] = act_t<ACT>(fmaf(frag_get<T>(f, jj), sc[jj], sh[jj]), p.pro.act);
                    if (p.pro.bscale) {
#pragma unroll
                        for (int jj = 0; jj < 16; ++jj) v[jj] *= p.pro.bscale[(long)pb * p.cin + kq + jj];
                    }
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) frag_set<T>(f, jj, v[jj]);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const T* brow = ldsB + (t * 32 + r) * p.ldb + gl * 32 + 16 * hh;
                    if constexpr (sizeof(T) == 2) {
                        bf16x8 b0 = *reinterpret_cast<const bf16x8*>(brow);
                        bf16x8 b1 = *reinterpret_cast<const bf16x8*>(brow + 8);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.v[0], b0, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.v[1], b1, acc[t], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 bq = *reinterpret_cast<const f32x4*>(brow + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.v[q][e], bq[e], acc[t], 0, 0, 0);
                        }
                    }
                }
            };
            Frag<T> cur, nxt;
            if (have_pre) cur = pre; else fload(cur, ab, 0);
            int g_begin = 0, g_end = 0;
            for (int g = 0; g < p.total_groups; ++g) {
                if (g == g_end) {                                  // next weight chunk (one chunk and one staging when resident)
                    g_begin = g_end;
                    g_end = min(p.total_groups, g_begin + p.groups_per_chunk);
                    if (!(resident && loaded)) {
                        __syncthreads();
                        constexpr int V = 16 / (int)sizeof(T);
                        const int len = (g_end - g_begin) * 32;
                        const int vec_per_row = len / V;
                        for (int i2 = tid; i2 < N_BLK * vec_per_row; i2 += 256) {
                            const int row = i2 / vec_per_row, col = (i2 - row * vec_per_row) * V;
                            f32x4 val = f32x4{0};
                            if (n0 + row < p.N)
                                val = *reinterpret_cast<const f32x4*>(wg + (long)(n0 + row) * wrow + g_begin * 32 + col);
                            *reinterpret_cast<f32x4*>(ldsB + row * p.ldb + col) = val;
                        }
                        __syncthreads();
                        loaded = true;
                    }
                }
                if (g + 1 < p.total_groups) {
                    fload(nxt, ab, g + 1);
                    consume(cur, g, g - g_begin);
                    cur = nxt;
                } else {
                    consume(cur, g, g - g_begin);
                }
            }
            const int tn = tile + gs.nbx;
            have_pre = tn < p.ntiles;
            if (have_pre) {
                long mn = (long)tn * 128 + wave * 32 + r;
                if (mn > p.M - 1) mn = p.M - 1;
                fload(pre, xin + mn * p.ldx + 16 * hh, 0);
            }
        } else {
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

        }

        // ---------------- epilogue: lane r <-> output channel n0+t*32+r -------------------------
        // NTS column tiles are transposed through the per-wave LDS stage together and leave as whole rows: with two tiles
        // (64 output channels) a row is one 128-byte line in bf16 - written half a line at a time (one tile after the
        // other) the same kernel lost a quarter of its bandwidth (profiles/r03_access_pattern_probe.txt: read 32 + write 64
        // 3.4 vs 5.8 TB/s).  Four-tile launches (low-resolution levels, column blocks of 128) keep one tile per pass.
        constexpr int NTS = NT == 2 ? 2 : 1;                   // tiles staged together
        constexpr int SR = 32 * NTS + 1;                       // stage row stride (floats)
        constexpr int LPR = 4 * NTS;                           // lanes per row in the store phase (8 channels each)
        const long mbase = (long)tile * 128 + wave * 32;
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += NTS) {
#pragma unroll
            for (int tt = 0; tt < NTS; ++tt) {
                const int t = t0 + tt;
                const int n = n0 + t * 32 + r;
                // bias is per output channel; in pixel-shuffle mode GEMM column n = quadrant*cout + co
                const float bv = (p.bias && n < p.N) ? p.bias[OUT_MODE == ISA_OUT_SHUFFLE2 ? n % p.cout : n] : 0.f;
                constexpr bool has_ep = EP;
                const float es = (has_ep && n < p.N) ? p.ep_scale[n] : 1.f, eh = (has_ep && n < p.N) ? p.ep_shift[n] : 0.f;
                float s = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                    float v = acc[t][i] + bv;
                    if (has_ep) v = act_apply(fmaf(v, es, eh), p.ep_act);
                    if (mbase + row < p.M) { s += v; s2 += v * v; }
                    stage[row * SR + tt * 32 + r] = v;
                }
                if (p.stats) {
                    s += __shfl_xor(s, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                    st_sum[t] += s; st_sq[t] += s2;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // row-wise stores: lane -> (row = lane / LPR, 8 channels at (lane % LPR) * 8): the LPR lanes of a row write its
            // 32 * NTS channels as ONE contiguous piece per store instruction.  (The first mapping - two lanes per row, two
            // 16-byte stores each - put non-adjacent 16-byte pieces of every row into each instruction: a pure memory kernel
            // with that store pattern runs 14-42 % below the contiguous one.)
#pragma unroll
            for (int pass = 0; pass < LPR / 2; ++pass) {
                const int row = lane / LPR + (64 / LPR) * pass, cseg = (lane % LPR) * 8;
                const long mr = mbase + row;
                const int nseg = n0 + t0 * 32 + cseg;
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
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = stage[row * SR + cseg + j];
                    if (EP && p.res) {                             // residual branch (8 channels of this row)
                        const T* rs = reinterpret_cast<const T*>(p.res) + mr * p.ldres + nseg;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (nseg + j < p.N) v[j] += st<T>::ld(rs + j);
                    }
                    const bool full = (nseg + 8 <= p.N) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
                    if (full) {
                        if (p.accumulate) {
                            float old[8];
                            load8<T>(dst, old);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += old[j];
                        }
                        store8<T>(dst, v);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (nseg + j < p.N) {
                                float o = v[j];
                                if (p.accumulate) o += st<T>::ld(dst + j);
                                st<T>::stv(dst + j, o);
                            }
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


// ---------------------------------------------------------------------------------------------------------------
// LDS-tiled variant for the low-resolution 1x1 layers (and their data gradients): M <= 128 K pixels, K = 128..1024.
// There the problem is a small GEMM, not a stream: with the kernel above every 128-pixel tile re-stages its whole
// weight slice (M = 4096, N = K = 1024: 64 MB of L2->LDS weight traffic for 17 MB of activations) and the K loop is a
// chain of K/32 dependent steps on one workgroup per CU - 15-18 us for 6-25 MB problems, ~175 launches per step.
// Here a workgroup owns a 128 x (64 | 128) output tile, BOTH operands go through double-buffered LDS tiles of 64
// contraction steps (rows padded by 16 B: conflict-free ds_read_b128 fragments, the same K order in A and B), the
// next tile's global loads are in registers while the MFMAs of the current one run, one barrier per step.  The lazy
// prologue is applied once per element when the A tile is written to LDS.  Epilogue as above (bias, statistics,
// transpose through LDS, 16-byte row stores, optional accumulate).
template <int WN, int PRO, bool EP = false>
__global__ __launch_bounds__(256) void conv_gemm_tiled_kernel(GemmParams p) {
    constexpr int BM = 128, BK = 64, BN = 64 * WN, LDT = BK + 8;
    constexpr bool HAS_PRO = PRO != 0;
    constexpr int ACT = PRO == 1 ? ISA_ACT_RELU6 : ACT_RT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem);                  // [2][BM][LDT]
    bf16_t* sB = sA + 2 * BM * LDT;                                // [2][BN][LDT]
    float* pro_tab = reinterpret_cast<float*>(sB + 2 * BN * LDT);  // [2][K]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const GroupSel gs = group_sel(p.G);
    if (gs.g) {                                                    // statistic group (see conv_gemm_kernel)
        p.x = reinterpret_cast<const bf16_t*>(p.x) + (long)gs.g * p.M * p.ldx;
        p.y = reinterpret_cast<bf16_t*>(p.y) + (long)gs.g * p.M * p.ldy;
        p.pro.scale = goff(p.pro.scale, (long)gs.g * p.kp); p.pro.shift = goff(p.pro.shift, (long)gs.g * p.kp);
        p.pro.bscale = goff(p.pro.bscale, (long)gs.g * (p.M / ((long)p.mh * p.mw)) * p.cin);
        p.stats = goff(p.stats, (long)gs.g * ISA_STAT_R * 2 * p.N);
    }
    const int tm = gs.bx / tiles_n, tn = gs.bx - tm * tiles_n;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN, K = p.kp;
    const bf16_t* xin = reinterpret_cast<const bf16_t*>(p.x);
    const bf16_t* wg = reinterpret_cast<const bf16_t*>(p.w);
    bool fin = false;                                              // pending finalize: see conv_gemm_kernel
    if constexpr (HAS_PRO && !EP) fin = p.fin.stats != nullptr;
    if constexpr (HAS_PRO) {
        for (int k = tid; k < K; k += 256) {
            if (fin && k < p.cin) continue;
            pro_tab[k] = (!fin && p.pro.scale) ? p.pro.scale[k] : 1.f;
            pro_tab[K + k] = (!fin && p.pro.shift) ? p.pro.shift[k] : 0.f;
        }
    }
    // staging: thread -> rows srow + 32 i, 8 contraction elements at sseg
    const int srow = tid >> 3, sseg = (tid & 7) * 8;
    long aoff[4];
    const float* bsrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long m = m0 + srow + 32 * i;
        if (m > p.M - 1) m = p.M - 1;                              // rows >= M read the last valid row; the epilogue masks them
        aoff[i] = m * p.ldx + sseg;
        bsrow[i] = (HAS_PRO && p.pro.bscale) ? p.pro.bscale + (m / ((long)p.mh * p.mw)) * p.cin + sseg : nullptr;
    }
    if constexpr (HAS_PRO && !EP) {
        if (fin) {                                                 // see conv_gemm_kernel: touch the first tile's rows, finalize
            unsigned probe[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) probe[i] = *reinterpret_cast<const unsigned*>(xin + aoff[i]);
            asm volatile("" ::: "memory");
            bn_fin_inline<256>(p.fin, p.cin, p.G, gs.g, pro_tab, K, tid);
            asm volatile("" :: "v"(probe[0]), "v"(probe[1]), "v"(probe[2]), "v"(probe[3]));
        }
    }
    // two register sets: the loads of K tile kt + 2 are issued while tile kt is multiplied and tile kt + 1 is written
    // to LDS - with one workgroup per CU on these small problems nothing else covers the memory latency
    struct Regs { bf16x8 a[4], b[BN / 32]; };
    long boff[BN / 32];
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
        int n = n0 + srow + 32 * i;
        if (n > p.N - 1) n = p.N - 1;                              // rows >= N read the last valid row; zeroed when stashed
        boff[i] = (long)n * K + sseg;
    }
    auto fetch = [&](Regs& R, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) R.a[i] = *reinterpret_cast<const bf16x8*>(xin + aoff[i] + kt * BK);
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) R.b[i] = *reinterpret_cast<const bf16x8*>(wg + boff[i] + kt * BK);
        asm volatile("" ::: "memory");                              // keep the prefetch here (LLVM sinks loads to their use)
    };
    auto stash = [&](const Regs& R, int buf, int kt) {
        const bf16x8 (&ra)[4] = R.a;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 o = ra[i];
            if constexpr (HAS_PRO) {
                const int k = kt * BK + sseg;
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(pro_tab + k), s1 = *reinterpret_cast<const f32x4*>(pro_tab + k + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(pro_tab + K + k), h1 = *reinterpret_cast<const f32x4*>(pro_tab + K + k + 4);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = act_t<ACT>(fmaf((float)ra[i][j], j < 4 ? s0[j & 3] : s1[j & 3], j < 4 ? h0[j & 3] : h1[j & 3]), p.pro.act);
                if (bsrow[i]) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= bsrow[i][kt * BK + j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
            }
            *reinterpret_cast<bf16x8*>(sA + (buf * BM + srow + 32 * i) * LDT + sseg) = o;
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i)
            *reinterpret_cast<bf16x8*>(sB + (buf * BN + srow + 32 * i) * LDT + sseg) = (n0 + srow + 32 * i < p.N) ? R.b[i] : bf16x8{0};
    };

    f32x16 acc[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x16{0};
    const int KT = K / BK;
    Regs R0, R1;
    fetch(R0, 0);
    if (KT > 1) fetch(R1, 1);
    __syncthreads();                                               // pro_tab
    stash(R0, 0, 0);
    __syncthreads();
    auto step = [&](int kt, Regs& Rnext, Regs& Rfree) {            // Rnext holds tile kt + 1; Rfree (tile kt, already in LDS) takes kt + 2
        const int buf = kt & 1;
        if (kt + 2 < KT) fetch(Rfree, kt + 2);
        const bf16_t* tA = sA + (buf * BM + wm * 64 + r) * LDT + 8 * hh;
        const bf16_t* tB = sB + (buf * BN + wn * 32 * WN + r) * LDT + 8 * hh;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[2], b[WN];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(tA + i * 32 * LDT + s * 16);
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(tB + j * 32 * LDT + s * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) stash(Rnext, buf ^ 1, kt + 1);
        __syncthreads();
    };
    for (int kt = 0; kt < KT; kt += 2) {
        step(kt, R1, R0);
        if (kt + 1 < KT) step(kt + 1, R0, R1);
    }

    // ---------------- epilogue (the tile buffers are free now): lane r <-> output channel -----------------------
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 33);
    float st_sum[WN], st_sq[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) { st_sum[j] = 0.f; st_sq[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long mbase = m0 + wm * 64 + i * 32;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int nb = n0 + wn * 32 * WN + j * 32;
            const int n = nb + r;
            const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
            constexpr bool has_ep = EP;
            const float es = (has_ep && n < p.N) ? p.ep_scale[n] : 1.f, eh = (has_ep && n < p.N) ? p.ep_shift[n] : 0.f;
            float s = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
                float v = acc[i][j][e] + bv;
                if (has_ep) v = act_apply(fmaf(v, es, eh), p.ep_act);
                if (mbase + row < p.M) { s += v; s2 += v * v; }
                stage[row * 33 + r] = v;
            }
            if (p.stats) {
                s += __shfl_xor(s, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                st_sum[j] += s; st_sq[j] += s2;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int half = 0; half < 2; ++half) {                 // four lanes per row: contiguous 32-channel pieces (see conv_gemm_kernel)
                const int row = (lane >> 2) + 16 * half, cseg = (lane & 3) * 8;
                const long mr = mbase + row;
                const int nseg = nb + cseg;
                if (mr < p.M && nseg < p.N) {
                    bf16_t* dst = reinterpret_cast<bf16_t*>(p.y) + mr * p.ldy + nseg;
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = stage[row * 33 + cseg + q];
                    if (EP && p.res) {
                        const bf16_t* rs = reinterpret_cast<const bf16_t*>(p.res) + mr * p.ldres + nseg;
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (nseg + q < p.N) v[q] += st<bf16_t>::ld(rs + q);
                    }
                    const bool full = (nseg + 8 <= p.N) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
                    if (full) {
                        if (p.accumulate) {
                            float old[8];
                            load8<bf16_t>(dst, old);
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] += old[q];
                        }
                        store8<bf16_t>(dst, v);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            if (nseg + q < p.N) {
                                float o = v[q];
                                if (p.accumulate) o += st<bf16_t>::ld(dst + q);
                                st<bf16_t>::stv(dst + q, o);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (p.stats) {
        __shared__ float sred[2 * 128];
        if (tid < 2 * BN) sred[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int c = wn * 32 * WN + j * 32 + r;
            if (hh == 0) { atomicAdd(&sred[c], st_sum[j]); atomicAdd(&sred[BN + c], st_sq[j]); }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, n = n0 + (tid - which * BN);
            if (n < p.N) {
                float* rep = p.stats + (blockIdx.x & (ISA_STAT_R - 1)) * 2 * p.N;
                atomicAdd(rep + which * p.N + n, sred[tid]);
            }
        }
    }
}

template <int WN, bool EP>
int launch_tiled_ep(const GemmParams& p, bool has_pro, hipStream_t s) {
    constexpr int BN = 64 * WN, LDT = 72;
    const size_t tiles = 2 * (size_t)(128 + BN) * LDT * 2;
    const size_t lds = tiles + (has_pro ? 2 * (size_t)p.kp * 4 : 0);
    const long grid = ((p.M + 127) / 128) * ((p.N + BN - 1) / BN) * p.G;      // p.M is per group
    const int idx = has_pro && p.pro.act == ISA_ACT_RELU6 ? 1 : (has_pro ? 2 : 0);
    const void* fn = idx == 1 ? reinterpret_cast<const void*>(&conv_gemm_tiled_kernel<WN, 1, EP>)
                   : idx == 2 ? reinterpret_cast<const void*>(&conv_gemm_tiled_kernel<WN, 2, EP>)
                              : reinterpret_cast<const void*>(&conv_gemm_tiled_kernel<WN, 0, EP>);
    if (lds > 64 * 1024) {
        static bool configured[3] = {false, false, false};
        if (!configured[idx]) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return ISA_ELAUNCH;
            configured[idx] = true;
        }
    }
    if (idx == 1) hipLaunchKernelGGL((conv_gemm_tiled_kernel<WN, 1, EP>), dim3((unsigned)grid), dim3(256), lds, s, p);
    else if (idx == 2) hipLaunchKernelGGL((conv_gemm_tiled_kernel<WN, 2, EP>), dim3((unsigned)grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv_gemm_tiled_kernel<WN, 0, EP>), dim3((unsigned)grid), dim3(256), lds, s, p);
    return launch_status();
}
template <int WN>
int launch_tiled(const GemmParams& p, bool has_pro, hipStream_t s) {
    return p.ep_scale ? launch_tiled_ep<WN, true>(p, has_pro, s) : launch_tiled_ep<WN, false>(p, has_pro, s);
}

template <typename T, int NT, int IN_MODE, int OUT_MODE>
int launch2(const GemmParams& p, bool has_pro, dim3 grid, size_t lds, hipStream_t s) {
    if constexpr (IN_MODE == ISA_IN_1X1 && OUT_MODE == ISA_OUT_PLAIN) {
        static const bool fast_ok = !(getenv("ISA_GEMM_FAST") && atoi(getenv("ISA_GEMM_FAST")) == 0);
        // Measured: on HBM-cold single launches (KBENCH_ROTATE) K = 128 gains (58.7 -> 51.9 us at 128x128), K = 64 is neutral and
        // K = 32 loses a little (31.0 vs 29.5 us); inside the training step taking it for every eligible launch is the best
        // setting (same-session A/B, images/s: off 590.8, >= 4 groups 590.2, >= 2 groups 591.1, all 593.0).  With a prologue
        // the wider tiles would cross an occupancy step, so those stay on the general path.
        static const int fast_min_groups = getenv("ISA_GEMM_FAST_MIN_GROUPS") ? atoi(getenv("ISA_GEMM_FAST_MIN_GROUPS")) : 1;
        if (fast_ok && p.cin == p.kp && p.total_groups >= fast_min_groups && !p.ep_scale && (!has_pro || NT == 1)) {
            if (has_pro && p.pro.act == ISA_ACT_RELU6) hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 1, false, true>), grid, dim3(256), lds, s, p);
            else if (has_pro) hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 2, false, true>), grid, dim3(256), lds, s, p);
            else hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 0, false, true>), grid, dim3(256), lds, s, p);
            return launch_status();
        }
    }
    if constexpr (OUT_MODE == ISA_OUT_PLAIN && IN_MODE != ISA_IN_GATHER2) {
        if (p.ep_scale) {
            if (has_pro && p.pro.act == ISA_ACT_RELU6) hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 1, true>), grid, dim3(256), lds, s, p);
            else if (has_pro) hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 2, true>), grid, dim3(256), lds, s, p);
            else hipLaunchKernelGGL((conv_gemm_kernel<T, NT, IN_MODE, OUT_MODE, 0, true>), grid, dim3(256), lds, s, p);
            return launch_status();
        }
    }
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
    if constexpr (sizeof(T) == 2) {
        // low-resolution 1x1 layers: small GEMMs, both operands through LDS (conv_gemm_tiled_kernel)
        static const long tiled_max_m = getenv("ISA_GEMM_TILED_MAX_M") ? atol(getenv("ISA_GEMM_TILED_MAX_M")) : 65536;
        static const int tiled_min_k = getenv("ISA_GEMM_TILED_MIN_K") ? atoi(getenv("ISA_GEMM_TILED_MIN_K")) : 128;
        if (in_mode == ISA_IN_1X1 && out_mode == ISA_OUT_PLAIN && p.cin == p.kp && p.kp >= tiled_min_k && p.kp % 64 == 0 && p.kp <= 2048 &&
            p.N >= 64 && p.N % 16 == 0 && p.M <= tiled_max_m)
        {
            static const long wide_min_wgs = getenv("ISA_GEMM_TILED_WIDE_MIN") ? atol(getenv("ISA_GEMM_TILED_WIDE_MIN")) : 512;
            // 128-wide column tiles only when they still give >= 2 workgroups per CU: the second resident workgroup is
            // what keeps the MFMA pipe busy while the first one writes its next tile to LDS and waits at the barrier
            return (p.N >= 128 && ((p.M + 127) / 128) * ((p.N + 127) / 128) * p.G >= wide_min_wgs) ? launch_tiled<2>(p, has_pro, s)
                                                                                               : launch_tiled<1>(p, has_pro, s);
        }
    }
    int nt = p.N <= 32 ? 1 : (p.N <= 64 ? 2 : 4);
    // low-resolution levels: few 128-pixel tiles but hundreds of output channels.  Narrower column tiles put more
    // workgroups on the chip (the re-read A operand is L2-resident at these sizes): 4096 px x 512 ch ran as 128
    // workgroups on 256 CUs.
    const long tiles_m = (p.M + 127) / 128 * p.G;
    while (nt > 1 && tiles_m * ((p.N + 32 * nt - 1) / (32 * nt)) < 512) nt /= 2;
    const int n_blk = 32 * nt;
    // weight chunk sized to <= 48 KB of LDS
    int gpc = (48 * 1024) / (n_blk * 32 * (int)sizeof(T));
    if (gpc < 1) gpc = 1;
    if (gpc > p.total_groups) gpc = p.total_groups;
    p.groups_per_chunk = gpc;
    p.ldb = gpc * 32 + 16 / (int)sizeof(T);
    const size_t b_bytes = ((size_t)n_blk * p.ldb * sizeof(T) + 15) & ~(size_t)15;
    const size_t lds = b_bytes + (has_pro ? 2 * (size_t)p.kp * 4 : 0) + 4 * 32 * (nt == 2 ? 65 : 33) * 4;
    p.ntiles = (int)((p.M + 127) / 128);
    const int gy = (p.N + n_blk - 1) / n_blk;
    int gx = p.ntiles * p.G;
    static const int wg_per_cu = getenv("ISA_GEMM_WG_PER_CU") ? atoi(getenv("ISA_GEMM_WG_PER_CU")) : 3;
    const int cap = max(1, (256 * wg_per_cu) / gy);  // three resident workgroups per CU (measured: 2 -> 36.5 ms, 3 -> 36.3, 4 -> 36.9)
    if (gx > cap) gx = cap;
    gx = (int)group_grid(gx, p.G);
    dim3 grid(gx, gy);
    switch (nt) {
        case 1: return launch1<T, 1>(p, has_pro, in_mode, out_mode, grid, lds, s);
        case 2: return launch1<T, 2>(p, has_pro, in_mode, out_mode, grid, lds, s);
        default: return launch1<T, 4>(p, has_pro, in_mode, out_mode, grid, lds, s);
    }
}

}  // namespace

static int conv_gemm_impl(const isa_tensor* x, const isa_pro* pro, const void* w, int32_t kp,
                          const float* bias, const isa_tensor* y, int32_t in_mode,
                          int32_t out_mode, float* stats, int32_t accumulate, const isa_conv_ep* ep, void* stream) {
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
    if (ep) {
        if (!ep->scale || !ep->shift || stats || accumulate || out_mode != ISA_OUT_PLAIN || in_mode == ISA_IN_GATHER2) return ISA_EINVAL;
        p.ep_scale = ep->scale; p.ep_shift = ep->shift; p.ep_act = ep->act;
        if (ep->res) {
            const isa_tensor* r = ep->res;
            if (!tensor_ok(r, 1) || r->dtype != y->dtype || r->n != y->n || r->h != y->h || r->w != y->w || r->c != y->c) return ISA_EINVAL;
            p.res = r->data; p.ldres = r->ld;
        }
    }
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
    // statistic groups matter only where per-channel statistics or constants are involved; everything else treats the
    // batch as one problem
    p.G = 1;
    const int G = tensor_groups(x);
    if (G > 1 && (stats || (pro && (pro->scale || pro->shift)))) {
        if (in_mode != ISA_IN_1X1 || out_mode != ISA_OUT_PLAIN || ep || x->n % G || tensor_groups(y) != G) return ISA_EINVAL;
        p.G = G; p.M /= G;
    }
    if (pro && pro->fin) {
        if (!fin_valid(pro)) return ISA_EINVAL;
        if (!ep) p.fin = make_fin(pro);
        else if (int rc = fin_standalone(pro, x->c, G, as_stream(stream))) return rc;
    }
    if (!ep && in_mode == ISA_IN_3X3 && out_mode == ISA_OUT_PLAIN && x->dtype == ISA_BF16 && !has_pro && !stats && kp == 32 &&
        y->c <= 32)
        return conv3x3_tiled_launch(x, w, bias, y, accumulate, as_stream(stream));
    if (x->dtype == ISA_BF16) return launch0<bf16_t>(p, has_pro, in_mode, out_mode, as_stream(stream));
    return launch0<float>(p, has_pro, in_mode, out_mode, as_stream(stream));
}

extern "C" int isa_conv_gemm(const isa_tensor* x, const isa_pro* pro, const void* w, int32_t kp,
                             const float* bias, const isa_tensor* y, int32_t in_mode,
                             int32_t out_mode, float* stats, int32_t accumulate, void* stream) {
    return conv_gemm_impl(x, pro, w, kp, bias, y, in_mode, out_mode, stats, accumulate, nullptr, stream);
}

extern "C" int isa_conv_gemm_ep(const isa_tensor* x, const isa_pro* pro, const void* w, int32_t kp,
                                const float* bias, const isa_tensor* y, int32_t in_mode, const isa_conv_ep* ep, void* stream) {
    if (!ep) return ISA_EINVAL;
    return conv_gemm_impl(x, pro, w, kp, bias, y, in_mode, ISA_OUT_PLAIN, nullptr, 0, ep, stream);
}
