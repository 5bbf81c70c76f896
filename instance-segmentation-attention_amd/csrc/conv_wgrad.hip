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
#include <vector>
#include "common.hpp"
#include "tr_lds.hpp"

int conv3x3_wgrad_tiled_launch(const isa_tensor* x, const isa_tensor* dy, float* dw, float* dbias, float* ws, long ws_floats,
                               isa_slab_arena* sa,
                               hipStream_t s);     // conv3x3_tiled.hip

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
    long ws_floats;            // host side: capacity of ws
    int G;                     // statistic groups (plain 1x1 only): M, nchunks are per group (common.hpp)
    isa_slab_arena* sa;        // host side: deferred folds (may be NULL)
    float* ws;                 // [gridDim.x][gridDim.y][taps][TN*TK*1024 (+ TN*32 bias sums)] partial slabs
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
    const GroupSel gs = group_sel(p.G);
    if (gs.g) {                                             // this workgroup's statistic group: rows [g*M, (g+1)*M)
        p.x = reinterpret_cast<const T*>(p.x) + (long)gs.g * p.M * p.ldx;
        p.dy = reinterpret_cast<const T*>(p.dy) + (long)gs.g * p.M * p.ldd;
        p.pro.scale = goff(p.pro.scale, (long)gs.g * p.cin); p.pro.shift = goff(p.pro.shift, (long)gs.g * p.cin);
        p.pro.bscale = goff(p.pro.bscale, (long)gs.g * (p.M / ((long)p.mh * p.mw)) * p.cin);
    }
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
    const int cgn = lane % VN, rown = lane / VN, cgk = lane % VK, rowk = lane / VK;
    constexpr int RPN = 64 / VN, RPK = 64 / VK;
    const int cn0 = n0 + cgn * 8, ck0 = k0 + cgk * 8;
    const bool has_pro = p.pro.scale || p.pro.shift || p.pro.bscale || p.pro.act != ISA_ACT_NONE;
    float psc[8], psh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = min(ck0 + j, p.cin - 1);
        psc[j] = p.pro.scale ? p.pro.scale[c] : 1.f;
        psh[j] = p.pro.shift ? p.pro.shift[c] : 0.f;
    }

    // global -> registers for one chunk (prologue applied to x here)
    auto fetch = [&](long chunk) {
        const long mbase = chunk * PM;
#pragma unroll
        for (int v = 0; v < LOADS_N; ++v) {
            const long m = mbase + v * RPN + rown;
#pragma unroll
            for (int j = 0; j < 8; ++j) rd[v][j] = 0.f;
            if (m < p.M && cn0 < p.N) {
                long row = m;
                if (p.out_mode == ISA_OUT_SHUFFLE2) {
                    const unsigned mu = (unsigned)m, q = mu / (unsigned)p.mw;
                    const int px = (int)(mu - q * (unsigned)p.mw);
                    const unsigned pb = q / (unsigned)p.mh; const int py = (int)(q - pb * (unsigned)p.mh);
                    row = ((long)pb * p.dh + 2 * py + (tap >> 1)) * p.dw_ + 2 * px + (tap & 1);
                }
                ldvec<T>(din + row * p.ldd + cn0, rd[v], p.N - cn0);
            }
        }
#pragma unroll
        for (int v = 0; v < LOADS_K; ++v) {
            const long m = mbase + v * RPK + rowk;
#pragma unroll
            for (int j = 0; j < 8; ++j) rx[v][j] = 0.f;
            if (m < p.M && ck0 < p.cin) {
                long row = m; bool ok = true; long pb = 0;
                if (!plain || p.pro.bscale) {
                    const unsigned mu = (unsigned)m, q = mu / (unsigned)p.mw;
                    const int px = (int)(mu - q * (unsigned)p.mw);
                    pb = q / (unsigned)p.mh; const int py = (int)(q - (unsigned)pb * (unsigned)p.mh);
                    if (p.in_mode == ISA_IN_3X3) {
                        const int sy = py + tap / 3 - 1, sx = px + tap % 3 - 1;
                        ok = sy >= 0 && sy < p.xh && sx >= 0 && sx < p.xw;
                        row = (pb * p.xh + sy) * p.xw + sx;
                    }
                }
                if (ok) {
                    ldvec<T>(xin + row * p.ldx + ck0, rx[v], p.cin - ck0);
                    if (has_pro) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float z = act_t<ACT>(fmaf(rx[v][j], psc[j], psh[j]), p.pro.act);
                            if (p.pro.bscale) z *= p.pro.bscale[pb * p.cin + min(ck0 + j, p.cin - 1)];
                            rx[v][j] = (ck0 + j < p.cin) ? z : 0.f;
                        }
                    }
                }
            }
        }
    };
    auto stash = [&]() {          // registers -> wave-private LDS slab
#pragma unroll
        for (int v = 0; v < LOADS_N; ++v) {
            f32x4 a, b;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = rd[v][j]; b[j] = rd[v][4 + j]; }
            float* d = sD + (v * RPN + rown) * LDN + cgn * 8;
            *reinterpret_cast<f32x4*>(d) = a;
            *reinterpret_cast<f32x4*>(d + 4) = b;
        }
#pragma unroll
        for (int v = 0; v < LOADS_K; ++v) {
            f32x4 a, b;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = rx[v][j]; b[j] = rx[v][4 + j]; }
            float* d = sX + (v * RPK + rowk) * LDK + cgk * 8;
            *reinterpret_cast<f32x4*>(d) = a;
            *reinterpret_cast<f32x4*>(d + 4) = b;
        }
    };

    const long stride = (long)gs.nbx * 4;
    long chunk = (long)gs.bx * 4 + wave;
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
    // one partial slab per workgroup, raw fragment layout (256-byte coalesced stores); wgrad_reduce_kernel
    // sums the gridDim.x slabs of each (group, tap) in a fixed order and adds into dW: no atomics, so no
    // same-line serialisation at the memory side, and bitwise-reproducible weight gradients.
    constexpr int SLABF = TN * TK * 1024 + TN * 32;
    float* slab = p.ws + (((long)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * SLABF;
#pragma unroll
    for (int e = 0; e < TN * TK * 16; ++e) slab[e * 64 + lane] = red[e * 64 + lane];
    for (int i = lane; i < TN * 32; i += 64) slab[TN * TK * 1024 + i] = red[ACC_FLOATS + i];
}

// second stage of the weight gradient: dW (+ dbias) += sum over the gx partial slabs
struct WgReduce {
    const float* ws; float* dw; float* dbias; const int32_t* kmap;
    int gx, gy, taps, groups_k, tn, tk, N, cin, ksrc, out_mode;
    int rsplit;                 // adders per dW address; chosen so that the launch fills the chip
};
// one workgroup's share of a fold: block `bx` of the slab elements, tile group `group`, z = tap * rsplit + split.
// Each thread folds 1/rsplit of the gx slabs of one fragment element (coalesced 256-byte reads, independent
// loads) and adds it with ONE atomic: rsplit adders per address instead of gx.
__device__ __forceinline__ void fold_fragments(const WgReduce& q, int bx, int nbx, int group, int z) {
    const int slabf = q.tn * q.tk * 1024 + q.tn * 32;
    const int tap = z / q.rsplit, split = z % q.rsplit;
    const int gn = group / q.groups_k, gk = group % q.groups_k;
    const int n0 = gn * q.tn * 32, k0 = gk * q.tk * 32;
    const int per = (q.gx + q.rsplit - 1) / q.rsplit;
    const int b0 = split * per, b1 = min(q.gx, b0 + per);
    if (b0 >= b1) return;
    for (int idx = bx * 256 + threadIdx.x; idx < slabf; idx += nbx * 256) {
        float s = 0.f;
#pragma unroll 8
        for (int b = b0; b < b1; ++b)
            s += q.ws[(((long)b * q.gy + group) * q.taps + tap) * slabf + idx];
        if (idx < q.tn * q.tk * 1024) {
            const int lane = idx & 63, e = (idx >> 6) & 15, t = idx >> 10;
            const int i = t / q.tk, j = t - i * q.tk;
            const int r = lane & 31, hh = lane >> 5;
            const int kd = k0 + j * 32 + r;
            const int n = n0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (kd >= q.cin || n >= q.N) continue;
            const int k = q.kmap ? q.kmap[kd] : kd;
            if (k < 0) continue;
            long off;
            if (q.out_mode == ISA_OUT_SHUFFLE2) off = ((long)k * q.N + n) * 4 + tap;       // [K][Co][2][2]
            else off = ((long)n * q.ksrc + k) * q.taps + tap;                              // [N][K][kh][kw]
            atomicAdd(q.dw + off, s);
        } else if (q.dbias && gk == 0 && (tap == 0 || q.out_mode == ISA_OUT_SHUFFLE2)) {   // transposed conv: every quadrant's pixels
            const int n = n0 + (idx - q.tn * q.tk * 1024);
            if (n < q.N) atomicAdd(q.dbias + n, s);
        }
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgReduce q) {
    // grid = (slab elements / 256, tile groups, taps * rsplit)
    fold_fragments(q, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z);
}

// depthwise tile slabs [gx][ncb][10*cb] (dwconv_tiled.hip): dw[c][t] += sum_b slab[b][cblk][t*cb+cc]; dbias from row 9
__device__ __forceinline__ void fold_depthwise(const float* ws, int nblk, int ncb, int cbw, int C, int csrc, float* dw,
                                               float* dbias, int cblk, int split, int rsplit) {
    const int per = (nblk + rsplit - 1) / rsplit;
    const int b0 = split * per, b1 = min(nblk, b0 + per);
    if (b0 >= b1) return;
    for (int i = threadIdx.x; i < 10 * cbw; i += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += ws[((long)b * ncb + cblk) * 10 * cbw + i];
        const int tp = i / cbw, c = cblk * cbw + (i - tp * cbw);
        if (c >= csrc || c >= C) continue;
        if (tp < 9) atomicAdd(dw + c * 9 + tp, s);
        else if (dbias) atomicAdd(dbias + c, s);
    }
}

// the deferred folds of a whole backward pass, FOLD_CHUNK descriptors per launch (kernel arguments: a captured
// hipGraph keeps them by value); a workgroup finds its descriptor by binary search over the block prefix
constexpr int FOLD_CHUNK = 32;
struct FoldChunk { FoldDesc d[FOLD_CHUNK]; int n; };
__global__ __launch_bounds__(256) void wgrad_fold_kernel(FoldChunk ch) {
    int lo = 0, hi = ch.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ch.d[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const FoldDesc& d = ch.d[lo];
    const int lb = blockIdx.x - d.first_block;
    if (d.kind == 0) {
        const int slabf = d.tn * d.tk * 1024 + d.tn * 32;
        const int nbx = (slabf + 255) / 256;
        WgReduce q{d.ws, d.dw, d.dbias, d.kmap, d.gx, d.gy, d.taps, d.groups_k, d.tn, d.tk, d.N, d.cin, d.ksrc, d.out_mode, d.rsplit};
        fold_fragments(q, lb % nbx, nbx, (lb / nbx) % d.gy, lb / (nbx * d.gy));
    } else {
        fold_depthwise(d.ws, d.gx, d.gy, d.tk, d.N, d.ksrc, d.dw, d.dbias, lb % d.gy, lb / d.gy, d.rsplit);
    }
}

constexpr long DEFER_MIN_FLOATS = 16L << 20;       // an arena with less than this left refuses the call (ISA_ENOMEM)

static int launch_reduce(const WgParams& p, int gx, int gy, int tn, int tk, hipStream_t s) {
    const int slabf = tn * tk * 1024 + tn * 32;
    // enough splits for >= ~512 workgroups (a 512-slab reduce of one tile group ran as 144), at most 64 adders per
    // address and never more splits than slabs
    const long base_blocks = (long)cdiv(slabf, 256) * gy * p.taps;
    int rsplit = (int)((512 + base_blocks - 1) / base_blocks);
    if (rsplit > 64) rsplit = 64;
    if (rsplit > gx) rsplit = gx;
    if (rsplit < 1) rsplit = 1;
    if (p.sa && p.sa->from_arena) {
        FoldDesc d{p.ws, p.dw, p.dbias, p.kmap, 0, gx, gy, p.taps, p.groups_k, tn, tk, p.N, p.cin, p.ksrc, p.out_mode, 0, 0, 0};
        if (defer_push(p.sa, d, (long)gx * gy * p.taps * slabf)) return ISA_OK;
    }
    WgReduce q{p.ws, p.dw, p.dbias, p.kmap, gx, gy, p.taps, p.groups_k, tn, tk, p.N, p.cin, p.ksrc, p.out_mode, rsplit};
    dim3 grid(cdiv(slabf, 256), gy, p.taps * rsplit);
    hipLaunchKernelGGL(wgrad_reduce_kernel, grid, dim3(256), 0, s, q);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------
// bf16 storage: the same contraction on v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate, which
// otherwise bounds this kernel: 2*M*N*K flops at 157 TF/s ~ the HBM time of the two operands).
// Operands A[i=n][k=pixel 8h+j] / B[k=pixel 8h+j][j=k-channel] need 8 consecutive PIXELS of one
// channel per lane: the tiles are staged row-major [pixel][channel] in LDS (coalesced 16-byte loads,
// prologue applied in fp32 then rounded once, exactly like conv_gemm's forward operand) and fetched
// with ds_read_b64_tr_b16, the hardware transpose read (lane (r,h) element j <- tile[8h+j][r];
// verified by scripts/probes/tr_probe.hip).  Row stride is kept == 64 (mod 256) bytes so the four
// 64-byte row pieces a half-wave touches land on disjoint banks.
// ---------------------------------------------------------------------------------------------
template <int TN, int TK, int ACT>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(WgParams p) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    constexpr int PMB = 32;                                   // pixels per wave-chunk = 2 MFMA k-steps
    constexpr int SN = TrStride<TN * 32>::bytes, SK = TrStride<TK * 32>::bytes;
    constexpr int SLAB = PMB * (SN + SK);
    constexpr int VN = TN * 4, VK = TK * 4;
    constexpr int LOADS_N = PMB * VN / 64, LOADS_K = PMB * VK / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int gn = blockIdx.y / p.groups_k, gk = blockIdx.y % p.groups_k;
    const int tap = blockIdx.z;
    const int n0 = gn * TN * 32, k0 = gk * TK * 32;
    char* sD = ldsb + wave * SLAB;
    char* sX = sD + PMB * SN;
    const GroupSel gs = group_sel(p.G);
    if (gs.g) {                                             // statistic group (see conv_wgrad_kernel)
        p.x = reinterpret_cast<const bf16_t*>(p.x) + (long)gs.g * p.M * p.ldx;
        p.dy = reinterpret_cast<const bf16_t*>(p.dy) + (long)gs.g * p.M * p.ldd;
        p.pro.scale = goff(p.pro.scale, (long)gs.g * p.cin); p.pro.shift = goff(p.pro.shift, (long)gs.g * p.cin);
        p.pro.bscale = goff(p.pro.bscale, (long)gs.g * (p.M / ((long)p.mh * p.mw)) * p.cin);
    }
    const bf16_t* xin = reinterpret_cast<const bf16_t*>(p.x);
    const bf16_t* din = reinterpret_cast<const bf16_t*>(p.dy);
    const bool plain = p.in_mode == ISA_IN_1X1 && p.out_mode == ISA_OUT_PLAIN;

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = f32x16{0};
    float dbs[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) dbs[i] = 0.f;
    bf16x8 rd[LOADS_N], rx[LOADS_K];
    const long nchunks = (p.M + PMB - 1) / PMB;
    // a lane keeps ONE 8-channel group of each operand for the whole kernel (VN, VK divide 64), so the
    // prologue constants are loaded once into registers instead of per element per chunk
    const int cgn = lane % VN, rown = lane / VN, cgk = lane % VK, rowk = lane / VK;
    constexpr int RPN = 64 / VN, RPK = 64 / VK;           // pixel rows covered per load instruction
    const int cn0 = n0 + cgn * 8, ck0 = k0 + cgk * 8;
    const bool has_aff = p.pro.scale || p.pro.shift;
    const bool has_pro = has_aff || p.pro.bscale || p.pro.act != ISA_ACT_NONE;
    float psc[8], psh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = min(ck0 + j, p.cin - 1);
        psc[j] = p.pro.scale ? p.pro.scale[c] : 1.f;
        psh[j] = p.pro.shift ? p.pro.shift[c] : 0.f;
    }

    auto fetch = [&](long chunk) {
        const long mbase = chunk * PMB;
#pragma unroll
        for (int v = 0; v < LOADS_N; ++v) {
            const long m = mbase + v * RPN + rown;
            rd[v] = bf16x8{0};
            if (m < p.M && cn0 < p.N) {
                long row = m;
                if (p.out_mode == ISA_OUT_SHUFFLE2) {
                    const unsigned mu = (unsigned)m, q = mu / (unsigned)p.mw;
                    const int px = (int)(mu - q * (unsigned)p.mw);
                    const unsigned pb = q / (unsigned)p.mh; const int py = (int)(q - pb * (unsigned)p.mh);
                    row = ((long)pb * p.dh + 2 * py + (tap >> 1)) * p.dw_ + 2 * px + (tap & 1);
                }
                const bf16_t* src = din + row * p.ldd + cn0;
                if (cn0 + 8 <= p.N) rd[v] = *reinterpret_cast<const bf16x8*>(src);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) rd[v][j] = (cn0 + j < p.N) ? src[j] : (bf16_t)0.f;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < LOADS_K; ++v) {
            const long m = mbase + v * RPK + rowk;
            rx[v] = bf16x8{0};
            if (m < p.M && ck0 < p.cin) {
                long row = m; bool ok = true; long pb = 0;
                if (!plain || p.pro.bscale) {
                    const unsigned mu = (unsigned)m, q = mu / (unsigned)p.mw;
                    const int px = (int)(mu - q * (unsigned)p.mw);
                    pb = q / (unsigned)p.mh; const int py = (int)(q - (unsigned)pb * (unsigned)p.mh);
                    if (p.in_mode == ISA_IN_3X3) {
                        const int sy = py + tap / 3 - 1, sx = px + tap % 3 - 1;
                        ok = sy >= 0 && sy < p.xh && sx >= 0 && sx < p.xw;
                        row = (pb * p.xh + sy) * p.xw + sx;
                    }
                }
                if (ok) {
                    const bf16_t* src = xin + row * p.ldx + ck0;
                    bf16x8 raw;
                    if (ck0 + 8 <= p.cin) raw = *reinterpret_cast<const bf16x8*>(src);
                    else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) raw[j] = (ck0 + j < p.cin) ? src[j] : (bf16_t)0.f;
                    }
                    if (has_pro) {
                        // wave-uniform tests hoisted out of the element loop (the per-image multiplier and the channel
                        // tail were a branch and a select per element)
                        float z[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) z[j] = act_t<ACT>(fmaf((float)raw[j], psc[j], psh[j]), p.pro.act);
                        if (p.pro.bscale) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) z[j] *= p.pro.bscale[pb * p.cin + min(ck0 + j, p.cin - 1)];
                        }
                        if (ck0 + 8 > p.cin) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) if (ck0 + j >= p.cin) z[j] = 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) raw[j] = (bf16_t)z[j];
                    }
                    rx[v] = raw;
                }
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int v = 0; v < LOADS_N; ++v)
            *reinterpret_cast<bf16x8*>(sD + (v * RPN + rown) * SN + cgn * 16) = rd[v];
#pragma unroll
        for (int v = 0; v < LOADS_K; ++v)
            *reinterpret_cast<bf16x8*>(sX + (v * RPK + rowk) * SK + cgk * 16) = rx[v];
    };

    const long stride = (long)gs.nbx * 4;
    long chunk = (long)gs.bx * 4 + wave;
    if (chunk < nchunks) fetch(chunk);
    for (; chunk < nchunks; chunk += stride) {
        stash();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (chunk + stride < nchunks) fetch(chunk + stride);
#pragma unroll
        for (int s = 0; s < PMB / 16; ++s) {
            bf16x8 a[TN], b[TK];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                a[i] = tr_frag(sD, SN, 16 * s, i * 32, lane);
#pragma unroll
                for (int j = 0; j < 8; ++j) dbs[i] += (float)a[i][j];
            }
#pragma unroll
            for (int j = 0; j < TK; ++j) b[j] = tr_frag(sX, SK, 16 * s, j * 32, lane);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    __syncthreads();
    float* red = reinterpret_cast<float*>(ldsb);
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
                const float sv = dbs[i] + __shfl_xor(dbs[i], 32, 64);
                if (hh == 0) { float* q = red + ACC_FLOATS + i * 32 + r; *q = (w == 0 ? 0.f : *q) + sv; }
            }
        }
        __syncthreads();
    }
    if (wave != 0) return;
    // one partial slab per workgroup, raw fragment layout (256-byte coalesced stores); wgrad_reduce_kernel
    // sums the gridDim.x slabs of each (group, tap) in a fixed order and adds into dW: no atomics, so no
    // same-line serialisation at the memory side, and bitwise-reproducible weight gradients.
    constexpr int SLABF = TN * TK * 1024 + TN * 32;
    float* slab = p.ws + (((long)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * SLABF;
#pragma unroll
    for (int e = 0; e < TN * TK * 16; ++e) slab[e * 64 + lane] = red[e * 64 + lane];
    for (int i = lane; i < TN * 32; i += 64) slab[TN * TK * 1024 + i] = red[ACC_FLOATS + i];
}

template <int TN, int TK>
int launch_wg_bf16(WgParams& p, int groups_n, hipStream_t s) {
    constexpr int SN = TrStride<TN * 32>::bytes, SK = TrStride<TK * 32>::bytes;
    const size_t slab = (size_t)32 * (SN + SK) * 4;
    const size_t redb = ((size_t)TN * TK * 16 * 64 + TN * 32) * 4;
    const size_t lds = slab > redb ? slab : redb;
    const int gy = groups_n * p.groups_k;
    const long nchunks = (p.M + 31) / 32;
    const long slabf = (long)TN * TK * 1024 + TN * 32;
    long want = (nchunks + 3) / 4 * p.G;
    long cap = (256L * 2) / ((long)gy * p.taps);
    if (int rc = defer_ws(p.sa, &p.ws, &p.ws_floats)) return rc;
    const long ws_cap = p.ws_floats / (slabf * gy * p.taps);
    if (ws_cap < p.G) return p.sa ? ISA_ENOMEM : ISA_EINVAL;   // workspace too small for even one slab set per group
    if (cap > ws_cap) cap = ws_cap;
    if (cap < 1) cap = 1;
    const int gx = (int)group_grid(want < cap ? want : cap, p.G);
    dim3 grid(gx, gy, p.taps);
    if (p.pro.act == ISA_ACT_RELU6) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<TN, TK, ISA_ACT_RELU6>), grid, dim3(256), lds, s, p);
    else if (p.pro.act == ISA_ACT_LEAKY && TN == 1 && TK == 1) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<TN, TK, ISA_ACT_LEAKY>), grid, dim3(256), lds, s, p);
    else if (p.pro.act == ISA_ACT_NONE) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<TN, TK, ISA_ACT_NONE>), grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<TN, TK, ACT_RT>), grid, dim3(256), lds, s, p);
    if (launch_status() != ISA_OK) return ISA_ELAUNCH;
    return launch_reduce(p, gx, gy, TN, TK, s);
}


template <typename T, int TN, int TK>
int launch_wg(WgParams& p, int groups_n, hipStream_t s) {
    constexpr int LDN = TN * 32 + 4, LDK = TK * 32 + 4;
    const size_t slab = (size_t)PM * (LDN + LDK) * 4 * 4;
    const size_t redb = ((size_t)TN * TK * 16 * 64 + TN * 32) * 4;
    const size_t lds = slab > redb ? slab : redb;
    const int gy = groups_n * p.groups_k;
    const long slabf = (long)TN * TK * 1024 + TN * 32;
    long want = (p.nchunks + 3) / 4 * p.G;
    long cap = (256L * 2) / ((long)gy * p.taps);
    if (int rc = defer_ws(p.sa, &p.ws, &p.ws_floats)) return rc;
    const long ws_cap = p.ws_floats / (slabf * gy * p.taps);
    if (ws_cap < p.G) return p.sa ? ISA_ENOMEM : ISA_EINVAL;
    if (cap > ws_cap) cap = ws_cap;
    if (cap < 1) cap = 1;
    const int gx = (int)group_grid(want < cap ? want : cap, p.G);
    dim3 grid(gx, gy, p.taps);
    if (p.pro.act == ISA_ACT_RELU6) hipLaunchKernelGGL((conv_wgrad_kernel<T, TN, TK, ISA_ACT_RELU6>), grid, dim3(256), lds, s, p);
    else if (p.pro.act == ISA_ACT_NONE) hipLaunchKernelGGL((conv_wgrad_kernel<T, TN, TK, ISA_ACT_NONE>), grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<T, TN, TK, ACT_RT>), grid, dim3(256), lds, s, p);
    if (launch_status() != ISA_OK) return ISA_ELAUNCH;
    return launch_reduce(p, gx, gy, TN, TK, s);
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
    if constexpr (sizeof(T) == 2) {
        if (tn == 1 && tk == 1) return launch_wg_bf16<1, 1>(p, groups_n, s);
        if (tn == 1 && tk == 2) return launch_wg_bf16<1, 2>(p, groups_n, s);
        if (tn == 2 && tk == 1) return launch_wg_bf16<2, 1>(p, groups_n, s);
        if (tn == 2 && tk == 2) return launch_wg_bf16<2, 2>(p, groups_n, s);
        if (tn == 1 && tk == 4) return launch_wg_bf16<1, 4>(p, groups_n, s);
        if (tn == 4 && tk == 1) return launch_wg_bf16<4, 1>(p, groups_n, s);
        return ISA_EINVAL;
    }
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
                              const int32_t* kmap, int32_t ksrc, float* ws, int64_t ws_floats,
                              isa_slab_arena* defer, void* stream) {
    if (pro && pro->fin) { if (int rc = fin_standalone(pro, x->c, tensor_groups(x), as_stream(stream))) return rc; }   // no in-kernel form here
    if (!tensor_ok(x, 8) || !tensor_ok(dy, 8) || !dw || x->dtype != dy->dtype || (!ws && !defer)) return ISA_EINVAL;
    if (in_mode == ISA_IN_GATHER2) return ISA_EINVAL;
    WgParams p{};
    p.x = x->data; p.xh = x->h; p.xw = x->w; p.cin = x->c; p.ldx = x->ld;
    p.dy = dy->data; p.dh = dy->h; p.dw_ = dy->w; p.ldd = dy->ld;
    p.mh = x->h; p.mw = x->w; p.M = (long)x->n * x->h * x->w;
    p.pro = make_pro(pro); p.dw = dw; p.dbias = dbias; p.kmap = kmap; p.ksrc = ksrc > 0 ? ksrc : x->c;
    p.in_mode = in_mode; p.out_mode = out_mode; p.ws = ws; p.ws_floats = ws_floats; p.sa = defer;
    if (out_mode == ISA_OUT_SHUFFLE2) {
        if (in_mode != ISA_IN_1X1 || dy->h != 2 * x->h || dy->w != 2 * x->w || dy->n != x->n) return ISA_EINVAL;
        p.taps = 4; p.N = dy->c;             // dbias: the four quadrant slabs each carry the column sums of their pixels
    } else {
        if (dy->h != x->h || dy->w != x->w || dy->n != x->n) return ISA_EINVAL;
        p.taps = in_mode == ISA_IN_3X3 ? 9 : 1; p.N = dy->c;
    }
    p.G = 1;
    const int G = tensor_groups(x);
    if (G > 1 && (p.pro.scale || p.pro.shift)) {            // only the per-channel prologue constants differ by group
        if (in_mode != ISA_IN_1X1 || out_mode != ISA_OUT_PLAIN || x->n % G) return ISA_EINVAL;
        p.G = G; p.M /= G;
    }
    p.nchunks = (p.M + PM - 1) / PM;
    if (in_mode == ISA_IN_3X3 && out_mode == ISA_OUT_PLAIN && x->dtype == ISA_BF16 && pro_trivial(p.pro) && !kmap &&
        x->c <= 32 && dy->c <= 32 && p.ksrc == x->c)
        return conv3x3_wgrad_tiled_launch(x, dy, dw, dbias, ws, ws_floats, defer, as_stream(stream));
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
// 16-byte loads: a lane owns one 8-channel group of a pixel (item = pixel * cg + group; grid_keep_cg keeps the group
// fixed per lane), the lanes of a wave that share a group fold by shuffles, lanes < cg add into the LDS row.  The
// element-wise walk above moved 2 bytes per lane per load (71 us for a 67 MB tensor).
template <typename T>
__global__ __launch_bounds__(256) void colsum8_kernel(const T* x, long pixels, int c, int ld, float* out) {
    extern __shared__ float red[];
    const int cg = (c + 7) / 8;
    for (int i = threadIdx.x; i < c; i += 256) red[i] = 0.f;
    __syncthreads();
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int last = -1;
    for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < pixels * cg; item += (long)gridDim.x * 256) {
        const int c0 = (int)(item % cg) * 8; const long pix = item / cg;
        if (c0 != last && last >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { if (last + j < c) atomicAdd(&red[last + j], s[j]); s[j] = 0.f; }
        }
        last = c0;
        float v[8];
        load8g<T>(x + pix * ld + c0, v, min(8, c - c0));
#pragma unroll
        for (int j = 0; j < 8; ++j) if (c0 + j < c) s[j] += v[j];
    }
    const int lane = threadIdx.x & 63;
    const bool fixed = ((long)gridDim.x * 256) % cg == 0 && cg < 64;
    const int cfix = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cg) * 8;
    if (fixed) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = fold_stride(s[j], cg, lane);
        if (lane < cg) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (cfix + j < c) atomicAdd(&red[cfix + j], s[j]);
        }
    } else if (last >= 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (last + j < c) atomicAdd(&red[last + j], s[j]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256)
        if (red[i] != 0.f) atomicAdd(out + i, red[i]);
}
}  // namespace

extern "C" int isa_colsum(const isa_tensor* x, float* out, void* stream) {
    if (!tensor_ok(x, 1) || !out) return ISA_EINVAL;
    const long pixels = (long)x->n * x->h * x->w;
    if (tensor_ok(x, 8)) {
        const int cg = (x->c + 7) / 8;
        const int grid = grid_keep_cg(grid_cap(cdiv(pixels * cg, 256), 1024), cg);
        if (x->dtype == ISA_BF16)
            hipLaunchKernelGGL(colsum8_kernel<bf16_t>, dim3(grid), dim3(256), x->c * 4, as_stream(stream),
                               (const bf16_t*)x->data, pixels, x->c, x->ld, out);
        else
            hipLaunchKernelGGL(colsum8_kernel<float>, dim3(grid), dim3(256), x->c * 4, as_stream(stream),
                               (const float*)x->data, pixels, x->c, x->ld, out);
        return launch_status();
    }
    const int grid = grid_cap(cdiv(pixels, 64), 512);
    if (x->dtype == ISA_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(grid), dim3(256), x->c * 4, as_stream(stream),
                           (const bf16_t*)x->data, pixels, x->c, x->ld, out);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, dim3(grid), dim3(256), x->c * 4, as_stream(stream),
                           (const float*)x->data, pixels, x->c, x->ld, out);
    return launch_status();
}

// second-stage reduction for other translation units that write conv_wgrad-format slabs (conv_fused_bwd.hip,
// conv3x3_tiled.hip): one tile group, `taps` slab sets per workgroup
int wgrad_slab_reduce_launch(float* ws, float* dw, float* dbias, int gx, int tn, int tk, int N, int cin, int taps,
                             isa_slab_arena* sa, hipStream_t s) {
    WgParams p{};
    p.sa = sa;
    p.ws = ws; p.dw = dw; p.dbias = dbias; p.kmap = nullptr; p.taps = taps; p.groups_k = 1; p.N = N; p.cin = cin;
    p.ksrc = cin; p.out_mode = ISA_OUT_PLAIN;
    return launch_reduce(p, gx, 1, tn, tk, s);
}

// ---- deferred folds ---------------------------------------------------------------------------------
int defer_ws(isa_slab_arena* a, float** ws, long* ws_floats) {
    if (!a) return ISA_OK;
    a->from_arena = false;
    if (a->floats - a->used < DEFER_MIN_FLOATS) return ISA_ENOMEM;
    a->from_arena = true;
    *ws_floats = a->floats - a->used;
    *ws = a->base + a->used;
    return ISA_OK;
}

bool defer_push(isa_slab_arena* a, FoldDesc f, long used_floats) {
    if (!a || !a->from_arena || f.ws != a->base + a->used || used_floats > a->floats - a->used) return false;
    a->from_arena = false;
    // ~32 slabs per adder (independent coalesced loads); the whole backward pass supplies the parallelism
    int rsplit = (f.gx + 31) / 32;
    if (rsplit > 64) rsplit = 64;
    if (rsplit < 1) rsplit = 1;
    f.rsplit = rsplit;
    if (f.kind == 0) f.blocks = cdiv(f.tn * f.tk * 1024 + f.tn * 32, 256) * f.gy * f.taps * rsplit;
    else f.blocks = f.gy * rsplit;
    a->tab.push_back(f);
    a->used += (used_floats + 63) & ~63L;
    if (a->used > a->peak) a->peak = a->used;
    return true;
}

extern "C" int isa_slab_arena_create(float* region, int64_t region_floats, isa_slab_arena** out) {
    if (!out || !region || region_floats < DEFER_MIN_FLOATS || ((uintptr_t)region & 255)) return ISA_EINVAL;
    *out = new isa_slab_arena{region, (long)region_floats, 0, 0, false, {}};
    return ISA_OK;
}

extern "C" int isa_slab_arena_destroy(isa_slab_arena* a) {
    delete a;
    return ISA_OK;
}

extern "C" int isa_slab_arena_begin(isa_slab_arena* a) {
    if (!a) return ISA_EINVAL;
    a->used = 0; a->from_arena = false;
    a->tab.clear();
    return ISA_OK;
}

extern "C" int isa_slab_arena_flush(isa_slab_arena* a, void* stream, int32_t* n_folds, int64_t* floats_used) {
    if (!a) return ISA_EINVAL;
    if (n_folds) *n_folds = (int32_t)a->tab.size();
    if (floats_used) *floats_used = a->used;
    for (size_t i0 = 0; i0 < a->tab.size(); i0 += FOLD_CHUNK) {
        FoldChunk ch;
        ch.n = (int)(a->tab.size() - i0 < (size_t)FOLD_CHUNK ? a->tab.size() - i0 : FOLD_CHUNK);
        int blocks = 0;
        for (int i = 0; i < ch.n; ++i) { ch.d[i] = a->tab[i0 + i]; ch.d[i].first_block = blocks; blocks += ch.d[i].blocks; }
        for (int i = ch.n; i < FOLD_CHUNK; ++i) ch.d[i] = FoldDesc{};
        if (blocks > 0) hipLaunchKernelGGL(wgrad_fold_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), ch);
        if (launch_status() != ISA_OK) { a->tab.clear(); return ISA_ELAUNCH; }
    }
    a->tab.clear();
    return ISA_OK;
}
