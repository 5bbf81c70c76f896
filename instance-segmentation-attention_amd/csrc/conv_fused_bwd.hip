// Fused backward of   x --1x1 conv (W[N][K])--> y --BN(train)(+act)--> ...   for bf16 storage and N, K <= 64:
//     dy   = BN-backward(g, y)                         (isa_bn_bwd_apply, never stored)
//     dW  += dy^T . pro(x)                             (isa_conv_wgrad)
//     dx (+)= dy . W                                   (isa_conv_gemm data gradient)
//     xbn:  sums of BN(x)'s backward over the finished dx   (isa_bn_bwd_reduce of the layer that made x)
// in ONE pass: reads g, y, x once and writes dx once (6 narrow-tensor passes instead of 12-13).
//
// Built on the bf16 weight-gradient kernel's structure (conv_wgrad.hip): every WAVE owns 32-pixel chunks,
// staged row-major in a wave-private LDS slab; no __syncthreads in the main loop; the next chunk's raw
// 16-byte loads are in flight while the current chunk's MFMAs run (the BN/ReLU6 arithmetic is done when the
// registers are stashed to LDS, not when they are fetched).
//   wgrad:  dW[n][k] += sum_px dy[px][n] x[px][k]   A/B fragments by ds_read_b64_tr_b16 (pixel-contraction)
//   dgrad:  dx[px][k] = sum_n  dy[px][n] W[n][k]    A = ds_read_b128 rows of the SAME dy slab,
//                                                    B = W fragments held in registers for the whole kernel
//   the 32xK dx tile goes through the x slab (bf16) so global stores are 16-byte row pieces.
#include "common.hpp"
#include "tr_lds.hpp"

int wgrad_slab_reduce_launch(float* ws, float* dw, float* dbias, int gx, int tn, int tk, int N, int cin, int taps,
                             isa_slab_arena* sa, hipStream_t s);

namespace {

struct PwParams {
    const bf16_t *g, *y, *x; bf16_t* dx; const float* w; float* dw;
    int N, K, ldg, ldy, ldx, lddx; long M;
    const float *ysc, *ysh, *ymu, *yis, *yred; float ycnt_inv; int yact; float *ydgamma, *ydbeta;
    const float *xsc, *xsh, *xmu, *xis; int xact; float* xred;
    int accumulate; float* ws;
    const bf16_t* addend; int lda;       // optional: dx += addend (the residual branch's gradient, saves its axpy pass)
    int G;                               // statistic groups: M is per group (common.hpp)
};

// XMODE 0: plain x; 1: x = relu6(scale*x+shift) with BN(x) sums produced; 2: runtime prologue, sums if p.xred
template <int TN, int TK, int YACT, int XMODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pw_bn_bwd_kernel(PwParams p) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    constexpr int PMB = 32;
    constexpr int SN = TrStride<TN * 32>::bytes, SK = TrStride<TK * 32>::bytes;
    constexpr int SLAB = PMB * (SN + SK);
    constexpr int VN = TN * 4, VK = TK * 4;
    constexpr int LOADS_N = PMB * VN / 64, LOADS_K = PMB * VK / 64;
    constexpr int RPN = 64 / VN, RPK = 64 / VK;
    constexpr int XACT = XMODE == 1 ? ISA_ACT_RELU6 : (XMODE == 0 ? ISA_ACT_NONE : ACT_RT);
    constexpr int NS = TN * 2;                                   // 16-wide contraction steps over N (data gradient)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    char* sD = ldsb + wave * SLAB;
    char* sX = sD + PMB * SN;
    const GroupSel gs = group_sel(p.G);
    if (gs.g) {                                                  // this workgroup's statistic group: rows [g*M, (g+1)*M)
        const long gm = (long)gs.g * p.M;
        p.g += gm * p.ldg; p.y += gm * p.ldy; p.x += gm * p.ldx; p.dx += gm * p.lddx;
        if (p.addend) p.addend += gm * p.lda;
        p.ysc += gs.g * p.N; p.ysh += gs.g * p.N; p.ymu += gs.g * p.N; p.yis += gs.g * p.N; p.yred += gs.g * ISA_STAT_R * 2 * p.N;
        p.xsc = goff(p.xsc, (long)gs.g * p.K); p.xsh = goff(p.xsh, (long)gs.g * p.K); p.xmu = goff(p.xmu, (long)gs.g * p.K);
        p.xis = goff(p.xis, (long)gs.g * p.K); p.xred = goff(p.xred, (long)gs.g * ISA_STAT_R * 2 * p.K);
    }
    const bool want_xred = XMODE == 1 || (XMODE == 2 && p.xred != nullptr);

    // ---- per-lane constants: a lane keeps ONE 8-channel group of each operand
    const int cgn = lane % VN, rown = lane / VN, cgk = lane % VK, rowk = lane / VK;
    const int cn0 = cgn * 8, ck0 = cgk * 8;
    const bool nok = cn0 < p.N, kok = ck0 < p.K;
    // per-channel constants live in LDS (shared by the four waves) and are re-read at each stash: holding the
    // 9 x 8 floats per lane in registers costs a whole occupancy step
    float* cst = reinterpret_cast<float*>(ldsb + 4 * SLAB);       // [9][64]
    // W (N x K fp32, <= 16 KB) is staged through the still-unused slab area with coalesced loads: building the MFMA
    // fragments straight from global memory was 8 * NS * TK dependent scalar loads per lane - a phase trace of a small
    // launch showed 38 k of its 85 k cycles (64 -> 64 channels) in that prologue
    float* wl = reinterpret_cast<float*>(ldsb);
    for (int i = tid; i < p.N * p.K; i += 256) wl[i] = p.w[i];
    if (tid < 64) {
        const int cn = min(tid, p.N - 1), ck = min(tid, p.K - 1);
        cst[0 * 64 + tid] = p.ysc[cn]; cst[1 * 64 + tid] = p.ysh[cn]; cst[2 * 64 + tid] = p.ymu[cn];
        float r0 = 0.f, r1 = 0.f;                                             // fold the ISA_STAT_R replicas
#pragma unroll
        for (int r = 0; r < ISA_STAT_R; ++r) { r0 += p.yred[r * 2 * p.N + cn]; r1 += p.yred[r * 2 * p.N + p.N + cn]; }
        cst[3 * 64 + tid] = p.yis[cn] * (r1 * p.ycnt_inv);                    // invstd * mean(g' * yhat)
        cst[4 * 64 + tid] = r0 * p.ycnt_inv;                                  // mean(g')
        if (gs.bx == 0 && tid < p.N) {                                        // BN(y) parameter gradients (per group)
            if (p.ydgamma) atomicAdd(p.ydgamma + tid, r1);
            if (p.ydbeta) atomicAdd(p.ydbeta + tid, r0);
        }
        cst[5 * 64 + tid] = (XMODE && p.xsc) ? p.xsc[ck] : 1.f; cst[6 * 64 + tid] = (XMODE && p.xsh) ? p.xsh[ck] : 0.f;
        cst[7 * 64 + tid] = (XMODE && p.xmu) ? p.xmu[ck] : 0.f; cst[8 * 64 + tid] = (XMODE && p.xis) ? p.xis[ck] : 1.f;
    }
    __syncthreads();
    auto ldc = [&](int which, int c0, float (&v)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(cst + which * 64 + c0), b = *reinterpret_cast<const f32x4*>(cst + which * 64 + c0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    };
    // ---- W fragments for the data gradient: B[n = 16 s + 8 hh + jj][k = 32 j + r], rounded to bf16 once
    bf16x8 wb[NS][TK];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int n = 16 * s + 8 * hh + jj, k = 32 * j + r;
                wb[s][j][jj] = (bf16_t)((n < p.N && k < p.K) ? wl[n * p.K + k] : 0.f);
            }
    __syncthreads();                                              // the slabs reuse W's staging area

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = f32x16{0};
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }

    bf16x8 rg[LOADS_N], ry[LOADS_N], rx[LOADS_K], rxc[LOADS_K];
    const long nchunks = (p.M + PMB - 1) / PMB;
    auto fetch = [&](long chunk) {                               // raw loads only: nothing here waits on memory
        const long mbase = chunk * PMB;
#pragma unroll
        for (int v = 0; v < LOADS_N; ++v) {
            const long m = mbase + v * RPN + rown;
            rg[v] = bf16x8{0}; ry[v] = bf16x8{0};
            if (m < p.M && nok) {
                rg[v] = *reinterpret_cast<const bf16x8*>(p.g + m * p.ldg + cn0);
                ry[v] = *reinterpret_cast<const bf16x8*>(p.y + m * p.ldy + cn0);
            }
        }
#pragma unroll
        for (int v = 0; v < LOADS_K; ++v) {
            const long m = mbase + v * RPK + rowk;
            rx[v] = bf16x8{0};
            if (m < p.M && kok) rx[v] = *reinterpret_cast<const bf16x8*>(p.x + m * p.ldx + ck0);
        }
    };
    auto stash = [&](long chunk) {                               // BN-backward / prologue arithmetic, then LDS
        const long mbase = chunk * PMB;
        {
        float ysc[8], ysh[8], ymu[8], yq[8], yk0[8];
        ldc(0, cn0, ysc); ldc(1, cn0, ysh); ldc(2, cn0, ymu); ldc(3, cn0, yq); ldc(4, cn0, yk0);
#pragma unroll
        for (int v = 0; v < LOADS_N; ++v) {
            const bool ok = (mbase + v * RPN + rown) < p.M && nok;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float yy = (float)ry[v][j];
                const float dz = (float)rg[v][j] * act_grad_t<YACT>(fmaf(yy, ysc[j], ysh[j]), p.yact);
                const float val = ysc[j] * (dz - yk0[j] - (yy - ymu[j]) * yq[j]);
                o[j] = (bf16_t)((ok && cn0 + j < p.N) ? val : 0.f);
            }
            *reinterpret_cast<bf16x8*>(sD + (v * RPN + rown) * SN + cgn * 16) = o;
        }
        }
        float xsc[8], xsh[8];
        ldc(5, ck0, xsc); ldc(6, ck0, xsh);
#pragma unroll
        for (int v = 0; v < LOADS_K; ++v) {
            const bool ok = (mbase + v * RPK + rowk) < p.M && kok;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xr = (float)rx[v][j];
                const float z = XMODE ? act_t<XACT>(fmaf(xr, xsc[j], xsh[j]), p.xact) : xr;
                o[j] = (bf16_t)((ok && ck0 + j < p.K) ? z : 0.f);
            }
            *reinterpret_cast<bf16x8*>(sX + (v * RPK + rowk) * SK + cgk * 16) = o;
            rxc[v] = rx[v];                                       // raw x of THIS chunk for the BN(x) sums
        }
    };

    const long stride = (long)gs.nbx * 4;
    long chunk = (long)gs.bx * 4 + wave;
    if (chunk < nchunks) fetch(chunk);
    for (; chunk < nchunks; chunk += stride) {
        stash(chunk);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (chunk + stride < nchunks) fetch(chunk + stride);
        // operands of this chunk's store epilogue (old dx / residual-branch gradient): requested now, consumed after
        // the MFMAs, so their latency is not exposed at the end of the trip
        constexpr bool EPI_PREFETCH = TK == 1;                    // wider x tiles have no registers to spare
        bf16x8 repi_old[EPI_PREFETCH ? LOADS_K : 1], repi_add[EPI_PREFETCH ? LOADS_K : 1];
        if constexpr (EPI_PREFETCH) {
            const long mb = chunk * PMB;
#pragma unroll
            for (int v = 0; v < LOADS_K; ++v) {
                const long m = mb + v * RPK + rowk;
                repi_old[v] = bf16x8{0}; repi_add[v] = bf16x8{0};
                if (m < p.M && kok) {
                    if (p.accumulate) repi_old[v] = *reinterpret_cast<const bf16x8*>(p.dx + m * p.lddx + ck0);
                    if (p.addend) repi_add[v] = *reinterpret_cast<const bf16x8*>(p.addend + m * p.lda + ck0);
                }
            }
        }
        // ---- weight gradient
#pragma unroll
        for (int s = 0; s < PMB / 16; ++s) {
            bf16x8 a[TN], b[TK];
#pragma unroll
            for (int i = 0; i < TN; ++i) a[i] = tr_frag(sD, SN, 16 * s, i * 32, lane);
#pragma unroll
            for (int j = 0; j < TK; ++j) b[j] = tr_frag(sX, SK, 16 * s, j * 32, lane);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        // ---- data gradient for the 32 pixels of this chunk
        f32x16 dxa[TK];
#pragma unroll
        for (int j = 0; j < TK; ++j) dxa[j] = f32x16{0};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sD + r * SN + (16 * s + 8 * hh) * 2);
#pragma unroll
            for (int j = 0; j < TK; ++j)
                dxa[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wb[s][j], dxa[j], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                          // all transpose reads of sX are done
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * hh;
                *reinterpret_cast<bf16_t*>(sX + row * SK + (32 * j + r) * 2) = (bf16_t)dxa[j][e];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const long mbase = chunk * PMB;
        float xsc[8], xsh[8], xmu[8], xis[8];
        if (want_xred) { ldc(5, ck0, xsc); ldc(6, ck0, xsh); ldc(7, ck0, xmu); ldc(8, ck0, xis); }
#pragma unroll
        for (int v = 0; v < LOADS_K; ++v) {
            const long m = mbase + v * RPK + rowk;
            if (m < p.M && kok) {
                bf16x8 o = *reinterpret_cast<const bf16x8*>(sX + (v * RPK + rowk) * SK + cgk * 16);
                bf16_t* dst = p.dx + m * p.lddx + ck0;
                if (p.accumulate) {
                    bf16x8 old;
                    if constexpr (EPI_PREFETCH) old = repi_old[v]; else old = *reinterpret_cast<const bf16x8*>(dst);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)o[j] + (float)old[j]);
                }
                if (p.addend) {
                    bf16x8 ad;
                    if constexpr (EPI_PREFETCH) ad = repi_add[v]; else ad = *reinterpret_cast<const bf16x8*>(p.addend + m * p.lda + ck0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)o[j] + (float)ad[j]);
                }
                *reinterpret_cast<bf16x8*>(dst) = o;
                if (want_xred) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xr = (float)rxc[v][j];
                        const float dz = (float)o[j] * act_grad_t<XACT>(fmaf(xr, xsc[j], xsh[j]), p.xact);
                        s0[j] += dz; s1[j] += dz * ((xr - xmu[j]) * xis[j]);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                          // sX is rewritten by the next stash
    }

    // ---- fold the four waves' weight-gradient fragments, one slab per workgroup (conv_wgrad.hip layout)
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
        }
        __syncthreads();
    }
    constexpr int SLABF = TN * TK * 1024 + TN * 32;
    float* slab = p.ws + (long)blockIdx.x * SLABF;
    if (wave == 0) {
#pragma unroll
        for (int e = 0; e < TN * TK * 16; ++e) slab[e * 64 + lane] = red[e * 64 + lane];
        for (int i = lane; i < TN * 32; i += 64) slab[TN * TK * 1024 + i] = 0.f;     // no conv bias on this path
    }
    if (want_xred) {
        __syncthreads();
        float* xr = red;                                          // [2][TK*32]
        for (int i = tid; i < 2 * TK * 32; i += 256) xr[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v0 = row_fold<VK>(s0[j]), v1 = row_fold<VK>(s1[j]);      // lanes with equal lane % VK share channels
            if ((lane & 15) < VK) { atomicAdd(&xr[ck0 + j], v0); atomicAdd(&xr[TK * 32 + ck0 + j], v1); }
        }
        __syncthreads();
        for (int i = tid; i < 2 * TK * 32; i += 256) {
            const int which = i / (TK * 32), c = i - which * TK * 32;
            if (c < p.K && xr[i] != 0.f)
                atomicAdd(p.xred + (blockIdx.x & (ISA_STAT_R - 1)) * 2 * p.K + which * p.K + c, xr[i]);
        }
    }
}

template <int TN, int TK, int YACT, int XMODE>
int launch_inst(PwParams& p, long ws_floats, isa_slab_arena* sa, hipStream_t s) {
    constexpr int SN = TrStride<TN * 32>::bytes, SK = TrStride<TK * 32>::bytes;
    constexpr size_t slab = (size_t)32 * (SN + SK) * 4 + 9 * 64 * 4;
    constexpr size_t redb = ((size_t)TN * TK * 16 * 64 + TN * 32) * 4;
    constexpr size_t lds = slab > redb ? slab : redb;
    const long nchunks = (p.M + 31) / 32;
    const long slabf = (long)TN * TK * 1024 + TN * 32;
    long gx = (nchunks + 3) / 4 * p.G;
    if (gx > 256 * 2) gx = 256 * 2;                                  // 2 resident workgroups per CU; fewer slabs to reduce
    if (int rc = defer_ws(sa, &p.ws, &ws_floats)) return rc;
    const long ws_cap = ws_floats / slabf;
    if (ws_cap < p.G) return sa ? ISA_ENOMEM : ISA_EINVAL;
    if (gx > ws_cap) gx = ws_cap;
    gx = group_grid(gx, p.G);
    hipLaunchKernelGGL((pw_bn_bwd_kernel<TN, TK, YACT, XMODE>), dim3((unsigned)gx), dim3(256), lds, s, p);
    if (launch_status() != ISA_OK) return ISA_ELAUNCH;
    return wgrad_slab_reduce_launch(p.ws, p.dw, nullptr, (int)gx, TN, TK, p.N, p.K, 1, sa, s);
}

template <int TN, int TK>
int launch_tile(PwParams& p, int xmode, long ws_floats, isa_slab_arena* sa, hipStream_t s) {
    const bool y6 = p.yact == ISA_ACT_RELU6, y0 = p.yact == ISA_ACT_NONE;
#define PW_X(YA) (xmode == 0 ? launch_inst<TN, TK, YA, 0>(p, ws_floats, sa, s) \
                : xmode == 1 ? launch_inst<TN, TK, YA, 1>(p, ws_floats, sa, s) \
                             : launch_inst<TN, TK, YA, 2>(p, ws_floats, sa, s))
    if (y6) return PW_X(ISA_ACT_RELU6);
    if (y0) return PW_X(ISA_ACT_NONE);
    return PW_X(ACT_RT);
#undef PW_X
}

}  // namespace

extern "C" int isa_conv1x1_bn_backward(const isa_tensor* g, const isa_tensor* y, const isa_bn_bwd* ybn,
                                       const isa_tensor* x, const isa_pro* xpro, const isa_bn_bwd* xbn,
                                       const float* w, float* dw, const isa_tensor* dx, int32_t accumulate,
                                       const isa_tensor* addend, float* ws, int64_t ws_floats, isa_slab_arena* defer,
                                       void* stream) {
    if (xpro && xpro->fin) { if (int rc = fin_standalone(xpro, x->c, tensor_groups(x), as_stream(stream))) return rc; }   // no in-kernel form here
    if (!tensor_ok(g, 8) || !tensor_ok(y, 8) || !tensor_ok(x, 8) || !tensor_ok(dx, 8)) return ISA_EINVAL;
    if (g->dtype != ISA_BF16 || y->dtype != ISA_BF16 || x->dtype != ISA_BF16 || dx->dtype != ISA_BF16) return ISA_EINVAL;
    if (!ybn || !ybn->scale || !ybn->shift || !ybn->mean || !ybn->invstd || !ybn->red || !(ybn->count > 0)) return ISA_EINVAL;
    if (!w || !dw || (!ws && !defer)) return ISA_EINVAL;
    const int N = g->c, K = x->c;
    if (N > 64 || K > 64 || N % 8 || K % 8) return ISA_EINVAL;
    if (y->c != N || dx->c != K) return ISA_EINVAL;
    const isa_tensor* ts[3] = {y, x, dx};
    for (const isa_tensor* t : ts)
        if (t->n != g->n || t->h != g->h || t->w != g->w) return ISA_EINVAL;
    const ProDev xp = make_pro(xpro);
    if (xp.bscale) return ISA_EINVAL;
    if (xbn && (!xbn->mean || !xbn->invstd || !xbn->out_red)) return ISA_EINVAL;
    PwParams p{};
    p.g = (const bf16_t*)g->data; p.y = (const bf16_t*)y->data; p.x = (const bf16_t*)x->data; p.dx = (bf16_t*)dx->data;
    p.w = w; p.dw = dw; p.N = N; p.K = K; p.ldg = g->ld; p.ldy = y->ld; p.ldx = x->ld; p.lddx = dx->ld;
    p.G = tensor_groups(g);                                       // BN(y) constants and sums are per statistic group
    if (g->n % p.G) return ISA_EINVAL;
    p.M = (long)(g->n / p.G) * g->h * g->w;
    p.ysc = ybn->scale; p.ysh = ybn->shift; p.ymu = ybn->mean; p.yis = ybn->invstd; p.yred = ybn->red;
    p.ycnt_inv = 1.f / ybn->count; p.yact = ybn->act; p.ydgamma = ybn->dgamma; p.ydbeta = ybn->dbeta;
    p.xsc = xp.scale; p.xsh = xp.shift; p.xact = xp.act;
    p.xmu = xbn ? xbn->mean : nullptr; p.xis = xbn ? xbn->invstd : nullptr; p.xred = xbn ? xbn->out_red : nullptr;
    p.accumulate = accumulate; p.ws = ws;
    if (addend) {
        if (!tensor_ok(addend, 8) || addend->dtype != ISA_BF16 || addend->c != K || addend->n != g->n || addend->h != g->h ||
            addend->w != g->w) return ISA_EINVAL;
        p.addend = (const bf16_t*)addend->data; p.lda = addend->ld;
    }
    int xmode = 0;
    if (!pro_trivial(xp) || xbn) xmode = (xbn && xp.act == ISA_ACT_RELU6) ? 1 : 2;
    hipStream_t s = as_stream(stream);
    const int tn = (N + 31) / 32, tk = (K + 31) / 32;
    if (tn == 1 && tk == 1) return launch_tile<1, 1>(p, xmode, ws_floats, defer, s);
    if (tn == 1 && tk == 2) return launch_tile<1, 2>(p, xmode, ws_floats, defer, s);
    if (tn == 2 && tk == 1) return launch_tile<2, 1>(p, xmode, ws_floats, defer, s);
    return launch_tile<2, 2>(p, xmode, ws_floats, defer, s);
}
