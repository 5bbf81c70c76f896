// Whole-block forward fusion of InvertedV1Residual in eval mode (MobileNetDenseASPP.py:68-93 under model.eval()):
//     y = BN2(pw1x1(ReLU6(BN1(dw3x3(x))))) (+ x)
// in ONE pass over HBM: the depthwise output - the block's expanded tensor - lives only in LDS.  The op-granular path
// (isa_dwconv3x3 + isa_conv_gemm_ep) writes it (C channels) and reads it back; here a block costs read C + write C'
// (+ the residual read, an L2 hit on the tile just staged) instead of 3C + C': 2.3x fewer bytes for the 64 -> 32 blocks
// of the 256x256 level.  Eval-mode BatchNorm is a constant per-channel affine (the cached running statistics), so no
// batch statistics separate the stages; the train-mode form of the same fusion needs a statistics pre-pass and a
// recomputing backward.
//
// gfx950 structure: a workgroup (256 threads) owns 8 x 32-pixel tiles, persistent over tiles.
//   * per 32-channel block: the (8+2) x (32+2) halo is staged in LDS in its storage type (16-byte global loads land
//     unconverted; the NEXT block's or tile's loads are already in registers while the current one is computed; an fp32
//     halo cost 49 KB and left one workgroup per CU: 92 -> 85 us for the 64 -> 32 block at 256x256 x 16), the stencil runs
//     as in dwconv_tiled.hip (lane = 8 channels x 4 consecutive x, 18 LDS row vectors feed 36 packed FMAs), BN1 + ReLU6 are
//     applied to the
//     accumulators and the 256 x 32 result goes to the A tile in LDS as bf16, rows padded by 16 B;
//   * after the last channel block: A[256 px][C] x W[C'][C] on v_mfma_f32_32x32x16_bf16 (a wave owns two 32-pixel rows
//     of the tile; both operands are conflict-free ds_read_b128 row reads, W staged once per workgroup), BN2 affine on
//     the accumulators (lane = output channel), transpose through a per-wave LDS stage, residual add and 16-byte row
//     stores.
#include "common.hpp"

namespace {

constexpr int TH = 8, TW = 32, CB = 32;
// halo pixel stride: bf16 tiles 40 elements (80 B: the four x-groups of a ds_read_b128 phase tile 256 B), fp32 tiles 36 floats
template <typename HT> struct halo_stride { static constexpr int v = 36; };
template <> struct halo_stride<bf16_t> { static constexpr int v = 40; };
constexpr int HALO = (TH + 2) * (TW + 2);
constexpr int NIT = (HALO * 4 + 255) / 256;              // 16-byte halo slots per thread and channel block

struct DwPwParams {
    const bf16_t* x; int n, h, w, c, ldx;
    const bf16_t* wdw; int wld;                          // depthwise taps [9][wld]
    const float *s1, *h1;                                // BN1 scale / shift [c]
    const bf16_t* wpw; int kp;                           // 1x1 weights [N][kp]
    const float *s2, *h2; int act2;                      // BN2 scale / shift [N], activation after it
    const bf16_t* res; int ldres;
    bf16_t* y; int N, ldy;
    int tiles_x, tiles_y; long ntiles;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fma8(const float (&x)[8], const float (&w)[8], float (&acc)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 r = __builtin_elementwise_fma(f32x2{x[j], x[j + 1]}, f32x2{w[j], w[j + 1]}, f32x2{acc[j], acc[j + 1]});
        acc[j] = r[0]; acc[j + 1] = r[1];
    }
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}

// HT: type of the halo tile in LDS.  float for C = 32 (49 KB: two workgroups per CU still fit and the stencil reads need no
// bf16 -> fp32 conversions - 4.5 per output element, as many instructions as its packed FMAs), bf16 for wider inputs.
template <int NT, typename HT>   // output channels = 32 * NT
__global__ __launch_bounds__(256) void dwpw_eval_kernel(DwPwParams p) {
    constexpr int PS = halo_stride<HT>::v;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = p.c, LDA = C + 8, LDW = C + 8;          // bf16 elements; +16 B keeps ds_read_b128 rows conflict-free
    HT* halo = reinterpret_cast<HT*>(smem);                                         // [HALO][PS]; the epilogue's stage later
    bf16_t* sA = reinterpret_cast<bf16_t*>(halo + HALO * PS);                       // [256][LDA]
    bf16_t* sW = sA + 256 * LDA;                                                    // [32 NT][LDW]
    float* wts = reinterpret_cast<float*>(sW + 32 * NT * LDW);                      // [9][C]
    float* bn1 = wts + 9 * C;                                                       // [2][C]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int cg = tid & 3, g = tid >> 2, row = g >> 3, x0 = (g & 7) * 4;
    const int ncb = C / CB;

    for (int i = tid; i < 9 * C; i += 256) { const int tp = i / C, cc = i - tp * C; wts[i] = (float)p.wdw[(long)tp * p.wld + cc]; }
    for (int i = tid; i < C; i += 256) { bn1[i] = p.s1[i]; bn1[C + i] = p.h1[i]; }
    for (int i = tid; i < 32 * NT * (C / 8); i += 256) {                            // 16-byte pieces of the 1x1 weights
        const int n = i / (C / 8), k8 = (i - n * (C / 8)) * 8;
        bf16x8 v = bf16x8{0};
        if (n < p.N) v = *reinterpret_cast<const bf16x8*>(p.wpw + (long)n * p.kp + k8);
        *reinterpret_cast<bf16x8*>(sW + n * LDW + k8) = v;
    }

    // tile-invariant halo slot geometry
    int hpix[NIT], hoff[NIT], hrc[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pix = (tid + it * 256) >> 2;
        const int rr = pix / (TW + 2), cc = pix - rr * (TW + 2);
        hpix[it] = pix < HALO ? pix : -1;
        hrc[it] = (rr << 16) | cc;
        hoff[it] = (rr * p.w + cc) * p.ldx + cg * 8;
    }
    // flat sequence of (tile, channel block) steps of this workgroup; the loads of step s + 1 are in registers while
    // step s is computed
    const long nsteps_total = p.ntiles * ncb;
    auto step_coords = [&](long s, int& b, int& ty, int& tx, int& cb) {
        const long t = s / ncb; cb = (int)(s - t * ncb);
        tx = (int)(t % p.tiles_x); const long q = t / p.tiles_x; ty = (int)(q % p.tiles_y); b = (int)(q / p.tiles_y);
    };
    bf16x8 regs[NIT]; bool ok[NIT];
    auto issue = [&](long s) {
        int b, ty, tx, cb; step_coords(s, b, ty, tx, cb);
        const bf16_t* base = p.x + (((long)b * p.h + ty * TH - 1) * p.w + tx * TW - 1) * p.ldx + cb * CB;
        const int rlo = 1 - ty * TH, rhi = p.h + 1 - ty * TH, clo = 1 - tx * TW, chi = p.w + 1 - tx * TW;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rr = hrc[it] >> 16, cc = hrc[it] & 0xffff;
            ok[it] = hpix[it] >= 0 && rr >= rlo && rr < rhi && cc >= clo && cc < chi;
            regs[it] = bf16x8{0};
            if (ok[it]) regs[it] = *reinterpret_cast<const bf16x8*>(base + hoff[it]);
        }
        asm volatile("" ::: "memory");
    };
    auto stash = [&]() {                                   // registers -> halo tile, unconverted (zero outside the image)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (hpix[it] < 0) continue;
            if constexpr (sizeof(HT) == 2) *reinterpret_cast<bf16x8*>(halo + hpix[it] * PS + cg * 8) = regs[it];
            else {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (float)regs[it][j];
                store8<float>(reinterpret_cast<float*>(halo) + hpix[it] * PS + cg * 8, o);
            }
        }
    };

    long s = blockIdx.x;                                   // this workgroup's tiles: blockIdx.x, + gridDim.x, ... (ncb steps each)
    const long tstride = gridDim.x;
    long tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    s = tile * ncb;
    issue(s);
    __syncthreads();                                       // weights, constants
    while (true) {
        int b, ty, tx, cb; step_coords(s, b, ty, tx, cb);
        stash();
        __syncthreads();
        // next step of this workgroup: the next channel block of the tile, or the first block of its next tile
        long sn = s + 1;
        if (cb == ncb - 1) sn = (tile + tstride) * ncb;
        const bool more = sn < nsteps_total;
        if (more) issue(sn);
        {   // ---- depthwise stencil on channel block cb, BN1 + ReLU6, bf16 into the A tile
            const float* wc = wts + cb * CB + cg * 8;
            float acc[4][8];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy) {
                float wr[3][8], in[6][8];
#pragma unroll
                for (int k = 0; k < 3; ++k) ld8(wc + (dy * 3 + k) * C, wr[k]);
#pragma unroll
                for (int k = 0; k < 6; ++k) load8<HT>(halo + ((row + dy) * (TW + 2) + x0 + k) * PS + cg * 8, in[k]);
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int k = 0; k < 3; ++k) fma8(in[o + k], wr[k], acc[o]);
            }
            float sc[8], sh[8];
            ld8(bn1 + cb * CB + cg * 8, sc); ld8(bn1 + C + cb * CB + cg * 8, sh);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float raw = (float)(bf16_t)acc[o][j];        // the op-granular path stores the raw output as bf16
                    v[j] = (bf16_t)__builtin_amdgcn_fmed3f(fmaf(raw, sc[j], sh[j]), 0.f, 6.f);
                }
                *reinterpret_cast<bf16x8*>(sA + (row * TW + x0 + o) * LDA + cb * CB + cg * 8) = v;
            }
        }
        __syncthreads();                                   // halo consumed (it is restaged next), A block written
        if (cb == ncb - 1) {
            // ---- 1x1 conv on the finished A tile: wave w owns tile rows 2w, 2w + 1 (32 pixels each)
            // one 32-pixel tile row at a time: NT accumulators live (two rows at once put the 64-channel variant at 264
            // registers, one wave per SIMD)
            const bf16_t* tB = sW + r * LDW + 8 * hh;
            float* stage = reinterpret_cast<float*>(halo) + wave * (32 * 33);
#pragma unroll 1
            for (int i = 0; i < 2; ++i) {
                f32x16 acc[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = f32x16{0};
                const bf16_t* tA = sA + (wave * 64 + i * 32 + r) * LDA + 8 * hh;
                for (int k = 0; k < C; k += 16) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(tA + k);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const bf16x8 bb = *reinterpret_cast<const bf16x8*>(tB + j * 32 * LDW + k);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[j], 0, 0, 0);
                    }
                }
                // ---- epilogue: BN2 affine (lane r <-> output channel), transpose through the (idle) halo area, residual, stores
                const int trow = wave * 2 + i;                                   // tile row = image row ty*8 + trow
                const long pix0 = ((long)b * p.h + ty * TH + trow) * p.w + tx * TW;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = j * 32 + r;
                    const float es = n < p.N ? p.s2[n] : 1.f, eh = n < p.N ? p.h2[n] : 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rw = (e & 3) + 8 * (e >> 2) + 4 * hh;
                        stage[rw * 33 + r] = act_apply(fmaf(acc[j][e], es, eh), p.act2);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int half = 0; half < 2; ++half) {         // four lanes per pixel row: contiguous 32-channel pieces
                        const int px = (lane >> 2) + 16 * half, cseg = (lane & 3) * 8;
                        const int nseg = j * 32 + cseg;
                        if (nseg < p.N) {
                            float v[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] = stage[px * 33 + cseg + q];
                            if (p.res) {
                                float r0[8];
                                load8<bf16_t>(p.res + (pix0 + px) * p.ldres + nseg, r0);
#pragma unroll
                                for (int q = 0; q < 8; ++q) v[q] += r0[q];
                            }
                            store8<bf16_t>(p.y + (pix0 + px) * p.ldy + nseg, v);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            tile += tstride;
            __syncthreads();                               // the stage area is the halo tile: restaged next
        }
        if (!more) break;
        s = sn;
    }
}

template <int NT, typename HT>
int launch_dwpw(const DwPwParams& p, hipStream_t s) {
    const int C = p.c;
    const size_t lds = (size_t)HALO * halo_stride<HT>::v * sizeof(HT) + (size_t)256 * (C + 8) * 2 + (size_t)32 * NT * (C + 8) * 2 + (size_t)9 * C * 4 + (size_t)2 * C * 4;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dwpw_eval_kernel<NT, HT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess) return ISA_ELAUNCH;
        configured = true;
    }
    if (lds > 150 * 1024) return ISA_EINVAL;
    const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
    long gx = 256L * (per_cu > 3 ? 3 : per_cu);
    if (gx > p.ntiles) gx = p.ntiles;
    hipLaunchKernelGGL((dwpw_eval_kernel<NT, HT>), dim3((unsigned)gx), dim3(256), lds, s, p);
    return launch_status();
}

}  // namespace

extern "C" int isa_dwpw_eval(const isa_tensor* x, const void* w_dw, const float* bn1_scale, const float* bn1_shift,
                             const void* w_pw, int32_t kp, const isa_conv_ep* ep, const isa_tensor* y, void* stream) {
    if (!tensor_ok(x, 8) || !tensor_ok(y, 8) || !w_dw || !bn1_scale || !bn1_shift || !w_pw || !ep || !ep->scale || !ep->shift)
        return ISA_EINVAL;
    if (x->dtype != ISA_BF16 || y->dtype != ISA_BF16) return ISA_EDTYPE;
    if (x->n != y->n || x->h != y->h || x->w != y->w) return ISA_EINVAL;
    // shapes this kernel is built for (the 256x256 ... 32x32 levels of the backbone): whole tiles, whole channel blocks
    if (x->c % CB || x->c > 128 || y->c % 8 || y->c > 64 || x->h % TH || x->w % TW || kp < x->c || kp % 32) return ISA_EINVAL;
    DwPwParams p{};
    p.x = (const bf16_t*)x->data; p.n = x->n; p.h = x->h; p.w = x->w; p.c = x->c; p.ldx = x->ld;
    p.wdw = (const bf16_t*)w_dw; p.wld = ((x->c + 7) / 8) * 8;
    p.s1 = bn1_scale; p.h1 = bn1_shift;
    p.wpw = (const bf16_t*)w_pw; p.kp = kp;
    p.s2 = ep->scale; p.h2 = ep->shift; p.act2 = ep->act;
    if (ep->res) {
        const isa_tensor* r = ep->res;
        if (!tensor_ok(r, 8) || r->dtype != ISA_BF16 || r->n != y->n || r->h != y->h || r->w != y->w || r->c != y->c) return ISA_EINVAL;
        p.res = (const bf16_t*)r->data; p.ldres = r->ld;
    }
    p.y = (bf16_t*)y->data; p.N = y->c; p.ldy = y->ld;
    p.tiles_x = x->w / TW; p.tiles_y = x->h / TH;
    p.ntiles = (long)x->n * p.tiles_x * p.tiles_y;
    // fp32 halo where it costs no occupancy: C = 32 (two workgroups per CU either way) and C = 128 (one either way)
    if (x->c == 32 || x->c == 128) return y->c <= 32 ? launch_dwpw<1, float>(p, as_stream(stream)) : launch_dwpw<2, float>(p, as_stream(stream));
    return y->c <= 32 ? launch_dwpw<1, bf16_t>(p, as_stream(stream)) : launch_dwpw<2, bf16_t>(p, as_stream(stream));
}
