// The reference's QK^T / softmax / .V operators (modules/utils.py), dead at HEAD but named by the task:
//   a19  ScaledDotProductAttention.forward (utils.py:316-327) as configured by MultiHeadAttention
//        (utils.py:167-225): few queries (one query point) against L = H*W keys, d_k = d_v = 12.
//   a20  _ScalePDAttention.forward core (utils.py:276-299): per-pixel softmax over a 3x3 dilated
//        neighbourhood (9 keys) and the weighted sum of the 9 values.
//   a21  Decoder.forward (utils.py:59-69): sigmoid(<q_b, enc[b,:,p]>) for every pixel.
// All three are ~1 FLOP/byte: pure HBM streaming, no MFMA (the contraction is 12-24 deep per key).
//
// a19 design (gfx950): the production kernel is sdp_split_kernel below - split-L streaming ("flash decoding") for the
// reference's head width d = 12: the keys of a batch element are split over workgroups, K and V stream ONCE through an
// LDS tile with the next tile's 16-byte loads in registers (rows are only 24-48 bytes: a lane-per-key global access
// would touch 64 lines per instruction), lane = key with a private online-softmax state, a merge kernel combines the
// partials.  sdp_kernel (first below) is the one-workgroup-per-query form kept ONLY for other head widths (d <= 32,
// d != 12, heads = 1), which the reference's configuration (config.py:22-25) never produces; it is reached through the
// same entry point and covered by the d = 16 / 32 cases of tests/test_gpu_byname.py.
#include "common.hpp"

typedef _Float16 f16_t;
template <> struct st<f16_t> {
    static constexpr int dtype = 2;
    static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    static __device__ __forceinline__ void stv(f16_t* p, float v) { *p = (f16_t)v; }
};

namespace {

constexpr int KT = 256;         // keys per LDS tile
constexpr int DMAX = 32;        // supported head dim (reference: 12)

struct SdpParams {
    const void* q; const void* k; const void* v; const uint8_t* mask;   // mask[bh, lq, L] != 0 => masked
    void* out; float* attn;
    int lq, dk, dv; long L; float inv_temp;
};

template <typename T>
__global__ __launch_bounds__(256) void sdp_kernel(SdpParams p) {
    __shared__ __attribute__((aligned(16))) float sK[KT * DMAX];
    __shared__ __attribute__((aligned(16))) float sV[KT * DMAX];
    __shared__ float sQ[DMAX];
    __shared__ float sM[4], sS[4], sO[4][DMAX];
    const int bh = blockIdx.x, qi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* kp = reinterpret_cast<const T*>(p.k) + (long)bh * p.L * p.dk;
    const T* vp = reinterpret_cast<const T*>(p.v) + (long)bh * p.L * p.dv;
    const uint8_t* mk = p.mask ? p.mask + ((long)bh * p.lq + qi) * p.L : nullptr;
    float* at = p.attn ? p.attn + ((long)bh * p.lq + qi) * p.L : nullptr;
    if (tid < p.dk) sQ[tid] = st<T>::ld(reinterpret_cast<const T*>(p.q) + ((long)bh * p.lq + qi) * p.dk + tid) * p.inv_temp;
    float m = -INFINITY, s = 0.f, o[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) o[j] = 0.f;
    for (long l0 = 0; l0 < p.L; l0 += KT) {
        const int nk = (int)min((long)KT, p.L - l0);
        __syncthreads();
        // stage the tile: the [nk, d] slabs are contiguous in memory -> fully coalesced element loads
        for (int i = tid; i < nk * p.dk; i += 256) sK[i] = st<T>::ld(kp + l0 * p.dk + i);
        for (int i = tid; i < nk * p.dv; i += 256) sV[i] = st<T>::ld(vp + l0 * p.dv + i);
        __syncthreads();
        if (tid < nk) {
            const long l = l0 + tid;
            float sc = 0.f;
#pragma unroll
            for (int j = 0; j < DMAX; ++j) if (j < p.dk) sc = fmaf(sQ[j], sK[tid * p.dk + j], sc);
            const bool masked = mk && mk[l] != 0;
            if (masked) sc = -INFINITY;
            if (at) at[l] = sc;
            if (!masked) {
                const float mn = fmaxf(m, sc);
                const float a = expf(m - mn), e = expf(sc - mn);      // exp(-inf)=0 on the first key
                s = s * a + e;
#pragma unroll
                for (int j = 0; j < DMAX; ++j) if (j < p.dv) o[j] = o[j] * a + e * sV[tid * p.dv + j];
                m = mn;
            }
        }
    }
    // merge the 256 online-softmax states: wave shuffle tree, then 4 partials through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
        const float mn = fmaxf(m, m2);
        const float a = (m == -INFINITY) ? 0.f : expf(m - mn), b = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
        s = s * a + s2 * b;
#pragma unroll
        for (int j = 0; j < DMAX; ++j) if (j < p.dv) o[j] = o[j] * a + __shfl_xor(o[j], off, 64) * b;
        m = mn;
    }
    if (lane == 0) {
        sM[wave] = m; sS[wave] = s;
#pragma unroll
        for (int j = 0; j < DMAX; ++j) if (j < p.dv) sO[wave][j] = o[j];
    }
    __syncthreads();
    float M = -INFINITY;
    for (int w = 0; w < 4; ++w) M = fmaxf(M, sM[w]);
    float S = 0.f;
    for (int w = 0; w < 4; ++w) S += (sM[w] == -INFINITY) ? 0.f : sS[w] * expf(sM[w] - M);
    if (tid < p.dv) {
        float acc = 0.f;
        for (int w = 0; w < 4; ++w) acc += (sM[w] == -INFINITY) ? 0.f : sO[w][tid] * expf(sM[w] - M);
        // all keys masked: softmax of all -inf is NaN in the reference (utils.py:323-325); keep that
        st<T>::stv(reinterpret_cast<T*>(p.out) + ((long)bh * p.lq + qi) * p.dv + tid, acc / S);
    }
    if (at) {
        __syncthreads();
        const float invS = 1.f / S;
        for (long l = tid; l < p.L; l += 256) at[l] = expf(at[l] - M) * invS;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// a19, the north-star form: split-L streaming attention for FEW queries against L = H*W keys (65 536 at 256x256,
// 1 048 576 at 1024x1024), d_k = d_v = D.
//   * grid = (nsplit, batch): every workgroup owns a contiguous range of keys of one batch element and ALL heads of
//     it (MultiHeadAttention projects K/V to [b, L, n_head*d]: the heads are interleaved inside a row, so a head-major
//     permute - utils.py:200-202 - is never materialised; G = heads per row, G = 1 for plain [bh, L, d] operands).
//     nsplit is chosen so that >= ~4 workgroups per CU exist: one query per image used to run as 32 workgroups.
//   * K/V tiles of KT keys are staged in LDS in their storage type with 16-byte coalesced loads (the tile is one
//     contiguous slab of KT*G*D elements); the NEXT tile's loads are issued before the current tile is scored, wait in
//     registers and are written to the (single) LDS buffer afterwards (issue-early / write-late): HBM latency overlaps
//     the arithmetic and the small LDS footprint lets 6 workgroups share a CU;
//   * lane = key: the lane reads its row with ds_read_b128 / b64 (row stride G*D elements: conflict-free for the 24-
//     and 48-byte rows of D = 12), scores it against the queries held in registers and folds it into a private
//     online-softmax state (m, s, o[D]) per (head, query);
//   * the 256 states are merged with wave shuffles and a 4-entry LDS exchange; each workgroup writes ONE partial
//     (m, s, o) per (head, query); sdp_merge_kernel combines the nsplit partials (flash-decoding merge) and
//     sdp_attn_norm_kernel turns the raw scores left in `attn` into probabilities when the caller wants them.
//   * storage f32, bf16 or f16 (ISA_F16: the fp16 attention path of BASELINE configs[4]); accumulation fp32.
// 1 FLOP/byte: HBM-bound; algorithmic bytes = K and V once = 2*L*G*D*sizeof(T) per batch element.

constexpr int SDP_QMAX = 4;      // (heads per row) x (queries) handled per pass over K/V

struct SdpSplitParams {
    const void* q; const void* k; const void* v; const uint8_t* mask; float* attn; float* part;
    int B, lq; long L; float inv_temp;
    int mask_per_head;           // mask index = head*B + b (reference repeats the mask per head) or just b
    int q0;                      // first query of this pass
    long keys_per_split;
};

template <typename T, int D, int G, int NQ, int KT>
__global__ __launch_bounds__(256) void sdp_split_kernel(SdpSplitParams p) {
    constexpr int LD = G * D;                               // elements per K / V row
    constexpr int ROWB = LD * (int)sizeof(T);               // bytes per row
    constexpr int VW = ROWB % 16 == 0 ? 16 : (ROWB % 8 == 0 ? 8 : 4);
    constexpr int NV = ROWB / VW;
    constexpr int TILE_B = KT * ROWB;                       // bytes of one K (or V) tile
    constexpr int NLD = (TILE_B / 16 + 255) / 256;          // 16-byte vectors per thread and tile and operand
    constexpr int KPL = KT / 256;                           // keys per lane per tile
    static_assert(KT % 256 == 0 && TILE_B % 16 == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [K tile | V tile]: ONE buffer - the next tile waits in
                                                                  // registers, so twice as many workgroups (bytes in flight) fit a CU
    __shared__ float sM[4][G * NQ], sS[4][G * NQ], sO[4][G * NQ][D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, split = blockIdx.x;
    const long l_begin = (long)split * p.keys_per_split;
    const long l_end = min(p.L, l_begin + p.keys_per_split);
    const char* kg = reinterpret_cast<const char*>(p.k) + ((long)b * p.L) * ROWB;
    const char* vg = reinterpret_cast<const char*>(p.v) + ((long)b * p.L) * ROWB;
    // queries -> registers, pre-scaled by 1/temperature
    float qr[G * NQ][D];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int qi = min(p.q0 + i, p.lq - 1);
            const T* qp = reinterpret_cast<const T*>(p.q) + ((long)b * p.lq + qi) * LD + g * D;
#pragma unroll
            for (int j = 0; j < D; ++j) qr[g * NQ + i][j] = st<T>::ld(qp + j) * p.inv_temp;
        }
    float m[G * NQ], s[G * NQ], o[G * NQ][D];
#pragma unroll
    for (int r = 0; r < G * NQ; ++r) {
        m[r] = -INFINITY; s[r] = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) o[r][j] = 0.f;
    }
    f32x4 pk[NLD], pv[NLD];
    // mask bytes of the lane's own keys travel with the tile: requested a tile ahead like K and V (read at the point of
    // use they were a dependent global load in the scoring loop: +35 % on the whole kernel)
    uint8_t pmask[KPL][G * NQ], cmask[KPL][G * NQ];
#pragma unroll
    for (int u = 0; u < KPL; ++u)
#pragma unroll
        for (int r = 0; r < G * NQ; ++r) { pmask[u][r] = 0; cmask[u][r] = 0; }
    auto issue = [&](long l0) {                             // all 16-byte loads of one tile; clamped, never out of range
        const long lim = (l_end - l0) * ROWB;               // valid bytes of this tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            long off = ((long)tid + i * 256) * 16;
            if (off + 16 > lim) off = lim >= 16 ? lim - 16 : 0;     // tail lanes re-read the last vector (not used)
            pk[i] = *reinterpret_cast<const f32x4*>(kg + l0 * ROWB + off);
            pv[i] = *reinterpret_cast<const f32x4*>(vg + l0 * ROWB + off);
        }
        if (p.mask) {
#pragma unroll
            for (int u = 0; u < KPL; ++u) {
                const long l = min(l0 + tid + u * 256, l_end - 1);
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
                        const int qi = min(p.q0 + i, p.lq - 1);
                        pmask[u][g * NQ + i] = p.mask[((long)(p.mask_per_head ? g * p.B + b : b) * p.lq + qi) * p.L + l];
                    }
            }
        }
    };
    auto take_mask = [&]() {
#pragma unroll
        for (int u = 0; u < KPL; ++u)
#pragma unroll
            for (int r = 0; r < G * NQ; ++r) cmask[u][r] = pmask[u][r];
    };
    auto commit = [&]() {
        char* kt = smem; char* vt = kt + TILE_B;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int off = (tid + i * 256) * 16;
            if (off < TILE_B) {
                *reinterpret_cast<f32x4*>(kt + off) = pk[i];
                *reinterpret_cast<f32x4*>(vt + off) = pv[i];
            }
        }
    };
    auto row = [&](const char* base, float (&dst)[LD]) {    // one row, vector reads, converted to fp32
        T tmp[LD];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (VW == 16) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(tmp) + i * 16) = *reinterpret_cast<const f32x4*>(base + i * 16);
            else if constexpr (VW == 8) *reinterpret_cast<long*>(reinterpret_cast<char*>(tmp) + i * 8) = *reinterpret_cast<const long*>(base + i * 8);
            else *reinterpret_cast<int*>(reinterpret_cast<char*>(tmp) + i * 4) = *reinterpret_cast<const int*>(base + i * 4);
        }
#pragma unroll
        for (int j = 0; j < LD; ++j) dst[j] = (float)tmp[j];
    };
    if (l_begin < l_end) { issue(l_begin); commit(); }
    __syncthreads();
    for (long l0 = l_begin; l0 < l_end; l0 += KT) {
        const bool more = l0 + KT < l_end;
        take_mask();                                         // this tile's mask bytes (requested with its K / V)
        if (more) issue(l0 + KT);                            // in flight while this tile is scored
        const char* kt = smem; const char* vt = kt + TILE_B;
#pragma unroll
        for (int u = 0; u < KPL; ++u) {
            const int key = tid + u * 256;
            const long l = l0 + key;
            if (l < l_end) {
                float kr[LD], vr[LD];
                row(kt + key * ROWB, kr);
                row(vt + key * ROWB, vr);
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
                        const int r = g * NQ + i, qi = p.q0 + i;
                        if (qi >= p.lq) continue;
                        float sc = 0.f;
#pragma unroll
                        for (int j = 0; j < D; ++j) sc = fmaf(qr[r][j], kr[g * D + j], sc);
                        const bool masked = cmask[u][r] != 0;
                        if (masked) sc = -INFINITY;
                        if (p.attn) p.attn[((long)(g * p.B + b) * p.lq + qi) * p.L + l] = sc;
                        if (!masked) {
                            // online softmax with a LAZY rescale: a key that does not raise the lane's running maximum (all
                            // but ~ln(keys per lane) of them on real scores) costs one exp and D fused multiply-adds; only a
                            // new maximum rescales s and o.  The unconditional form (two exps, a multiply and an FMA per
                            // element of o for every key) made the bf16 / f16 kernel instruction-bound at 3.6 TB/s.
                            if (sc > m[r]) {
                                const float a = __expf(m[r] - sc);                       // exp(-inf) = 0 on the first key
                                s[r] = s[r] * a + 1.f;
#pragma unroll
                                for (int j = 0; j < D; ++j) o[r][j] = o[r][j] * a + vr[g * D + j];
                                m[r] = sc;
                            } else {
                                const float e = __expf(sc - m[r]);
                                s[r] += e;
#pragma unroll
                                for (int j = 0; j < D; ++j) o[r][j] = fmaf(e, vr[g * D + j], o[r][j]);
                            }
                        }
                    }
            }
        }
        __syncthreads();                                     // every lane has read its rows
        if (more) commit();
        __syncthreads();
    }
    // merge the 256 online-softmax states: wave shuffle tree, then 4 partials through LDS
#pragma unroll
    for (int r = 0; r < G * NQ; ++r) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float m2 = __shfl_xor(m[r], off, 64), s2 = __shfl_xor(s[r], off, 64);
            const float mn = fmaxf(m[r], m2);
            const float a = (m[r] == -INFINITY) ? 0.f : __expf(m[r] - mn), bb = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
            s[r] = s[r] * a + s2 * bb;
#pragma unroll
            for (int j = 0; j < D; ++j) o[r][j] = o[r][j] * a + __shfl_xor(o[r][j], off, 64) * bb;
            m[r] = mn;
        }
        if (lane == 0) {
            sM[wave][r] = m[r]; sS[wave][r] = s[r];
#pragma unroll
            for (int j = 0; j < D; ++j) sO[wave][r][j] = o[r][j];
        }
    }
    __syncthreads();
    // partial of this workgroup: part[((g*B + b)*lq + qi)][split][2 + D]
    if (tid < G * NQ * (2 + D)) {
        const int r = tid / (2 + D), c = tid - r * (2 + D);
        const int g = r / NQ, qi = p.q0 + (r - g * NQ);
        if (qi < p.lq) {
            float M = -INFINITY;
            for (int w = 0; w < 4; ++w) M = fmaxf(M, sM[w][r]);
            float val;
            if (c == 0) val = M;
            else {
                val = 0.f;
                for (int w = 0; w < 4; ++w) {
                    const float sc = (sM[w][r] == -INFINITY) ? 0.f : __expf(sM[w][r] - M);
                    val += (c == 1 ? sS[w][r] : sO[w][r][c - 2]) * sc;
                }
            }
            p.part[(((long)(g * p.B + b) * p.lq + qi) * gridDim.x + split) * (2 + D) + c] = val;
        }
    }
}

// combine the nsplit partials of one (head*B + b, query): out = sum_i o_i e^(m_i - M) / sum_i s_i e^(m_i - M)
// out layout: [b, lq, G*D] (heads interleaved, the layout MultiHeadAttention's fc consumes; == [bh, lq, D] for G = 1)
template <typename T>
__global__ __launch_bounds__(64) void sdp_merge_kernel(const float* part, int nsplit, int D, int G, int B, int lq, T* out, float* ms) {
    const int row = blockIdx.x;                              // (g*B + b)*lq + qi
    const int lane = threadIdx.x;
    const float* pr = part + (long)row * nsplit * (2 + D);
    float M = -INFINITY;
    for (int i = lane; i < nsplit; i += 64) M = fmaxf(M, pr[i * (2 + D)]);
    M = wave_max(M);
    // one pass over the lane's partials: the weight e^(m_i - M) is computed once and applied to s_i and all D values of o_i
    // (a pass per output element recomputed it D times and made this one-wave kernel 16 us long at 256 partials)
    float S = 0.f, acc[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) acc[j] = 0.f;
    for (int i = lane; i < nsplit; i += 64) {
        const float* q = pr + i * (2 + D);
        const float mi = q[0];
        if (mi == -INFINITY) continue;
        const float w = __expf(mi - M);
        S += q[1] * w;
#pragma unroll
        for (int j = 0; j < DMAX; ++j) if (j < D) acc[j] = fmaf(q[2 + j], w, acc[j]);
    }
    S = wave_sum(S);
    const int g = row / (B * lq), rem = row - g * B * lq, b = rem / lq, qi = rem - b * lq;
#pragma unroll
    for (int j = 0; j < DMAX; ++j) {
        if (j >= D) break;
        const float a = wave_sum(acc[j]);
        // all keys masked: softmax of all -inf is NaN in the reference (utils.py:323-325); 0/0 keeps that
        if (lane == 0) st<T>::stv(out + ((long)b * lq + qi) * (G * D) + g * D + j, a / S);
    }
    if (lane == 0 && ms) { ms[2 * row] = M; ms[2 * row + 1] = S; }
}

__global__ __launch_bounds__(256) void sdp_attn_norm_kernel(float* attn, const float* ms, long L) {
    const int row = blockIdx.y;
    const float M = ms[2 * row], inv = 1.f / ms[2 * row + 1];
    float* at = attn + (long)row * L;
    for (long l = ((long)blockIdx.x * 256 + threadIdx.x) * 4; l < L; l += (long)gridDim.x * 1024) {
        if (l + 4 <= L) {
            f32x4 v = *reinterpret_cast<f32x4*>(at + l);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __expf(v[j] - M) * inv;
            *reinterpret_cast<f32x4*>(at + l) = v;
        } else {
            for (long k = l; k < L; ++k) at[k] = __expf(at[k] - M) * inv;
        }
    }
}

// MultiHeadAttention `last=True` branch (utils.py:203-209,310-313): sigmoid(q . k^T), no temperature, no softmax.
// out[(g*B + b), qi, l]; same operand layout as sdp_split_kernel.
template <typename T>
__global__ __launch_bounds__(256) void sdp_sigmoid_kernel(const T* q, const T* k, int D, int G, int B, int lq, long L, float* out, int sig) {
    const int b = blockIdx.y;
    const int LD = G * D;
    for (long l = (long)blockIdx.x * 256 + threadIdx.x; l < L; l += (long)gridDim.x * 256) {
        const T* kr = k + ((long)b * L + l) * LD;
        for (int g = 0; g < G; ++g)
            for (int qi = 0; qi < lq; ++qi) {
                const T* qp = q + ((long)b * lq + qi) * LD + g * D;
                float a = 0.f;
                for (int j = 0; j < D; ++j) a = fmaf(st<T>::ld(qp + j), st<T>::ld(kr + g * D + j), a);
                out[((long)(g * B + b) * lq + qi) * L + l] = sig ? 1.f / (1.f + __expf(-a)) : a;
            }
    }
}

// Small dense layers of the attention wrappers: y[r, :] = LayerNorm(W x[r, :] + bias + residual[r, :]) (LayerNorm and
// residual optional): MultiHeadAttention's w_qs on the few query rows and fc + layer_norm on the attended rows
// (utils.py:189-201); R rows, K, N <= 64.  One wave per row.
__global__ __launch_bounds__(64) void linear_ln_kernel(const float* x, const float* w, const float* bias, const float* res,
                                                       const float* gamma, const float* beta, float eps, int K, int N,
                                                       float* y) {
    const int r = blockIdx.x, n = threadIdx.x;
    float acc = 0.f;
    if (n < N) {
        acc = bias ? bias[n] : 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(w[n * K + k], x[(long)r * K + k], acc);
        if (res) acc += res[(long)r * N + n];
    }
    if (gamma) {
        const float mean = wave_sum(n < N ? acc : 0.f) / N;
        const float d = n < N ? acc - mean : 0.f;
        const float var = wave_sum(d * d) / N;                       // biased, as nn.LayerNorm
        acc = d * rsqrtf(var + eps) * gamma[n < N ? n : 0] + (beta ? beta[n < N ? n : 0] : 0.f);
    }
    if (n < N) y[(long)r * N + n] = acc;
}

// InstanceNorm2d(out + residual) of _ScalePDAttention (utils.py:301-302; affine=False, biased variance, eps 1e-5)
template <typename T>
__global__ __launch_bounds__(256) void inorm_stats_kernel(const T* x, int ldx, const T* res, int ldr, int c, long hw, float* sums) {
    const int b = blockIdx.y;
    extern __shared__ float red[];                                   // [2c]
    for (int i = threadIdx.x; i < 2 * c; i += 256) red[i] = 0.f;
    __syncthreads();
    const int lanes_c = c;                                           // c <= 64
    const int rows = 256 / lanes_c, ch = threadIdx.x % lanes_c, rsub = threadIdx.x / lanes_c;
    if (rsub < rows) {
        float s1 = 0.f, s2 = 0.f;
        for (long pix = (long)blockIdx.x * rows + rsub; pix < hw; pix += (long)gridDim.x * rows) {
            const float v = st<T>::ld(x + (b * hw + pix) * ldx + ch) + st<T>::ld(res + (b * hw + pix) * ldr + ch);
            s1 += v; s2 += v * v;
        }
        atomicAdd(&red[ch], s1); atomicAdd(&red[c + ch], s2);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * c; i += 256) atomicAdd(sums + (long)b * 2 * c + i, red[i]);
}
template <typename T>
__global__ __launch_bounds__(256) void inorm_apply_kernel(const T* x, int ldx, const T* res, int ldr, int c, long hw, const float* sums,
                                                          float eps, T* out, int ldo, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % c); const long pix = i / c; const long b = pix / hw;
        const float mean = sums[b * 2 * c + ch] / hw;
        const float var = fmaxf(sums[b * 2 * c + c + ch] / hw - mean * mean, 0.f);
        const float v = st<T>::ld(x + pix * ldx + ch) + st<T>::ld(res + pix * ldr + ch);
        st<T>::stv(out + pix * ldo + ch, (v - mean) * rsqrtf(var + eps));
    }
}

// a20: local dilated attention on NHWC tensors.  q,k: [n,h,w,dk]  v: [n,h,w,dv]  nomask: fp32 [n,h*w]
struct LocalParams { const void* q; const void* k; const void* v; const float* nomask; void* out;
                     int n, h, w, dk, dv, ldq, ldk, ldv, ldo, d; };
template <typename T>
__global__ __launch_bounds__(256) void local_attn_kernel(LocalParams p) {
    const long pixels = (long)p.n * p.h * p.w;
    const float scale = rsqrtf((float)p.dk);
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < pixels; pix += (long)gridDim.x * 256) {
        const int x = (int)(pix % p.w); const long t = pix / p.w; const int y = (int)(t % p.h); const long b = t / p.h;
        float qv[DMAX];
        const T* qp = reinterpret_cast<const T*>(p.q) + pix * p.ldq;
#pragma unroll
        for (int j = 0; j < DMAX; ++j) qv[j] = j < p.dk ? st<T>::ld(qp + j) : 0.f;
        float lg[9]; long src[9];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int yy = y + (i / 3 - 1) * p.d, xx = x + (i % 3 - 1) * p.d;
            // zero padding (F.pad, utils.py:279-281): K = V = 0 and nomask = 0 -> logit 0, NOT masked
            if (yy < 0 || yy >= p.h || xx < 0 || xx >= p.w) { src[i] = -1; lg[i] = 0.f; }
            else {
                src[i] = (b * p.h + yy) * p.w + xx;
                const T* kp = reinterpret_cast<const T*>(p.k) + src[i] * p.ldk;
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < DMAX; ++j) if (j < p.dk) a = fmaf(qv[j], st<T>::ld(kp + j), a);
                lg[i] = p.nomask[src[i]] != 0.f ? -INFINITY : a * scale;
            }
            mx = fmaxf(mx, lg[i]);
        }
        float se = 0.f, pr[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { pr[i] = (mx == -INFINITY) ? 0.f : expf(lg[i] - mx); se += pr[i]; }
        const float inv = se > 0.f ? 1.f / se : 0.f;             // all 9 masked: NaN -> 0 (utils.py:296)
        T* op = reinterpret_cast<T*>(p.out) + pix * p.ldo;
        for (int j = 0; j < p.dv; ++j) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (src[i] >= 0) a = fmaf(pr[i] * inv, st<T>::ld(reinterpret_cast<const T*>(p.v) + src[i] * p.ldv + j), a);
            st<T>::stv(op + j, a);
        }
    }
}

// a21: out[b,p] = sigmoid(sum_c q[b,c] * enc[b,p,c])   (enc NHWC, out fp32 map)
template <typename T>
__global__ __launch_bounds__(256) void point_query_kernel(const float* q, const T* enc, int c, int ld, long hw, long pixels, float* out) {
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < pixels; pix += (long)gridDim.x * 256) {
        const long b = pix / hw;
        float a = 0.f;
        for (int j = 0; j < c; ++j) a = fmaf(q[b * c + j], st<T>::ld(enc + pix * ld + j), a);
        out[pix] = 1.f / (1.f + expf(-a));
    }
}

}  // namespace

template <typename T, int D, int G, int NQ>
static int sdp_launch_tile(const SdpSplitParams& p, int tile_keys, dim3 grid, hipStream_t s) {
#define SDP_KT(KT) do { \
        const size_t lds = (size_t)2 * KT * G * D * sizeof(T); \
        if (lds > 64 * 1024) { \
            static bool configured = false; \
            if (!configured) { \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&sdp_split_kernel<T, D, G, NQ, KT>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISA_ELAUNCH; \
                configured = true; \
            } \
        } \
        hipLaunchKernelGGL((sdp_split_kernel<T, D, G, NQ, KT>), grid, dim3(256), lds, s, p); \
        return launch_status(); \
    } while (0)
    switch (tile_keys) {
        case 256: SDP_KT(256);
        case 512: SDP_KT(512);
        case 1024: SDP_KT(1024);
        default: return ISA_EINVAL;
    }
#undef SDP_KT
}

template <typename T>
static int sdp_fast(const void* q, const void* k, const void* v, const uint8_t* mask, void* out, float* attn, int B, int G,
                    int lq, long L, int D, float temperature, int mask_per_head, float* ws, long ws_floats, int tile_keys,
                    hipStream_t s) {
    if (D != 12 || (G != 1 && G != 2)) return 1;                      // not a fast-path shape: caller falls back
    if (tile_keys == 0) tile_keys = 512;
    while (tile_keys > 256 && (size_t)2 * tile_keys * G * D * sizeof(T) > 160 * 1024) tile_keys /= 2;   // one K + one V tile in LDS
    const long rows = (long)G * B * lq;
    // splits: >= ~4 workgroups per CU, whole tiles per split, bounded by the workspace
    long nsplit = (1024 + B - 1) / B;
    // at least 4096 keys (16 per lane) per workgroup: below that the fixed cost of a workgroup (query setup, the shuffle
    // merge of 256 online-softmax states) outweighs its keys - measured 1024 / 2048 / 4096 / 8192: 77 / 49 / 35 / 39 us at
    // L = 1M, batch 1 and 32 / 29 / 28 / 38 us at L = 65 536, batch 16 (bf16)
    long min_keys = tile_keys > 4096 ? tile_keys : 4096;
    const long max_split = (L + min_keys - 1) / min_keys;
    if (nsplit > max_split) nsplit = max_split;
    const long per_row = 2 + D;
    while (nsplit > 1 && rows * nsplit * per_row + 2 * rows > ws_floats) nsplit /= 2;
    if (rows * nsplit * per_row + 2 * rows > ws_floats) return ISA_ENOMEM;
    long kps = (L + nsplit - 1) / nsplit;
    kps = (kps + tile_keys - 1) / tile_keys * tile_keys;
    nsplit = (L + kps - 1) / kps;
    float* part = ws; float* ms = ws + rows * nsplit * per_row;
    SdpSplitParams p{q, k, v, mask, attn, part, B, lq, L, 1.f / temperature, mask_per_head, 0, kps};
    dim3 grid((unsigned)nsplit, (unsigned)B);
    const int nq_per_pass = SDP_QMAX / G;
    for (int q0 = 0; q0 < lq; q0 += nq_per_pass) {
        p.q0 = q0;
        int rc;
        const int nq = lq - q0 < nq_per_pass ? lq - q0 : nq_per_pass;
        if (G == 1) rc = nq == 1 ? sdp_launch_tile<T, 12, 1, 1>(p, tile_keys, grid, s) : sdp_launch_tile<T, 12, 1, 4>(p, tile_keys, grid, s);
        else rc = nq == 1 ? sdp_launch_tile<T, 12, 2, 1>(p, tile_keys, grid, s) : sdp_launch_tile<T, 12, 2, 2>(p, tile_keys, grid, s);
        if (rc != ISA_OK) return rc;
    }
    hipLaunchKernelGGL(sdp_merge_kernel<T>, dim3((unsigned)rows), dim3(64), 0, s, part, (int)nsplit, D, G, B, lq, (T*)out, ms);
    if (attn) {
        const int gx = (int)((L + 1023) / 1024 < 64 ? (L + 1023) / 1024 : 64);
        hipLaunchKernelGGL(sdp_attn_norm_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, s, attn, ms, L);
    }
    return launch_status();
}

// heads: number of heads interleaved in a K / V / q row (MultiHeadAttention's projected layout [b, L, heads*d]); 1 for
// plain [bh, L, d] operands.  ws: fp32 workspace for the split partials (>= 64 K floats is plenty), tile_keys: keys per
// LDS tile (0 = default 512; 256 / 512 / 1024 for the tile-size sweep).
extern "C" int isa_sdp_attention(const void* q, const void* k, const void* v, const uint8_t* mask, void* out,
                                 float* attn, int32_t bh, int32_t lq, int64_t L, int32_t dk, int32_t dv,
                                 float temperature, int32_t dtype, int32_t heads, int32_t mask_per_head,
                                 float* ws, int64_t ws_floats, int32_t tile_keys, void* stream) {
    if (!q || !k || !v || !out || bh <= 0 || lq <= 0 || L <= 0 || dk <= 0 || dk > DMAX || dv <= 0 || dv > DMAX ||
        temperature <= 0.f || heads < 1) return ISA_EINVAL;
    if (dtype != ISA_F32 && dtype != ISA_BF16 && dtype != ISA_F16) return ISA_EDTYPE;
    hipStream_t s = as_stream(stream);
    if (dk == dv && ws) {
        int rc = 1;
        if (dtype == ISA_BF16) rc = sdp_fast<bf16_t>(q, k, v, mask, out, attn, bh, heads, lq, L, dk, temperature, mask_per_head, ws, ws_floats, tile_keys, s);
        else if (dtype == ISA_F16) rc = sdp_fast<f16_t>(q, k, v, mask, out, attn, bh, heads, lq, L, dk, temperature, mask_per_head, ws, ws_floats, tile_keys, s);
        else rc = sdp_fast<float>(q, k, v, mask, out, attn, bh, heads, lq, L, dk, temperature, mask_per_head, ws, ws_floats, tile_keys, s);
        if (rc != 1) return rc;
    }
    // general head dims (<= 32, contiguous [bh, L, d] operands only): one workgroup per (batch*head, query)
    if (heads != 1 || dtype == ISA_F16) return ISA_EINVAL;
    SdpParams p{q, k, v, mask, out, attn, lq, dk, dv, (long)L, 1.f / temperature};
    dim3 grid(bh, lq);
    if (dtype == ISA_BF16) hipLaunchKernelGGL(sdp_kernel<bf16_t>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(sdp_kernel<float>, grid, dim3(256), 0, s, p);
    return launch_status();
}


// ---------------------------------------------------------------------------------------------------------------
// Backward of the streaming attention core (autograd of utils.py:316-327 in eval mode), from what the forward pass
// already returns: attn (the normalised probabilities, zero on masked keys) and out.
//   D_q  = dO_q . out_q                      (= sum_j p_qj dP_qj, because out = sum_j p_qj v_j)
//   dP_qj = dO_q . v_j      dS_qj = p_qj (dP_qj - D_q)
//   dV_j = sum_q p_qj dO_q  dK_j = sum_q dS_qj q_q / T      dQ_q = sum_j dS_qj k_j / T
// One pass over K and V: a lane owns keys j, j + stride, ...; dK_j / dV_j are written once, the dQ partials of a lane
// live in registers (lq * d <= 96), are folded through LDS per workgroup and added to the fp32 dq with one atomic each.
constexpr int SDPB_MAXQD = 96;
template <typename T>
__global__ __launch_bounds__(256) void sdp_bwd_kernel(const T* q, const T* k, const T* v, const float* attn, const T* out, const T* dout,
                                                      float* dq, T* dk, T* dv, int B, int heads, int lq, long L, int d, int dvd,
                                                      float inv_t) {
    __shared__ float sq[SDPB_MAXQD], sdo[SDPB_MAXQD], sD[8];
    __shared__ float sS[8][256];                                  // dS of this chunk's keys, per query
    __shared__ float sK[256][33];                                 // the chunk's keys (fp32, +1 pad: conflict-free column reads)
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int ldq = heads * d, ldv = heads * dvd;
    const int tid = threadIdx.x;
    for (int i = tid; i < lq * d; i += 256) sq[i] = st<T>::ld(q + ((long)b * lq + i / d) * ldq + h * d + i % d);
    for (int i = tid; i < lq * dvd; i += 256) sdo[i] = st<T>::ld(dout + ((long)b * lq + i / dvd) * ldv + h * dvd + i % dvd);
    __syncthreads();
    if (tid < lq) {
        float s = 0.f;
        for (int e = 0; e < dvd; ++e) s += sdo[tid * dvd + e] * st<T>::ld(out + ((long)b * lq + tid) * ldv + h * dvd + e);
        sD[tid] = s;
    }
    __syncthreads();
    const float* arow = attn + ((long)h * B + b) * lq * L;          // attn rows are head-major ((h*B + b)*lq + q)
    const int qi_b = tid / d, e_b = tid - qi_b * d;                  // phase-B role: one (query, channel) of dQ
    float acc = 0.f;
    for (long j0 = (long)blockIdx.x * 256; j0 < L; j0 += (long)gridDim.x * 256) {
        const long j = j0 + tid;
        const bool live = j < L;
        float kj[32], vj[32], dkj[32], dvj[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            kj[e] = (live && e < d) ? st<T>::ld(k + ((long)b * L + j) * ldq + h * d + e) : 0.f;
            vj[e] = (live && e < dvd) ? st<T>::ld(v + ((long)b * L + j) * ldv + h * dvd + e) : 0.f;
            dkj[e] = 0.f; dvj[e] = 0.f;
            sK[tid][e] = kj[e];
        }
        for (int qi = 0; qi < lq; ++qi) {
            const float pj = live ? arow[(long)qi * L + j] : 0.f;
            float dP = 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) if (e < dvd) dP += sdo[qi * dvd + e] * vj[e];
            const float dS = pj * (dP - sD[qi]) * inv_t;
            sS[qi][tid] = dS;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                if (e < dvd) dvj[e] += pj * sdo[qi * dvd + e];
                if (e < d) dkj[e] += dS * sq[qi * d + e];
            }
        }
        if (live) {
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                if (e < d) st<T>::stv(dk + ((long)b * L + j) * ldq + h * d + e, dkj[e]);
                if (e < dvd) st<T>::stv(dv + ((long)b * L + j) * ldv + h * dvd + e, dvj[e]);
            }
        }
        __syncthreads();
        if (tid < lq * d) {
#pragma unroll 8
            for (int jj = 0; jj < 256; ++jj) acc += sS[qi_b][jj] * sK[jj][e_b];
        }
        __syncthreads();
    }
    if (tid < lq * d && acc != 0.f) atomicAdd(dq + ((long)b * lq + qi_b) * ldq + h * d + e_b, acc);
}

extern "C" int isa_sdp_attention_bwd(const void* q, const void* k, const void* v, const float* attn, const void* out,
                                     const void* dout, float* dq, void* dk, void* dv, int32_t bh, int32_t lq, int64_t L,
                                     int32_t dkd, int32_t dvd, float temperature, int32_t dtype, int32_t heads, void* stream) {
    if (!q || !k || !v || !attn || !out || !dout || !dq || !dk || !dv) return ISA_EINVAL;
    if (bh <= 0 || lq <= 0 || lq > 8 || L <= 0 || dkd <= 0 || dkd > 32 || dvd <= 0 || dvd > 32 || heads <= 0) return ISA_EINVAL;
    if (lq * dkd > SDPB_MAXQD || lq * dvd > SDPB_MAXQD || !(temperature > 0.f)) return ISA_EINVAL;
    long gx = (L + 255) / 256;
    const long cap = (2048 + (long)bh * heads - 1) / ((long)bh * heads);      // ~8 workgroups per CU in total (44 KB of LDS each)
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)(bh * heads));
    hipStream_t s = as_stream(stream);
    const float it = 1.f / temperature;
#define SDPB(T) hipLaunchKernelGGL(sdp_bwd_kernel<T>, grid, dim3(256), 0, s, (const T*)q, (const T*)k, (const T*)v, attn, (const T*)out, \
                                   (const T*)dout, dq, (T*)dk, (T*)dv, (int)bh, (int)heads, (int)lq, (long)L, (int)dkd, (int)dvd, it)
    if (dtype == ISA_F32) SDPB(float);
    else if (dtype == ISA_BF16) SDPB(bf16_t);
    else if (dtype == ISA_F16) SDPB(f16_t);
    else return ISA_EINVAL;
#undef SDPB
    return launch_status();
}

extern "C" int isa_sdp_scores(const void* q, const void* k, float* out, int32_t b, int32_t heads, int32_t lq, int64_t L,
                              int32_t d, int32_t dtype, int32_t sigmoid, void* stream) {
    if (!q || !k || !out || b <= 0 || heads < 1 || lq <= 0 || L <= 0 || d <= 0) return ISA_EINVAL;
    dim3 grid(grid_cap(cdiv(L, 256), 1024), b);
    hipStream_t s = as_stream(stream);
    if (dtype == ISA_BF16) hipLaunchKernelGGL(sdp_sigmoid_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k, d, heads, b, lq, (long)L, out, sigmoid);
    else if (dtype == ISA_F16) hipLaunchKernelGGL(sdp_sigmoid_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)q, (const f16_t*)k, d, heads, b, lq, (long)L, out, sigmoid);
    else if (dtype == ISA_F32) hipLaunchKernelGGL(sdp_sigmoid_kernel<float>, grid, dim3(256), 0, s, (const float*)q, (const float*)k, d, heads, b, lq, (long)L, out, sigmoid);
    else return ISA_EDTYPE;
    return launch_status();
}

extern "C" int isa_linear_ln(const float* x, const float* w, const float* bias, const float* residual, const float* gamma,
                             const float* beta, float eps, int32_t rows, int32_t k, int32_t n, float* y, void* stream) {
    if (!x || !w || !y || rows <= 0 || k <= 0 || n <= 0 || n > 64) return ISA_EINVAL;
    hipLaunchKernelGGL(linear_ln_kernel, dim3(rows), dim3(64), 0, as_stream(stream), x, w, bias, residual, gamma, beta, eps, k, n, y);
    return launch_status();
}

extern "C" int isa_instance_norm_res(const isa_tensor* x, const isa_tensor* res, const isa_tensor* out, float eps,
                                     float* sums /* zeroed [n][2c] */, void* stream) {
    if (!tensor_ok(x, 1) || !tensor_ok(res, 1) || !tensor_ok(out, 1) || !sums || x->c > 64 || res->c != x->c || out->c != x->c ||
        x->dtype != res->dtype || x->dtype != out->dtype || x->n != res->n || x->h != res->h || x->w != res->w) return ISA_EINVAL;
    const long hw = (long)x->h * x->w, total = hw * x->n * x->c;
    hipStream_t s = as_stream(stream);
    dim3 g1(grid_cap(cdiv(hw, 256), 256), x->n);
    const int g2 = grid_cap(cdiv(total, 256));
    if (x->dtype == ISA_BF16) {
        hipLaunchKernelGGL(inorm_stats_kernel<bf16_t>, g1, dim3(256), 2 * x->c * 4, s, (const bf16_t*)x->data, x->ld, (const bf16_t*)res->data, res->ld, x->c, hw, sums);
        hipLaunchKernelGGL(inorm_apply_kernel<bf16_t>, dim3(g2), dim3(256), 0, s, (const bf16_t*)x->data, x->ld, (const bf16_t*)res->data, res->ld, x->c, hw, sums, eps, (bf16_t*)out->data, out->ld, total);
    } else {
        hipLaunchKernelGGL(inorm_stats_kernel<float>, g1, dim3(256), 2 * x->c * 4, s, (const float*)x->data, x->ld, (const float*)res->data, res->ld, x->c, hw, sums);
        hipLaunchKernelGGL(inorm_apply_kernel<float>, dim3(g2), dim3(256), 0, s, (const float*)x->data, x->ld, (const float*)res->data, res->ld, x->c, hw, sums, eps, (float*)out->data, out->ld, total);
    }
    return launch_status();
}

extern "C" int isa_local_attention(const isa_tensor* q, const isa_tensor* k, const isa_tensor* v, const float* nomask,
                                   const isa_tensor* out, int32_t dilation, void* stream) {
    if (!tensor_ok(q, 1) || !tensor_ok(k, 1) || !tensor_ok(v, 1) || !tensor_ok(out, 1) || !nomask || dilation < 1 ||
        q->c != k->c || q->c > DMAX || v->c > DMAX || out->c != v->c || q->dtype != k->dtype || q->dtype != v->dtype ||
        q->dtype != out->dtype) return ISA_EINVAL;
    LocalParams p{q->data, k->data, v->data, nomask, out->data, q->n, q->h, q->w, q->c, v->c, q->ld, k->ld, v->ld, out->ld, dilation};
    const int grid = grid_cap(cdiv((long)q->n * q->h * q->w, 256));
    if (q->dtype == ISA_BF16) hipLaunchKernelGGL(local_attn_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(local_attn_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), p);
    return launch_status();
}

extern "C" int isa_point_query(const float* q, const isa_tensor* enc, float* out, void* stream) {
    if (!q || !tensor_ok(enc, 1) || !out) return ISA_EINVAL;
    const long hw = (long)enc->h * enc->w, pixels = hw * enc->n;
    const int grid = grid_cap(cdiv(pixels, 256));
    if (enc->dtype == ISA_BF16)
        hipLaunchKernelGGL(point_query_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), q, (const bf16_t*)enc->data, enc->c, enc->ld, hw, pixels, out);
    else
        hipLaunchKernelGGL(point_query_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), q, (const float*)enc->data, enc->c, enc->ld, hw, pixels, out);
    return launch_status();
}
