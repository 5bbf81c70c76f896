// The reference's QK^T / softmax / .V operators (modules/utils.py), dead at HEAD but named by the task:
//   a19  ScaledDotProductAttention.forward (utils.py:316-327) as configured by MultiHeadAttention
//        (utils.py:167-225): few queries (one query point) against L = H*W keys, d_k = d_v = 12.
//   a20  _ScalePDAttention.forward core (utils.py:276-299): per-pixel softmax over a 3x3 dilated
//        neighbourhood (9 keys) and the weighted sum of the 9 values.
//   a21  Decoder.forward (utils.py:59-69): sigmoid(<q_b, enc[b,:,p]>) for every pixel.
// All three are ~1 FLOP/byte: pure HBM streaming, no MFMA (the contraction is 12-24 deep per key).
//
// a19 design (gfx950): one 256-thread workgroup per (batch*head, query).  K and V are streamed ONCE:
// tiles of 256 keys x d are staged in LDS with 16-byte coalesced loads (rows are only 24-48 bytes, so
// a lane-per-key global access would touch 64 lines per instruction), then each lane scores its key
// from LDS and folds it into a private online softmax state (running max m, sum s, acc[d_v]).  The 256
// states are merged with wave shuffles (64 -> 1) and a 4-entry LDS exchange.  Raw scores go to the
// attn output on the fly and are normalised by a second, L-float pass (the reference returns attn).
#include "common.hpp"

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

extern "C" int isa_sdp_attention(const void* q, const void* k, const void* v, const uint8_t* mask, void* out,
                                 float* attn, int32_t bh, int32_t lq, int64_t L, int32_t dk, int32_t dv,
                                 float temperature, int32_t dtype, void* stream) {
    if (!q || !k || !v || !out || bh <= 0 || lq <= 0 || L <= 0 || dk <= 0 || dk > DMAX || dv <= 0 || dv > DMAX ||
        temperature <= 0.f) return ISA_EINVAL;
    SdpParams p{q, k, v, mask, out, attn, lq, dk, dv, (long)L, 1.f / temperature};
    dim3 grid(bh, lq);
    if (dtype == ISA_BF16) hipLaunchKernelGGL(sdp_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), p);
    else if (dtype == ISA_F32) hipLaunchKernelGGL(sdp_kernel<float>, grid, dim3(256), 0, as_stream(stream), p);
    else return ISA_EDTYPE;
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
