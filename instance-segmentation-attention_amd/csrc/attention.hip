// Attention mask head kernels (reference: modules/utils.py:457-663, attenet2.py:304-347):
// spatial additive attention with a masked softmax over all H*W positions, masked BatchNorm,
// per-instance hard attention (masked softmax over the instance's pixels), on-device sampling
// (argmax), pyramid targets / position-code channels and the attention gate.
//
// Shapes: feature tensors are NHWC views (T = bf16|f32).  Single-channel maps (scores, masks,
// probabilities, targets) are kept as plain fp32 [n, h*w] arrays: they are 1/24..1/512 of the
// traffic of the feature tensors they are derived from, so precision is free there and the
// softmax / argmax paths stay bit-stable across storage modes.
// Row-wide reductions (L = h*w = 65 536 at 256^2) use one 1024-thread workgroup per row:
// wave shuffles, then a 16-entry LDS exchange — the row (256 KB) lives in L2 across the passes.
#include "common.hpp"

namespace {

struct View { void* data; int n, h, w, c, ld; };
static inline View mkview(const isa_tensor* t) { return View{t->data, t->n, t->h, t->w, t->c, t->ld}; }

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}

// ---- a7 step 1: dot[p] = m[p]*<w, x[p,:]> + bias ; chansum[b,c] += m[p]*x[p,c] -----------------
template <typename T>
__global__ __launch_bounds__(256) void mask_dot_kernel(View x, const float* m, const float* w, const float* bias,
                                                       float* dot, float* chansum) {
    extern __shared__ float red[];      // [C]
    const int C = x.c, b = blockIdx.y;
    for (int i = threadIdx.x; i < C; i += 256) red[i] = 0.f;
    __syncthreads();
    const long hw = (long)x.h * x.w;
    const int cg = (C + 7) / 8;
    // lane -> (pixel, channel group): consecutive lanes walk consecutive 16-byte pieces of NHWC rows.  The host rounds
    // the grid so that the stride is a multiple of cg (grid_keep_cg): a lane keeps ONE channel group, and the cg
    // lanes of a pixel are adjacent.  Uniform trip count (tail lanes idle) so that shuffles see every lane.
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const long total = hw * cg, step = (long)gridDim.x * 256;
    const bool fixed = step % cg == 0 && cg < 64;
    const int cfix = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cg) * 8;
    int last = -1;
    for (long base = (long)blockIdx.x * 256; base < total; base += step) {
        const long item = base + threadIdx.x;
        const bool live = item < total;
        const int c0 = live ? (int)(item % cg) * 8 : cfix; const long pix = live ? item / cg : 0;
        if (!fixed && c0 != last && last >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { if (last + j < C) atomicAdd(&red[last + j], acc[j]); acc[j] = 0.f; }
        }
        last = c0;
        float part = 0.f;
        if (live) {
            float v[8];
            load8g<T>(reinterpret_cast<const T*>(x.data) + ((long)b * hw + pix) * x.ld + c0, v, min(8, C - c0));
            const float mm = m[(long)b * hw + pix];
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (c0 + j < C) { d = fmaf(w[c0 + j], v[j], d); acc[j] = fmaf(mm, v[j], acc[j]); }
            }
            part = mm * d + (c0 == 0 ? bias[0] : 0.f);
        }
        // the cg partial dots of one pixel sit in adjacent lanes: fold them into the group's first lane when the whole
        // group is inside this wave (one float atomic per pixel instead of cg), otherwise every lane adds its own
        const int cgi = c0 >> 3;
        float tot = part;
        for (int k = 1; k < cg; ++k) { const float t = __shfl_down(part, k, 64); if (cgi == 0) tot += t; }
        const bool whole = fixed && lane - cgi >= 0 && lane - cgi + cg - 1 < 64;
        if (live) {
            if (whole) { if (cgi == 0) atomicAdd(dot + (long)b * hw + pix, tot); }
            else atomicAdd(dot + (long)b * hw + pix, part);
        }
    }
    if (fixed) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fold_stride(acc[j], cg, lane);
        if (lane < cg) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (cfix + j < C) atomicAdd(&red[cfix + j], acc[j]);
        }
    } else if (last >= 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (last + j < C) atomicAdd(&red[last + j], acc[j]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256)
        if (red[i] != 0.f) atomicAdd(chansum + (long)b * C + i, red[i]);
}

// ---- a7 step 2: beta = msum * softmax_{masked}(fcw*tanh(dot + ht) + fcb); one block per image ----
__global__ __launch_bounds__(1024) void sp_softmax_kernel(const float* dot, const float* m, const float* chansum,
                                                          const float* lh, const float* fcw, const float* fcb,
                                                          int C, long L, float* beta, float* rowstat) {
    __shared__ float sh[16];
    __shared__ float s_ht;
    const int b = blockIdx.x;
    if (threadIdx.x < 64) {
        float a = 0.f;
        for (int c = threadIdx.x; c < C; c += 64) a += lh[c] * chansum[(long)b * C + c];
        a = wave_sum(a);
        if (threadIdx.x == 0) s_ht = a / (float)L;
    }
    __syncthreads();
    const float ht = s_ht, fw = fcw[0], fb = fcb[0];
    const float* d = dot + (long)b * L; const float* mm = m + (long)b * L;
    // the score z = fcw * tanh(dot + ht) + fcb is evaluated ONCE per pixel and parked in the output row (-inf outside the
    // mask); the sum and the final pass read it back from L2.  Three passes that each re-evaluated tanhf kept one CU busy
    // for 90 us per launch (one 1024-thread workgroup owns a row: 64 pixels x 3 tanhf + 2 expf per thread).
    float* bt = beta + (long)b * L;
    float mx = -INFINITY, cnt = 0.f;
    for (long p = threadIdx.x; p < L; p += 1024) {
        float z = -INFINITY;
        if (mm[p] >= 0.5f) { z = fmaf(fw, tanhf(d[p] + ht), fb); mx = fmaxf(mx, z); cnt += 1.f; }
        bt[p] = z;
    }
    mx = block_max(mx, sh);
    cnt = block_sum(cnt, sh);
    float se = 0.f;
    for (long p = threadIdx.x; p < L; p += 1024) { const float z = bt[p]; if (z != -INFINITY) se += expf(z - mx); }
    se = block_sum(se, sh);
    const float k = cnt > 0.f ? cnt / se : 0.f;
    for (long p = threadIdx.x; p < L; p += 1024) { const float z = bt[p]; bt[p] = z != -INFINITY ? k * expf(z - mx) : 0.f; }
    if (threadIdx.x == 0 && rowstat) { rowstat[4 * b] = mx; rowstat[4 * b + 1] = se; rowstat[4 * b + 2] = cnt; rowstat[4 * b + 3] = ht; }
}

// ---- a7 step 3: per-channel sum / sumsq of x*beta (BatchNorm statistics of `Base*beta`) ----------
template <typename T>
__global__ __launch_bounds__(256) void scaled_stats_kernel(View x, const float* beta, float* stats) {
    extern __shared__ float red[];      // [2C]
    const int C = x.c;
    for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
    __syncthreads();
    const long pixels = (long)x.n * x.h * x.w;
    const int cg = (C + 7) / 8;
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int last = -1;
    for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < pixels * cg; item += (long)gridDim.x * 256) {
        const int c0 = (int)(item % cg) * 8; const long pix = item / cg;
        if (c0 != last && last >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (last + j < C) { atomicAdd(&red[last + j], s0[j]); atomicAdd(&red[C + last + j], s1[j]); }
                s0[j] = 0.f; s1[j] = 0.f;
            }
        }
        last = c0;
        float v[8];
        load8g<T>(reinterpret_cast<const T*>(x.data) + pix * x.ld + c0, v, min(8, C - c0));
        const float bb = beta[pix];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = v[j] * bb; s0[j] += t; s1[j] += t * t; }
    }
    {   // a lane keeps one channel group when the stride is a multiple of cg (host: grid_keep_cg): fold per wave
        const int lane = threadIdx.x & 63;
        const bool fixed = ((long)gridDim.x * 256) % cg == 0 && cg < 64;
        const int cfix = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cg) * 8;
        if (fixed) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] = fold_stride(s0[j], cg, lane); s1[j] = fold_stride(s1[j], cg, lane); }
            if (lane < cg) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (cfix + j < C) { atomicAdd(&red[cfix + j], s0[j]); atomicAdd(&red[C + cfix + j], s1[j]); }
            }
        } else if (last >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (last + j < C) { atomicAdd(&red[last + j], s0[j]); atomicAdd(&red[C + last + j], s1[j]); }
        }
    }
    __syncthreads();
    float* rep = stats + (blockIdx.x & (ISA_STAT_R - 1)) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        if (red[i] != 0.f) atomicAdd(rep + i, red[i]);
}

// ---- a7 step 4: out = x + (scale*x*beta + shift) * m -------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sp_apply_kernel(View x, const float* beta, const float* m, const float* scale,
                                                       const float* shift, View out) {
    const int C = x.c, cg = (C + 7) / 8;
    const long pixels = (long)x.n * x.h * x.w;
    for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < pixels * cg; item += (long)gridDim.x * 256) {
        const int c0 = (int)(item % cg) * 8; const long pix = item / cg;
        const int nv = min(8, C - c0);
        float v[8];
        load8g<T>(reinterpret_cast<const T*>(x.data) + pix * x.ld + c0, v, nv);
        const float bb = beta[pix], mm = m[pix];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = min(c0 + j, C - 1);
            v[j] = v[j] + fmaf(scale[c], v[j] * bb, shift[c]) * mm;
        }
        store8g<T>(reinterpret_cast<T*>(out.data) + pix * out.ld + c0, v, nv);
    }
}

// ---- a8 maskBN on a single-channel map, fused with the following 3x3 mean * sem (utils.py:636-645)
// pass 1: per image A[b] = sum e*m, M[b] = sum m ; pass 2 (needs mean): V[b] = sum (e-mean)^2*m
template <typename T>
__global__ __launch_bounds__(1024) void maskbn_stats_kernel(View e, const float* m, const float* mean_in, float* out, long L) {
    __shared__ float sh[16];
    const int b = blockIdx.x;
    const T* ep = reinterpret_cast<const T*>(e.data) + (long)b * L * e.ld;
    const float* mp = m + (long)b * L;
    float a = 0.f, c = 0.f;
    if (!mean_in) {
        for (long p = threadIdx.x; p < L; p += 1024) { const float mm = mp[p]; a += st<T>::ld(ep + p * e.ld) * mm; c += mm; }
        a = block_sum(a, sh); c = block_sum(c, sh);
        if (threadIdx.x == 0) { out[2 * b] = a; out[2 * b + 1] = c; }
    } else {
        const float mu = mean_in[0];
        for (long p = threadIdx.x; p < L; p += 1024) { const float d = st<T>::ld(ep + p * e.ld) - mu; a += d * d * mp[p]; }
        a = block_sum(a, sh);
        if (threadIdx.x == 0) out[b] = a;
    }
}
// tiny: combine per-image sums into mean / var (+ running stats with the reference's inverted momentum)
__global__ void maskbn_finalize_kernel(const float* am, const float* v, int n, int stage, float* mean_var,
                                       float* running_mean, float* running_var, float f, int train) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!train) { mean_var[0] = running_mean[0]; mean_var[1] = running_var[0]; return; }
    if (stage == 0) {
        float s = 0.f;
        for (int b = 0; b < n; ++b) s += am[2 * b] / (am[2 * b + 1] + 1.f);
        mean_var[0] = s / n;
    } else {
        float s = 0.f;
        for (int b = 0; b < n; ++b) s += v[b] / (am[2 * b + 1] + 1.f);
        mean_var[1] = s / n;
        running_mean[0] = running_mean[0] * f + (1.f - f) * mean_var[0];
        running_var[0] = running_var[0] * f + (1.f - f) * mean_var[1];
    }
}
// merge[p] = sem[p] * (1/9) * sum_{q in 3x3(p), in bounds} ((e[q]-mean)/sqrt(var+eps)*w + b)
template <typename T>
__global__ __launch_bounds__(256) void maskbn_apply_pool_kernel(View e, const float* sem, const float* mean_var,
                                                                const float* w, const float* bb, float eps, float* merge) {
    const long pixels = (long)e.n * e.h * e.w;
    const float mu = mean_var[0], k = w[0] / sqrtf(mean_var[1] + eps), b0 = bb[0];
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < pixels; pix += (long)gridDim.x * 256) {
        const int x = (int)(pix % e.w); const long q = pix / e.w; const int y = (int)(q % e.h); const long b = q / e.h;
        float acc = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < e.h && xx >= 0 && xx < e.w)
                    acc += (st<T>::ld(reinterpret_cast<const T*>(e.data) + ((b * e.h + yy) * e.w + xx) * e.ld) - mu) * k + b0;
            }
        merge[pix] = sem[pix] * acc * (1.f / 9.f);
    }
}

// ---- a9/a10: alpha[b,:] = softmax over the pixels of instance idx[b] of merge[b,:] (0 if empty) ---
// nsrc: images behind the rows - row b reads image b % nsrc (all decoder iterations in one launch: rows [it*nsrc + image])
__global__ __launch_bounds__(1024) void ins_softmax_kernel(const float* merge, const int64_t* ins, const int32_t* idx,
                                                           int nobj, long L, float* alpha, float* rowstat, int nsrc) {
    __shared__ float sh[16];
    const int b = blockIdx.x, bi = b % nsrc;
    const int64_t* plane = ins + ((long)bi * nobj + idx[b]) * L;
    const float* z = merge + (long)bi * L;
    // the int64 instance plane is read ONCE: the masked score (-inf outside the instance) is parked in the output row and the
    // two later passes read that
    float* al = alpha + (long)b * L;
    float mx = -INFINITY;
    for (long p = threadIdx.x; p < L; p += 1024) {
        const float v = plane[p] != 0 ? z[p] : -INFINITY;
        mx = fmaxf(mx, v);
        al[p] = v;
    }
    mx = block_max(mx, sh);
    float se = 0.f;
    for (long p = threadIdx.x; p < L; p += 1024) { const float v = al[p]; if (v != -INFINITY) se += expf(v - mx); }
    se = block_sum(se, sh);
    const float inv = se > 0.f ? 1.f / se : 0.f;
    for (long p = threadIdx.x; p < L; p += 1024) { const float v = al[p]; al[p] = v != -INFINITY ? expf(v - mx) * inv : 0.f; }
    if (threadIdx.x == 0 && rowstat) { rowstat[2 * b] = mx; rowstat[2 * b + 1] = se; }
}

// ---- chunked row softmax (a7 step 2 and a9/a10 with many workgroups per row) ---------------------------------------
// One 1024-thread workgroup per row keeps n of the 256 CUs busy for 86 us (n = 16 ... 32 rows of 65 536 pixels).  Here a
// row is cut into S chunks of <= 4096 pixels, one 256-thread workgroup each:
//   pass 1 (row_score_kernel): the masked score z of the chunk (16 values per thread, kept in registers), parked in the
//           output row (-inf outside the mask), and the chunk's online-softmax partials part[b][s] = {max, sum exp(z - max), count};
//   pass 2 (row_norm_kernel):  row max / sum from the S partials (sum_s e_s exp(m_s - max)), then the chunk is normalised
//           in place; chunk 0 writes rowstat.
// MODE 0: beta = count * softmax(fcw tanh(dot + ht) + fcb) over m >= 0.5 (SpatialAttentionLayer);  MODE 1: alpha = softmax
// of merge over the pixels of instance idx[b] (HardAttentionLayer).  Same values as the one-workgroup kernels up to the
// order of the fp32 sums.
constexpr int ROW_CHUNK = 4096, ROW_MAX_CHUNKS = ISA_ROW_CHUNKS;
struct RowScore {
    const float *dot, *m, *chansum, *lh, *fcw, *fcb; int C;                 // MODE 0
    const float* merge; const int64_t* ins; const int32_t* idx; int nobj, nsrc;   // MODE 1
    long L; int S; float* out; float* part; float* rowstat;
};
template <int MODE>
__global__ __launch_bounds__(256) void row_score_kernel(RowScore q) {
    __shared__ float sh[4];
    __shared__ float s_ht;
    const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const long p0 = (long)s * ROW_CHUNK, p1 = min(q.L, p0 + ROW_CHUNK);
    float ht = 0.f, fw = 0.f, fb = 0.f;
    const float *src0 = nullptr, *msk = nullptr; const int64_t* plane = nullptr;
    if constexpr (MODE == 0) {
        if (tid < 64) {
            float a = 0.f;
            for (int c = tid; c < q.C; c += 64) a += q.lh[c] * q.chansum[(long)b * q.C + c];
            a = wave_sum(a);
            if (tid == 0) s_ht = a / (float)q.L;
        }
        __syncthreads();
        ht = s_ht; fw = q.fcw[0]; fb = q.fcb[0];
        src0 = q.dot + (long)b * q.L; msk = q.m + (long)b * q.L;
        if (s == 0 && tid == 0 && q.rowstat) q.rowstat[4 * b + 3] = ht;
    } else {
        const int bi = b % q.nsrc;
        plane = q.ins + ((long)bi * q.nobj + q.idx[b]) * q.L;
        src0 = q.merge + (long)bi * q.L;
    }
    float* o = q.out + (long)b * q.L;
    float z[ROW_CHUNK / 256];
    float mx = -INFINITY, cnt = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_CHUNK / 256; ++i) {
        const long p = p0 + tid + 256 * i;
        float v = -INFINITY;
        if (p < p1) {
            if constexpr (MODE == 0) { if (msk[p] >= 0.5f) { v = fmaf(fw, tanhf(src0[p] + ht), fb); cnt += 1.f; } }
            else { if (plane[p] != 0) v = src0[p]; }
            o[p] = v;
        }
        z[i] = v; mx = fmaxf(mx, v);
    }
    mx = block_max(mx, sh);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_CHUNK / 256; ++i) if (z[i] != -INFINITY) se += expf(z[i] - mx);
    se = block_sum(se, sh);
    if constexpr (MODE == 0) cnt = block_sum(cnt, sh);
    if (tid == 0) { float* pt = q.part + ((long)b * q.S + s) * 4; pt[0] = mx; pt[1] = se; pt[2] = cnt; }
}
template <int MODE>
__global__ __launch_bounds__(256) void row_norm_kernel(float* out, const float* part, int S, long L, float* rowstat) {
    const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    // every thread folds the S <= 64 partials itself (the same values in the same order in every workgroup of the row)
    float mx = -INFINITY;
    for (int i = 0; i < S; ++i) mx = fmaxf(mx, part[((long)b * S + i) * 4]);
    float se = 0.f, cnt = 0.f;
    for (int i = 0; i < S; ++i) {
        const float* pt = part + ((long)b * S + i) * 4;
        if (pt[0] != -INFINITY) se += pt[1] * expf(pt[0] - mx);
        cnt += pt[2];
    }
    const float k = MODE == 0 ? (cnt > 0.f ? cnt / se : 0.f) : (se > 0.f ? 1.f / se : 0.f);
    float* o = out + (long)b * L;
    const long p0 = (long)s * ROW_CHUNK, p1 = min(L, p0 + ROW_CHUNK);
    for (long p = p0 + tid; p < p1; p += 256) { const float z = o[p]; o[p] = z != -INFINITY ? k * expf(z - mx) : 0.f; }
    if (s == 0 && tid == 0 && rowstat) {
        if (MODE == 0) { rowstat[4 * b] = mx; rowstat[4 * b + 1] = se; rowstat[4 * b + 2] = cnt; }
        else { rowstat[2 * b] = mx; rowstat[2 * b + 1] = se; }
    }
}

// ---- a11: s_t[b] = argmax_p alpha[b,p], first maximum wins (torch.argmax) -------------------------
// `race` (optional): the exponential race of DecoderLayer.sample's training branch - argmax_p alpha[p] / race[p] with
// race ~ Exp(1) draws one index from Multinomial(alpha) (torch.multinomial's own single-sample form, attenet2.py:321)
__global__ __launch_bounds__(1024) void row_argmax_kernel(const float* a, const float* race, long L, int32_t* out) {
    __shared__ float shv[16];
    __shared__ int shi[16];
    const int b = blockIdx.x;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (long p = threadIdx.x; p < L; p += 1024) {
        float v = a[(long)b * L + p];
        if (race) v = v / race[(long)b * L + p];
        if (v > best) { best = v; bi = (int)p; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { shv[wave] = best; shi[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i)
            if (shv[i] > best || (shv[i] == best && shi[i] < bi)) { best = shv[i]; bi = shi[i]; }
        out[b] = bi;
    }
}

// ---- a12: pyramid targets: tgt[b,y,x] = max over the f x f block of plane ins[b, idx[b]] ---------
// (idx == nullptr: `src` is an fp32 map instead, used for the semantic mask `mask_all`)
// nsrc: images behind the n output maps - map b reads image b % nsrc (all decoder iterations, or copies of one map)
__global__ __launch_bounds__(256) void pool_target_kernel(const int64_t* ins, const int32_t* idx, const float* src,
                                                          int nobj, int n, int H, int W, int f, float* out, int nsrc) {
    const int h = H / f, w = W / f;
    const long total = (long)n * h * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % w); const long q = i / w; const int y = (int)(q % h); const int b = (int)(q / h);
        float v = 0.f;
        if (idx) {
            const int64_t* plane = ins + ((long)(b % nsrc) * nobj + idx[b]) * H * W;
            for (int dy = 0; dy < f; ++dy)
                for (int dx = 0; dx < f; ++dx) if (plane[(long)(y * f + dy) * W + x * f + dx] != 0) v = 1.f;
        } else {
            const float* plane = src + (long)(b % nsrc) * H * W;
            v = -INFINITY;
            for (int dy = 0; dy < f; ++dy)
                for (int dx = 0; dx < f; ++dx) v = fmaxf(v, plane[(long)(y * f + dy) * W + x * f + dx]);
        }
        out[i] = v;
    }
}

// ---- a12/a13: auxiliary concat channels: [mask_all | 2*nb code bits | point marker] ---------------
// (utils.py:1027-1045 conPosition + the mask_all cat at :1085).  s_t is the flat full-res index.
template <typename T>
__global__ __launch_bounds__(256) void concat_aux_kernel(View dst, const float* mask_all, const int32_t* s_t,
                                                         int W_full, int f, int nb, int mask_n) {
    const int naux = 2 * nb + 2;
    const long total = (long)dst.n * dst.h * dst.w * naux;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % naux); const long pix = i / naux;
        const int x = (int)(pix % dst.w); const long q = pix / dst.w; const int y = (int)(q % dst.h); const int b = (int)(q / dst.h);
        float v;
        if (ch == 0) v = mask_all[((long)(b % mask_n) * dst.h + y) * dst.w + x];       // mask_n images behind dst.n (iterations share it)
        else {
            const int s = s_t[b], r = s / W_full, c = s % W_full;
            const bool here = (r / f == y) && (c / f == x);
            const int t = ch - 1;                       // 0..2nb-1 code bits (row bits MSB first, then col), 2nb marker
            if (!here) v = 0.f;
            else if (t == 2 * nb) v = 1.f;
            else {
                const int rem = t < nb ? r % f : c % f;
                const int bit = t < nb ? nb - 1 - t : 2 * nb - 1 - t;
                v = (float)((rem >> bit) & 1);
            }
        }
        st<T>::stv(reinterpret_cast<T*>(dst.data) + pix * dst.ld + ch, v);
    }
}

// ---- a13 gate: out = up * softmax2(bilinear_x2(pred_prev))[1]  (utils.py:1047-1056) -------------
// bilinear, align_corners=False, exact 2x: source coord (o+0.5)/2-0.5 clamped at 0.
__device__ __forceinline__ void bil_src(int o, int n_in, int& i0, int& i1, float& w1) {
    float s = (o + 0.5f) * 0.5f - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s; i1 = min(i0 + 1, n_in - 1); w1 = s - (float)i0;
}
template <typename T>
__device__ __forceinline__ float gate_at(const View& pred, int b, int y, int x) {
    int y0, y1, x0, x1; float wy, wx;
    bil_src(y, pred.h, y0, y1, wy); bil_src(x, pred.w, x0, x1, wx);
    const T* base = reinterpret_cast<const T*>(pred.data) + (long)b * pred.h * pred.w * pred.ld;
    auto diff = [&](int yy, int xx) { const T* q = base + ((long)yy * pred.w + xx) * pred.ld; return st<T>::ld(q + 1) - st<T>::ld(q); };
    const float top = diff(y0, x0) * (1.f - wx) + diff(y0, x1) * wx;
    const float bot = diff(y1, x0) * (1.f - wx) + diff(y1, x1) * wx;
    const float u = top * (1.f - wy) + bot * wy;       // upsampled (l1 - l0)
    return 1.f / (1.f + expf(-u));
}
template <typename T>
__global__ __launch_bounds__(256) void gate_kernel(View up, View pred, View out, float* gmap) {
    const int C = up.c, cg = (C + 7) / 8;
    const long pixels = (long)up.n * up.h * up.w;
    for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < pixels * cg; item += (long)gridDim.x * 256) {
        const int c0 = (int)(item % cg) * 8; const long pix = item / cg;
        const int x = (int)(pix % up.w); const long q = pix / up.w; const int y = (int)(q % up.h); const int b = (int)(q / up.h);
        const float g = gate_at<T>(pred, b, y, x);
        if (gmap && c0 == 0) gmap[pix] = g;
        const int nv = min(8, C - c0);
        float v[8];
        load8g<T>(reinterpret_cast<const T*>(up.data) + pix * up.ld + c0, v, nv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= g;
        store8g<T>(reinterpret_cast<T*>(out.data) + pix * out.ld + c0, v, nv);
    }
}

// ---- a17 losses: per-image partial sums of a 2-class prediction against a {0,1} target ------------
// sums[b][0..6] = { sum p1*t, sum p1, sum t, sum focal, sum ce, sum p1^2, count }
template <typename T>
__global__ __launch_bounds__(256) void mask_loss_sums_kernel(View pred, const float* target, const int64_t* onehot,
                                                             float* sums) {
    __shared__ float sh[32];
    const int b = blockIdx.y;
    const long L = (long)pred.h * pred.w;
    float a[7] = {0, 0, 0, 0, 0, 0, 0};
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < L; p += (long)gridDim.x * 256) {
        const T* q = reinterpret_cast<const T*>(pred.data) + ((long)b * L + p) * pred.ld;
        const float l0 = st<T>::ld(q), l1 = st<T>::ld(q + 1);
        // target: fp32 map, or channel 1 of an int64 NCHW one-hot [n,2,h,w]
        const float t = target ? target[(long)b * L + p] : (float)onehot[((long)b * 2 + 1) * L + p];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), inv = 1.f / (e0 + e1);
        const float p0 = e0 * inv, p1 = e1 * inv;
        const float lse = mx + logf(e0 + e1);
        const float p0c = fminf(fmaxf(p0, 1e-7f), 1.f - 1e-7f), p1c = fminf(fmaxf(p1, 1e-7f), 1.f - 1e-7f);
        a[0] += p1 * t; a[1] += p1; a[2] += t;
        a[3] += -(1.f - p1) * (1.f - p1) * logf(p1c) * t - (1.f - p0) * (1.f - p0) * logf(p0c) * (1.f - t);
        a[4] += lse - (t > 0.5f ? l1 : l0);
        a[5] += p1 * p1; a[6] += 1.f;
    }
    block_sums_atomic<7>(a, sh, sums + 8 * b);
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
    do { if ((dtype) == ISA_BF16) { CALL_BF16; } else { CALL_F32; } } while (0)

extern "C" int isa_mask_dot(const isa_tensor* x, const float* m, const float* w, const float* bias,
                            float* dot, float* chansum, void* stream) {
    if (!tensor_ok(x, 8) || !m || !w || !bias || !dot || !chansum) return ISA_EINVAL;
    const long items = (long)x->h * x->w * ((x->c + 7) / 8);
    dim3 grid(grid_keep_cg(grid_cap(cdiv(items, 256), 256), (x->c + 7) / 8), x->n);
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(mask_dot_kernel<bf16_t>, grid, dim3(256), x->c * 4, as_stream(stream), mkview(x), m, w, bias, dot, chansum),
        hipLaunchKernelGGL(mask_dot_kernel<float>, grid, dim3(256), x->c * 4, as_stream(stream), mkview(x), m, w, bias, dot, chansum));
    return launch_status();
}

extern "C" int isa_sp_softmax(const float* dot, const float* m, const float* chansum, const float* lh,
                              const float* fcw, const float* fcb, int32_t n, int32_t c, int64_t L,
                              float* beta, float* rowstat, float* part, void* stream) {
    if (!dot || !m || !chansum || !lh || !fcw || !fcb || !beta || n <= 0 || L <= 0) return ISA_EINVAL;
    const long S = (L + ROW_CHUNK - 1) / ROW_CHUNK;
    if (part && S <= ROW_MAX_CHUNKS) {
        RowScore q{};
        q.dot = dot; q.m = m; q.chansum = chansum; q.lh = lh; q.fcw = fcw; q.fcb = fcb; q.C = c;
        q.L = L; q.S = (int)S; q.out = beta; q.part = part; q.rowstat = rowstat;
        hipLaunchKernelGGL(row_score_kernel<0>, dim3((unsigned)S, n), dim3(256), 0, as_stream(stream), q);
        hipLaunchKernelGGL(row_norm_kernel<0>, dim3((unsigned)S, n), dim3(256), 0, as_stream(stream), beta, part, (int)S, (long)L, rowstat);
        return launch_status();
    }
    hipLaunchKernelGGL(sp_softmax_kernel, dim3(n), dim3(1024), 0, as_stream(stream), dot, m, chansum, lh, fcw, fcb,
                       c, (long)L, beta, rowstat);
    return launch_status();
}

extern "C" int isa_scaled_stats(const isa_tensor* x, const float* beta, float* stats, void* stream) {
    if (!tensor_ok(x, 8) || !beta || !stats) return ISA_EINVAL;
    const long items = (long)x->n * x->h * x->w * ((x->c + 7) / 8);
    const int grid = grid_keep_cg(grid_cap(cdiv(items, 256), 1024), (x->c + 7) / 8);
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(scaled_stats_kernel<bf16_t>, dim3(grid), dim3(256), 2 * x->c * 4, as_stream(stream), mkview(x), beta, stats),
        hipLaunchKernelGGL(scaled_stats_kernel<float>, dim3(grid), dim3(256), 2 * x->c * 4, as_stream(stream), mkview(x), beta, stats));
    return launch_status();
}

extern "C" int isa_sp_apply(const isa_tensor* x, const float* beta, const float* m, const float* scale,
                            const float* shift, const isa_tensor* out, void* stream) {
    if (!tensor_ok(x, 8) || !tensor_ok(out, 8) || x->c != out->c || x->dtype != out->dtype || !beta || !m || !scale || !shift)
        return ISA_EINVAL;
    const long items = (long)x->n * x->h * x->w * ((x->c + 7) / 8);
    const int grid = grid_cap(cdiv(items, 256));
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(sp_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(x), beta, m, scale, shift, mkview(out)),
        hipLaunchKernelGGL(sp_apply_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(x), beta, m, scale, shift, mkview(out)));
    return launch_status();
}

extern "C" int isa_maskbn_stats(const isa_tensor* e, const float* m, const float* mean_in, float* out, void* stream) {
    if (!tensor_ok(e, 1) || !m || !out) return ISA_EINVAL;
    const long L = (long)e->h * e->w;
    DISPATCH_T(e->dtype,
        hipLaunchKernelGGL(maskbn_stats_kernel<bf16_t>, dim3(e->n), dim3(1024), 0, as_stream(stream), mkview(e), m, mean_in, out, L),
        hipLaunchKernelGGL(maskbn_stats_kernel<float>, dim3(e->n), dim3(1024), 0, as_stream(stream), mkview(e), m, mean_in, out, L));
    return launch_status();
}

extern "C" int isa_maskbn_finalize(const float* am, const float* v, int32_t n, int32_t stage, float* mean_var,
                                   float* running_mean, float* running_var, float f, int32_t train, void* stream) {
    if (!mean_var || !running_mean || !running_var) return ISA_EINVAL;
    hipLaunchKernelGGL(maskbn_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), am, v, n, stage, mean_var,
                       running_mean, running_var, f, train);
    return launch_status();
}

extern "C" int isa_maskbn_apply_pool(const isa_tensor* e, const float* sem, const float* mean_var, const float* w,
                                     const float* b, float eps, float* merge, void* stream) {
    if (!tensor_ok(e, 1) || !sem || !mean_var || !w || !b || !merge) return ISA_EINVAL;
    const int grid = grid_cap(cdiv((long)e->n * e->h * e->w, 256));
    DISPATCH_T(e->dtype,
        hipLaunchKernelGGL(maskbn_apply_pool_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(e), sem, mean_var, w, b, eps, merge),
        hipLaunchKernelGGL(maskbn_apply_pool_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(e), sem, mean_var, w, b, eps, merge));
    return launch_status();
}

extern "C" int isa_ins_softmax(const float* merge, const int64_t* ins, const int32_t* idx, int32_t n, int32_t nobj,
                               int64_t L, float* alpha, float* rowstat, int32_t nsrc, float* part, void* stream) {
    if (!merge || !ins || !idx || !alpha || n <= 0) return ISA_EINVAL;
    if (nsrc <= 0) nsrc = n;
    if (n % nsrc) return ISA_EINVAL;
    const long S = (L + ROW_CHUNK - 1) / ROW_CHUNK;
    if (part && S <= ROW_MAX_CHUNKS) {
        RowScore q{};
        q.merge = merge; q.ins = ins; q.idx = idx; q.nobj = nobj; q.nsrc = nsrc;
        q.L = L; q.S = (int)S; q.out = alpha; q.part = part; q.rowstat = rowstat;
        hipLaunchKernelGGL(row_score_kernel<1>, dim3((unsigned)S, n), dim3(256), 0, as_stream(stream), q);
        hipLaunchKernelGGL(row_norm_kernel<1>, dim3((unsigned)S, n), dim3(256), 0, as_stream(stream), alpha, part, (int)S, (long)L, rowstat);
        return launch_status();
    }
    hipLaunchKernelGGL(ins_softmax_kernel, dim3(n), dim3(1024), 0, as_stream(stream), merge, ins, idx, nobj, (long)L, alpha, rowstat, nsrc);
    return launch_status();
}

extern "C" int isa_row_argmax(const float* a, const float* race, int32_t n, int64_t L, int32_t* out, void* stream) {
    if (!a || !out || n <= 0) return ISA_EINVAL;
    hipLaunchKernelGGL(row_argmax_kernel, dim3(n), dim3(1024), 0, as_stream(stream), a, race, (long)L, out);
    return launch_status();
}

namespace {
// sem_seg_argmax = GT.argmax(1) of the int64 one-hot target (reseg.py:118), as the fp32 {0,1} map the head consumes
__global__ __launch_bounds__(256) void onehot_map_kernel(const int64_t* oh, long hw, long total, float* out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / hw, p = i - b * hw;
        out[i] = oh[(b * 2 + 1) * hw + p] > oh[(b * 2) * hw + p] ? 1.f : 0.f;     // first maximum wins (torch.argmax)
    }
}
// F.dropout2d masks from uniform draws: mask = (u < keep) / keep
__global__ __launch_bounds__(256) void dropout_mask_kernel(const float* u, long n, float keep, float* out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = u[i] < keep ? 1.f / keep : 0.f;
}
// softmax over the channels of an NHWC logit map, written as NCHW fp32 (Model.predict, model.py:486)
template <typename T>
__global__ __launch_bounds__(256) void softmax_nchw_kernel(View x, float* out) {
    const long hw = (long)x.h * x.w, pixels = hw * x.n;
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < pixels; pix += (long)gridDim.x * 256) {
        const T* p = reinterpret_cast<const T*>(x.data) + pix * x.ld;
        float m = -INFINITY;
        for (int c = 0; c < x.c; ++c) m = fmaxf(m, st<T>::ld(p + c));
        float s = 0.f;
        for (int c = 0; c < x.c; ++c) s += expf(st<T>::ld(p + c) - m);
        const long b = pix / hw, q = pix - b * hw;
        for (int c = 0; c < x.c; ++c) out[(b * x.c + c) * hw + q] = expf(st<T>::ld(p + c) - m) / s;
    }
}
}  // namespace

extern "C" int isa_onehot_map(const int64_t* onehot, int32_t n, int64_t hw, float* out, void* stream) {
    if (!onehot || !out || n <= 0 || hw <= 0) return ISA_EINVAL;
    hipLaunchKernelGGL(onehot_map_kernel, dim3(grid_cap(cdiv(n * hw, 256))), dim3(256), 0, as_stream(stream), onehot, (long)hw, (long)n * hw, out);
    return launch_status();
}

extern "C" int isa_dropout_mask(const float* u, int64_t n, float keep, float* out, void* stream) {
    if (!u || !out || n <= 0 || !(keep > 0.f)) return ISA_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_cap(cdiv(n, 256), 256)), dim3(256), 0, as_stream(stream), u, (long)n, keep, out);
    return launch_status();
}

extern "C" int isa_softmax_nchw(const isa_tensor* x, float* out, void* stream) {
    if (!tensor_ok(x, 1) || !out) return ISA_EINVAL;
    const int grid = grid_cap(cdiv((long)x->n * x->h * x->w, 256));
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(softmax_nchw_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(x), out),
        hipLaunchKernelGGL(softmax_nchw_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(x), out));
    return launch_status();
}

extern "C" int isa_pool_target(const int64_t* ins, const int32_t* idx, const float* src, int32_t nobj, int32_t n,
                               int32_t H, int32_t W, int32_t f, float* out, int32_t nsrc, void* stream) {
    if ((!ins && !src) || (ins && !idx) || !out || f < 1 || H % f || W % f || n <= 0) return ISA_EINVAL;
    if (nsrc <= 0) nsrc = n;
    if (n % nsrc) return ISA_EINVAL;
    const int grid = grid_cap(cdiv((long)n * (H / f) * (W / f), 256));
    hipLaunchKernelGGL(pool_target_kernel, dim3(grid), dim3(256), 0, as_stream(stream), ins, ins ? idx : nullptr, src, nobj, n, H, W, f, out, nsrc);
    return launch_status();
}

extern "C" int isa_concat_aux(const isa_tensor* dst, const float* mask_all, const int32_t* s_t, int32_t W_full,
                              int32_t f, int32_t nb, int32_t mask_n, void* stream) {
    if (!tensor_ok(dst, 1) || !mask_all || !s_t || dst->c != 2 * nb + 2) return ISA_EINVAL;
    if (mask_n <= 0) mask_n = dst->n;
    if (dst->n % mask_n) return ISA_EINVAL;
    const int grid = grid_cap(cdiv((long)dst->n * dst->h * dst->w * dst->c, 256));
    DISPATCH_T(dst->dtype,
        hipLaunchKernelGGL(concat_aux_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(dst), mask_all, s_t, W_full, f, nb, mask_n),
        hipLaunchKernelGGL(concat_aux_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(dst), mask_all, s_t, W_full, f, nb, mask_n));
    return launch_status();
}

extern "C" int isa_gate(const isa_tensor* up, const isa_tensor* pred, const isa_tensor* out, float* gmap, void* stream) {
    if (!tensor_ok(up, 8) || !tensor_ok(pred, 1) || !tensor_ok(out, 8) || pred->c != 2 || up->c != out->c ||
        up->h != 2 * pred->h || up->w != 2 * pred->w || up->dtype != pred->dtype || up->dtype != out->dtype) return ISA_EINVAL;
    const long items = (long)up->n * up->h * up->w * ((up->c + 7) / 8);
    const int grid = grid_cap(cdiv(items, 256));
    DISPATCH_T(up->dtype,
        hipLaunchKernelGGL(gate_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(up), mkview(pred), mkview(out), gmap),
        hipLaunchKernelGGL(gate_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(up), mkview(pred), mkview(out), gmap));
    return launch_status();
}

extern "C" int isa_mask_loss_sums(const isa_tensor* pred, const float* target, const int64_t* onehot, float* sums,
                                  void* stream) {
    if (!tensor_ok(pred, 1) || pred->c != 2 || (!target && !onehot) || !sums) return ISA_EINVAL;
    const long L = (long)pred->h * pred->w;
    dim3 grid(grid_cap(cdiv(L, 256), 64), pred->n);
    DISPATCH_T(pred->dtype,
        hipLaunchKernelGGL(mask_loss_sums_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), mkview(pred), target, onehot, sums),
        hipLaunchKernelGGL(mask_loss_sums_kernel<float>, grid, dim3(256), 0, as_stream(stream), mkview(pred), target, onehot, sums));
    return launch_status();
}
