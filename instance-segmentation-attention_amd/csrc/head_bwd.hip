// Backward of the attention mask head and the losses (reference: attenet2.py:86-141,204-290 losses;
// utils.py:484-523 SpatialAttentionLayer; utils.py:568-591 maskBN; utils.py:631-655 hard attention;
// utils.py:1047-1056 gate; utils.py:402-420 squeeze-excite).  The reference gets these from autograd;
// here each is a hand-derived kernel.  Same data conventions as attention.hip: NHWC feature views
// (T = bf16|f32), fp32 [n, h*w] single-channel maps, one 1024-thread workgroup per row for row-wide
// reductions, registers -> LDS atomics -> one global atomic per channel for channel reductions.
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

// ---------------------------------------------------------------------------------------------
// loss assembly (one workgroup): per-level coefficients for the prediction gradients, REINFORCE
// advantage, EMA baseline, and the four reported scalars.  attenet2.py:239-290.
// sums[lvl][b][8] as produced by isa_mask_loss_sums; coef[lvl][b][4] = {c_t, c_1, c_focal, c_ce}:
//   dL/dp1 = c_t*t + c_1 + c_focal*dfocal/dp1 ;  dL/dp0 = c_focal*dfocal/dp0 ;  dL/dl1 += c_ce*(p1-t)
// ---------------------------------------------------------------------------------------------
struct HeadLossParams {
    const float* sums; const float* alpha; const int32_t* s_t; long L; int B, iters;
    float w[5]; float ce_weight, lambda_l, lambda_r, inv_iter;
    float* baseline; int training;
    float* coef; float* adv; float* scal;
};
__global__ __launch_bounds__(1024) void head_loss_kernel(HeadLossParams p) {
    __shared__ float sh[16];
    const int b = threadIdx.x;
    const bool on = b < p.B;
    const int rows = p.iters * p.B;                         // images per level in sums / coef: [level][iteration][image]
    float base = p.baseline[0];
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    // the iterations in order: the EMA baseline of iteration i is what iteration i + 1 starts from (attenet2.py:266)
    for (int it = 0; it < p.iters; ++it) {
        const int row = it * p.B + b;
        float loss_pred = 0.f, d4 = 0.f, d4sq = 0.f, ce_sum = 0.f, cnt4 = 0.f;
        if (on) {
            for (int l = 0; l < 5; ++l) {
                const float* s = p.sums + ((long)l * rows + row) * 8;
                const float A = s[0], S = s[1], T = s[2], F = s[3], cnt = s[6];
                const float den = S + T + 1.f;
                const float D = 1.f - (2.f * A + 1.f) / den;
                loss_pred += p.w[l] * (p.ce_weight * F / cnt + D);
                const float g = p.inv_iter / (float)p.B * p.lambda_l * p.w[l];
                float* c = p.coef + ((long)l * rows + row) * 4;
                c[0] = p.training ? g * (-2.f / den) : 0.f;
                c[1] = p.training ? g * (2.f * A + 1.f) / (den * den) : 0.f;
                c[2] = p.training ? g * p.ce_weight / cnt : 0.f;
                c[3] = 0.f;
                if (l == 4) {
                    d4 = D; ce_sum = s[4]; cnt4 = cnt;
                    d4sq = 1.f - (2.f * A + 1.f) / (s[5] + T + 1.f);       // dice with time=2 (eval branch)
                }
            }
        }
        const float lp = -d4;                                   // log_p_y = -eval_dice
        const float mean_lp = block_sum(on ? lp : 0.f, sh) / (float)p.B;
        const float ce = block_sum(on ? ce_sum : 0.f, sh) / block_sum(on ? cnt4 : 0.f, sh);
        const float dsum = block_sum(on ? d4 : 0.f, sh);
        if (p.training) base = 0.9f * base + 0.1f * mean_lp;
        float per_img = 0.f;
        if (on && p.training) {
            const float picked = p.alpha[(long)row * p.L + p.s_t[row]];
            const float loss_r = -(lp - base) * logf(picked);
            per_img = p.lambda_l * loss_pred + p.lambda_r * loss_r;
            p.adv[row] = p.inv_iter / (float)p.B * p.lambda_r * (lp - base);
        }
        const float tot = block_sum(per_img, sh);
        const float d2sum = block_sum(on ? d4sq : 0.f, sh);
        if (p.training) {
            acc0 += p.inv_iter * tot / (float)p.B;                      // ins_cost without the NaN entropy term
            acc1 += p.inv_iter * (ce + dsum);                           // criterion
        } else {
            acc0 += p.inv_iter * d2sum / (float)p.B;
            acc1 += p.inv_iter * (ce + dsum / (float)p.B);
        }
        acc2 += p.inv_iter * ce;
        acc3 += p.inv_iter * dsum / (float)p.B;
    }
    if (threadIdx.x == 0) {
        if (p.training) p.baseline[0] = base;
        p.scal[0] += acc0; p.scal[1] += acc1; p.scal[2] += acc2; p.scal[3] += acc3;
    }
}

// trainer-side semantic losses (model.py:255-269): CE (mean over all pixels) + Dice(time=1, fg, mean)
__global__ __launch_bounds__(1024) void sem_loss_kernel(const float* sums, int B, float* coef, float* scal) {
    __shared__ float sh[16];
    const int b = threadIdx.x; const bool on = b < B;
    float D = 0.f, ce = 0.f, cnt = 0.f;
    if (on) {
        const float* s = sums + (long)b * 8;
        const float A = s[0], S = s[1], T = s[2];
        const float den = S + T + 1.f;
        D = 1.f - (2.f * A + 1.f) / den; ce = s[4]; cnt = s[6];
        float* c = coef + (long)b * 4;
        c[0] = (1.f / B) * (-2.f / den); c[1] = (1.f / B) * (2.f * A + 1.f) / (den * den); c[2] = 0.f;
    }
    const float tot_cnt = block_sum(cnt, sh);
    const float ce_sum = block_sum(ce, sh);
    const float dmean = block_sum(D, sh) / (float)B;
    if (on) coef[(long)b * 4 + 3] = 1.f / tot_cnt;
    if (threadIdx.x == 0) { scal[0] = ce_sum / tot_cnt; scal[1] = dmean; }
}

// d(pred logits) from the coefficients above
template <typename T>
__global__ __launch_bounds__(256) void mask_loss_grad_kernel(View pred, const float* target, const int64_t* onehot,
                                                             const float* coef, View dpred, int accumulate) {
    const int b = blockIdx.y;
    const long L = (long)pred.h * pred.w;
    const float c_t = coef[4 * b], c_1 = coef[4 * b + 1], c_f = coef[4 * b + 2], c_ce = coef[4 * b + 3];
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < L; p += (long)gridDim.x * 256) {
        const T* q = reinterpret_cast<const T*>(pred.data) + ((long)b * L + p) * pred.ld;
        const float l0 = st<T>::ld(q), l1 = st<T>::ld(q + 1);
        const float t = target ? target[(long)b * L + p] : (float)onehot[((long)b * 2 + 1) * L + p];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), inv = 1.f / (e0 + e1);
        const float p0 = e0 * inv, p1 = e1 * inv;
        float dp1 = c_t * t + c_1, dp0 = 0.f;
        if (c_f != 0.f) {
            if (p1 > 1e-7f && p1 < 1.f - 1e-7f) dp1 += c_f * (-(1.f - p1) * (1.f - p1) * t / p1);
            if (p0 > 1e-7f && p0 < 1.f - 1e-7f) dp0 += c_f * (-(1.f - p0) * (1.f - p0) * (1.f - t) / p0);
        }
        const float g1 = p1 * p0 * (dp1 - dp0) + c_ce * (p1 - t);
        T* d = reinterpret_cast<T*>(dpred.data) + ((long)b * L + p) * dpred.ld;
        float o0 = -g1, o1 = g1;
        if (accumulate) { o0 += st<T>::ld(d); o1 += st<T>::ld(d + 1); }
        st<T>::stv(d, o0); st<T>::stv(d + 1, o1);
    }
}

// ---------------------------------------------------------------------------------------------
// REINFORCE path: d merge[b,p] += adv[b] * (alpha[b,p] - [p == s_t[b]]) on the instance's pixels
// ---------------------------------------------------------------------------------------------
// rows [it*nsrc + image] of `iters` decoder iterations; a workgroup owns pixels of ONE image and walks its iterations, so
// the read-modify-write of dmerge needs no atomics
__global__ __launch_bounds__(256) void ins_softmax_bwd_kernel(const float* alpha, const int64_t* ins, const int32_t* idx,
                                                              const int32_t* s_t, const float* adv, int nobj, long L,
                                                              float* dmerge, int nsrc, int iters) {
    const int bi = blockIdx.y;
    for (int it = 0; it < iters; ++it) {
        const int b = it * nsrc + bi;
        const int64_t* plane = ins + ((long)bi * nobj + idx[b]) * L;
        const float a = adv[b]; const int s = s_t[b];
        for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < L; p += (long)gridDim.x * 256)
            if (plane[p] != 0) dmerge[(long)bi * L + p] += a * (alpha[(long)b * L + p] - (p == s ? 1.f : 0.f));
    }
}

// ---------------------------------------------------------------------------------------------
// maskBN + AvgPool3x3*sem backward (single-channel map e, T activation)
// de_bn[q] = (1/9) sum_{p in N(q)} sem[p]*dmerge[p]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float pooled_grad(const float* sem, const float* dmerge, long b, int y, int x, int h, int w) {
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                const long q = (b * h + yy) * w + xx;
                acc += sem[q] * dmerge[q];
            }
        }
    return acc * (1.f / 9.f);
}
// red[0]=sum dEhat, red[1]=sum dEhat*(e-mean), red[2]=sum de_bn*ehat (dW), red[3]=sum de_bn (dB)
template <typename T>
__global__ __launch_bounds__(256) void maskbn_bwd_reduce_kernel(View e, const float* sem, const float* dmerge,
                                                                const float* mean_var, const float* w, float eps, float* red) {
    __shared__ float sh[16];
    const long pixels = (long)e.n * e.h * e.w;
    const float mu = mean_var[0], inv = 1.f / sqrtf(mean_var[1] + eps), ww = w[0];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < pixels; pix += (long)gridDim.x * 256) {
        const int x = (int)(pix % e.w); const long q = pix / e.w; const int y = (int)(q % e.h); const long b = q / e.h;
        const float g = pooled_grad(sem, dmerge, b, y, x, e.h, e.w);
        const float d = st<T>::ld(reinterpret_cast<const T*>(e.data) + pix * e.ld) - mu;
        a0 += g * ww; a1 += g * ww * d; a2 += g * d * inv; a3 += g;
    }
    float a[4] = {a0, a1, a2, a3};
    block_sums_atomic<4>(a, sh, red);
}
// tiny: k[0]=dvar, k[1]=dmean; param grads
__global__ void maskbn_bwd_finalize_kernel(const float* red, const float* am, const float* mean_var, int n, float eps,
                                           int train, float* k, float* dw, float* db) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    atomicAdd(dw, red[2]); atomicAdd(db, red[3]);
    if (!train) { k[0] = 0.f; k[1] = 0.f; return; }
    const float var = mean_var[1], mu = mean_var[0];
    const float inv = 1.f / sqrtf(var + eps);
    const float dvar = red[1] * (-0.5f) * inv * inv * inv;
    float Q = 0.f;                               // d var / d mean
    for (int b = 0; b < n; ++b) Q += -2.f * (am[2 * b] - mu * am[2 * b + 1]) / (am[2 * b + 1] + 1.f);
    Q /= (float)n;
    k[0] = dvar; k[1] = -inv * red[0] + dvar * Q;
}
template <typename T>
__global__ __launch_bounds__(256) void maskbn_bwd_apply_kernel(View e, const float* sem, const float* dmerge,
                                                               const float* mean_var, const float* w, float eps, const float* am,
                                                               const float* k, int train, View de, int accumulate) {
    const long pixels = (long)e.n * e.h * e.w;
    const float mu = mean_var[0], inv = 1.f / sqrtf(mean_var[1] + eps), ww = w[0];
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < pixels; pix += (long)gridDim.x * 256) {
        const int x = (int)(pix % e.w); const long q = pix / e.w; const int y = (int)(q % e.h); const long b = q / e.h;
        float g = pooled_grad(sem, dmerge, b, y, x, e.h, e.w) * ww * inv;
        if (train) {
            const float d = st<T>::ld(reinterpret_cast<const T*>(e.data) + pix * e.ld) - mu;
            g += sem[pix] / ((float)e.n * (am[2 * b + 1] + 1.f)) * (2.f * k[0] * d + k[1]);
        }
        T* o = reinterpret_cast<T*>(de.data) + pix * de.ld;
        if (accumulate) g += st<T>::ld(o);
        st<T>::stv(o, g);
    }
}

// ---------------------------------------------------------------------------------------------
// SpatialAttentionLayer backward.  Forward (attention.hip): v = x*beta, out = x + (sc*v+sh)*m with
// BN(v) batch statistics (mean, invstd), beta = cnt*softmax_m(fw*tanh(dot+ht)+fb),
// dot = m*<wv,x>+bv, ht = <lh, sum_p m*x>/L.
// ---------------------------------------------------------------------------------------------
struct SpBwd {
    View dout, x, dx;
    const float *beta, *m, *scale, *mean, *invstd, *red, *ddot, *dht, *lh, *wv;
    float inv_count, inv_L; int train, accumulate;
    float* out_red; float* dbeta_map; float* gw; float* sdot;
};
// pass 1: red[c] += sum dq, red[C+c] += sum dq*vhat       (dq = dout*m)
template <typename T>
__global__ __launch_bounds__(256) void sp_bwd_reduce_kernel(SpBwd p) {
    extern __shared__ float red[];
    const int C = p.x.c, cg = (C + 7) / 8;
    for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
    __syncthreads();
    const long pixels = (long)p.x.n * p.x.h * p.x.w;
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mu[8], is[8];
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
        if (c0 != last) {                      // a lane keeps its channel group (grid_keep_cg): loaded once
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int c = min(c0 + j, C - 1); mu[j] = p.mean[c]; is[j] = p.invstd[c]; }
        }
        last = c0;
        const int nv = min(8, C - c0);
        float d[8], xv[8];
        load8g<T>(reinterpret_cast<const T*>(p.dout.data) + pix * p.dout.ld + c0, d, nv);
        load8g<T>(reinterpret_cast<const T*>(p.x.data) + pix * p.x.ld + c0, xv, nv);
        const float bb = p.beta[pix], mm = p.m[pix];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dq = d[j] * mm;
            const float vh = (xv[j] * bb - mu[j]) * is[j];
            s0[j] += dq; s1[j] += dq * vh;
        }
    }
    {
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
    for (int i = threadIdx.x; i < 2 * C; i += 256)
        if (red[i] != 0.f) atomicAdd(p.out_red + i, red[i]);
}
// per-channel constants of the BN(v) backward, hoisted out of the element loops: r0 = red[c]/count, r1 = red[C+c]/count
struct SpConst { float sc[8], mu[8], is[8], r0[8], r1[8]; };
__device__ __forceinline__ void sp_load_const(const SpBwd& p, int c0, int C, SpConst& k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = min(c0 + j, C - 1);
        k.sc[j] = p.scale[c]; k.mu[j] = p.mean[c]; k.is[j] = p.invstd[c];
        k.r0[j] = p.train ? p.red[c] * p.inv_count : 0.f;
        k.r1[j] = p.train ? p.red[C + c] * p.inv_count : 0.f;
    }
}
__device__ __forceinline__ float sp_dv(float sc, float mu, float is, float r0, float r1, int train, float dq, float xv, float bb) {
    if (!train) return sc * dq;
    const float vh = (xv * bb - mu) * is;
    return sc * (dq - r0 - vh * r1);
}
// pass 2: dbeta[p] = sum_c dv[p,c]*x[p,c].  One lane owns a pixel (all its channel groups: the 16-byte loads of
// neighbouring lanes tile the pixel rows completely) and stores the sum: the three-lanes-per-pixel version spent
// its time in 3 global float atomics per pixel and 7 per-channel constant loads per element (131 us at 16x256x256x24).
template <typename T>
__global__ __launch_bounds__(256) void sp_bwd_dbeta_kernel(SpBwd p) {
    extern __shared__ float cst[];            // [5][C]: scale, mean, invstd, red0/count, red1/count
    const int C = p.x.c, cg = (C + 7) / 8;
    for (int c = threadIdx.x; c < C; c += 256) {
        cst[c] = p.scale[c]; cst[C + c] = p.mean[c]; cst[2 * C + c] = p.invstd[c];
        cst[3 * C + c] = p.train ? p.red[c] * p.inv_count : 0.f;
        cst[4 * C + c] = p.train ? p.red[C + c] * p.inv_count : 0.f;
    }
    __syncthreads();
    const long pixels = (long)p.x.n * p.x.h * p.x.w;
    const unsigned hw = (unsigned)p.x.h * (unsigned)p.x.w;
    const bool aligned = hw % 256 == 0;
    __shared__ float shs[4];
    // a workgroup walks a CONTIGUOUS range of 256-pixel blocks: with rows that are a multiple of 256 pixels (every configured
    // size) it stays inside one image for many trips, and the row's sum of beta * dbeta - what pass 3 needs before it can
    // touch a pixel - leaves as one block reduction and one atomic per workgroup and image (one atomic per wave and trip,
    // 1024 on the same address per row, made this pass 106 us instead of 44)
    const long nblk = (pixels + 255) / 256, per = (nblk + gridDim.x - 1) / gridDim.x;
    const long b0 = (long)blockIdx.x * per, b1 = min(nblk, b0 + per);
    float tacc = 0.f; unsigned cur_row = b0 < b1 ? (unsigned)(b0 * 256) / hw : 0u;
    for (long blk = b0; blk < b1; ++blk) {                                                        // workgroup-uniform trips
        const long base = blk * 256, pix = base + threadIdx.x;
        if (aligned) {
            const unsigned rw = (unsigned)base / hw;
            if (rw != cur_row) {
                const float tb = block_sum(tacc, shs);
                if (threadIdx.x == 0 && tb != 0.f) atomicAdd(p.sdot + cur_row, tb);
                tacc = 0.f; cur_row = rw;
            }
        }
        float t = 0.f; int row = 0;
        if (pix < pixels) {
            const float bb = p.beta[pix], mm = p.m[pix];
            float acc = 0.f;
            for (int g = 0; g < cg; ++g) {
                const int c0 = g * 8, nv = min(8, C - c0);
                float d[8], xv[8];
                load8g<T>(reinterpret_cast<const T*>(p.dout.data) + pix * p.dout.ld + c0, d, nv);
                load8g<T>(reinterpret_cast<const T*>(p.x.data) + pix * p.x.ld + c0, xv, nv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = c0 + j;
                    if (c < C) acc += sp_dv(cst[c], cst[C + c], cst[2 * C + c], cst[3 * C + c], cst[4 * C + c], p.train, d[j] * mm, xv[j], bb) * xv[j];
                }
            }
            p.dbeta_map[pix] = acc;
            t = bb * acc;
            if (!aligned) row = (int)((unsigned)pix / hw);             // n * h * w < 2^32 (checked on the host)
        }
        if (aligned) tacc += t;
        else {
            const int row0 = __shfl(row, 0, 64);
            if (__all(row == row0 || pix >= pixels)) {
                t = wave_sum(t);
                if ((threadIdx.x & 63) == 0 && t != 0.f) atomicAdd(p.sdot + row0, t);
            } else if (t != 0.f) atomicAdd(p.sdot + row, t);
        }
    }
    if (aligned && b0 < b1) {
        const float tb = block_sum(tacc, shs);
        if (threadIdx.x == 0 && tb != 0.f) atomicAdd(p.sdot + cur_row, tb);
    }
}
// pass 3 (one workgroup per image): softmax + tanh backward on the maps.
// rowstat[b] = {max, sumexp, cnt, ht}; outputs ddot map, dht[b], atomics into d fcw / d fcb
__global__ __launch_bounds__(1024) void sp_bwd_row_kernel(const float* beta, const float* dbeta, const float* dot, const float* m,
                                                          const float* rowstat, const float* fcw, long L, float* ddot, float* dht,
                                                          float* dfcw, float* dfcb) {
    __shared__ float sh[16];
    const int b = blockIdx.x;
    const float cnt = rowstat[4 * b + 2], ht = rowstat[4 * b + 3], fw = fcw[0];
    const float* be = beta + (long)b * L; const float* db = dbeta + (long)b * L;
    float s = 0.f;
    for (long p = threadIdx.x; p < L; p += 1024) s += be[p] * db[p];
    const float S = cnt > 0.f ? block_sum(s, sh) / cnt : 0.f;
    float a_w = 0.f, a_b = 0.f, a_h = 0.f;
    for (long p = threadIdx.x; p < L; p += 1024) {
        float du = 0.f;
        if (m[(long)b * L + p] >= 0.5f) {
            const float dz = be[p] * (db[p] - S);
            const float t = tanhf(dot[(long)b * L + p] + ht);
            a_w += dz * t; a_b += dz;
            du = dz * fw * (1.f - t * t);
            a_h += du;
        }
        ddot[(long)b * L + p] = du;
    }
    a_w = block_sum(a_w, sh); a_b = block_sum(a_b, sh); a_h = block_sum(a_h, sh);
    if (threadIdx.x == 0) { atomicAdd(dfcw, a_w); atomicAdd(dfcb, a_b); dht[b] = a_h; }
}
// pass 3 with many workgroups per row: the row's sum of beta * dbeta comes from pass 2 (sdot), so every chunk of 4096 pixels is
// independent; the three row sums leave as one atomic per workgroup (dht is part of the zeroed scratch).  The one-workgroup
// kernel above kept n = 16 CUs busy for 76 us.
constexpr int SPB_CHUNK = 4096;
__global__ __launch_bounds__(256) void sp_bwd_rowc_kernel(const float* beta, const float* dbeta, const float* dot, const float* m,
                                                          const float* rowstat, const float* fcw, const float* sdot, long L,
                                                          float* ddot, float* dht, float* dfcw, float* dfcb) {
    __shared__ float sh[4];
    const int b = blockIdx.y;
    const float cnt = rowstat[4 * b + 2], ht = rowstat[4 * b + 3], fw = fcw[0];
    const float S = cnt > 0.f ? sdot[b] / cnt : 0.f;
    const long o = (long)b * L, p0 = (long)blockIdx.x * SPB_CHUNK, p1 = min(L, p0 + SPB_CHUNK);
    float a_w = 0.f, a_b = 0.f, a_h = 0.f;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        float du = 0.f;
        if (m[o + p] >= 0.5f) {
            const float dz = beta[o + p] * (dbeta[o + p] - S);
            const float t = tanhf(dot[o + p] + ht);
            a_w += dz * t; a_b += dz;
            du = dz * fw * (1.f - t * t);
            a_h += du;
        }
        ddot[o + p] = du;
    }
    a_w = block_sum(a_w, sh); a_b = block_sum(a_b, sh); a_h = block_sum(a_h, sh);
    if (threadIdx.x == 0) { atomicAdd(dfcw, a_w); atomicAdd(dfcb, a_b); atomicAdd(dht + b, a_h); }
}
// pass 4: dx (+)= dout + dv*beta + ddot*m*wv + (dht[b]*lh/L)*m ;  gw[c] += sum_p ddot*m*x
template <typename T>
__global__ __launch_bounds__(256) void sp_bwd_dx_kernel(SpBwd p) {
    extern __shared__ float red[];
    const int C = p.x.c, cg = (C + 7) / 8;
    for (int i = threadIdx.x; i < C; i += 256) red[i] = 0.f;
    __syncthreads();
    const long pixels = (long)p.x.n * p.x.h * p.x.w, hw = (long)p.x.h * p.x.w;
    float gacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wv[8], lh[8];
    SpConst k;
    int last = -1;
    for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < pixels * cg; item += (long)gridDim.x * 256) {
        const int c0 = (int)(item % cg) * 8; const long pix = item / cg;
        if (c0 != last && last >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { if (last + j < C) atomicAdd(&red[last + j], gacc[j]); gacc[j] = 0.f; }
        }
        if (c0 != last) {                      // a lane keeps its channel group (grid_keep_cg): loaded once
            sp_load_const(p, c0, C, k);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int c = min(c0 + j, C - 1); wv[j] = p.wv[c]; lh[j] = p.lh[c]; }
        }
        last = c0;
        const int nv = min(8, C - c0);
        const long b = pix / hw;
        float d[8], xv[8], o[8];
        load8g<T>(reinterpret_cast<const T*>(p.dout.data) + pix * p.dout.ld + c0, d, nv);
        load8g<T>(reinterpret_cast<const T*>(p.x.data) + pix * p.x.ld + c0, xv, nv);
        const float bb = p.beta[pix], mm = p.m[pix], dd = p.ddot[pix], hb = p.dht[b] * p.inv_L;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dv = sp_dv(k.sc[j], k.mu[j], k.is[j], k.r0[j], k.r1[j], p.train, d[j] * mm, xv[j], bb);
            o[j] = d[j] + dv * bb + dd * mm * wv[j] + hb * lh[j] * mm;
            gacc[j] += dd * mm * xv[j];
        }
        T* dst = reinterpret_cast<T*>(p.dx.data) + pix * p.dx.ld + c0;
        if (p.accumulate) {
            float old[8]; load8g<T>(dst, old, nv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += old[j];
        }
        store8g<T>(dst, o, nv);
    }
    {
        const int lane = threadIdx.x & 63;
        const bool fixed = ((long)gridDim.x * 256) % cg == 0 && cg < 64;
        const int cfix = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cg) * 8;
        if (fixed) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gacc[j] = fold_stride(gacc[j], cg, lane);
            if (lane < cg) {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (cfix + j < C) atomicAdd(&red[cfix + j], gacc[j]);
            }
        } else if (last >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (last + j < C) atomicAdd(&red[last + j], gacc[j]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256)
        if (red[i] != 0.f) atomicAdd(p.gw + i, red[i]);
}
// tiny: parameter gradients of the layer
__global__ void sp_bwd_params_kernel(const float* red, const float* gw, const float* dht, const float* chansum, int n, int C,
                                     float inv_L, int train, float* d_gamma, float* d_beta, float* d_wv, float* d_bv, float* d_lh) {
    const int c = threadIdx.x;
    if (c < C) {
        if (train) { atomicAdd(d_gamma + c, red[C + c]); atomicAdd(d_beta + c, red[c]); }
        atomicAdd(d_wv + c, gw[c]);
        float a = 0.f;
        for (int b = 0; b < n; ++b) a += dht[b] * chansum[(long)b * C + c];
        atomicAdd(d_lh + c, a * inv_L);
    }
    if (c == 0) {
        float a = 0.f;
        for (int b = 0; b < n; ++b) a += dht[b];      // sum_p ddot = sum_b dht[b]
        atomicAdd(d_bv, a);
    }
}

// ---------------------------------------------------------------------------------------------
// gate backward: out = up*g(pred).  hi-res pass: dup (+)= dout*g, du[p] = (sum_c dout*up)*g(1-g);
// lo-res pass: transposed bilinear of du into d pred (channel 1: +, channel 0: -).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bil_src(int o, int n_in, int& i0, int& i1, float& w1) {
    float s = (o + 0.5f) * 0.5f - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s; i1 = min(i0 + 1, n_in - 1); w1 = s - (float)i0;
}
// POW2: the channel-group count is a power of two <= 64, so the lanes of one pixel are adjacent in a wave: the
// per-pixel dot product is folded with shuffles and stored once (no atomics, shift/mask instead of 64-bit
// division).  Otherwise one float atomic per lane into the zeroed du.
template <typename T, bool POW2>
__global__ __launch_bounds__(256) void gate_bwd_hi_kernel(View dout, View up, const float* gmap, View dup, int accumulate, float* du, int sh) {
    const int C = up.c, cg = (C + 7) / 8;
    const long pixels = (long)up.n * up.h * up.w;
    const long total = pixels * cg;
    for (long base = (long)blockIdx.x * 256; base < total; base += (long)gridDim.x * 256) {
        const long item = base + threadIdx.x;
        const bool live = item < total;
        int c0 = 0; long pix = 0;
        if (POW2) { c0 = (int)(item & (cg - 1)) * 8; pix = item >> sh; }
        else { c0 = (int)(item % cg) * 8; pix = item / cg; }
        float acc = 0.f, g = 0.f;
        if (live) {
            const int nv = min(8, C - c0);
            float d[8], u[8], o[8];
            load8g<T>(reinterpret_cast<const T*>(dout.data) + pix * dout.ld + c0, d, nv);
            load8g<T>(reinterpret_cast<const T*>(up.data) + pix * up.ld + c0, u, nv);
            g = gmap[pix];
#pragma unroll
            for (int j = 0; j < 8; ++j) { if (c0 + j < C) acc += d[j] * u[j]; o[j] = d[j] * g; }
            T* dst = reinterpret_cast<T*>(dup.data) + pix * dup.ld + c0;
            if (accumulate) {
                float old[8]; load8g<T>(dst, old, nv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += old[j];
            }
            store8g<T>(dst, o, nv);
        }
        if (POW2) {
            for (int off = 1; off < cg; off <<= 1) acc += __shfl_xor(acc, off, 64);
            if (live && c0 == 0) du[pix] = acc * g * (1.f - g);
        } else if (live) {
            atomicAdd(du + pix, acc * g * (1.f - g));
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_lo_kernel(const float* du, int H, int W, View dpred, int accumulate) {
    // each low-res pixel gathers from the <=4x4 hi-res pixels whose bilinear footprint includes it
    const long pixels = (long)dpred.n * dpred.h * dpred.w;
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < pixels; pix += (long)gridDim.x * 256) {
        const int X = (int)(pix % dpred.w); const long q = pix / dpred.w; const int Y = (int)(q % dpred.h); const long b = q / dpred.h;
        float acc = 0.f;
        for (int y = max(0, 2 * Y - 2); y <= min(H - 1, 2 * Y + 2); ++y) {
            int y0, y1; float wy;
            bil_src(y, dpred.h, y0, y1, wy);
            const float ky = (y0 == Y ? 1.f - wy : 0.f) + (y1 == Y ? wy : 0.f);
            if (ky == 0.f) continue;
            for (int x = max(0, 2 * X - 2); x <= min(W - 1, 2 * X + 2); ++x) {
                int x0, x1; float wx;
                bil_src(x, dpred.w, x0, x1, wx);
                const float kx = (x0 == X ? 1.f - wx : 0.f) + (x1 == X ? wx : 0.f);
                if (kx != 0.f) acc += ky * kx * du[(b * H + y) * W + x];
            }
        }
        T* d = reinterpret_cast<T*>(dpred.data) + pix * dpred.ld;
        float o0 = -acc, o1 = acc;
        if (accumulate) { o0 += st<T>::ld(d); o1 += st<T>::ld(d + 1); }
        st<T>::stv(d, o0); st<T>::stv(d + 1, o1);
    }
}

// ---------------------------------------------------------------------------------------------
// squeeze-excite backward.  forward: gate = sigmoid(W2 relu(W1 mean + b1) + b2), y = x*gate
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(View dxa, View x, float* dg) {
    extern __shared__ float red[];
    const int C = x.c, cg = C / 8, b = blockIdx.y;
    for (int i = threadIdx.x; i < C; i += 256) red[i] = 0.f;
    __syncthreads();
    const long hw = (long)x.h * x.w;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int last = -1;
    for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < hw * cg; item += (long)gridDim.x * 256) {
        const int c0 = (int)(item % cg) * 8; const long pix = (long)b * hw + item / cg;
        if (c0 != last && last >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { atomicAdd(&red[last + j], s[j]); s[j] = 0.f; }
        }
        last = c0;
        float d[8], xv[8];
        load8<T>(reinterpret_cast<const T*>(dxa.data) + pix * dxa.ld + c0, d);
        load8<T>(reinterpret_cast<const T*>(x.data) + pix * x.ld + c0, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += d[j] * xv[j];
    }
    if (last >= 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&red[last + j], s[j]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256)
        if (red[i] != 0.f) atomicAdd(dg + (long)b * C + i, red[i]);
}
// one workgroup per image: dgate -> (dW2, db2, dW1, db1) atomics, dmean[b,c]
__global__ void se_fc_bwd_kernel(const float* dg, const float* gate, const float* hid, const float* mean, const float* w1,
                                 const float* w2, int c, int hidden, float* dw1, float* db1, float* dw2, float* db2, float* dmean) {
    extern __shared__ float sm[];          // da2[c] | dh[hidden]
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
        const float g = gate[(long)b * c + i];
        const float da = dg[(long)b * c + i] * g * (1.f - g);
        sm[i] = da;
        atomicAdd(db2 + i, da);
        for (int j = 0; j < hidden; ++j) atomicAdd(dw2 + i * hidden + j, da * hid[(long)b * hidden + j]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < hidden; j += blockDim.x) {
        float a = 0.f;
        for (int i = 0; i < c; ++i) a += w2[i * hidden + j] * sm[i];
        a = hid[(long)b * hidden + j] > 0.f ? a : 0.f;           // relu'
        sm[c + j] = a;
        atomicAdd(db1 + j, a);
        for (int i = 0; i < c; ++i) atomicAdd(dw1 + j * c + i, a * mean[(long)b * c + i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += blockDim.x) {
        float a = 0.f;
        for (int j = 0; j < hidden; ++j) a += w1[j * c + i] * sm[c + j];
        dmean[(long)b * c + i] = a;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(View dxa, const float* gate, const float* dmean, float inv_hw, View dx,
                                                           int accumulate) {
    const int C = dxa.c, cg = C / 8;
    const long pixels = (long)dxa.n * dxa.h * dxa.w, hw = (long)dxa.h * dxa.w;
    for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < pixels * cg; item += (long)gridDim.x * 256) {
        const int c0 = (int)(item % cg) * 8; const long pix = item / cg; const long b = pix / hw;
        float d[8], o[8];
        load8<T>(reinterpret_cast<const T*>(dxa.data) + pix * dxa.ld + c0, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = d[j] * gate[b * C + c0 + j] + dmean[b * C + c0 + j] * inv_hw;
        T* dst = reinterpret_cast<T*>(dx.data) + pix * dx.ld + c0;
        if (accumulate) {
            float old[8]; load8<T>(dst, old);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += old[j];
        }
        store8<T>(dst, o);
    }
}

// dst (+)= src * s[b,c]
template <typename T>
// fold > 1: src holds fold * dst.n images; dst[b] (+)= sum_g src[g*n + b] * s[g*n + b] (the gradient of a tensor that was
// broadcast to `fold` statistic groups with a per-image scale each)
__global__ __launch_bounds__(256) void scale_bc_kernel(View src, const float* s, View dst, int accumulate, int fold) {
    const int C = src.c, cg = (C + 7) / 8;
    const long pixels = (long)dst.n * src.h * src.w, hw = (long)src.h * src.w;
    for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < pixels * cg; item += (long)gridDim.x * 256) {
        const int c0 = (int)(item % cg) * 8; const long pix = item / cg; const long b = pix / hw;
        const int nv = min(8, C - c0);
        float d[8];
        load8g<T>(reinterpret_cast<const T*>(src.data) + pix * src.ld + c0, d, nv);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] *= s[b * C + min(c0 + j, C - 1)];
        for (int g = 1; g < fold; ++g) {
            float e[8];
            load8g<T>(reinterpret_cast<const T*>(src.data) + (g * pixels + pix) * src.ld + c0, e, nv);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = fmaf(e[j], s[(g * dst.n + b) * C + min(c0 + j, C - 1)], d[j]);
        }
        T* o = reinterpret_cast<T*>(dst.data) + pix * dst.ld + c0;
        if (accumulate) {
            float old[8]; load8g<T>(o, old, nv);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] += old[j];
        }
        store8g<T>(o, d, nv);
    }
}

// ---------------------------------------------------------------------------------------------
// optimizer: global grad-norm clip (torch.nn.utils.clip_grad_norm_) + Adadelta with weight decay
// (torch.optim.Adadelta: lr, rho=0.9, eps=1e-6), model.py:145-166,273-278, on the flat buffers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, long n, float scale, float* out) {
    __shared__ float sh[16];
    float a = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float v = g[i] * scale; a += v * v; }
    a = block_sum(a, sh);
    if (threadIdx.x == 0) atomicAdd(out, a);
}
__global__ __launch_bounds__(256) void adadelta_kernel(float* p, const float* g, float* sq, float* acc, long n, float lr, float rho,
                                                       float eps, float wd, const float* sqnorm, float max_norm, float gscale,
                                                       const float* lr_dev) {
    if (lr_dev) lr = lr_dev[0];       // step size read at run time: a captured hipGraph follows ReduceLROnPlateau
    float clip = 1.f;
    if (max_norm > 0.f) { const float tn = sqrtf(sqnorm[0]); clip = fminf(1.f, max_norm / (tn + 1e-6f)); }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float w = p[i];
        const float gr = g[i] * gscale * clip + wd * w;
        const float s = rho * sq[i] + (1.f - rho) * gr * gr;
        const float d = sqrtf(acc[i] + eps) / sqrtf(s + eps) * gr;
        acc[i] = rho * acc[i] + (1.f - rho) * d * d;
        sq[i] = s;
        p[i] = w - lr * d;
    }
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
    do { if ((dtype) == ISA_BF16) { CALL_BF16; } else { CALL_F32; } } while (0)

extern "C" int isa_head_loss(const float* sums, const float* alpha, const int32_t* s_t, int64_t L, int32_t B,
                             const float* level_w, float ce_weight, float lambda_l, float lambda_r, float inv_iter,
                             float* baseline, int32_t training, float* coef, float* adv, float* scal, int32_t iters, void* stream) {
    if (!sums || !alpha || !s_t || !level_w || !baseline || !coef || !adv || !scal || B <= 0 || B > 1024) return ISA_EINVAL;
    if (iters < 1) iters = 1;
    HeadLossParams p{};
    p.sums = sums; p.alpha = alpha; p.s_t = s_t; p.L = L; p.B = B; p.iters = iters;
    for (int i = 0; i < 5; ++i) p.w[i] = level_w[i];
    p.ce_weight = ce_weight; p.lambda_l = lambda_l; p.lambda_r = lambda_r; p.inv_iter = inv_iter;
    p.baseline = baseline; p.training = training; p.coef = coef; p.adv = adv; p.scal = scal;
    hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), p);
    return launch_status();
}

extern "C" int isa_sem_loss(const float* sums, int32_t B, float* coef, float* scal, void* stream) {
    if (!sums || !coef || !scal || B <= 0 || B > 1024) return ISA_EINVAL;
    hipLaunchKernelGGL(sem_loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), sums, B, coef, scal);
    return launch_status();
}

extern "C" int isa_mask_loss_grad(const isa_tensor* pred, const float* target, const int64_t* onehot, const float* coef,
                                  const isa_tensor* dpred, int32_t accumulate, void* stream) {
    if (!tensor_ok(pred, 1) || !tensor_ok(dpred, 1) || pred->c != 2 || dpred->c != 2 || (!target && !onehot) || !coef) return ISA_EINVAL;
    const long L = (long)pred->h * pred->w;
    dim3 grid(grid_cap(cdiv(L, 256), 128), pred->n);
    DISPATCH_T(pred->dtype,
        hipLaunchKernelGGL(mask_loss_grad_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), mkview(pred), target, onehot, coef, mkview(dpred), accumulate),
        hipLaunchKernelGGL(mask_loss_grad_kernel<float>, grid, dim3(256), 0, as_stream(stream), mkview(pred), target, onehot, coef, mkview(dpred), accumulate));
    return launch_status();
}

extern "C" int isa_ins_softmax_bwd(const float* alpha, const int64_t* ins, const int32_t* idx, const int32_t* s_t,
                                   const float* adv, int32_t n, int32_t nobj, int64_t L, float* dmerge, int32_t nsrc, void* stream) {
    if (!alpha || !ins || !idx || !s_t || !adv || !dmerge || n <= 0) return ISA_EINVAL;
    if (nsrc <= 0) nsrc = n;
    if (n % nsrc) return ISA_EINVAL;
    dim3 grid(grid_cap(cdiv(L, 256), 128), nsrc);
    hipLaunchKernelGGL(ins_softmax_bwd_kernel, grid, dim3(256), 0, as_stream(stream), alpha, ins, idx, s_t, adv, nobj, (long)L, dmerge,
                       nsrc, n / nsrc);
    return launch_status();
}

extern "C" int isa_maskbn_bwd(const isa_tensor* e, const float* sem, const float* dmerge, const float* mean_var,
                              const float* w, float eps, const float* am, int32_t train, float* red4, float* k2,
                              float* dw, float* db, const isa_tensor* de, int32_t accumulate, void* stream) {
    if (!tensor_ok(e, 1) || !tensor_ok(de, 1) || !sem || !dmerge || !mean_var || !w || !am || !red4 || !k2 || !dw || !db) return ISA_EINVAL;
    const int grid = grid_cap(cdiv((long)e->n * e->h * e->w, 256));
    hipStream_t s = as_stream(stream);
    DISPATCH_T(e->dtype,
        hipLaunchKernelGGL(maskbn_bwd_reduce_kernel<bf16_t>, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, s, mkview(e), sem, dmerge, mean_var, w, eps, red4),
        hipLaunchKernelGGL(maskbn_bwd_reduce_kernel<float>, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, s, mkview(e), sem, dmerge, mean_var, w, eps, red4));
    hipLaunchKernelGGL(maskbn_bwd_finalize_kernel, dim3(1), dim3(64), 0, s, red4, am, mean_var, e->n, eps, train, k2, dw, db);
    DISPATCH_T(e->dtype,
        hipLaunchKernelGGL(maskbn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, mkview(e), sem, dmerge, mean_var, w, eps, am, k2, train, mkview(de), accumulate),
        hipLaunchKernelGGL(maskbn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, s, mkview(e), sem, dmerge, mean_var, w, eps, am, k2, train, mkview(de), accumulate));
    return launch_status();
}

extern "C" int isa_sp_bwd(const isa_tensor* dout, const isa_tensor* x, const float* beta, const float* m, const float* dot,
                          const float* rowstat, const float* chansum, const float* scale, const float* mean,
                          const float* invstd, const float* wv, const float* lh, const float* fcw, float count, int32_t train,
                          float* scratch /* zeroed: [2C red | C gw | n dht | n sdot | n*L dbeta | n*L ddot], n rounded up to 4 */,
                          const isa_tensor* dx, int32_t accumulate,
                          float* d_gamma, float* d_beta, float* d_wv, float* d_bv, float* d_lh, float* d_fcw, float* d_fcb,
                          void* stream) {
    if (!tensor_ok(dout, 8) || !tensor_ok(x, 8) || !tensor_ok(dx, 8) || dout->c != x->c || dx->c != x->c || !scratch) return ISA_EINVAL;
    const int C = x->c, n = x->n; const long L = (long)x->h * x->w;
    if ((long)n * L >= (1L << 32)) return ISA_EINVAL;
    const int n4 = ((n + 3) / 4) * 4;
    float* red = scratch; float* gw = red + 2 * C; float* dht = gw + C; float* sdot = dht + n4; float* dbeta = sdot + n4; float* ddot = dbeta + n * L;
    SpBwd p{};
    p.dout = mkview(dout); p.x = mkview(x); p.dx = mkview(dx);
    p.beta = beta; p.m = m; p.scale = scale; p.mean = mean; p.invstd = invstd; p.red = red; p.ddot = ddot; p.dht = dht;
    p.lh = lh; p.wv = wv; p.inv_count = 1.f / count; p.inv_L = 1.f / (float)L; p.train = train; p.accumulate = accumulate;
    p.out_red = red; p.dbeta_map = dbeta; p.gw = gw; p.sdot = sdot;
    hipStream_t s = as_stream(stream);
    const long items = (long)n * L * ((C + 7) / 8);
    const int grid = grid_keep_cg(grid_cap(cdiv(items, 256)), (C + 7) / 8);
    if (train)
        DISPATCH_T(x->dtype,
            hipLaunchKernelGGL(sp_bwd_reduce_kernel<bf16_t>, dim3(grid), dim3(256), 2 * C * 4, s, p),
            hipLaunchKernelGGL(sp_bwd_reduce_kernel<float>, dim3(grid), dim3(256), 2 * C * 4, s, p));
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(sp_bwd_dbeta_kernel<bf16_t>, dim3(grid_cap(cdiv((long)n * L, 256))), dim3(256), 5 * C * 4, s, p),
        hipLaunchKernelGGL(sp_bwd_dbeta_kernel<float>, dim3(grid_cap(cdiv((long)n * L, 256))), dim3(256), 5 * C * 4, s, p));
    hipLaunchKernelGGL(sp_bwd_rowc_kernel, dim3((unsigned)((L + SPB_CHUNK - 1) / SPB_CHUNK), n), dim3(256), 0, s, beta, dbeta, dot, m,
                       rowstat, fcw, sdot, L, ddot, dht, d_fcw, d_fcb);
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(sp_bwd_dx_kernel<bf16_t>, dim3(grid), dim3(256), C * 4, s, p),
        hipLaunchKernelGGL(sp_bwd_dx_kernel<float>, dim3(grid), dim3(256), C * 4, s, p));
    hipLaunchKernelGGL(sp_bwd_params_kernel, dim3(1), dim3(((C + 63) / 64) * 64), 0, s, red, gw, dht, chansum, n, C, p.inv_L, train,
                       d_gamma, d_beta, d_wv, d_bv, d_lh);
    return launch_status();
}

extern "C" int isa_gate_bwd(const isa_tensor* dout, const isa_tensor* up, const float* gmap, const isa_tensor* dup,
                            int32_t acc_up, float* du /* zeroed [n*H*W] */, const isa_tensor* dpred, int32_t acc_pred, void* stream) {
    if (!tensor_ok(dout, 8) || !tensor_ok(up, 8) || !tensor_ok(dup, 8) || !tensor_ok(dpred, 1) || !gmap || !du || dpred->c != 2) return ISA_EINVAL;
    hipStream_t s = as_stream(stream);
    const long items = (long)up->n * up->h * up->w * ((up->c + 7) / 8);
    const int g1 = grid_cap(cdiv(items, 256)), g2 = grid_cap(cdiv((long)dpred->n * dpred->h * dpred->w, 256));
    const int cgs = (up->c + 7) / 8;
    int sh = 0;
    while ((1 << sh) < cgs) ++sh;
    if ((1 << sh) == cgs && cgs <= 64) {
        DISPATCH_T(up->dtype,
            hipLaunchKernelGGL((gate_bwd_hi_kernel<bf16_t, true>), dim3(g1), dim3(256), 0, s, mkview(dout), mkview(up), gmap, mkview(dup), acc_up, du, sh),
            hipLaunchKernelGGL((gate_bwd_hi_kernel<float, true>), dim3(g1), dim3(256), 0, s, mkview(dout), mkview(up), gmap, mkview(dup), acc_up, du, sh));
    } else {
        DISPATCH_T(up->dtype,
            hipLaunchKernelGGL((gate_bwd_hi_kernel<bf16_t, false>), dim3(g1), dim3(256), 0, s, mkview(dout), mkview(up), gmap, mkview(dup), acc_up, du, sh),
            hipLaunchKernelGGL((gate_bwd_hi_kernel<float, false>), dim3(g1), dim3(256), 0, s, mkview(dout), mkview(up), gmap, mkview(dup), acc_up, du, sh));
    }
    DISPATCH_T(up->dtype,
        hipLaunchKernelGGL(gate_bwd_lo_kernel<bf16_t>, dim3(g2), dim3(256), 0, s, du, up->h, up->w, mkview(dpred), acc_pred),
        hipLaunchKernelGGL(gate_bwd_lo_kernel<float>, dim3(g2), dim3(256), 0, s, du, up->h, up->w, mkview(dpred), acc_pred));
    return launch_status();
}

extern "C" int isa_se_bwd(const isa_tensor* dxa, const isa_tensor* x, const float* gate, const float* hid, const float* mean,
                          const float* w1, const float* w2, int32_t hidden, float* dg /* zeroed [n*c] */, float* dmean /* [n*c] */,
                          float* dw1, float* db1, float* dw2, float* db2, const isa_tensor* dx, int32_t accumulate, void* stream) {
    if (!tensor_ok(dxa, 8) || !tensor_ok(x, 8) || !tensor_ok(dx, 8) || x->c % 8 || dxa->c != x->c || dx->c != x->c) return ISA_EINVAL;
    hipStream_t s = as_stream(stream);
    const int C = x->c, n = x->n;
    const long items = (long)x->h * x->w * (C / 8);
    dim3 g1(grid_cap(cdiv(items, 256), 128), n);
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(se_bwd_reduce_kernel<bf16_t>, g1, dim3(256), C * 4, s, mkview(dxa), mkview(x), dg),
        hipLaunchKernelGGL(se_bwd_reduce_kernel<float>, g1, dim3(256), C * 4, s, mkview(dxa), mkview(x), dg));
    hipLaunchKernelGGL(se_fc_bwd_kernel, dim3(n), dim3(64), (C + hidden) * 4, s, dg, gate, hid, mean, w1, w2, C, hidden, dw1, db1, dw2, db2, dmean);
    const int g3 = grid_cap(cdiv((long)n * items, 256));
    DISPATCH_T(x->dtype,
        hipLaunchKernelGGL(se_bwd_apply_kernel<bf16_t>, dim3(g3), dim3(256), 0, s, mkview(dxa), gate, dmean, 1.f / ((float)x->h * x->w), mkview(dx), accumulate),
        hipLaunchKernelGGL(se_bwd_apply_kernel<float>, dim3(g3), dim3(256), 0, s, mkview(dxa), gate, dmean, 1.f / ((float)x->h * x->w), mkview(dx), accumulate));
    return launch_status();
}

extern "C" int isa_scale_bc(const isa_tensor* src, const float* s_bc, const isa_tensor* dst, int32_t accumulate, void* stream) {
    if (!tensor_ok(src, 8) || !tensor_ok(dst, 8) || src->c != dst->c || src->dtype != dst->dtype || !s_bc) return ISA_EINVAL;
    if (src->h != dst->h || src->w != dst->w || dst->n <= 0 || src->n % dst->n) return ISA_EINVAL;
    const int fold = src->n / dst->n;                  // 1: plain; G: sum over the G groups of src
    const long items = (long)dst->n * src->h * src->w * ((src->c + 7) / 8);
    const int grid = grid_cap(cdiv(items, 256));
    DISPATCH_T(src->dtype,
        hipLaunchKernelGGL(scale_bc_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(src), s_bc, mkview(dst), accumulate, fold),
        hipLaunchKernelGGL(scale_bc_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), mkview(src), s_bc, mkview(dst), accumulate, fold));
    return launch_status();
}

extern "C" int isa_sqnorm(const float* g, int64_t n, float scale, float* out /* zeroed */, void* stream) {
    if (!g || !out || n <= 0) return ISA_EINVAL;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid_cap(cdiv(n, 256), 1024)), dim3(256), 0, as_stream(stream), g, (long)n, scale, out);
    return launch_status();
}

extern "C" int isa_adadelta(float* p, const float* g, float* sq, float* acc, int64_t n, float lr, float rho, float eps, float wd,
                            const float* sqnorm, float max_norm, float gscale, const float* lr_dev, void* stream) {
    if (!p || !g || !sq || !acc || n <= 0) return ISA_EINVAL;
    hipLaunchKernelGGL(adadelta_kernel, dim3(grid_cap(cdiv(n, 256), 2048)), dim3(256), 0, as_stream(stream), p, g, sq, acc, (long)n, lr,
                       rho, eps, wd, sqnorm, max_norm, gscale, lr_dev);
    return launch_status();
}
