// Nearest-neighbour resize of the annotation planes, the `ann_resizer` of the reference's collate function
// (code/lib/dataset.py:162,168 -> utils.py:26-27 -> torchvision Resize -> PIL Image.resize(NEAREST)), applied on the
// host to each of the 32 instance planes and the semantic map of every image (dataset.py:293-320).
// Pillow (third-party, absent from /root/reference, version unpinned there) resizes NEAREST through
// ImagingScaleAffine (libImaging/Geometry.c): source column of output x is (int)xo with xo = 0.5*s, then xo += s per
// column (s = in/out, accumulated in double), rows likewise.  The tables are built on the host with exactly that
// loop in IEEE double - bit-identical to Pillow - and travel as kernel arguments; the device pass is a gather of
// pixel vectors.  oracle/resize_ref.py restates the same rule and is pinned against the installed Pillow.
#include "common.hpp"

namespace {

constexpr int TAB_MAX = 768;                      // output rows / columns per table (3 KB of kernel arguments)
struct ResizeTabs { uint16_t y[TAB_MAX]; uint16_t x[TAB_MAX]; };

template <int V>
__global__ __launch_bounds__(256) void resize_nearest_kernel(const uint8_t* src, uint8_t* dst, int n, int h0, int w0, int c,
                                                             int h, int w, ResizeTabs t) {
    const int cv = c / V;
    const long per_img = (long)h * w * cv, total = (long)n * per_img;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per_img); const long r = i - (long)b * per_img;
        const int v = (int)(r % cv); const long pix = r / cv;
        const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
        const long so = (((long)b * h0 + t.y[y]) * w0 + t.x[x]) * c + (long)v * V;
        const long doff = (((long)b * h + y) * w + x) * c + (long)v * V;
        if (V == 16) *reinterpret_cast<uint4*>(dst + doff) = *reinterpret_cast<const uint4*>(src + so);
        else dst[doff] = src[so];
    }
}

// Pillow's ImagingScaleAffine column table: xo starts at half a step and accumulates
void scale_table(int n_in, int n_out, uint16_t* tab) {
    const double a = (double)n_in / (double)n_out;
    double o = a * 0.5;
    for (int i = 0; i < n_out; ++i) {
        int s = o < 0.0 ? -1 : (int)o;
        if (s < 0) s = 0;
        if (s > n_in - 1) s = n_in - 1;             // Pillow leaves such pixels untouched; cannot happen for a pure scale
        tab[i] = (uint16_t)s;
        o += a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Bilinear image resize, the `img_resizer` of the reference (code/lib/dataset.py:160-161,166-167 and prediction.py:37:
// utils.py:26-27 -> PIL.Image.resize((w, h), BILINEAR)), bit-identical to Pillow's libImaging/Resample.c for 8-bit
// channels: a separable triangle filter whose support grows with the down-scaling factor, coefficients computed in
// double and rounded to 22-bit fixed point (here by a small kernel, in IEEE double with contraction off: the same
// operations in the same order), a horizontal pass into a uint8 intermediate, then a vertical pass; every pass starts
// from 1 << 21 and clips to [0, 255].  oracle/resize_ref.py restates it and is pinned against the installed Pillow.
constexpr int RS_PREC = 32 - 8 - 2;

__global__ __launch_bounds__(256) void bilin_coeffs_kernel(int n_in, int n_out, int ksize, int* bounds, int* kk) {
#pragma clang fp contract(off)
    const int xx = blockIdx.x * 256 + threadIdx.x;
    if (xx >= n_out) return;
    const double scale = (double)n_in / (double)n_out;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > n_in) xmax = n_in;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        ww += t < 1.0 ? 1.0 - t : 0.0;
    }
    for (int x = 0; x < ksize; ++x) {
        int v = 0;
        if (x < xmax) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            double w = t < 1.0 ? 1.0 - t : 0.0;
            if (ww != 0.0) w /= ww;
            v = w < 0.0 ? (int)(-0.5 + w * (double)(1 << RS_PREC)) : (int)(0.5 + w * (double)(1 << RS_PREC));
        }
        kk[(long)xx * ksize + x] = v;
    }
    bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
}

// out[b, y, xx, :] = clip8((sum_x in[b, y, xmin + x, :] * k[x] + 2^21) >> 22): one thread per output pixel, c <= 4
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* src, uint8_t* dst, int n, int h, int w_in, int w_out, int c,
                                                         const int* bounds, const int* kk, int ksize) {
    const long total = (long)n * h * w_out;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int xx = (int)(i % w_out); const long row = i / w_out;
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const int* k = kk + (long)xx * ksize;
        const uint8_t* sp = src + (row * w_in + xmin) * c;
        int acc[4] = {1 << (RS_PREC - 1), 1 << (RS_PREC - 1), 1 << (RS_PREC - 1), 1 << (RS_PREC - 1)};
        for (int x = 0; x < cnt; ++x)
            for (int ch = 0; ch < c; ++ch) acc[ch] += (int)sp[x * c + ch] * k[x];
        for (int ch = 0; ch < c; ++ch) dst[i * c + ch] = (uint8_t)min(max(acc[ch] >> RS_PREC, 0), 255);
    }
}
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* src, uint8_t* dst, int n, int h_in, int h_out, int w, int c,
                                                         const int* bounds, const int* kk, int ksize) {
    const long total = (long)n * h_out * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % w); const long t = i / w; const int yy = (int)(t % h_out); const long b = t / h_out;
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const int* k = kk + (long)yy * ksize;
        const uint8_t* sp = src + ((b * h_in + ymin) * w + x) * c;
        int acc[4] = {1 << (RS_PREC - 1), 1 << (RS_PREC - 1), 1 << (RS_PREC - 1), 1 << (RS_PREC - 1)};
        for (int y = 0; y < cnt; ++y)
            for (int ch = 0; ch < c; ++ch) acc[ch] += (int)sp[(long)y * w * c + ch] * k[y];
        for (int ch = 0; ch < c; ++ch) dst[i * c + ch] = (uint8_t)min(max(acc[ch] >> RS_PREC, 0), 255);
    }
}

static int bilin_ksize(int n_in, int n_out) {
    const double scale = (double)n_in / (double)n_out, fs = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(fs) * 2 + 1;
}

}  // namespace

extern "C" int64_t isa_resize_bilinear_ws_bytes(int32_t n, int32_t h0, int32_t w0, int32_t c, int32_t h, int32_t w) {
    if (n <= 0 || h0 <= 0 || w0 <= 0 || c <= 0 || h <= 0 || w <= 0) return 0;
    const int64_t ints = 2L * w + (int64_t)w * bilin_ksize(w0, w) + 2L * h + (int64_t)h * bilin_ksize(h0, h);
    return ((ints * 4 + 255) & ~255L) + (int64_t)n * h0 * w * c;
}

extern "C" int isa_resize_bilinear_u8(const uint8_t* src, int32_t n, int32_t h0, int32_t w0, int32_t c, uint8_t* dst,
                                      int32_t h, int32_t w, void* ws, int64_t ws_bytes, void* stream) {
    if (!src || !dst || src == dst || n <= 0 || h0 <= 0 || w0 <= 0 || c <= 0 || c > 4 || h <= 0 || w <= 0 || !ws) return ISA_EINVAL;
    if (ws_bytes < isa_resize_bilinear_ws_bytes(n, h0, w0, c, h, w)) return ISA_ENOMEM;
    hipStream_t s = as_stream(stream);
    const int kx = bilin_ksize(w0, w), ky = bilin_ksize(h0, h);
    int* xb = reinterpret_cast<int*>(ws); int* xk = xb + 2 * w; int* yb = xk + (long)w * kx; int* yk = yb + 2 * h;
    const int64_t ints = 2L * w + (int64_t)w * kx + 2L * h + (int64_t)h * ky;
    uint8_t* tmp = reinterpret_cast<uint8_t*>(ws) + ((ints * 4 + 255) & ~255L);
    const uint8_t* hsrc = src;
    if (w0 != w) {                                    // horizontal pass first (ImagingResample), skipped on an unchanged axis
        hipLaunchKernelGGL(bilin_coeffs_kernel, dim3(cdiv(w, 256)), dim3(256), 0, s, w0, w, kx, xb, xk);
        uint8_t* out = h0 != h ? tmp : dst;
        hipLaunchKernelGGL(resample_h_kernel, dim3(grid_cap(cdiv((long)n * h0 * w, 256), 4096)), dim3(256), 0, s, src, out, n, h0, w0, w, c, xb, xk, kx);
        hsrc = out;
    }
    if (h0 != h) {
        hipLaunchKernelGGL(bilin_coeffs_kernel, dim3(cdiv(h, 256)), dim3(256), 0, s, h0, h, ky, yb, yk);
        hipLaunchKernelGGL(resample_v_kernel, dim3(grid_cap(cdiv((long)n * h * w, 256), 4096)), dim3(256), 0, s, hsrc, dst, n, h0, h, w, c, yb, yk, ky);
    } else if (w0 == w) {
        if (hipMemcpyAsync(dst, src, (size_t)n * h * w * c, hipMemcpyDeviceToDevice, s) != hipSuccess) return ISA_ELAUNCH;
    }
    return launch_status();
}

extern "C" int isa_resize_nearest_u8(const uint8_t* src, int32_t n, int32_t h0, int32_t w0, int32_t c, uint8_t* dst,
                                     int32_t h, int32_t w, void* stream) {
    if (!src || !dst || src == dst || n <= 0 || h0 <= 0 || w0 <= 0 || c <= 0 || h <= 0 || w <= 0) return ISA_EINVAL;
    if (h > TAB_MAX || w > TAB_MAX || h0 > 65535 || w0 > 65535) return ISA_EINVAL;
    ResizeTabs t;
    scale_table(h0, h, t.y);
    scale_table(w0, w, t.x);
    const bool vec = (c % 16 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0);
    const long items = (long)n * h * w * (vec ? c / 16 : c);
    const int grid = grid_cap(cdiv(items, 256), 256 * 16);
    if (vec) hipLaunchKernelGGL(resize_nearest_kernel<16>, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, h0, w0, c, h, w, t);
    else hipLaunchKernelGGL(resize_nearest_kernel<1>, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, h0, w0, c, h, w, t);
    return launch_status();
}
