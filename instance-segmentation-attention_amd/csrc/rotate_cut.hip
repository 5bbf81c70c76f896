// The two non-D4 augmentations the reference ships ENABLED (code/settings/CVPPP/training_settings.py:40 ROTATION = True,
// :50 CENTER_CUT = True), as AlignCollate.__preprocess applies them per image on the host with PIL (code/lib/dataset.py:
// 236-269), on the device - byte / index work, HBM streaming:
//   * isa_rotate_nearest_u8: the annotation rotator (dataset.py:142,243-249 -> preprocess.py:311-327 Image.rotate(angle,
//     NEAREST, expand=True)) for the instance planes and the semantic map: Pillow's Geometry.c affine_fixed - nearest
//     neighbour in 16.16 fixed point, zero fill - as one gather pass over pixel vectors.  The six coefficients come from
//     the host (Image.rotate's matrix is Python float arithmetic; isa_amd.data.rotate_geometry restates it).
//   * isa_rotate_bilinear_u8: the image rotator (dataset.py:140,241 -> preprocess.py:330-365 rotate_with_random_bg):
//     Geometry.c ImagingGenericTransform + bilinear_filter32RGB - IEEE double arithmetic with contraction off, (UINT8)
//     truncation - and the composite over the drawn background colour (alpha is 0 or 255 only: a select).
//   * isa_cover_rows_u8 / isa_plane_sums_u8 / isa_crop_planes_u8: CenterCut (dataset.py:252-269 -> preprocess.py:239-264):
//     the candidate centres (pixels covered by exactly one plane), the per-plane window sums of the has-object filter,
//     and the window crop with the surviving planes compacted and zero planes appended (dataset.py:304-311).
// Bit-exact against oracle/rotate_ref.py, which is pinned against the installed Pillow (tests/test_oracle_rotate.py).
#include "common.hpp"

namespace {

struct RotFixed { int a0, a1, a2, a3, a4, a5; };

template <int V>   // V bytes per lane: 16 (c % 16 == 0, aligned) or 1
__global__ __launch_bounds__(256) void rotate_nearest_kernel(const uint8_t* src, uint8_t* dst, int n, int h0, int w0, int c, int h, int w,
                                                             RotFixed k) {
    const int cv = c / V;
    const long per_img = (long)h * w * cv, total = (long)n * per_img;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per_img); const long r = i - (long)b * per_img;
        const int v = (int)(r % cv); const long pix = r / cv;
        const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
        // affine_fixed accumulates xx = a2 + y * a1 + x * a0 by additions in int (the four corners were checked to fit);
        // the same numbers here from 64-bit products
        const int xin = (int)(((long long)k.a2 + (long long)y * k.a1 + (long long)x * k.a0) >> 16);
        const int yin = (int)(((long long)k.a5 + (long long)y * k.a4 + (long long)x * k.a3) >> 16);
        const bool ok = xin >= 0 && xin < w0 && yin >= 0 && yin < h0;
        const long so = (((long)b * h0 + yin) * w0 + xin) * c + (long)v * V, doff = i * V;
        if (V == 16) *reinterpret_cast<uint4*>(dst + doff) = ok ? *reinterpret_cast<const uint4*>(src + so) : uint4{0, 0, 0, 0};
        else dst[doff] = ok ? src[so] : (uint8_t)0;
    }
}

struct RotDouble { double m[6]; uint8_t bg[4]; };

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void rotate_bilinear_kernel(const uint8_t* src, uint8_t* dst, int n, int h0, int w0, int c, int h, int w,
                                                              RotDouble k) {
    const long per_img = (long)h * w, total = (long)n * per_img;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per_img); const long pix = i - (long)b * per_img;
        const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
        // Geometry.c affine_transform: the output pixel's centre through the matrix
        double xin = k.m[0] * (x + 0.5) + k.m[1] * (y + 0.5) + k.m[2];
        double yin = k.m[3] * (x + 0.5) + k.m[4] * (y + 0.5) + k.m[5];
        uint8_t* o = dst + i * c;
        if (xin < 0.0 || xin >= (double)w0 || yin < 0.0 || yin >= (double)h0) {       // transparent: the background shows
            for (int ch = 0; ch < c; ++ch) o[ch] = k.bg[ch];
            continue;
        }
        xin -= 0.5; yin -= 0.5;
        const double fx = floor(xin), fy = floor(yin);
        const double dx = xin - fx, dy = yin - fy;
        const int xi = (int)fx, yi = (int)fy;
        const int xa = min(max(xi, 0), w0 - 1), xb = min(max(xi + 1, 0), w0 - 1);
        const int ya = min(max(yi, 0), h0 - 1);
        const bool has2 = yi + 1 >= 0 && yi + 1 < h0;
        const uint8_t* r0 = src + ((long)b * h0 + ya) * w0 * c;
        const uint8_t* r1 = src + ((long)b * h0 + (has2 ? yi + 1 : ya)) * w0 * c;
        for (int ch = 0; ch < c; ++ch) {
            const double p00 = r0[xa * c + ch], p01 = r0[xb * c + ch];
            double v1 = p00 + (p01 - p00) * dx, v2 = v1;
            if (has2) { const double p10 = r1[xa * c + ch], p11 = r1[xb * c + ch]; v2 = p10 + (p11 - p10) * dx; }
            v1 = v1 + (v2 - v1) * dy;
            o[ch] = (uint8_t)v1;                                                         // (UINT8) v1: truncation
        }
    }
}

// single[b,y,x] = (sum_k planes[b,y,x,k] == 1); row_counts[b,y] = number of such pixels in the row (one wave per row)
__global__ __launch_bounds__(64) void cover_rows_kernel(const uint8_t* planes, int h, int w, int k, uint8_t* single, int32_t* row_counts) {
    const long row = blockIdx.x;                       // b * h + y
    int cnt = 0;
    for (int x = threadIdx.x; x < w; x += 64) {
        const uint8_t* p = planes + (row * w + x) * k;
        int s = 0;
        for (int j = 0; j < k; ++j) s += p[j];
        const int one = s == 1;
        single[row * w + x] = (uint8_t)one;
        cnt += one;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (threadIdx.x == 0) row_counts[row] = cnt;
}

// sums[b][c] += window sums of every channel (exact integers; int64 atomics via unsigned long long)
__global__ __launch_bounds__(256) void plane_sums_kernel(const uint8_t* src, int H, int W, int c, int y0, int x0, int h, int w,
                                                         unsigned long long* sums) {
    extern __shared__ unsigned int part[];             // [c]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < c; i += 256) part[i] = 0;
    __syncthreads();
    const long items = (long)h * w * c;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % c); const long pix = i / c;
        const int y = (int)(pix / w), x = (int)(pix - (long)y * w);
        const unsigned v = src[(((long)b * H + y0 + y) * W + x0 + x) * c + ch];
        if (v) atomicAdd(&part[ch], v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256)
        if (part[i]) atomicAdd(sums + (long)b * c + i, (unsigned long long)part[i]);
}

// dst[b,y,x,j] = chan[j] >= 0 ? src[b, y0+y, x0+x, chan[j]] : 0
__global__ __launch_bounds__(256) void crop_planes_kernel(const uint8_t* src, int n, int H, int W, int c, int y0, int x0, uint8_t* dst,
                                                          int h, int w, int cd, const int32_t* chan) {
    const long total = (long)n * h * w * cd;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i % cd); const long pix = i / cd;
        const int x = (int)(pix % w); const long q = pix / w; const int y = (int)(q % h); const int b = (int)(q / h);
        const int sc = chan ? chan[j] : j;
        dst[i] = (sc >= 0 && sc < c) ? src[(((long)b * H + y0 + y) * W + x0 + x) * c + sc] : (uint8_t)0;
    }
}

}  // namespace

extern "C" int isa_rotate_nearest_u8(const uint8_t* src, int32_t n, int32_t h0, int32_t w0, int32_t c, uint8_t* dst, int32_t h,
                                     int32_t w, const int32_t* coef_host, void* stream) {
    if (!src || !dst || src == dst || !coef_host || n <= 0 || h0 <= 0 || w0 <= 0 || c <= 0 || h <= 0 || w <= 0) return ISA_EINVAL;
    if (h0 > 32767 || w0 > 32767 || h > 32767 || w > 32767) return ISA_EINVAL;            // the range Pillow's 16.16 fixed-point path covers
    const RotFixed k{coef_host[0], coef_host[1], coef_host[2], coef_host[3], coef_host[4], coef_host[5]};
    const bool vec = (c % 16 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0);
    const long items = (long)n * h * w * (vec ? c / 16 : c);
    const int grid = grid_cap(cdiv(items, 256), 256 * 16);
    if (vec) hipLaunchKernelGGL(rotate_nearest_kernel<16>, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, h0, w0, c, h, w, k);
    else hipLaunchKernelGGL(rotate_nearest_kernel<1>, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, h0, w0, c, h, w, k);
    return launch_status();
}

extern "C" int isa_rotate_bilinear_u8(const uint8_t* src, int32_t n, int32_t h0, int32_t w0, int32_t c, uint8_t* dst, int32_t h,
                                      int32_t w, const double* matrix_host, const uint8_t* bg_host, void* stream) {
    if (!src || !dst || src == dst || !matrix_host || !bg_host || n <= 0 || h0 <= 0 || w0 <= 0 || c <= 0 || c > 4 || h <= 0 || w <= 0)
        return ISA_EINVAL;
    RotDouble k{};
    for (int i = 0; i < 6; ++i) k.m[i] = matrix_host[i];
    for (int i = 0; i < c; ++i) k.bg[i] = bg_host[i];
    const int grid = grid_cap(cdiv((long)n * h * w, 256), 256 * 16);
    hipLaunchKernelGGL(rotate_bilinear_kernel, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, h0, w0, c, h, w, k);
    return launch_status();
}

extern "C" int isa_cover_rows_u8(const uint8_t* planes, int32_t n, int32_t h, int32_t w, int32_t k, uint8_t* single,
                                 int32_t* row_counts, void* stream) {
    if (!planes || !single || !row_counts || n <= 0 || h <= 0 || w <= 0 || k <= 0) return ISA_EINVAL;
    hipLaunchKernelGGL(cover_rows_kernel, dim3((unsigned)((long)n * h)), dim3(64), 0, as_stream(stream), planes, h, w, k, single, row_counts);
    return launch_status();
}

extern "C" int isa_plane_sums_u8(const uint8_t* src, int32_t n, int32_t H, int32_t W, int32_t c, int32_t y0, int32_t x0, int32_t h,
                                 int32_t w, int64_t* sums, void* stream) {
    if (!src || !sums || n <= 0 || c <= 0 || c > 4096 || y0 < 0 || x0 < 0 || h <= 0 || w <= 0 || y0 + h > H || x0 + w > W) return ISA_EINVAL;
    dim3 grid(grid_cap(cdiv((long)h * w * c, 256), 512), n);
    hipLaunchKernelGGL(plane_sums_kernel, grid, dim3(256), c * sizeof(unsigned int), as_stream(stream), src, H, W, c, y0, x0, h, w,
                       reinterpret_cast<unsigned long long*>(sums));
    return launch_status();
}

extern "C" int isa_crop_planes_u8(const uint8_t* src, int32_t n, int32_t H, int32_t W, int32_t c, int32_t y0, int32_t x0, uint8_t* dst,
                                  int32_t h, int32_t w, int32_t cd, const int32_t* chan_dev, void* stream) {
    if (!src || !dst || src == dst || n <= 0 || c <= 0 || cd <= 0 || y0 < 0 || x0 < 0 || h <= 0 || w <= 0 || y0 + h > H || x0 + w > W)
        return ISA_EINVAL;
    const int grid = grid_cap(cdiv((long)n * h * w * cd, 256), 256 * 16);
    hipLaunchKernelGGL(crop_planes_kernel, dim3(grid), dim3(256), 0, as_stream(stream), src, n, H, W, c, y0, x0, dst, h, w, cd, chan_dev);
    return launch_status();
}
