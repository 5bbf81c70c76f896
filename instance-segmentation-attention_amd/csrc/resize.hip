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

}  // namespace

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
