// The exact (index-permuting) augmentations of AlignCollate.__preprocess (code/lib/dataset.py:185-233) on the device:
// horizontal flip, vertical flip, transpose, rotation by a multiple of 90 degrees (preprocess.py:171 FLIP_LEFT_RIGHT,
// :218 FLIP_TOP_BOTTOM, :286 TRANSPOSE, :323 Image.rotate(angle, expand=True) with angle in {0,90,180,270} = the
// counter-clockwise quarter turns), applied in that order to a uint8 NHWC batch: the RGB image [n,s,s,3], the
// semantic map [n,s,s,1] and the instance planes [n,s,s,32] all take the same per-image op code.  On the host the
// reference runs four PIL transposes per instance plane per image (32 planes); here the composed permutation is one
// gather pass: byte work, HBM streaming (pixel vectors of 16 bytes when the channel count allows).
// op code: bit0 hflip, bit1 vflip, bit2 transpose, bits 3-4 quarter turns.  Rectangular images (the reference augments
// the ORIGINAL image, e.g. 2362 x 672, before the resize): an op that transposes XOR turns an odd number of quarters
// exchanges the axes, so a call covers images whose ops agree in that parity (`swap_axes`), dst is [n,w,h,c] then.
#include "common.hpp"

namespace {

// source pixel of output pixel (y, x) of an h x w image under op (flips, then transpose, then q quarter turns)
__device__ __forceinline__ void d4_source(int op, int h, int w, int y, int x, int& sy, int& sx) {
    const int q = (op >> 3) & 3;
    const int th = (op & 4) ? w : h, tw = (op & 4) ? h : w;  // dims after the transpose stage
    int y1 = y, x1 = x;
    if (q == 1) { y1 = x; x1 = tw - 1 - y; }                // np.rot90(a, 1)[y, x] = a[x, aw-1-y]
    else if (q == 2) { y1 = th - 1 - y; x1 = tw - 1 - x; }
    else if (q == 3) { y1 = th - 1 - x; x1 = y; }           // np.rot90(a, 3)[y, x] = a[ah-1-x, y]
    if (op & 4) { const int t = y1; y1 = x1; x1 = t; }      // transpose
    if (op & 2) y1 = h - 1 - y1;                             // vertical flip
    if (op & 1) x1 = w - 1 - x1;                             // horizontal flip
    sy = y1; sx = x1;
}

template <int V>   // V bytes per lane: 16 (c % 16 == 0, aligned) or 1
__global__ __launch_bounds__(256) void d4_kernel(const uint8_t* src, uint8_t* dst, int n, int h, int w, int c, int swap, const int32_t* ops) {
    const int cv = c / V;
    const int oh = swap ? w : h, ow = swap ? h : w;
    const long per_img = (long)oh * ow * cv, total = (long)n * per_img;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per_img); const long r = i - (long)b * per_img;
        const int v = (int)(r % cv); const long pix = r / cv;
        const int y = (int)(pix / ow), x = (int)(pix - (long)y * ow);
        const int op = ops[b] & 31;
        int sy, sx;
        d4_source(op, h, w, y, x, sy, sx);
        // an op of the other axis parity has no image of this shape (h != w): the caller groups by parity; zeros mark a misuse
        const bool ok = h == w || ((((op >> 2) ^ (op >> 3)) & 1) == swap);
        const long so = (((long)b * h + sy) * w + sx) * c + (long)v * V, doff = i * V;
        if (V == 16) *reinterpret_cast<uint4*>(dst + doff) = ok ? *reinterpret_cast<const uint4*>(src + so) : uint4{0, 0, 0, 0};
        else dst[doff] = ok ? src[so] : (uint8_t)0;
    }
}

}  // namespace

extern "C" int isa_d4_augment(const uint8_t* src, uint8_t* dst, int32_t n, int32_t h, int32_t w, int32_t c, int32_t swap_axes,
                              const int32_t* ops_dev, void* stream) {
    if (!src || !dst || src == dst || !ops_dev || n <= 0 || h <= 0 || w <= 0 || c <= 0) return ISA_EINVAL;
    const int swap = swap_axes ? 1 : 0;
    const bool vec = (c % 16 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0);
    const long items = (long)n * h * w * (vec ? c / 16 : c);
    const int grid = grid_cap(cdiv(items, 256), 256 * 16);
    if (vec) hipLaunchKernelGGL(d4_kernel<16>, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, h, w, c, swap, ops_dev);
    else hipLaunchKernelGGL(d4_kernel<1>, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, h, w, c, swap, ops_dev);
    return launch_status();
}
