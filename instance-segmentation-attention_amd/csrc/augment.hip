// The exact (index-permuting) augmentations of AlignCollate.__preprocess (code/lib/dataset.py:185-233) on the device:
// horizontal flip, vertical flip, transpose, rotation by a multiple of 90 degrees (preprocess.py:171 FLIP_LEFT_RIGHT,
// :218 FLIP_TOP_BOTTOM, :286 TRANSPOSE, :323 Image.rotate(angle, expand=True) with angle in {0,90,180,270} = the
// counter-clockwise quarter turns), applied in that order to a uint8 NHWC batch: the RGB image [n,s,s,3], the
// semantic map [n,s,s,1] and the instance planes [n,s,s,32] all take the same per-image op code.  On the host the
// reference runs four PIL transposes per instance plane per image (32 planes); here the composed permutation is one
// gather pass: byte work, HBM streaming (pixel vectors of 16 bytes when the channel count allows).
// op code: bit0 hflip, bit1 vflip, bit2 transpose, bits 3-4 quarter turns.  Square images only (a transposing op on
// a non-square image changes the tensor shape: the collated batch is square, data_settings.py IMAGE_HEIGHT/WIDTH).
#include "common.hpp"

namespace {

// source pixel of output pixel (y, x) on an s x s image under op
__device__ __forceinline__ void d4_source(int op, int s, int y, int x, int& sy, int& sx) {
    const int q = (op >> 3) & 3;
    int y1 = y, x1 = x;
    if (q == 1) { y1 = x; x1 = s - 1 - y; }                 // np.rot90(a, 1)[y, x] = a[x, s-1-y]
    else if (q == 2) { y1 = s - 1 - y; x1 = s - 1 - x; }
    else if (q == 3) { y1 = s - 1 - x; x1 = y; }
    if (op & 4) { const int t = y1; y1 = x1; x1 = t; }      // transpose
    if (op & 2) y1 = s - 1 - y1;                             // vertical flip
    if (op & 1) x1 = s - 1 - x1;                             // horizontal flip
    sy = y1; sx = x1;
}

template <int V>   // V bytes per lane: 16 (c % 16 == 0, aligned) or 1
__global__ __launch_bounds__(256) void d4_kernel(const uint8_t* src, uint8_t* dst, int n, int s, int c, const int32_t* ops) {
    const int cv = c / V;
    const long per_img = (long)s * s * cv, total = (long)n * per_img;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per_img); const long r = i - (long)b * per_img;
        const int v = (int)(r % cv); const long pix = r / cv;
        const int y = (int)(pix / s), x = (int)(pix - (long)y * s);
        int sy, sx;
        d4_source(ops[b] & 31, s, y, x, sy, sx);
        const long so = (((long)b * s + sy) * s + sx) * c + (long)v * V, doff = (((long)b * s + y) * s + x) * c + (long)v * V;
        if (V == 16) *reinterpret_cast<uint4*>(dst + doff) = *reinterpret_cast<const uint4*>(src + so);
        else dst[doff] = src[so];
    }
}

}  // namespace

extern "C" int isa_d4_augment(const uint8_t* src, uint8_t* dst, int32_t n, int32_t s, int32_t c, const int32_t* ops_dev,
                              void* stream) {
    if (!src || !dst || src == dst || !ops_dev || n <= 0 || s <= 0 || c <= 0) return ISA_EINVAL;
    const bool vec = (c % 16 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0);
    const long items = (long)n * s * s * (vec ? c / 16 : c);
    const int grid = grid_cap(cdiv(items, 256), 256 * 16);
    if (vec) hipLaunchKernelGGL(d4_kernel<16>, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, s, c, ops_dev);
    else hipLaunchKernelGGL(d4_kernel<1>, dim3(grid), dim3(256), 0, as_stream(stream), src, dst, n, s, c, ops_dev);
    return launch_status();
}
