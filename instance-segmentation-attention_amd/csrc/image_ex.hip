// ImageEx + ToTensor + Standardization (code/lib/utils.py:90-113, code/lib/preprocess.py:192-195) as one pass:
// uint8 RGB [n,h,w,3] -> the network's NHWC input view, 21 channels (ld 24, pad channels written as zero), value
// (x - 0.5) * 2 with x = [rgb(0..255), L*a*b*, HSV, YUV, YCbCr, HED, YIQ].  Reads 3 bytes, writes 48 (bf16) per
// pixel: HBM streaming, the transcendental work (3 pow, 3 cbrt, 3 log per pixel) stays under the memory time.
// The six conversions restate scikit-image's published formulas (>= 0.17); see oracle/image_ex_ref.py for the
// version caveat ("parity unpinned": the reference's dependency is absent and unversioned).
#include "common.hpp"

namespace {

__device__ __forceinline__ float srgb_lin(float c) { return c > 0.04045f ? powf((c + 0.055f) * (1.f / 1.055f), 2.4f) : c * (1.f / 12.92f); }
__device__ __forceinline__ float lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.f / 116.f; }

template <typename T>
__global__ __launch_bounds__(256) void image_ex_kernel(const uint8_t* rgb, long pixels, T* out, int ld) {
    // the input is 8-bit: sRGB linearisation (pow) and the stain optical density (log) have 256 possible values
    // each, so they are tabulated once per workgroup instead of evaluated three times per pixel
    __shared__ float lut_lin[256], lut_od[256];
    {
        const float c = (float)threadIdx.x * (1.f / 255.f);
        lut_lin[threadIdx.x] = srgb_lin(c);
        lut_od[threadIdx.x] = logf(fmaxf(c, 1e-6f)) * -0.07238241365054197f;
    }
    __syncthreads();
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long)gridDim.x * 256) {
        const int Ri = rgb[3 * p], Gi = rgb[3 * p + 1], Bi = rgb[3 * p + 2];
        const float R = (float)Ri, G = (float)Gi, B = (float)Bi;
        const float r = R * (1.f / 255.f), g = G * (1.f / 255.f), b = B * (1.f / 255.f);
        float v[24];
        v[0] = R; v[1] = G; v[2] = B;
        {   // CIE L*a*b*, D65 / 2 degree observer
            const float lr = lut_lin[Ri], lg = lut_lin[Gi], lb = lut_lin[Bi];
            const float X = (0.412453f * lr + 0.357580f * lg + 0.180423f * lb) * (1.f / 0.95047f);
            const float Y = 0.212671f * lr + 0.715160f * lg + 0.072169f * lb;
            const float Z = (0.019334f * lr + 0.119193f * lg + 0.950227f * lb) * (1.f / 1.08883f);
            const float fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
            v[3] = 116.f * fy - 16.f; v[4] = 500.f * (fx - fy); v[5] = 200.f * (fy - fz);
        }
        {   // HSV: later branches win ties (red, green, blue), hue in [0,1)
            const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b)), d = mx - mn;
            float h = 0.f, s = 0.f;
            if (d != 0.f) {
                s = d / mx;
                if (r == mx) h = (g - b) / d;
                if (g == mx) h = 2.f + (b - r) / d;
                if (b == mx) h = 4.f + (r - g) / d;
                h *= (1.f / 6.f);
                h -= floorf(h);
            }
            v[6] = h; v[7] = s; v[8] = mx;
        }
        v[9] = 0.299f * r + 0.587f * g + 0.114f * b;                               // YUV
        v[10] = -0.14714119f * r - 0.28886916f * g + 0.43601035f * b;
        v[11] = 0.61497538f * r - 0.51496512f * g - 0.10001026f * b;
        v[12] = 65.481f * r + 128.553f * g + 24.966f * b + 16.f;                   // YCbCr (studio range)
        v[13] = -37.797f * r - 74.203f * g + 112.0f * b + 128.f;
        v[14] = 112.0f * r - 93.786f * g - 18.214f * b + 128.f;
        {   // HED stain separation: log(max(rgb,1e-6))/log(1e-6) @ inv(rgb_from_hed), clipped at 0
            const float dr = lut_od[Ri], dg = lut_od[Gi], db = lut_od[Bi];
            v[15] = fmaxf(1.8779827369f * dr - 0.0659080622f * dg - 0.6019073634f * db, 0.f);
            v[16] = fmaxf(-1.0076786863f * dr + 1.1347303725f * dg - 0.4804141885f * db, 0.f);
            v[17] = fmaxf(-0.5561158182f * dr - 0.1355217986f * dg + 1.573588072f * db, 0.f);
        }
        v[18] = 0.299f * r + 0.587f * g + 0.114f * b;                              // YIQ
        v[19] = 0.59590059f * r - 0.27455667f * g - 0.32134392f * b;
        v[20] = 0.21153661f * r - 0.52273617f * g + 0.31119955f * b;
#pragma unroll
        for (int j = 0; j < 21; ++j) v[j] = (v[j] - 0.5f) * 2.f;
        v[21] = 0.f; v[22] = 0.f; v[23] = 0.f;
        T* dst = out + p * ld;
        if (ld >= 24) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float w8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w8[j] = v[8 * q + j];
                store8<T>(dst + 8 * q, w8);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 21; ++j) st<T>::stv(dst + j, v[j]);
        }
    }
}

}  // namespace

extern "C" int isa_image_ex(const uint8_t* rgb, const isa_tensor* out, void* stream) {
    if (!rgb || !out || !out->data || out->c != 21 || out->ld < 21 || out->n <= 0 || out->h <= 0 || out->w <= 0) return ISA_EINVAL;
    if (out->dtype != ISA_F32 && out->dtype != ISA_BF16) return ISA_EINVAL;
    if (out->ld >= 24 && (out->ld % 8 || reinterpret_cast<uintptr_t>(out->data) % 16)) return ISA_EALIGN;
    const long pixels = (long)out->n * out->h * out->w;
    const int grid = grid_cap(cdiv(pixels, 256), 256 * 16);
    if (out->dtype == ISA_BF16)
        hipLaunchKernelGGL(image_ex_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), rgb, pixels, (bf16_t*)out->data, out->ld);
    else
        hipLaunchKernelGGL(image_ex_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), rgb, pixels, (float*)out->data, out->ld);
    return launch_status();
}
