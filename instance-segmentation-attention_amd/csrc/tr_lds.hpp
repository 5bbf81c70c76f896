// LDS transpose-read fragments for bf16 MFMA operands whose contraction index is the pixel.
// Tiles are staged row-major [pixel][channel]; ds_read_b64_tr_b16 returns, for lane (r, h), element j
// <- tile[8h+j][r] (verified by scripts/probes/tr_probe.hip).  Row stride == 64 (mod 256) bytes keeps the
// four 64-byte row pieces a half-wave touches on disjoint banks.
#pragma once
#include "common.hpp"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_S16X4(ptr) ((__attribute__((address_space(3))) s16x4*)(ptr))

template <int CH> struct TrStride { static constexpr int bytes = (CH * 2) % 128 == 0 ? CH * 2 + 64 : CH * 2; };

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int stride_bytes, int pix0, int chan0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const char* a = tile + (pix0 + 8 * (g >> 1) + q) * stride_bytes + (chan0 + 16 * (g & 1) + 4 * pp) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(a + 4 * stride_bytes));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
