// Tail of AlignCollate.__call__ (code/lib/dataset.py:349-379) on the device: the collate function ends by widening the
// uint8 instance planes [bs,h,w,K] to int64, permuting them to [bs,K,h,w], and turning the {0,1} semantic map into an
// int64 one-hot [bs,2,h,w].  Done on the host that is 17.8 bytes/pixel/plane-set of PCIe traffic per step (285 MB at
// bs=16, 256x256, K=32); handing over the uint8 arrays and expanding here moves 33 bytes per pixel over PCIe and
// writes the int64 targets at HBM speed.  Integer work, bit-exact against oracle/collate_ref.py.
//
// Layout: a workgroup takes 256 consecutive pixels of one image: the [256][K] byte tile is read with 16-byte loads
// into LDS (row stride K+4 bytes: the 4-byte column reads of the transposed walk spread over the banks), then for each
// plane k the 256 lanes write 256 consecutive int64 (2 KB contiguous per store instruction).
#include "common.hpp"

namespace {

constexpr int TP = 256;          // pixels per tile

__global__ __launch_bounds__(256) void collate_targets_kernel(const uint8_t* ins, const uint8_t* sem, int n, long hw, int K,
                                                              int64_t* ins_out, int64_t* sem_out) {
    extern __shared__ uint8_t tile[];                 // [TP][K + 4]
    const int stride = K + 4;
    const long tiles_per_img = (hw + TP - 1) / TP;
    for (long t = blockIdx.x; t < (long)n * tiles_per_img; t += gridDim.x) {
        const long b = t / tiles_per_img, p0 = (t % tiles_per_img) * TP;
        const int np = (int)min((long)TP, hw - p0);
        const uint8_t* src = ins + (b * hw + p0) * K;
        const long bytes = (long)np * K;
        __syncthreads();                               // the previous tile has been consumed
        if ((K & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            for (long i = threadIdx.x * 16L; i < bytes; i += 256 * 16L) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + i);
                const int p = (int)(i / K), k = (int)(i % K);
                uint32_t* d = reinterpret_cast<uint32_t*>(tile + p * stride + k);   // stride and k are multiples of 4
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (long i = threadIdx.x; i < bytes; i += 256) tile[(i / K) * stride + (i % K)] = src[i];
        }
        __syncthreads();
        const int p = threadIdx.x;
        if (p < np) {
            int64_t* dst = ins_out + (b * K) * hw + p0 + p;
            for (int k = 0; k < K; ++k) dst[(long)k * hw] = (int64_t)tile[p * stride + k];
            if (sem) {
                const uint8_t v = sem[b * hw + p0 + p];
                sem_out[(b * 2) * hw + p0 + p] = v == 0;          // np.eye(2)[v]: channel c = (v == c)
                sem_out[(b * 2 + 1) * hw + p0 + p] = v == 1;
            }
        }
    }
}

}  // namespace

extern "C" int isa_collate_targets(const uint8_t* ins, const uint8_t* sem, int32_t n, int32_t h, int32_t w, int32_t k,
                                   int64_t* ins_out, int64_t* sem_out, void* stream) {
    if (!ins || !ins_out || n <= 0 || h <= 0 || w <= 0 || k <= 0 || k > 252 || (sem && !sem_out)) return ISA_EINVAL;
    const long hw = (long)h * w;
    const long tiles = (long)n * ((hw + TP - 1) / TP);
    const int grid = grid_cap(tiles, 256 * 8);
    hipLaunchKernelGGL(collate_targets_kernel, dim3(grid), dim3(256), (size_t)TP * (k + 4), as_stream(stream),
                       ins, sem, n, hw, k, ins_out, sem_out);
    return launch_status();
}
