// Does the ACCESS PATTERN of conv_gemm's streaming kernel cost bandwidth?  (its 256x256 launches stream at
// 3.6-3.9 TB/s where a copy reaches 5.2.)  Pure memory kernels, no MFMA, on buffers far larger than the Infinity Cache:
//   read  A: 16 bytes per lane, consecutive lanes consecutive addresses (1 KiB contiguous per wave instruction)
//   read  B: conv_gemm's A-fragment pattern - lane (r, hh) reads the two 16-byte halves of bytes [32 hh, 32 hh + 32) of pixel r's
//            row (row stride = K * 2 bytes): every wave instruction touches 32 rows and half of the bytes it pulls in
//   write A: 16 bytes per lane contiguous;  write B: conv_gemm's epilogue pattern - lane -> (row = lane >> 1, 16 channels at
//            (lane & 1) * 16), two 16-byte stores per lane, row stride N * 2 bytes
// and the combinations a 1x1 conv K -> N moves (read K channels, write N channels per pixel).
// build: hipcc --offload-arch=gfx950 -O3 -w scripts/probes/access_pattern_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid-stride over 32-pixel wave tiles; K, N in bf16 channels (row bytes = 2K / 2N)
template <int RMODE, int WMODE>
__global__ __launch_bounds__(256) void stream_kernel(const char* in, char* out, long pixels, int K, int N, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const long tiles = pixels / 128;
    f32x4 acc = {0, 0, 0, 0};
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long p0 = t * 128 + wave * 32;                       // first pixel of this wave's 32-pixel tile
        if (RMODE == 0) {                                          // contiguous: the tile is 32 * 2K bytes
            const char* base = in + p0 * 2 * K;
            for (int o = lane * 16; o < 64 * K; o += 1024) { const f32x4 v = *reinterpret_cast<const f32x4*>(base + o); acc += v; }
        } else if (RMODE == 1) {                                   // row per lane, 32 bytes per K32 group
            const char* row = in + (p0 + r) * 2 * K + 32 * hh;
            for (int g = 0; g < K / 32; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(row + g * 64), b = *reinterpret_cast<const f32x4*>(row + g * 64 + 16);
                acc += a; acc += b;
            }
        }
        if (WMODE == 0) {
            char* base = out + p0 * 2 * N;
            for (int o = lane * 16; o < 64 * N; o += 1024) *reinterpret_cast<f32x4*>(base + o) = acc;
        } else if (WMODE == 1) {
            const int row = lane >> 1, cseg = (lane & 1) * 32;    // 16 channels = 32 bytes per lane, as two 16-byte stores
            for (int n0 = 0; n0 < 2 * N; n0 += 64) {
                char* dst = out + (p0 + row) * 2 * N + n0 + cseg;
                *reinterpret_cast<f32x4*>(dst) = acc; *reinterpret_cast<f32x4*>(dst + 16) = acc;
            }
        }
    }
    if (acc[0] == 12345.f) sink[0] = acc[1];
}

template <int RMODE, int WMODE>
float run(const char* in, char* out, long pixels, int K, int N, float* sink, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<RMODE, WMODE>), dim3(grid), dim3(256), 0, 0, in, out, pixels, K, N, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const long pixels = 8L << 20;                                  // 8 M pixels: K = 64 -> 1 GiB read
    char *in, *out; float* sink;
    hipMalloc(&in, pixels * 2 * 64); hipMalloc(&out, pixels * 2 * 64); hipMalloc(&sink, 4);
    hipMemset(in, 0, pixels * 2 * 64);
    const int grids[2] = {768, 2048};
    for (int gi = 0; gi < 2; ++gi) {
        const int grid = grids[gi];
        printf("grid %d workgroups x 256 threads, 8 M pixels\n", grid);
        for (int K = 32; K <= 64; K += 32) {
            const double rb = (double)pixels * 2 * K;
            float a = run<0, 2>(in, out, pixels, K, 32, sink, grid), b = run<1, 2>(in, out, pixels, K, 32, sink, grid);
            printf("  read only, K = %2d:  contiguous %.0f GB/s   row-per-lane %.0f GB/s\n", K, rb / a / 1e6, rb / b / 1e6);
        }
        for (int N = 32; N <= 64; N += 32) {
            const double wb = (double)pixels * 2 * N;
            float a = run<2, 0>(in, out, pixels, 32, N, sink, grid), b = run<2, 1>(in, out, pixels, 32, N, sink, grid);
            printf("  write only, N = %2d: contiguous %.0f GB/s   row-pair-per-lane %.0f GB/s\n", N, wb / a / 1e6, wb / b / 1e6);
        }
        const int KN[3][2] = {{64, 32}, {32, 64}, {64, 64}};
        for (int i = 0; i < 3; ++i) {
            const int K = KN[i][0], N = KN[i][1];
            const double tb = (double)pixels * 2 * (K + N);
            float a = run<0, 0>(in, out, pixels, K, N, sink, grid), b = run<1, 1>(in, out, pixels, K, N, sink, grid);
            float c = run<1, 0>(in, out, pixels, K, N, sink, grid), d = run<0, 1>(in, out, pixels, K, N, sink, grid);
            printf("  read %2d + write %2d:  contiguous/contiguous %.0f   conv_gemm patterns %.0f   (row read + contiguous write %.0f, contiguous read + row-pair write %.0f) GB/s\n",
                   K, N, tb / a / 1e6, tb / b / 1e6, tb / c / 1e6, tb / d / 1e6);
        }
    }
    return 0;
}
