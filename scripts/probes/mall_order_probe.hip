// Does the ORDER in which a kernel walks a tensor that the previous kernel just wrote matter?  Backward kernels re-read
// 130-540 MB tensors their predecessor produced; the Infinity Cache is 256 MB.  If it behaves like an LRU, a consumer that
// walks the tensor in the producer's order finds the head evicted by the tail (no hits once the tensor exceeds the cache),
// while one that walks it BACKWARDS starts on the most recently written bytes.
//   W: 1024 persistent workgroups write the buffer in increasing 64 KB chunks;
//   R fwd / R rev: the same grid reads it in increasing / decreasing chunk order, timed right after W (hipEvents).
// build: hipcc --offload-arch=gfx950 -O3 -w scripts/probes/mall_order_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr long CHUNK = 65536;

__global__ __launch_bounds__(256) void write_kernel(char* buf, long nchunks, float v) {
    const f32x4 x = {v, v, v, v};
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x)
        for (int o = threadIdx.x * 16; o < CHUNK; o += 4096) *reinterpret_cast<f32x4*>(buf + c * CHUNK + o) = x;
}
__global__ __launch_bounds__(256) void read_kernel(const char* buf, long nchunks, int rev, float* sink) {
    f32x4 acc = {0, 0, 0, 0};
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long cc = rev ? nchunks - 1 - c : c;
        for (int o = threadIdx.x * 16; o < CHUNK; o += 4096) acc += *reinterpret_cast<const f32x4*>(buf + cc * CHUNK + o);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) *sink = 1.f;
}

int main() {
    float* sink; hipMalloc(&sink, 4);
    char* big; hipMalloc(&big, 1L << 30);                       // flush buffer
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long sizes_mb[] = {64, 128, 192, 256, 320, 384, 512, 768};
    printf("%8s %12s %12s %12s %12s\n", "MB", "R fwd us", "R rev us", "fwd TB/s", "rev TB/s");
    for (long mb : sizes_mb) {
        const long bytes = mb << 20, nchunks = bytes / CHUNK;
        char* buf; hipMalloc(&buf, bytes);
        float t[2] = {0, 0};
        for (int rev = 0; rev < 2; ++rev) {
            float best = 1e9f, sum = 0.f; int reps = 8;
            for (int r = 0; r < reps + 2; ++r) {
                hipLaunchKernelGGL(write_kernel, dim3(1024), dim3(256), 0, 0, big, (1L << 30) / CHUNK, 0.f);   // evict everything
                hipLaunchKernelGGL(write_kernel, dim3(1024), dim3(256), 0, 0, buf, nchunks, 1.f);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(256), 0, 0, buf, nchunks, rev, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r >= 2) { sum += ms; if (ms < best) best = ms; }
            }
            t[rev] = sum / reps * 1e3f;
        }
        printf("%8ld %12.1f %12.1f %12.2f %12.2f\n", mb, t[0], t[1], bytes / t[0] / 1e6, bytes / t[1] / 1e6);
        hipFree(buf);
    }
    return 0;
}
