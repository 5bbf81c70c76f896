// What does a grid-wide barrier cost on MI355X next to a kernel boundary?  (Input to the block-persistent design of
// a whole InvertedResidual block as ONE kernel whose phases are separated by statistics-only grid barriers.)
//   * barrier: every workgroup adds 1 to a device counter (agent-scope atomic, executed at the memory side) and one lane
//     polls it with sc1 loads until all have arrived; bounded spin (a stuck barrier sets an error flag and every wave
//     leaves: the grid always drains);
//   * the grid is one or two workgroups per CU, so all workgroups are co-resident;
//   * compared with the same number of dependent empty kernels (launch + drain + the L2 write-back of a kernel boundary).
// build: hipcc --offload-arch=gfx950 -O3 scripts/probes/grid_barrier_probe.hip -o gpurun_out/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void barrier_loop(unsigned* counter, int rounds, unsigned* err, float* sink, const float* src, int touch) {
    const unsigned nwg = gridDim.x;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        if (touch) acc += src[((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)r * nwg * 256) % (1u << 24)];   // some memory traffic per phase
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * nwg;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 22)) { *err = 1; break; }          // never hang: leave and report
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (touch && acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void empty_kernel(float* sink, const float* src, int touch, int r) {
    if (touch) { const float v = src[((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)r * gridDim.x * 256) % (1u << 24)]; if (v == 12345.678f) sink[0] = v; }
}

int main() {
    unsigned *counter, *err; float *sink, *src;
    hipMalloc(&counter, 4); hipMalloc(&err, 4); hipMalloc(&sink, 4); hipMalloc(&src, (size_t)(1u << 24) * 4);
    hipMemset(src, 0, (size_t)(1u << 24) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 200;
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
        for (int touch = 0; touch <= 1; ++touch) {
            const int grid = 256 * wg_per_cu;
            float best_b = 1e9f, best_k = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
                hipEventRecord(e0);
                hipLaunchKernelGGL(barrier_loop, dim3(grid), dim3(256), 0, 0, counter, rounds, err, sink, src, touch);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best_b) best_b = ms;
                hipEventRecord(e0);
                for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, sink, src, touch, r);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1); if (ms < best_k) best_k = ms;
            }
            unsigned h_err = 0; hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost);
            printf("grid %4d workgroups, %s: grid barrier %.2f us each (one kernel, %d barriers)%s;  dependent kernel launches %.2f us each (eager stream)\n",
                   grid, touch ? "one 4-byte load per thread and phase" : "no memory traffic", best_b * 1e3f / rounds, rounds,
                   h_err ? "  [BARRIER TIMED OUT]" : "", best_k * 1e3f / rounds);
        }
    // the same launches replayed from a hipGraph (what the training step does)
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, sink, src, 1, r);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipGraph replay of %d dependent 256-workgroup kernels (one load per thread): %.2f us each\n", rounds, ms * 1e3f / rounds);
    return 0;
}
