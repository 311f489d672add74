// micro-benchmark: cost of a device-wide barrier inside ONE kernel (atomic counter + agent-scope fences) against a kernel boundary, with a
// producer/consumer data hand-off across the barrier (every workgroup writes a line, then reads a line written by another XCD's workgroup).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned nblocks, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                        // release: make this workgroup's writes visible device-wide
        atomicAdd(ctr, 1u);
        const unsigned target = (gen + 1) * nblocks;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();                                        // acquire
    }
    __syncthreads();
    gen++;
}
__global__ void k_phases(float* buf, unsigned* ctr, int nphase, int* bad) {
    unsigned gen = 0; const unsigned nb = gridDim.x;
    for (int p = 0; p < nphase; p++) {
        buf[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)(p * 1000 + blockIdx.x);
        grid_barrier(ctr, nb, gen);
        const unsigned other = (blockIdx.x + nb / 2 + 1) % nb;    // a block that (very likely) ran on another XCD
        const float v = buf[(size_t)other * 256 + threadIdx.x];
        if (v != (float)(p * 1000 + other)) atomicAdd(bad, 1);
        grid_barrier(ctr, nb, gen);                             // nobody overwrites before everybody has read
    }
}
__global__ void k_write(float* buf, int p) { buf[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)(p * 1000 + blockIdx.x); }
__global__ void k_read(const float* buf, int p, int* bad) { const unsigned nb = gridDim.x, other = (blockIdx.x + nb / 2 + 1) % nb; if (buf[(size_t)other * 256 + threadIdx.x] != (float)(p * 1000 + other)) atomicAdd(bad, 1); }
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float* buf; unsigned* ctr; int* bad; hipMalloc(&buf, 1024 * 256 * 4); hipMalloc(&ctr, 4); hipMalloc(&bad, 4);
    for (int nb : {64, 140, 256, 512}) {
        const int nphase = 50, reps = 20;
        hipMemset(bad, 0, 4);
        double best = 1e30;
        for (int r = 0; r < reps; r++) {
            hipMemsetAsync(ctr, 0, 4, st); hipStreamSynchronize(st);
            auto t0 = std::chrono::high_resolution_clock::now();
            hipLaunchKernelGGL(k_phases, dim3(nb), dim3(256), 0, st, buf, ctr, nphase, bad);
            hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
            if (us < best) best = us;
        }
        int hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("%4d workgroups: %.2f us per grid barrier (incl. the write/read between), stale reads: %d\n", nb, best / (2.0 * nphase), hb);
        // the same hand-off with kernel boundaries, replayed from a graph
        hipGraph_t g; hipGraphExec_t ge; hipMemset(bad, 0, 4);
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int p = 0; p < nphase; p++) { hipLaunchKernelGGL(k_write, dim3(nb), dim3(256), 0, st, buf, p); hipLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, st, buf, p, bad); }
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int r = 0; r < reps; r++) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("                 %.2f us per kernel boundary (graph of %d tiny kernels), stale reads: %d\n", us / reps / (2.0 * nphase), 2 * nphase, hb);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
