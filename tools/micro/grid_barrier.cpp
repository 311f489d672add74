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
// r05: the XCD-HIERARCHICAL form of the MI355X guide ("barrier-xcd": 4.1 us at 256 workgroups there vs 7.4 for one flat counter): arrivals go to the
// counter of the workgroup's own XCD (s_getreg XCC_ID: 8 counters, 32 arrivers each at 256 workgroups); the LAST arriver of an XCD does the release fence, arrives on the top
// counter and waits for all 8 XCDs there, does the acquire and bumps its XCD's generation word; everybody else polls that word (relaxed) and does ONE acquire fence.
// ws: [0..7] per-XCD arrival counters, [8] top counter, [16..23] per-XCD generation words, one 64-byte line apart in the real layout (x16 below)
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }
__device__ __forceinline__ void grid_barrier_xcd(unsigned* ws, const unsigned* per_xcd /* workgroups resident on each XCD, counted by a census launch */, unsigned nxcd_active, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        unsigned* cnt = ws + 16 * x; unsigned* top = ws + 16 * 8; unsigned* gw = ws + 16 * (16 + x);
        const unsigned target = gen + 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // this workgroup's writes leave its L2 before it arrives
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned a = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == target * per_xcd[x]) {                     // the XCD's last arriver: up to the top level
            __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int sp = 0; sp < 2000000 && __hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target * nxcd_active; sp++) __builtin_amdgcn_s_sleep(1);      // bounded: a placement that differs from the census must not hang the GPU
            __hip_atomic_store(gw, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (int sp = 0; sp < 2000000 && __hip_atomic_load(gw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; sp++) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    gen++;
}
__global__ void k_census(unsigned* per_xcd) { if (threadIdx.x == 0) atomicAdd(per_xcd + xcc_id(), 1u); }
__global__ void k_phases_xcd(float* buf, unsigned* ws, const unsigned* per_xcd, unsigned nxcd_active, int nphase, int* bad) {
    unsigned gen = 0; const unsigned nb = gridDim.x;
    for (int p = 0; p < nphase; p++) {
        buf[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)(p * 1000 + blockIdx.x);
        grid_barrier_xcd(ws, per_xcd, nxcd_active, gen);
        const unsigned other = (blockIdx.x + nb / 2 + 1) % nb;
        const float v = buf[(size_t)other * 256 + threadIdx.x];
        if (v != (float)(p * 1000 + other)) atomicAdd(bad, 1);
        grid_barrier_xcd(ws, per_xcd, nxcd_active, gen);
    }
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
        if (nb <= 256) {      // XCD-hierarchical form (one workgroup per CU at most: every workgroup resident; the census counts who sits where for THIS grid size)
            unsigned *ws, *px; hipMalloc(&ws, 16 * 32 * 4); hipMalloc(&px, 8 * 4);
            hipMemset(px, 0, 32); hipLaunchKernelGGL(k_census, dim3(nb), dim3(256), 0, st, px); hipStreamSynchronize(st);
            unsigned hpx[8]; hipMemcpy(hpx, px, 32, hipMemcpyDeviceToHost); unsigned act = 0, tot = 0; for (int x = 0; x < 8; x++) { act += hpx[x] > 0; tot += hpx[x]; }
            hipMemset(bad, 0, 4); double bx = 1e30;
            for (int r = 0; r < reps; r++) {
                hipMemsetAsync(ws, 0, 16 * 32 * 4, st); hipStreamSynchronize(st);
                auto t0 = std::chrono::high_resolution_clock::now();
                hipLaunchKernelGGL(k_phases_xcd, dim3(nb), dim3(256), 0, st, buf, ws, px, act, nphase, bad);
                hipStreamSynchronize(st);
                const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
                if (us < bx) bx = us;
            }
            hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("                 %.2f us per XCD-hierarchical barrier (census: %u workgroups on %u XCDs; placement is the same for equal grids), stale reads: %d\n", bx / (2.0 * nphase), tot, act, hb);
            hipFree(ws); hipFree(px);
        }
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
