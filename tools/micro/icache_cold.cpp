// icache_cold.cpp -- what does COLD straight-line code cost on gfx950?  One workgroup per CU executes N dependent v_add_u32 (4-byte encodings, 16
// per 64-byte line) once, as straight-line code (cold instruction cache at kernel start) or as a 64-instruction loop (hot after one pass);
// s_memtime around the block, per-workgroup.  Build: hipcc --offload-arch=gfx950 -O3 icache_cold.cpp -o icache_cold
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define N_STRAIGHT 4096
__global__ void k_straight(unsigned long long* out, int dummy) {
    unsigned v = threadIdx.x + dummy;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(".rept 4096\n v_add_u32_e32 %0, %0, %0\n .endr" : "+v"(v));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = v; }
}
__global__ void k_loop(unsigned long long* out, int dummy) {
    unsigned v = threadIdx.x + dummy;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 4096 / 64; i++) asm volatile(".rept 64\n v_add_u32_e32 %0, %0, %0\n .endr" : "+v"(v));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = v; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 2 * 256 * 8);
    std::vector<unsigned long long> h(512);
    for (int wg : {256, 1}) for (int threads : {64, 256}) for (int rep = 0; rep < 3; rep++) {
        for (int which = 0; which < 2; which++) {
            if (which == 0) hipLaunchKernelGGL(k_straight, dim3(wg), dim3(threads), 0, 0, d, rep); else hipLaunchKernelGGL(k_loop, dim3(wg), dim3(threads), 0, 0, d, rep);
            hipDeviceSynchronize(); hipMemcpy(h.data(), d, 2 * wg * 8, hipMemcpyDeviceToHost);
            std::vector<double> t; for (int i = 0; i < wg; i++) t.push_back((double)h[2 * i]);
            std::sort(t.begin(), t.end());
            printf("%s wg=%3d threads=%3d rep %d: %d instr, median %.0f ticks (%.2f ticks/instr, %.1f ticks per 64-B line), min %.0f max %.0f  [s_memtime ticks; 100 MHz => 10 ns]\n",
                   which ? "loop    " : "straight", wg, threads, rep, N_STRAIGHT, t[t.size() / 2], t[t.size() / 2] / N_STRAIGHT, t[t.size() / 2] / (N_STRAIGHT / 16.0), t.front(), t.back());
        }
    }
    return 0;
}
