// launch_cost.cpp -- what an EMPTY dependent launch costs inside a hipGraph as a function of grid size, dynamic LDS, register budget and kernel-argument
// size (MI355X).  r03: the fused backward launches of the train step cost 4.7 us with every workgroup returning at once (DQN_PROBE=11).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
struct Arg { int a[340]; };   // 1360 bytes, the size of k_dwdx_lds's argument block
__global__ void k_small() {}
__global__ void k_arg(Arg a, float* p) { if (a.a[3] == 12345) p[0] = 1.0f; }
__global__ __launch_bounds__(256) void k_regs(float* p, int n) {   // ~128 VGPRs: a long dependent chain that is never executed (n == 0)
    float v[96];
    if (n > 0) { for (int i = 0; i < 96; i++) v[i] = p[threadIdx.x + 256 * i]; for (int j = 0; j < n; j++) for (int i = 0; i < 96; i++) v[i] = v[i] * v[(i + 1) % 96] + 1.0f; float s = 0; for (int i = 0; i < 96; i++) s += v[i]; p[threadIdx.x] = s; }
}
template <class F> double run(hipStream_t st, F enq, int nk, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nk; i++) enq();
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; i++) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < reps; i++) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return us / reps / nk;
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    float* p; hipMalloc(&p, 64 << 20); hipMemset(p, 0, 64 << 20);
    Arg a; for (int i = 0; i < 340; i++) a.a[i] = i;
    printf("graph of 20 dependent EMPTY kernels, 200 replays: us per kernel\n");
    for (int lds : {0, 17408, 34816}) for (int grid : {32, 256, 512, 1024, 1536, 2048, 4096})
        printf("no args   grid %5d x 256 thr, LDS %5d: %.2f\n", grid, lds, run(st, [&] { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), lds, st); }, 20, 200));
    for (int grid : {32, 1024, 2048})
        printf("1360-B args grid %5d x 256 thr, LDS 34816: %.2f\n", grid, run(st, [&] { hipLaunchKernelGGL(k_arg, dim3(grid), dim3(256), 34816, st, a, p); }, 20, 200));
    for (int grid : {32, 1024, 2048})
        printf("128-VGPR kernel grid %5d x 256 thr, LDS 34816: %.2f\n", grid, run(st, [&] { hipLaunchKernelGGL(k_regs, dim3(grid), dim3(256), 34816, st, p, 0); }, 20, 200));
    for (int thr : {64, 128, 512, 1024})
        printf("no args   grid %5d x %4d thr, LDS 34816: %.2f\n", 1024 * 256 / thr, thr, run(st, [&] { hipLaunchKernelGGL(k_small, dim3(1024 * 256 / thr), dim3(thr), 34816, st); }, 20, 200));
    return 0;
}
