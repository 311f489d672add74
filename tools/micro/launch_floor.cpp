// micro-benchmark: per-dispatch floor of dependent kernels replayed from a hipGraph on one stream (MI355X)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_empty() {}
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] + 1.0f; }
struct Big { int a[40]; };
__global__ void k_bigarg(Big b, float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = (float)b.a[3]; }
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
    Big b; for (int i = 0; i < 40; i++) b.a[i] = i;
    printf("graph of 20 kernels, 200 replays: us per kernel\n");
    printf("empty <<<1,64>>>            %.2f\n", run(st, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }, 20, 200));
    printf("empty <<<1024,256>>>        %.2f\n", run(st, [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st); }, 20, 200));
    printf("empty <<<1024,256>>> 40KB LDS %.2f\n", run(st, [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 40960, st); }, 20, 200));
    printf("bigarg <<<1,64>>>           %.2f\n", run(st, [&] { hipLaunchKernelGGL(k_bigarg, dim3(1), dim3(64), 0, st, b, p); }, 20, 200));
    printf("touch 1M floats (8 MB rw)   %.2f\n", run(st, [&] { hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, st, p, 1 << 20); }, 20, 200));
    printf("touch 16M floats (128MB rw) %.2f\n", run(st, [&] { hipLaunchKernelGGL(k_touch, dim3(65536), dim3(256), 0, st, p, 16 << 20); }, 20, 200));
    printf("single graph launch of 1 empty kernel: %.2f us\n", run(st, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }, 1, 500));
    return 0;
}
