// mfma_lds_ceiling.cpp -- what the forward tile's inner loop can reach with NOTHING but LDS fragment reads and MFMAs (MI355X, fp32 16x16x4; peak 157.3 TFLOP/s):
// a 256-thread workgroup = 4 waves, wave w owns 16 columns x (16*NT) channels; per K tile of 32: 8 k-steps x (1 A read + NT B reads, ds_read_b32) + 8*NT MFMAs.
// Variants: regs = operands from registers (pure MFMA issue), lds = fragment reads as in k_fwd_lds, lds+bar = plus the two barriers per K tile.  Occupancy via grid.
// r04: calibrates how much of the config-5 gap (48 % whole-step) is the tile shape itself.     hipcc --offload-arch=gfx950 -O3 mfma_lds_ceiling.cpp -o mfma_lds_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
template <int NT, int MODE>   // MODE 0 regs, 1 lds, 2 lds + barriers
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    constexpr int SA = 80, SB = (16 * NT) % 32 == 0 ? 16 * NT + 16 : 16 * NT + 32;
    extern __shared__ float lds[];
    float* As = lds; float* Bs = lds + 32 * SA;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 32 * SA + 32 * SB; i += 256) lds[i] = 1.0f + 1e-6f * (float)i;
    __syncthreads();
    f32x4 acc[NT];
    for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float af = 1.0f + lane * 1e-6f, bf0 = 1.0f - lane * 1e-6f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int st = 0; st < 8; st++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = MFMA(af, bf0, acc[t]);
        } else {
            const float* Ab = As + kq * SA + 16 * wave + l15; const float* Bb = Bs + kq * SB + l15;
            float a[8], b[8][NT];
#pragma unroll
            for (int st = 0; st < 8; st++) { a[st] = Ab[4 * st * SA];
#pragma unroll
                for (int t = 0; t < NT; t++) b[st][t] = Bb[4 * st * SB + 16 * t]; }
#pragma unroll
            for (int st = 0; st < 8; st++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = MFMA(a[st], b[st][t], acc[t]);
            if (MODE == 2) { __syncthreads(); if (tid == 0 && it == iters + 5) lds[0] = 2.0f; __syncthreads(); }
            asm volatile("" ::: "memory");
        }
    }
    float s = 0; for (int t = 0; t < NT; t++) s += acc[t].x + acc[t].y + acc[t].z + acc[t].w;
    if (s == 12345.678f) out[tid] = s;
}
template <int NT, int MODE> void run(const char* name, float* out) {
    const int iters = 4000; const size_t lds = (32 * 80 + 32 * 96) * 4 * 2;      // two buffers' worth, like the real kernel (bounds the residency the same way)
    for (int wg_per_cu : {1, 2, 3, 4, 6, 8}) {
        const int grid = 256 * wg_per_cu;
        hipLaunchKernelGGL((k<NT, MODE>), dim3(grid), dim3(256), lds, 0, out, 10); hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0); hipLaunchKernelGGL((k<NT, MODE>), dim3(grid), dim3(256), lds, 0, out, iters); hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double flop = (double)grid * 4 /*waves*/ * iters * 8 * NT * 2048.0;
        printf("%-10s NT=%d  %d workgroups/CU: %7.1f TFLOP/s (%.0f %% of 157.3)\n", name, NT, wg_per_cu, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3 * 100);
    }
}
int main() {
    float* out; hipMalloc(&out, 4096);
    run<4, 0>("regs", out); run<4, 1>("lds", out); run<4, 2>("lds+bar", out);
    run<2, 1>("lds", out); run<2, 2>("lds+bar", out); run<1, 1>("lds", out);
    return 0;
}
