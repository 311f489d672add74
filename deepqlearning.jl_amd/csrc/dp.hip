// dp.hip -- data-parallel exchange without an all-reduce of the big weight gradients.
// For a wide dense layer the weight gradient is dW = X^T dpre summed over the global batch.  Instead of all-reducing dW
// ((K+1)*N floats; 2 x 6.4 MB for the two 3136x512 layers) every rank ALL-GATHERS its operands -- X (K x B) and dpre (N x B),
// 0.53 MB -- plus its small gradients (conv layers, heads), and then computes the big dW over the world*B gathered samples
// locally with the ordinary dW kernel (rank-major sample order: exactly the contraction a single GPU would run on the
// concatenated batch) and sums the small gradients over ranks in ascending order.  One collective of 0.86 MB per rank
// instead of 13.2 MB through a ring, and every replica performs bit-identical arithmetic.
#include "common.h"

// send[dst + i] = src[(i / B) * ld + (i % B)]     (B = ld = 1: plain copy; else rows of B samples out of a [.][ld] arena)
__global__ void k_dp_pack(DpPackArgs A) {
    const DpRegion& R = A.r[blockIdx.y];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < R.n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long row = R.B > 1 ? i / (unsigned)R.B : i; const unsigned col = R.B > 1 ? (unsigned)(i - row * (unsigned)R.B) : 0u;
        float v;
        if (R.S > 1) {          // a conv layer's split-K dW slabs: reduced here (ascending) instead of by a launch of their own
            v = slab_sum(R.src + i, (size_t)R.per_s, R.S);
        } else v = R.B > 1 ? R.src[row * (unsigned)R.ld + col] : R.src[i];
        A.send[R.dst + i] = v;
    }
}
void launch_dp_pack(hipStream_t st, const DpPackArgs& a) {
    unsigned long long mx = 1; for (int i = 0; i < a.n; i++) mx = a.r[i].n > mx ? a.r[i].n : mx;
    unsigned bx = (unsigned)((mx + 255) / 256); if (bx > 512) bx = 512;
    hipLaunchKernelGGL(k_dp_pack, dim3(bx, a.n), dim3(256), 0, st, a);
}
// grad[dst + i] = sum over ranks r (ascending) of recv[r * stride + src + i]
__global__ void k_dp_unpack_sum(DpSumArgs A) {
    const DpRange& R = A.r[blockIdx.y];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < R.n; i += (unsigned long long)gridDim.x * blockDim.x) {
        float tot = A.recv[R.src + i];
        for (int r = 1; r < A.world; r++) tot = tot + A.recv[(unsigned long long)r * A.stride + R.src + i];
        A.grad[R.dst + i] = tot;
    }
}
void launch_dp_unpack_sum(hipStream_t st, const DpSumArgs& a) {
    unsigned long long mx = 1; for (int i = 0; i < a.n; i++) mx = a.r[i].n > mx ? a.r[i].n : mx;
    unsigned bx = (unsigned)((mx + 255) / 256); if (bx > 512) bx = 512;
    hipLaunchKernelGGL(k_dp_unpack_sum, dim3(bx, a.n), dim3(256), 0, st, a);
}
