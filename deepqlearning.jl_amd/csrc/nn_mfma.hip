// nn_mfma.hip -- fp32 MFMA (v_mfma_f32_16x16x4_f32) kernels for the large contractions of batch_train!.
// (stub: filled in once the VALU path is parity-green on the GPU)
#include "common.h"
bool launch_mfma_fwd(hipStream_t, const LayerDev&, const float*, const float*, int, int, int, float*, float*) { return false; }
bool launch_mfma_dw(hipStream_t, const LayerDev&, const float*, int, const float*, int, float*, float*) { return false; }
bool launch_mfma_dx(hipStream_t, const LayerDev&, const float*, const float*, int, float*, float*, const float*, const float*, int, int) { return false; }
