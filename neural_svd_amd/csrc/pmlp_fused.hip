// Fused MFMA path (placeholder until the gfx950 kernels land): reports "unsupported" so every shape
// takes the generic kernels.
#include "nsvd_kernels.h"

bool nsvd_fused_supported(const nsvd_model_desc&, int) { return false; }
size_t nsvd_fused_workspace_bytes(const nsvd_model_desc&, int) { return 0; }
int nsvd_fused_forward(const nsvd_model_desc&, const nsvd_params&, const nsvd_problem&, const float*, int, float*,
                       float*, void*, int, hipStream_t) { return NSVD_EUNSUPPORTED; }
int nsvd_fused_backward(const nsvd_model_desc&, const nsvd_params&, const nsvd_problem&, const float*, int,
                        const float*, const nsvd_params&, void*, hipStream_t) { return NSVD_EUNSUPPORTED; }
