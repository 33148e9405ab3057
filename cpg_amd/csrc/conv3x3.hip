// placeholder until the specialised 3x3 s1 p1 kernels land: everything routes to the generic path
#include "cpg_common.h"
extern "C" int cpg_conv3x3_supported(const cpg_conv_desc *d) { (void)d; return 0; }
int cpg_conv3x3_fwd(const cpg_conv_desc *, const float *, const float *, const float *, float, const float *, float *, hipStream_t) { return CPG_E_UNSUPPORTED; }
int cpg_conv3x3_dgrad(const cpg_conv_desc *, const float *, const float *, const float *, float, float *, hipStream_t) { return CPG_E_UNSUPPORTED; }
