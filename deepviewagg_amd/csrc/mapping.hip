// placeholder until the mapping-build kernels land (same commit series)
#include "dva_common.h"
extern "C" {
int64_t dva_visibility_workspace_bytes(const dva_camera*, int64_t) { return DVA_ERR_UNSUPPORTED; }
int dva_visibility(const float*, int64_t, const dva_camera*, const uint8_t*, int64_t*, int64_t*,
                   int64_t*, float*, double*, double*, int64_t*, void*, int64_t, void*) {
  return DVA_ERR_UNSUPPORTED;
}
int dva_mapping_features(const float*, const int64_t*, const float*, const double*, const float*,
                         const float*, const float*, const float*, const dva_camera*, int64_t, float*,
                         int32_t*, void*) {
  return DVA_ERR_UNSUPPORTED;
}
}
