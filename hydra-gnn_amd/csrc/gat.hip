// K3 -- GAT edge softmax + weighted aggregation (implemented in a later milestone of this round).
#include "kernels.h"
extern "C" int hmp_gat_fwd(const float*, int32_t, const float*, const float*, const float*, const float*, hmp_plan, hmp_gat_args,
                           float*, float*, int32_t, void*) {
  using namespace hmp;
  HMP_FAIL(HMP_E_STATE, "hmp_gat_fwd: not built in this library revision");
}
extern "C" int hmp_gat_bwd(const float*, int32_t, const float*, int32_t, const float*, const float*, const float*, const float*,
                           const float*, hmp_plan, hmp_gat_args, float*, float*, int32_t, float*, float*, float*, float*, void*) {
  using namespace hmp;
  HMP_FAIL(HMP_E_STATE, "hmp_gat_bwd: not built in this library revision");
}
