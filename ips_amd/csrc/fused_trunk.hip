// fused_trunk.hip - LDS-resident fused trunk for 1x32x32 patches (placeholder hook:
// returns "unsupported" until the fused kernel lands; trunk.hip then uses the
// layer-by-layer kernels of conv.hip).
#include "ipsx_common.h"

namespace ipsx {
bool fused_trunk_supported(const ipsx_trunk*) { return false; }
int fused_trunk_encode(const ipsx_trunk*, const float*, int64_t, float*, hipStream_t) {
    return fail(IPSX_EINVAL, "fused trunk not available");
}
}  // namespace ipsx
