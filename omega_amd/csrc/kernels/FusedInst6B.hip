// FusedInst6B.hip -- explicit instantiations of launchFusedT (FusedKernelsImpl.h): one of the translation units the fused RHS
// is compiled in.
#include "FusedKernelsImpl.h"

namespace OMEGA {
OMEGA_FUSED_INSTANCES_6B(OMEGA_FUSED_DEFINE)
OMEGA_FUSED_INSTANCES_6N(OMEGA_FUSED_DEFINE)
} // namespace OMEGA
