// FusedInst8.hip -- explicit instantiations of launchFusedT (FusedKernelsImpl.h): one of the translation units the fused RHS
// is compiled in.
#include "FusedKernelsImpl.h"

namespace OMEGA {
OMEGA_FUSED_INSTANCES_8(OMEGA_FUSED_DEFINE)
} // namespace OMEGA
