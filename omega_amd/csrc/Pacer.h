// Pacer.h -- the reference's timer names as roctx ranges.
//
// The reference brackets its launches with Pacer::start / Pacer::stop (share/pacer/Pacer.cpp:153-200, GPTL timers
// written to omega.timing), e.g. "Tend:computeAllTendencies", "AuxState:vertexAuxState1", "RK4:haloExch"
// (components/omega/src/ocn/Tendencies.cpp:270-588, AuxiliaryState.cpp:76-175, timeStepping/*.cpp).  Here the same
// names open and close roctx ranges, so `rocprofv3 --marker-trace --kernel-trace` lines the kernels of this build up
// under the timer names of an Omega run.  Without a profiler attached a range costs two calls into the roctx stub.
// The fused kernels cover several reference timers at once; their ranges are named "Tend:fused:<level>" and list
// the reference timers they replace.  The ranges bracket the ENQUEUE of asynchronous launches on the host: which
// kernels ran under a range is what the profiler's correlation ids say, not the range's own duration.  start / stop
// pairs of one level keep a per-thread depth, so that changing the timing level between a start and its stop cannot
// unbalance the roctx stack.
#ifndef OMEGA_AMD_PACER_H
#define OMEGA_AMD_PACER_H

namespace OMEGA {
namespace Pacer {

// (defined in Device.cpp: code that includes this header links libomega_amd only, not the roctx library)
/// timers above this level are not emitted (reference: Pacer::setTimingLevel, Pacer.cpp:138-150)
int &timingLevel();
bool start(const char *Name, int Level = 0);
bool stop(const char *Name, int Level = 0);
/// scoped start / stop
/// (remembers whether its push happened: a change of the timing level inside the range does not unbalance the stack)
struct Range {
   bool Pushed;
   Range(const char *Name, int Level = 0) : Pushed(Level <= timingLevel()) {
      if (Pushed)
         start(Name, -1000000);
   }
   ~Range() {
      if (Pushed)
         stop(nullptr, -1000000);
   }
   Range(const Range &)            = delete;
   Range &operator=(const Range &) = delete;
};

} // namespace Pacer
} // namespace OMEGA
#endif
