// GraphCache.h -- replay of launch-bound kernel sequences as HIP graphs.
//
// A fused RHS is 7 launches, a stage-fused RK4 step 28; on small meshes (QU240: 7 k cells, or the per-GPU share of
// a partitioned mesh) each kernel runs for a few microseconds and the sequence is bound by the host's launch rate.
// A sequence is identified by a key (every pointer and scalar that enters the launches); the first time a key is
// seen the sequence runs directly (lazy allocations happen here), the second time it is captured from the stream
// into a graph, afterwards the instantiated graph is launched: one host call per sequence.
// Only sequences that consist of kernel launches on ONE non-default stream qualify (no host callbacks, no
// synchronisation, no allocation) -- the callers check that.
// Measured on MI355X: replay does not shorten even the smallest workload (QU240-sized:
// 86 us direct, 91 us replayed) -- the sequences are bound by the GPU's per-kernel latency, not by the host -- so
// replay is OFF unless asked for (Tendencies::UseGraphs / RungeKutta4Stepper::UseGraphs, or OMEGA_GRAPHS=1).
#ifndef OMEGA_AMD_GRAPHCACHE_H
#define OMEGA_AMD_GRAPHCACHE_H

#include "Base.h"

#include <cstdlib>
#include <cstring>
#include "Tuning.h"

#include <functional>

namespace OMEGA {

class GraphCache {
 public:
   using Key = std::vector<unsigned long long>;
   ~GraphCache() { clear(); }
   void clear() {
      for (Entry &E : Entries)
         if (E.Exec)
            (void)hipGraphExecDestroy(E.Exec);
      Entries.clear();
   }
   /// default of the UseGraphs switches (option Graphs = 1), and the global veto (Graphs = 0): Tuning.h
   static bool defaultOn() { return tuning().Graphs > 0; }
   static bool enabled() { return tuning().Graphs != 0; }
   template <class T> static void add(Key &K, const T &V) {
      unsigned long long W[(sizeof(T) + 7) / 8] = {};
      std::memcpy(W, &V, sizeof(T));
      for (unsigned long long X : W)
         K.push_back(X);
   }
   I8 NReplays = 0, NCaptures = 0;

   /// Runs Launch() -- directly, or captured / replayed as a graph on S.
   void run(const Key &K, hipStream_t S, const std::function<void()> &Launch) {
      if (!enabled() || S == nullptr) { // the legacy default stream cannot be captured
         Launch();
         return;
      }
      Entry *E = nullptr;
      for (Entry &X : Entries)
         if (X.K == K)
            E = &X;
      if (!E) { // first sight: run directly (allocations, table uploads)
         if (Entries.size() >= MaxEntries) {
            if (Entries.front().Exec)
               (void)hipGraphExecDestroy(Entries.front().Exec);
            Entries.erase(Entries.begin());
         }
         Entries.push_back(Entry{K, nullptr, false});
         Launch();
         return;
      }
      if (!E->Exec && !E->Failed) {
         hipGraph_t G = nullptr;
         if (hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            E->Failed = true;
            Launch();
            return;
         }
         bool Threw = false;
         try {
            Launch();
         } catch (...) {
            Threw = true;
         }
         const hipError_t R = hipStreamEndCapture(S, &G);
         if (Threw || R != hipSuccess || !G ||
             hipGraphInstantiate(&E->Exec, G, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (G)
               (void)hipGraphDestroy(G);
            E->Exec   = nullptr;
            E->Failed = true;
            OMEGA_REQUIRE(!Threw, "GraphCache: a launch failed during stream capture");
            Launch(); // nothing ran during the capture: run the sequence for real
            return;
         }
         (void)hipGraphDestroy(G);
         ++NCaptures;
      }
      if (E->Exec) {
         HIP_CHECK(hipGraphLaunch(E->Exec, S));
         ++NReplays;
      } else {
         Launch();
      }
   }

 private:
   struct Entry {
      Key K;
      hipGraphExec_t Exec;
      bool Failed;
   };
   static constexpr size_t MaxEntries = 16;
   std::vector<Entry> Entries;
};

} // namespace OMEGA
#endif
