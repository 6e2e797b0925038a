// FusedKernels.hip -- entry points of the fused right-hand side: which instantiation of launchFusedT a mesh takes.
// The kernel bodies and launchFusedT itself are in FusedKernelsImpl.h; the instantiations are compiled by FusedInst*.hip.
#include "FusedKernelsImpl.h"

namespace OMEGA {

const char *FusedKernelNames[FusedNumKernels] = {"", "", "", "", "", "", ""};

bool fusedRHSSupported(const MeshView &M, int K) {
   const size_t MaxRows = (size_t)(M.NEdgesSize > M.NCellsSize ? M.NEdgesSize : M.NCellsSize);
   const size_t Rows    = MaxRows > (size_t)M.NVerticesSize ? MaxRows : (size_t)M.NVerticesSize;
   return M.MaxEdges >= 5 && M.MaxEdges <= 8 && Rows * (size_t)levelPitch(K) * 8 <= (size_t)BufOOB;
}

OMEGA_FUSED_INSTANCES_5(OMEGA_FUSED_DECLARE)
OMEGA_FUSED_INSTANCES_5N(OMEGA_FUSED_DECLARE)
OMEGA_FUSED_INSTANCES_6A(OMEGA_FUSED_DECLARE)
OMEGA_FUSED_INSTANCES_6B(OMEGA_FUSED_DECLARE)
OMEGA_FUSED_INSTANCES_6N(OMEGA_FUSED_DECLARE)
OMEGA_FUSED_INSTANCES_7(OMEGA_FUSED_DECLARE)
OMEGA_FUSED_INSTANCES_7N(OMEGA_FUSED_DECLARE)
OMEGA_FUSED_INSTANCES_8(OMEGA_FUSED_DECLARE)

/// does the stage-fused variant cover this mesh / option set?  (same conditions launchFusedT checks
/// on its way to CellPVFinalBody)
static bool stageFusedSupported(const MeshView &M, const TendParams &P, Real *EdgeScratch) {
   return isDefaultTermSet(P) && M.CellPVOK && M.CellPVFinalOK && EdgeScratch;
}

bool launchFusedRHS(const MeshView &M, int K, int NT, const TendParams &P, const AuxPtrs &A, Real *HTend, Real *UTend,
                    Real *TrTend, const Real *H, const Real *U, const Real *Tr, hipStream_t S, hipEvent_t *Ev,
                    Real *EdgeScratch, const StageUpdate *Stage, const MeshView *Narrow) {
   const bool Fast = isDefaultTermSet(P);
   if (Stage && !stageFusedSupported(M, P, EdgeScratch))
      return false;
   // narrow cell tables (HorzMesh::narrowView): the sweeps run the (MaxEdges-1)-slot kernels on them, the cells with
   // MaxEdges edges go through list launches on M.  (Not with run-time option flags and 8 slots: the merged level-1
   // kernel is not instantiated for that, and both widths must take the same level-1 structure.)
   if (Narrow && tuning().NarrowTables != 0 && EdgeScratch && (Fast || Narrow->MaxEdges <= 6)) {
      switch (Narrow->MaxEdges) {
#define OMEGA_NARROW_CASE(MN_)                                                                                     \
   case MN_:                                                                                                       \
      if (Fast)                                                                                                    \
         launchFusedT<MN_, true, MN_, true>(*Narrow, K, NT, P, A, HTend, UTend, TrTend, H, U, Tr, S, Ev, EdgeScratch, Stage, &M); \
      else                                                                                                         \
         launchFusedT<MN_, false, MN_, true>(*Narrow, K, NT, P, A, HTend, UTend, TrTend, H, U, Tr, S, Ev, EdgeScratch, nullptr, &M); \
      return true;
#ifndef OMEGA_ONLY_ME6
         OMEGA_NARROW_CASE(5)
         OMEGA_NARROW_CASE(6)
         OMEGA_NARROW_CASE(7)
#endif
#undef OMEGA_NARROW_CASE
      default:
         break;
      }
   }
   // (the sweeps' valence: MaxEdges, or MaxEdges-1 where that is what most cells have -- default term set, ME >= 6)
#define OMEGA_DISPATCH_DOM(ME_)                                                                                    \
   do {                                                                                                            \
      if constexpr ((ME_) >= 6) {                                                                                  \
         if (M.DomM1) {                                                                                            \
            launchFusedT<ME_, true, (ME_)-1>(M, K, NT, P, A, HTend, UTend, TrTend, H, U, Tr, S, Ev, EdgeScratch,   \
                                             Stage);                                                               \
            break;                                                                                                 \
         }                                                                                                         \
      }                                                                                                            \
      launchFusedT<ME_, true>(M, K, NT, P, A, HTend, UTend, TrTend, H, U, Tr, S, Ev, EdgeScratch, Stage);          \
   } while (0)
#define OMEGA_CASE(ME_)                                                                                            \
   case ME_:                                                                                                       \
      if (Fast)                                                                                                    \
         OMEGA_DISPATCH_DOM(ME_);                                                                                 \
      else                                                                                                         \
         launchFusedT<ME_, false>(M, K, NT, P, A, HTend, UTend, TrTend, H, U, Tr, S, Ev, EdgeScratch, nullptr);    \
      break;
   switch (M.MaxEdges) {
#ifndef OMEGA_ONLY_ME6 // (measurement builds of a kernel experiment: hexagon meshes only, a quarter of the compile time)
      OMEGA_CASE(5)
#endif
      OMEGA_CASE(6)
#ifndef OMEGA_ONLY_ME6
      OMEGA_CASE(7)
      OMEGA_CASE(8)
#endif
   default:
      return false; // callers check fusedRHSSupported()
   }
#undef OMEGA_CASE
#undef OMEGA_DISPATCH_DOM
   return true;
}

} // namespace OMEGA
