// Tendencies.cpp -- see Tendencies.h.
#include "Tendencies.h"
#include "Pacer.h"
#include "kernels/KernelCommon.h"

namespace OMEGA {

std::string Tendencies::fusedLimit(size_t NCellsSize, size_t NEdgesSize, size_t NVerticesSize, int MaxEdges, int K) {
   if (MaxEdges < 5 || MaxEdges > 8)
      return "MaxEdges = " + std::to_string(MaxEdges) + " is outside 5..8, the table widths the fused RHS is instantiated for";
   const size_t Rows = std::max(NCellsSize, std::max(NEdgesSize, NVerticesSize)), RowBytes = (size_t)levelPitch(K) * 8;
   if (Rows * RowBytes > (size_t)FusedMaxPlaneBytes)
      return "an array plane of " + std::to_string(Rows) + " rows x " + std::to_string(RowBytes) + " B exceeds the 4 GiB that the fused "
             "kernels' 32-bit buffer offsets address: at most " + std::to_string((size_t)FusedMaxPlaneBytes / RowBytes) + " rows (cells, "
             "edges or vertices, halo included) per rank at " + std::to_string(K) + " levels -- ~ " +
             std::to_string((size_t)FusedMaxPlaneBytes / RowBytes / 3) + " cells of a hexagon mesh; partition the mesh over more ranks "
             "(several ranks may share one GPU)";
   return "";
}

Tendencies::Tendencies(const std::string &, const HorzMesh *Mesh_, int K, int NT, const TendParams &Options,
                       bool AllowReferenceStructured)
    : Params(Options), Mesh(Mesh_), NVertLayers(K), NTracers(NT) {
   if (!fusedRHSSupported(Mesh->view(), K)) {
      const std::string Why = fusedLimit((size_t)Mesh->NCellsSize, (size_t)Mesh->NEdgesSize, (size_t)Mesh->NVerticesSize,
                                         Mesh->view().MaxEdges, K);
      OMEGA_REQUIRE(AllowReferenceStructured,
                    "Tendencies: this mesh is outside the fused RHS (" + Why + "); pass AllowReferenceStructured "
                    "(omg_tend_create_reference_structured) to accept the reference-structured 23-launch path, ~ 5 x slower");
      UseFusedRHS = false;
   }
   // Tendency arrays (Tendencies.cpp:233-238)
   LayerThicknessTend = Array2DReal::levels("LayerThicknessTend", Mesh->NCellsSize, K);
   NormalVelocityTend = Array2DReal::levels("NormalVelocityTend", Mesh->NEdgesSize, K);
   TracerTend         = Array3DReal::levels("TracerTend", NT > 0 ? NT : 1, Mesh->NCellsSize, K);
   // the running PV sums of the fused RHS: allocated here, not at the first evaluation (no allocation inside a step)
   if (fusedRHSSupported(Mesh->view(), K))
      EdgeScratch = Array2DReal::levels("EdgeScratch", Mesh->NEdgesSize, K);
}

Tendencies::~Tendencies() {
   for (auto &Set : TimingEvents)
      for (auto &E : Set)
         (void)hipEventDestroy(E);
}

void Tendencies::enableKernelTiming(bool On) { TimingOn = On; }

int Tendencies::collectKernelTimes(double *MsSum) {
   for (int I = 0; I < FusedNumKernels; ++I)
      MsSum[I] = 0.0;
   const int N = (int)TimingEvents.size();
   for (auto &Set : TimingEvents) {
      HIP_CHECK(hipEventSynchronize(Set[FusedNumKernels]));
      for (int I = 0; I < FusedNumKernels; ++I) {
         float Ms = 0.f;
         HIP_CHECK(hipEventElapsedTime(&Ms, Set[I], Set[I + 1]));
         MsSum[I] += Ms;
      }
      for (auto &E : Set)
         (void)hipEventDestroy(E);
   }
   TimingEvents.clear();
   return N;
}

TendParams Tendencies::paramsFor(const AuxiliaryState *Aux) const {
   TendParams P          = Params;
   P.FluxThicknessUpwind = Aux->LayerThicknessAux.FluxThickEdgeChoice == FluxThickEdgeOption::Upwind;
   P.FluxTracerUpwind    = Aux->TracerAux.TracersOnEdgeChoice == FluxTracerEdgeOption::Upwind;
   P.WindInterpIsotropic = Aux->WindForcingAux.InterpChoice == InterpCellToEdgeOption::Isotropic;
   return P;
}

// Tendencies.cpp:257-297
void Tendencies::computeThicknessTendenciesOnly(const OceanState *State, const AuxiliaryState *Aux, int ThickLvl, int VelLvl,
                                                hipStream_t S) {
   Pacer::Range Timer("Tend:computeThicknessTendenciesOnly", 1);
   Array2DReal NormalVelEdge;
   OMEGA_REQUIRE(State->getNormalVelocity(NormalVelEdge, VelLvl) == 0, "Tendencies: bad velocity time level");
   launchThicknessTendOnly(Mesh->view(), NVertLayers, paramsFor(Aux), Aux->ptrs(), LayerThicknessTend.Ptr,
                           NormalVelEdge.Ptr, S);
   if (CustomThicknessTend) { // Tendencies.cpp:288-291
      Pacer::Range T2("Tend:customThicknessTend", 2);
      CustomThicknessTend(LayerThicknessTend, State, Aux, ThickLvl, VelLvl, ModelTime, S);
   }
}
// Tendencies.cpp:301-423
void Tendencies::computeVelocityTendenciesOnly(const OceanState *State, const AuxiliaryState *Aux, int ThickLvl, int VelLvl,
                                               hipStream_t S) {
   Pacer::Range Timer("Tend:computeVelocityTendenciesOnly", 1);
   Array2DReal NormalVelEdge;
   OMEGA_REQUIRE(State->getNormalVelocity(NormalVelEdge, VelLvl) == 0, "Tendencies: bad velocity time level");
   launchVelocityTendOnly(Mesh->view(), NVertLayers, paramsFor(Aux), Aux->ptrs(), NormalVelocityTend.Ptr,
                          NormalVelEdge.Ptr, S);
   if (CustomVelocityTend) { // Tendencies.cpp:416-419
      Pacer::Range T2("Tend:customVelocityTend", 2);
      CustomVelocityTend(NormalVelocityTend, State, Aux, ThickLvl, VelLvl, ModelTime, S);
   }
}
// Tendencies.cpp:427-486
void Tendencies::computeTracerTendenciesOnly(const OceanState *State, const AuxiliaryState *Aux,
                                             const Array3DReal &TracerArray, int, int VelLvl, hipStream_t S) {
   Pacer::Range Timer("Tend:computeTracerTendenciesOnly", 1);
   Array2DReal NormalVelEdge;
   OMEGA_REQUIRE(State->getNormalVelocity(NormalVelEdge, VelLvl) == 0, "Tendencies: bad velocity time level");
   launchTracerTendOnly(Mesh->view(), NVertLayers, NTracers, paramsFor(Aux), Aux->ptrs(), TracerTend.Ptr,
                        NormalVelEdge.Ptr, TracerArray.Ptr, S);
}
// Tendencies.cpp:488-519
void Tendencies::computeThicknessTendencies(const OceanState *State, const AuxiliaryState *Aux, int ThickLvl, int VelLvl,
                                            hipStream_t S) {
   Pacer::Range Timer("Tend:computeThicknessTendencies", 1);
   Array2DReal LayerThick, NormVel;
   OMEGA_REQUIRE(State->getLayerThickness(LayerThick, ThickLvl) == 0 && State->getNormalVelocity(NormVel, VelLvl) == 0,
                 "Tendencies: bad time level");
   const TendParams P = paramsFor(Aux);
   Pacer::start("Tend:computeLayerThickAux", 2);
   launchLayerThickAuxEdge(Mesh->view(), NVertLayers, Aux->ptrs(), LayerThick.Ptr, NormVel.Ptr, P.FluxThicknessUpwind, S);
   Pacer::stop("Tend:computeLayerThickAux", 2);
   computeThicknessTendenciesOnly(State, Aux, ThickLvl, VelLvl, S);
}
// Tendencies.cpp:521-535
void Tendencies::computeVelocityTendencies(const OceanState *State, const AuxiliaryState *Aux, int ThickLvl, int VelLvl,
                                           hipStream_t S) {
   Pacer::Range Timer("Tend:computeVelocityTendencies", 1);
   Aux->computeMomAux(State, ThickLvl, VelLvl, S);
   computeVelocityTendenciesOnly(State, Aux, ThickLvl, VelLvl, S);
}
// Tendencies.cpp:537-575
void Tendencies::computeTracerTendencies(const OceanState *State, const AuxiliaryState *Aux,
                                         const Array3DReal &TracerArray, int ThickLvl, int VelLvl, hipStream_t S) {
   Pacer::Range Timer("Tend:computeTracerTendencies", 1);
   Array2DReal LayerThick, NormVel;
   OMEGA_REQUIRE(State->getLayerThickness(LayerThick, ThickLvl) == 0 && State->getNormalVelocity(NormVel, VelLvl) == 0,
                 "Tendencies: bad time level");
   const TendParams P = paramsFor(Aux);
   Pacer::start("Tend:computeTracerAuxEdge", 2);
   launchEdgeAuxState4(Mesh->view(), NVertLayers, NTracers, Aux->ptrs(), NormVel.Ptr, LayerThick.Ptr, TracerArray.Ptr,
                       P.FluxTracerUpwind, S);
   Pacer::stop("Tend:computeTracerAuxEdge", 2);
   Pacer::start("Tend:computeTracerAuxCell", 2);
   launchCellAuxState4(Mesh->view(), NVertLayers, NTracers, Aux->ptrs(), TracerArray.Ptr, S);
   Pacer::stop("Tend:computeTracerAuxCell", 2);
   computeTracerTendenciesOnly(State, Aux, TracerArray, ThickLvl, VelLvl, S);
}
// Tendencies.cpp:579-600
bool Tendencies::computeAllTendenciesStage(const OceanState *State, const AuxiliaryState *Aux,
                                           const Array3DReal &TracerArray, int ThickLvl, int VelLvl,
                                           const StageUpdate &Stage, hipStream_t S) {
   Pacer::Range Timer("Tend:computeAllTendencies", 1);
   if (!(UseFusedRHS && fusedRHSSupported(Mesh->view(), NVertLayers)))
      return false;
   if (CustomThicknessTend || CustomVelocityTend)
      return false; // the custom terms are added to the stored tendencies: needs the plain sequence
   Array2DReal LayerThick, NormVel;
   OMEGA_REQUIRE(State->getLayerThickness(LayerThick, ThickLvl) == 0 && State->getNormalVelocity(NormVel, VelLvl) == 0,
                 "Tendencies: bad time level");
   return launchFusedRHS(Mesh->view(), NVertLayers, NTracers, paramsFor(Aux), Aux->ptrs(), LayerThicknessTend.Ptr,
                         NormalVelocityTend.Ptr, TracerTend.Ptr, LayerThick.Ptr, NormVel.Ptr, TracerArray.Ptr, S, nullptr,
                         EdgeScratch.Ptr, &Stage, Mesh->narrowView());
}

void Tendencies::computeAllTendencies(const OceanState *State, const AuxiliaryState *Aux, const Array3DReal &TracerArray,
                                      int ThickLvl, int VelLvl, hipStream_t S) {
   Pacer::Range Timer("Tend:computeAllTendencies", 1);
   // (a mesh outside the fused RHS -- Tendencies::fusedLimit -- only gets here if the caller accepted the 23-launch path
   // when the object was made: the constructor fails otherwise)
   if (UseFusedRHS && fusedRHSSupported(Mesh->view(), NVertLayers)) {
      Array2DReal LayerThick, NormVel;
      OMEGA_REQUIRE(State->getLayerThickness(LayerThick, ThickLvl) == 0 &&
                        State->getNormalVelocity(NormVel, VelLvl) == 0,
                    "Tendencies: bad time level");
      hipEvent_t *Ev = nullptr;
      if (TimingOn && TimingEvents.size() < 4096) {
         TimingEvents.emplace_back(FusedNumKernels + 1);
         for (auto &E : TimingEvents.back())
            HIP_CHECK(hipEventCreate(&E));
         Ev = TimingEvents.back().data();
      }
      const TendParams P = paramsFor(Aux);
      auto Launch        = [&]() {
         launchFusedRHS(Mesh->view(), NVertLayers, NTracers, P, Aux->ptrs(), LayerThicknessTend.Ptr,
                        NormalVelocityTend.Ptr, TracerTend.Ptr, LayerThick.Ptr, NormVel.Ptr, TracerArray.Ptr, S, Ev,
                        EdgeScratch.Ptr, nullptr, Mesh->narrowView());
      };
      // wind forcing reads the stress arrays through a non-tile kernel too, still plain launches: capturable
      if (graphsOn() && !Ev && !CustomThicknessTend && !CustomVelocityTend) {
         GraphCache::Key Key;
         GraphCache::add(Key, LayerThick.Ptr), GraphCache::add(Key, NormVel.Ptr), GraphCache::add(Key, TracerArray.Ptr);
         GraphCache::add(Key, Aux), GraphCache::add(Key, P), GraphCache::add(Key, S);
         GraphCache::add(Key, tuningGeneration()); // (the kernel structure options are read at every launch)
         Graphs.run(Key, S, Launch);
      } else {
         Launch();
      }
      if ((CustomThicknessTend || CustomVelocityTend) && MaterialiseAuxForCustom)
         Aux->computeAll(State, TracerArray, ThickLvl, VelLvl, S); // the hooks may read the AuxiliaryState
      if (CustomThicknessTend)
         CustomThicknessTend(LayerThicknessTend, State, Aux, ThickLvl, VelLvl, ModelTime, S);
      if (CustomVelocityTend)
         CustomVelocityTend(NormalVelocityTend, State, Aux, ThickLvl, VelLvl, ModelTime, S);
      return;
   }
   Aux->computeAll(State, TracerArray, ThickLvl, VelLvl, S);
   computeThicknessTendenciesOnly(State, Aux, ThickLvl, VelLvl, S);
   computeVelocityTendenciesOnly(State, Aux, ThickLvl, VelLvl, S);
   computeTracerTendenciesOnly(State, Aux, TracerArray, ThickLvl, VelLvl, S);
}

} // namespace OMEGA
