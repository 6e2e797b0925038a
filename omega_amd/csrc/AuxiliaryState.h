// AuxiliaryState.h -- diagnostic ("auxiliary") fields of the RHS.
// Interface and array names after the reference (components/omega/src/ocn/
// AuxiliaryState.h:37-82 and auxiliaryVars/*.h public members).
#ifndef OMEGA_AMD_AUXILIARYSTATE_H
#define OMEGA_AMD_AUXILIARYSTATE_H

#include "Base.h"
#include "HorzMesh.h"
#include "OceanState.h"
#include "kernels/Kernels.h"

namespace OMEGA {

enum class FluxThickEdgeOption { Center, Upwind };
enum class FluxTracerEdgeOption { Center, Upwind };
enum class InterpCellToEdgeOption { Anisotropic, Isotropic };

struct KineticAuxVars {
   Array2DReal KineticEnergyCell, VelocityDivCell;
};
struct LayerThicknessAuxVars {
   Array2DReal FluxLayerThickEdge, MeanLayerThickEdge, SshCell;
   FluxThickEdgeOption FluxThickEdgeChoice = FluxThickEdgeOption::Center;
};
struct VorticityAuxVars {
   Array2DReal RelVortVertex, NormRelVortVertex, NormPlanetVortVertex, NormRelVortEdge, NormPlanetVortEdge;
   Array2DReal InvThickVertex; ///< 1/LayerThickVertex: internal to the fused RHS (not a reference array)
};
struct VelocityDel2AuxVars {
   Array2DReal Del2Edge, Del2DivCell, Del2RelVortVertex;
};
struct WindForcingAuxVars {
   Array1DReal NormalStressEdge, ZonalStressCell, MeridStressCell;
   InterpCellToEdgeOption InterpChoice = InterpCellToEdgeOption::Isotropic;
};
struct TracerAuxVars {
   Array3DReal HTracersEdge, Del2TracersCell;
   FluxTracerEdgeOption TracersOnEdgeChoice = FluxTracerEdgeOption::Center;
};

class AuxiliaryState : public Registry<AuxiliaryState> {
 public:
   AuxiliaryState(const std::string &Name, const HorzMesh *Mesh, Halo *MeshHalo, int NVertLayers, int NTracers);

   KineticAuxVars KineticAux;
   LayerThicknessAuxVars LayerThicknessAux;
   VorticityAuxVars VorticityAux;
   VelocityDel2AuxVars VelocityDel2Aux;
   WindForcingAuxVars WindForcingAux;
   TracerAuxVars TracerAux;

   /// AuxiliaryState::computeMomAux (AuxiliaryState.cpp:60-143)
   void computeMomAux(const OceanState *State, int ThickTimeLevel, int VelTimeLevel, hipStream_t S) const;
   /// AuxiliaryState::computeAll (AuxiliaryState.cpp:146-185)
   void computeAll(const OceanState *State, const Array3DReal &TracerArray, int ThickTimeLevel, int VelTimeLevel,
                   hipStream_t S) const;
   /// exchange of the non-computed aux variables (AuxiliaryState.cpp:312-323)
   I4 exchangeHalo(hipStream_t S);
   // ---- the reference's signatures (AuxiliaryState.h:74-85): on this object's `Stream` (default: the null stream)
   hipStream_t Stream = nullptr;
   void computeMomAux(const OceanState *State, int ThickTimeLevel, int VelTimeLevel) const {
      computeMomAux(State, ThickTimeLevel, VelTimeLevel, Stream);
   }
   void computeAll(const OceanState *State, const Array3DReal &TracerArray, int ThickTimeLevel, int VelTimeLevel) const {
      computeAll(State, TracerArray, ThickTimeLevel, VelTimeLevel, Stream);
   }
   I4 exchangeHalo() { return exchangeHalo(Stream); }

   AuxPtrs ptrs() const;
   const HorzMesh *Mesh;
   Halo *MeshHalo;
   std::string Name;
   int NVertLayers, NTracers;
};

} // namespace OMEGA
#endif
