// AuxiliaryState.cpp -- see AuxiliaryState.h.
#include "AuxiliaryState.h"
#include "Halo.h"
#include "Pacer.h"

namespace OMEGA {

AuxiliaryState::AuxiliaryState(const std::string &Name_, const HorzMesh *Mesh_, Halo *MeshHalo_, int K, int NT)
    : Mesh(Mesh_), MeshHalo(MeshHalo_), Name(Name_), NVertLayers(K), NTracers(NT) {
   const int NC = Mesh->NCellsSize, NE = Mesh->NEdgesSize, NV = Mesh->NVerticesSize;
   const int NTa = NT > 0 ? NT : 1;
   KineticAux.KineticEnergyCell         = Array2DReal::levels("KineticEnergyCell", NC, K);
   KineticAux.VelocityDivCell           = Array2DReal::levels("VelocityDivCell", NC, K);
   LayerThicknessAux.FluxLayerThickEdge = Array2DReal::levels("FluxLayerThickEdge", NE, K);
   LayerThicknessAux.MeanLayerThickEdge = Array2DReal::levels("MeanLayerThickEdge", NE, K);
   LayerThicknessAux.SshCell            = Array2DReal::levels("SshCell", NC, K);
   VorticityAux.RelVortVertex           = Array2DReal::levels("RelVortVertex", NV, K);
   VorticityAux.NormRelVortVertex       = Array2DReal::levels("NormRelVortVertex", NV, K);
   VorticityAux.NormPlanetVortVertex    = Array2DReal::levels("NormPlanetVortVertex", NV, K);
   VorticityAux.InvThickVertex          = Array2DReal::levels("InvThickVertex", NV, K);
   VorticityAux.NormRelVortEdge         = Array2DReal::levels("NormRelVortEdge", NE, K);
   VorticityAux.NormPlanetVortEdge      = Array2DReal::levels("NormPlanetVortEdge", NE, K);
   VelocityDel2Aux.Del2Edge             = Array2DReal::levels("Del2Edge", NE, K);
   VelocityDel2Aux.Del2DivCell          = Array2DReal::levels("Del2DivCell", NC, K);
   VelocityDel2Aux.Del2RelVortVertex    = Array2DReal::levels("Del2RelVortVertex", NV, K);
   WindForcingAux.NormalStressEdge      = Array1DReal("NormalStressEdge", NE);
   WindForcingAux.ZonalStressCell       = Array1DReal("ZonalStressCell", NC);
   WindForcingAux.MeridStressCell       = Array1DReal("MeridStressCell", NC);
   TracerAux.HTracersEdge               = Array3DReal::levels("HTracersEdge", NTa, NE, K);
   TracerAux.Del2TracersCell            = Array3DReal::levels("Del2TracersCell", NTa, NC, K);
}

AuxPtrs AuxiliaryState::ptrs() const {
   AuxPtrs A;
   A.KineticEnergyCell    = KineticAux.KineticEnergyCell.Ptr;
   A.VelocityDivCell      = KineticAux.VelocityDivCell.Ptr;
   A.FluxLayerThickEdge   = LayerThicknessAux.FluxLayerThickEdge.Ptr;
   A.MeanLayerThickEdge   = LayerThicknessAux.MeanLayerThickEdge.Ptr;
   A.SshCell              = LayerThicknessAux.SshCell.Ptr;
   A.RelVortVertex        = VorticityAux.RelVortVertex.Ptr;
   A.NormRelVortVertex    = VorticityAux.NormRelVortVertex.Ptr;
   A.NormPlanetVortVertex = VorticityAux.NormPlanetVortVertex.Ptr;
   A.InvThickVertex       = VorticityAux.InvThickVertex.Ptr;
   A.NormRelVortEdge      = VorticityAux.NormRelVortEdge.Ptr;
   A.NormPlanetVortEdge   = VorticityAux.NormPlanetVortEdge.Ptr;
   A.Del2Edge             = VelocityDel2Aux.Del2Edge.Ptr;
   A.Del2DivCell          = VelocityDel2Aux.Del2DivCell.Ptr;
   A.Del2RelVortVertex    = VelocityDel2Aux.Del2RelVortVertex.Ptr;
   A.HTracersEdge         = TracerAux.HTracersEdge.Ptr;
   A.Del2TracersCell      = TracerAux.Del2TracersCell.Ptr;
   A.NormalStressEdge     = WindForcingAux.NormalStressEdge.Ptr;
   A.ZonalStressCell      = WindForcingAux.ZonalStressCell.Ptr;
   A.MeridStressCell      = WindForcingAux.MeridStressCell.Ptr;
   return A;
}

void AuxiliaryState::computeMomAux(const OceanState *State, int ThickTimeLevel, int VelTimeLevel, hipStream_t S) const {
   Array2DReal LayerThickCell, NormalVelEdge;
   OMEGA_REQUIRE(State->getLayerThickness(LayerThickCell, ThickTimeLevel) == 0, "AuxiliaryState: bad thickness time level");
   OMEGA_REQUIRE(State->getNormalVelocity(NormalVelEdge, VelTimeLevel) == 0, "AuxiliaryState: bad velocity time level");
   const MeshView &M = Mesh->view();
   const AuxPtrs A   = ptrs();
   const int K       = NVertLayers;
   const int Upwind  = LayerThicknessAux.FluxThickEdgeChoice == FluxThickEdgeOption::Upwind;
   Pacer::Range Timer("AuxState:computeMomAux", 1);
   auto Timed = [](const char *Name, auto &&Launch) { // the reference's level-2 timer around each launch
      Pacer::Range T2(Name, 2);
      Launch();
   };
   Timed("AuxState:vertexAuxState1", [&] { launchVertexAuxState1(M, K, A, LayerThickCell.Ptr, NormalVelEdge.Ptr, S); }); // :79-85
   Timed("AuxState:cellAuxState1", [&] { launchCellAuxState1(M, K, A, NormalVelEdge.Ptr, S); });                       // :88-93
   Timed("AuxState:edgeAuxState1", [&] {
      launchEdgeAuxState1(M, A, WindForcingAux.InterpChoice == InterpCellToEdgeOption::Isotropic, S); // :99-103
   });
   Timed("AuxState:edgeAuxState2", [&] { launchEdgeAuxState2(M, K, A, LayerThickCell.Ptr, NormalVelEdge.Ptr, Upwind, S); }); // :106-115
   Timed("AuxState:vertexAuxState2", [&] { launchVertexAuxState2(M, K, A, S); });                                     // :118-123
   Timed("AuxState:cellAuxState2", [&] { launchCellAuxState2(M, K, A, S); });                                         // :126-131
   Timed("AuxState:cellAuxState3", [&] { launchCellAuxState3(M, K, A, LayerThickCell.Ptr, S); });                     // :134-140
}

void AuxiliaryState::computeAll(const OceanState *State, const Array3DReal &TracerArray, int ThickTimeLevel,
                                int VelTimeLevel, hipStream_t S) const {
   Array2DReal LayerThickCell, NormalVelEdge;
   OMEGA_REQUIRE(State->getLayerThickness(LayerThickCell, ThickTimeLevel) == 0, "AuxiliaryState: bad thickness time level");
   OMEGA_REQUIRE(State->getNormalVelocity(NormalVelEdge, VelTimeLevel) == 0, "AuxiliaryState: bad velocity time level");
   Pacer::Range Timer("AuxState:computeAll", 1);
   computeMomAux(State, ThickTimeLevel, VelTimeLevel, S);
   const MeshView &M = Mesh->view();
   const AuxPtrs A   = ptrs();
   const int Upwind  = TracerAux.TracersOnEdgeChoice == FluxTracerEdgeOption::Upwind;
   Pacer::start("AuxState:edgeAuxState4", 2);
   launchEdgeAuxState4(M, NVertLayers, NTracers, A, NormalVelEdge.Ptr, LayerThickCell.Ptr, TracerArray.Ptr, Upwind, S); // :165-171
   Pacer::stop("AuxState:edgeAuxState4", 2);
   Pacer::start("AuxState:cellAuxState4", 2);
   launchCellAuxState4(M, NVertLayers, NTracers, A, TracerArray.Ptr, S);                                               // :176-182
   Pacer::stop("AuxState:cellAuxState4", 2);
}

I4 AuxiliaryState::exchangeHalo(hipStream_t S) {
   if (!MeshHalo)
      return 0;
   Array2DReal Z, Mv;
   // 1-D arrays are exchanged as (N, 1)
   Z.Ptr = WindForcingAux.ZonalStressCell.Ptr, Z.Ext[0] = Mesh->NCellsSize, Z.Ext[1] = 1, Z.Pitch = 1;
   Mv.Ptr = WindForcingAux.MeridStressCell.Ptr, Mv.Ext[0] = Mesh->NCellsSize, Mv.Ext[1] = 1, Mv.Pitch = 1;
   I4 Err = MeshHalo->exchangeFullArrayHalo(Z, OnCell, S);
   Err += MeshHalo->exchangeFullArrayHalo(Mv, OnCell, S);
   return Err;
}

} // namespace OMEGA
