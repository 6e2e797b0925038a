// OceanState.h -- prognostic state containers: OceanState (LayerThickness, NormalVelocity
// with NTimeLevels circular time levels) and Tracers.  Interfaces follow the reference
// (components/omega/src/ocn/OceanState.h:85-149, OceanState.cpp:247-407;
//  components/omega/src/ocn/Tracers.h, Tracers.cpp:269,457-496).  Tracers is an instance
// class here (the reference's is a static registry; the name/group registry is out of scope).
#ifndef OMEGA_AMD_OCEANSTATE_H
#define OMEGA_AMD_OCEANSTATE_H

#include "Base.h"
#include "HorzMesh.h"

namespace OMEGA {

class Halo;

class OceanState : public Registry<OceanState> {
 public:
   OceanState(const std::string &Name, const HorzMesh *Mesh, Halo *MeshHalo, int NVertLayers, int NTimeLevels);

   std::string Name;
   I4 NCellsOwned, NCellsAll, NCellsSize, NEdgesOwned, NEdgesAll, NEdgesSize;
   I4 NTimeLevels, NVertLayers;
   std::vector<Array2DReal> LayerThickness; ///< [NTimeLevels] (NCellsSize, NVertLayers)
   std::vector<Array2DReal> NormalVelocity; ///< [NTimeLevels] (NEdgesSize, NVertLayers)

   /// TimeLevel: 1 new, 0 current, -1 previous ...  (OceanState.cpp:394-407)
   I4 getTimeIndex(I4 &TimeIndex, I4 TimeLevel) const;
   I4 getLayerThickness(Array2DReal &LayerThick, I4 TimeLevel) const;
   I4 getNormalVelocity(Array2DReal &NormVel, I4 TimeLevel) const;
   I4 exchangeHalo(I4 TimeLevel, hipStream_t S);
   void updateTimeLevels(hipStream_t S); ///< halo exchange of level 1, then rotate
   void rotateTimeLevels();              ///< index rotation only (caller already exchanged)
   I4 copyToDevice(const Real *HostLayerThick, const Real *HostNormVel, I4 TimeLevel);
   I4 copyToHost(Real *HostLayerThick, Real *HostNormVel, I4 TimeLevel) const;

 private:
   Halo *MeshHalo;
   I4 CurTimeIndex = 0;
};

class Tracers {
 public:
   Tracers(const HorzMesh *Mesh, Halo *MeshHalo, int NVertLayers, int NTracers, int NTimeLevels);
   I4 NTracers, NTimeLevels, NVertLayers, NCellsOwned, NCellsAll, NCellsSize;
   std::vector<Array3DReal> TracerArrays; ///< [NTimeLevels] (NTracers, NCellsSize, NVertLayers)
   I4 getNumTracers() const { return NTracers; }
   I4 getTimeIndex(I4 &TimeIndex, I4 TimeLevel) const;
   I4 getAll(Array3DReal &TracerArray, I4 TimeLevel) const;
   I4 exchangeHalo(I4 TimeLevel, hipStream_t S);
   void updateTimeLevels(hipStream_t S);
   void rotateTimeLevels();
   I4 copyToDevice(const Real *HostTracers, I4 TimeLevel);
   I4 copyToHost(Real *HostTracers, I4 TimeLevel) const;

 private:
   Halo *MeshHalo;
   I4 CurTimeIndex = 0;
};

} // namespace OMEGA
#endif
