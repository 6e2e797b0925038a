// OceanState.h -- prognostic state containers: OceanState (LayerThickness, NormalVelocity
// with NTimeLevels circular time levels) and Tracers.  Interfaces follow the reference
// (components/omega/src/ocn/OceanState.h:85-149, OceanState.cpp:247-407;
//  components/omega/src/ocn/Tracers.h, Tracers.cpp:269,457-496).  The tracer arrays live in
// TracerStore INSTANCES (several states may coexist: tests, provisional buffers); `Tracers` is the
// reference's all-static interface forwarding to a default instance, so that reference call sites
// (Tracers::getAll(Array, Level), Tracers::updateTimeLevels()) compile unchanged.  The name / group /
// Field registry of the reference's Tracers is out of scope.
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
   // ---- the reference's signatures (OceanState.h:113-129): on this object's `Stream` (default: the null stream)
   hipStream_t Stream = nullptr;
   I4 exchangeHalo(I4 TimeLevel) { return exchangeHalo(TimeLevel, Stream); }
   void updateTimeLevels() { updateTimeLevels(Stream); }

 private:
   Halo *MeshHalo;
   I4 CurTimeIndex = 0;
};

class TracerStore {
 public:
   TracerStore(const HorzMesh *Mesh, Halo *MeshHalo, int NVertLayers, int NTracers, int NTimeLevels);
   ~TracerStore(); ///< (a store that is the static interface's default stops being it)
   TracerStore(const TracerStore &) = delete;
   TracerStore &operator=(const TracerStore &) = delete;
   hipStream_t Stream = nullptr; ///< stream of the static `Tracers` interface's calls on this store
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

/// The reference's interface (Tracers.h: "all of variables and methods are static"): forwards to the default
/// TracerStore -- the one made by Tracers::init, or any store handed to Tracers::setDefault (not owned then).
class Tracers {
 public:
   /// allocates the default store (the reference's init() reads the tracer list from the config: out of scope)
   static TracerStore *init(const HorzMesh *Mesh, Halo *MeshHalo, int NVertLayers, int NTracers, int NTimeLevels);
   static void setDefault(TracerStore *Store); ///< use a caller-owned store as the default (nullptr: none)
   static TracerStore *getDefault();           ///< nullptr if there is none
   static I4 clear();                          ///< Tracers::clear: drops the default store
   static I4 getNumTracers();
   static I4 getTimeIndex(I4 &TimeIndex, I4 TimeLevel);
   static I4 getAll(Array3DReal &TracerArray, I4 TimeLevel); ///< Tracers.cpp:269; -1 without a default store
   static I4 exchangeHalo(I4 TimeLevel);                     ///< Tracers.cpp:457-467
   static void updateTimeLevels();                           ///< Tracers.cpp:473-496: exchange of the new level, then rotate
};

} // namespace OMEGA
#endif
