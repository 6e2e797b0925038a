// OceanState.cpp -- see OceanState.h.
#include "OceanState.h"
#include "Halo.h"

namespace OMEGA {

OceanState::OceanState(const std::string &Name_, const HorzMesh *Mesh, Halo *MeshHalo_, int NVertLayers_,
                       int NTimeLevels_)
    : Name(Name_), MeshHalo(MeshHalo_) {
   OMEGA_REQUIRE(NTimeLevels_ >= 1 && NTimeLevels_ <= 5, "OceanState: NTimeLevels must be in [1,5]");
   NCellsOwned = Mesh->NCellsOwned, NCellsAll = Mesh->NCellsAll, NCellsSize = Mesh->NCellsSize;
   NEdgesOwned = Mesh->NEdgesOwned, NEdgesAll = Mesh->NEdgesAll, NEdgesSize = Mesh->NEdgesSize;
   NVertLayers = NVertLayers_;
   NTimeLevels = NTimeLevels_;
   for (int I = 0; I < NTimeLevels; ++I) {
      LayerThickness.push_back(Array2DReal::levels("LayerThickness" + std::to_string(I), NCellsSize, NVertLayers));
      NormalVelocity.push_back(Array2DReal::levels("NormalVelocity" + std::to_string(I), NEdgesSize, NVertLayers));
   }
}

// OceanState::getTimeIndex (OceanState.cpp:394-407)
I4 OceanState::getTimeIndex(I4 &TimeIndex, I4 TimeLevel) const {
   if (NTimeLevels > 1 && (TimeLevel > 1 || (TimeLevel + NTimeLevels) <= 1))
      return -1;
   TimeIndex = (TimeLevel + CurTimeIndex + NTimeLevels) % NTimeLevels;
   return 0;
}
I4 OceanState::getLayerThickness(Array2DReal &A, I4 TimeLevel) const {
   I4 Idx;
   if (getTimeIndex(Idx, TimeLevel) != 0)
      return -1;
   A = LayerThickness[Idx];
   return 0;
}
I4 OceanState::getNormalVelocity(Array2DReal &A, I4 TimeLevel) const {
   I4 Idx;
   if (getTimeIndex(Idx, TimeLevel) != 0)
      return -1;
   A = NormalVelocity[Idx];
   return 0;
}
// OceanState::exchangeHalo (OceanState.cpp:333-345)
I4 OceanState::exchangeHalo(I4 TimeLevel, hipStream_t S) {
   I4 Idx;
   if (getTimeIndex(Idx, TimeLevel) != 0)
      return -1;
   if (!MeshHalo)
      return 0;
   I4 Err = MeshHalo->exchangeFullArrayHalo(LayerThickness[Idx], OnCell, S);
   Err += MeshHalo->exchangeFullArrayHalo(NormalVelocity[Idx], OnEdge, S);
   return Err;
}
// OceanState::updateTimeLevels (OceanState.cpp:349-365)
void OceanState::updateTimeLevels(hipStream_t S) {
   OMEGA_REQUIRE(NTimeLevels > 1, "OceanState: can't update time levels for NTimeLevels == 1");
   exchangeHalo(1, S);
   rotateTimeLevels();
}
void OceanState::rotateTimeLevels() { CurTimeIndex = (CurTimeIndex + 1) % NTimeLevels; }

I4 OceanState::copyToDevice(const Real *HH, const Real *HU, I4 TimeLevel) {
   I4 Idx;
   if (getTimeIndex(Idx, TimeLevel) != 0)
      return -1;
   if (HH)
      OMEGA::copyToDevice(LayerThickness[Idx], HH);
   if (HU)
      OMEGA::copyToDevice(NormalVelocity[Idx], HU);
   return 0;
}
I4 OceanState::copyToHost(Real *HH, Real *HU, I4 TimeLevel) const {
   I4 Idx;
   if (getTimeIndex(Idx, TimeLevel) != 0)
      return -1;
   if (HH)
      OMEGA::copyToHost(HH, LayerThickness[Idx]);
   if (HU)
      OMEGA::copyToHost(HU, NormalVelocity[Idx]);
   return 0;
}

TracerStore::TracerStore(const HorzMesh *Mesh, Halo *MeshHalo_, int NVertLayers_, int NTracers_, int NTimeLevels_)
    : MeshHalo(MeshHalo_) {
   OMEGA_REQUIRE(NTimeLevels_ >= 1 && NTracers_ >= 0, "Tracers: bad sizes");
   NTracers = NTracers_, NTimeLevels = NTimeLevels_, NVertLayers = NVertLayers_;
   NCellsOwned = Mesh->NCellsOwned, NCellsAll = Mesh->NCellsAll, NCellsSize = Mesh->NCellsSize;
   for (int I = 0; I < NTimeLevels; ++I)
      TracerArrays.push_back(Array3DReal::levels("TracerArrays" + std::to_string(I), NTracers > 0 ? NTracers : 1,
                                                 NCellsSize, NVertLayers));
}
I4 TracerStore::getTimeIndex(I4 &TimeIndex, I4 TimeLevel) const {
   if (NTimeLevels > 1 && (TimeLevel > 1 || (TimeLevel + NTimeLevels) <= 1))
      return -1;
   TimeIndex = (TimeLevel + CurTimeIndex + NTimeLevels) % NTimeLevels;
   return 0;
}
I4 TracerStore::getAll(Array3DReal &A, I4 TimeLevel) const {
   I4 Idx;
   if (getTimeIndex(Idx, TimeLevel) != 0)
      return -1;
   A = TracerArrays[Idx];
   return 0;
}
I4 TracerStore::exchangeHalo(I4 TimeLevel, hipStream_t S) {
   I4 Idx;
   if (getTimeIndex(Idx, TimeLevel) != 0)
      return -1;
   if (!MeshHalo || NTracers == 0)
      return 0;
   return MeshHalo->exchangeFullArrayHalo(TracerArrays[Idx], OnCell, S);
}
void TracerStore::updateTimeLevels(hipStream_t S) {
   OMEGA_REQUIRE(NTimeLevels > 1, "Tracers: can't update time levels for NTimeLevels == 1");
   exchangeHalo(1, S);
   rotateTimeLevels();
}
void TracerStore::rotateTimeLevels() { CurTimeIndex = (CurTimeIndex + 1) % NTimeLevels; }
I4 TracerStore::copyToDevice(const Real *H, I4 TimeLevel) {
   I4 Idx;
   if (getTimeIndex(Idx, TimeLevel) != 0)
      return -1;
   if (NTracers > 0)
      OMEGA::copyToDevice(TracerArrays[Idx], H);
   return 0;
}
I4 TracerStore::copyToHost(Real *H, I4 TimeLevel) const {
   I4 Idx;
   if (getTimeIndex(Idx, TimeLevel) != 0)
      return -1;
   if (NTracers > 0)
      OMEGA::copyToHost(H, TracerArrays[Idx]);
   return 0;
}

// ---- the reference's static interface (Tracers.h), on the default store ----
namespace {
std::unique_ptr<TracerStore> &ownedDefaultStore() {
   static std::unique_ptr<TracerStore> P;
   return P;
}
TracerStore *&defaultStore() {
   static TracerStore *P = nullptr;
   return P;
}
} // namespace
TracerStore::~TracerStore() {
   if (defaultStore() == this) // never leave the static interface pointing at a store that is gone
      defaultStore() = nullptr;
}
TracerStore *Tracers::init(const HorzMesh *Mesh, Halo *MeshHalo, int NVertLayers, int NTracers, int NTimeLevels) {
   ownedDefaultStore().reset(new TracerStore(Mesh, MeshHalo, NVertLayers, NTracers, NTimeLevels));
   return defaultStore() = ownedDefaultStore().get();
}
void Tracers::setDefault(TracerStore *Store) {
   if (Store != ownedDefaultStore().get())
      ownedDefaultStore().reset();
   defaultStore() = Store;
}
TracerStore *Tracers::getDefault() { return defaultStore(); }
I4 Tracers::clear() {
   ownedDefaultStore().reset();
   defaultStore() = nullptr;
   return 0;
}
I4 Tracers::getNumTracers() { return defaultStore() ? defaultStore()->NTracers : 0; }
I4 Tracers::getTimeIndex(I4 &TimeIndex, I4 TimeLevel) {
   return defaultStore() ? defaultStore()->getTimeIndex(TimeIndex, TimeLevel) : -1;
}
I4 Tracers::getAll(Array3DReal &TracerArray, I4 TimeLevel) {
   return defaultStore() ? defaultStore()->getAll(TracerArray, TimeLevel) : -1;
}
I4 Tracers::exchangeHalo(I4 TimeLevel) {
   return defaultStore() ? defaultStore()->exchangeHalo(TimeLevel, defaultStore()->Stream) : -1;
}
void Tracers::updateTimeLevels() {
   OMEGA_REQUIRE(defaultStore(), "Tracers::updateTimeLevels: no default tracer store (Tracers::init / setDefault)");
   defaultStore()->updateTimeLevels(defaultStore()->Stream);
}

} // namespace OMEGA
