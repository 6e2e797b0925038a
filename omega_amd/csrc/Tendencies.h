// Tendencies.h -- RHS tendencies of layer thickness, normal velocity and tracers.
// Interface after the reference (components/omega/src/ocn/Tendencies.h:73-144;
// Tendencies.cpp:217-600).  The `TimeInstant` argument of the reference is only forwarded
// to the custom-tendency hooks (Tendencies.cpp:288-293): here the methods take the stream
// instead, and the hooks (CustomThicknessTend / CustomVelocityTend below) receive the model
// time as `ModelTime` seconds, which the time steppers set for every stage.
#ifndef OMEGA_AMD_TENDENCIES_H
#define OMEGA_AMD_TENDENCIES_H

#include "AuxiliaryState.h"
#include "Base.h"
#include "GraphCache.h"
#include "kernels/Kernels.h"

#include <functional>

namespace OMEGA {

class Tendencies : public Registry<Tendencies> {
 public:
   Tendencies(const std::string &Name, const HorzMesh *Mesh, int NVertLayers, int NTracers, const TendParams &Options);

   Array2DReal LayerThicknessTend; ///< (NCellsSize, NVertLayers)
   Array2DReal NormalVelocityTend; ///< (NEdgesSize, NVertLayers)
   Array3DReal TracerTend;         ///< (NTracers, NCellsSize, NVertLayers)

   TendParams Params; ///< enable flags and coefficients (readTendConfig)
   /// Fused RHS for computeAllTendencies (default on); off = the reference's launch
   /// structure (every AuxiliaryState array materialised).
   bool UseFusedRHS = true;
   /// Replay the fused RHS as a HIP graph when it is called again with the same arrays on a non-default stream
   /// (GraphCache.h): one host call instead of 7 launches.  Default off (no gain measured); never used while
   /// kernel timing or custom tendencies are on.
   bool UseGraphs = false;
   GraphCache Graphs;
   /// the object's switch, or the option Graphs = 1 (read when asked, not when the object was made)
   bool graphsOn() const { return UseGraphs || GraphCache::defaultOn(); }

   /// Custom tendencies (Tendencies.h:51-53, 182-183): called at the end of the thickness / velocity
   /// group with the tendency array, the state / aux state, the two time levels and the model time
   /// (here: ModelTime seconds since the reference time instead of a TimeInstant, plus the stream).
   using CustomTendencyType = std::function<void(const Array2DReal &, const OceanState *, const AuxiliaryState *, int,
                                                 int, R8 TimeSeconds, hipStream_t)>;
   CustomTendencyType CustomThicknessTend, CustomVelocityTend;
   /// model time handed to the custom tendencies; the time steppers set it for every stage
   /// (RungeKutta4Stepper.cpp:87 StageTime, RungeKutta2Stepper.cpp:44,58, ForwardBackwardStepper.cpp:50,59,67)
   R8 ModelTime = 0.0;

   void computeThicknessTendenciesOnly(const OceanState *State, const AuxiliaryState *AuxState, int ThickTimeLevel,
                                       int VelTimeLevel, hipStream_t S);
   void computeVelocityTendenciesOnly(const OceanState *State, const AuxiliaryState *AuxState, int ThickTimeLevel,
                                      int VelTimeLevel, hipStream_t S);
   void computeTracerTendenciesOnly(const OceanState *State, const AuxiliaryState *AuxState,
                                    const Array3DReal &TracerArray, int ThickTimeLevel, int VelTimeLevel,
                                    hipStream_t S);
   void computeThicknessTendencies(const OceanState *State, const AuxiliaryState *AuxState, int ThickTimeLevel,
                                   int VelTimeLevel, hipStream_t S);
   void computeVelocityTendencies(const OceanState *State, const AuxiliaryState *AuxState, int ThickTimeLevel,
                                  int VelTimeLevel, hipStream_t S);
   void computeTracerTendencies(const OceanState *State, const AuxiliaryState *AuxState,
                                const Array3DReal &TracerArray, int ThickTimeLevel, int VelTimeLevel, hipStream_t S);
   /// computeAllTendencies with a Runge-Kutta stage update folded into the kernels that produce the
   /// tendencies (kernels/Kernels.h: StageUpdate).  Returns false -- nothing launched -- when the
   /// stage-fused kernels do not cover this mesh / option set; the caller then uses the plain sequence.
   bool computeAllTendenciesStage(const OceanState *State, const AuxiliaryState *AuxState, const Array3DReal &TracerArray,
                                  int ThickTimeLevel, int VelTimeLevel, const StageUpdate &Stage, hipStream_t S);
   void computeAllTendencies(const OceanState *State, const AuxiliaryState *AuxState, const Array3DReal &TracerArray,
                             int ThickTimeLevel, int VelTimeLevel, hipStream_t S);

   /// Per-kernel timing of the fused RHS with HIP events on the launch stream (bench.py's
   /// roofline leg): while enabled every computeAllTendencies call records 7 events.
   void enableKernelTiming(bool On);
   /// Sum of the recorded durations per kernel [FusedNumKernels] in ms and the number of
   /// recorded RHS evaluations; synchronises, then clears the recordings.
   int collectKernelTimes(double *MsSum);
   ~Tendencies();

   const HorzMesh *Mesh;
   int NVertLayers, NTracers;
   /// The fused RHS never materialises the edge-located auxiliary arrays; custom tendency hooks receive the
   /// AuxiliaryState and may read it (reference: computeAllTendencies computes the auxiliary state first,
   /// Tendencies.cpp:591): with this on (default) AuxiliaryState::computeAll runs before the hooks are called.
   bool MaterialiseAuxForCustom = true;

 private:
   Array2DReal EdgeScratch; ///< running PV sums of the fused RHS (allocated by the constructor)
   bool TimingOn = false;
   bool WarnedUnfused = false;
   std::vector<std::vector<hipEvent_t>> TimingEvents;
   /// AuxiliaryState options are read by AuxiliaryState::readConfigOptions in the reference;
   /// the kernels take them through TendParams, so sync them from the AuxState in use.
   TendParams paramsFor(const AuxiliaryState *AuxState) const;
};

} // namespace OMEGA
#endif
