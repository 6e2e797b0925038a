// Tendencies.h -- RHS tendencies of layer thickness, normal velocity and tracers.
// Interface after the reference (components/omega/src/ocn/Tendencies.h:73-144;
// Tendencies.cpp:217-600).  Every compute method exists in two forms: the native one takes the HIP
// stream (the `TimeInstant` of the reference is only forwarded to the custom-tendency hooks,
// Tendencies.cpp:288-293: here `ModelTime` seconds, set by the time steppers for every stage), and
// the REFERENCE'S OWN SIGNATURE (..., TimeInstant Time), which sets ModelTime from the instant and
// runs on the object's `Stream` -- a reference call site compiles unchanged.
#ifndef OMEGA_AMD_TENDENCIES_H
#define OMEGA_AMD_TENDENCIES_H

#include "AuxiliaryState.h"
#include "Base.h"
#include "GraphCache.h"
#include "TimeMgr.h"
#include "kernels/Kernels.h"

#include <functional>
#include <type_traits>

namespace OMEGA {

/// A custom tendency hook (Tendencies.h:51-53): holds a callable of EITHER form --
///   native:     void(const Array2DReal &Tend, const OceanState *, const AuxiliaryState *, int ThickLvl, int VelLvl,
///                    R8 ModelTimeSeconds, hipStream_t S)                  (launch on S)
///   reference:  void(Array2DReal Tend, const OceanState *, const AuxiliaryState *, int ThickLvl, int VelLvl, TimeInstant Time)
/// A reference-form hook knows no stream: its work goes to the null stream as in the reference; when the tendencies run
/// on another stream the adapter orders the two by host synchronisation before and after the call.
class CustomTendencyType {
 public:
   using Native = std::function<void(const Array2DReal &, const OceanState *, const AuxiliaryState *, int, int, R8, hipStream_t)>;
   CustomTendencyType() = default;
   CustomTendencyType(std::nullptr_t) {}
   template <class C, std::enable_if_t<std::is_invocable_v<C &, const Array2DReal &, const OceanState *, const AuxiliaryState *,
                                                           int, int, R8, hipStream_t>, int> = 0>
   CustomTendencyType(C F_) : F(std::move(F_)) {}
   template <class C, std::enable_if_t<!std::is_invocable_v<C &, const Array2DReal &, const OceanState *, const AuxiliaryState *,
                                                            int, int, R8, hipStream_t> &&
                                           std::is_invocable_v<C &, Array2DReal, const OceanState *, const AuxiliaryState *, int,
                                                               int, TimeInstant>, int> = 0>
   CustomTendencyType(C G)
       : F([G](const Array2DReal &Tend, const OceanState *St, const AuxiliaryState *Aux, int ThickLvl, int VelLvl, R8 Seconds,
               hipStream_t S) mutable {
            if (S)
               HIP_CHECK(hipStreamSynchronize(S));
            G(Tend, St, Aux, ThickLvl, VelLvl, TimeInstant::fromSeconds(Seconds));
            if (S)
               HIP_CHECK(hipStreamSynchronize(nullptr));
         }) {}
   explicit operator bool() const { return (bool)F; }
   void operator()(const Array2DReal &Tend, const OceanState *St, const AuxiliaryState *Aux, int ThickLvl, int VelLvl, R8 Seconds,
                   hipStream_t S) const {
      F(Tend, St, Aux, ThickLvl, VelLvl, Seconds, S);
   }

 private:
   Native F;
};

class Tendencies : public Registry<Tendencies> {
 public:
   /// Fails (OmegaError naming the limit) when the mesh is outside the fused RHS -- an array plane of 4 GiB or more
   /// (32-bit buffer offsets: ~ 2.2 M cells x 80 levels PER RANK; several ranks may share a GPU) or MaxEdges outside
   /// 5..8 -- unless AllowReferenceStructured: then computeAllTendencies takes the reference-structured 23-launch path
   /// (~ 5 x the time) for good, a decision of the caller's instead of a surprise.
   Tendencies(const std::string &Name, const HorzMesh *Mesh, int NVertLayers, int NTracers, const TendParams &Options,
              bool AllowReferenceStructured = false);
   /// the test behind it, on sizes alone (no mesh, no device): "" if the fused RHS covers them, else the reason
   static std::string fusedLimit(size_t NCellsSize, size_t NEdgesSize, size_t NVerticesSize, int MaxEdges, int NVertLayers);

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
   /// (native form: ModelTime seconds + the stream; or the reference's form with a TimeInstant).
   using CustomTendencyType = OMEGA::CustomTendencyType;
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

   // ---- the reference's signatures (Tendencies.h:73-102): ModelTime <- Time, launches on `Stream`
   hipStream_t Stream = nullptr; ///< stream of the reference-signature methods (default: the null stream, as Kokkos')
   void computeThicknessTendencies(const OceanState *State, const AuxiliaryState *AuxState, int ThickTimeLevel,
                                   int VelTimeLevel, TimeInstant Time) {
      ModelTime = Time.getSeconds();
      computeThicknessTendencies(State, AuxState, ThickTimeLevel, VelTimeLevel, Stream);
   }
   void computeVelocityTendencies(const OceanState *State, const AuxiliaryState *AuxState, int ThickTimeLevel,
                                  int VelTimeLevel, TimeInstant Time) {
      ModelTime = Time.getSeconds();
      computeVelocityTendencies(State, AuxState, ThickTimeLevel, VelTimeLevel, Stream);
   }
   void computeTracerTendencies(const OceanState *State, const AuxiliaryState *AuxState, const Array3DReal &TracerArray,
                                int ThickTimeLevel, int VelTimeLevel, TimeInstant Time) {
      ModelTime = Time.getSeconds();
      computeTracerTendencies(State, AuxState, TracerArray, ThickTimeLevel, VelTimeLevel, Stream);
   }
   void computeAllTendencies(const OceanState *State, const AuxiliaryState *AuxState, const Array3DReal &TracerArray,
                             int ThickTimeLevel, int VelTimeLevel, TimeInstant Time) {
      ModelTime = Time.getSeconds();
      computeAllTendencies(State, AuxState, TracerArray, ThickTimeLevel, VelTimeLevel, Stream);
   }
   void computeThicknessTendenciesOnly(const OceanState *State, const AuxiliaryState *AuxState, int ThickTimeLevel,
                                       int VelTimeLevel, TimeInstant Time) {
      ModelTime = Time.getSeconds();
      computeThicknessTendenciesOnly(State, AuxState, ThickTimeLevel, VelTimeLevel, Stream);
   }
   void computeVelocityTendenciesOnly(const OceanState *State, const AuxiliaryState *AuxState, int ThickTimeLevel,
                                      int VelTimeLevel, TimeInstant Time) {
      ModelTime = Time.getSeconds();
      computeVelocityTendenciesOnly(State, AuxState, ThickTimeLevel, VelTimeLevel, Stream);
   }
   void computeTracerTendenciesOnly(const OceanState *State, const AuxiliaryState *AuxState, const Array3DReal &TracerArray,
                                    int ThickTimeLevel, int VelTimeLevel, TimeInstant Time) {
      ModelTime = Time.getSeconds();
      computeTracerTendenciesOnly(State, AuxState, TracerArray, ThickTimeLevel, VelTimeLevel, Stream);
   }

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
   std::vector<std::vector<hipEvent_t>> TimingEvents;
   /// AuxiliaryState options are read by AuxiliaryState::readConfigOptions in the reference;
   /// the kernels take them through TendParams, so sync them from the AuxState in use.
   TendParams paramsFor(const AuxiliaryState *AuxState) const;
};

} // namespace OMEGA
#endif
