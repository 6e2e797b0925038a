// TimeStepper.h -- time stepping schemes built on Tendencies / AuxiliaryState / Halo.
// Interface after the reference (components/omega/src/timeStepping/TimeStepper.h:57-237):
// doStep, the six update kernels, and the ForwardBackward / RungeKutta2 / RungeKutta4
// schemes (ForwardBackwardStepper.cpp:27-82, RungeKutta2Stepper.cpp:27-73,
// RungeKutta4Stepper.cpp:25-137).  The clock / alarm machinery of the reference is out of
// scope: the step is a TimeInterval (TimeMgr.h), and `Real * TimeInterval` coefficients go through
// the same integer-fraction arithmetic as the reference's TimeMgr so they are bit-equal.
// Every method exists in the native form (coefficient in seconds + the HIP stream) and with the
// REFERENCE'S OWN SIGNATURE (TimeInterval coefficient / TimeInstant &SimTime, no stream: the
// object's `Stream`), so that a reference call site -- ocnRun's `Stepper->doStep(State, SimTime)`,
// a stepper subclass written like RungeKutta4Stepper.cpp:68-137 -- compiles unchanged.
#ifndef OMEGA_AMD_TIMESTEPPER_H
#define OMEGA_AMD_TIMESTEPPER_H

#include "AuxiliaryState.h"
#include "GraphCache.h"
#include "Halo.h"
#include "OceanState.h"
#include "Tendencies.h"
#include "TimeMgr.h"

namespace OMEGA {

enum class TimeStepperType { ForwardBackward, RungeKutta4, RungeKutta2, Invalid };

class TimeStepper {
 public:
   TimeStepper(const std::string &Name, TimeStepperType Type, int NTimeLevels, R8 TimeStepSeconds);
   virtual ~TimeStepper() = default;

   /// plain factory: the caller owns the object
   static TimeStepper *make(const std::string &Name, TimeStepperType Type, R8 TimeStepSeconds);
   /// the reference's registry (TimeStepper::create / get / getDefault / erase / clear, TimeStepper.h:87-139):
   /// create makes the scheme, attaches the data and finalises it; the registry owns it
   static TimeStepper *create(const std::string &Name, TimeStepperType Type, R8 TimeStepSeconds, Tendencies *Tend,
                              AuxiliaryState *AuxState, const HorzMesh *Mesh, Halo *MeshHalo, TracerStore *Trc);
   static TimeStepper *get(const std::string &Name);
   static TimeStepper *getDefault() { return get("Default"); }
   static void erase(const std::string &Name);
   static void clear();
   static TimeStepperType getFromStr(const std::string &In); ///< TimeStepper.h:64-75

   /// attach the objects the scheme works on (TimeStepper::attachData)
   /// (Trc == nullptr: the default store of the static Tracers interface, as in the reference)
   void attachData(Tendencies *Tend, AuxiliaryState *AuxState, const HorzMesh *Mesh, Halo *MeshHalo, TracerStore *Trc);
   /// after attachData: creates whatever a step needs, so that doStep allocates nothing (base: the halo's job tables
   /// and message buffers for the end-of-step exchange of h, u and the tracers)
   virtual void finalizeInit();

   /// advance State (and the attached tracers) by one step on stream S
   virtual void doStep(OceanState *State, hipStream_t S) = 0;
   /// the reference's signature (TimeStepper.h:82-84): one step from SimTime on this object's `Stream`; SimTime is
   /// advanced by TimeStep (the reference advances its StepClock and reads the time back, RungeKutta4Stepper.cpp:134-135).
   /// const as in the reference: the step counter and the launch caches a step touches are bookkeeping, not state.
   void doStep(OceanState *State, TimeInstant &SimTime) const;
   hipStream_t Stream = nullptr; ///< stream of the reference-signature methods (default: the null stream, as Kokkos')

   // update kernels (TimeStepper.cpp:378-524); Coeff is a multiple of the time step
   void updateThicknessByTend(OceanState *State1, int TimeLevel1, OceanState *State2, int TimeLevel2, R8 CoeffSeconds,
                              hipStream_t S) const;
   void updateVelocityByTend(OceanState *State1, int TimeLevel1, OceanState *State2, int TimeLevel2, R8 CoeffSeconds,
                             hipStream_t S) const;
   void updateStateByTend(OceanState *State1, int TimeLevel1, OceanState *State2, int TimeLevel2, R8 CoeffSeconds,
                          hipStream_t S) const;
   void updateTracersByTend(const Array3DReal &NextTracers, const Array3DReal &CurTracers, OceanState *State1,
                            int TimeLevel1, OceanState *State2, int TimeLevel2, R8 CoeffSeconds, hipStream_t S) const;
   void weightTracers(const Array3DReal &NextTracers, const Array3DReal &CurTracers, OceanState *CurState,
                      int TimeLevel1, hipStream_t S) const;
   void accumulateTracersUpdate(const Array3DReal &AccumTracer, R8 CoeffSeconds, hipStream_t S) const;
   void finalizeTracersUpdate(const Array3DReal &NextTracers, OceanState *State, int TimeLevel, hipStream_t S) const;
   // ... and with the reference's signatures (TimeStepper.h:174-237): Coeff.get(seconds), this object's `Stream`
   void updateStateByTend(OceanState *State1, int TimeLevel1, OceanState *State2, int TimeLevel2, TimeInterval Coeff) const {
      updateStateByTend(State1, TimeLevel1, State2, TimeLevel2, Coeff.getSeconds(), Stream);
   }
   void updateThicknessByTend(OceanState *State1, int TimeLevel1, OceanState *State2, int TimeLevel2,
                              TimeInterval Coeff) const {
      updateThicknessByTend(State1, TimeLevel1, State2, TimeLevel2, Coeff.getSeconds(), Stream);
   }
   void updateVelocityByTend(OceanState *State1, int TimeLevel1, OceanState *State2, int TimeLevel2,
                             TimeInterval Coeff) const {
      updateVelocityByTend(State1, TimeLevel1, State2, TimeLevel2, Coeff.getSeconds(), Stream);
   }
   void updateTracersByTend(const Array3DReal &NextTracers, const Array3DReal &CurTracers, OceanState *State1,
                            int TimeLevel1, OceanState *State2, int TimeLevel2, TimeInterval Coeff) const {
      updateTracersByTend(NextTracers, CurTracers, State1, TimeLevel1, State2, TimeLevel2, Coeff.getSeconds(), Stream);
   }
   void weightTracers(const Array3DReal &NextTracers, const Array3DReal &CurTracers, OceanState *CurState,
                      int TimeLevel1) const {
      weightTracers(NextTracers, CurTracers, CurState, TimeLevel1, Stream);
   }
   void accumulateTracersUpdate(const Array3DReal &AccumTracer, TimeInterval Coeff) const {
      accumulateTracersUpdate(AccumTracer, Coeff.getSeconds(), Stream);
   }
   void finalizeTracersUpdate(const Array3DReal &NextTracers, OceanState *State, int TimeLevel) const {
      finalizeTracersUpdate(NextTracers, State, TimeLevel, Stream);
   }

   /// seconds of (Mult * TimeStep), through TimeFrac arithmetic
   /// (components/omega/src/infra/TimeMgr.cpp:193-283, 747-767, 956-1000, 382-391)
   static R8 coeffSeconds(R8 Mult, R8 TimeStepSeconds);
   R8 coeff(R8 Mult) const { return (Mult * TimeStep).getSeconds(); }

   std::string Name;
   TimeStepperType Type;
   int NTimeLevels;
   TimeInterval TimeStep;  ///< (TimeStepper.h:247) `RKB[Stage] * TimeStep` is a TimeInterval, as in the reference
   R8 TimeStepSeconds;     ///< TimeStep in seconds
   TimeInterval getTimeStep() const { return TimeStep; } ///< TimeStepper.h:129
   I8 NStepsDone = 0;
   /// model time in seconds since the reference time: StartTime + NStepsDone*TimeStep.  The schemes
   /// hand the stage times to Tendencies::ModelTime (the reference passes a TimeInstant).
   R8 StartTime = 0.0;
   R8 simTime() const { return StartTime + (R8)NStepsDone * TimeStepSeconds; }
   /// TimeStepper::changeTimeStep (TimeStepper.h:141-143): the model time reached so far is kept
   void changeTimeStep(R8 NewTimeStepSeconds) {
      OMEGA_REQUIRE(NewTimeStepSeconds > 0, "TimeStepper: time step must be positive");
      StartTime  = simTime();
      NStepsDone = 0;
      TimeStepSeconds = NewTimeStepSeconds;
      TimeStep        = TimeInterval(NewTimeStepSeconds, TimeUnits::Seconds);
   }
   void changeTimeStep(const TimeInterval &NewTimeStep) { changeTimeStep(NewTimeStep.getSeconds()); } ///< TimeStepper.h:141

 protected:
   /// end-of-step: halo exchange of the new level, then rotate (State->updateTimeLevels();
   /// Tracers::updateTimeLevels()) -- h, u and tracers travel in one message per neighbour
   void updateTimeLevels(OceanState *State, hipStream_t S) const;
   /// first thing in every doStep: a peer-wire wait of an EARLIER step that gave up is reported now (the status word is
   /// host memory: no synchronisation); the halo of that step was left untouched and the state is not to be trusted
   void requireHealthyWire() const;
   Tendencies *Tend         = nullptr;
   AuxiliaryState *AuxState = nullptr;
   const HorzMesh *Mesh     = nullptr;
   Halo *MeshHalo           = nullptr;
   TracerStore *Trc         = nullptr;
};

class ForwardBackwardStepper : public TimeStepper {
 public:
   ForwardBackwardStepper(const std::string &Name, R8 Dt) : TimeStepper(Name, TimeStepperType::ForwardBackward, 2, Dt) {}
   void doStep(OceanState *State, hipStream_t S) override;
   using TimeStepper::doStep;
};

class RungeKutta2Stepper : public TimeStepper {
 public:
   RungeKutta2Stepper(const std::string &Name, R8 Dt) : TimeStepper(Name, TimeStepperType::RungeKutta2, 2, Dt) {}
   void doStep(OceanState *State, hipStream_t S) override;
   using TimeStepper::doStep;
};

class RungeKutta4Stepper : public TimeStepper {
 public:
   RungeKutta4Stepper(const std::string &Name, R8 Dt);
   void finalizeInit() override;
   void doStep(OceanState *State, hipStream_t S) override;
   using TimeStepper::doStep;

 protected:
   static constexpr int NStages = 4;
   R8 RKA[NStages], RKB[NStages], RKC[NStages];
   std::unique_ptr<OceanState> ProvisState;
   Array3DReal ProvisTracers;

 public:
   /// Fold the stage updates into the RHS kernels (kernels/Kernels.h: StageUpdate) when the fused RHS
   /// covers the mesh / options: same arithmetic, element by element, without the 19 streaming update
   /// launches per step.  The tendency arrays are then only stored if StoreStageTendencies is set.
   bool FuseStageUpdates      = true;
   bool StoreStageTendencies  = false;
   /// With fused stages and neighbours: start each halo exchange as soon as the band of cells whose values
   /// travel is final and let it run on a communication stream while the stage's interior cells are still
   /// being computed (kernels/Kernels.h: StageUpdate::AfterBand); the next consumer waits on an event.
   bool OverlapHaloExchange   = true;
   /// Without neighbours (one rank) the 28 launches of a stage-fused step are replayed as one HIP graph per
   /// time-level parity when the step runs on a non-default stream (GraphCache.h).
   bool UseGraphs             = false; ///< (or the option Graphs = 1, read at every step)
   GraphCache Graphs;
   ~RungeKutta4Stepper() override;

 protected:
   /// second provisional buffer: a stage reads one and writes the other (neighbours still gather the input)
   std::unique_ptr<OceanState> ProvisState2;
   Array3DReal ProvisTracers2;
   bool doStepFused(OceanState *State, hipStream_t S);
   bool StageFusedKnownGood = false; ///< a direct (un-captured) stage-fused step has succeeded on this configuration
   // overlapped exchange: communication stream, "band is final" and "halo is in place" events
   hipStream_t CommStream = nullptr;
   hipEvent_t EvBand = nullptr, EvDone = nullptr, EvFork = nullptr;
   void ensureCommStream();
   bool ExchangePending = false;
   struct ExchangeJob {
      RungeKutta4Stepper *Self;
      hipStream_t S;
      Array2DReal H, U;
      Array3DReal *Tr;
      int NT;
      bool Provis = false; ///< the mid-step exchange of the provisional state (timer "RK4:haloExchProvis")
   };
   static void startExchangeThunk(void *Job);
   void startExchange(const ExchangeJob &Job);
   void joinExchange(hipStream_t S);
};

} // namespace OMEGA
#endif
