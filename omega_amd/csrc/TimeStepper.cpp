// TimeStepper.cpp -- see TimeStepper.h.
#include "TimeStepper.h"
#include "Pacer.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>

namespace OMEGA {

// ---- TimeFrac arithmetic of the reference's TimeMgr (TimeMgr.h) ----
namespace {
I8 fracGCD(I8 A, I8 B) {
   A = std::llabs(A);
   B = std::llabs(B);
   if (A == 0)
      return B ? B : 1;
   if (B == 0)
      return A;
   while (B) {
      I8 T = A % B;
      A    = B;
      B    = T;
   }
   return A;
}
} // namespace
// TimeFrac::simplify (TimeMgr.cpp:956-1000)
void TimeFrac::simplify() {
   OMEGA_REQUIRE(Denom != 0, "TimeFrac: zero denominator");
   I8 W;
   if (std::llabs((W = Numer / Denom)) >= 1) {
      Whole += W;
      Numer %= Denom;
   }
   if (Whole > 0 && ((Numer < 0 && Denom > 0) || (Denom < 0 && Numer > 0))) {
      Whole--;
      Numer += Denom;
   } else if ((Whole < 0 && (Numer > 0 && Denom > 0)) || (Denom < 0 && Numer < 0)) {
      Whole++;
      Numer -= Denom;
   }
   if (Denom < 0) {
      Denom *= -1;
      Numer *= -1;
   }
   const I8 G = fracGCD(Numer, Denom);
   Numer /= G;
   Denom /= G;
}
// TimeFrac::setSeconds (TimeMgr.cpp:193-283): continued-fraction conversion
TimeFrac TimeFrac::fromSeconds(R8 Seconds) {
   TimeFrac F;
   const R8 Rabs = std::fabs(Seconds);
   OMEGA_REQUIRE(!((Rabs > 0.0 && Rabs < 1e-17) || Rabs > 1e18), "TimeStepper: time value out of range");
   const int Sign = (Seconds < 0) ? -1 : 1;
   R8 Target      = Rabs;
   if (Target == 0.0)
      return F;
   if (Target >= 1.0) {
      const I8 W = (I8)Rabs;
      Target -= (R8)W;
      F.Whole = Sign * W;
      if (Target < 1e-17)
         return F;
   }
   const R8 P = std::pow(10.0, -(DBL_DIG - (int)std::log10(Rabs)));
   R8 R       = Target;
   I8 Npp = 0, Np = 1, Dpp = 1, Dp = 0, A, N, D;
   for (;;) {
      A = (I8)R;
      N = A * Np + Npp;
      D = A * Dp + Dpp;
      if (std::fabs((R8)N / (R8)D - Target) < P)
         break;
      const R8 Fr = R - (R8)A;
      if (Fr < 1e-17)
         break;
      R   = 1.0 / Fr;
      Npp = Np;
      Np  = N;
      Dpp = Dp;
      Dp  = D;
   }
   F.Numer = N * Sign;
   F.Denom = D;
   F.simplify();
   return F;
}
// TimeFrac::operator+ / operator- (TimeMgr.cpp:625-679): over the least common denominator
TimeFrac TimeFrac::operator+(const TimeFrac &O) const {
   TimeFrac S;
   S.Denom = Denom / fracGCD(Denom, O.Denom) * O.Denom;
   S.Numer = Numer * (S.Denom / Denom) + O.Numer * (S.Denom / O.Denom);
   S.Whole = Whole + O.Whole;
   S.simplify();
   return S;
}
TimeFrac TimeFrac::operator-(const TimeFrac &O) const {
   TimeFrac S;
   S.Denom = Denom / fracGCD(Denom, O.Denom) * O.Denom;
   S.Numer = Numer * (S.Denom / Denom) - O.Numer * (S.Denom / O.Denom);
   S.Whole = Whole - O.Whole;
   S.simplify();
   return S;
}
// TimeFrac::operator*(R8) (TimeMgr.cpp:747-767)
TimeFrac TimeFrac::operator*(R8 Multiplier) const {
   const TimeFrac M = fromSeconds(Multiplier);
   TimeFrac P;
   P.Denom = Denom * M.Denom;
   P.Numer = (Whole * Denom + Numer) * (M.Whole * M.Denom + M.Numer);
   P.simplify();
   return P;
}
TimeFrac TimeFrac::operator*(I4 Multiplier) const {
   TimeFrac P;
   P.Whole = Whole * Multiplier;
   P.Numer = Numer * Multiplier;
   P.Denom = Denom;
   P.simplify();
   return P;
}
void TimeInterval::set(R8 Length, TimeUnits Units) {
   OMEGA_REQUIRE(Units == TimeUnits::Seconds || Units == TimeUnits::Minutes || Units == TimeUnits::Hours,
                 "TimeInterval: only non-calendar units (seconds, minutes, hours) are supported");
   Interval = TimeFrac::fromSeconds(Length);
   if (Units == TimeUnits::Minutes)
      Interval = Interval * (I4)60;
   else if (Units == TimeUnits::Hours)
      Interval = Interval * (I4)3600;
}
void TimeInterval::get(R8 &Length, TimeUnits Units) const {
   OMEGA_REQUIRE(Units == TimeUnits::Seconds || Units == TimeUnits::Minutes || Units == TimeUnits::Hours,
                 "TimeInterval: only non-calendar units (seconds, minutes, hours) are supported");
   Length = Interval.getSeconds();
   if (Units == TimeUnits::Minutes)
      Length /= 60.0;
   else if (Units == TimeUnits::Hours)
      Length /= 3600.0;
}

R8 TimeStepper::coeffSeconds(R8 Mult, R8 TimeStepSeconds) {
   // (Real * TimeInterval, then TimeInterval::get(seconds): TimeStepper.cpp:392-393)
   return (TimeFrac::fromSeconds(TimeStepSeconds) * Mult).getSeconds();
}

TimeStepper::TimeStepper(const std::string &Name_, TimeStepperType Type_, int NTimeLevels_, R8 Dt)
    : Name(Name_), Type(Type_), NTimeLevels(NTimeLevels_), TimeStep(Dt, TimeUnits::Seconds), TimeStepSeconds(Dt) {
   OMEGA_REQUIRE(Dt > 0, "TimeStepper: time step must be positive");
}

// TimeStepper.h:82-84.  The model time of the stages is SimTime + RKC * TimeStep in the reference; here the schemes
// derive it from StartTime + NStepsDone * TimeStep, so the step is anchored at SimTime first.
void TimeStepper::doStep(OceanState *State, TimeInstant &SimTime) const {
   TimeStepper *Self = const_cast<TimeStepper *>(this);
   Self->StartTime   = SimTime.getSeconds();
   Self->NStepsDone  = 0;
   Self->doStep(State, Stream);
   SimTime += TimeStep;
}

TimeStepperType TimeStepper::getFromStr(const std::string &In) {
   if (In == "Forward-Backward")
      return TimeStepperType::ForwardBackward;
   if (In == "RungeKutta4")
      return TimeStepperType::RungeKutta4;
   if (In == "RungeKutta2")
      return TimeStepperType::RungeKutta2;
   return TimeStepperType::Invalid;
}

namespace {
std::map<std::string, std::unique_ptr<TimeStepper>> &allSteppers() {
   static std::map<std::string, std::unique_ptr<TimeStepper>> M;
   return M;
}
} // namespace
TimeStepper *TimeStepper::create(const std::string &Name, TimeStepperType Type, R8 Dt, Tendencies *T, AuxiliaryState *A,
                                 const HorzMesh *M, Halo *H, TracerStore *Tr) {
   auto &All = allSteppers();
   if (All.find(Name) != All.end())
      return nullptr;
   TimeStepper *St = make(Name, Type, Dt);
   All[Name].reset(St);
   St->attachData(T, A, M, H, Tr);
   St->finalizeInit();
   return St;
}
TimeStepper *TimeStepper::get(const std::string &Name) {
   auto It = allSteppers().find(Name);
   return It == allSteppers().end() ? nullptr : It->second.get();
}
void TimeStepper::erase(const std::string &Name) { allSteppers().erase(Name); }
void TimeStepper::clear() { allSteppers().clear(); }

TimeStepper *TimeStepper::make(const std::string &Name, TimeStepperType Type, R8 Dt) {
   switch (Type) {
   case TimeStepperType::ForwardBackward:
      return new ForwardBackwardStepper(Name, Dt);
   case TimeStepperType::RungeKutta4:
      return new RungeKutta4Stepper(Name, Dt);
   case TimeStepperType::RungeKutta2:
      return new RungeKutta2Stepper(Name, Dt);
   default:
      OMEGA_ABORT("TimeStepper::make: unknown time stepper type");
   }
}

void TimeStepper::attachData(Tendencies *T, AuxiliaryState *A, const HorzMesh *M, Halo *H, TracerStore *Tr) {
   Tend     = T;
   AuxState = A;
   Mesh     = M;
   MeshHalo = H;
   Trc      = Tr ? Tr : Tracers::getDefault();
}

void TimeStepper::finalizeInit() {
   if (MeshHalo && MeshHalo->NNghbr > 0 && Tend)
      MeshHalo->reserveState(Tend->LayerThicknessTend, Tend->NormalVelocityTend, Tend->NTracers > 0 ? &Tend->TracerTend : nullptr,
                             Tend->NTracers);
}

// ---- update kernels ----
void TimeStepper::updateThicknessByTend(OceanState *S1, int L1, OceanState *S2, int L2, R8 C, hipStream_t S) const {
   Array2DReal H1, H2;
   OMEGA_REQUIRE(S1->getLayerThickness(H1, L1) == 0 && S2->getLayerThickness(H2, L2) == 0,
                 "TimeStepper updateThickness: error retrieving layer thick");
   launchUpdateByTend(Mesh->NCellsAll, H1.Pitch, H1.Ptr, H2.Ptr, Tend->LayerThicknessTend.Ptr, C, S);
}
void TimeStepper::updateVelocityByTend(OceanState *S1, int L1, OceanState *S2, int L2, R8 C, hipStream_t S) const {
   Array2DReal U1, U2;
   OMEGA_REQUIRE(S1->getNormalVelocity(U1, L1) == 0 && S2->getNormalVelocity(U2, L2) == 0,
                 "TimeStepper updateVelocity: error retrieving velocity");
   launchUpdateByTend(Mesh->NEdgesAll, U1.Pitch, U1.Ptr, U2.Ptr, Tend->NormalVelocityTend.Ptr, C, S);
}
void TimeStepper::updateStateByTend(OceanState *S1, int L1, OceanState *S2, int L2, R8 C, hipStream_t S) const {
   updateThicknessByTend(S1, L1, S2, L2, C, S);
   updateVelocityByTend(S1, L1, S2, L2, C, S);
}
void TimeStepper::updateTracersByTend(const Array3DReal &Next, const Array3DReal &Cur, OceanState *S1, int L1,
                                      OceanState *S2, int L2, R8 C, hipStream_t S) const {
   Array2DReal H1, H2;
   OMEGA_REQUIRE(S1->getLayerThickness(H1, L1) == 0 && S2->getLayerThickness(H2, L2) == 0,
                 "TimeStepper updateTracers: error retrieving layer thick");
   launchUpdateTracersByTend(Trc ? Trc->NTracers : 0, Mesh->NCellsAll, Mesh->NCellsSize, H1.Pitch, Next.Ptr, Cur.Ptr,
                             H1.Ptr, H2.Ptr, Tend->TracerTend.Ptr, C, S);
}
void TimeStepper::weightTracers(const Array3DReal &Next, const Array3DReal &Cur, OceanState *St, int L1,
                                hipStream_t S) const {
   Array2DReal H;
   OMEGA_REQUIRE(St->getLayerThickness(H, L1) == 0, "TimeStepper weightTracers: bad time level");
   launchWeightTracers(Trc ? Trc->NTracers : 0, Mesh->NCellsAll, Mesh->NCellsSize, H.Pitch, Next.Ptr, Cur.Ptr, H.Ptr,
                       S);
}
void TimeStepper::accumulateTracersUpdate(const Array3DReal &Accum, R8 C, hipStream_t S) const {
   launchAccumulateTracers(Trc ? Trc->NTracers : 0, Mesh->NCellsAll, Mesh->NCellsSize, Accum.Pitch, Accum.Ptr,
                           Tend->TracerTend.Ptr, C, S);
}
void TimeStepper::finalizeTracersUpdate(const Array3DReal &Next, OceanState *St, int L, hipStream_t S) const {
   Array2DReal H;
   OMEGA_REQUIRE(St->getLayerThickness(H, L) == 0, "TimeStepper finalizeTracers: bad time level");
   launchFinalizeTracers(Trc ? Trc->NTracers : 0, Mesh->NCellsAll, Mesh->NCellsSize, H.Pitch, Next.Ptr, H.Ptr, S);
}

void TimeStepper::requireHealthyWire() const {
   OMEGA_REQUIRE(!MeshHalo || MeshHalo->checkWire() == 0,
                 "TimeStepper: a halo exchange of an earlier step failed" + MeshHalo->wireError());
}

void TimeStepper::updateTimeLevels(OceanState *State, hipStream_t S) const {
   if (MeshHalo && MeshHalo->NNghbr > 0) {
      Array2DReal H, U;
      Array3DReal Tr;
      State->getLayerThickness(H, 1);
      State->getNormalVelocity(U, 1);
      const int NT = Trc ? Trc->NTracers : 0;
      if (NT > 0)
         Trc->getAll(Tr, 1);
      // (the reference's name for this exchange: "RK4:haloExch" / "RK2:haloExch" / "ForwardBackward:haloExch", level 3)
      Pacer::Range Timer(Type == TimeStepperType::RungeKutta4   ? "RK4:haloExch"
                         : Type == TimeStepperType::RungeKutta2 ? "RK2:haloExch"
                                                                : "ForwardBackward:haloExch",
                         3);
      OMEGA_REQUIRE(MeshHalo->exchangeState(H, U, NT > 0 ? &Tr : nullptr, NT, S) == 0,
                    "TimeStepper: halo exchange failed" + MeshHalo->wireError());
   }
   State->rotateTimeLevels();
   if (Trc)
      Trc->rotateTimeLevels();
}

// ---- ForwardBackwardStepper::doStep (ForwardBackwardStepper.cpp:27-82) ----
void ForwardBackwardStepper::doStep(OceanState *State, hipStream_t S) {
   requireHealthyWire();
   const int CurLevel = 0, NextLevel = 1;
   Array3DReal CurTracerArray, NextTracerArray;
   OMEGA_REQUIRE(Trc->getAll(CurTracerArray, CurLevel) == 0 && Trc->getAll(NextTracerArray, NextLevel) == 0,
                 "ForwardBackward doStep: error retrieving tracers");
   const R8 Dt = coeff(1.0);
   const R8 T0 = simTime();
   // R_h^{n} = RHS_h(u^{n}, h^{n}, t^{n});  h^{n+1} = h^{n} + R_h^{n}
   Tend->ModelTime = T0;
   Tend->computeThicknessTendencies(State, AuxState, CurLevel, CurLevel, S);
   updateThicknessByTend(State, NextLevel, State, CurLevel, Dt, S);
   // R_phi^{n};  phi^{n+1} = (phi^{n} * h^{n} + R_phi^{n}) / h^{n+1}
   Tend->computeTracerTendencies(State, AuxState, CurTracerArray, CurLevel, CurLevel, S);
   updateTracersByTend(NextTracerArray, CurTracerArray, State, NextLevel, State, CurLevel, Dt, S);
   // R_u^{n+1} = RHS_u(u^{n}, h^{n+1}, t^{n+1});  u^{n+1} = u^{n} + R_u^{n+1}
   Tend->ModelTime = T0 + Dt;
   Tend->computeVelocityTendencies(State, AuxState, NextLevel, CurLevel, S);
   updateVelocityByTend(State, NextLevel, State, CurLevel, Dt, S);
   updateTimeLevels(State, S);
   ++NStepsDone;
}

// ---- RungeKutta2Stepper::doStep (RungeKutta2Stepper.cpp:27-73) ----
void RungeKutta2Stepper::doStep(OceanState *State, hipStream_t S) {
   requireHealthyWire();
   const int CurLevel = 0, NextLevel = 1;
   Array3DReal CurTracerArray, NextTracerArray;
   OMEGA_REQUIRE(Trc->getAll(CurTracerArray, CurLevel) == 0 && Trc->getAll(NextTracerArray, NextLevel) == 0,
                 "RungeKutta2 doStep: error retrieving tracers");
   const R8 Half = coeff(0.5), Full = coeff(1.0);
   const R8 T0 = simTime();
   Tend->ModelTime = T0;
   Tend->computeAllTendencies(State, AuxState, CurTracerArray, CurLevel, CurLevel, S);
   updateStateByTend(State, NextLevel, State, CurLevel, Half, S);
   updateTracersByTend(NextTracerArray, CurTracerArray, State, NextLevel, State, CurLevel, Half, S);
   Tend->ModelTime = T0 + Half;
   Tend->computeAllTendencies(State, AuxState, NextTracerArray, NextLevel, NextLevel, S);
   updateStateByTend(State, NextLevel, State, CurLevel, Full, S);
   updateTracersByTend(NextTracerArray, CurTracerArray, State, NextLevel, State, CurLevel, Full, S);
   updateTimeLevels(State, S);
   ++NStepsDone;
}

// ---- RungeKutta4Stepper (RungeKutta4Stepper.cpp:17-137) ----
RungeKutta4Stepper::RungeKutta4Stepper(const std::string &Name, R8 Dt)
    : TimeStepper(Name, TimeStepperType::RungeKutta4, 2, Dt) {
   RKA[0] = 0, RKA[1] = 1. / 2, RKA[2] = 1. / 2, RKA[3] = 1;
   RKB[0] = 1. / 6, RKB[1] = 1. / 3, RKB[2] = 1. / 3, RKB[3] = 1. / 6;
   RKC[0] = 0, RKC[1] = 1. / 2, RKC[2] = 1. / 2, RKC[3] = 1;
}

void RungeKutta4Stepper::finalizeInit() {
   OMEGA_REQUIRE(Tend && Mesh && Trc, "RungeKutta4Stepper: attachData before finalizeInit");
   const int K = Tend->LayerThicknessTend.Ext[1];
   const int NT = Trc->NTracers;
   ProvisState.reset(new OceanState("Provis" + Name, Mesh, MeshHalo, K, 1)); // 1 time level (:56-60)
   ProvisTracers = Array3DReal::levels("ProvisTracers", NT > 0 ? NT : 1, Mesh->NCellsSize, K);
   // Everything a step needs is created here, as the reference does (RungeKutta4Stepper.cpp:43-64), never inside doStep:
   // the second provisional buffer of the stage-fused form, and with neighbours the communication stream, its events and
   // the halo's job tables and message buffers for the state exchange (h + u + tracers in one message per neighbour).
   ProvisState2.reset(new OceanState("Provis2" + Name, Mesh, MeshHalo, K, 1));
   ProvisTracers2 = Array3DReal::levels("ProvisTracers2", NT > 0 ? NT : 1, Mesh->NCellsSize, K);
   if (MeshHalo && MeshHalo->NNghbr > 0)
      ensureCommStream();
   TimeStepper::finalizeInit();
}

// The same scheme with every stage's updates applied in the epilogue of the kernels that produce
// the tendencies.  Stage s computes R = RHS(q_in) and, element by element,
//    q^{n+1} (+)= RKB[s]*dt*R          (first stage: = q^n + ..., tracers thickness-weighted)
//    q_out     = q^n + RKA[s+1]*dt*R   (the next stage's input; tracers divided by the new thickness)
// which is what weightTracers / updateStateByTend / accumulateTracersUpdate / updateTracersByTend /
// finalizeTracersUpdate do in separate sweeps.  q_in and q_out alternate between two buffers.
RungeKutta4Stepper::~RungeKutta4Stepper() {
   if (EvBand)
      (void)hipEventDestroy(EvBand);
   if (EvDone)
      (void)hipEventDestroy(EvDone);
   if (EvFork)
      (void)hipEventDestroy(EvFork);
   if (CommStream)
      (void)hipStreamDestroy(CommStream);
}

void RungeKutta4Stepper::startExchangeThunk(void *Job) {
   auto *J = static_cast<ExchangeJob *>(Job);
   J->Self->startExchange(*J);
}
// Called by the RHS launcher between the band and the interior part of a stage: everything a neighbour
// receives is final on stream S.  Pack, send / receive and unpack run on the communication stream.
void RungeKutta4Stepper::ensureCommStream() {
   if (CommStream)
      return;
   // the highest priority the device offers: the band launches and the pack / unpack kernels on this stream are
   // small and everything else waits for them, the interior launch next to them fills the GPU for much longer
   int Least = 0, Greatest = 0;
   HIP_CHECK(hipDeviceGetStreamPriorityRange(&Least, &Greatest));
   HIP_CHECK(hipStreamCreateWithPriority(&CommStream, hipStreamNonBlocking, Greatest));
   HIP_CHECK(hipEventCreateWithFlags(&EvBand, hipEventDisableTiming));
   HIP_CHECK(hipEventCreateWithFlags(&EvDone, hipEventDisableTiming));
   HIP_CHECK(hipEventCreateWithFlags(&EvFork, hipEventDisableTiming));
   noteDeviceResource(4);
}
void RungeKutta4Stepper::startExchange(const ExchangeJob &Job) {
   ensureCommStream();
   Pacer::Range Timer(Job.Provis ? "RK4:haloExchProvis" : "RK4:haloExch", 3);
   HIP_CHECK(hipEventRecord(EvBand, Job.S));
   HIP_CHECK(hipStreamWaitEvent(CommStream, EvBand, 0));
   OMEGA_REQUIRE(MeshHalo->exchangeState(Job.H, Job.U, Job.NT > 0 ? Job.Tr : nullptr, Job.NT, CommStream) == 0,
                 "RungeKutta4: overlapped halo exchange failed" + MeshHalo->wireError());
   HIP_CHECK(hipEventRecord(EvDone, CommStream));
   ExchangePending = true;
}
void RungeKutta4Stepper::joinExchange(hipStream_t S) {
   if (ExchangePending)
      HIP_CHECK(hipStreamWaitEvent(S, EvDone, 0));
   ExchangePending = false;
}

bool RungeKutta4Stepper::doStepFused(OceanState *State, hipStream_t S) {
   const int CurLevel = 0, NextLevel = 1;
   const int NT = Trc->NTracers;
   const int K  = Tend->LayerThicknessTend.Ext[1];
   (void)K;
   Array3DReal NextTr, CurTr;
   Array2DReal CurH, CurU, NextH, NextU;
   OMEGA_REQUIRE(Trc->getAll(CurTr, CurLevel) == 0 && Trc->getAll(NextTr, NextLevel) == 0,
                 "RungeKutta4 doStep: error retrieving tracers");
   State->getLayerThickness(CurH, CurLevel), State->getNormalVelocity(CurU, CurLevel);
   State->getLayerThickness(NextH, NextLevel), State->getNormalVelocity(NextU, NextLevel);
   OceanState *Prov[2]   = {ProvisState.get(), ProvisState2.get()};
   Array3DReal *ProvT[2] = {&ProvisTracers, &ProvisTracers2};
   const bool Exchanges  = MeshHalo && MeshHalo->NNghbr > 0;
   const bool Overlap    = Exchanges && OverlapHaloExchange;
   ExchangeJob Job{this, S, {}, {}, nullptr, NT};
   bool FirstStageOk = true;
   auto RunStages    = [&]() {
   for (int Stage = 0; Stage < NStages; ++Stage) {
      StageUpdate Su;
      Su.CB        = coeff(RKB[Stage]);
      Su.CA        = Stage + 1 < NStages ? coeff(RKA[Stage + 1]) : 0.0;
      Su.First     = Stage == 0;
      Su.Last      = Stage == NStages - 1;
      Su.StoreTend = StoreStageTendencies ? 1 : 0;
      Su.NextH = NextH.Ptr, Su.NextU = NextU.Ptr, Su.NextTr = NextTr.Ptr;
      Su.CurH = CurH.Ptr, Su.CurU = CurU.Ptr, Su.CurTr = CurTr.Ptr;
      OceanState *Out = Prov[Stage % 2];
      Array2DReal OutH, OutU;
      Out->getLayerThickness(OutH, CurLevel), Out->getNormalVelocity(OutU, CurLevel);
      Su.ProvH = OutH.Ptr, Su.ProvU = OutU.Ptr, Su.ProvTr = ProvT[Stage % 2]->Ptr;
      // How far the sweeps of this stage have to go (halo layers are prefixes of the local numbering).  An evaluation
      // reaches two cells far (the del4 terms), so with the input valid on every layer:
      //  * a stage whose output is exchanged at once (overlapped: stage 1, last) is read on owned elements only: level 3
      //    runs on the send band + interior (below), level 1 through layer 2 (level 2 keeps its full sweeps);
      //  * the stage before it (0, 2) feeds that evaluation: tracers through layer 2, and every edge of those cells --
      //    finished in the thread of the edge's second cell -- through layer 3.  Only at HaloWidth >= 4, where these
      //    layers are valid at all; at the reference's default 3 the outer layers' values enter the next evaluation as
      //    they are (RungeKutta4Stepper.cpp:107 "depends on halo width"), so nothing is left out there.
      const int HaloW = (int)Mesh->NCellsHaloH.size();
      if (Exchanges && HaloW >= 4 && (Stage == 0 || Stage == 2))
         Su.NCellsTr = Mesh->NCellsHaloH(1), Su.NCellsVel = Mesh->NCellsHaloH(2);
      if (Overlap && (Stage == 1 || Stage == NStages - 1)) {
         // this stage's output is exchanged next: the provisional state before stage 2 (:107-113), the
         // new state at the end of the step (:130-131)
         if (Stage == 1)
            Job.H = OutH, Job.U = OutU, Job.Tr = ProvT[Stage % 2], Job.Provis = true;
         else
            Job.H = NextH, Job.U = NextU, Job.Tr = &NextTr, Job.Provis = false;
         Su.AfterBand = &RungeKutta4Stepper::startExchangeThunk, Su.AfterBandCtx = &Job;
         Su.HaloOutputsReplaced = 1; // Prov (stage 1) / Next (last stage): every halo element arrives with the exchange
         if (HaloW >= 3)
            Su.NCellsL1 = Mesh->NCellsHaloH(1);
         ensureCommStream();
         Su.BandStream = CommStream, Su.BandReady = EvFork; // the band launches go where the exchange follows them
      }
      bool Ok;
      if (Stage == 0) {
         Ok = Tend->computeAllTendenciesStage(State, AuxState, CurTr, CurLevel, CurLevel, Su, S);
         if (!Ok) {
            FirstStageOk = false; // nothing has been touched: the caller runs the plain sequence
            return;
         }
      } else {
         OceanState *In = Prov[(Stage - 1) % 2];
         if (Stage == 2 && Exchanges) { // depends on the halo width (:107-113)
            if (Overlap) {
               joinExchange(S); // started by stage 1 when its band was final
            } else {
               Array2DReal H, U;
               In->getLayerThickness(H, CurLevel), In->getNormalVelocity(U, CurLevel);
               Pacer::Range Timer("RK4:haloExchProvis", 3);
               OMEGA_REQUIRE(MeshHalo->exchangeState(H, U, NT > 0 ? ProvT[(Stage - 1) % 2] : nullptr, NT, S) == 0,
                             "RungeKutta4: provisional halo exchange failed" + MeshHalo->wireError());
            }
         }
         Ok = Tend->computeAllTendenciesStage(In, AuxState, *ProvT[(Stage - 1) % 2], CurLevel, CurLevel, Su, S);
         OMEGA_REQUIRE(Ok, "RungeKutta4: stage-fused RHS became unavailable mid-step");
      }
   }
   };
   if (!Exchanges && (UseGraphs || GraphCache::defaultOn()) && StageFusedKnownGood && !Tend->CustomThicknessTend && !Tend->CustomVelocityTend) {
      // one rank: nothing but kernel launches on S -- replay them as a graph (keyed by everything that enters them)
      GraphCache::Key Key;
      GraphCache::add(Key, State), GraphCache::add(Key, CurH.Ptr), GraphCache::add(Key, NextH.Ptr);
      GraphCache::add(Key, CurU.Ptr), GraphCache::add(Key, NextU.Ptr), GraphCache::add(Key, CurTr.Ptr);
      GraphCache::add(Key, NextTr.Ptr), GraphCache::add(Key, TimeStepSeconds), GraphCache::add(Key, (int)StoreStageTendencies);
      GraphCache::add(Key, Tend), GraphCache::add(Key, AuxState), GraphCache::add(Key, Tend->Params), GraphCache::add(Key, S);
      GraphCache::add(Key, (int)Tend->UseFusedRHS);
      GraphCache::add(Key, tuningGeneration()); // (the kernel structure options are read at every launch)
      GraphCache::add(Key, (int)AuxState->LayerThicknessAux.FluxThickEdgeChoice);
      GraphCache::add(Key, (int)AuxState->TracerAux.TracersOnEdgeChoice);
      GraphCache::add(Key, (int)AuxState->WindForcingAux.InterpChoice);
      Graphs.run(Key, S, RunStages);
   } else {
      RunStages();
   }
   if (!FirstStageOk)
      return false;
   StageFusedKnownGood = true;
   if (Overlap) { // the end-of-step exchange was started by the last stage: wait for it, then rotate
      joinExchange(S);
      State->rotateTimeLevels();
      Trc->rotateTimeLevels();
   } else {
      updateTimeLevels(State, S);
   }
   ++NStepsDone;
   return true;
}

void RungeKutta4Stepper::doStep(OceanState *State, hipStream_t S) {
   if (!ProvisState)
      finalizeInit();
   requireHealthyWire();
   if (FuseStageUpdates && doStepFused(State, S))
      return;
   const int CurLevel = 0, NextLevel = 1;
   Array3DReal NextTracerArray, CurTracerArray;
   OMEGA_REQUIRE(Trc->getAll(CurTracerArray, CurLevel) == 0 && Trc->getAll(NextTracerArray, NextLevel) == 0,
                 "RungeKutta4 doStep: error retrieving tracers");
   const int NT = Trc->NTracers;
   const R8 T0  = simTime();
   for (int Stage = 0; Stage < NStages; ++Stage) {
      Tend->ModelTime = T0 + coeff(RKC[Stage]); // StageTime (:87)
      if (Stage == 0) {
         // R^{(0)} = RHS(q^{n}, t^{n});  q^{n+1} = q^{n} + dt * RKB[0] * R^{(0)}
         weightTracers(NextTracerArray, CurTracerArray, State, CurLevel, S);
         Tend->computeAllTendencies(State, AuxState, CurTracerArray, CurLevel, CurLevel, S);
         updateStateByTend(State, NextLevel, State, CurLevel, coeff(RKB[Stage]), S);
         accumulateTracersUpdate(NextTracerArray, coeff(RKB[Stage]), S);
      } else {
         // q^{provis} = q^{n} + RKA[stage]*dt*R^{(s-1)};  R^{(s)} = RHS(q^{provis});  q^{n+1} += RKB[stage]*dt*R^{(s)}
         updateStateByTend(ProvisState.get(), CurLevel, State, CurLevel, coeff(RKA[Stage]), S);
         updateTracersByTend(ProvisTracers, CurTracerArray, ProvisState.get(), CurLevel, State, CurLevel,
                             coeff(RKA[Stage]), S);
         if (Stage == 2 && MeshHalo && MeshHalo->NNghbr > 0) { // depends on the halo width (:107-113)
            Array2DReal H, U;
            ProvisState->getLayerThickness(H, CurLevel);
            ProvisState->getNormalVelocity(U, CurLevel);
            Pacer::Range Timer("RK4:haloExchProvis", 3);
            OMEGA_REQUIRE(MeshHalo->exchangeState(H, U, NT > 0 ? &ProvisTracers : nullptr, NT, S) == 0,
                          "RungeKutta4: provisional halo exchange failed" + MeshHalo->wireError());
         }
         Tend->computeAllTendencies(ProvisState.get(), AuxState, ProvisTracers, CurLevel, CurLevel, S);
         updateStateByTend(State, NextLevel, State, NextLevel, coeff(RKB[Stage]), S);
         accumulateTracersUpdate(NextTracerArray, coeff(RKB[Stage]), S);
      }
   }
   finalizeTracersUpdate(NextTracerArray, State, NextLevel, S);
   updateTimeLevels(State, S);
   ++NStepsDone;
}

} // namespace OMEGA
