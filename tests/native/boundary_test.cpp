// boundary_test.cpp -- a consumer of the C++ boundary (omega_amd/csrc/*.h): compiled with hipcc against the
// library's headers, linked with libomega_amd.so, driving the path by the reference's class and method names
// (Decomp / Halo / HorzMesh / OceanState / AuxiliaryState / Tendencies / TimeStepper registries: create, get,
// getDefault, erase, clear; Tendencies::computeAllTendencies; RungeKutta4Stepper::doStep through TimeStepper;
// Halo::exchangeFullArrayHalo; DivergenceOnCell; custom tendencies as std::function with a HIP kernel defined
// HERE).  Checks:
//   1. fused RHS and one RK4 step against golden vectors (raw little-endian doubles written by the pytest wrapper
//      from tests/golden/*.npz), bit for bit, on a mesh read from an MPAS file through MeshFile;
//   2. the reference's time-stepper known answer (test/timeStepping/TimeStepperTest.cpp:375-388): du/dt = -0.5 u as
//      the custom velocity tendency, T = 1, dt = 0.2 and 0.1, L-inf error orders 4 / 1 / 2 +- 0.1.
//   3. (r6) the REFERENCE'S CALL-SITE FORMS: a stepper subclass written HERE against the reference's signatures only --
//      static Tracers::getAll, `SimTime + RKC[Stage] * TimeStep`, computeAllTendencies(..., TimeInstant),
//      updateStateByTend(..., TimeInterval), ProvisState->exchangeHalo(Level), Halo::exchangeFullArrayHalo(Array, OnCell),
//      State->updateTimeLevels(), Tracers::updateTimeLevels(), no stream anywhere (the scheme of
//      RungeKutta4Stepper.cpp:68-137, in this file's own words) -- gives the golden RK4 vectors bit for bit, as does
//      TimeStepper::doStep(OceanState *, TimeInstant &); a custom tendency in the reference's form (Array2DReal, ...,
//      TimeInstant) receives the stage times.
// usage: boundary_test <mesh.nc> <golden_dir> <K> <NT>
#include "AuxiliaryState.h"
#include "Decomp.h"
#include "Halo.h"
#include "HorzMesh.h"
#include "HorzOperators.h"
#include "MeshIO.h"
#include "OceanState.h"
#include "Tendencies.h"
#include "TimeStepper.h"
#include "Pacer.h"
#include "PeerWire.h"
#include "Tuning.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

using namespace OMEGA;

static int Failures = 0;
#define CHECK(cond, what)                                                                                          \
   do {                                                                                                            \
      if (!(cond)) {                                                                                               \
         std::printf("FAIL: %s (%s:%d)\n", what, __FILE__, __LINE__);                                              \
         ++Failures;                                                                                               \
      }                                                                                                            \
   } while (0)

static std::vector<double> readBin(const std::string &Path) {
   std::ifstream F(Path, std::ios::binary | std::ios::ate);
   if (!F)
      throw std::runtime_error("cannot open " + Path);
   const size_t N = (size_t)F.tellg() / sizeof(double);
   std::vector<double> V(N);
   F.seekg(0);
   F.read(reinterpret_cast<char *>(V.data()), (std::streamsize)(N * sizeof(double)));
   return V;
}

// global [nGlobal][K] -> local [rowsSize][K] through 1-based ids (sentinel row zero)
static std::vector<double> toLocal(const std::vector<double> &G, const HostArrayI4 &ID, int RowsSize, int K, int NT = 1,
                                   size_t NGlobal = 0) {
   std::vector<double> L((size_t)NT * RowsSize * K, 0.0);
   for (int T = 0; T < NT; ++T)
      for (int R = 0; R + 1 < RowsSize; ++R)
         std::memcpy(&L[((size_t)T * RowsSize + R) * K], &G[((size_t)T * NGlobal + (ID(R) - 1)) * K], K * sizeof(double));
   return L;
}
static bool sameRows(const std::vector<double> &Local, const std::vector<double> &Glob, const HostArrayI4 &ID, int NOwned,
                     int RowsSize, int K, int NT = 1, size_t NGlobal = 0) {
   for (int T = 0; T < NT; ++T)
      for (int R = 0; R < NOwned; ++R)
         if (std::memcmp(&Local[((size_t)T * RowsSize + R) * K], &Glob[((size_t)T * NGlobal + (ID(R) - 1)) * K],
                         K * sizeof(double)) != 0)
            return false;
   return true;
}

// DecayVelocityTendency of the reference test (TimeStepperTest.cpp:49-73) as this program's own HIP kernel
__global__ void decayKernel(double *Tend, const double *U, int NRows, int K, int Pitch, double Coeff) {
   const int I = blockIdx.x * blockDim.x + threadIdx.x;
   if (I < NRows * K) {
      const int R = I / K, L = I - R * K;
      Tend[(size_t)R * Pitch + L] -= Coeff * U[(size_t)R * Pitch + L];
   }
}

// ---- 3. a time stepper written against the reference's signatures (TimeStepper.h:82-84, 174-237; Tendencies.h:73-102;
// Tracers.h:119,195-199; OceanState.h:113-129; Halo.h:767) -- no hipStream_t, no seconds, no TracerStore ----
class CallSiteRK4 : public RungeKutta4Stepper {
 public:
   CallSiteRK4(const std::string &Name, R8 Dt) : RungeKutta4Stepper(Name, Dt) {}
   std::vector<double> StageSeconds; ///< what the stages passed as model time
   void step(OceanState *State, TimeInstant &SimTime) {
      const int Cur = 0, Next = 1;
      Array3DReal CurTr, NextTr;
      if (Tracers::getAll(CurTr, Cur) != 0 || Tracers::getAll(NextTr, Next) != 0)
         throw std::runtime_error("CallSiteRK4: Tracers::getAll failed");
      weightTracers(NextTr, CurTr, State, Cur);
      OceanState *In             = State;
      const Array3DReal *InTr    = &CurTr;
      for (int Stage = 0; Stage < NStages; ++Stage) {
         const TimeInstant StageTime = SimTime + RKC[Stage] * TimeStep;
         StageSeconds.push_back(StageTime.getSeconds());
         if (Stage > 0) { // provisional state of this stage from the tendencies of the one before
            const TimeInterval A = RKA[Stage] * TimeStep;
            updateStateByTend(ProvisState.get(), Cur, State, Cur, A);
            updateTracersByTend(ProvisTracers, CurTr, ProvisState.get(), Cur, State, Cur, A);
            if (Stage == 2) {
               ProvisState->exchangeHalo(Cur);
               MeshHalo->exchangeFullArrayHalo(ProvisTracers, OnCell);
            }
            In = ProvisState.get(), InTr = &ProvisTracers;
         }
         Tend->computeAllTendencies(In, AuxState, *InTr, Cur, Cur, StageTime);
         const TimeInterval B = RKB[Stage] * TimeStep;
         updateStateByTend(State, Next, State, Stage == 0 ? Cur : Next, B);
         accumulateTracersUpdate(NextTr, B);
      }
      finalizeTracersUpdate(NextTr, State, Next);
      State->updateTimeLevels();
      Tracers::updateTimeLevels();
      SimTime += TimeStep;
   }
};

int main(int argc, char **argv) {
   if (argc < 5) {
      std::printf("usage: %s mesh.nc golden_dir K NT\n", argv[0]);
      return 2;
   }
   try {
      const std::string Dir = argv[2];
      const int K = std::atoi(argv[3]), NT = std::atoi(argv[4]);
      deviceInit(0);
      MeshFile File(argv[1]);

      // ---- objects through the registries, by the reference's names ----
      Decomp *DefDecomp = Decomp::create("Default", File.desc(), 1, 0, 3);
      CHECK(DefDecomp && Decomp::getDefault() == DefDecomp, "Decomp::create / getDefault");
      CHECK(Decomp::create("Default", File.desc(), 1, 0, 3) == nullptr, "second create with the same name must fail");
      Halo *DefHalo     = Halo::create("Default", DefDecomp);
      HorzMesh *DefMesh = HorzMesh::create("Default", DefDecomp, K);
      OceanState *State = OceanState::create("Default", DefMesh, DefHalo, K, 2);
      AuxiliaryState *Aux = AuxiliaryState::create("Default", DefMesh, DefHalo, K, NT);
      Tendencies *Tend    = Tendencies::create("Default", DefMesh, K, NT, TendParams{});
      TracerStore Trc(DefMesh, DefHalo, K, NT, 2);
      CHECK(HorzMesh::get("Default") == DefMesh && Tendencies::get("nope") == nullptr, "get by name");
      const size_t NCg = File.desc().NCells, NEg = File.desc().NEdges;

      // ---- 1. golden vectors ----
      const auto Hg = readBin(Dir + "/h.bin"), Ug = readBin(Dir + "/u.bin"), Trg = readBin(Dir + "/tr.bin");
      const auto Hl = toLocal(Hg, DefDecomp->CellIDH, DefMesh->NCellsSize, K);
      const auto Ul = toLocal(Ug, DefDecomp->EdgeIDH, DefMesh->NEdgesSize, K);
      const auto Tl = toLocal(Trg, DefDecomp->CellIDH, DefMesh->NCellsSize, K, NT, NCg);
      State->copyToDevice(Hl.data(), Ul.data(), 0);
      Trc.copyToDevice(Tl.data(), 0);
      hipStream_t S;
      HIP_CHECK(hipStreamCreate(&S));
      Array3DReal TracerArray;
      Trc.getAll(TracerArray, 0);
      Tendencies::getDefault()->computeAllTendencies(State, Aux, TracerArray, 0, 0, S);
      HIP_CHECK(hipStreamSynchronize(S));
      std::vector<double> HT((size_t)DefMesh->NCellsSize * K), UT((size_t)DefMesh->NEdgesSize * K),
          TT((size_t)NT * DefMesh->NCellsSize * K);
      copyToHost(HT.data(), Tend->LayerThicknessTend);
      copyToHost(UT.data(), Tend->NormalVelocityTend);
      copyToHost(TT.data(), Tend->TracerTend);
      CHECK(sameRows(HT, readBin(Dir + "/hTend.bin"), DefDecomp->CellIDH, DefMesh->NCellsOwned, DefMesh->NCellsSize, K),
            "LayerThicknessTend equals the golden vector");
      CHECK(sameRows(UT, readBin(Dir + "/uTend.bin"), DefDecomp->EdgeIDH, DefMesh->NEdgesOwned, DefMesh->NEdgesSize, K),
            "NormalVelocityTend equals the golden vector");
      CHECK(sameRows(TT, readBin(Dir + "/trTend.bin"), DefDecomp->CellIDH, DefMesh->NCellsOwned, DefMesh->NCellsSize, K, NT,
                     NCg),
            "TracerTend equals the golden vector");

      TimeStepper *Stepper = TimeStepper::create("Default", TimeStepper::getFromStr("RungeKutta4"), 600.0, Tend, Aux,
                                                 DefMesh, DefHalo, &Trc);
      CHECK(Stepper && TimeStepper::getDefault() == Stepper, "TimeStepper::create / getDefault");
      Stepper->doStep(State, S);
      HIP_CHECK(hipStreamSynchronize(S));
      std::vector<double> H1(HT.size()), U1(UT.size()), T1(TT.size());
      State->copyToHost(H1.data(), U1.data(), 0);
      Trc.copyToHost(T1.data(), 0);
      CHECK(sameRows(H1, readBin(Dir + "/rk4_h.bin"), DefDecomp->CellIDH, DefMesh->NCellsOwned, DefMesh->NCellsSize, K),
            "RK4 layer thickness equals the golden vector");
      CHECK(sameRows(U1, readBin(Dir + "/rk4_u.bin"), DefDecomp->EdgeIDH, DefMesh->NEdgesOwned, DefMesh->NEdgesSize, K),
            "RK4 normal velocity equals the golden vector");
      CHECK(sameRows(T1, readBin(Dir + "/rk4_tr.bin"), DefDecomp->CellIDH, DefMesh->NCellsOwned, DefMesh->NCellsSize, K, NT,
                     NCg),
            "RK4 tracers equal the golden vector");

      // Halo::exchangeFullArrayHalo by its reference name (one rank: nothing to move, must return 0)
      Array2DReal Hc;
      State->getLayerThickness(Hc, 0);
      CHECK(DefHalo->exchangeFullArrayHalo(Hc, OnCell, S) == 0 && State->exchangeHalo(0, S) == 0, "exchangeFullArrayHalo");
      // (round 3) the reference's other element types by the same name, the timer ranges and the options of this build
      {
         Array1DI4 Ids("Ids", DefMesh->NCellsSize);
         Array2DI4 Pairs("Pairs", DefMesh->NEdgesSize, 2);
         Array1DReal Depth("Depth", DefMesh->NCellsSize);
         CHECK(DefHalo->exchangeFullArrayHalo(Ids, OnCell, S) == 0 && DefHalo->exchangeFullArrayHalo(Pairs, OnEdge, S) == 0 &&
                   DefHalo->exchangeFullArrayHalo(Depth, OnCell, S) == 0,
               "exchangeFullArrayHalo for Array1DI4 / Array2DI4 / Array1DReal");
         Pacer::Range Timer("boundary_test:customRange", 1); // the reference's Pacer::start / stop as a roctx range
         int V = -1;
         CHECK(getTuningOption("NarrowTables", V) && V == 1 && !setTuningOption("NoSuchOption", 1), "tuning options");
         CHECK(DefMesh->narrowView() == nullptr && DefMesh->view().NWideCells >= 0,
               "narrowView: a mesh of hexagons stored 6 wide needs no second table set");
         PeerWire Wire(1, 0, 4096);
         char Handle[PeerWire::HandleBytes];
         Wire.localHandle(Handle);
         Wire.connect(Handle);
         CHECK(Wire.connected() && Wire.status() == 0 && Wire.mailboxBytes() >= 4096, "PeerWire: one-rank wire");
      }

      // HorzOperators functor object
      {
         Array2DReal Div = Array2DReal::levels("Div", DefMesh->NCellsSize, K), Un;
         State->getNormalVelocity(Un, 0);
         DivergenceOnCell DivOp(DefMesh);
         DivOp(Div, Un, S);
         Aux->computeMomAux(State, 0, 0, S);
         HIP_CHECK(hipStreamSynchronize(S));
         std::vector<double> A((size_t)DefMesh->NCellsSize * K), B(A.size());
         copyToHost(A.data(), Div);
         copyToHost(B.data(), Aux->KineticAux.VelocityDivCell);
         // VelocityDivCell uses the (Dv*InvArea*Sign)*u product order, DivergenceOnCell (Dv*Sign)*u*InvArea: equal
         // to rounding, not bitwise
         double MaxRel = 0, Mx = 0;
         for (size_t I = 0; I < A.size(); ++I)
            Mx = std::fmax(Mx, std::fabs(B[I]));
         for (size_t I = 0; I < (size_t)DefMesh->NCellsOwned * K; ++I)
            MaxRel = std::fmax(MaxRel, std::fabs(A[I] - B[I]) / Mx);
         CHECK(Mx > 0 && MaxRel < 1e-14, "DivergenceOnCell agrees with KineticAuxVars' divergence");
      }

      // ---- 2. TimeStepperTest: orders 4 / 1 / 2 with du/dt = -0.5 u ----
      {
         const int K1 = 1;
         OceanState *TestState = OceanState::create("TestState", DefMesh, DefHalo, K1, 2);
         AuxiliaryState *TestAux = AuxiliaryState::create("TestAuxState", DefMesh, DefHalo, K1, NT);
         TendParams Off;
         Off.ThicknessFluxTendencyEnable = Off.PVTendencyEnable = Off.KETendencyEnable = Off.SSHTendencyEnable = 0;
         Off.VelDiffTendencyEnable = Off.VelHyperDiffTendencyEnable = Off.TracerHorzAdvTendencyEnable = 0;
         Off.TracerDiffTendencyEnable = Off.TracerHyperDiffTendencyEnable = 0;
         Tendencies *TestTend = Tendencies::create("TestTendencies", DefMesh, K1, NT, Off);
         const double Coeff   = 0.5;
         TestTend->CustomVelocityTend = [Coeff, DefMesh](const Array2DReal &NormalVelTend, const OceanState *St,
                                                         const AuxiliaryState *, int, int VelLvl, R8, hipStream_t Str) {
            Array2DReal NormalVelEdge;
            St->getNormalVelocity(NormalVelEdge, VelLvl);
            const int N = DefMesh->NEdgesAll * NormalVelTend.Ext[1];
            hipLaunchKernelGGL(decayKernel, dim3((N + 255) / 256), dim3(256), 0, Str, NormalVelTend.Ptr, NormalVelEdge.Ptr,
                               DefMesh->NEdgesAll, NormalVelTend.Ext[1], NormalVelTend.Pitch, Coeff);
         };
         TracerStore TestTrc(DefMesh, DefHalo, K1, NT, 2);
         const double TimeEnd = 1.0, Exact = std::exp(-Coeff * TimeEnd);
         const struct {
            const char *Name;
            double Order;
         } Schemes[3] = {{"RungeKutta4", 4.0}, {"Forward-Backward", 1.0}, {"RungeKutta2", 2.0}};
         for (const auto &Sch : Schemes) {
            double Err[2];
            double Dt = 0.2;
            for (int Ref = 0; Ref < 2; ++Ref, Dt /= 2) {
               const int NSteps = (int)std::ceil(TimeEnd / Dt);
               TimeStepper *St  = TimeStepper::create("TestTimeStepper", TimeStepper::getFromStr(Sch.Name), TimeEnd / NSteps,
                                                      TestTend, TestAux, DefMesh, DefHalo, &TestTrc);
               std::vector<double> One1((size_t)DefMesh->NCellsSize * K1, 1.0), OneE((size_t)DefMesh->NEdgesSize * K1, 1.0),
                   OneT((size_t)NT * DefMesh->NCellsSize * K1, 1.0);
               TestState->copyToDevice(One1.data(), OneE.data(), 0);
               TestTrc.copyToDevice(OneT.data(), 0);
               for (int I = 0; I < NSteps; ++I)
                  St->doStep(TestState, S);
               HIP_CHECK(hipStreamSynchronize(S));
               TestState->copyToHost(One1.data(), OneE.data(), 0);
               double E = 0;
               for (int R = 0; R < DefMesh->NEdgesOwned; ++R)
                  E = std::fmax(E, std::fabs(OneE[R] - Exact));
               Err[Ref] = E;
               TimeStepper::erase("TestTimeStepper");
            }
            const double Rate = std::log2(Err[0] / Err[1]);
            std::printf("%-17s errors %.3e %.3e  order %.3f (expected %.0f)\n", Sch.Name, Err[0], Err[1], Rate, Sch.Order);
            CHECK(std::fabs(Rate - Sch.Order) <= 0.1, "time stepper convergence order");
         }
         Tendencies::erase("TestTendencies");
         CHECK(Tendencies::get("TestTendencies") == nullptr, "erase");
      }
      // ---- 3. the reference's call-site forms ----
      {
         HIP_CHECK(hipDeviceSynchronize());
         State->copyToDevice(Hl.data(), Ul.data(), 0);     // (level 0 = the current level, whatever the rotation so far)
         Trc.copyToDevice(Tl.data(), 0);
         Tracers::setDefault(&Trc);
         CHECK(Tracers::getDefault() == &Trc && Tracers::getNumTracers() == NT, "Tracers: static interface on a default store");
         CallSiteRK4 Mine("CallSite", 600.0);
         Mine.attachData(Tend, Aux, DefMesh, DefHalo, nullptr); // nullptr: the static Tracers' store, as in the reference
         Mine.finalizeInit();
         TimeInstant SimTime = TimeInstant::fromSeconds(1200.0);
         // a custom tendency in the REFERENCE'S form (Tendencies.h:51-53): records `Time - ReferenceTime` in seconds
         std::vector<double> Seen;
         const TimeInstant ReferenceTime = TimeInstant::fromSeconds(200.0);
         Tend->CustomThicknessTend       = [&Seen, ReferenceTime](Array2DReal, const OceanState *, const AuxiliaryState *, int, int,
                                                            TimeInstant Time) {
            R8 ElapsedSec;
            TimeInterval Elapsed = Time - ReferenceTime;
            Elapsed.get(ElapsedSec, TimeUnits::Seconds);
            Seen.push_back(ElapsedSec);
         };
         Mine.step(State, SimTime);
         HIP_CHECK(hipDeviceSynchronize());
         Tend->CustomThicknessTend = nullptr;
         CHECK(SimTime == TimeInstant::fromSeconds(1800.0) && (SimTime - ReferenceTime).getSeconds() == 1600.0, "SimTime advanced by TimeStep");
         CHECK(Mine.StageSeconds == (std::vector<double>{1200.0, 1500.0, 1500.0, 1800.0}), "stage times SimTime + RKC * TimeStep");
         CHECK(Seen == (std::vector<double>{1000.0, 1300.0, 1300.0, 1600.0}), "reference-form custom tendency saw Time - ReferenceTime");
         State->copyToHost(H1.data(), U1.data(), 0);
         Trc.copyToHost(T1.data(), 0);
         CHECK(sameRows(H1, readBin(Dir + "/rk4_h.bin"), DefDecomp->CellIDH, DefMesh->NCellsOwned, DefMesh->NCellsSize, K) &&
                   sameRows(U1, readBin(Dir + "/rk4_u.bin"), DefDecomp->EdgeIDH, DefMesh->NEdgesOwned, DefMesh->NEdgesSize, K) &&
                   sameRows(T1, readBin(Dir + "/rk4_tr.bin"), DefDecomp->CellIDH, DefMesh->NCellsOwned, DefMesh->NCellsSize, K,
                            NT, NCg),
               "a stepper written against the reference's signatures gives the golden RK4 vectors");
         // ... and TimeStepper::doStep(OceanState *, TimeInstant &) of the library's own scheme, through a const pointer
         State->copyToDevice(Hl.data(), Ul.data(), 0);
         Trc.copyToDevice(Tl.data(), 0);
         const TimeStepper *ConstStepper = TimeStepper::getDefault();
         TimeInstant T2                  = TimeInstant::fromSeconds(0.0);
         ConstStepper->doStep(State, T2);
         HIP_CHECK(hipDeviceSynchronize());
         CHECK(T2.getSeconds() == 600.0 && ConstStepper->getTimeStep() == TimeInterval(10.0, TimeUnits::Minutes), "doStep(State, SimTime)");
         State->copyToHost(H1.data(), U1.data(), 0);
         Trc.copyToHost(T1.data(), 0);
         CHECK(sameRows(H1, readBin(Dir + "/rk4_h.bin"), DefDecomp->CellIDH, DefMesh->NCellsOwned, DefMesh->NCellsSize, K) &&
                   sameRows(U1, readBin(Dir + "/rk4_u.bin"), DefDecomp->EdgeIDH, DefMesh->NEdgesOwned, DefMesh->NEdgesSize, K) &&
                   sameRows(T1, readBin(Dir + "/rk4_tr.bin"), DefDecomp->CellIDH, DefMesh->NCellsOwned, DefMesh->NCellsSize, K,
                            NT, NCg),
               "doStep(OceanState *, TimeInstant &) gives the golden RK4 vectors");
         // the other reference-signature methods compile and run: group tendencies, aux state, state exchange
         Array3DReal Tr0;
         Tracers::getAll(Tr0, 0);
         Tend->computeThicknessTendencies(State, Aux, 0, 0, T2);
         Tend->computeVelocityTendenciesOnly(State, Aux, 0, 0, T2);
         Tend->computeTracerTendencies(State, Aux, Tr0, 0, 0, T2);
         Aux->computeAll(State, Tr0, 0, 0);
         Aux->computeMomAux(State, 0, 0);
         CHECK(State->exchangeHalo(0) == 0 && Tracers::exchangeHalo(0) == 0 && Aux->exchangeHalo() == 0 &&
                   DefHalo->exchangeFullArrayHalo(Tr0, OnCell) == 0,
               "reference-signature exchanges");
         HIP_CHECK(hipDeviceSynchronize());
         Tracers::setDefault(nullptr);
         CHECK(Tracers::getAll(Tr0, 0) == -1, "Tracers::getAll without a default store returns an error code");
      }
      HIP_CHECK(hipStreamDestroy(S));
      // reference shutdown order (TimeStepperTest.cpp finalizeTimeStepperTest)
      TimeStepper::clear();
      Tendencies::clear();
      AuxiliaryState::clear();
      OceanState::clear();
      HorzMesh::clear();
      Halo::clear();
      Decomp::clear();
      CHECK(Decomp::getDefault() == nullptr, "clear");
   } catch (const std::exception &E) {
      std::printf("FAIL: exception: %s\n", E.what());
      return 1;
   }
   if (Failures == 0)
      std::printf("boundary_test OK\n");
   return Failures == 0 ? 0 : 1;
}
