// Kernels.h -- host-callable launchers of the hand-written HIP kernels (gfx950).
// Every launcher is asynchronous on the given stream and takes raw device pointers.
#ifndef OMEGA_AMD_KERNELS_H
#define OMEGA_AMD_KERNELS_H

#include "../HorzMesh.h"

namespace OMEGA {

/// Device pointers of every AuxiliaryState array
/// (reference: components/omega/src/ocn/auxiliaryVars/*.h public members).
struct AuxPtrs {
   Real *KineticEnergyCell, *VelocityDivCell;                      // C x K
   Real *FluxLayerThickEdge, *MeanLayerThickEdge;                  // E x K
   Real *SshCell;                                                  // C x K
   Real *RelVortVertex, *NormRelVortVertex, *NormPlanetVortVertex; // V x K
   Real *InvThickVertex; // V x K: 1/LayerThickVertex (VorticityAuxVars.h:47), fused RHS only -- see launchVertexAuxState1
   Real *NormRelVortEdge, *NormPlanetVortEdge;                     // E x K
   Real *Del2Edge, *Del2DivCell, *Del2RelVortVertex;               // E, C, V x K
   Real *HTracersEdge, *Del2TracersCell;                           // NT x E x K, NT x C x K
   Real *NormalStressEdge, *ZonalStressCell, *MeridStressCell;     // E, C, C
};

/// Enable flags and coefficients (reference: Tendencies::readTendConfig,
/// components/omega/src/ocn/Tendencies.cpp:123-213; AuxiliaryState::readConfigOptions,
/// components/omega/src/ocn/AuxiliaryState.cpp:259-308; defaults configs/Default.yml:25-52)
struct TendParams {
   int ThicknessFluxTendencyEnable = 1, PVTendencyEnable = 1, KETendencyEnable = 1, SSHTendencyEnable = 1,
       VelDiffTendencyEnable = 1, VelHyperDiffTendencyEnable = 1, WindForcingTendencyEnable = 0,
       BottomDragTendencyEnable = 0, TracerHorzAdvTendencyEnable = 1, TracerDiffTendencyEnable = 1,
       TracerHyperDiffTendencyEnable = 1;
   int FluxThicknessUpwind = 0, FluxTracerUpwind = 0, WindInterpIsotropic = 1;
   Real ViscDel2 = 1.0e3, ViscDel4 = 1.2e11, DivFactor = 1.0, EddyDiff2 = 10.0, EddyDiff4 = 0.0, Density0 = 1026.0,
        BottomDragCoeff = 0.0;
};

// ---- AuxiliaryState launches, one per reference parallelFor (AuxiliaryState.cpp:79-182) ----
/// StoreNorm: write NormRelVortVertex / NormPlanetVortVertex (the reference's arrays); StoreInv: write
/// InvThickVertex instead, from which the fused PV kernels rebuild both with the same multiplication
/// (RelVort*Inv, FVertex*Inv: one vertex array less to write and to gather)
void launchVertexAuxState1(const MeshView &M, int K, const AuxPtrs &A, const Real *H, const Real *U, hipStream_t S,
                           bool StoreNorm = true, bool StoreInv = false);
/// the vertices of a list only, storing RelVort and 1/LayerThickVertex (the vertices the merged level-1 kernel leaves out)
void launchVertexAuxState1List(const MeshView &M, int K, const AuxPtrs &A, const Real *H, const Real *U, hipStream_t S,
                               const I4 *Vertices, int N);
void launchCellAuxState1(const MeshView &M, int K, const AuxPtrs &A, const Real *U, hipStream_t S);
void launchEdgeAuxState1(const MeshView &M, const AuxPtrs &A, int Isotropic, hipStream_t S);
void launchEdgeAuxState2(const MeshView &M, int K, const AuxPtrs &A, const Real *H, const Real *U, int FluxUpwind,
                         hipStream_t S);
void launchVertexAuxState2(const MeshView &M, int K, const AuxPtrs &A, hipStream_t S);
void launchCellAuxState2(const MeshView &M, int K, const AuxPtrs &A, hipStream_t S);
void launchCellAuxState3(const MeshView &M, int K, const AuxPtrs &A, const Real *H, hipStream_t S);
void launchEdgeAuxState4(const MeshView &M, int K, int NT, const AuxPtrs &A, const Real *U, const Real *H,
                         const Real *Tr, int TracerUpwind, hipStream_t S);
void launchCellAuxState4(const MeshView &M, int K, int NT, const AuxPtrs &A, const Real *Tr, hipStream_t S);
/// Tendencies::computeThicknessTendencies' computeLayerThickAux (Tendencies.cpp:507-512)
void launchLayerThickAuxEdge(const MeshView &M, int K, const AuxPtrs &A, const Real *H, const Real *U, int FluxUpwind,
                             hipStream_t S);

// ---- Tendencies::compute*TendenciesOnly: all enabled terms of a group in one launch,
//      accumulated in registers in the reference's term order, one store per element ----
void launchThicknessTendOnly(const MeshView &M, int K, const TendParams &P, const AuxPtrs &A, Real *HTend,
                             const Real *U, hipStream_t S);
void launchVelocityTendOnly(const MeshView &M, int K, const TendParams &P, const AuxPtrs &A, Real *UTend,
                            const Real *U, hipStream_t S);
void launchTracerTendOnly(const MeshView &M, int K, int NT, const TendParams &P, const AuxPtrs &A, Real *TrTend,
                          const Real *U, const Real *Tr, hipStream_t S);

// ---- fused RHS (Tendencies::computeAllTendencies): see FusedKernels.hip ----
/// Kernel order (also the index of the optional timing events, Ev[i] recorded BEFORE kernel i,
/// Ev[7] after the last): 0 vertex L1, 1 cell L1, 2 cell L2, 3 vertex L2, 4 + 5 edge L3 (cell-centric
/// PV: side-0 sums, then side-1 sums + remaining terms; otherwise one edge kernel in slot 4),
/// 6 cell L3.  FusedKernelNames[i] is set by the launcher to the kernel actually used ("" = none).
constexpr int FusedNumKernels = 7;
/// MaxEdges in [5,8] and every array plane <= FusedMaxPlaneBytes (32-bit byte offsets inside a plane; the value is the
/// size every buffer resource of the fused kernels is given, FusedKernelsImpl.h: BufOOB)
constexpr unsigned FusedMaxPlaneBytes = 0xffffff00u;
bool fusedRHSSupported(const MeshView &M, int K);
extern const char *FusedKernelNames[FusedNumKernels];
/// Runge-Kutta stage update folded into the kernels that produce the tendencies (the arithmetic of
/// TimeStepper::updateStateByTend / updateTracersByTend / weightTracers / accumulateTracersUpdate /
/// finalizeTracersUpdate, TimeStepper.cpp:378-524, applied element by element in the epilogue):
///   Next  = (First ? Cur : Next) + CB*Tend          tracers: (First ? CurTr*CurH : NextTr) + CB*Tend,
///                                                            and / NextH(new) when Last
///   Prov  = Cur + CA*Tend   (unless Last)           tracers: (CurTr*CurH + CA*Tend) / ProvH(new)
/// Prov* are OUT buffers distinct from the RHS inputs (neighbours still read the inputs).
struct StageUpdate {
   Real CB = 0, CA = 0; ///< seconds
   int First = 0, Last = 0, StoreTend = 0;
   Real *NextH = nullptr, *NextU = nullptr, *NextTr = nullptr;
   const Real *CurH = nullptr, *CurU = nullptr, *CurTr = nullptr;
   Real *ProvH = nullptr, *ProvU = nullptr, *ProvTr = nullptr;
   /// Overlap of the halo exchange that follows this stage with the stage's own interior work: when
   /// AfterBand is set, the kernels that finish u and the tracers run first over BandCells (halo cells and
   /// the owned cells whose values any neighbour receives: HorzMesh::BandCells), then AfterBand(Ctx) is
   /// called -- the stepper starts the exchange on its communication stream there -- and the same
   /// kernels continue over InteriorCells while the messages travel.
   void (*AfterBand)(void *) = nullptr;
   void *AfterBandCtx        = nullptr;
   /// Set together with AfterBand when the exchange started there delivers EVERY halo element of EVERY array this
   /// stage writes that is read again before the next such exchange (RK4: stage 1 writes Prov -- exchanged now -- and
   /// Next, whose halo nobody reads before the end-of-step exchange replaces it; the last stage writes Next only).
   /// The kernels that finish u and the tracers then skip the halo cells that finish nothing owned here
   /// (MeshView::BandSendCells instead of BandCells).  Ignored when StoreTend is set: stored tendencies keep their
   /// halo values.
   int HaloOutputsReplaced = 0;
   /// Optional (both or none): the band part is launched on BandStream -- the stepper's communication stream, on
   /// which the exchange follows in stream order -- after everything queued on S so far (BandReady is recorded on S
   /// and waited for there).  The interior part on S then shares the GPU with the band launch, which is too small to
   /// fill it, instead of waiting for it.  The two parts write disjoint elements and read only the stage's inputs.
   hipStream_t BandStream = nullptr;
   hipEvent_t BandReady   = nullptr;
   /// Sweep lengths (0 = all local cells; the local numbering is owned, halo layer 1, 2, ... so a prefix is "through
   /// layer i"): how far the merged level-1 kernel (NCellsL1) and the level-3 velocity / tracer kernels (NCellsVel,
   /// NCellsTr) have to go for everything this rank still reads of the stage's results.  Set by the stepper from the
   /// halo layer bounds; ignored when StoreTend is set.
   I4 NCellsL1 = 0, NCellsVel = 0, NCellsTr = 0;
};
/// Returns false (nothing launched) when Stage != nullptr and the stage-fused kernels do not cover
/// this mesh / option set; the caller then runs the plain RHS followed by the update kernels.
bool launchFusedRHS(const MeshView &M, int K, int NT, const TendParams &P, const AuxPtrs &A, Real *HTend, Real *UTend,
                    Real *TrTend, const Real *H, const Real *U, const Real *Tr, hipStream_t S,
                    hipEvent_t *Ev = nullptr, Real *EdgeScratch = nullptr, const StageUpdate *Stage = nullptr,
                    const MeshView *Narrow = nullptr);
/// Narrow: HorzMesh::narrowView() -- the per-(cell, slot) tables stored MaxEdges-1 wide; the sweeps then run the
/// (MaxEdges-1)-slot kernels on them and the cells with MaxEdges edges (M.WideCells) go through list launches on M
/// EdgeScratch: optional [NEdgesSize][K] work array for the cell-centric PV sums (faster path)

// ---- reductions (base/Reductions.h): double-double local sums on the device ----
/// sum_i A[i] (B == nullptr) or sum_i A[i]*B[i] over N values, accumulated in double-double (Knuth
/// two-sum per element, the ddSum combination of Reductions.h:24-35 between threads / workgroups);
/// HiLo[0] + HiLo[1] is the sum, HiLo[0] its rounded value.  Synchronises the stream.
void localSumDD(const Real *A, const Real *B, size_t N, hipStream_t S, double HiLo[2]);
/// element rows [0, NRows) of a [RowsSize][K] array times a per-row weight (e.g. AreaCell): sum_r W[r]*sum_k A[r][k]*B[r][k]
/// (Pitch = row pitch of A and B in values)
void localWeightedSumDD(const Real *W, const Real *A, const Real *B, int NRows, int K, int Pitch, hipStream_t S,
                        double HiLo[2]);
/// ddSum (Reductions.h:24-35) over NPairs (hi, lo) pairs in order: how per-rank partial sums are combined
void combineDD(const double *Pairs, int NPairs, double HiLo[2]);

// ---- ManufacturedSolution custom tendencies (CustomTendencyTerms.cpp:112-208) ----
struct ManufacturedParams {
   Real H0, Eta0, Kx, Ky, AngFreq, Grav, ViscDel2, ViscDel4;
   int VelDiffTendencyEnable, VelHyperDiffTendencyEnable;
};
void launchManufacturedThickness(int NCells, int K, Real *Tend, const Real *XCell, const Real *YCell,
                                 const ManufacturedParams &P, Real ElapsedSec, hipStream_t S);
void launchManufacturedVelocity(int NEdges, int K, Real *Tend, const Real *XEdge, const Real *YEdge, const Real *FEdge,
                                const Real *AngleEdge, const ManufacturedParams &P, Real ElapsedSec, hipStream_t S);

// ---- HorzOperators (HorzOperators.h:9-187): sweeps over elements [0, N) x K levels of arrays with row pitch Pitch ----
void launchDivergenceOnCell(const MeshView &M, int N, int K, int Pitch, Real *DivCell, const Real *VecEdge, hipStream_t S);
void launchGradientOnEdge(const MeshView &M, int N, int K, int Pitch, Real *GradEdge, const Real *ScalarCell, hipStream_t S);
void launchCurlOnVertex(const MeshView &M, int N, int K, int Pitch, Real *CurlVertex, const Real *VecEdge, hipStream_t S);
void launchTangentialReconOnEdge(const MeshView &M, int N, int K, int Pitch, Real *ReconEdge, const Real *VecEdge,
                                 hipStream_t S);
void launchInterpCellToEdge(const MeshView &M, int N, Real *ArrayEdge, const Real *ArrayCell, int Isotropic,
                            hipStream_t S);

// ---- TimeStepper update kernels (TimeStepper.cpp:378-524) ----
void launchUpdateByTend(int NRows, int K, Real *X1, const Real *X2, const Real *Tend, Real Coeff, hipStream_t S);
void launchUpdateTracersByTend(int NT, int NRows, int RowsSize, int K, Real *NextTr, const Real *CurTr, const Real *H1,
                               const Real *H2, const Real *TrTend, Real Coeff, hipStream_t S);
void launchWeightTracers(int NT, int NRows, int RowsSize, int K, Real *NextTr, const Real *CurTr, const Real *HCur,
                         hipStream_t S);
void launchAccumulateTracers(int NT, int NRows, int RowsSize, int K, Real *Accum, const Real *TrTend, Real Coeff,
                             hipStream_t S);
void launchFinalizeTracers(int NT, int NRows, int RowsSize, int K, Real *NextTr, const Real *HNext, hipStream_t S);

// ---- Halo pack / unpack (Halo.h:324-414, 566-653): message layout Buf[(T*NList + I)*K + k] per neighbour ----

/// Every row of one exchange in ONE launch: job j = (piece, row) copies the K values of row `row` of
/// piece `piece` (base pointers in HaloBases; a 3-D array is its [NT*RowsSize][K] plane stack) to / from buffer
/// row j.  The job tables live on the device (Halo::Plan).
constexpr int HaloMaxPieces = 4;
struct HaloBases {
   void *P[HaloMaxPieces];
};
/// (buffer rows are compact, K values of ElemBytes (4: I4 / R4, 8: R8 / I8) bytes each; array rows have pitch Pitch values)
void launchHaloPackAll(void *Buf, const HaloBases &B, const I4 *Jobs, size_t NRows, int K, int Pitch, int ElemBytes,
                       hipStream_t S);
/// SkipIfSet: optional device-readable status word (PeerWire); if it is non-zero when the kernel runs nothing is unpacked
void launchHaloUnpackAll(const HaloBases &B, const void *Buf, const I4 *Jobs, size_t NRows, int K, int Pitch,
                         int ElemBytes, hipStream_t S, const int *SkipIfSet = nullptr);

} // namespace OMEGA
#endif
