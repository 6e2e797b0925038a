// TendKernels.hip -- Tendencies::compute{Thickness,Velocity,Tracer}TendenciesOnly
// (components/omega/src/ocn/Tendencies.cpp:257-486) as one HIP kernel per group.  The
// reference zero-fills the tendency array and then runs one parallelFor per enabled term,
// each read-modify-writing the array; here the same terms are accumulated in registers in
// the same order and stored once.  Term functors: components/omega/src/ocn/TendencyTerms.h.
// Compiled with -ffp-contract=off (bit-for-bit the reference's operation order).
#include "KernelCommon.h"
#include "Kernels.h"

#include <algorithm>
#include <vector>

namespace OMEGA {

// ---------------------------------------------------------------------------------------
// ThicknessFluxDivOnCell (TendencyTerms.h:35-58)
struct ThickTendBody {
   MeshView M;
   int K;
   int Enabled;
   const Real *Flux, *U;
   Real *Tend;
   struct Lds {
      Real *DvS, *InvA;
      int *Edge, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME = M.MaxEdges;
      return ldsRound8(sizeof(Real) * Tile * ME) + ldsRound8(sizeof(Real) * Tile) +
             ldsRound8(sizeof(int) * Tile * ME) + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      const int ME = M.MaxEdges;
      LdsCarver C{P};
      Lds L;
      L.DvS  = C.take<Real>(Tile * ME);
      L.InvA = C.take<Real>(Tile);
      L.Edge = C.take<int>(Tile * ME);
      L.N    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME = M.MaxEdges;
      for (int I = Tid; I < Cnt * ME; I += NThr) {
         const size_t G = (size_t)First * ME + I;
         L.DvS[I]       = M.DvSignOnCell[G];
         L.Edge[I]      = M.EdgesOnCell[G];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         L.N[I]    = M.NEdgesOnCell[First + I];
         L.InvA[I] = M.InvAreaCell[First + I];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int ICell, int Kv) const {
      const int ME = M.MaxEdges;
      T TendV      = splat<T>(0.0);
      if (Enabled) {
         T DivTmp       = splat<T>(0.0);
         const int N    = L.N[Le];
         const Real InvA = L.InvA[Le];
         for (int J = 0; J < N; ++J) {
            const int JEdge = L.Edge[Le * ME + J];
            DivTmp -= L.DvS[Le * ME + J] * ldk<T>(Flux, JEdge, K, Kv) * ldk<T>(U, JEdge, K, Kv) * InvA;
         }
         TendV -= DivTmp;
      }
      stk<T>(Tend, ICell, K, Kv, TendV);
   }
};
void launchThicknessTendOnly(const MeshView &M, int K, const TendParams &P, const AuxPtrs &A, Real *HTend,
                             const Real *U, hipStream_t S) {
   ThickTendBody B{M, K, P.ThicknessFluxTendencyEnable, A.FluxLayerThickEdge, U, HTend};
   launchTile(B, M.NCellsAll, K, S);
}

// ---------------------------------------------------------------------------------------
// Velocity terms: PotentialVortHAdvOnEdge :81-108, KEGradOnEdge :127-140, SSHGradOnEdge
// :159-173, VelocityDiffusionOnEdge :195-219, VelocityHyperDiffOnEdge :244-269,
// WindForcingOnEdge :291-301, BottomDragOnEdge :319-334
struct VelTendBody {
   MeshView M;
   int K;
   TendParams P;
   AuxPtrs A;
   const Real *U;
   Real *Tend;
   int KLog = 0; ///< number of levels (K is the row pitch): set by launchTile
   struct Lds {
      Real *W, *InvDc, *InvDv, *Mask, *MaskGrav, *C2, *C4;
      int *EoE, *C0, *C1, *V0, *V1, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME2 = M.MaxEdges2;
      return ldsRound8(sizeof(Real) * Tile * ME2) + ldsRound8(sizeof(Real) * Tile) * 6 +
             ldsRound8(sizeof(int) * Tile * ME2) + ldsRound8(sizeof(int) * Tile) * 5;
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      const int ME2 = M.MaxEdges2;
      LdsCarver C{Ptr};
      Lds L;
      L.W        = C.take<Real>(Tile * ME2);
      L.InvDc    = C.take<Real>(Tile);
      L.InvDv    = C.take<Real>(Tile);
      L.Mask     = C.take<Real>(Tile);
      L.MaskGrav = C.take<Real>(Tile);
      L.C2       = C.take<Real>(Tile);
      L.C4       = C.take<Real>(Tile);
      L.EoE      = C.take<int>(Tile * ME2);
      L.C0       = C.take<int>(Tile);
      L.C1       = C.take<int>(Tile);
      L.V0       = C.take<int>(Tile);
      L.V1       = C.take<int>(Tile);
      L.N        = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME2   = M.MaxEdges2;
      const Real Grav = 9.80665; // TendencyTerms.h:176
      for (int I = Tid; I < Cnt * ME2; I += NThr) {
         const size_t G = (size_t)First * ME2 + I;
         L.W[I]         = M.WeightsOnEdge[G];
         L.EoE[I]       = M.EdgesOnEdge[G];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         const int E     = First + I;
         const Real Mask = M.EdgeMask1D[E];
         L.InvDc[I]      = M.InvDcEdge[E];
         L.InvDv[I]      = M.InvDvEdge[E];
         L.Mask[I]       = Mask;
         L.MaskGrav[I]   = Mask * Grav;                                 // :169
         L.C2[I]         = Mask * P.ViscDel2 * M.MeshScalingDel2[E];    // :217
         L.C4[I]         = Mask * P.ViscDel4 * M.MeshScalingDel4[E];    // :267
         L.C0[I]         = M.CellsOnEdge[2 * E];
         L.C1[I]         = M.CellsOnEdge[2 * E + 1];
         L.V0[I]         = M.VerticesOnEdge[2 * E];
         L.V1[I]         = M.VerticesOnEdge[2 * E + 1];
         L.N[I]          = M.NEdgesOnEdge[E];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IEdge, int Kv) const {
      const int ME2 = M.MaxEdges2;
      const int C0 = L.C0[Le], C1 = L.C1[Le], V0 = L.V0[Le], V1 = L.V1[Le];
      const Real InvDc = L.InvDc[Le], InvDv = L.InvDv[Le];
      T TendV = splat<T>(0.0);
      if (P.PVTendencyEnable) {
         T VortTmp   = splat<T>(0.0);
         const T QRe = ldk<T>(A.NormRelVortEdge, IEdge, K, Kv);
         const T QFe = ldk<T>(A.NormPlanetVortEdge, IEdge, K, Kv);
         const int N = L.N[Le];
         for (int J = 0; J < N; ++J) {
            const int JEdge = L.EoE[Le * ME2 + J];
            const T NormVort =
                (QRe + QFe + ldk<T>(A.NormRelVortEdge, JEdge, K, Kv) + ldk<T>(A.NormPlanetVortEdge, JEdge, K, Kv)) *
                0.5;
            VortTmp += L.W[Le * ME2 + J] * ldk<T>(A.FluxLayerThickEdge, JEdge, K, Kv) * ldk<T>(U, JEdge, K, Kv) *
                       NormVort;
         }
         TendV += L.Mask[Le] * VortTmp;
      }
      if (P.KETendencyEnable)
         TendV -= L.Mask[Le] * (ldk<T>(A.KineticEnergyCell, C1, K, Kv) - ldk<T>(A.KineticEnergyCell, C0, K, Kv)) *
                  InvDc;
      if (P.SSHTendencyEnable)
         TendV -= L.MaskGrav[Le] * (ldk<T>(A.SshCell, C1, K, Kv) - ldk<T>(A.SshCell, C0, K, Kv)) * InvDc;
      if (P.VelDiffTendencyEnable) {
         const T Del2U = ((ldk<T>(A.VelocityDivCell, C1, K, Kv) - ldk<T>(A.VelocityDivCell, C0, K, Kv)) * InvDc -
                          (ldk<T>(A.RelVortVertex, V1, K, Kv) - ldk<T>(A.RelVortVertex, V0, K, Kv)) * InvDv);
         TendV += L.C2[Le] * Del2U;
      }
      if (P.VelHyperDiffTendencyEnable) {
         const T Del2U =
             (P.DivFactor * (ldk<T>(A.Del2DivCell, C1, K, Kv) - ldk<T>(A.Del2DivCell, C0, K, Kv)) * InvDc -
              (ldk<T>(A.Del2RelVortVertex, V1, K, Kv) - ldk<T>(A.Del2RelVortVertex, V0, K, Kv)) * InvDv);
         TendV -= L.C4[Le] * Del2U;
      }
      constexpr int W = VecW<T>::W;
      if (P.WindForcingTendencyEnable && Kv == 0) { // acts on level 0 only (:294)
         const Real InvThickEdge = 1. / A.MeanLayerThickEdge[(size_t)IEdge * K];
         setc(TendV, 0, getc(TendV, 0) + L.Mask[Le] * InvThickEdge * A.NormalStressEdge[IEdge] / P.Density0);
      }
      if (P.BottomDragTendencyEnable && (Kv + 1) * W >= KLog) { // bottom level KBot = K-1 (:323)
         const int KBot          = KLog - 1;
         const int Comp          = KBot - Kv * W;
         const Real VelNormEdge  = sqrt(A.KineticEnergyCell[(size_t)C0 * K + KBot] + A.KineticEnergyCell[(size_t)C1 * K + KBot]);
         const Real InvThickEdge = 1. / A.MeanLayerThickEdge[(size_t)IEdge * K + KBot];
         setc(TendV, Comp,
              getc(TendV, Comp) - L.Mask[Le] * P.BottomDragCoeff * VelNormEdge * InvThickEdge * U[(size_t)IEdge * K + KBot]);
      }
      stk<T>(Tend, IEdge, K, Kv, TendV);
   }
};
void launchVelocityTendOnly(const MeshView &M, int K, const TendParams &P, const AuxPtrs &A, Real *UTend,
                            const Real *U, hipStream_t S) {
   VelTendBody B{M, K, P, A, U, UTend};
   launchTile(B, M.NEdgesAll, K, S);
}

// ---------------------------------------------------------------------------------------
// Tracer terms: TracerHorzAdvOnCell :349-373, TracerDiffOnCell :394-426,
// TracerHyperDiffOnCell :449-480; tracer loop inside the thread.
struct TracerTendBody {
   MeshView M;
   int K, NT;
   TendParams P;
   const Real *U, *Tr, *HTr, *HMean, *Del2Tr;
   Real *Tend;
   struct Lds {
      Real *MDvS, *Df2, *Df4, *InvA;
      int *Edge, *C0, *C1, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME = M.MaxEdges;
      return ldsRound8(sizeof(Real) * Tile * ME) * 3 + ldsRound8(sizeof(Real) * Tile) +
             ldsRound8(sizeof(int) * Tile * ME) * 3 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      const int ME = M.MaxEdges;
      LdsCarver C{Ptr};
      Lds L;
      L.MDvS = C.take<Real>(Tile * ME);
      L.Df2  = C.take<Real>(Tile * ME);
      L.Df4  = C.take<Real>(Tile * ME);
      L.InvA = C.take<Real>(Tile);
      L.Edge = C.take<int>(Tile * ME);
      L.C0   = C.take<int>(Tile * ME);
      L.C1   = C.take<int>(Tile * ME);
      L.N    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME = M.MaxEdges;
      for (int I = Tid; I < Cnt * ME; I += NThr) {
         const size_t G = (size_t)First * ME + I;
         L.MDvS[I]      = M.MaskDvSignOnCell[G];
         L.Df2[I]       = M.Diff2CoefOnCell[G];
         L.Df4[I]       = M.Diff4CoefOnCell[G];
         L.Edge[I]      = M.EdgesOnCell[G];
         L.C0[I]        = M.CellsOnEdgeOnCell[2 * G];
         L.C1[I]        = M.CellsOnEdgeOnCell[2 * G + 1];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         L.N[I]    = M.NEdgesOnCell[First + I];
         L.InvA[I] = M.InvAreaCell[First + I];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int ICell, int Kv) const {
      const int ME    = M.MaxEdges;
      const int N     = L.N[Le];
      const Real InvA = L.InvA[Le];
      const size_t CStride = (size_t)M.NCellsSize * K, EStride = (size_t)M.NEdgesSize * K;
      for (int Lt = 0; Lt < NT; ++Lt) {
         T TendV = splat<T>(0.0);
         if (P.TracerHorzAdvTendencyEnable) {
            T HAdvTmp       = splat<T>(0.0);
            const Real *HTrL = HTr + Lt * EStride;
            for (int J = 0; J < N; ++J) {
               const int JEdge = L.Edge[Le * ME + J];
               HAdvTmp -= L.MDvS[Le * ME + J] * ldk<T>(HTrL, JEdge, K, Kv) * ldk<T>(U, JEdge, K, Kv) * InvA;
            }
            TendV -= HAdvTmp;
         }
         if (P.TracerDiffTendencyEnable) {
            T DiffTmp       = splat<T>(0.0);
            const Real *TrL = Tr + Lt * CStride;
            for (int J = 0; J < N; ++J) {
               const T Grad = ldk<T>(TrL, L.C1[Le * ME + J], K, Kv) - ldk<T>(TrL, L.C0[Le * ME + J], K, Kv);
               DiffTmp -= L.Df2[Le * ME + J] * ldk<T>(HMean, L.Edge[Le * ME + J], K, Kv) * Grad;
            }
            TendV += P.EddyDiff2 * DiffTmp * InvA;
         }
         if (P.TracerHyperDiffTendencyEnable) {
            T HypTmp        = splat<T>(0.0);
            const Real *D2L = Del2Tr + Lt * CStride;
            for (int J = 0; J < N; ++J) {
               const T Grad = ldk<T>(D2L, L.C1[Le * ME + J], K, Kv) - ldk<T>(D2L, L.C0[Le * ME + J], K, Kv);
               HypTmp -= L.Df4[Le * ME + J] * Grad;
            }
            TendV -= P.EddyDiff4 * HypTmp * InvA;
         }
         stk<T>(Tend + Lt * CStride, ICell, K, Kv, TendV);
      }
   }
};
void launchTracerTendOnly(const MeshView &M, int K, int NT, const TendParams &P, const AuxPtrs &A, Real *TrTend,
                          const Real *U, const Real *Tr, hipStream_t S) {
   if (NT <= 0)
      return;
   TracerTendBody B{M, K, NT, P, U, Tr, A.HTracersEdge, A.MeanLayerThickEdge, A.Del2TracersCell, TrTend};
   launchTile(B, M.NCellsAll, K, S);
}

// ---------------------------------------------------------------------------------------
// TimeStepper update kernels (components/omega/src/timeStepping/TimeStepper.cpp:378-524):
// pure streaming, 16-byte accesses, grid-stride.
template <class F> __global__ void __launch_bounds__(256) streamKernel(F Fn, size_t NVec) {
   for (size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x; I < NVec; I += (size_t)gridDim.x * blockDim.x)
      Fn(I);
}
template <class F> static void launchStream(const F &Fn, size_t NVec, hipStream_t S) {
   if (NVec == 0)
      return;
   size_t Blocks = (NVec + 255) / 256;
   if (Blocks > 8192)
      Blocks = 8192;
   hipLaunchKernelGGL((streamKernel<F>), dim3((unsigned)Blocks), dim3(256), 0, S, Fn, NVec);
}

// (for all streaming update kernels below `K` is the ROW LENGTH of the arrays = their pitch: callers pass
// Array.Pitch, so padded levels are swept too -- they hold zeros / garbage nobody reads)
// X1 = X2 + Coeff*Tend over NRows*K contiguous values (updateThicknessByTend :378-401,
// updateVelocityByTend :407-430)
template <class T> struct UpdateFn {
   Real *X1;
   const Real *X2, *Tend;
   Real Coeff;
   __device__ void operator()(size_t I) const {
      const T A = reinterpret_cast<const T *>(X2)[I], B = reinterpret_cast<const T *>(Tend)[I];
      reinterpret_cast<T *>(X1)[I] = A + Coeff * B;
   }
};
void launchUpdateByTend(int NRows, int K, Real *X1, const Real *X2, const Real *Tend, Real Coeff, hipStream_t S) {
   const size_t N = (size_t)NRows * K;
   if (N % 2 == 0)
      launchStream(UpdateFn<dv2>{X1, X2, Tend, Coeff}, N / 2, S);
   else
      launchStream(UpdateFn<double>{X1, X2, Tend, Coeff}, N, S);
}

// tracer kernels: index I runs over NRows*K (cell, level); the tracer loop is inside so the
// thickness arrays are read once for all tracers.
template <class T> struct UpdateTracersFn { // updateTracersByTend :447-469
   int NT;
   size_t Stride; // RowsSize*K / W
   Real *NextTr;
   const Real *CurTr, *H1, *H2, *TrTend;
   Real Coeff;
   __device__ void operator()(size_t I) const {
      const T Hn = reinterpret_cast<const T *>(H1)[I], Hc = reinterpret_cast<const T *>(H2)[I];
      for (int L = 0; L < NT; ++L) {
         const size_t J = L * Stride + I;
         reinterpret_cast<T *>(NextTr)[J] =
             (reinterpret_cast<const T *>(CurTr)[J] * Hc + Coeff * reinterpret_cast<const T *>(TrTend)[J]) / Hn;
      }
   }
};
void launchUpdateTracersByTend(int NT, int NRows, int RowsSize, int K, Real *NextTr, const Real *CurTr, const Real *H1,
                               const Real *H2, const Real *TrTend, Real Coeff, hipStream_t S) {
   const size_t N = (size_t)NRows * K, St = (size_t)RowsSize * K;
   if (NT <= 0)
      return;
   if (N % 2 == 0 && St % 2 == 0)
      launchStream(UpdateTracersFn<dv2>{NT, St / 2, NextTr, CurTr, H1, H2, TrTend, Coeff}, N / 2, S);
   else
      launchStream(UpdateTracersFn<double>{NT, St, NextTr, CurTr, H1, H2, TrTend, Coeff}, N, S);
}

template <class T> struct WeightTracersFn { // weightTracers :473-487
   int NT;
   size_t Stride;
   Real *NextTr;
   const Real *CurTr, *H;
   __device__ void operator()(size_t I) const {
      const T Hc = reinterpret_cast<const T *>(H)[I];
      for (int L = 0; L < NT; ++L) {
         const size_t J                   = L * Stride + I;
         reinterpret_cast<T *>(NextTr)[J] = reinterpret_cast<const T *>(CurTr)[J] * Hc;
      }
   }
};
void launchWeightTracers(int NT, int NRows, int RowsSize, int K, Real *NextTr, const Real *CurTr, const Real *HCur,
                         hipStream_t S) {
   const size_t N = (size_t)NRows * K, St = (size_t)RowsSize * K;
   if (NT <= 0)
      return;
   if (N % 2 == 0 && St % 2 == 0)
      launchStream(WeightTracersFn<dv2>{NT, St / 2, NextTr, CurTr, HCur}, N / 2, S);
   else
      launchStream(WeightTracersFn<double>{NT, St, NextTr, CurTr, HCur}, N, S);
}

template <class T> struct AccumTracersFn { // accumulateTracersUpdate :492-507
   int NT;
   size_t Stride;
   Real *Accum;
   const Real *TrTend;
   Real Coeff;
   __device__ void operator()(size_t I) const {
      for (int L = 0; L < NT; ++L) {
         const size_t J = L * Stride + I;
         reinterpret_cast<T *>(Accum)[J] += Coeff * reinterpret_cast<const T *>(TrTend)[J];
      }
   }
};
void launchAccumulateTracers(int NT, int NRows, int RowsSize, int K, Real *Accum, const Real *TrTend, Real Coeff,
                             hipStream_t S) {
   const size_t N = (size_t)NRows * K, St = (size_t)RowsSize * K;
   if (NT <= 0)
      return;
   if (N % 2 == 0 && St % 2 == 0)
      launchStream(AccumTracersFn<dv2>{NT, St / 2, Accum, TrTend, Coeff}, N / 2, S);
   else
      launchStream(AccumTracersFn<double>{NT, St, Accum, TrTend, Coeff}, N, S);
}

template <class T> struct FinalizeTracersFn { // finalizeTracersUpdate :511-524
   int NT;
   size_t Stride;
   Real *NextTr;
   const Real *H;
   __device__ void operator()(size_t I) const {
      const T Hn = reinterpret_cast<const T *>(H)[I];
      for (int L = 0; L < NT; ++L) {
         const size_t J = L * Stride + I;
         reinterpret_cast<T *>(NextTr)[J] /= Hn;
      }
   }
};
void launchFinalizeTracers(int NT, int NRows, int RowsSize, int K, Real *NextTr, const Real *HNext, hipStream_t S) {
   const size_t N = (size_t)NRows * K, St = (size_t)RowsSize * K;
   if (NT <= 0)
      return;
   if (N % 2 == 0 && St % 2 == 0)
      launchStream(FinalizeTracersFn<dv2>{NT, St / 2, NextTr, HNext}, N / 2, S);
   else
      launchStream(FinalizeTracersFn<double>{NT, St, NextTr, HNext}, N, S);
}

// ---------------------------------------------------------------------------------------
// Halo pack / unpack (components/omega/src/base/Halo.h:324-414, 566-653).  The message layout is the
// reference's: per neighbour and array Buf[(T*NList + I)*K + k] (2-D arrays: T = 0 only), compact rows of K.
// All rows of one exchange (every neighbour, every array) in one launch: buffer row j <-> row Jobs[2j+1] of the
// plane stack of piece Jobs[2j].  threadIdx.x walks the level chunks of a row (coalesced on both sides),
// threadIdx.y the rows of the workgroup.
template <class T, bool Pack> __global__ void __launch_bounds__(256)
haloCopyAllKernel(T *Buf, HaloBases B, const I4 *Jobs, size_t NRows, int KV, int Pitch, const int *Skip) {
   // (KV, Pitch: row length / row pitch in units of T)
   // Skip (unpack over the peer wire): the wire's status word; non-zero = a wait for a neighbour's message gave up, the
   // mailbox holds a stale or partial message -- the halo is left as it was and the host learns it from the status
   if (Skip && *reinterpret_cast<const volatile int *>(Skip) != 0)
      return;
   for (size_t J = (size_t)blockIdx.x * blockDim.y + threadIdx.y; J < NRows; J += (size_t)gridDim.x * blockDim.y) {
      const int Piece  = Jobs[2 * J];
      const size_t Row = (size_t)(unsigned)Jobs[2 * J + 1];
      T *A             = static_cast<T *>(B.P[Piece]) + Row * Pitch;
      T *Bf            = Buf + J * KV;
      for (int Kv = threadIdx.x; Kv < KV; Kv += blockDim.x) {
         if (Pack)
            Bf[Kv] = A[Kv];
         else
            A[Kv] = Bf[Kv];
      }
   }
}
template <class T, bool Pack>
static void launchHaloCopyT(void *Buf, const HaloBases &B, const I4 *Jobs, size_t NRows, int KV, int Pitch, hipStream_t S,
                            const int *Skip) {
   int TX = 1;
   while (TX < KV && TX < 64)
      TX *= 2;
   const int TY  = 256 / TX;
   size_t Blocks = (NRows + TY - 1) / TY;
   if (Blocks > 8192)
      Blocks = 8192;
   hipLaunchKernelGGL((haloCopyAllKernel<T, Pack>), dim3((unsigned)Blocks), dim3(TX, TY), 0, S, static_cast<T *>(Buf), B,
                      Jobs, NRows, KV, Pitch, Skip);
   HIP_CHECK(hipGetLastError());
}
template <bool Pack>
static void launchHaloCopyAll(void *Buf, const HaloBases &B, const I4 *Jobs, size_t NRows, int K, int Pitch, int ElemBytes,
                              hipStream_t S, const int *Skip = nullptr) {
   if (NRows == 0)
      return;
   // the widest unit that divides both the row length and the row pitch: 16, 8 or 4 bytes
   const size_t RowB = (size_t)K * ElemBytes, PitchB = (size_t)Pitch * ElemBytes;
   if (RowB % 16 == 0 && PitchB % 16 == 0)
      launchHaloCopyT<dv2, Pack>(Buf, B, Jobs, NRows, (int)(RowB / 16), (int)(PitchB / 16), S, Skip);
   else if (RowB % 8 == 0 && PitchB % 8 == 0)
      launchHaloCopyT<double, Pack>(Buf, B, Jobs, NRows, (int)(RowB / 8), (int)(PitchB / 8), S, Skip);
   else
      launchHaloCopyT<int, Pack>(Buf, B, Jobs, NRows, (int)(RowB / 4), (int)(PitchB / 4), S, Skip);
}
void launchHaloPackAll(void *Buf, const HaloBases &B, const I4 *Jobs, size_t NRows, int K, int Pitch, int ElemBytes,
                       hipStream_t S) {
   launchHaloCopyAll<true>(Buf, B, Jobs, NRows, K, Pitch, ElemBytes, S);
}
void launchHaloUnpackAll(const HaloBases &B, const void *Buf, const I4 *Jobs, size_t NRows, int K, int Pitch, int ElemBytes,
                         hipStream_t S, const int *SkipIfSet) {
   launchHaloCopyAll<false>(const_cast<void *>(Buf), B, Jobs, NRows, K, Pitch, ElemBytes, S, SkipIfSet);
}

// ---------------------------------------------------------------------------------------
// ManufacturedSolution (CustomTendencyTerms.cpp): one thread per (element, level); the source term
// is level independent, so it is evaluated once per element row and added to every level.
__global__ void manufacturedThicknessKernel(int N, int K, int Pitch, Real *Tend, const Real *XCell, const Real *YCell,
                                            ManufacturedParams P, Real T) {
   const int I = blockIdx.x * blockDim.y + threadIdx.y;
   if (I >= N)
      return;
   const Real Phase = P.Kx * XCell[I] + P.Ky * YCell[I] - P.AngFreq * T;
   const Real Src   = P.Eta0 * (-P.H0 * (P.Kx + P.Ky) * sin(Phase) - P.AngFreq * cos(Phase) +
                              P.Eta0 * (P.Kx + P.Ky) * cos(2.0 * Phase)); // :139-142
   for (int Kl = threadIdx.x; Kl < K; Kl += blockDim.x)
      Tend[(size_t)I * Pitch + Kl] += Src;
}
__global__ void manufacturedVelocityKernel(int N, int K, int Pitch, Real *Tend, const Real *XEdge, const Real *YEdge,
                                           const Real *FEdge, const Real *AngleEdge, ManufacturedParams P, Real T) {
   const int I = blockIdx.x * blockDim.y + threadIdx.y;
   if (I >= N)
      return;
   const Real Kx2 = P.Kx * P.Kx, Ky2 = P.Ky * P.Ky, Kx4 = Kx2 * Kx2, Ky4 = Ky2 * Ky2;
   const Real Phase       = P.Kx * XEdge[I] + P.Ky * YEdge[I] - P.AngFreq * T;
   const Real SourceTerm0 = P.AngFreq * sin(Phase) - 0.5 * P.Eta0 * (P.Kx + P.Ky) * sin(2.0 * Phase);
   Real U = P.Eta0 * ((-FEdge[I] + P.Grav * P.Kx) * cos(Phase) + SourceTerm0);
   Real V = P.Eta0 * ((FEdge[I] + P.Grav * P.Ky) * cos(Phase) + SourceTerm0);
   if (P.VelDiffTendencyEnable) { // :188-191
      U += P.ViscDel2 * P.Eta0 * (Kx2 + Ky2) * cos(Phase);
      V += P.ViscDel2 * P.Eta0 * (Kx2 + Ky2) * cos(Phase);
   }
   if (P.VelHyperDiffTendencyEnable) { // :192-197
      U -= P.ViscDel4 * P.Eta0 * ((Kx4 + Ky4 + Kx2 * Ky2) * cos(Phase));
      V -= P.ViscDel4 * P.Eta0 * ((Kx4 + Ky4 + Kx2 * Ky2) * cos(Phase));
   }
   const Real Src = cos(AngleEdge[I]) * U + sin(AngleEdge[I]) * V;
   for (int Kl = threadIdx.x; Kl < K; Kl += blockDim.x)
      Tend[(size_t)I * Pitch + Kl] += Src;
}
static dim3 rowBlock(int K) {
   int TX = 1;
   while (TX < K && TX < 64)
      TX <<= 1;
   return dim3(TX, 256 / TX, 1);
}
void launchManufacturedThickness(int N, int K, Real *Tend, const Real *XCell, const Real *YCell,
                                 const ManufacturedParams &P, Real T, hipStream_t S) {
   if (N <= 0)
      return;
   const dim3 B = rowBlock(K);
   hipLaunchKernelGGL(manufacturedThicknessKernel, dim3((N + B.y - 1) / B.y), B, 0, S, N, K, levelPitch(K), Tend, XCell, YCell, P, T);
   HIP_CHECK(hipGetLastError());
}
void launchManufacturedVelocity(int N, int K, Real *Tend, const Real *XEdge, const Real *YEdge, const Real *FEdge,
                                const Real *AngleEdge, const ManufacturedParams &P, Real T, hipStream_t S) {
   if (N <= 0)
      return;
   const dim3 B = rowBlock(K);
   hipLaunchKernelGGL(manufacturedVelocityKernel, dim3((N + B.y - 1) / B.y), B, 0, S, N, K, levelPitch(K), Tend, XEdge, YEdge, FEdge,
                      AngleEdge, P, T);
   HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------
// Reproducible sums (reference: components/omega/src/base/Reductions.h).  The reference's host path
// accumulates in double-double (Knuth) and combines ranks with the ddSum MPI operator; its device path
// falls back to a plain parallelReduce.  Here the device path keeps the double-double arithmetic:
// (hi, lo) per thread, combined with the same ddSum expression across the workgroup and across
// workgroups, so the result does not depend on the launch geometry or the partition to within
// double-double rounding (~1e-32 relative).
struct DD {
   double Hi, Lo;
};
__host__ __device__ inline DD ddAdd(DD A, DD B) { // ddSum, Reductions.h:24-35
   const double T1 = A.Hi + B.Hi;
   const double E  = T1 - A.Hi;
   const double T2 = ((B.Hi - E) + (A.Hi - (T1 - E))) + A.Lo + B.Lo;
   DD R;
   R.Hi = T1 + T2;
   R.Lo = T2 - ((T1 + T2) - T1);
   return R;
}
__host__ __device__ inline DD ddAddScalar(DD S, double Ai) { // Knuth accumulation, Reductions.h:171-179
   const double T1 = Ai + S.Hi;
   const double E  = T1 - Ai;
   const double T2 = ((S.Hi - E) + (Ai - (T1 - E))) + S.Lo;
   DD R;
   R.Hi = T1 + T2;
   R.Lo = T2 - ((T1 + T2) - T1);
   return R;
}
__device__ inline DD blockReduceDD(DD V) {
   __shared__ DD Sh[256];
   const int T = threadIdx.x;
   Sh[T]       = V;
   __syncthreads();
   for (int Off = 128; Off > 0; Off >>= 1) {
      if (T < Off)
         Sh[T] = ddAdd(Sh[T], Sh[T + Off]);
      __syncthreads();
   }
   return Sh[0];
}
__global__ void __launch_bounds__(256) sumDDKernel(const Real *A, const Real *B, size_t N, DD *Partial) {
   DD Acc{0.0, 0.0};
   for (size_t I = (size_t)blockIdx.x * 256 + threadIdx.x; I < N; I += (size_t)gridDim.x * 256)
      Acc = ddAddScalar(Acc, B ? A[I] * B[I] : A[I]);
   const DD R = blockReduceDD(Acc);
   if (threadIdx.x == 0)
      Partial[blockIdx.x] = R;
}
__global__ void __launch_bounds__(256) weightedSumDDKernel(const Real *W, const Real *A, const Real *B, int NRows, int K,
                                                           int Pitch, DD *Partial) {
   DD Acc{0.0, 0.0};
   const size_t N = (size_t)NRows * K;
   for (size_t I = (size_t)blockIdx.x * 256 + threadIdx.x; I < N; I += (size_t)gridDim.x * 256) {
      const size_t R = I / K, J = R * Pitch + (I - R * K);
      const double V = B ? A[J] * B[J] : A[J];
      Acc            = ddAddScalar(Acc, W[R] * V);
   }
   const DD R = blockReduceDD(Acc);
   if (threadIdx.x == 0)
      Partial[blockIdx.x] = R;
}
static void finishDD(DD *PartialD, int NB, hipStream_t S, double HiLo[2]) {
   std::vector<DD> H(NB);
   HIP_CHECK(hipMemcpyAsync(H.data(), PartialD, NB * sizeof(DD), hipMemcpyDeviceToHost, S));
   HIP_CHECK(hipStreamSynchronize(S));
   DD Acc{0.0, 0.0};
   for (int I = 0; I < NB; ++I)
      Acc = ddAdd(Acc, H[I]);
   HiLo[0] = Acc.Hi, HiLo[1] = Acc.Lo;
   HIP_CHECK(hipFree(PartialD));
}
void localSumDD(const Real *A, const Real *B, size_t N, hipStream_t S, double HiLo[2]) {
   const int NB = (int)std::min<size_t>(1024, (N + 255) / 256 ? (N + 255) / 256 : 1);
   DD *PartialD = nullptr;
   HIP_CHECK(hipMalloc(&PartialD, NB * sizeof(DD)));
   hipLaunchKernelGGL(sumDDKernel, dim3(NB), dim3(256), 0, S, A, B, N, PartialD);
   HIP_CHECK(hipGetLastError());
   finishDD(PartialD, NB, S, HiLo);
}
void localWeightedSumDD(const Real *W, const Real *A, const Real *B, int NRows, int K, int Pitch, hipStream_t S,
                        double HiLo[2]) {
   const size_t N = (size_t)NRows * K;
   const int NB   = (int)std::min<size_t>(1024, (N + 255) / 256 ? (N + 255) / 256 : 1);
   DD *PartialD   = nullptr;
   HIP_CHECK(hipMalloc(&PartialD, NB * sizeof(DD)));
   hipLaunchKernelGGL(weightedSumDDKernel, dim3(NB), dim3(256), 0, S, W, A, B, NRows, K, Pitch, PartialD);
   HIP_CHECK(hipGetLastError());
   finishDD(PartialD, NB, S, HiLo);
}
/// host-side ddSum over an array of (hi, lo) pairs, in order (the combination across ranks)
void combineDD(const double *Pairs, int NPairs, double HiLo[2]) {
   DD Acc{0.0, 0.0};
   for (int I = 0; I < NPairs; ++I)
      Acc = ddAdd(Acc, DD{Pairs[2 * I], Pairs[2 * I + 1]});
   HiLo[0] = Acc.Hi, HiLo[1] = Acc.Lo;
}

} // namespace OMEGA
